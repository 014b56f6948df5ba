// Developer harness: host-side cost of hipMemsetAsync vs a kernel launch (what the GPU waits for after a host decision).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void k_clear(double* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d; hipMalloc(&d, 4096);
  double* big; const size_t nb = (size_t)12672 * 12672; hipMalloc(&big, nb * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipStreamSynchronize(s);
    double t0 = now();
    for (int i = 0; i < 200; ++i) hipMemsetAsync(d, 0, 4, s);
    double t1 = now(); hipStreamSynchronize(s); double t2 = now();
    printf("hipMemsetAsync(4 B): host %.1f us per call (drain %.1f us per call)\n", (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6);
    t0 = now();
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, d);
    t1 = now(); hipStreamSynchronize(s); t2 = now();
    printf("empty kernel launch: host %.1f us per call (drain %.1f us per call)\n", (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6);
    t0 = now();
    for (int i = 0; i < 20; ++i) hipMemsetAsync(big, 0, nb * 8, s);
    t1 = now(); hipStreamSynchronize(s); t2 = now();
    printf("hipMemsetAsync(1.28 GB): host %.1f us per call, %.3f ms per call = %.2f TB/s\n", (t1 - t0) / 20 * 1e6, (t2 - t0) / 20 * 1e3, nb * 8 / ((t2 - t0) / 20) / 1e12);
    t0 = now();
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_clear, dim3(4096), dim3(256), 0, s, big, nb);
    t1 = now(); hipStreamSynchronize(s); t2 = now();
    printf("k_clear(1.28 GB, 8 B per lane): host %.1f us per call, %.3f ms per call = %.2f TB/s\n", (t1 - t0) / 20 * 1e6, (t2 - t0) / 20 * 1e3, nb * 8 / ((t2 - t0) / 20) / 1e12);
  }
  return 0;
}
