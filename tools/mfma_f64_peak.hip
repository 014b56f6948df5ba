// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (calibrates the fp64 matrix peak used
// as roofline denominator in bench.py).  hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a, double b) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int threads, const char* name) {
  double* out; hipMalloc(&out, sizeof(double) * blocks * threads);
  int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001, 0.9999999);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waves = (double)blocks * threads / 64;
  double flops = waves * iters * NACC * 2.0 * 16 * 16 * 4;
  printf("%-28s blocks=%d threads=%d  %.3f ms  %.2f TFLOP/s\n", name, blocks, threads, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(256, 256, "1 wave/SIMD, 4 acc");
  run<8>(256, 256, "1 wave/SIMD, 8 acc");
  run<4>(512, 256, "2 waves/SIMD, 4 acc");
  run<4>(1024, 256, "4 waves/SIMD, 4 acc");
  run<1>(1024, 256, "4 waves/SIMD, 1 acc");
  run<2>(256, 256, "1 wave/SIMD, 2 acc");
  return 0;
}
