// Micro-benchmark: throughput of fp64 atomic adds to scattered addresses, device scope vs workgroup scope
// (the latter is only correct when a single XCD touches an address during the kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SCOPE>
__global__ void __launch_bounds__(256) k(double* buf, size_t n_mask, int iters, int xcd_local) {
  unsigned lane_id = blockIdx.x * 256 + threadIdx.x;
  unsigned long long st = lane_id * 0x9E3779B97F4A7C15ull + 12345;
  unsigned xcd = blockIdx.x & 7;
  for (int i = 0; i < iters; ++i) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    size_t idx = (st >> 20) & n_mask;
    // groups of 8 consecutive doubles (one 64-B segment) like the grid columns of one cell row
    idx = (idx & ~(size_t)7) | (threadIdx.x & 7);
    if (xcd_local) idx = (idx & ~((size_t)7 << 10)) | ((size_t)xcd << 10);   // partition the address space by XCD
    if (SCOPE == 0) unsafeAtomicAdd(buf + idx, 1.0);
    else __hip_atomic_fetch_add(buf + idx, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
int main() {
  size_t n = (size_t)1 << 27;  // 1 GiB of doubles
  double* buf; hipMalloc(&buf, n * 8); hipMemset(buf, 0, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = 256 * 8, iters = 512;
  for (int local = 0; local < 2; ++local)
    for (int scope = 0; scope < 2; ++scope) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (scope == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, n - 1, iters, local);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, n - 1, iters, local);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("xcd_local=%d scope=%s: %.3f ms  %.1f G atomics/s\n", local, scope ? "workgroup" : "agent", ms, (double)blocks * 256 * iters / ms / 1e6);
      }
    }
  return 0;
}
