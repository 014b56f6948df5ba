// Developer harness (not shipped): the chain's 64 x 64 diagonal-block factorisation in isolation -- the blocked panels of round 4
// (chain_factor_blocked: one wavefront pivots in registers, three update with MFMA and build the inverse), checked against a
// long-double LDL^T on the host and timed with the 100 MHz wall clock inside one workgroup, phase by phase.  (The
// barrier-per-pivot-pair loop of round 3 it replaced measured 17.9 us per block in this harness against 9.9: profiles/r04_diag.txt.)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form tools/bench_diag.hip -o tools/bin/bench_diag
#define CBA_DEV_SWITCHES 1
#define CBA_DIAGLOG 1
#include "../camera_calibration_amd/csrc/kernels_linalg.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
namespace cba { void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); } }
using namespace cba;

__global__ void __launch_bounds__(256) k_diag_test(const double* __restrict__ Tin, double* __restrict__ Lout, double* __restrict__ Iout,
                                                   double* __restrict__ rdout, int iters, unsigned long long* __restrict__ ticks, int mode,
                                                   int* __restrict__ status) {
  __shared__ double smem[3 * kInner * TS];
  double* sV = smem;
  double* sW = smem + kInner * TS;
  double* sT = smem + 2 * kInner * TS;
  double* s_rd = sW + kInner;
  const int tid = threadIdx.x;
  for (int e = tid; e < kInner * kInner; e += 256) {
    const int m = e >> 6, n = e & 63;
    sT[m * TS + n] = (n >= m) ? Tin[m * kInner + n] : 1.0e3 + 0.37 * m + n;      // junk below the diagonal: must not matter
  }
  __syncthreads();
  unsigned long long total = 0;
  bool bad_any = false;
  for (int it = 0; it < iters; ++it) {
    for (int e = tid; e < kInner * TS; e += 256) { sW[e] = (e % TS) < kInner ? sT[e] : sW[e]; sV[e] = -7.0; }
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    bool bad;
    bad = chain_factor_blocked(sW, sV, s_rd);
    total += wall_clock64() - t0;
    bad_any |= bad;
    __syncthreads();
  }
  if (tid == 0) { ticks[0] = total; if (bad_any) atomicExch(status, 2); }
  for (int e = tid; e < kInner * kInner; e += 256) {
    const int m = e >> 6, n = e & 63;
    Lout[e] = (mode == 0 && n < m) ? 0.0 : sV[m * TS + n];                              // (the chain's store masks the same way)
    Iout[e] = (mode == 0 && (n >> 4) < (m >> 4)) ? 0.0 : sW[m * TS + n];
  }
  if (tid < kInner) rdout[tid] = s_rd[(tid >> 4) * TS + (tid & 15)];
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int n = kInner;
  std::vector<double> T(n * n);
  // symmetric, indefinite-safe: A A^T scaled + a diagonal, like a damped Schur block
  std::vector<double> A(n * n);
  for (int i = 0; i < n * n; ++i) A[i] = ((double)((i * 2654435761u) % 2001) / 1000.0 - 1.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0; for (int k = 0; k < n; ++k) s += A[i * n + k] * A[j * n + k];
      T[i * n + j] = 0.05 * s + (i == j ? 3.0 : 0.0);
    }
  // reference: long-double LDL^T and the inverse of the unit factor
  std::vector<long double> L(n * n, 0.0L), d(n), W(n * n), M(n * n, 0.0L);
  for (int i = 0; i < n * n; ++i) W[i] = T[i];
  for (int s = 0; s < n; ++s) {
    d[s] = W[s * n + s];
    for (int i = s + 1; i < n; ++i) L[i * n + s] = W[i * n + s] / d[s];
    for (int i = s + 1; i < n; ++i) for (int j = s + 1; j < n; ++j) W[i * n + j] -= L[i * n + s] * d[s] * L[j * n + s];
    L[s * n + s] = 1.0L;
  }
  for (int j = 0; j < n; ++j) {           // column j of L^-1
    M[j * n + j] = 1.0L;
    for (int i = j + 1; i < n; ++i) { long double s = 0; for (int k = j; k < i; ++k) s += L[i * n + k] * M[k * n + j]; M[i * n + j] = -s; }
  }
  double *dT, *dL, *dI, *dr; unsigned long long* dt; int* dst;
  hipMalloc(&dT, 8 * n * n); hipMalloc(&dL, 8 * n * n); hipMalloc(&dI, 8 * n * n); hipMalloc(&dr, 8 * n); hipMalloc(&dt, 8); hipMalloc(&dst, 4);
  hipMemcpy(dT, T.data(), 8 * n * n, hipMemcpyHostToDevice);
  unsigned long long* dlog; hipMalloc(&dlog, 8 * 32);
  for (int mode = 0; mode < 1; ++mode) {
    hipMemset(dst, 0, 4);
    for (int rep = 0; rep < 2; ++rep) {
      // second repetition of the blocked variant with the phase stamps on
      unsigned long long* lp = (mode == 0 && rep == 1) ? dlog : nullptr;
      hipMemset(dlog, 0, 8 * 32);
      hipMemcpyToSymbol(HIP_SYMBOL(g_diaglog), &lp, sizeof(lp));
      hipLaunchKernelGGL(k_diag_test, dim3(1), dim3(256), 0, 0, dT, dL, dI, dr, iters, dt, mode, dst);
      hipDeviceSynchronize();
    }
    if (mode == 0) {
      unsigned long long hl[32]; hipMemcpy(hl, dlog, 8 * 32, hipMemcpyDeviceToHost);
      auto us = [&](int a, int b) { return (double)(hl[b] - hl[a]) / 100.0 / iters; };
      printf("   phases (us, barrier to barrier): A0 %.2f (panel %.2f) | C0 %.2f | A1 %.2f (panel %.2f) | C1 %.2f | A2 %.2f (panel %.2f) | C2 %.2f | A3 %.2f (panel %.2f, + M_33 %.2f) | E1 %.2f | E2 %.2f\n",
             us(0, 1), us(0, 10), us(1, 2), us(2, 3), us(2, 11), us(3, 4), us(4, 5), us(4, 12), us(5, 6), us(6, 7), us(6, 13), us(13, 14), us(7, 8), us(8, 9));
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d: launch failed: %s\n", mode, hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<double> hL(n * n), hI(n * n), hr(n); unsigned long long ht; int hs;
    hipMemcpy(hL.data(), dL, 8 * n * n, hipMemcpyDeviceToHost); hipMemcpy(hI.data(), dI, 8 * n * n, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), dr, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost); hipMemcpy(&hs, dst, 4, hipMemcpyDeviceToHost);
    double eL = 0, eD = 0, eI = 0, eR = 0, eZ = 0, mI = 0;
    bool nan = false;
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        const double vl = hL[j * n + i], vi = hI[j * n + i];        // [j][i]: L(i, j) / d ; [q = j][p = i]: M(i, j)
        if (!(vl == vl) || !(vi == vi)) nan = true;
        if (i > j) eL = std::max(eL, (double)fabsl(vl - L[i * n + j]));
        else if (i == j) eD = std::max(eD, (double)fabsl((vl - d[j]) / d[j]));
        else eZ = std::max(eZ, std::fabs(vl));
        if (i >= j) { eI = std::max(eI, (double)fabsl(vi - M[i * n + j])); mI = std::max(mI, (double)fabsl(M[i * n + j])); }
        else eZ = std::max(eZ, std::fabs(vi));
      }
    for (int i = 0; i < n; ++i) eR = std::max(eR, (double)fabsl(hr[i] * d[i] - 1.0L));
    printf("%s: %.3f us per block (%d blocks) | status %d%s | L abs err %.2e  d rel %.2e  invL abs %.2e (max %.2f)  1/d rel %.2e  below-diagonal max %.1e\n",
           mode == 0 ? "blocked panels (round 4)" : "pivot pairs   (round 3)", (double)ht / 100.0 / iters, iters, hs, nan ? "  NaN!" : "",
           eL, eD, eI, mI, eR, eZ);
  }
  return 0;
}
