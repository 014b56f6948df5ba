// Developer harness (not shipped): the 128 x 128 fp64 MFMA GEMM alone, at the shapes of the super-panel updates.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form tools/bench_gemm.hip -o tools/bin/bench_gemm
#define CBA_DEV_SWITCHES 1
#include "../camera_calibration_amd/csrc/kernels_linalg.hip"
#include <cstdio>
#include <vector>
namespace cba { void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); } }
using namespace cba;
int main() {
  const int n = 12672, K = 2304;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double *A, *B, *S;
  hipMalloc(&A, sizeof(double) * (size_t)K * n); hipMalloc(&B, sizeof(double) * (size_t)K * n); hipMalloc(&S, sizeof(double) * (size_t)n * n);
  std::vector<double> h((size_t)K * n);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((double)((i * 2654435761u) % 2001) / 1000.0 - 1.0) * 0.05;
  hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((double)((i * 40503u + 7) % 1999) / 1000.0 - 1.0) * 0.05;
  hipMemcpy(B, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMemset(S, 0, sizeof(double) * (size_t)n * n);
  const int ms_[3] = {84, 71, 54}, ks_[3] = {1920, 1664, 2176};
  for (int rep = 0; rep < 2; ++rep)
    for (int c = 0; c < 3; ++c) {
      GemmArgs u{};
      const int off = n - ms_[c] * 128;
      u.A = A; u.lda = n; u.B = B; u.ldb = n; u.K = ks_[c]; u.C = S; u.ldc = n; u.Cin = S; u.ldcin = n;
      u.m_off = off; u.m_tiles = ms_[c]; u.n_off = off; u.n_tiles = ms_[c]; u.upper = 1; u.diag = 0;
      launch_gemm<128, 128, 64, 64, true>(u, nullptr);
      hipEventRecord(e0);
      for (int r = 0; r < 4; ++r) launch_gemm<128, 128, 64, 64, true>(u, nullptr);
      hipEventRecord(e1);
      float t; hipEventSynchronize(e1); hipEventElapsedTime(&t, e0, e1); t /= 4;
      const double tiles = ms_[c] * (ms_[c] + 1) / 2.0;
      printf("upper update m = %d tiles, K = %d: %.3f ms  %.2f TFLOP/s\n", ms_[c], ks_[c], t, tiles * 2.0 * 128 * 128 * ks_[c] / t / 1e9);
    }
  // checksum so that variants can be compared
  std::vector<double> row(n); hipMemcpy(row.data(), S + (size_t)(n - 200) * n, sizeof(double) * n, hipMemcpyDeviceToHost);
  double cs = 0; for (int i = 0; i < n; ++i) cs += row[i] * (1 + i % 7);
  printf("checksum %.12e\n", cs);
  return 0;
}
