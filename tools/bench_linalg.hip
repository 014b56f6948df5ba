// Developer harness (not shipped): times the linear-algebra kernels of the engine in isolation.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form tools/bench_linalg.hip -o tools/bin/bench_linalg
#define CBA_DEV_SWITCHES 1   // epilogue modes + CBA_* environment switches exist only in this harness
#include "../camera_calibration_amd/csrc/kernels_linalg.hip"
#include <cstdio>
#include <chrono>
#include <vector>
#include <cmath>
#include <algorithm>
namespace cba { void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); } }
using namespace cba;
namespace cba {

int ldlt_back_solve(const double* S, int n_fact, int ld, int zcol, const LdltWorkspace& w, double* x, hipStream_t s);
}
static float timeit(hipEvent_t e0, hipEvent_t e1) { float ms; hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms; }
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 12544;
  int K = argc > 2 ? atoi(argv[2]) : 3008;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // SPD matrix S = G^T G / K + I built on the device with the Schur GEMM itself
  double *A, *S, *H;
  hipMalloc(&A, sizeof(double) * (size_t)K * n); hipMalloc(&S, sizeof(double) * (size_t)n * n); hipMalloc(&H, sizeof(double) * (size_t)n * n);
  std::vector<double> hA((size_t)K * n);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((double)((i * 2654435761u) % 2001) / 1000.0 - 1.0) * 0.05;
  hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
  hipMemset(H, 0, sizeof(double) * (size_t)n * n);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    schur_gemm(A, A, K, n, H, S, n, n, n - 1, 1, -1.0 * K * 0.001, nullptr, nullptr);  // S = -lambda' I - A^T A  (negative definite: fine for LDL^T)
    hipEventRecord(e1);
    float ms = timeit(e0, e1);
    double nt = n / 128.0, tiles = nt * (nt + 1) / 2;
    printf("schur_gemm n=%d K=%d: %.3f ms  %.2f TFLOP/s\n", n, K, ms, tiles * 2.0 * 128 * 128 * K / ms / 1e9);
  }
  // trailing-update shaped GEMMs (C -= A^T B, upper tiles) for several K
  for (int mode : {0, 1, 2, 0}) for (int Kt : {16, 256, 512}) {
    GemmArgs u{}; u.epi_mode = mode;
    u.A = A; u.lda = n; u.B = A; u.ldb = n; u.K = Kt; u.C = S; u.ldc = n; u.Cin = S; u.ldcin = n;
    u.m_off = 0; u.m_tiles = n / 128; u.n_off = 0; u.n_tiles = n / 128; u.upper = 1; u.diag = 0;
    launch_gemm<128, 128, 64, 64, true>(u, nullptr);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch_gemm<128, 128, 64, 64, true>(u, nullptr);
    hipEventRecord(e1);
    float ms = timeit(e0, e1) / 5;
    double nt = n / 128.0, tiles = nt * (nt + 1) / 2;
    printf("trailing-shaped gemm mode=%d n=%d K=%d: %.3f ms  %.2f TFLOP/s\n", mode, n, Kt, ms, tiles * 2.0 * 128 * 128 * Kt / ms / 1e9);
  }
  // the same with distinct operand panels (what the factorisation's updates look like: A = L in place, B = panel buffer)
  for (int nn : {n, 10880, 8832, 6144}) for (int Kt : {256, 512}) {
    if (nn > n || K < 1504 + Kt) continue;
    GemmArgs u{};
    u.A = A; u.lda = n; u.B = A + (size_t)1504 * n; u.ldb = n; u.K = Kt; u.C = S; u.ldc = n; u.Cin = S; u.ldcin = n;
    u.m_off = 0; u.m_tiles = nn / 128; u.n_off = 0; u.n_tiles = nn / 128; u.upper = 1; u.diag = 0;
    launch_gemm<128, 128, 64, 64, true>(u, nullptr);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch_gemm<128, 128, 64, 64, true>(u, nullptr);
    hipEventRecord(e1);
    float ms = timeit(e0, e1) / 5;
    double nt = nn / 128, tiles = nt * (nt + 1) / 2;
    printf("distinct-operand gemm n=%d K=%d: %.3f ms  %.2f TFLOP/s\n", nn, Kt, ms, tiles * 2.0 * 128 * 128 * Kt / ms / 1e9);
  }
  {  // ... and on the engine's main stream (CU mask without the CUs reserved for the pivot chain)
    hipStream_t msk; make_main_stream(&msk);
    for (int nn : {10880, 6144}) for (int Kt : {256, 512}) {
      if (nn > n || K < 1504 + Kt) continue;
      GemmArgs u{};
      u.A = A; u.lda = n; u.B = A + (size_t)1504 * n; u.ldb = n; u.K = Kt; u.C = S; u.ldc = n; u.Cin = S; u.ldcin = n;
      u.m_off = 0; u.m_tiles = nn / 128; u.n_off = 0; u.n_tiles = nn / 128; u.upper = 1; u.diag = 0;
      launch_gemm<128, 128, 64, 64, true>(u, msk);
      hipEventRecord(e0, msk);
      for (int r = 0; r < 5; ++r) launch_gemm<128, 128, 64, 64, true>(u, msk);
      hipEventRecord(e1, msk);
      float ms = timeit(e0, e1) / 5;
      double nt = nn / 128, tiles = nt * (nt + 1) / 2;
      printf("distinct-operand gemm on the masked main stream n=%d K=%d: %.3f ms  %.2f TFLOP/s\n", nn, Kt, ms, tiles * 2.0 * 128 * 128 * Kt / ms / 1e9);
    }

  }
  LdltWorkspace w; ldlt_workspace_alloc(w, n);
  hipMemset(w.status, 0, 4);
  // diag kernel alone, 200 launches on the first block of a scratch copy
  double* T; hipMalloc(&T, sizeof(double) * (size_t)64 * n);
  hipMemcpy(T, S, sizeof(double) * (size_t)64 * n, hipMemcpyDeviceToDevice);
  hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_ldlt_diag<64>, dim3(1), dim3(256), 0, 0, T, n, 0, w.dvec, w.invLt, w.status);
  hipEventRecord(e1);
  printf("k_ldlt_diag<64>: %.2f us per launch (200 back-to-back)\n", timeit(e0, e1) * 1000 / 200);
  hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_ldlt_diag<0>, dim3(1), dim3(256), 0, 0, T, n, 0, w.dvec, w.invLt, w.status);
  hipEventRecord(e1);
  printf("k_ldlt_diag<0> (load/store only): %.2f us per launch\n", timeit(e0, e1) * 1000 / 200);
#define TIME_DIAG(NS) \
  hipMemcpy(T, S, sizeof(double) * (size_t)64 * n, hipMemcpyDeviceToDevice); \
  hipEventRecord(e0); \
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_ldlt_diag<NS>, dim3(1), dim3(256), 0, 0, T, n, 0, w.dvec, w.invLt, w.status); \
  hipEventRecord(e1); \
  printf("k_ldlt_diag<%d>: %.2f us per launch\n", NS, timeit(e0, e1) * 1000 / 200);
  TIME_DIAG(8) TIME_DIAG(16) TIME_DIAG(17) TIME_DIAG(24) TIME_DIAG(32) TIME_DIAG(48) TIME_DIAG(64)
  hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_ldlt_diag<64>, dim3(256), dim3(256), 0, 0, T, n, 0, w.dvec, w.invLt, w.status);
  hipEventRecord(e1);
  printf("k_ldlt_diag<64> x256 blocks (same data, clock test): %.2f us per launch\n", timeit(e0, e1) * 1000 / 200);
  int n_fact = n - 128;
  hipStream_t ms; make_main_stream(&ms);
  printf("panel CUs reserved: %d\n", panel_cu_count());
  for (int rep = 0; rep < 2; ++rep) {
    schur_gemm(A, A, K, n, H, S, n, n, n - 1, 1, -1.0 * K * 0.001, nullptr, nullptr);
    hipDeviceSynchronize();
    GemmStats gs;
    hipMemset(w.status, 0, 4);
#ifdef CBA_TLOG
    const int n_tags = (n / 64 + 1) * kTlKinds;
    static unsigned long long* tl = nullptr;
    if (!tl) { hipMalloc(&tl, sizeof(unsigned long long) * 2 * n_tags); hipMemcpyToSymbol(HIP_SYMBOL(g_tlog), &tl, sizeof(tl)); }
    {
      std::vector<unsigned long long> init(2 * n_tags);
      for (int i = 0; i < n_tags; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
      hipMemcpy(tl, init.data(), sizeof(unsigned long long) * 2 * n_tags, hipMemcpyHostToDevice);
    }
#endif
    hipEventRecord(e0, ms);
    auto h0 = std::chrono::steady_clock::now();
    ldlt_factor(S, n_fact, n, w, ms, &gs);
    auto h1 = std::chrono::steady_clock::now();
    hipEventRecord(e1, ms);
    float ms_ = timeit(e0, e1);
    printf("host enqueue time of ldlt_factor: %.3f ms\n", std::chrono::duration<double, std::milli>(h1 - h0).count());
    printf("ldlt_factor n_fact=%d: %.3f ms  (trailing %.3f TFLOP -> %.2f TFLOP/s overall)\n", n_fact, ms_, gs.flops / 1e12, gs.flops / ms_ / 1e9);
#ifdef CBA_TLOG
    if (rep == 1) {
      std::vector<unsigned long long> lg(2 * n_tags);
      hipMemcpy(lg.data(), tl, sizeof(unsigned long long) * 2 * n_tags, hipMemcpyDeviceToHost);
      static const char* names[kTlKinds] = {"diag", "near", "scale", "mid_trsm", "mid_upd", "chain_trsm", "chain_upd", "a'", "panel_solve",
                                            "a''n", "a''rest", "bulk", "Xn", "", "", ""};
      const char* pe = getenv("TL_PANELS");       // e.g. "0,6656,9216,11776": first column of the panels to print
      std::vector<int> want;
      if (pe) { for (const char* c = pe; *c;) { want.push_back(atoi(c)); while (*c && *c != ',') ++c; if (*c) ++c; } }
      else want = {0, 4096, 6656, 9216, 11776};
      // panel spans (diag of the panel's first block to the next panel's first diag)
      printf("panel starts (us since first diag), width, span:\n");
      const double t00 = (double)lg[0];
      for (int k0 = 0; k0 < n_fact;) {
        const int pw = panel_width_at(k0, n_fact);
        const int nxt = k0 + pw;
        const double a = (lg[2 * ((k0 / 64) * kTlKinds)] - t00) / 100.0;
        const double b = nxt < n_fact ? (lg[2 * ((nxt / 64) * kTlKinds)] - t00) / 100.0 : NAN;
        printf("  panel %6d w %3d start %9.1f span %8.1f\n", k0, pw, a, b - a);
        k0 = nxt;
      }
      for (int k0 : want) {
        if (k0 >= n_fact) continue;
        const int pw = panel_width_at(k0, n_fact);
        struct Ev { double a, b; int blk, kind; };
        std::vector<Ev> evs;
        const double t0 = (double)lg[2 * ((k0 / 64) * kTlKinds)];
        for (int blk = k0 / 64; blk < (k0 + pw) / 64 + 1 && blk * kTlKinds < n_tags; ++blk)
          for (int kd = 0; kd < kTlKinds; ++kd) {
            const unsigned long long a = lg[2 * (blk * kTlKinds + kd)], b = lg[2 * (blk * kTlKinds + kd) + 1];
            if (a == ~0ull || b == 0) continue;
            if (blk == (k0 + pw) / 64 && kd != kTlDiag) continue;
            evs.push_back({(a - t0) / 100.0, (b - t0) / 100.0, blk, kd});
          }
        std::sort(evs.begin(), evs.end(), [](const Ev& x, const Ev& y) { return x.a < y.a; });
        printf("timeline of panel at column %d (width %d), us relative to its first diag:\n", k0, pw);
        for (const Ev& e : evs) printf("  %8.1f %8.1f  dur %7.1f  blk %3d  %s\n", e.a, e.b, e.b - e.a, e.blk, names[e.kind]);
      }
    }
#endif
    double* x; hipMalloc(&x, sizeof(double) * n);
    hipEventRecord(e0);
    ldlt_back_solve(S, n_fact, n, n - 1, w, x, nullptr);
    hipEventRecord(e1);
    printf("ldlt_back_solve: %.3f ms\n", timeit(e0, e1));
    hipFree(x);
  }
  int st; hipMemcpy(&st, w.status, 4, hipMemcpyDeviceToHost);
  printf("status %d\n", st);
  return 0;
}
