// Developer harness (not shipped): the three launches of the grid-first factorisation on a synthetic matrix with the structure of
// BASELINE configs[1] (banded grid part by the plan, dense border), without the observation stages: per-launch times, residual of the
// solution on the host, chain timelines of the block-sparse launch and of the border launch (TAILLOG).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -w tools/bench_gridfirst.hip \
//         camera_calibration_amd/csrc/gridfirst_plan.hip -o tools/bin/bench_gridfirst
//   tools/bin/bench_gridfirst [strips] [gw gh N P]        env: TAILLOG=1 REPS=n
#define CBA_DEV_SWITCHES 1
#define CBA_TAILLOG 1
#include "../camera_calibration_amd/csrc/kernels_linalg.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
namespace cba { void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); } }
using namespace cba;

static float timeit(hipEvent_t e0, hipEvent_t e1) { float ms; hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms; }
static inline double rnd(size_t i) { return (double)((i * 2654435761ull) % 2001ull) / 1000.0 - 1.0; }

int main(int argc, char** argv) {
  const int strips = argc > 1 ? atoi(argv[1]) : 2;
  const int gw = argc > 5 ? atoi(argv[2]) : 84, gh = argc > 5 ? atoi(argv[3]) : 60, N = argc > 5 ? atoi(argv[4]) : 500, P = argc > 5 ? atoi(argv[5]) : 815;
  const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 3;
  cba_camera cam{CBA_CENTRAL_GENERIC, 2048, 1456, 0, 0, 2047, 1455, gw, gh};
  GfPlan pl;
  const int single = getenv("SINGLE_TILES") ? 1 : 0;
  if (gf_build_plan(&cam, 1, N, P, strips, single, &pl) != CBA_OK) { printf("plan failed\n"); return 1; }
  const int n = pl.n_pad, nf = pl.n_fact, Gf = pl.Gf;
  printf("plan: strips %d, Gf %d, border %d, n_fact %d, n_pad %d, chains %zu, tasks %zu (list 0: %d), model GFLOP %.1f / %.1f / %.1f\n", pl.strips[0], Gf, pl.n_border,
         nf, n, pl.chains.size(), pl.tasks.size(), pl.n_tasks0, pl.flops_grid / 1e9, pl.flops_update / 1e9, pl.flops_border / 1e9);
  // ---- synthetic F: structural tiles of the grid part filled with small entries, dense border, diagonally dominant ----
  std::vector<double> hF((size_t)n * n, 0.0);
  const double off = 2e-4;
  for (size_t t = 0; t + 1 < pl.grid_tiles.size(); t += 2) {
    const int r = pl.grid_tiles[t], c = pl.grid_tiles[t + 1];
    for (int i = 0; i < 64; ++i)
      for (int j = 0; j < 64; ++j) {
        const int fi = 64 * r + i, fj = 64 * c + j;
        if (fj <= fi) continue;
        if (pl.grid_of_f[fi] < 0 || pl.grid_of_f[fj] < 0) continue;
        hF[(size_t)fi * n + fj] = off * rnd((size_t)fi * 7919 + fj);
      }
  }
  for (int fi = 0; fi < Gf + pl.n_border; ++fi) {
    if (fi < Gf && pl.grid_of_f[fi] < 0) continue;
    for (int fj = std::max(fi + 1, Gf); fj < Gf + pl.n_border; ++fj) hF[(size_t)fi * n + fj] = off * rnd((size_t)fi * 104729 + fj);
  }
  for (int i = 0; i < n; ++i) {
    const bool real = (i < Gf) ? pl.grid_of_f[i] >= 0 : (i < Gf + pl.n_border);
    hF[(size_t)i * n + i] = real ? 4.0 + 0.5 * rnd(i) : 1.0;
  }
  std::vector<double> hb(n, 0.0);
  for (int i = 0; i < Gf + pl.n_border; ++i) if (i >= Gf || pl.grid_of_f[i] >= 0) hb[i] = std::sin(0.37 * i) + 0.25;
  for (int i = 0; i < nf; ++i) hF[(size_t)i * n + (n - 1)] = hb[i];
  double *F0, *F, *Xb, *x;
  hipMalloc(&F0, sizeof(double) * (size_t)n * n); hipMalloc(&F, sizeof(double) * (size_t)n * n);
  hipMalloc(&Xb, sizeof(double) * (size_t)Gf * (n - Gf)); hipMemset(Xb, 0, sizeof(double) * (size_t)Gf * (n - Gf));
  hipMalloc(&x, sizeof(double) * n);
  hipMemcpy(F0, hF.data(), sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice);
  GfDevice g;
  hipMalloc(&g.tasks, sizeof(GfTask) * pl.tasks.size()); hipMemcpy(g.tasks, pl.tasks.data(), sizeof(GfTask) * pl.tasks.size(), hipMemcpyHostToDevice);
  hipMalloc(&g.ivals, sizeof(GfIval) * pl.ivals.size()); hipMemcpy(g.ivals, pl.ivals.data(), sizeof(GfIval) * pl.ivals.size(), hipMemcpyHostToDevice);
  hipMalloc(&g.chains, sizeof(GfChain) * pl.chains.size()); hipMemcpy(g.chains, pl.chains.data(), sizeof(GfChain) * pl.chains.size(), hipMemcpyHostToDevice);
  hipMalloc(&g.rowmask, sizeof(uint64_t) * pl.rowmask.size()); hipMemcpy(g.rowmask, pl.rowmask.data(), sizeof(uint64_t) * pl.rowmask.size(), hipMemcpyHostToDevice);
  g.n_tasks0 = pl.n_tasks0; g.n_tasks1 = (int)pl.tasks.size() - pl.n_tasks0; g.n_chains = (int)pl.chains.size(); g.nbg = pl.nbg; g.nbf = pl.nbf;
  g.mask_words = pl.mask_words; g.flops_grid = pl.flops_grid;
  prepare_device_streams();
  hipStream_t ms; make_main_stream(&ms);
  LdltWorkspace w; ldlt_workspace_alloc(w, n, pl.nbg);
  hipEvent_t e[5]; for (auto& ev : e) hipEventCreate(&ev);
  unsigned long long* tl = nullptr;
  const int nb_all = nf / 64;
  if (getenv("TAILLOG")) { hipMalloc(&tl, sizeof(unsigned long long) * 16 * nb_all); }
  double best[4] = {1e30, 1e30, 1e30, 1e30};
  for (int rep = 0; rep < reps; ++rep) {
    hipMemcpy(F, F0, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice);
    hipMemset(w.status, 0, 4);
    hipDeviceSynchronize();
    GemmStats gs;
    // the three launches separately (ldlt_factor_gridfirst inlined so that events can sit between them)
    if (tl && rep == reps - 1) { hipMemset(tl, 0, sizeof(unsigned long long) * 16 * nb_all); hipMemcpyToSymbol(HIP_SYMBOL(g_taillog), &tl, sizeof(tl)); }
    hipEventRecord(e[0], ms);
    ldlt_sparse(F, n, g, w, ms, &gs, Xb, n - Gf);
    hipEventRecord(e[1], ms);
    std::vector<unsigned long long> hs;
    if (tl && rep == reps - 1) { hipStreamSynchronize(ms); hs.resize(16 * (size_t)nb_all); hipMemcpy(hs.data(), tl, sizeof(unsigned long long) * hs.size(), hipMemcpyDeviceToHost); hipMemset(tl, 0, sizeof(unsigned long long) * 16 * nb_all); }
    {
      GemmArgs u{};
      u.A = F; u.lda = n; u.B = Xb - Gf; u.ldb = n - Gf; u.K = Gf;
      u.C = F; u.ldc = n; u.Cin = F; u.ldcin = n; u.diag = 0; u.upper = 1;
      const int tlc = (n - Gf) / 128;
      u.m_off = Gf; u.m_tiles = tlc; u.n_off = Gf; u.n_tiles = tlc;
      launch_gemm<128, 128, 64, 64, true>(u, ms);
    }
    hipEventRecord(e[2], ms);
    ldlt_factor(F, nf, n, w, ms, &gs, Gf);
    hipEventRecord(e[3], ms);
    ldlt_back_solve(F, nf, n, n - 1, w, x, ms, g.rowmask, g.mask_words);
    hipEventRecord(e[4], ms);
    hipStreamSynchronize(ms);
    int st = 0; hipMemcpy(&st, w.status, 4, hipMemcpyDeviceToHost);
    const double t[4] = {timeit(e[0], e[1]), timeit(e[1], e[2]), timeit(e[2], e[3]), timeit(e[3], e[4])};
    for (int i = 0; i < 4; ++i) best[i] = std::min(best[i], t[i]);
    printf("rep %d: grid rows %.3f ms, border update (dense) %.3f ms, border %.3f ms, back substitution %.3f ms, status %d\n", rep, t[0], t[1], t[2], t[3], st);
    if (tl && rep == reps - 1) {
      std::vector<unsigned long long> hb2(16 * (size_t)nb_all);
      hipMemcpy(hb2.data(), tl, sizeof(unsigned long long) * hb2.size(), hipMemcpyDeviceToHost);
      unsigned long long* nul = nullptr; hipMemcpyToSymbol(HIP_SYMBOL(g_taillog), &nul, sizeof(nul));
      static const char* names[9] = {"wait flags", "load U,P", "X = invL U", "X epilogue", "T product", "T write + publish tile", "T -> regs", "pivots", "epilogue + publish"};
      auto avg = [&](const std::vector<unsigned long long>& h, int b0, int b1, const char* what) {
        double acc[9] = {0}; int cnt = 0;
        for (int b = b0; b < b1; ++b) { if (h[16 * b + 9] == 0 || h[16 * b + 1] == 0) continue; for (int ph = 0; ph < 9; ++ph) acc[ph] += (double)(h[16 * b + ph + 1] - h[16 * b + ph]) / 100.0; ++cnt; }
        if (!cnt) return;
        printf("   %s, blocks %d-%d (us):", what, b0, b1);
        double tot = 0; for (int ph = 0; ph < 9; ++ph) { printf(" %s %.2f |", names[ph], acc[ph] / cnt); tot += acc[ph] / cnt; }
        printf(" total %.2f\n", tot);
      };
      for (size_t ci = 0; ci < pl.chains.size(); ++ci) {
        const int r0 = pl.chains[ci].r0, r1 = pl.chains[ci].r1;
        char nm[64]; snprintf(nm, sizeof nm, "grid chain %zu", ci);
        avg(hs, r0 + 1, r1, nm);
        printf("      span %.1f us for %d blocks = %.2f us / block\n", (double)(hs[16 * (r1 - 1) + 9] - hs[16 * r0]) / 100.0, r1 - r0, (double)(hs[16 * (r1 - 1) + 9] - hs[16 * r0]) / 100.0 / (r1 - r0));
      }
      const int nbb = pl.nbf - pl.nbg;
      avg(hb2, 1, std::min(nbb, 12), "border chain (start)");
      avg(hb2, nbb / 2 - 6, nbb / 2 + 6, "border chain (middle)");
      avg(hb2, nbb - 12, nbb, "border chain (end)");
      printf("      border chain span %.1f us for %d blocks = %.2f us / block\n", (double)(hb2[16 * (nbb - 1) + 9] - hb2[0]) / 100.0, nbb, (double)(hb2[16 * (nbb - 1) + 9] - hb2[0]) / 100.0 / nbb);
    }
  }
  printf("best: grid rows %.3f ms (%.1f GFLOP model), border update %.3f ms (dense: %.1f TFLOP/s), border %.3f ms, back substitution %.3f ms; sum %.3f ms\n", best[0],
         pl.flops_grid / 1e9, best[1], pl.flops_update / best[1] / 1e9, best[2], best[3], best[0] + best[1] + best[2] + best[3]);
  // residual on the host: r = F0 x - b over the real rows (F0 symmetric from its upper triangle)
  std::vector<double> hx(nf);
  hipMemcpy(hx.data(), x, sizeof(double) * nf, hipMemcpyDeviceToHost);
  std::vector<double> r(nf, 0.0);
  for (int i = 0; i < nf; ++i) {
    const double* row = &hF[(size_t)i * n];
    double acc = row[i] * hx[i];
    for (int j = i + 1; j < nf; ++j) { const double v = row[j]; if (v != 0.0) { acc += v * hx[j]; r[j] += v * hx[i]; } }
    r[i] += acc;
  }
  double rmax = 0, bmax = 0, xmax = 0; bool nan = false;
  for (int i = 0; i < nf; ++i) { rmax = std::max(rmax, std::fabs(r[i] - hb[i])); bmax = std::max(bmax, std::fabs(hb[i])); xmax = std::max(xmax, std::fabs(hx[i])); if (!(hx[i] == hx[i])) nan = true; }
  printf("residual |F x - b|max / |b|max = %.2e, |x|max %.3e%s\n", rmax / bmax, xmax, nan ? "  NaN!" : "");
  { double s1 = 0; for (int i = 0; i < nf; ++i) s1 += hx[i] * (1.0 + (i % 7)); printf("checksum of x: %.17g\n", s1); }
  return 0;
}
