// Developer experiment: per-workgroup timeline of the bulk GEMM launch (128 x 128 tiles, K = 512, upper triangle of
// n = 10880: 3655 tiles) on the null stream, on the engine's CU-masked main stream with and without the SE-balanced slot
// assignment.  Prints, per configuration: launch span, tile duration percentiles, rounds per CU, idle share of the CUs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -DCBA_WGLOG tools/gemm_wg_timeline.hip -o tools/bin/gemm_wg_timeline
#define CBA_DEV_SWITCHES 1
#include "../camera_calibration_amd/csrc/kernels_linalg.hip"
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
namespace cba { void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); } }
using namespace cba;
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  prepare_device_streams();
  const int n = 10880, K = 512;
  double *A, *S;
  hipMalloc(&A, sizeof(double) * (size_t)2 * K * n); hipMalloc(&S, sizeof(double) * (size_t)n * n);
  std::vector<double> hA((size_t)2 * K * n);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((double)((i * 2654435761u) % 2001) / 1000.0 - 1.0) * 0.05;
  hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
  hipMemset(S, 0, sizeof(double) * (size_t)n * n);
  const int max_blocks = 8192;
  unsigned long long* d_log; hipMalloc(&d_log, sizeof(unsigned long long) * 4 * max_blocks);
  hipMemcpyToSymbol(HIP_SYMBOL(g_wglog), &d_log, sizeof(d_log));
  hipStream_t msk; make_main_stream(&msk);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, hipStream_t st) {
    GemmArgs u{};
    u.A = A; u.lda = n; u.B = A + (size_t)K * n; u.ldb = n; u.K = K; u.C = S; u.ldc = n; u.Cin = S; u.ldcin = n;
    u.m_off = 0; u.m_tiles = n / 128; u.n_off = 0; u.n_tiles = n / 128; u.upper = 1; u.diag = 0;
    for (int r = 0; r < 6; ++r) launch_gemm<128, 128, 64, 64, true>(u, st);      // clocks up
    hipStreamSynchronize(st);
    hipMemset(d_log, 0, sizeof(unsigned long long) * 4 * max_blocks);
    hipEventRecord(e0, st);
    launch_gemm<128, 128, 64, 64, true>(u, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(4 * (size_t)max_blocks);
    hipMemcpy(h.data(), d_log, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    std::vector<double> dur;                       // us, workgroups that ran a tile (longer than 5 us)
    std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> per_cu;
    int empties = 0;
    for (int b = 0; b < max_blocks; ++b) {
      const unsigned long long t0 = h[4 * b], t1 = h[4 * b + 1];
      if (!t1) continue;
      const double us = (t1 - t0) / 100.0;
      if (us < 5.0) { ++empties; continue; }
      tmin = std::min(tmin, t0); tmax = std::max(tmax, t1);
      dur.push_back(us);
      const unsigned hw = (unsigned)h[4 * b + 3];
      per_cu[(unsigned)h[4 * b + 2] << 16 | ((hw >> 13) & 7) << 8 | ((hw >> 8) & 15)].push_back({t0, t1});
    }
    std::sort(dur.begin(), dur.end());
    const double span = (tmax - tmin) / 100.0;
    // busy share of a CU: union of its workgroups' intervals / span; tiles per CU
    double busy_sum = 0; size_t tmin_cu = 1 << 30, tmax_cu = 0; double busy2_sum = 0;
    for (auto& kv : per_cu) {
      auto v = kv.second; std::sort(v.begin(), v.end());
      unsigned long long covered = 0, cur_end = 0; double two = 0;
      for (auto& iv : v) { if (iv.second > cur_end) { covered += iv.second - std::max(iv.first, cur_end); cur_end = iv.second; } two += (iv.second - iv.first); }
      busy_sum += covered / 100.0 / span; busy2_sum += two / 100.0 / (2 * span);
      tmin_cu = std::min(tmin_cu, v.size()); tmax_cu = std::max(tmax_cu, v.size());
    }
    const double tiles = (double)dur.size();
    printf("%-40s %.3f ms (%.1f TFLOP/s), device span %.0f us; %d tiles + %d empty workgroups; tile us p10 %.0f p50 %.0f p90 %.0f max %.0f; CUs %zu, tiles per CU %zu..%zu; "
           "CU busy (>= 1 workgroup) %.1f %%, slot occupancy (2 per CU) %.1f %%\n",
           name, ms, tiles * 2.0 * 128 * 128 * K / ms / 1e9, span, (int)dur.size(), empties, dur[dur.size() / 10], dur[dur.size() / 2], dur[dur.size() * 9 / 10],
           dur.back(), per_cu.size(), tmin_cu, tmax_cu, 100.0 * busy_sum / per_cu.size(), 100.0 * busy2_sum / per_cu.size());
  };
  run("null stream", nullptr);
  run(getenv("CBA_NO_SE_BALANCE") ? "masked main stream, static slots" : "masked main stream, SE-balanced", msk);
  run("null stream again", nullptr);
  return 0;
}
