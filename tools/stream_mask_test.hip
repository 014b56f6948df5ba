// Developer experiment: the bulk GEMM on different kinds of HIP streams
#include "../camera_calibration_amd/csrc/kernels_linalg.hip"
#include <cstdio>
#include <vector>
namespace cba { void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); } }
using namespace cba;
static float timeit(hipEvent_t e0, hipEvent_t e1) { float ms; hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms; }
int main(int argc, char** argv) {
  const int which = argc > 1 ? atoi(argv[1]) : 0;
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int n = 12672, K = 3008, nn = 10880, Kt = 512;
  double *A, *S;
  hipMalloc(&A, sizeof(double) * (size_t)K * n); hipMalloc(&S, sizeof(double) * (size_t)n * n);
  std::vector<double> hA((size_t)K * n);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((double)((i * 2654435761u) % 2001) / 1000.0 - 1.0) * 0.05;
  hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
  hipMemset(S, 0, sizeof(double) * (size_t)n * n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, hipStream_t st) {
    GemmArgs u{};
    u.A = A; u.lda = n; u.B = A + (size_t)1504 * n; u.ldb = n; u.K = Kt; u.C = S; u.ldc = n; u.Cin = S; u.ldcin = n;
    u.m_off = 0; u.m_tiles = nn / 128; u.n_off = 0; u.n_tiles = nn / 128; u.upper = 1; u.diag = 0;
    launch_gemm<128, 128, 64, 64, true>(u, st);
    hipEventRecord(e0, st);
    for (int r = 0; r < 5; ++r) launch_gemm<128, 128, 64, 64, true>(u, st);
    hipEventRecord(e1, st);
    float ms = timeit(e0, e1) / 5;
    double nt = nn / 128, tiles = nt * (nt + 1) / 2;
    if (name) printf("%-44s %.3f ms  %.2f TFLOP/s\n", name, ms, tiles * 2.0 * 128 * 128 * Kt / ms / 1e9);
  };
  auto masked = [&](const char* name, std::vector<int> off) {
    uint32_t m[8]; for (int i = 0; i < 8; ++i) m[i] = 0xffffffffu;
    for (int b : off) m[b >> 5] &= ~(1u << (b & 31));
    hipStream_t s; hipExtStreamCreateWithCUMask(&s, 8, m); run(name, s); hipStreamDestroy(s);
  };
  if (which == 7) {   // the user stream is the first queue the process uses
    hipStream_t s7; hipStreamCreateWithFlags(&s7, hipStreamNonBlocking);
    for (int r = 0; r < 10; ++r) run(r == 9 ? "first-used non-blocking stream (warm)" : nullptr, s7);
    run("null stream afterwards", nullptr);
    run("first-used non-blocking stream again", s7);
    return 0;
  }
  if (which == 9 || which == 10) {   // user streams created before the first launch, used after the null stream
    hipStream_t sa, sb;
    if (which == 9) { hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking); }
    else { uint32_t m[8]; for (int i = 0; i < 8; ++i) m[i] = 0xffffffffu; m[0] = 0xffffff00u; hipExtStreamCreateWithCUMask(&sa, 8, m);
           for (int i = 0; i < 8; ++i) m[i] = 0; m[0] = 0xffu; hipExtStreamCreateWithCUMask(&sb, 8, m); }
    for (int r = 0; r < 10; ++r) run(r == 9 ? "null stream (after warm-up)" : nullptr, nullptr);
    run("early-created stream a", sa);
    run("null stream again", nullptr);
    run("early-created stream a again", sa);
    return 0;
  }
  // warm the clocks up, then one configuration per process (a bad CU mask can hang the queue: run under `timeout`)
  for (int r = 0; r < 10; ++r) run(r == 9 ? "null stream (after warm-up)" : nullptr, nullptr);
  if (which == 8) {
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t s8; hipStreamCreateWithPriority(&s8, hipStreamNonBlocking, hi); run("high-priority stream", s8);
    hipStream_t s9; hipStreamCreateWithPriority(&s9, hipStreamNonBlocking, lo); run("low-priority stream", s9);
  }
  if (which == 1) { hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); run("non-blocking stream", s2); }
  if (which == 2) masked("mask: all 256 CUs", {});
  if (which == 3) masked("mask: bits 0-7 off", {0, 1, 2, 3, 4, 5, 6, 7});
  if (which == 4) masked("mask: bits 0,32,..,224 off", {0, 32, 64, 96, 128, 160, 192, 224});
  if (which == 5) masked("mask: bits 248-255 off", {248, 249, 250, 251, 252, 253, 254, 255});
  if (which == 6) masked("mask: bits 0-15 off", {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15});
  run("null stream again", nullptr);
  return 0;
}
