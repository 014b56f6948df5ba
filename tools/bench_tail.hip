// Developer harness (not shipped): the two-level factorisation (super-panels + final dataflow launch) for several sizes of the
// final launch -- same matrix, solutions compared with the first setting, all timed; chain timeline of the final launch (TAILLOG),
// helper-task breakdown (HELPLOG), the helpers' K loop alone (MMA_ONLY).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form tools/bench_tail.hip -o tools/bin/bench_tail
#define CBA_DEV_SWITCHES 1
#define CBA_TAILLOG 1
#include "../camera_calibration_amd/csrc/kernels_linalg.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
namespace cba { void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); } }
// The helpers' K loop of rounds 2-4 (register-staged: buffer loads -> scale by d -> ds_write -> barrier -> MFMA), kept here as the
// reference of the LDS-DMA loop the product uses now (tail_mma_dma): same sums bit for bit, 48-50 vs 56-60 TFLOP/s.
namespace cba {
template <bool SYM>
__device__ __forceinline__ void tail_mma_regstaged(v4f64 (&acc)[2][2], const double* A, const double* B, int ld, const double* dk, int K,
                                         double* sA, double* sB) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32, li = lane & 15, lk = lane >> 4;
  const int r = tid >> 5, c2 = 2 * (tid & 31);          // rows r, r + 8, r + 16, r + 24 of a slab
  const int nk = K / kTailKT;                             // K is a multiple of 64
  const __amdgpu_buffer_rsrc_t ra = tail_rsrc(A), rb = tail_rsrc(B), rd = tail_rsrc(dk);
  const int rowb = ld * 8;
  // staging registers written out as scalars, one set per slot (arrays here end up in scratch memory)
  v2f64_t a0_0, a0_1, a0_2, a0_3, a1_0, a1_1, a1_2, a1_3;
  v2f64_t b0_0 = {0, 0}, b0_1 = {0, 0}, b0_2 = {0, 0}, b0_3 = {0, 0}, b1_0 = {0, 0}, b1_1 = {0, 0}, b1_2 = {0, 0}, b1_3 = {0, 0};
  double d0_0, d0_1, d0_2, d0_3, d1_0, d1_1, d1_2, d1_3;
#define CBA_XLOAD1(slot_, q_, k0_)                                                               \
  {                                                                                              \
    const int o = ((k0_) + r + 8 * (q_)) * rowb + c2 * 8;                                        \
    a##slot_##_##q_ = tail_ld2(ra, o);                                                           \
    if constexpr (!SYM) b##slot_##_##q_ = tail_ld2(rb, o);                                       \
    d##slot_##_##q_ = tail_ld1(rd, ((k0_) + r + 8 * (q_)) * 8);                                  \
  }
#define CBA_XLOAD(slot_, k0_) { CBA_XLOAD1(slot_, 0, k0_) CBA_XLOAD1(slot_, 1, k0_) CBA_XLOAD1(slot_, 2, k0_) CBA_XLOAD1(slot_, 3, k0_) }
#define CBA_XSTORE1(buf_, slot_, q_)                                                             \
  {                                                                                              \
    double* qa = sA + (buf_) * kTailKT * TS + (r + 8 * (q_)) * TS + c2;                          \
    double* qb = sB + (buf_) * kTailKT * TS + (r + 8 * (q_)) * TS + c2;                          \
    qa[0] = a##slot_##_##q_.x * d##slot_##_##q_; qa[1] = a##slot_##_##q_.y * d##slot_##_##q_;    \
    if constexpr (SYM) { qb[0] = a##slot_##_##q_.x; qb[1] = a##slot_##_##q_.y; }                 \
    else { qb[0] = b##slot_##_##q_.x; qb[1] = b##slot_##_##q_.y; }                               \
  }
#define CBA_XSTORE(buf_, slot_) { CBA_XSTORE1(buf_, slot_, 0) CBA_XSTORE1(buf_, slot_, 1) CBA_XSTORE1(buf_, slot_, 2) CBA_XSTORE1(buf_, slot_, 3) }
  CBA_XLOAD(0, 0);
  CBA_XLOAD(1, (1 < nk ? 1 : nk - 1) * kTailKT);
  CBA_XSTORE(0, 0);
  __syncthreads();
#define CBA_XSTEP(slot_, next_slot_)                                                                         \
  if (kb0 + (slot_) < nk) {                                                                                  \
    const int kb = kb0 + (slot_);                                                                            \
    const int buf = kb & 1;                                                                                  \
    CBA_XLOAD(slot_, (kb + 2 < nk ? kb + 2 : nk - 1) * kTailKT);                                             \
    const double* a_s = sA + buf * kTailKT * TS;                                                             \
    const double* b_s = sB + buf * kTailKT * TS;                                                             \
    /* operands of k-step kk + 4 are read before the MFMAs of step kk are issued (the compiler's own order, read -> wait -> */ \
    /* 4 MFMAs, left the matrix pipe idle for an LDS round trip per step) */                                 \
    double af[2][2], bf[2][2];                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) af[0][i] = a_s[lk * TS + wm0 + i * 16 + li];               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) bf[0][j] = b_s[lk * TS + wn0 + j * 16 + li];               \
    _Pragma("unroll") for (int kk = 0; kk < kTailKT; kk += 4) {                                              \
      const int cur = (kk >> 2) & 1, nxt = cur ^ 1;                                                          \
      if (kk + 4 < kTailKT) {                                                                                \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) af[nxt][i] = a_s[(kk + 4 + lk) * TS + wm0 + i * 16 + li]; \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) bf[nxt][j] = b_s[(kk + 4 + lk) * TS + wn0 + j * 16 + li]; \
      }                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                        \
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);      \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
    CBA_XSTORE(buf ^ 1, next_slot_);                                                                         \
    __syncthreads();                                                                                         \
  }
#pragma nounroll
  for (int kb0 = 0; kb0 < nk; kb0 += 2) {
    CBA_XSTEP(0, 1)
    CBA_XSTEP(1, 0)
  }
#undef CBA_XSTEP
#undef CBA_XLOAD
#undef CBA_XLOAD1
#undef CBA_XSTORE
#undef CBA_XSTORE1
}

}  // namespace cba
// (the 64 x 128 LDS-DMA K loop, tail_mma_dma2, moved into the product in round 5: REG2 tasks of k_ldlt_tail)
__global__ void __launch_bounds__(256, 2) k_mma2_only(const double* S, int ld, const double* dvec, int K, int ntc, double* out) {
  __shared__ double smem[2 * cba::kInner * cba::TS];
  const int c = blockIdx.x % ntc, r = (blockIdx.x / ntc) % ntc;
  cba::v4f64 acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (cba::v4f64){0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < K; k += 1024) {
    const int kk = K - k < 1024 ? K - k : 1024;
    cba::tail_mma_dma2(acc, S + (size_t)k * ld + r * cba::kInner, S + (size_t)k * ld + c * 2 * cba::kInner, ld, dvec + k, kk, smem);
  }
  // same checksum as k_mma_only over the two 64-column halves: half h of this tile = tile (r, 2 c + h) of the 64 x 64 kernel
  for (int h = 0; h < 2; ++h) {
    double v = 0;
    const int wv = threadIdx.x >> 6;
    // wave wv holds rows wm0 + 32, columns wn0 = (wv & 1) * 64 ... + 64: tiles j = 0..3 -> columns wn0 + 16 j; half h = (wv & 1)
    if ((wv & 1) == h) for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) v += acc[i][j][q];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && (wv & 1) == h) atomicAdd(&out[(blockIdx.x * 2 + h)], v);
  }
}
using namespace cba;

static float timeit(hipEvent_t e0, hipEvent_t e1) { float ms; hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms; }

struct Case { int n, n_fact; };
static double g_rate = 0; static int g_launches = 0;

// synthetic ceiling of the helpers' inner loop: every workgroup accumulates one 64 x 64 tile over K rows, no flags
template <bool SYM>
__global__ void __launch_bounds__(256, 2) k_mma_only(const double* S, int ld, const double* dvec, int K, int ntc, double* out, int variant) {
  __shared__ double smem[2 * kInner * TS];
  double* sA = smem + kInner * TS;
  double* sB = smem;
  const int c = blockIdx.x % ntc, r = (blockIdx.x / ntc) % ntc;
  v4f64 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < K; k += 1024) {
    const int kk = K - k < 1024 ? K - k : 1024;
    if (variant == 0) tail_mma_regstaged<SYM>(acc, S + (size_t)k * ld + r * kInner, S + (size_t)k * ld + c * kInner, ld, dvec + k, kk, sA, sB);
    else tail_mma_dma<SYM>(acc, S + (size_t)k * ld + r * kInner, S + (size_t)k * ld + c * kInner, ld, dvec + k, kk, smem);
  }
  double v = 0; for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 4; ++q) v += acc[i][j][q] * (1 + i + 2 * j + 4 * q);
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = v;
}

int main(int argc, char** argv) {
  std::vector<Case> cases;
  if (argc > 2) cases.push_back({atoi(argv[1]), atoi(argv[2])});
  else cases = {{1152, 1088}, {2304, 2240}, {12672, 12544}};
  std::vector<int> tails = {1024, 2048, 3072, 4096, 6144, 8192};
  if (const char* e = getenv("TAILS")) { tails.clear(); for (const char* c = e; *c;) { tails.push_back(atoi(c)); while (*c && *c != ',') ++c; if (*c) ++c; } }
  const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 3;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  prepare_device_streams();
  hipStream_t ms; make_main_stream(&ms);
  if (getenv("HELPLOG")) {
    // helper-task breakdown of ONE dataflow launch: rows [k0, k0 + W) of a super-panel (W = 0: a final tail from k0)
    const int n = 12672, n_fact = 12544, K = 1024;
    const int k0 = getenv("HL_K0") ? atoi(getenv("HL_K0")) : 0, W = getenv("HL_W") ? atoi(getenv("HL_W")) : 2048;
    double *A, *S, *H; hipMalloc(&A, sizeof(double) * (size_t)K * n); hipMalloc(&S, sizeof(double) * (size_t)n * n); hipMalloc(&H, sizeof(double) * (size_t)n * n);
    std::vector<double> hA((size_t)K * n);
    for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((double)((i * 2654435761u) % 2001) / 1000.0 - 1.0) * 0.05;
    hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
    hipMemset(H, 0, sizeof(double) * (size_t)n * n);
    schur_gemm(A, A, K, n, H, S, n, n, n_fact, 1, -1.0 * K * 0.001, nullptr, nullptr);
    hipDeviceSynchronize();
    LdltWorkspace w; ldlt_workspace_alloc(w, n);
    unsigned long long* hl; const size_t nlog = (size_t)1 << 20; hipMalloc(&hl, nlog * 64); hipMemset(hl, 0, nlog * 64);
    for (int rep = 0; rep < 2; ++rep) {
      if (rep == 1) hipMemcpyToSymbol(HIP_SYMBOL(g_helplog), &hl, sizeof(hl));
      GemmStats gs;
      hipEventRecord(e0, ms);
      ldlt_tail(S, W > 0 ? k0 + W : n_fact, n, k0, w, ms, &gs, W > 0 ? w.X : nullptr);
      hipEventRecord(e1, ms);
      printf("dataflow launch rows [%d, %d): %.3f ms\n", k0, W > 0 ? k0 + W : n_fact, timeit(e0, e1));
    }
    std::vector<unsigned long long> h(nlog * 8);
    hipMemcpy(h.data(), hl, nlog * 64, hipMemcpyDeviceToHost);
    double tot = 0, wait = 0, mma = 0, dwait = 0, epi = 0; long cnt = 0; unsigned long long tmin = ~0ull, tmax = 0;
    double by_kind[3][5] = {{0}};
    for (size_t i = 0; i < nlog; ++i) {
      const unsigned long long* e = &h[i * 8];
      if (e[4] == 0) continue;
      const double T = (double)(e[4] - e[0]) / 100.0;
      const int kd = (int)e[5];
      tot += T; wait += e[1] / 100.0; mma += e[2] / 100.0; dwait += e[3] / 100.0; epi += T - (e[1] + e[2] + e[3]) / 100.0; ++cnt;
      by_kind[kd][0] += 1; by_kind[kd][1] += T; by_kind[kd][2] += e[1] / 100.0; by_kind[kd][3] += e[2] / 100.0; by_kind[kd][4] += e[3] / 100.0;
      tmin = std::min(tmin, e[0]); tmax = std::max(tmax, e[4]);
    }
    printf("REG tasks logged: %ld, span %.1f us; workgroup-time: total %.0f us = wait rows %.1f %% + k-loop %.1f %% + wait diag %.1f %% + epilogue/other %.1f %%\n",
           cnt, (double)(tmax - tmin) / 100.0, tot, 100 * wait / tot, 100 * mma / tot, 100 * dwait / tot, 100 * epi / tot);
    printf("per REG task: %.1f us (wait rows %.1f, k-loop %.1f, wait diag %.1f, rest %.1f)\n", tot / cnt, wait / cnt, mma / cnt, dwait / cnt, epi / cnt);
    return 0;
  }
  if (getenv("MMA2_ONLY")) {
    const int n = 12672, K = 4096, ntc = 48;
    double *S, *dv, *out; hipMalloc(&S, sizeof(double) * (size_t)K * n); hipMalloc(&dv, sizeof(double) * K); hipMalloc(&out, 8 * 8192);
    std::vector<double> h((size_t)K * n); for (size_t i = 0; i < h.size(); ++i) h[i] = ((double)((i * 2654435761u) % 2001) / 1000.0 - 1.0) * 0.05;
    hipMemcpy(S, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> hd(K); for (int i = 0; i < K; ++i) hd[i] = 1.0 + 0.001 * (i % 97); hipMemcpy(dv, hd.data(), K * 8, hipMemcpyHostToDevice);
    // reference sums of two tiles from the host (tile (0, 0..1), plain loops) for one workgroup
    for (int grid : {128, 256, 512, 1024}) for (int rep = 0; rep < 2; ++rep) {
      hipMemset(out, 0, 8 * 8192);
      hipEventRecord(e0, ms);
      hipLaunchKernelGGL(k_mma2_only, dim3(grid), dim3(256), 0, ms, S, n, dv, K, ntc, out);
      hipEventRecord(e1, ms);
      const float t = timeit(e0, e1);
      double got[2]; hipMemcpy(got, out, 16, hipMemcpyDeviceToHost);
      double ref[2] = {0, 0};
      if (grid == 128 && rep == 0) {
        for (int hh = 0; hh < 2; ++hh) for (int k = 0; k < K; ++k) { double sa = 0, sb = 0; for (int m = 0; m < 64; ++m) sa += h[(size_t)k * n + m]; for (int c2 = 0; c2 < 64; ++c2) sb += h[(size_t)k * n + hh * 64 + c2]; ref[hh] += hd[k] * sa * sb; }
        printf("   tile sums device %.10e %.10e | host %.10e %.10e\n", got[0], got[1], ref[0], ref[1]);
      }
      printf("mma2_only (64x128, LDS-DMA, slabs of 16) grid %d K %d: %.3f ms  %.2f TFLOP/s\n", grid, K, t, grid * 2.0 * 64 * 128 * K / t / 1e9);
    }
    return 0;
  }
  if (getenv("MMA_ONLY")) {
    const int n = 12672, K = 4096, ntc = getenv("MMA_NTC") ? atoi(getenv("MMA_NTC")) : 96;
    double *S, *dv, *out; hipMalloc(&S, sizeof(double) * (size_t)K * n); hipMalloc(&dv, sizeof(double) * K); hipMalloc(&out, 8 * 8192);
    std::vector<double> h((size_t)K * n); for (size_t i = 0; i < h.size(); ++i) h[i] = ((double)((i * 2654435761u) % 2001) / 1000.0 - 1.0) * 0.05;
    hipMemcpy(S, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> hd(K); for (int i = 0; i < K; ++i) hd[i] = 1.0 + 0.001 * (i % 97); hipMemcpy(dv, hd.data(), K * 8, hipMemcpyHostToDevice);
    std::vector<double> ref(8192), got(8192);
    for (int variant = 0; variant < 2; ++variant)
    for (int grid : {256, 512, 1024, 2048}) for (int rep = 0; rep < 2; ++rep) {
      hipMemset(out, 0, 8 * 8192);
      hipEventRecord(e0, ms);
      if (getenv("MMA_SYM")) hipLaunchKernelGGL(k_mma_only<true>, dim3(grid), dim3(256), 0, ms, S, n, dv, K, ntc, out, variant);
      else hipLaunchKernelGGL(k_mma_only<false>, dim3(grid), dim3(256), 0, ms, S, n, dv, K, ntc, out, variant);
      hipEventRecord(e1, ms);
      const float t = timeit(e0, e1);
      hipMemcpy(got.data(), out, 8 * 8192, hipMemcpyDeviceToHost);
      if (variant == 0 && grid == 2048) ref = got;
      double dev = 0, mx = 0; for (int i = 0; i < grid * 4 && i < 8192; ++i) { dev = std::max(dev, std::fabs(got[i] - ref[i])); mx = std::max(mx, std::fabs(ref[i])); }
      printf("mma_only %s grid %d K %d: %.3f ms  %.2f TFLOP/s%s\n", variant ? "LDS-DMA" : "register-staged", grid, K, t, grid * 2.0 * 64 * 64 * K / t / 1e9,
             variant ? (dev == 0 ? "  [bit-identical to the register-staged loop]" : "  [DIFFERS]") : "");
      if (variant && dev != 0) printf("   max dev %.3e of %.3e\n", dev, mx);
    }
    return 0;
  }
  for (const Case& cs : cases) {
    const int n = cs.n, n_fact = cs.n_fact, K = 1024;
    printf("=== n_pad %d n_fact %d ===\n", n, n_fact);
    double *A, *S0, *S, *H, *x;
    hipMalloc(&A, sizeof(double) * (size_t)K * n); hipMalloc(&S0, sizeof(double) * (size_t)n * n); hipMalloc(&S, sizeof(double) * (size_t)n * n);
    hipMalloc(&H, sizeof(double) * (size_t)n * n); hipMalloc(&x, sizeof(double) * n);
    std::vector<double> hA((size_t)K * n);
    for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((double)((i * 2654435761u) % 2001) / 1000.0 - 1.0) * 0.05;
    hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
    hipMemset(H, 0, sizeof(double) * (size_t)n * n);
    // S0 = -lambda' I - A^T A (negative definite, fine for LDL^T); rows >= n_fact: identity; last column: right-hand side
    schur_gemm(A, A, K, n, H, S0, n, n, n_fact, 1, -1.0 * K * 0.001, nullptr, nullptr);
    hipDeviceSynchronize();
    {
      std::vector<double> row(n);
      for (int r = n_fact; r < n; ++r) {           // padding rows: unit diagonal, zero elsewhere (as the engine keeps them)
        std::fill(row.begin(), row.end(), 0.0); row[r] = 1.0;
        hipMemcpy(S0 + (size_t)r * n, row.data(), sizeof(double) * n, hipMemcpyHostToDevice);
      }
      std::vector<double> col(n_fact);
      for (int r = 0; r < n_fact; ++r) col[r] = std::sin(0.37 * r) + 0.25;
      hipMemcpy2D(S0 + (n - 1), sizeof(double) * n, col.data(), sizeof(double), sizeof(double), n_fact, hipMemcpyHostToDevice);
      // columns [n_fact, n - 1) of the factored rows: zero
      if (n - 1 > n_fact) hipMemset2D(S0 + n_fact, sizeof(double) * n, 0, sizeof(double) * (n - 1 - n_fact), n_fact);
    }
    LdltWorkspace w; ldlt_workspace_alloc(w, n);
    auto run = [&](int tail, std::vector<double>* hx, std::vector<double>* hS, std::vector<double>* hd, double* best_ms, double* tail_ms) {
      w.tail_rows = tail;
      *best_ms = 1e30; *tail_ms = 0;
      int st = 0;
      for (int rep = 0; rep < reps; ++rep) {
        hipMemcpy(S, S0, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice);
        hipMemset(w.status, 0, 4);
        hipDeviceSynchronize();
        GemmStats gs;
        hipEventRecord(e0, ms);
        ldlt_factor(S, n_fact, n, w, ms, &gs);
        hipEventRecord(e1, ms);
        const double t = timeit(e0, e1);
        GemmStats g2;
        ldlt_collect_spans(w, &g2);
        if (t < *best_ms) { *best_ms = t; *tail_ms = ldlt_tail_last_ms(w); g_rate = g2.seconds > 0 ? g2.flops / g2.seconds / 1e12 : 0; g_launches = g2.launches; }
        hipMemcpy(&st, w.status, 4, hipMemcpyDeviceToHost);
        if (st) break;
      }
      w.tail_timed = false;
      {
        // back substitution: panels of 256 (round 2) against the dataflow launch, same factor
        std::vector<double> xa(n_fact), xb(n_fact);
        float tp = 1e30f, td = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          w.back_dataflow = false;
          hipEventRecord(e0, ms); ldlt_back_solve(S, n_fact, n, n - 1, w, x, ms); hipEventRecord(e1, ms); tp = std::min(tp, timeit(e0, e1));
          hipMemcpy(xa.data(), x, sizeof(double) * n_fact, hipMemcpyDeviceToHost);
          w.back_dataflow = true;
          hipEventRecord(e0, ms); ldlt_back_solve(S, n_fact, n, n - 1, w, x, ms); hipEventRecord(e1, ms); td = std::min(td, timeit(e0, e1));
        }
        hipMemcpy(xb.data(), x, sizeof(double) * n_fact, hipMemcpyDeviceToHost);
        double dmax = 0, xm = 0; for (int i = 0; i < n_fact; ++i) { dmax = std::max(dmax, std::fabs(xa[i] - xb[i])); xm = std::max(xm, std::fabs(xa[i])); }
        printf("   back substitution: panels %.3f ms, dataflow %.3f ms, |dx| / |x|max %.2e\n", tp, td, dmax / xm);
      }
      hipStreamSynchronize(ms);
      hx->resize(n_fact); hipMemcpy(hx->data(), x, sizeof(double) * n_fact, hipMemcpyDeviceToHost);
      hd->resize(n_fact); hipMemcpy(hd->data(), w.dvec, sizeof(double) * n_fact, hipMemcpyDeviceToHost);
      if (hS) { hS->resize((size_t)n * n); hipMemcpy(hS->data(), S, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost); }
      return st;
    };
    std::vector<double> xr, Sr, dr;
    double ms_ref, tms;
    if (getenv("PAIRS")) setenv("CBA_TAIL_PAIR", "0", 1);          // reference = single-tile tasks
    int st = run(tails.back(), &xr, n <= 4096 ? &Sr : nullptr, &dr, &ms_ref, &tms);
    printf("reference (tail %d): %.3f ms  status %d   [128x128 GEMM launches: %d at %.1f TFLOP/s per launch]\n", tails.back(), ms_ref, st, g_launches, g_rate);
    double xmax = 0; for (double v : xr) xmax = std::max(xmax, std::fabs(v));
    { double s1 = 0, s2 = 0; for (size_t i = 0; i < xr.size(); ++i) { s1 += xr[i] * (1.0 + (i % 7)); s2 += xr[i] * xr[i]; } printf("checksum of x (compare across builds): %.17g %.17g\n", s1, s2); }
    std::vector<int> pair_modes = {-1};
    if (const char* e = getenv("PAIRS")) { pair_modes.clear(); for (const char* c = e; *c;) { pair_modes.push_back(atoi(c)); while (*c && *c != ',') ++c; if (*c) ++c; } }
    for (int tail : tails) for (int pm : pair_modes) {
      if (pm >= 0) { setenv("CBA_TAIL_PAIR", pm ? "1" : "0", 1); printf("-- CBA_TAIL_PAIR=%d\n", pm); }
      std::vector<double> xt, St, dt;
      double ms_t;
      st = run(tail, &xt, n <= 4096 ? &St : nullptr, &dt, &ms_t, &tms);
      double dx = 0, dd = 0, dmax = 0, dS = 0, smax = 0;
      for (int i = 0; i < n_fact; ++i) { dx = std::max(dx, std::fabs(xt[i] - xr[i])); dd = std::max(dd, std::fabs(dt[i] - dr[i])); dmax = std::max(dmax, std::fabs(dr[i])); }
      bool nan = false; for (double v : xt) if (!(v == v)) nan = true;
      if (!St.empty())
        for (int r = 0; r < n_fact; ++r) for (int c = r; c < n; ++c) {
          const double a = St[(size_t)r * n + c], b = Sr[(size_t)r * n + c];
          dS = std::max(dS, std::fabs(a - b)); smax = std::max(smax, std::fabs(b));
        }
      int t0 = 0;                                  // first row of the final launch (the loop of ldlt_factor)
      { const int sw = super_width(), tr = ldlt_tail_rows(w); while (n_fact - t0 > tr + sw / 2 && n - (t0 + sw) >= 1024) t0 += super_width_at(n, t0, sw); }
      if (getenv("TAILLOG") && n_fact - t0 >= 512) {
        // one more run with the chain timeline
        const int nb = (n_fact - t0) / 64;
        unsigned long long* tl; hipMalloc(&tl, sizeof(unsigned long long) * 16 * nb); hipMemset(tl, 0, sizeof(unsigned long long) * 16 * nb);
        hipMemcpyToSymbol(HIP_SYMBOL(g_taillog), &tl, sizeof(tl));
        std::vector<double> x2, d2; double m2, tm2;
        run(tail, &x2, nullptr, &d2, &m2, &tm2);
        std::vector<unsigned long long> h(16 * (size_t)nb);
        hipMemcpy(h.data(), tl, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        unsigned long long* nul = nullptr; hipMemcpyToSymbol(HIP_SYMBOL(g_taillog), &nul, sizeof(nul));
        hipFree(tl);
        static const char* names[9] = {"wait flags", "load U,P", "X = invL U", "X epilogue", "T product", "T write + publish tile", "T -> regs", "pivots", "epilogue + publish"};
        auto avg = [&](int b0, int b1) {
          double acc[9] = {0}; int cnt = 0;
          for (int b = std::max(b0, 1); b < b1; ++b) { for (int ph = 0; ph < 9; ++ph) acc[ph] += (double)(h[16 * b + ph + 1] - h[16 * b + ph]) / 100.0; ++cnt; }
          printf("   chain phases, blocks %d-%d (us):", b0, b1);
          double tot = 0; for (int ph = 0; ph < 9; ++ph) { printf(" %s %.2f |", names[ph], acc[ph] / cnt); tot += acc[ph] / cnt; }
          printf(" total %.2f\n", tot);
        };
        avg(1, std::min(nb, 8)); avg(nb / 2 - 4, nb / 2 + 4); avg(nb - 8, nb);
        printf("   chain span %.1f us for %d blocks = %.2f us / block\n", (double)(h[16 * (nb - 1) + 9] - h[0]) / 100.0, nb, (double)(h[16 * (nb - 1) + 9] - h[0]) / 100.0 / nb);
      }
      printf("tail %5d (starts at row %5d, %4d rows): %.3f ms total, tail launch %.3f ms, status %d | x rel %.2e  d rel %.2e  L rel %.2e%s  [GEMM launches: %d at %.1f TFLOP/s]\n",
             tail, t0, n_fact - t0, ms_t, tms, st, dx / xmax, dd / dmax, smax > 0 ? dS / smax : 0.0, nan ? "  NaN!" : "", g_launches, g_rate);
    }
    // residual of the solution with the last setting on small cases (host, O(n^2))
    ldlt_workspace_free(w);
    hipFree(A); hipFree(S0); hipFree(S); hipFree(H); hipFree(x);
  }
  return 0;
}
