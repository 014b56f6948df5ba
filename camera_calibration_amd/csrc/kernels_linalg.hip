// Schur-complement stage of the bundle-adjustment engine (gfx950): block inverses, D^-1 B, the fp64
// MFMA GEMM  S = H_dd + lambda I - B^T (D^-1 B), and a blocked LDL^T factorisation / solve of the
// reduced system.  Restates what LMOptimizer::SolveWithSchurComplementDenseOffDiag computes
// (libvis/src/libvis/lm_optimizer.h:1247-1369 in the reference tree); Eigen's dense LDLT call sites
// (:1289, :1361) become hand-written kernels.
//
// Storage: every symmetric matrix keeps its upper triangle in row-major order (as the reference's
// accumulator writes it, lm_optimizer_update_accumulator.h:212,256).  Reading the same memory as a
// column-major matrix gives the lower triangle, so the factorisation below is a textbook
// right-looking, lower, column-major LDL^T whose "columns" are contiguous memory rows -- every panel
// operation streams contiguous rows and every product is the one GEMM shape
//        C[m][n] (-)= sum_k A[k][m] * B[k][n]      (A, B: K x ld row-major, "K-major" operands)
// which feeds v_mfma_f64_16x16x4_f64 directly from LDS rows.
// Leading dimensions are padded to multiples of 128 and K to multiples of 16 so the hot loops carry
// no bounds checks; padded diagonal entries are set to 1.
#include <cmath>
#include <cstdlib>

#include "cba_internal.h"
#include "gridfirst_plan.h"
#include <algorithm>
#include <map>
#include <mutex>

namespace cba {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// per-block inverse (bs <= 6) of D_i + lambda I by LDL^T with diagonal pivoting, and D^-1 b.
// One lane per block; blocks hold only their upper triangle.
// ------------------------------------------------------------------------------------------------
__global__ void k_block_inverse(const double* __restrict__ Dblk, const double* __restrict__ bblk, double lambda, int bs,
                                int nb, double* __restrict__ Dinv, double* __restrict__ dinvb, int* __restrict__ status) {
  int blk = blockIdx.x * blockDim.x + threadIdx.x;
  if (blk >= nb) return;
  double A[6][6], Inv[6][6];
  for (int r = 0; r < bs; ++r)
    for (int c = 0; c < bs; ++c) {
      int lo = r < c ? r : c, hi = r < c ? c : r;
      A[r][c] = Dblk[(size_t)blk * bs * bs + lo * bs + hi] + (r == c ? lambda : 0.0);
      Inv[r][c] = (r == c) ? 1.0 : 0.0;
    }
  // symmetric Gauss-Jordan with diagonal pivoting on the remaining diagonal (exact for any
  // non-singular symmetric block, definite or not)
  int perm[6];
  bool used[6];
  for (int i = 0; i < bs; ++i) used[i] = false;
  bool bad = false;
  for (int step = 0; step < bs; ++step) {
    int p = -1; double best = -1.0;
    for (int i = 0; i < bs; ++i)
      if (!used[i] && fabs(A[i][i]) > best) { best = fabs(A[i][i]); p = i; }
    perm[step] = p; used[p] = true;
    double piv = A[p][p];
    if (!(fabs(piv) > 0.0)) { bad = true; break; }
    double ip = 1.0 / piv;
    for (int c = 0; c < bs; ++c) { A[p][c] *= ip; Inv[p][c] *= ip; }
    for (int r = 0; r < bs; ++r) {
      if (r == p) continue;
      double f = A[r][p];
      if (f == 0.0) continue;
      for (int c = 0; c < bs; ++c) { A[r][c] -= f * A[p][c]; Inv[r][c] -= f * Inv[p][c]; }
    }
  }
  (void)perm;
  if (bad) atomicExch(status, 1);
  for (int r = 0; r < bs; ++r) {
    double acc = 0.0;
    for (int c = 0; c < bs; ++c) {
      double v = bad ? NAN : 0.5 * (Inv[r][c] + Inv[c][r]);
      Dinv[(size_t)blk * bs * bs + r * bs + c] = v;
      acc += v * bblk[(size_t)blk * bs + c];
    }
    dinvb[(size_t)blk * bs + r] = acc;
  }
}
// The same elimination for the block size of the path (6: pose blocks) with every loop unrolled and the pivot row picked by
// selects, so that both matrices stay in registers (the generic kernel indexes its arrays with the run-time pivot: they live in
// scratch memory, 95 us for 500 blocks on the critical path of every solve).  Same operations in the same order.
template <int BS>
__global__ void __launch_bounds__(64) k_block_inverse_fixed(const double* __restrict__ Dblk, const double* __restrict__ bblk, double lambda,
                                                            int nb, double* __restrict__ Dinv, double* __restrict__ dinvb, int* __restrict__ status) {
  const int blk = blockIdx.x * blockDim.x + threadIdx.x;
  if (blk >= nb) return;
  double A[BS][BS], Inv[BS][BS];
#pragma unroll
  for (int r = 0; r < BS; ++r)
#pragma unroll
    for (int c = 0; c < BS; ++c) {
      const int lo = r < c ? r : c, hi = r < c ? c : r;
      A[r][c] = Dblk[(size_t)blk * BS * BS + lo * BS + hi] + (r == c ? lambda : 0.0);
      Inv[r][c] = (r == c) ? 1.0 : 0.0;
    }
  unsigned used = 0;
  bool bad = false;
#pragma unroll
  for (int step = 0; step < BS; ++step) {
    int p = -1; double best = -1.0;
#pragma unroll
    for (int i = 0; i < BS; ++i)
      if (!((used >> i) & 1u) && fabs(A[i][i]) > best) { best = fabs(A[i][i]); p = i; }
    if (bad) continue;                       // (the generic kernel leaves its loop here)
    if (p < 0) { bad = true; continue; }     // nothing comparable left on the diagonal (NaN)
    used |= 1u << p;
    double pa[BS], pi[BS], piv = 0.0;
#pragma unroll
    for (int i = 0; i < BS; ++i)
      if (i == p) {
        piv = A[i][i];
#pragma unroll
        for (int c = 0; c < BS; ++c) { pa[c] = A[i][c]; pi[c] = Inv[i][c]; }
      }
    if (!(fabs(piv) > 0.0)) { bad = true; continue; }
    const double ip = 1.0 / piv;
#pragma unroll
    for (int c = 0; c < BS; ++c) { pa[c] *= ip; pi[c] *= ip; }
    double colp[BS];                         // A[r][p] of every row, before the row operations of this step
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      double v = 0.0;
#pragma unroll
      for (int c = 0; c < BS; ++c) if (c == p) v = A[r][c];
      colp[r] = v;
    }
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      if (r == p) {
#pragma unroll
        for (int c = 0; c < BS; ++c) { A[r][c] = pa[c]; Inv[r][c] = pi[c]; }
      } else {
        const double f = colp[r];
        if (f != 0.0) {
#pragma unroll
          for (int c = 0; c < BS; ++c) { A[r][c] -= f * pa[c]; Inv[r][c] -= f * pi[c]; }
        }
      }
    }
  }
  if (bad) atomicExch(status, 1);
  double bb[BS];
#pragma unroll
  for (int c = 0; c < BS; ++c) bb[c] = bblk[(size_t)blk * BS + c];
#pragma unroll
  for (int r = 0; r < BS; ++r) {
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < BS; ++c) {
      const double v = bad ? NAN : 0.5 * (Inv[r][c] + Inv[c][r]);
      Dinv[(size_t)blk * BS * BS + r * BS + c] = v;
      acc += v * bb[c];
    }
    dinvb[(size_t)blk * BS + r] = acc;
  }
}
int launch_block_inverse(const double* Dblk, const double* bblk, double lambda, int bs, int nb, double* Dinv,
                         double* dinvb, int* status, hipStream_t s) {
  if (nb == 0) return CBA_OK;
  if (bs == 6) hipLaunchKernelGGL(k_block_inverse_fixed<6>, dim3((nb + 63) / 64), dim3(64), 0, s, Dblk, bblk, lambda, nb, Dinv, dinvb, status);
  else hipLaunchKernelGGL(k_block_inverse, dim3((nb + 63) / 64), dim3(64), 0, s, Dblk, bblk, lambda, bs, nb, Dinv, dinvb, status);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// W[blk*bs + r][col] = sum_k Dinv[blk][r][k] * B[blk*bs + k][col]   (lm_optimizer.h:1302-1310)
__global__ void __launch_bounds__(256) k_dinv_times_B(const double* __restrict__ Dinv, const double* __restrict__ B, int bs,
                                                      int dd, int ld, double* __restrict__ W) {
  int blk = blockIdx.y;
  int col = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ double sD[36];
  if ((int)threadIdx.x < bs * bs) sD[threadIdx.x] = Dinv[(size_t)blk * bs * bs + threadIdx.x];
  __syncthreads();
  if (col >= dd) return;
  double b[6];
  for (int k = 0; k < bs; ++k) b[k] = B[((size_t)blk * bs + k) * ld + col];
  for (int r = 0; r < bs; ++r) {
    double acc = 0.0;
    for (int k = 0; k < bs; ++k) acc += sD[r * bs + k] * b[k];
    W[((size_t)blk * bs + r) * ld + col] = acc;
  }
}

// y[k] = base[k] - sum_j M[k][j] v[j]; one wavefront per row
__global__ void __launch_bounds__(256) k_gemv_n(const double* __restrict__ M, int K, int n, int ld, const double* __restrict__ v,
                                                const double* __restrict__ base, double* __restrict__ y) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= K) return;
  const double* r = M + (size_t)row * ld;
  double acc = 0.0;
  if ((((size_t)r | (size_t)v) & 15) == 0) {
    // 16 bytes per lane, four independent loads in flight per lane (8-byte loads one at a time: 3 TB/s on a 300 MB matrix)
    const double2* r2 = reinterpret_cast<const double2*>(r);
    const double2* v2 = reinterpret_cast<const double2*>(v);
    const int n2 = n >> 1;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int j = lane;
    for (; j + 192 < n2; j += 256) {
      const double2 m0 = r2[j], m1 = r2[j + 64], m2 = r2[j + 128], m3 = r2[j + 192];
      const double2 w0 = v2[j], w1 = v2[j + 64], w2 = v2[j + 128], w3 = v2[j + 192];
      a0 += m0.x * w0.x + m0.y * w0.y; a1 += m1.x * w1.x + m1.y * w1.y;
      a2 += m2.x * w2.x + m2.y * w2.y; a3 += m3.x * w3.x + m3.y * w3.y;
    }
    for (; j < n2; j += 64) { const double2 m0 = r2[j], w0 = v2[j]; a0 += m0.x * w0.x + m0.y * w0.y; }
    acc = (a0 + a1) + (a2 + a3);
    if ((n & 1) && lane == 0) acc += r[n - 1] * v[n - 1];
  } else {
    for (int j = lane; j < n; j += 64) acc += r[j] * v[j];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) y[row] = (base ? base[row] : 0.0) - acc;
}
int launch_gemv_n(const double* M, int K, int n, int ld, const double* v, const double* base, double* y, hipStream_t s) {
  if (K == 0) return CBA_OK;
  hipLaunchKernelGGL(k_gemv_n, dim3((K + 3) / 4), dim3(256), 0, s, M, K, n, ld, v, base, y);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ------------------------------------------------------------------------------------------------
// fp64 MFMA GEMM:  C[m][n] = Cin[m][n] + diag(m==n) - sum_k A[k][m] B[k][n]     (SUB = true)
//                  C[m][n] =                           sum_k A[k][m] B[k][n]     (SUB = false)
// Block tile TM x TN, 4 wavefronts, KT = 16 rows of A and B per LDS stage (row stride padded by 16
// doubles so the four k-rows of one MFMA operand fetch land in disjoint bank halves).
// v_mfma_f64_16x16x4_f64 operand map: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15];
// result D: column j = l&15, row i = (l>>4) + 4*reg.
// Tile selection: `upper` launches only tiles whose column range reaches the diagonal (n_tile >= m_tile).
// The linear block index is permuted so that the blocks of one XCD (blockIdx % 8) own consecutive
// tiles of the same tile row and share its A panel in that XCD's L2.
// ------------------------------------------------------------------------------------------------
constexpr int KT = 16;
// K slab of the block-sparse Schur launch: 12 rows = the pose blocks of exactly TWO imagesets (6 rows each), so a slab never straddles
// a third or fourth imageset as a 16-row slab (2.7 imagesets) does -- the product skips a slab only if ALL its rows are zero in one of
// the two column tiles, and the imagesets are ordered so that neighbours have similar footprints (cba_set_observations).  Modelled
// from the observation lists: 0.88 of the 16-row slabs' work at cfg 2 (0.76 against the Z-order of round 4); 48 instead of 64 MFMAs
// per wavefront and barrier.  The dense launches (super-panel updates) keep KT = 16.
constexpr int kSchurSlab = 12;
constexpr int kSchurChunk = 64;   // tiles per XCD chunk of a block-sparse launch

struct GemmArgs {
  const double* A; int lda;     // K x lda, column offset already applied for m_begin = 0 of this call
  const double* B; int ldb;
  int K;                        // multiple of the launch's slab (KT dense, kSchurSlab block-sparse)
  double* C; int ldc;
  const double* Cin; int ldcin; // may alias C
  int m_tiles, n_tiles;         // tile counts of this call
  int m_off, n_off;             // element offsets of tile (0,0) inside C (and A/B column spaces)
  int upper;                    // only tiles with (n_off + tn*TN + TN - 1) >= (m_off + tm*TM)
  int n_real;                   // rows/cols < n_real get diag_add, others 1.0 (only if diag)
  int diag;                     // add to diagonal entries
  const double* diag_add_ptr;   // device scalar (lambda) or null
  double diag_add;              // host scalar used when diag_add_ptr == null
  long long total_tiles;
  int chunk;                    // tiles per XCD chunk (set by launch_gemm)
  const unsigned long long* kmask;  // optional block-sparsity mask [column tile][kmask_words], bit = K slab of kSchurSlab rows
  int kmask_words;
  const int* chunk_order;           // block-sparse launches: permutation of the 64-tile chunks, heaviest first (null = as enumerated)
  int n_chunks;
  int strips;                   // set by launch_gemm: strip-blocked tile order (square upper dense launches)
  int col_group, col_stride;    // distributed factorisation: owned column groups (tiles per group, group stride); 0 = all columns
  int keep_col_p1;              // 1 + a column of C the launch must not write (the right-hand side kept in S's last column); 0 = none
  int slab16;                   // block-sparse launch with 16-row K slabs (the border update of the grid-first order); 0 = slabs of kSchurSlab rows
  int tile_list_entries;        // slots of a tile_list launch
  const int4* tile_list;        // optional explicit order of the launch's tiles (tm, tn, s0, s1): slot b runs tile_list[b], tm = -1: no tile.
                                // The dispatcher hands workgroups out in slot order as slots come free, i.e. list scheduling: with the
                                // tiles sorted by executed K slabs, heaviest first, the light tiles fill the gaps behind the heavy ones.
                                // s1 > 0: a PART of the tile -- K slabs [s0, s1) only, added to C with fp64 atomics (the other part(s) of
                                // the tile are entries of their own and run whenever: a tile with all slabs is a third of the launch's
                                // makespan, two halves are not); s1 = 0: the whole tile, plain read-modify-write
};

// Developer switches are compiled only into the bench harness (tools/bench_tail.hip, tools/bench_diag.hip define CBA_DEV_SWITCHES): the
// product library has no epilogue modes and reads no CBA_* environment variables.
#ifdef CBA_DEV_SWITCHES
#define CBA_GETENV(name_) getenv(name_)
#else
#define CBA_GETENV(name_) ((const char*)nullptr)
#endif

// slot -> position in the launch's tile enumeration (>= total_tiles: no tile)
__device__ __forceinline__ long long gemm_slot_tile(const GemmArgs& g, long long b) {
  const long long q = b >> 3;
  const long long cq = q / g.chunk;
  long long c = cq * 8 + (b & 7);
  if (g.chunk_order) {
    // the eight chunks of a round (one per XCD) are neighbours in the order by executed K slabs: the XCDs finish together and
    // the light chunks of the sparse grid x grid region come last instead of leaving CUs idle behind dense ones
    if (c >= g.n_chunks) return g.total_tiles;
    c = g.chunk_order[c];
  }
  return c * g.chunk + (q - cq * g.chunk);
}

// col_group launches (distributed factorisation): first column of owned tile column tn, and the number of tile rows of the
// launch that reach the diagonal of that column
__host__ __device__ inline int colgroup_col0(const GemmArgs& g, int tn, int TN) {
  return g.n_off + ((tn / g.col_group) * g.col_stride * g.col_group + tn % g.col_group) * TN;
}
template <int TM, int TN>
__host__ __device__ inline int colgroup_strip_rows(const GemmArgs& g, int tn_last) {
  const int last = colgroup_col0(g, tn_last, TN) + TN - 1;       // last column of the strip
  if (last < g.m_off) return 0;
  const int rows = (last - g.m_off) / TM + 1;
  return rows < g.m_tiles ? rows : g.m_tiles;
}
// One output tile.  `b` is the linear slot of the tile (= blockIdx.x: workgroup b runs on XCD b % 8, observed dispatch
// order; only speed depends on it).  Returns false when the slot is past the last tile.
template <int TM, int TN, int WM, int WN, bool SUB, int KTT>
__device__ __forceinline__ bool gemm_tile(const GemmArgs& g, long long b) {
  static_assert(KTT % 4 == 0 && KTT >= 4 && KTT <= 16, "a stage is KTT / 4 MFMA k-steps; every wavefront moves KTT / 4 rows of each operand");
  constexpr int LDA_S = TM + 16, LDB_S = TN + 16;
  constexpr int WAVES_N = TN / WN;
  constexpr int MI = WM / 16, NJ = WN / 16;

  // ---- tile decode (XCD-aware permutation of the linear slot) ----
  // Tiles are dealt to the XCDs in chunks of 64 consecutive tiles of the row-major upper-triangle order: the 64
  // workgroup slots of an XCD share one A panel (and neighbouring B panels) in its L2, and chunks from
  // all parts of the matrix land on every XCD, which balances the block-sparse K loops.
  const long long t = g.tile_list ? b : gemm_slot_tile(g, b);
  if (t >= (g.tile_list ? (long long)g.tile_list_entries : g.total_tiles)) return false;
  int tm, tn;
  bool part = false;            // a K range of the tile (tile_list): starts from zero, leaves through atomics
  int s_lo = 0, s_hi = g.K / KTT;
  if (g.tile_list) {
    const int4 tt = g.tile_list[t];
    tm = tt.x; tn = tt.y;
    if (tm < 0) return true;
    if (tt.w > 0) { part = true; s_lo = tt.z; s_hi = tt.w; }
  } else if (g.strips) {
    // Square upper-triangular launch, dense: tiles are enumerated strip by strip (kStripW tile columns), row by
    // row inside a strip, so that the ~64 workgroups in flight on an XCD form an 8 x 8 block sharing 8 A and
    // 8 B panels (row-major order: 1 A panel and 64 different B panels -- 4x the operand traffic).
    constexpr int kStripW = 8;
    long long rem = t;
    int s = 0, w = 0;
    for (;; ++s) {
      const int c0 = s * kStripW;
      w = g.n_tiles - c0 < kStripW ? g.n_tiles - c0 : kStripW;
      const long long cnt = (long long)c0 * w + (long long)w * (w + 1) / 2;   // full rows above + triangle
      if (rem < cnt) break;
      rem -= cnt;
    }
    const int c0 = s * kStripW;
    if (rem < (long long)c0 * w) { tm = (int)(rem / w); tn = c0 + (int)(rem - (long long)tm * w); }
    else {
      rem -= (long long)c0 * w;
      int i = 0;
      while (rem >= w - i) { rem -= w - i; ++i; }
      tm = c0 + i; tn = c0 + i + (int)rem;
    }
  } else if (g.upper) {
    // enumerate tile rows; row tm owns tiles tn in [first(tm), n_tiles)
    // first(tm) = smallest tn with n_off + tn*TN + TN - 1 >= m_off + tm*TM
    long long rem = t;
    tm = 0;
    for (;; ++tm) {
      long long mrow = (long long)g.m_off + (long long)tm * TM;
      long long first = (mrow > g.n_off) ? (mrow - g.n_off) / TN : 0;  // first tile whose columns reach row mrow
      long long cnt = g.n_tiles - first;
      if (cnt < 0) cnt = 0;
      if (rem < cnt) { tn = (int)(first + rem); break; }
      rem -= cnt;
    }
  } else if (g.col_group > 0) {
    // distributed factorisation: strips of 8 OWNED tile columns, row by row inside a strip down to the diagonal of the strip's
    // last column (colgroup_strip_rows): the ~64 workgroups in flight on an XCD share 8 A and 8 B panels, and only the few
    // tiles between the diagonals of a strip's column groups are enumerated in vain (skipped below)
    long long rem = t;
    int c0 = 0, w = 0;
    for (;; c0 += 8) {
      w = g.n_tiles - c0 < 8 ? g.n_tiles - c0 : 8;
      const long long cnt = (long long)colgroup_strip_rows<TM, TN>(g, c0 + w - 1) * w;
      if (rem < cnt) break;
      rem -= cnt;
    }
    tm = (int)(rem / w);
    tn = c0 + (int)(rem - (long long)tm * w);
  } else {
    tm = (int)(t / g.n_tiles);
    tn = (int)(t - (long long)tm * g.n_tiles);
  }
  const int m0 = g.m_off + tm * TM;
  int n0 = g.n_off + tn * TN;
  if (g.col_group > 0) {
    // distributed factorisation: the launch covers only the column groups this rank owns (every `col_stride`-th group of
    // `col_group` tiles); tiles below the diagonal are launched and skipped
    n0 = colgroup_col0(g, tn, TN);
    if (n0 + TN - 1 < m0) return true;
  }

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm0 = (wv / WAVES_N) * WM, wn0 = (wv % WAVES_N) * WN;
  const int li = lane & 15, lk = lane >> 4;

  // SUB: the accumulators start as -(Cin + diag) so that the read of the C tile overlaps the first
  // operand slab (and the co-resident workgroup's MFMAs) instead of sitting in the epilogue; the
  // result is C = -acc.  (Measured: the epilogue read cost 0.44 ms per 1.2 GB trailing update.)
  v4f64 acc[MI][NJ];
  // SUB: the accumulators start as -(Cin + diag) (see below); in the LDS-DMA variant the tile is loaded AFTER the
  // first operand slab has been put in flight so that the two HBM round trips overlap.
  auto preload_c = [&]() {
    if (SUB && !part) {
      double dadd0 = 0.0;
      if (g.diag) dadd0 = g.diag_add_ptr ? *g.diag_add_ptr : g.diag_add;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm0 + i * 16 + lk + 4 * r;
            const int n = n0 + wn0 + j * 16 + li;
            double cin = g.Cin[(size_t)m * g.ldcin + n];
            if (g.diag && m == n) cin += (m < g.n_real) ? dadd0 : 1.0;
            acc[i][j][r] = -cin;
          }
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
    }
  };

  const double* Ag = g.A + m0;
  const double* Bg = g.B + n0;
  const int nk = s_hi;          // (the whole K range unless the entry is a part of a tile)
  static_assert(TM == 128 && TN == 128, "only the 128 x 128 LDS-DMA tile is built (the register-staged panel variants went with the blocked schedule)");
  {
    // Stage pipeline with LDS-DMA (global_load_lds_dwordx4): each wavefront-instruction moves one
    // 1 KiB row segment (64 lanes x 16 B) of the K-major operand straight into its (padded) LDS row,
    // no staging registers.  The slab for stage kb+1 is in flight while the MFMAs consume stage kb;
    // one vmcnt(0) + barrier per stage.
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    // Four separate LDS arrays (not one array indexed by the stage parity): the compiler's wait-count insertion
    // only lets an LDS read run ahead of an LDS-DMA in flight when it can prove that the two touch different LDS
    // variables; with sA[buf] / sA[buf ^ 1] it put an s_waitcnt vmcnt(0) between the DMA issue and the first
    // ds_read of EVERY stage, i.e. the next slab was never in flight during the MFMAs of the current one.
    __shared__ double dA0[KTT * LDA_S], dA1[KTT * LDA_S], dB0[KTT * LDB_S], dB1[KTT * LDB_S];
#define CBA_DMA_STAGE(SA_, SB_, k0_)                                                                               \
  {                                                                                                                \
    _Pragma("unroll") for (int j = 0; j < KTT / 4; ++j) {                                                           \
      const int row = wv * (KTT / 4) + j;                                                                           \
      __builtin_amdgcn_global_load_lds((gbl_ptr)(Ag + (size_t)((k0_) + row) * g.lda + 2 * lane),                   \
                                       (lds_ptr)&SA_[row * LDA_S], 16, 0, 0);                                      \
      __builtin_amdgcn_global_load_lds((gbl_ptr)(Bg + (size_t)((k0_) + row) * g.ldb + 2 * lane),                   \
                                       (lds_ptr)&SB_[row * LDB_S], 16, 0, 0);                                      \
    }                                                                                                              \
  }
#define CBA_MMA_STAGE(SA_, SB_)                                                                                    \
  {                                                                                                                \
    _Pragma("unroll") for (int kk = 0; kk < KTT; kk += 4) {                                                         \
      double af[MI], bf[NJ];                                                                                       \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) af[i] = SA_[(kk + lk) * LDA_S + wm0 + i * 16 + li];           \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) bf[j] = SB_[(kk + lk) * LDB_S + wn0 + j * 16 + li];           \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                             \
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);                      \
    }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                               \
    __syncthreads();                                                                                               \
  }
    // Block-sparse K loop: `kmask` (optional) holds, per 128-column tile, one bit per 16-row K slab that
    // contains any non-zero.  A slab contributes to tile (tm, tn) only if both column tiles touch it --
    // in bundle adjustment an imageset's rows of B are non-zero only at the points it sees and the grid
    // cells it covers -- so the loop walks the set bits of mask[tm] & mask[tn].
    const unsigned long long* ma = g.kmask ? g.kmask + (size_t)(m0 >> 7) * g.kmask_words : nullptr;
    const unsigned long long* mb = g.kmask ? g.kmask + (size_t)(n0 >> 7) * g.kmask_words : nullptr;
    int mword = -1;
    unsigned long long mbits = 0;
    auto next_slab = [&](int after) -> int {
      if (!g.kmask) return after + 1;
      int s = after + 1;
      while (s < nk) {
        if ((s >> 6) != mword) { mword = s >> 6; mbits = ma[mword] & mb[mword]; }   // one load per 64 slabs
        unsigned long long bits = mbits >> (s & 63);
        if (bits) return s + __builtin_ctzll(bits);
        s = (s | 63) + 1;
      }
      return nk;
    };
    int kb = next_slab(s_lo - 1);
    if (kb < nk) CBA_DMA_STAGE(dA0, dB0, kb * KTT);
    preload_c();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    while (kb < nk) {
      int nxt = next_slab(kb);                       // slab kb is in dA0 / dB0
      if (nxt < nk) CBA_DMA_STAGE(dA1, dB1, nxt * KTT);
      CBA_MMA_STAGE(dA0, dB0);
      kb = nxt;
      if (kb >= nk) break;
      nxt = next_slab(kb);                           // slab kb is in dA1 / dB1
      if (nxt < nk) CBA_DMA_STAGE(dA0, dB0, nxt * KTT);
      CBA_MMA_STAGE(dA1, dB1);
      kb = nxt;
    }
#undef CBA_MMA_STAGE
#undef CBA_DMA_STAGE
  }

  // ---- epilogue ----
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm0 + i * 16 + lk + 4 * r;
        const int n = n0 + wn0 + j * 16 + li;
        double v = acc[i][j][r];
        if (SUB) v = -v;
        if (n + 1 == g.keep_col_p1) continue;
        if (part) atomicAdd(&g.C[(size_t)m * g.ldc + n], v);      // C += -(A^T B over the part's slabs)
        else g.C[(size_t)m * g.ldc + n] = v;
      }
  return true;
}

// One tile per workgroup.
template <int TM, int TN, int WM, int WN, bool SUB, int KTT = KT>
__global__ void __launch_bounds__(256) k_gemm_atb(GemmArgs g) {
  gemm_tile<TM, TN, WM, WN, SUB, KTT>(g, blockIdx.x);
}

static long long count_upper_tiles(int m_off, int n_off, int m_tiles, int n_tiles, int TM, int TN) {
  long long total = 0;
  for (int tm = 0; tm < m_tiles; ++tm) {
    long long mrow = (long long)m_off + (long long)tm * TM;
    long long first = (mrow > n_off) ? (mrow - n_off) / TN : 0;
    long long cnt = n_tiles - first;
    if (cnt > 0) total += cnt;
  }
  return total;
}

template <int TM, int TN, int WM, int WN, bool SUB>
static int launch_gemm(GemmArgs g, hipStream_t s) {
  g.total_tiles = g.upper ? count_upper_tiles(g.m_off, g.n_off, g.m_tiles, g.n_tiles, TM, TN)
                          : (long long)g.m_tiles * g.n_tiles;
  if (g.col_group > 0) {
    g.total_tiles = 0;
    for (int c0 = 0; c0 < g.n_tiles; c0 += 8) {
      const int w = g.n_tiles - c0 < 8 ? g.n_tiles - c0 : 8;
      g.total_tiles += (long long)colgroup_strip_rows<TM, TN>(g, c0 + w - 1) * w;
    }
  }
  if (g.total_tiles <= 0) return CBA_OK;
  // chunk = 64 tiles for big launches; small launches use smaller chunks so that all eight XCDs get work
  long long per = (g.total_tiles + 7) / 8;
  // dense launches: one contiguous range per XCD (best L2 reuse); block-sparse launches: chunks of 64
  // interleaved over the XCDs so that dense and sparse regions of the matrix are spread evenly
  g.chunk = (int)((g.kmask && per > kSchurChunk) ? kSchurChunk : (per < 1 ? 1 : per));   // col_group launches are dense: the few skipped tiles sit at the end of every strip
  static const int use_strips = CBA_GETENV("CBA_NO_STRIPS") ? 0 : 1;
  g.strips = (use_strips && TM == 128 && TN == 128 && g.upper && !g.kmask && g.m_off == g.n_off && g.m_tiles == g.n_tiles &&
              g.total_tiles >= 512) ? 1 : 0;
  long long chunks = (g.total_tiles + g.chunk - 1) / g.chunk;
  g.n_chunks = (int)chunks;
  if (g.chunk != kSchurChunk) g.chunk_order = nullptr;            // the order was built for chunks of kSchurChunk tiles
  long long blocks = ((chunks + 7) / 8) * 8 * g.chunk;
  if (g.tile_list) { blocks = g.tile_list_entries; g.strips = 0; }
  if (g.kmask && !g.slab16) hipLaunchKernelGGL((k_gemm_atb<TM, TN, WM, WN, SUB, kSchurSlab>), dim3((unsigned)blocks), dim3(256), 0, s, g);   // block-sparse: slabs of two pose blocks
  else hipLaunchKernelGGL((k_gemm_atb<TM, TN, WM, WN, SUB, KT>), dim3((unsigned)blocks), dim3(256), 0, s, g);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

int launch_dinv_times_B_ld(const double* Dinv, const double* B, int bs, int nb, int dd, int ld, double* W, hipStream_t s) {
  if (nb == 0 || dd == 0) return CBA_OK;
  hipLaunchKernelGGL(k_dinv_times_B, dim3((dd + 255) / 256, nb), dim3(256), 0, s, Dinv, B, bs, dd, ld, W);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}
// y[j*ystride] = base[j] - sum_k M[k][j] v[k] in two deterministic stages: kGemvChunks row chunks
// produce partial sums (coalesced along j), a second kernel adds them in fixed order.
constexpr int kGemvChunks = 64;
__global__ void __launch_bounds__(256) k_gemv_t_partial(const double* __restrict__ M, int K, int n, int ld,
                                                        const double* __restrict__ v, double* __restrict__ partial) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int c = blockIdx.y;
  int per = (K + kGemvChunks - 1) / kGemvChunks;
  int k0 = c * per, k1 = k0 + per < K ? k0 + per : K;
  if (j >= n) return;
  double acc = 0.0;
  for (int k = k0; k < k1; ++k) acc += M[(size_t)k * ld + j] * v[k];
  partial[(size_t)c * n + j] = acc;
}
// entries [n, n_zero) of y are set to zero -- the padding rows of the right-hand side column -- except the last one, which is the
// matrix's last diagonal entry when y is the last column of S (one, like every padding diagonal entry)
__global__ void __launch_bounds__(256) k_gemv_t_final(const double* __restrict__ partial, int n, const double* __restrict__ base,
                                                      double* __restrict__ y, int ystride, int n_zero) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) { if (j < n_zero) y[(size_t)j * ystride] = (j == n_zero - 1) ? 1.0 : 0.0; return; }
  double acc = 0.0;
  for (int c = 0; c < kGemvChunks; ++c) acc += partial[(size_t)c * n + j];
  y[(size_t)j * ystride] = (base ? base[j] : 0.0) - acc;
}
int launch_gemv_t_strided(const double* M, int K, int n, int ld, const double* v, const double* base, double* y,
                          int ystride, double* partial_ws, hipStream_t s) {
  if (n == 0) return CBA_OK;
  hipLaunchKernelGGL(k_gemv_t_partial, dim3((n + 255) / 256, kGemvChunks), dim3(256), 0, s, M, K, n, ld, v, partial_ws);
  hipLaunchKernelGGL(k_gemv_t_final, dim3((n + 255) / 256), dim3(256), 0, s, partial_ws, n, base, y, ystride, n);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}
// the two stages separately: the partial sums only need M and v, the final stage writes y (solve_system runs the first next to
// the Schur product and the second behind it)
int launch_gemv_t_partial(const double* M, int K, int n, int ld, const double* v, double* partial_ws, hipStream_t s) {
  if (n == 0) return CBA_OK;
  hipLaunchKernelGGL(k_gemv_t_partial, dim3((n + 255) / 256, kGemvChunks), dim3(256), 0, s, M, K, n, ld, v, partial_ws);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}
int launch_gemv_t_final(int n, const double* base, double* y, int ystride, const double* partial_ws, int n_zero, hipStream_t s) {
  if (n == 0 && n_zero == 0) return CBA_OK;
  const int m = n > n_zero ? n : n_zero;
  hipLaunchKernelGGL(k_gemv_t_final, dim3((m + 255) / 256), dim3(256), 0, s, partial_ws, n, base, y, ystride, n_zero);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}
int gemv_t_workspace_doubles(int n) { return kGemvChunks * n; }

// S = Hdd + lambda I - A^T B on the upper tiles (n_pad x n_pad, all leading dims = ld, multiples of 128)
// bit (tile t, slab k) = any non-zero in B[kSchurSlab k .. kSchurSlab k + kSchurSlab - 1][128t .. 128t+127]
__global__ void __launch_bounds__(256) k_touch_mask(const double* __restrict__ B, int ld, unsigned long long* __restrict__ mask,
                                                    int words) {
  const int slab = blockIdx.x, tile = blockIdx.y;
  const double* p = B + (size_t)slab * kSchurSlab * ld + (size_t)tile * 128;
  bool nz = false;
  for (int e = threadIdx.x; e < kSchurSlab * 128; e += 256) nz = nz || (p[(size_t)(e >> 7) * ld + (e & 127)] != 0.0);
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  if (nz) any = 1;
  __syncthreads();
  if (threadIdx.x == 0 && any) atomicOr(mask + (size_t)tile * words + (slab >> 6), 1ull << (slab & 63));
}
int schur_mask_words(int Kpad) { return (Kpad / kSchurSlab + 63) / 64; }
int schur_slab_rows() { return kSchurSlab; }
// Chunks of the block-sparse Schur launch (kSchurChunk consecutive upper tiles in row-major order) sorted by the K slabs they
// execute, heaviest first; `order` gets schur_chunk_count(n_pad) entries, or is left alone when the launch would not use chunks
int schur_chunk_count(int n_pad) {
  const long long tiles = (long long)(n_pad / 128) * (n_pad / 128 + 1) / 2;
  return ((tiles + 7) / 8 > kSchurChunk) ? (int)((tiles + kSchurChunk - 1) / kSchurChunk) : 0;
}
void schur_chunk_order(const unsigned long long* mask_host, int n_pad, int Kpad, int* order) {
  const int nt = n_pad / 128, words = schur_mask_words(Kpad), nc = schur_chunk_count(n_pad);
  if (nc == 0) return;
  std::vector<std::pair<long long, int>> work(nc);
  for (int c = 0; c < nc; ++c) work[c] = {0, c};
  long long t = 0;
  for (int tm = 0; tm < nt; ++tm)
    for (int tn = tm; tn < nt; ++tn, ++t) {
      long long slabs = 0;
      for (int w = 0; w < words; ++w) slabs += __builtin_popcountll(mask_host[(size_t)tm * words + w] & mask_host[(size_t)tn * words + w]);
      work[t / kSchurChunk].first += slabs + 2;            // + the C tile's read / write
    }
  std::stable_sort(work.begin(), work.end(), [](const std::pair<long long, int>& a, const std::pair<long long, int>& b) { return a.first > b.first; });
  for (int c = 0; c < nc; ++c) order[c] = work[c].second;
  // XCD x walks order[x], order[8 + x], order[16 + x], ... (gemm_slot_tile; the dispatcher deals workgroups to the XCDs round-robin): with
  // the list sorted, XCD 0 would get the heaviest chunk of EVERY octet and XCD 7 the lightest -- differences that add up to about one
  // heavy chunk (~ 18 % of an XCD's share at cfg 2).  Every other octet is dealt in reverse (boustrophedon): the totals even out.
  for (int r = 1; 8 * r + 8 <= nc; r += 2) std::reverse(order + 8 * r, order + 8 * r + 8);
}
int launch_touch_mask(const double* B, int Kpad, int n_pad, int ld, unsigned long long* mask, hipStream_t s) {
  const int words = schur_mask_words(Kpad);
  CBA_HIP(hipMemsetAsync(mask, 0, sizeof(unsigned long long) * (size_t)(n_pad / 128) * words, s));
  hipLaunchKernelGGL(k_touch_mask, dim3(Kpad / kSchurSlab, n_pad / 128), dim3(256), 0, s, B, ld, mask, words);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

int schur_gemm(const double* A, const double* B, int Kpad, int ldab, const double* Cin, double* C, int n_pad, int ld,
               int n_real, int add_diag, double lambda, const unsigned long long* kmask, hipStream_t s, const int* chunk_order = nullptr,
               int keep_col = -1) {
  GemmArgs g{};
  g.keep_col_p1 = keep_col + 1;
  g.kmask = kmask; g.kmask_words = schur_mask_words(Kpad);
  g.chunk_order = chunk_order;
  g.A = A; g.lda = ldab; g.B = B; g.ldb = ldab; g.K = Kpad;
  g.C = C; g.ldc = ld; g.Cin = Cin; g.ldcin = ld;
  g.m_tiles = n_pad / 128; g.n_tiles = n_pad / 128; g.m_off = 0; g.n_off = 0; g.upper = 1;
  g.n_real = n_real; g.diag = add_diag; g.diag_add_ptr = nullptr; g.diag_add = lambda;
  return launch_gemm<128, 128, 64, 64, true>(g, s);
}

// ------------------------------------------------------------------------------------------------
// LDL^T, lower/column-major view of "upper in row-major" storage.
//   kInner = 64 : diagonal blocks factored (and their unit-lower factors inverted) by the chain workgroup of a dataflow launch
//   kPanel = 256: panel width of the panel version of the back substitution
// ------------------------------------------------------------------------------------------------
constexpr int kInner = 64;
constexpr int kPanel = 256;
constexpr int kSuperMax = 4096;              // widest super-panel (rows factored by one dataflow launch in front of a bulk update)
constexpr int kTailMaxBlockRows = 192;      // the persistent tail launch covers at most this many 64-row blocks (flag storage)
// Reciprocal of a pivot: v_rcp_f64 refined by two Newton steps (the IEEE division expands to ~3x as
// many dependent instructions, and 1/d sits on the critical path of every elimination step).
__device__ __forceinline__ double pivot_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

constexpr int TS = kInner + 16;   // LDS row stride (doubles) of a staged K-slab / 64x64 tile

__device__ __forceinline__ void tile_mma_lds(v4f64 (&acc)[2][2], const double* Al, const double* Bl) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32, li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < kInner; kk += 4) {
    double af[2], bf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[i] = Al[(kk + lk) * TS + wm0 + i * 16 + li];
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[j] = Bl[(kk + lk) * TS + wn0 + j * 16 + li];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------------
// Dataflow factorisation of a block-row range in ONE persistent launch (ldlt_tail, k_ldlt_tail).
//
// A blocked schedule of separate launches (rounds 1-2: per 64-block a diagonal factor, a near step, solves and updates on four
// streams) is bound by its pivot chain below ~6000 remaining rows: launch gaps, waits for the last workgroup of the previous
// launch and cross-stream stalls (profiles/r02_factor_timeline_pairs.txt: 5.1 ms for 12 % of the flops at config 2).  Here the
// whole range is factored by one launch in which every 64 x 64 tile is a task and tasks synchronise through device-scope flags:
//
//   chain workgroup (the first one to arrive): for r = r0, r0 + 1, ...: X = invL_{r-1} U_{r-1,r} (both operands in LDS: the
//       inverse it has just computed never leaves the CU), L_{r-1,r} = X / d published, T_rr = P_r - L^T X, 64 pivots
//       (chain_factor_blocked), L_rr / d / invL_rr published.  No launch, no stream event, no other tile on its critical path.
//   helper workgroups: tasks drawn from one ticket counter in row-major order (a task only ever waits for tasks with smaller
//       tickets or for the chain, so the launch cannot deadlock however many workgroups are resident):
//       PRE(r)    U_{r,r+1} = A_{r,r+1} - sum_{k<r} (d_k L_kr)^T L_{k,r+1}       in place (what the chain's next step reads)
//       PART(r+1) P_{r+1}   = A_{r+1,r+1} - sum_{k<r} (d_k L_{k,r+1})^T L_{k,r+1}  in place
//       REG(r,c)  U = A_rc - sum_{k<r} (d_k L_kr)^T L_kc, then (after block r is factored) X = invL_r U, L_rc = X / d_r.
//   LEFT-looking: a tile is read once, accumulated in registers over all earlier block rows (one K loop that follows the
//   frontier of finished rows: as many ready rows per batch as there are, at most 32) and written once -- no read-modify-write
//   of the trailing matrix per panel.  Only S is read: the update uses d_k L_k^T L_k (the A fragments are scaled by d_k on their
//   way from LDS to the MFMA), no panel buffer.
//
// Cross-workgroup visibility: everything another workgroup reads is written with agent-scope stores (sc1, write-through) and
// read with agent-scope loads (sc1 buffer loads / sc1 LDS-DMA, 16 B per lane); a flag is raised after s_waitcnt vmcnt(0) + barrier.  Flags hold
// the number of the factorisation call ("epoch"), so nothing has to be cleared between calls.  Every spin is bounded
// (kTailTimeoutTicks of the 100 MHz clock): on a timeout the launch sets status 3, raises the abort flag and ends.
// ------------------------------------------------------------------------------------------------
struct TailArgs {
  double* S; int ld;
  int rt0, nr, ntc;                 // first tail block row, number of block rows to factor, number of block columns (64 wide)
  double* dvec; double* invLt; int* status;
  unsigned* tile_flag;              // [(r - rt0) * ntc + c]: L_rc published
  unsigned* diag_flag;              // [r - rt0]: block r factored (L_rr, d, invL_rr published)
  unsigned* upre_flag;              // [r - rt0]: U_{r,r+1} in place
  unsigned* part_flag;              // [r - rt0]: P_r in place
  unsigned* ctrl;                   // [1] abort, [2] role tickets, [3] CU of the chain workgroup, [8 + x] task tickets of list x
  unsigned epoch;
  int ntasks;
  int evict;                        // helper workgroups that share the chain's CU stop taking tasks
  double* X; int ldx; int x_c0;     // super-panel mode: X = d L of the tiles with column block >= x_c0 goes to X[(64 (r - rt0) + p) * ldx + col]
                                    // (the K-major B operand of the bulk update that follows); null = not needed
  int xcd_lists;                    // 1: one task list per XCD (column block c -> XCD c % 8), own list first; 0: one list
  int ntasks_x[8];                  // tasks per list
  int pair;                         // one list only: REG tasks take TWO adjacent column blocks (64 x 128 tile, kind 3) beyond the first
                                    // kTailNearSingles columns of a row (tail_task)
  // block-sparse launch (k_ldlt_sparse; gridfirst_plan.h): static task lists with K intervals, several pivot chains
  const GfTask* tasks;              // list 0 (ntasks_x[0] entries), then list 1 (ntasks_x[1])
  const GfIval* ivals;
  const GfChain* chains;            // role i < n_chains runs chain i; ctrl[kCtrlChainCu + i] = its CU
  int n_chains;
  int n_critical;                   // helper workgroups (roles n_chains ... n_chains + n_critical - 1) that serve list 0 first
  const unsigned long long* act;    // optional activity of the border tiles: [(c - x_c0) / 2][act_words], bit r = block row r of the 128-column
  int act_words;                    // tile can be non-zero (kernels_gridfirst.hip: k_gf_touch / k_gf_close); inactive tiles are neither computed nor read
};
// first block row >= k (< kend) whose bit is set / clear in `bits`; kend if there is none
__device__ __forceinline__ int bits_next(const unsigned long long* bits, int k, int kend, bool want_set) {
  while (k < kend) {
    unsigned long long w = bits[k >> 6];
    if (!want_set) w = ~w;
    w >>= (k & 63);
    if (w) { const int hit = k + __builtin_ctzll(w); return hit < kend ? hit : kend; }
    k = (k | 63) + 1;
  }
  return kend;
}
constexpr int kCtrlWords = 128;     // control words of a dataflow launch: [1] abort, [2] role tickets, [3] CU of the chain (dense launch),
constexpr int kCtrlChainCu = 16;    // [8 + x] task tickets of list x, [kCtrlChainCu + i] CU of chain i (block-sparse launch)
constexpr int kMaxChains = kCtrlWords - kCtrlChainCu;
constexpr unsigned long long kTailTimeoutTicks = 300000000ull;   // 3 s

typedef unsigned v4u32_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u32_t __attribute__((ext_vector_type(2)));
typedef double v2f64_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tail_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7ffffffe, 0x00020000);
}
// agent-scope (sc1) loads: 16 B / 8 B per lane, tracked by the compiler's wait counts
__device__ __forceinline__ v2f64_t tail_ld2(__amdgpu_buffer_rsrc_t rs, int byte_off, int soff = 0) {
  return __builtin_bit_cast(v2f64_t, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, soff, 16));
}
__device__ __forceinline__ double tail_ld1(__amdgpu_buffer_rsrc_t rs, int byte_off, int soff = 0) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, soff, 16));
}
__device__ __forceinline__ void tail_st1(__amdgpu_buffer_rsrc_t rs, int byte_off, int soff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32_t, v), rs, byte_off, soff, 16);
}
__device__ __forceinline__ void tail_st2(__amdgpu_buffer_rsrc_t rs, int byte_off, int soff, v2f64_t v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32_t, v), rs, byte_off, soff, 16);
}
__device__ __forceinline__ unsigned tail_ldflag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tail_stflag(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tail_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double tail_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// all stores of this workgroup are acknowledged, then one lane raises the flag
__device__ __forceinline__ void tail_publish(unsigned* flag, unsigned epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) tail_stflag(flag, epoch);
}
__device__ __forceinline__ void tail_abort(const TailArgs& t) {
  atomicExch(t.status, 3);
  tail_stflag(&t.ctrl[1], 1u);
}
// Waits until *f0 (and *f1, if given) carry the epoch.  false = the launch was aborted.  `slot`: an int in LDS.
__device__ __forceinline__ bool tail_wait(const TailArgs& t, const unsigned* f0, const unsigned* f1, volatile int* slot) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    unsigned spins = 0;
    while (tail_ldflag(f0) != t.epoch || (f1 && tail_ldflag(f1) != t.epoch)) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63u) == 0) {
        if (tail_ldflag(&t.ctrl[1]) != 0) { ok = 0; break; }
        if (wall_clock64() - t0 > kTailTimeoutTicks) { tail_abort(t); ok = 0; break; }
      }
    }
    *slot = ok;
  }
  __syncthreads();
  const int ok = *slot;
  return ok != 0;
}
// Number of consecutive block rows k, k + 1, ... (< kend, at most 32) whose tiles (row, ca) and (row, cb) are published; waits for
// at least one.  0 = aborted.  One wavefront polls 64 flags per round (a round costs an L2 round trip, ~2 us: with 16 rows per
// round the polling alone was 10 % of a helper's time in the final launch).
__device__ __forceinline__ int tail_wait_rows(const TailArgs& t, int k, int kend, int ca, int cb, volatile int* slot) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int row = k + (lane >> 1);
    const bool in = row < kend;
    const unsigned* f = t.tile_flag + (size_t)((in ? row : k) - t.rt0) * t.ntc + ((lane & 1) ? cb : ca);
    const unsigned long long t0 = wall_clock64();
    int n = 0;
    unsigned spins = 0;
    for (;;) {
      const bool ok = in && tail_ldflag(f) == t.epoch;
      const unsigned long long m = __ballot(ok);
      const unsigned long long both = m & (m >> 1) & 0x5555555555555555ull;
      n = 0;
      while (n < 32 && ((both >> (2 * n)) & 1ull)) ++n;
      if (n > 0) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63u) == 0) {
        if (tail_ldflag(&t.ctrl[1]) != 0) break;
        if (__builtin_amdgcn_readfirstlane((int)(wall_clock64() - t0 > kTailTimeoutTicks))) { if (lane == 0) tail_abort(t); break; }
      }
    }
    if (lane == 0) *slot = n;
  }
  __syncthreads();
  const int n = *slot;
  return n;
}

// The same for a REG2 task: rows k ... kend - 1 of column blocks ca, cb AND cb + 1 (21 rows x 3 flags per polling round).
__device__ __forceinline__ int tail_wait_rows3(const TailArgs& t, int k, int kend, int ca, int cb, volatile int* slot) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int ri = lane / 3, which = lane - 3 * ri;
    const int row = k + ri;
    const bool in = lane < 63 && row < kend;
    const unsigned* f = t.tile_flag + (size_t)((in ? row : k) - t.rt0) * t.ntc + (which == 0 ? ca : cb + which - 1);
    const unsigned long long t0 = wall_clock64();
    int n = 0;
    unsigned spins = 0;
    for (;;) {
      const bool ok = in && tail_ldflag(f) == t.epoch;
      const unsigned long long m = __ballot(ok);
      n = 0;
      while (n < 21 && ((m >> (3 * n)) & 7ull) == 7ull) ++n;
      if (n > 0) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63u) == 0) {
        if (tail_ldflag(&t.ctrl[1]) != 0) break;
        if (__builtin_amdgcn_readfirstlane((int)(wall_clock64() - t0 > kTailTimeoutTicks))) { if (lane == 0) tail_abort(t); break; }
      }
    }
    if (lane == 0) *slot = n;
  }
  __syncthreads();
  const int n = *slot;
  return n;
}

// acc (64 x 64, 4 waves x 32 x 32) += sum_{k < K} (dk[k] A[k][m]) B[k][n]; A, B: K rows of `ld` doubles, written by other workgroups
// of this launch (agent-scope loads).  SYM: B == A (loaded once).  Slabs of kTailKT = 32 rows, the next one in flight while the
// MFMAs consume the current one; one s_waitcnt vmcnt(0) + barrier per slab.
// Operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write, no per-slab v_mul of the
// staged rows -- the A fragments are scaled by d_k after their ds_read (2 v_mul_f64 per 4 MFMAs).  Rounds 2-4 staged the slabs
// through registers (load, scale, ds_write): that loop sat at 48-50 TFLOP/s over the chip whatever the prefetch depth, slab height
// or cache policy (profiles/r04_helper_kloop_*); this one reaches 56-60 in the same harness with bit-identical sums
// (profiles/r04_helper_kloop_lds_dma.txt; the register-staged loop lives on in tools/bench_tail.hip as the reference).
// One DMA instruction moves 1 KiB = two 64-column rows to CONSECUTIVE LDS addresses, so slab row k sits in "pair" k & 15, half
// k >> 4, pairs 144 doubles apart: the four K rows 4 j + lk of an MFMA step then fall into both halves of the LDS banks (288 dwords
// = 32 mod 64 per pair).
constexpr int kTailKT = 32;
// The DMA is issued from inline asm: issued through the builtin, the compiler's wait-count insertion cannot tell the two stage
// buffers inside one __shared__ array apart and puts s_waitcnt vmcnt(0) in front of every ds_read (k_gemm_atb solves that with
// four separate arrays; here the two 64 x TS tiles of the chain have to stay one array).  The waits are explicit, as there.
// sm: 4 slabs of kDmaSlab doubles (A0, B0, A1, B1) + 2 x 32 doubles of d; ends with a barrier.
constexpr int kDmaPair = 2 * kInner + 16;
constexpr int kDmaSlab = (kTailKT / 2) * kDmaPair;
constexpr int kDmaDoubles = 4 * kDmaSlab + 2 * kTailKT;
// (M0 = LDS base of the DMA is written here without being declared clobbered -- the compiler rejects it as a reserved register.
// Nothing else in k_ldlt_tail may use M0.  That is enforced at BUILD time: camera_calibration_amd/build.py: check_tail_m0
// disassembles the kernel after every compile and fails the build unless every M0 access in it is one of these s_mov_b32 directly
// in front of its s_nop + global_load_lds, and no instruction with an implicit M0 operand appears; tests/test_host_hygiene.py runs
// the same check and shows that it catches a foreign M0 use.)
__device__ __forceinline__ void tail_dma16(const double* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void tail_dma4(const double* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 sc1" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ const double* tail_uniform(const double* p) {      // a wave-uniform pointer the compiler keeps in VGPRs
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const double*>(((unsigned long long)hi << 32) | lo);
}
template <bool SYM>
__device__ __forceinline__ void tail_mma_dma(v4f64 (&acc)[2][2], const double* A_, const double* B_, int ld_, const double* dk_, int K,
                                             double* sm) {
  const double* A = tail_uniform(A_);
  const double* B = tail_uniform(B_);
  const double* dk = tail_uniform(dk_);
  const int ld = __builtin_amdgcn_readfirstlane(ld_);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32, li = lane & 15, lk = lane >> 4;
  const int nk = K / kTailKT;                             // K is a multiple of 64
  const unsigned lds0 = (unsigned)(size_t)sm;
  // wavefront wv moves pairs 4 wv ... 4 wv + 3 of each operand: lanes 0-31 slab row p, lanes 32-63 slab row p + 16
  const unsigned rowb = (unsigned)ld * 8u;
  const unsigned vo = (unsigned)(4 * wv + (lane >> 5) * 16) * rowb + (unsigned)(lane & 31) * 16u;
  const unsigned la = lds0 + (unsigned)(4 * wv * kDmaPair) * 8u;
#define CBA_DSTAGE(buf_, k0_)                                                                                 \
  {                                                                                                           \
    const double* ga = A + (size_t)(k0_) * ld;                                                                \
    const double* gb = B + (size_t)(k0_) * ld;                                                                \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
      tail_dma16(ga, vo + q * rowb, la + (unsigned)((buf_) * 2 * kDmaSlab + q * kDmaPair) * 8u);              \
      if constexpr (!SYM) tail_dma16(gb, vo + q * rowb, la + (unsigned)((buf_) * 2 * kDmaSlab + kDmaSlab + q * kDmaPair) * 8u); \
    }                                                                                                         \
    if (wv == 0) tail_dma4(dk + (k0_), (unsigned)lane * 4u, lds0 + (unsigned)(4 * kDmaSlab + (buf_) * kTailKT) * 8u); \
  }
#define CBA_DOFF(j_) ((((4 * (j_)) & 15) * kDmaPair) + ((j_) >> 2) * kInner)
#define CBA_DMMA(buf_)                                                                                        \
  {                                                                                                           \
    const double* a_s = sm + (buf_) * 2 * kDmaSlab + lk * kDmaPair + wm0 + li;                                \
    const double* b_s = sm + (buf_) * 2 * kDmaSlab + (SYM ? 0 : kDmaSlab) + lk * kDmaPair + wn0 + li;         \
    const double* d_s = sm + 4 * kDmaSlab + (buf_) * kTailKT + lk;                                            \
    double af[2][2], bf[2][2], dv[2];                                                                         \
    dv[0] = d_s[0];                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) af[0][i] = a_s[CBA_DOFF(0) + i * 16];                       \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) bf[0][j] = b_s[CBA_DOFF(0) + j * 16];                       \
    _Pragma("unroll") for (int s = 0; s < kTailKT / 4; ++s) {                                                 \
      const int cur = s & 1, nxt = cur ^ 1;                                                                   \
      if (s + 1 < kTailKT / 4) {                                                                              \
        dv[nxt] = d_s[4 * (s + 1)];                                                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) af[nxt][i] = a_s[CBA_DOFF(s + 1) + i * 16];             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) bf[nxt][j] = b_s[CBA_DOFF(s + 1) + j * 16];             \
      }                                                                                                       \
      af[cur][0] *= dv[cur]; af[cur][1] *= dv[cur];                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);       \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
    }                                                                                                         \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
    __syncthreads();                                                                                          \
  }
  CBA_DSTAGE(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma nounroll
  for (int kb = 0; kb < nk; kb += 2) {
    if (kb + 1 < nk) CBA_DSTAGE(1, (kb + 1) * kTailKT);
    CBA_DMMA(0)
    if (kb + 1 < nk) {
      if (kb + 2 < nk) CBA_DSTAGE(0, (kb + 2) * kTailKT);
      CBA_DMMA(1)
    }
  }
#undef CBA_DMMA
#undef CBA_DOFF
#undef CBA_DSTAGE
}


#ifdef CBA_DEV_SWITCHES
// ---- round-5 variants of the helpers' K loop, measured in situ and NOT adopted (profiles/r05_pair_tasks_in_situ.txt, r05_ring_kloop_in_situ.txt); compiled into the
// ---- bench harness only (tools/bench_tail.hip), the product library does not contain them
// Ring variant of the 64 x 64 loop (round 5): slabs of 16 K rows in FOUR stages, three slabs in flight.  In situ the operands come
// over the fabric (L2 hit rate 27 % in the final launch, profiles/r05_tail_traffic.txt) with a latency that one slab of look-ahead
// (~1.8 us of MFMAs at two workgroups per CU) does not cover: SQ_WAIT_ANY is 24 % of the wave cycles of the final launch against 10 %
// in the bulk GEMM.  The L2-resident synthetic loop cannot show that (there the deeper ring lost 3 % to its extra barriers).
// Every wavefront issues the same number of DMA instructions per stage (the 128-byte d slab redundantly, all to the same place), so
// that `s_waitcnt vmcnt(2 x per stage)` means "my part of the oldest slab in flight has landed" for all of them.
constexpr int kRingKT = 16, kRingStages = 4;
constexpr int kRingSlab = (kRingKT / 2) * kDmaPair;       // 1152 doubles
constexpr int kRingStage = 2 * kRingSlab;
constexpr int kRingDoubles = kRingStages * kRingStage + kRingStages * 32;
template <bool SYM>
__device__ __forceinline__ void tail_mma_ring(v4f64 (&acc)[2][2], const double* A_, const double* B_, int ld_, const double* dk_, int K,
                                              double* sm) {
  const double* A = tail_uniform(A_);
  const double* B = tail_uniform(B_);
  const double* dk = tail_uniform(dk_);
  const int ld = __builtin_amdgcn_readfirstlane(ld_);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32, li = lane & 15, lk = lane >> 4;
  const int nk = K / kRingKT;                             // K is a multiple of 64: nk is a multiple of 4
  const unsigned lds0 = (unsigned)(size_t)sm;
  const unsigned rowb = (unsigned)ld * 8u;
  // wavefront wv moves pairs 2 wv, 2 wv + 1 of each operand: lanes 0-31 slab row p, lanes 32-63 slab row p + 8
  const unsigned vo = (unsigned)(2 * wv + (lane >> 5) * 8) * rowb + (unsigned)(lane & 31) * 16u;
  const unsigned la = lds0 + (unsigned)(2 * wv * kDmaPair) * 8u;
#define CBA_RSTAGE(buf_, k0_)                                                                                 \
  {                                                                                                           \
    const double* ga = A + (size_t)(k0_) * ld;                                                                \
    const double* gb = B + (size_t)(k0_) * ld;                                                                \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                           \
      tail_dma16(ga, vo + q * rowb, la + (unsigned)((buf_) * kRingStage + q * kDmaPair) * 8u);                \
      if constexpr (!SYM) tail_dma16(gb, vo + q * rowb, la + (unsigned)((buf_) * kRingStage + kRingSlab + q * kDmaPair) * 8u); \
    }                                                                                                         \
    tail_dma4(dk + (k0_), (unsigned)(lane & 31) * 4u, lds0 + (unsigned)(kRingStages * kRingStage + (buf_) * 32) * 8u); \
  }
#define CBA_ROFF(j_) ((((4 * (j_)) & 7) * kDmaPair) + ((j_) >> 1) * kInner)
#define CBA_RMMA(buf_)                                                                                        \
  {                                                                                                           \
    const double* a_s = sm + (buf_) * kRingStage + lk * kDmaPair + wm0 + li;                                  \
    const double* b_s = sm + (buf_) * kRingStage + (SYM ? 0 : kRingSlab) + lk * kDmaPair + wn0 + li;          \
    const double* d_s = sm + kRingStages * kRingStage + (buf_) * 32 + lk;                                     \
    double af[2][2], bf[2][2], dv[2];                                                                         \
    dv[0] = d_s[0];                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) af[0][i] = a_s[CBA_ROFF(0) + i * 16];                       \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) bf[0][j] = b_s[CBA_ROFF(0) + j * 16];                       \
    _Pragma("unroll") for (int s = 0; s < kRingKT / 4; ++s) {                                                 \
      const int cur = s & 1, nxt = cur ^ 1;                                                                   \
      if (s + 1 < kRingKT / 4) {                                                                              \
        dv[nxt] = d_s[4 * (s + 1)];                                                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) af[nxt][i] = a_s[CBA_ROFF(s + 1) + i * 16];             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) bf[nxt][j] = b_s[CBA_ROFF(s + 1) + j * 16];             \
      }                                                                                                       \
      af[cur][0] *= dv[cur]; af[cur][1] *= dv[cur];                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);       \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
    }                                                                                                         \
  }
  // DMA instructions per wavefront and stage: 2 (A) + 2 (B, unless SYM) + 1 (d)
#define CBA_RWAIT(kb_)                                                                                        \
  {                                                                                                           \
    if ((kb_) + 2 < nk) { if (SYM) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); } \
    else if ((kb_) + 1 < nk) { if (SYM) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); } \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    __syncthreads();                                                                                          \
  }
  CBA_RSTAGE(0, 0);
  CBA_RSTAGE(1, kRingKT);
  CBA_RSTAGE(2, 2 * kRingKT);
#pragma nounroll
  for (int kb = 0; kb < nk; kb += 4) {
    CBA_RWAIT(kb)     if (kb + 3 < nk) CBA_RSTAGE(3, (kb + 3) * kRingKT);  CBA_RMMA(0)
    CBA_RWAIT(kb + 1) if (kb + 4 < nk) CBA_RSTAGE(0, (kb + 4) * kRingKT);  CBA_RMMA(1)
    CBA_RWAIT(kb + 2) if (kb + 5 < nk) CBA_RSTAGE(1, (kb + 5) * kRingKT);  CBA_RMMA(2)
    CBA_RWAIT(kb + 3) if (kb + 6 < nk) CBA_RSTAGE(2, (kb + 6) * kRingKT);  CBA_RMMA(3)
  }
  __syncthreads();                                       // the staging area is free again
#undef CBA_RWAIT
#undef CBA_RMMA
#undef CBA_ROFF
#undef CBA_RSTAGE
}

#endif  // CBA_DEV_SWITCHES (the ring variant)

// The same loop for TWO adjacent column blocks (round 5): acc (64 x 128, 4 waves x 32 x 64) += sum_{k < K} (dk[k] A[k][m]) B[k][n],
// B 128 columns wide.  Why: the final dataflow launch is bound by what its operands cost on the FABRIC, not by the matrix pipe -- per
// dispatch PMC (profiles/r05_tail_traffic.txt): 11.1 GiB FETCH_SIZE raw = 23 GB corrected in 5.3 ms = 4.4 TB/s over the whole launch
// against ~6.3 TB/s a copy reaches, L2 hit rate 27 % (the 64 tasks on an XCD stream 65 different strips through 4 MB), MFMA-busy
// 65 %.  A 64 x 64 tile moves 2 x 64 x 8 bytes per K row for 2 x 64 x 64 flops (8 flop / byte); a 64 x 128 tile moves 3 x 64 x 8 for
// twice the flops (10.7 flop / byte): a quarter of the bytes gone, the A strip fetched once for two tiles.
// Slabs of kT2 = 16 K rows so that two stages fit next to each other in the 80 KB of a workgroup (two workgroups per CU): per
// stage A = 8 pairs of 64-column rows (as above: rows k and k + 8 share a DMA instruction), B = 16 rows of 128 columns (one DMA
// instruction each), rows 144 doubles apart (bank-conflict free for the four K rows of an MFMA step).  One barrier per 32 MFMAs of a
// wavefront, as in the 64 x 64 loop.  (Synthetic, operands L2-resident: 57-61 TFLOP/s against 56-60, tools/bench_tail.hip MMA2_ONLY.)
constexpr int kT2 = 16;                                   // slab height
constexpr int kA2Slab = (kT2 / 2) * kDmaPair;             // 1152 doubles: pairs of 64-column rows
constexpr int kB2Row = 2 * kInner + 16;                   // 144
constexpr int kB2Slab = kT2 * kB2Row;                     // 2304 doubles
constexpr int kStage2 = kA2Slab + kB2Slab;
constexpr int kDma2Doubles = 2 * kStage2 + 64;
__device__ __forceinline__ void tail_mma_dma2(v4f64 (&acc)[2][4], const double* A_, const double* B_, int ld_, const double* dk_, int K,
                                              double* sm) {
  const double* A = tail_uniform(A_);
  const double* B = tail_uniform(B_);
  const double* dk = tail_uniform(dk_);
  const int ld = __builtin_amdgcn_readfirstlane(ld_);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 64, li = lane & 15, lk = lane >> 4;
  const int nk = K / kT2;
  const unsigned lds0 = (unsigned)(size_t)sm;
  const unsigned rowb = (unsigned)ld * 8u;
  // A: wave wv moves pairs 2 wv, 2 wv + 1 (lanes 0-31 slab row p, lanes 32-63 slab row p + 8); B: rows 4 wv ... 4 wv + 3, one per instruction
  const unsigned voa = (unsigned)(2 * wv + (lane >> 5) * 8) * rowb + (unsigned)(lane & 31) * 16u;
  const unsigned vob = (unsigned)(4 * wv) * rowb + (unsigned)lane * 16u;
#define CBA_X2_STAGE(buf_, k0_)                                                                                \
  {                                                                                                            \
    const double* ga = A + (size_t)(k0_) * ld;                                                                 \
    const double* gb = B + (size_t)(k0_) * ld;                                                                 \
    const unsigned base = lds0 + (unsigned)((buf_) * kStage2) * 8u;                                            \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) tail_dma16(ga, voa + q * rowb, base + (unsigned)((2 * wv + q) * kDmaPair) * 8u); \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) tail_dma16(gb, vob + q * rowb, base + (unsigned)(kA2Slab + (4 * wv + q) * kB2Row) * 8u); \
    if (wv == 0) tail_dma4(dk + (k0_), (unsigned)lane * 4u, lds0 + (unsigned)(2 * kStage2 + (buf_) * 32) * 8u); \
  }
#define CBA_X2_AOFF(j_) ((((4 * (j_)) & 7) * kDmaPair) + ((j_) >> 1) * kInner)
#define CBA_X2_MMA(buf_)                                                                                       \
  {                                                                                                            \
    const double* a_s = sm + (buf_) * kStage2 + lk * kDmaPair + wm0 + li;                                      \
    const double* b_s = sm + (buf_) * kStage2 + kA2Slab + lk * kB2Row + wn0 + li;                              \
    const double* d_s = sm + 2 * kStage2 + (buf_) * 32 + lk;                                                   \
    double af[2][2], bf[2][4], dv[2];                                                                          \
    dv[0] = d_s[0];                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) af[0][i] = a_s[CBA_X2_AOFF(0) + i * 16];                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) bf[0][j] = b_s[j * 16];                                      \
    _Pragma("unroll") for (int s = 0; s < kT2 / 4; ++s) {                                                      \
      const int cur = s & 1, nxt = cur ^ 1;                                                                    \
      if (s + 1 < kT2 / 4) {                                                                                   \
        dv[nxt] = d_s[4 * (s + 1)];                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) af[nxt][i] = a_s[CBA_X2_AOFF(s + 1) + i * 16];           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) bf[nxt][j] = b_s[4 * (s + 1) * kB2Row + j * 16];         \
      }                                                                                                        \
      af[cur][0] *= dv[cur]; af[cur][1] *= dv[cur];                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
    __syncthreads();                                                                                           \
  }
  CBA_X2_STAGE(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma nounroll
  for (int kb = 0; kb < nk; kb += 2) {
    if (kb + 1 < nk) CBA_X2_STAGE(1, (kb + 1) * kT2);
    CBA_X2_MMA(0)
    if (kb + 1 < nk) {
      if (kb + 2 < nk) CBA_X2_STAGE(0, (kb + 2) * kT2);
      CBA_X2_MMA(1)
    }
  }
#undef CBA_X2_MMA
#undef CBA_X2_AOFF
#undef CBA_X2_STAGE
}

// ticket of list x -> task.  kind 0 = PRE(r), 1 = PART(r + 1), 2 = REG(r, c).  List x (of `nl` lists) holds the tasks whose column
// block c has c % nl == x, rows in increasing order -- a task only waits for tiles of earlier rows, so every list is in
// dependency order and the launch makes progress as long as each list's pending head is held by a running workgroup or nobody
// is left to wait for it (workgroups steal from the other lists once their own is empty).
__device__ __forceinline__ int tail_row_count(const TailArgs& t, int r, int x, int nl) {
  // columns c in [r + 1, ntc) with c % nl == x; column r + 1 counts twice (PRE + PART) while r + 1 is a row of the tail
  const int lo = r + 1;
  const int first = lo + ((x - lo % nl) + nl) % nl;
  int cnt = first < t.ntc ? (t.ntc - 1 - first) / nl + 1 : 0;
  if (r + 1 < t.nr && (r + 1) % nl == x) cnt += 1;
  return cnt;
}
#ifdef CBA_DEV_SWITCHES
// Pair mode (t.pair, one list): row r hands out PRE(r), PART(r + 1), the REG tasks of the first kTailNearSingles columns right of
// them one tile at a time (they feed the chain's next steps: latency matters), then REG2 tasks (kind 3) of two adjacent column blocks
// each, and a last single tile when the count is odd.  The last block row (nothing below it) keeps single tiles.
constexpr int kTailNearSingles = 2;
__host__ __device__ inline int tail_pair_row_count(int rt0, int nr, int ntc, int r) {
  (void)rt0;
  if (r + 1 >= nr) return ntc - nr;
  const int ncols = ntc - r - 2;
  const int s1 = ncols < kTailNearSingles ? ncols : kTailNearSingles;
  const int rem = ncols - s1;
  return 2 + s1 + rem / 2 + (rem & 1);
}
__device__ __forceinline__ void tail_task_pair(const TailArgs& t, int ticket, int* kind, int* r_out, int* c_out) {
  int r = t.rt0;
  for (; r < t.nr; ++r) {
    const int cnt = tail_pair_row_count(t.rt0, t.nr, t.ntc, r);
    if (ticket < cnt) break;
    ticket -= cnt;
  }
  *r_out = r;
  if (r + 1 >= t.nr) { *kind = 2; *c_out = r + 1 + ticket; return; }
  if (ticket < 2) { *kind = ticket; *c_out = r + 1; return; }
  int q = ticket - 2;
  const int ncols = t.ntc - r - 2;
  const int s1 = ncols < kTailNearSingles ? ncols : kTailNearSingles;
  if (q < s1) { *kind = 2; *c_out = r + 2 + q; return; }
  q -= s1;
  const int rem = ncols - s1;
  if (q < rem / 2) { *kind = 3; *c_out = r + 2 + s1 + 2 * q; return; }
  *kind = 2; *c_out = t.ntc - 1;
}
#endif
__device__ __forceinline__ void tail_task(const TailArgs& t, int ticket, int x, int nl, int* kind, int* r_out, int* c_out) {
#ifdef CBA_DEV_SWITCHES
  if (t.pair) { tail_task_pair(t, ticket, kind, r_out, c_out); return; }
#endif
  int r = t.rt0;
  for (; r < t.nr; ++r) {
    const int cnt = tail_row_count(t, r, x, nl);
    if (ticket < cnt) break;
    ticket -= cnt;
  }
  *r_out = r;
  const int lo = r + 1;
  const int first = lo + ((x - lo % nl) + nl) % nl;
  if (r + 1 < t.nr && first == r + 1) {
    if (ticket == 0) { *kind = 0; *c_out = r + 1; return; }
    if (ticket == 1) { *kind = 1; *c_out = r + 1; return; }
    *kind = 2; *c_out = first + (ticket - 1) * nl;          // ticket 2 -> first + nl
    return;
  }
  *kind = 2; *c_out = first + ticket * nl;
}

__device__ __forceinline__ unsigned tail_cu_id() {
  const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7u;          // HW_REG_XCC_ID
  const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);                // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
  return 0x80000000u | (xcc << 8) | ((hw >> 8) & 0xffu);
}

// developer timeline of the chain workgroup (tools/bench_tail.hip, -DCBA_TAILLOG): 100 MHz stamps per block and phase
#ifdef CBA_TAILLOG
__device__ unsigned long long* g_helplog = nullptr;   // per helper task (ticket): start, wait-rows ticks, k-loop ticks, diag-wait ticks, end, kind, r, c
#define HELP_NOW() (g_helplog ? wall_clock64() : 0ull)
__device__ unsigned long long* g_taillog = nullptr;
#define TAIL_STAMP(blk_, ph_) do { __builtin_amdgcn_sched_barrier(0); if (g_taillog && threadIdx.x == 0) g_taillog[(size_t)(blk_) * 16 + (ph_)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TAIL_STAMP(blk_, ph_) do { } while (0)
#define HELP_NOW() 0ull
#endif

// tile_mma_lds with an A operand that is a transposed unit-lower-triangular inverse carrying junk in the 16 x 16 tiles below
// its block diagonal (chain_factor_blocked): row block I of the result takes the k blocks <= I only
__device__ __forceinline__ void tile_mma_lds_lowerA(v4f64 (&acc)[2][2], const double* Al, const double* Bl) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32, li = lane & 15, lk = lane >> 4;
  const int mb0 = wm0 >> 4;
#pragma unroll
  for (int kk = 0; kk < kInner; kk += 4) {
    const int kb = kk >> 4;
    if (kb <= mb0 + 1) {
      double bf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = Bl[(kk + lk) * TS + wn0 + j * 16 + li];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (kb <= mb0 + i) {
          const double af = Al[(kk + lk) * TS + wm0 + i * 16 + li];
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf[j], acc[i][j], 0, 0, 0);
        }
    }
  }
}

// ---- blocked 64 x 64 LDL^T of the chain (round 4) ----
// The 64 pivots of a diagonal block used to be 32 barrier rounds of the whole workgroup (ldlt_diag_core: two pivots per
// LDS round trip + barrier, 447 ns per round = 14.3 us per block, the largest item on the critical path of the whole
// factorisation).  Here the block is factored in four panels of 16 columns:
//   * ONE wavefront (wave 0) factors a panel with no barrier and no LDS traffic inside: lane i holds row i of the panel
//     (16 registers), pivot row entries are broadcast with v_readlane into SGPRs, the 16 steps are fully unrolled;
//   * the other three wavefronts apply the panel to the rest of the block with v_mfma_f64_16x16x4 (rank-16 updates of
//     16 x 16 tiles, accumulators kept in registers across panels) and build L^-1 in product form
//     (X <- E_p X per panel: 16 x 16 unit-triangular inverses by forward substitution with LDS-broadcast operands,
//     everything else MFMA), off the pivot path: after the last pivot only M_33 and one more product level remain.
// Two barriers per panel instead of sixteen.  LDS (two 64 x TS tiles, as before):
//   sW  upper triangle (row <= col, 16 x 16 tiles (J, I), J <= I): the working matrix in upper storage W(i, j) at [j][i];
//       the rows of a factored panel hold d l (the K-major A operand of the updates); tiles are replaced by the transposed
//       inverse M^T ([q][p] = M(p, q), the layout the next chain step and the helpers multiply with) once they are dead.
//       strictly lower tiles (I, J), I > J: the inverse being built, natural layout [p][q] (B operand of its own updates);
//       junk afterwards -- consumers skip them / the global store writes zeros.
//   sV  L^T with d on the diagonal ([j][i] = L(i, j), i > j; zeros below): the tile that goes to S, and the B operand of the
//       updates.  Padding columns 64 .. 79 of rows 16 p .. 16 p + 15: the natural copy of M_pp (B operand).
//   s_rd (padding of sW rows 0 .. 3): 1 / d.
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
// Panel P (columns 16 P .. 16 P + 15) by one wavefront.  Returns true when a pivot was zero or NaN.
// No per-column lane masks in here (the compiler hoists every `lane > base + c` comparison out of the chain's block loop as an
// SGPR pair and then spills them: 546 SGPR spills, two v_readlane reloads per use): lanes above the diagonal of the panel's
// own 16 x 16 block carry junk through the loop -- their results land below the diagonal of sV, which nothing reads and the
// global store masks -- and the pivots are collected per lane with v_writelane.
template <int P>
__device__ __forceinline__ bool chain_panel(double* sW, double* sV, double* s_rd, int lane_in) {
  constexpr int base = 16 * P;
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = sW[(base + c) * TS + lane];        // W(lane, base + c); meaningful for lane >= base + c
  bool bad = false;
  int dlo = 0, dhi = 0;                                                  // lane base + c: d_c
  const bool below = lane >= base + 16;
  // the pivot of step c + 1 and its reciprocal are started as soon as column c + 1 has its update of step c, underneath the
  // rest of that step's updates (the reciprocal is a chain of five dependent fp64 operations)
  int slo = __builtin_amdgcn_readlane(__double2loint(a[0]), base);
  int shi = __builtin_amdgcn_readlane(__double2hiint(a[0]), base);
  double inv = pivot_rcp(__hiloint2double(shi, slo));
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(dlo) : "s"(slo), "n"(base + c));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(dhi) : "s"(shi), "n"(base + c));
    if (!(fabs(__hiloint2double(shi, slo)) > 0.0)) bad = true;
    const double l = a[c] * inv;
    if (c + 1 < 16) {
      const double v = readlane_f64(a[c], base + c + 1);
      // (round 5 measured the alternative -- the next pivot as d' = W(c+1, c+1) - v^2 / d on a dependency chain of its own, one fused
      // multiply-add behind 1 / d: panels 1.64-1.73 -> 1.79-1.90 us; with ONE wavefront issuing, the extra instructions cost more than the
      // shorter dependency chain saves: profiles/r05_pivot_chain.txt)
      a[c + 1] = __builtin_fma(-l, v, a[c + 1]);
      slo = __builtin_amdgcn_readlane(__double2loint(a[c + 1]), base + c + 1);
      shi = __builtin_amdgcn_readlane(__double2hiint(a[c + 1]), base + c + 1);
      inv = pivot_rcp(__hiloint2double(shi, slo));
    }
#pragma unroll
    for (int c2 = c + 2; c2 < 16; ++c2) {
      const double v = readlane_f64(a[c], base + c2);                    // d l_{c2}: the column entry before scaling
      a[c2] = __builtin_fma(-l, v, a[c2]);
      // (left alone, the scheduler hoists every v_readlane of the panel to the top and spills the SGPRs it cannot hold)
      if (((c2 - c) & 7) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    // d l of the rows below the panel back in place (the updates' A operand), L^T into sV.  Lanes left of the panel
    // hold the inverse being built in sW (lower tiles): they must not write there.
    // (branch-free: a branch here splits the panel into basic blocks and the updates get sunk towards their uses, with every
    // broadcast SGPR pair alive until then; the other lanes store to the sV slot that the next store overwrites)
    if (P < 3) { double* dst = below ? &sW[(base + c) * TS + lane] : &sV[(base + c) * TS + lane]; *dst = a[c]; }
    sV[(base + c) * TS + lane] = l;
    __builtin_amdgcn_sched_barrier(0);
  }
  if ((unsigned)(lane - base) < 16u) {
    const double dv = __hiloint2double(dhi, dlo);
    sV[lane * TS + lane] = dv;                                           // d on the diagonal
    s_rd[P * TS + lane - base] = pivot_rcp(dv);
  }
  return bad;
}
// 16 x 16 tile helpers; li = lane & 15, lk = lane >> 4.  MFMA result layout: element (lk + 4 r, li) in component r.
__device__ __forceinline__ void mma16(v4f64& acc, const double* Ak, const double* Bk, int li, int lk) {
  // acc[m][n] += sum_{k < 16} Ak[k][m] Bk[k][n]   (both K-major, row stride TS)
#pragma unroll
  for (int kk = 0; kk < 16; kk += 4)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ak[(kk + lk) * TS + li], Bk[(kk + lk) * TS + li], acc, 0, 0, 0);
}
__device__ __forceinline__ v4f64 ld16(const double* t, int li, int lk) {
  v4f64 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = t[(lk + 4 * r) * TS + li];
  return v;
}
__device__ __forceinline__ void st16(double* t, v4f64 v, int li, int lk) {
#pragma unroll
  for (int r = 0; r < 4; ++r) t[(lk + 4 * r) * TS + li] = v[r];
}
__device__ __forceinline__ void st16_t(double* t, v4f64 v, int li, int lk) {       // transposed
#pragma unroll
  for (int r = 0; r < 4; ++r) t[li * TS + lk + 4 * r] = v[r];
}
// M_pp = L_pp^-1 (16 x 16, unit lower): lane j (of every group of 16) carries column j through the forward substitution;
// the entries of L are wave-uniform LDS reads (broadcast).  Natural copy -> padding of sV, transposed copy (complete tile:
// zeros below its diagonal) -> diagonal tile of sW.
__device__ __forceinline__ void chain_inv16(double* sW, double* sV, int p, int lane_in) {
  int lane = lane_in;
  asm volatile("" : "+v"(lane));                             // (keeps the 16 comparisons below inside the chain's block loop)
  const int j = lane & 15, base = 16 * p;
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (i == j) ? 1.0 : 0.0;
  // right-looking order: the 15 - k updates of step k are independent of each other.  (Pinning that order with a scheduling
  // barrier per step cost 93 spilled registers in the launch and gained 0.1 us.)
#pragma unroll
  for (int k = 0; k < 15; ++k) {
#pragma unroll
    for (int i = k + 1; i < 16; ++i) x[i] = __builtin_fma(-sV[(base + k) * TS + base + i], x[k], x[i]);
  }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sV[(base + i) * TS + kInner + j] = x[i];               // natural: M(i, j)
      sW[(base + j) * TS + base + i] = x[i];                 // transposed: [q = j][p = i]
    }
  }
}
// out = -(A B) (FIRST) or Cin - A B, A = L_Ip (from sV), B natural; result natural -> sW lower tile (I, J)
__device__ __forceinline__ v4f64 chain_xupd(const double* sV, const double* Bk, int I, int p, const double* cin, int li, int lk) {
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  mma16(acc, sV + 16 * p * TS + 16 * I, Bk, li, lk);
  if (cin) { const v4f64 c = ld16(cin, li, lk); return c - acc; }
  return -acc;
}
// The whole block.  On entry sW holds T (upper triangle valid); on exit sV / sW / s_rd as described above.  All 256 lanes.
#ifdef CBA_DIAGLOG
__device__ unsigned long long* g_diaglog = nullptr;   // tools/bench_diag.hip: accumulated 100 MHz ticks per phase boundary
#define CHAIN_PH(n_) do { __builtin_amdgcn_sched_barrier(0); if (g_diaglog && threadIdx.x == 0) g_diaglog[n_] += wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CHAIN_PH(n_) do { } while (0)
#endif
__device__ __forceinline__ bool chain_factor_blocked(double* sW, double* sV, double* s_rd) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
  bool bad = false;
#define SW_T(J_, I_) (sW + 16 * (J_) * TS + 16 * (I_))
#define SV_PAD(p_) (sV + 16 * (p_) * TS + kInner)
  // update of tile (J, I) by panel P: acc += (d l)(rows J) L(rows I)^T
#define T_UPD(acc_, P_, J_, I_) mma16(acc_, sW + 16 * (P_) * TS + 16 * (J_), sV + 16 * (P_) * TS + 16 * (I_), li, lk)
  const v4f64 zero = {0.0, 0.0, 0.0, 0.0};
  v4f64 keep = zero;                                   // the tile this wave carries across panels: w1 (2,2), w2 (2,3), w3 (3,3)
  CHAIN_PH(0);
  // A0
  if (wv == 0) { bad |= chain_panel<0>(sW, sV, s_rd, lane); CHAIN_PH(10); }
  __syncthreads();
  CHAIN_PH(1);
  // C0: row 1 of the block
  if (wv >= 1) {
    v4f64 acc = zero;
    T_UPD(acc, 0, 1, wv);
    double* t = SW_T(1, wv);
    st16(t, ld16(t, li, lk) - acc, li, lk);
  }
  __syncthreads();
  CHAIN_PH(2);
  // A1
  if (wv == 0) { bad |= chain_panel<1>(sW, sV, s_rd, lane); CHAIN_PH(11); }
  else {
    if (wv == 1) { T_UPD(keep, 0, 2, 2); chain_inv16(sW, sV, 0, lane); }
    else if (wv == 2) T_UPD(keep, 0, 2, 3);
    else T_UPD(keep, 0, 3, 3);
  }
  __syncthreads();
  CHAIN_PH(3);
  // C1: row 2 of the block (w3 keeps (3,3))
  if (wv == 1) { T_UPD(keep, 1, 2, 2); double* t = SW_T(2, 2); st16(t, ld16(t, li, lk) - keep, li, lk); }
  else if (wv == 2) { T_UPD(keep, 1, 2, 3); double* t = SW_T(2, 3); st16(t, ld16(t, li, lk) - keep, li, lk); }
  else if (wv == 3) T_UPD(keep, 1, 3, 3);
  __syncthreads();
  CHAIN_PH(4);
  // A2: inverse, panel 0 applied: X_I0 = -L_I0 M_00
  if (wv == 0) { bad |= chain_panel<2>(sW, sV, s_rd, lane); CHAIN_PH(12); }
  else if (wv == 1) chain_inv16(sW, sV, 1, lane);
  else if (wv == 2) {
    st16(SW_T(1, 0), chain_xupd(sV, SV_PAD(0), 1, 0, nullptr, li, lk), li, lk);
    st16(SW_T(2, 0), chain_xupd(sV, SV_PAD(0), 2, 0, nullptr, li, lk), li, lk);
  } else st16(SW_T(3, 0), chain_xupd(sV, SV_PAD(0), 3, 0, nullptr, li, lk), li, lk);
  __syncthreads();
  CHAIN_PH(5);
  // C2: tile (3,3); M_10 = M_11 X_10; X_21 = -L_21 M_11, X_31 = -L_31 M_11
  if (wv == 3) { T_UPD(keep, 2, 3, 3); double* t = SW_T(3, 3); st16(t, ld16(t, li, lk) - keep, li, lk); }
  else if (wv == 1) {
    v4f64 m = zero;
    mma16(m, SW_T(1, 1), SW_T(1, 0), li, lk);
    st16(SW_T(1, 0), m, li, lk);
    st16_t(SW_T(0, 1), m, li, lk);
  } else if (wv == 2) {
    st16(SW_T(2, 1), chain_xupd(sV, SV_PAD(1), 2, 1, nullptr, li, lk), li, lk);
    st16(SW_T(3, 1), chain_xupd(sV, SV_PAD(1), 3, 1, nullptr, li, lk), li, lk);
  }
  __syncthreads();
  CHAIN_PH(6);
  // A3: last panel and its inverse on wave 0; M_22; panel 1 applied to column 0 of the inverse
  if (wv == 0) { bad |= chain_panel<3>(sW, sV, s_rd, lane); CHAIN_PH(13); chain_inv16(sW, sV, 3, lane); CHAIN_PH(14); }
  else if (wv == 1) chain_inv16(sW, sV, 2, lane);
  else if (wv == 2) st16(SW_T(2, 0), chain_xupd(sV, SW_T(1, 0), 2, 1, SW_T(2, 0), li, lk), li, lk);
  else st16(SW_T(3, 0), chain_xupd(sV, SW_T(1, 0), 3, 1, SW_T(3, 0), li, lk), li, lk);
  __syncthreads();
  CHAIN_PH(7);
  // E1: M_20 = M_22 X_20, M_21 = M_22 X_21, X_32 = -L_32 M_22
  if (wv == 1) {
    v4f64 m = zero;
    mma16(m, SW_T(2, 2), SW_T(2, 0), li, lk);
    st16(SW_T(2, 0), m, li, lk);
    st16_t(SW_T(0, 2), m, li, lk);
  } else if (wv == 2) {
    v4f64 m = zero;
    mma16(m, SW_T(2, 2), SW_T(2, 1), li, lk);
    st16(SW_T(2, 1), m, li, lk);
    st16_t(SW_T(1, 2), m, li, lk);
  } else if (wv == 3) st16(SW_T(3, 2), chain_xupd(sV, SV_PAD(2), 3, 2, nullptr, li, lk), li, lk);
  __syncthreads();
  CHAIN_PH(8);
  // E2: X_3J -= L_32 M_2J, then M_3J = M_33 X_3J (the wave's own tile goes through LDS to become a B operand)
  if (wv >= 1) {
    const int J = wv - 1;
    double* x = SW_T(3, J);
    if (J < 2) st16(x, chain_xupd(sV, SW_T(2, J), 3, 2, x, li, lk), li, lk);
    v4f64 m = zero;
    mma16(m, SW_T(3, 3), x, li, lk);
    st16_t(SW_T(J, 3), m, li, lk);
  }
  __syncthreads();
  CHAIN_PH(9);
#undef T_UPD
#undef SV_PAD
#undef SW_T
  return bad;
}

// ---- the chain workgroup ----
// Tile I/O goes through buffer instructions with ONE per-lane offset register (voffset) and a wave-uniform offset (soffset, an
// SGPR): with 64-bit global addresses the compiler kept 16 loop-invariant address pairs per tile alive across the pivot loop,
// spilled them, and every load then waited for a scratch reload AND the previous load (6 us for one tile).  Row-wise tile
// traffic moves full 512-byte rows per half wave (16 bytes per lane at a stride of 128 bytes -- 64 partial lines per
// instruction -- made the agent-scope stores of one tile take 12 us).
//
// Per block (measured, tools/bench_tail.hip -DCBA_TAILLOG, us): flags 0.65, U / P loads 1.4, X = invL U 2.0, its epilogue 1.0,
// T product 2.0, T to LDS + publication of L_{r-1,r} 0.75, T to registers 0.35, 64 pivots 14.7, epilogue 5.0 = 27.9.  Measured
// and dropped: polling / loading the next step's operands underneath the pivots from a hook in the pivot loop (per pair: pivots
// 14.3 -> 16.6 us; between the 16-step segments: 15.5-17 us) or inside the epilogue (32 us per block) -- the extra live state
// spills, and every spill reload waits for vmcnt(0), i.e. for the write-through acknowledgement of the stores in flight.
// Block rows [r_begin, r_end) (the dense launch: the whole tail).  start_dep: block r_begin has predecessors in the launch -- its
// diagonal tile arrives through a PARTFULL task (block-sparse launch: the chain of a camera's separators).
__device__ __forceinline__ void tail_chain(const TailArgs& t, double* sV, double* sW, const int r_begin, const int r_end, const int start_dep,
                                           unsigned* cu_word) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32, li = lane & 15, lk = lane >> 4;
  double* s_rd = sW + kInner;                     // 1 / d of the block factored last: padding columns of rows 0 .. 3 of sW
  volatile int* slot = reinterpret_cast<volatile int*>(sW + 4 * TS + kInner);            // padding of row 4
  const int ld = t.ld;
  const int acc_voff = ((wm0 + lk) * ld + wn0 + li) * 8;       // accumulator layout: element (i, jj, r4) at + ((16 i + 4 r4) ld + 16 jj) * 8
  // row-wise layout: instruction k of a wave moves rows rw + 2 k (lanes 0-31) and rw + 2 k + 1 (lanes 32-63), 16 bytes per lane
  const int rw = 16 * wv + (lane >> 5), cw = 2 * (lane & 31);
  const int u_voff = (rw * ld + cw) * 8;
  if (tid == 0 && t.evict) tail_stflag(cu_word, tail_cu_id());
  for (int r = r_begin; r < r_end; ++r) {
    const int j0 = kInner * r;
    const int b = r - t.rt0;
    TAIL_STAMP(b, 0);
    if (r > r_begin) {
      const __amdgpu_buffer_rsrc_t ru = tail_rsrc(t.S + (size_t)(j0 - kInner) * ld + j0);
      if (!tail_wait(t, &t.upre_flag[b - 1], &t.part_flag[b], slot)) return;
      TAIL_STAMP(b, 1);
      // U_{r-1,r} -> sV; P_r -> registers (accumulator layout): all 24 loads of a lane in flight together
      const __amdgpu_buffer_rsrc_t rp = tail_rsrc(t.S + (size_t)j0 * ld + j0);
      v2f64_t u[8];
      v4f64 P[2][2];
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = tail_ld2(ru, u_voff, 2 * k * ld * 8);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) P[i][jj][r4] = tail_ld1(rp, acc_voff, ((16 * i + 4 * r4) * ld + 16 * jj) * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) { sV[(rw + 2 * k) * TS + cw] = u[k].x; sV[(rw + 2 * k) * TS + cw + 1] = u[k].y; }
      // everything this lane stored in the previous block's epilogue is acknowledged by now: the barrier below completes the
      // publication of block r - 1 at no cost on the chain
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) tail_stflag(&t.diag_flag[b - 1], t.epoch);
      TAIL_STAMP(b, 2);
      v4f64 X[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) X[i][jj] = (v4f64){0.0, 0.0, 0.0, 0.0};
      tile_mma_lds_lowerA(X, sW, sV);            // X[p][n] = sum_q invLt[q][p] U[q][n]  (sW carries junk below its block diagonal)
      __syncthreads();                           // every wave is done reading sW (inverse) and sV (U)
      TAIL_STAMP(b, 3);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int p = wm0 + i * 16 + lk + 4 * r4;
          const double rd = s_rd[(p >> 4) * TS + (p & 15)];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int n = wn0 + jj * 16 + li;
            const double x = X[i][jj][r4], l = x * rd;
            sW[p * TS + n] = l;
            sV[p * TS + n] = x;
            tail_st1(ru, acc_voff, ((16 * i + 4 * r4) * ld + 16 * jj) * 8, l);          // L_{r-1,r}
          }
        }
      __syncthreads();
      TAIL_STAMP(b, 4);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) X[i][jj] = (v4f64){0.0, 0.0, 0.0, 0.0};
      tile_mma_lds(X, sW, sV);                   // sum_p L[p][m] X[p][n]
      __syncthreads();
      TAIL_STAMP(b, 5);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int m = wm0 + i * 16 + lk + 4 * r4, n = wn0 + jj * 16 + li;
            sW[m * TS + n] = P[i][jj][r4] - X[i][jj][r4];
          }
      tail_publish(&t.tile_flag[(size_t)(b - 1) * t.ntc + r], t.epoch);       // L_{r-1,r}; its barrier also covers the tile in sW
      TAIL_STAMP(b, 6);
    } else {
      // first block of the chain: nothing to subtract, or (start_dep) everything subtracted by a PARTFULL task of this launch
      if (start_dep && !tail_wait(t, &t.part_flag[b], nullptr, slot)) return;
      for (int e = tid; e < kInner * kInner; e += 256) {
        const int m = e >> 6, n = e & 63;
        const double* src = t.S + (size_t)(j0 + m) * ld + j0 + n;
        sW[m * TS + n] = start_dep ? tail_ld(src) : *src;
      }
      __syncthreads();
    }
    {
      // four panels of 16 columns: wave 0 pivots in registers, waves 1-3 update with MFMA and build the inverse
      TAIL_STAMP(b, 7);
      const bool bad = chain_factor_blocked(sW, sV, s_rd);
      TAIL_STAMP(b, 8);
      if (bad && tid == 0) atomicExch(t.status, 2);
      // sV = L^T / d and sW = transposed inverse are complete tiles in LDS (behind the factorisation's last barrier): both leave
      // row by row, full 512-byte rows per half wave.  The tiles below the block diagonal of sW hold working copies of the
      // inverse; zeros go to memory in their place.
      const int rw2 = 16 * wv + (lane >> 5), cw2 = 2 * (lane & 31);
      const int s_voff = (rw2 * ld + cw2) * 8, i_voff = (rw2 * kInner + cw2) * 8;
      const __amdgpu_buffer_rsrc_t rs = tail_rsrc(t.S + (size_t)j0 * ld + j0);
      const __amdgpu_buffer_rsrc_t ri = tail_rsrc(t.invLt + (size_t)r * kInner * kInner);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int row = rw2 + 2 * k;
        v2f64_t v, w;
        v.x = sV[row * TS + cw2]; v.y = sV[row * TS + cw2 + 1];
        w.x = sW[row * TS + cw2]; w.y = sW[row * TS + cw2 + 1];
        if ((cw2 >> 4) < (row >> 4)) { w.x = 0.0; w.y = 0.0; }
        if (cw2 < row) v.x = 0.0;                             // below the diagonal of sV: junk of the panel loops
        if (cw2 + 1 < row) v.y = 0.0;
        tail_st2(rs, s_voff, 2 * k * ld * 8, v);
        tail_st2(ri, i_voff, 2 * k * kInner * 8, w);
      }
      if (tid < kInner) {
        const __amdgpu_buffer_rsrc_t rv = tail_rsrc(t.dvec + j0);
        tail_st1(rv, tid * 8, 0, sV[tid * TS + tid]);
      }
    }
    TAIL_STAMP(b, 9);
    // published by the next step (after its loads) -- or here, for the last block
    if (r + 1 == r_end) tail_publish(&t.diag_flag[b], t.epoch);
  }
}

// ---- helper workgroups ----
// REG2 task (round 5): tiles (r, c) and (r, c + 1) in one go -- the K loop on the 64 x 128 tile (tail_mma_dma2: the A strip is
// fetched once for both), then the 64 x 64 epilogue of a REG task twice with ONE load of invL_r.  false = the launch was aborted.
// Round 6: the border tiles of the block-sparse launch (k_ldlt_sparse) are REG2 tasks -- a 128-column pair is exactly the unit of the
// row strips' activity, and what the launch runs out of with several pivot chains is workgroup SLOTS: at the frontier of every chain
// one task per border column block is waiting for that chain's next diagonal block (4 chains x 133 column blocks at BASELINE
// configs[2] against 2 x 256 slots).  ivals / n_iv: the K intervals of the task (null: [rt0, r)); arow: the activity bits of the pair's
// 128-column tile (K rows whose tiles do not exist are skipped) or null.
__device__ __forceinline__ bool tail_helper_pair(const TailArgs& t, double* sV, double* sAB, int r, int c, volatile int* slot,
                                                 const GfIval* ivals, int n_iv, const unsigned long long* arow) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int ld = t.ld;
  static_assert(kDma2Doubles <= kInner * TS + (kInner - 1) * TS + kInner, "the 64 x 128 K-loop staging overruns the slots");
  v4f64 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  for (int iv = 0; iv < n_iv; ++iv) {
    int k = t.rt0, kend = r;
    if (ivals) { k = __builtin_amdgcn_readfirstlane(ivals[iv].k0); kend = __builtin_amdgcn_readfirstlane(ivals[iv].k1); }
    while (k < kend) {
      int run_end = kend;
      if (arow) {
        k = bits_next(arow, k, kend, true);
        if (k >= kend) break;
        run_end = bits_next(arow, k, kend, false);
      }
      const int nrows = tail_wait_rows3(t, k, run_end, r, c, slot);
      if (nrows <= 0) return false;
      const double* A = t.S + (size_t)k * kInner * ld + (size_t)r * kInner;
      const double* B = t.S + (size_t)k * kInner * ld + (size_t)c * kInner;
      tail_mma_dma2(acc, A, B, ld, t.dvec + (size_t)k * kInner, nrows * kInner, sV);
      k += nrows;
    }
  }
  // accumulator layout of the 64 x 128 tile: wave wv holds rows 32 (wv >> 1) + 16 i + lk + 4 r4, columns 64 (wv & 1) + 16 j + li,
  // i.e. waves 0 / 2 hold tile (r, c) and waves 1 / 3 hold tile (r, c + 1)
  const int wm0 = (wv >> 1) * 32, half = wv & 1;
  {
    // U = A_rc - acc, straight into the accumulator registers (the tiles were written before this launch: fetched now, one round trip
    // per task; prefetching them underneath the K loop would cost 64 more live registers)
    const __amdgpu_buffer_rsrc_t rt = tail_rsrc(t.S + (size_t)r * kInner * ld + (size_t)(c + half) * kInner);
    const int voff = ((wm0 + lk) * ld + li) * 8;
    double a[2][4][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) a[i][j][r4] = tail_ld1(rt, voff, ((16 * i + 4 * r4) * ld + 16 * j) * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) acc[i][j][r4] = a[i][j][r4] - acc[i][j][r4];
  }
  if (!tail_wait(t, &t.diag_flag[r - t.rt0], nullptr, slot)) return false;     // (its barrier: every wave is done with the K-loop staging)
  // invL_r (K-major, [q][p]) -> sAB; 1 / d_r of the rows of the 64 x 64 product layout
  const int pm0 = (wv >> 1) * 32, pn0 = (wv & 1) * 32;                         // 64 x 64 product: 4 waves x 32 x 32
  double rdr[2][4];
  {
    const __amdgpu_buffer_rsrc_t ri = tail_rsrc(t.invLt + (size_t)r * kInner * kInner);
    const __amdgpu_buffer_rsrc_t rd = tail_rsrc(t.dvec + (size_t)r * kInner);
    const int rw = 16 * wv + (lane >> 5), cw = 2 * (lane & 31);
    v2f64_t u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] = tail_ld2(ri, (rw * kInner + cw) * 8, 2 * k * kInner * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) rdr[i][r4] = tail_ld1(rd, (pm0 + lk) * 8, (16 * i + 4 * r4) * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) { sAB[(rw + 2 * k) * TS + cw] = u[k].x; sAB[(rw + 2 * k) * TS + cw + 1] = u[k].y; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) rdr[i][r4] = 1.0 / rdr[i][r4];
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h == 1) __syncthreads();                 // every wave is done reading tile 0's U from sV
    if (half == h) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) sV[(wm0 + 16 * i + lk + 4 * r4) * TS + 16 * j + li] = acc[i][j][r4];
    }
    __syncthreads();                             // U (and, for h = 0, invL_r) complete in LDS
    v4f64 x[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) x[i][jj] = (v4f64){0.0, 0.0, 0.0, 0.0};
    tile_mma_lds(x, sAB, sV);                    // X[p][n] = sum_q invLt[q][p] U[q][n]
    const __amdgpu_buffer_rsrc_t rt = tail_rsrc(t.S + (size_t)r * kInner * ld + (size_t)(c + h) * kInner);
    const int acc_voff = ((pm0 + lk) * ld + pn0 + li) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          tail_st1(rt, acc_voff, ((16 * i + 4 * r4) * ld + 16 * jj) * 8, x[i][jj][r4] * rdr[i][r4]);
    if (t.X && c + h >= t.x_c0) {
      double* Xt = t.X + (size_t)(r - t.rt0) * kInner * t.ldx + (size_t)(c + h) * kInner;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            Xt[(size_t)(pm0 + i * 16 + lk + 4 * r4) * t.ldx + pn0 + jj * 16 + li] = x[i][jj][r4];
    }
  }
  // both tiles with one acknowledgement wait: tail_publish = vmcnt(0) + barrier + flag store by one lane
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    tail_stflag(&t.tile_flag[(size_t)(r - t.rt0) * t.ntc + c], t.epoch);
    tail_stflag(&t.tile_flag[(size_t)(r - t.rt0) * t.ntc + c + 1], t.epoch);
  }
  return true;
}

// SPARSE (k_ldlt_sparse): tasks come from the plan's two lists (list 0 = what the chains wait for; the first n_critical helper roles
// serve it first, everybody else list 1 first), every task carries its K intervals, several chains own a CU each.
template <bool SPARSE>
__device__ __forceinline__ void tail_helper(const TailArgs& t, double* sV, double* sAB, const int role) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32, li = lane & 15, lk = lane >> 4;
  // K-loop staging (LDS-DMA): kDmaDoubles from the start of sV, running over into sAB (the two tiles are one array); the slots
  // sit behind it, in the padding of sAB's last row
  static_assert(kDmaDoubles <= kInner * TS + (kInner - 1) * TS + kInner, "the K-loop staging overruns the slots");
#ifdef CBA_DEV_SWITCHES
  static_assert(kRingDoubles <= kInner * TS + (kInner - 1) * TS + kInner, "the ring K-loop staging overruns the slots");
#endif
  volatile int* slot = reinterpret_cast<volatile int*>(sAB + (kInner - 1) * TS + kInner);
  volatile int* slot2 = reinterpret_cast<volatile int*>(sAB + (kInner - 1) * TS + kInner + 2);
  volatile int* slot3 = reinterpret_cast<volatile int*>(sAB + (kInner - 1) * TS + kInner + 4);
  const int ld = t.ld;
  const unsigned my_cu = tail_cu_id();
  const int nl = SPARSE ? 2 : (t.xcd_lists ? 8 : 1);
  const int my_list = SPARSE ? (role - t.n_chains < t.n_critical ? 0 : 1) : (t.xcd_lists ? (int)((my_cu >> 8) & 7u) : 0);
  for (;;) {
    __syncthreads();                             // the previous task is done with sV / sAB / the slots
    if (SPARSE && t.evict && tid < 64) {
      // a helper that shares a CU with one of the chains leaves (one flag per chain, polled by one wavefront)
      const bool hit = tid < t.n_chains && tail_ldflag(&t.ctrl[kCtrlChainCu + tid]) == my_cu;
      const unsigned long long any = __ballot(hit);
      if (tid == 0) *slot = any != 0ull ? 1 : 0;
    }
    if (SPARSE) __syncthreads();
    if (tid == 0) {
      int tk = -1, lst = 0;
      const bool evicted = t.evict && (SPARSE ? *slot != 0 : tail_ldflag(&t.ctrl[3]) == my_cu);
      if (!evicted && tail_ldflag(&t.ctrl[1]) == 0) {
        for (int d = 0; d < nl; ++d) {             // own list first, then the others
          const int x = (my_list + d) % nl;
          if (tail_ldflag(&t.ctrl[8 + x]) >= (unsigned)t.ntasks_x[x]) continue;
          const int k = (int)atomicAdd(&t.ctrl[8 + x], 1u);
          if (k < t.ntasks_x[x]) { tk = k; lst = x; break; }
        }
      }
      *slot2 = tk; *slot3 = lst;
    }
    __syncthreads();
    const int tk = *slot2;
    if (tk < 0) return;
    int kind, r, c, iv0 = 0, n_iv = 1;
    if (SPARSE) {
      const GfTask tsk = t.tasks[(*slot3 ? t.ntasks_x[0] : 0) + tk];
      kind = tsk.kind_n & 255; n_iv = tsk.kind_n >> 8; r = tsk.r; c = tsk.c; iv0 = tsk.iv0;
      kind = __builtin_amdgcn_readfirstlane(kind); n_iv = __builtin_amdgcn_readfirstlane(n_iv);
      r = __builtin_amdgcn_readfirstlane(r); c = __builtin_amdgcn_readfirstlane(c); iv0 = __builtin_amdgcn_readfirstlane(iv0);
    } else {
      tail_task(t, tk, *slot3, nl, &kind, &r, &c);
    }
    if (kind == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(3);
#ifdef CBA_DEV_SWITCHES
    if (!SPARSE && kind == 3) {
      if (!tail_helper_pair(t, sV, sAB, r, c, slot, nullptr, 1, nullptr)) return;
      continue;
    }
#endif
    if (SPARSE && kind == 4) {                   // REG2: the two border column blocks of one 128-column tile
      const unsigned long long* arow2 = t.act ? t.act + (size_t)((c - t.x_c0) >> 1) * t.act_words : nullptr;
      if (arow2 && !((arow2[r >> 6] >> (r & 63)) & 1ull)) {      // nothing touches this tile and no fill reaches it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          tail_stflag(&t.tile_flag[(size_t)(r - t.rt0) * t.ntc + c], t.epoch);
          tail_stflag(&t.tile_flag[(size_t)(r - t.rt0) * t.ntc + c + 1], t.epoch);
        }
        continue;
      }
      if (!tail_helper_pair(t, sV, sAB, r, c, slot, t.ivals + iv0, n_iv, arow2)) return;
      continue;
    }
    if (SPARSE && kind == 3) kind = 1;           // PARTFULL: a PART task whose intervals reach up to the row above the tile
    // border tile of the row strip: its 128-column tile's activity bits (uniform)
    const unsigned long long* arow = nullptr;
    if (SPARSE && t.act && kind == 2 && c >= t.x_c0) {
      arow = t.act + (size_t)((c - t.x_c0) >> 1) * t.act_words;
      if (!((arow[r >> 6] >> (r & 63)) & 1ull)) {          // nothing touches this tile and no fill reaches it: not computed, not read
        tail_publish(&t.tile_flag[(size_t)(r - t.rt0) * t.ntc + c], t.epoch);
        continue;
      }
    }
    const int ca = (kind == 1) ? c : r;          // column block of the A operand: PART is L_{k,r+1}^T d L_{k,r+1}
    v4f64 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) acc[i][jj] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const unsigned long long h_start = HELP_NOW();
    unsigned long long h_wait = 0, h_mma = 0;
    // the tile itself (written before this launch) is fetched now, underneath the K loop
    const int row0 = (kind == 1 ? c : r) * kInner;
    const __amdgpu_buffer_rsrc_t rt = tail_rsrc(t.S + (size_t)row0 * ld + (size_t)c * kInner);
    const int acc_voff = ((wm0 + lk) * ld + wn0 + li) * 8;
    double a_rc[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) a_rc[i][jj][r4] = tail_ld1(rt, acc_voff, ((16 * i + 4 * r4) * ld + 16 * jj) * 8);
    for (int iv = 0; iv < n_iv; ++iv) {
      int k = t.rt0, kend = r;
      if (SPARSE) {
        const GfIval v = t.ivals[iv0 + iv];
        k = __builtin_amdgcn_readfirstlane(v.k0); kend = __builtin_amdgcn_readfirstlane(v.k1);
      }
      while (k < kend) {
        int run_end = kend;
        if (SPARSE && arow) {                      // only the rows whose tile (k, c) exists
          k = bits_next(arow, k, kend, true);
          if (k >= kend) break;
          run_end = bits_next(arow, k, kend, false);
        }
        const unsigned long long h0 = HELP_NOW();
        const int nrows = tail_wait_rows(t, k, run_end, ca, c, slot);
        if (nrows <= 0) return;
        const unsigned long long h1 = HELP_NOW();
        const double* A = t.S + (size_t)k * kInner * ld + (size_t)ca * kInner;
        const double* B = t.S + (size_t)k * kInner * ld + (size_t)c * kInner;
#ifdef CBA_TAIL_RING
        if (kind == 1) tail_mma_ring<true>(acc, A, A, ld, t.dvec + (size_t)k * kInner, nrows * kInner, sV);
        else tail_mma_ring<false>(acc, A, B, ld, t.dvec + (size_t)k * kInner, nrows * kInner, sV);
#else
        if (kind == 1) tail_mma_dma<true>(acc, A, A, ld, t.dvec + (size_t)k * kInner, nrows * kInner, sV);
        else tail_mma_dma<false>(acc, A, B, ld, t.dvec + (size_t)k * kInner, nrows * kInner, sV);
#endif
        k += nrows;
        h_wait += h1 - h0; h_mma += HELP_NOW() - h1;
      }
    }
    const unsigned long long h_kend = HELP_NOW();
    // U = A_rc - acc
    if (kind != 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int m = wm0 + i * 16 + lk + 4 * r4, n = wn0 + jj * 16 + li;
            if (kind == 1 && n < m) continue;                               // diagonal tile: upper triangle only
            tail_st1(rt, acc_voff, ((16 * i + 4 * r4) * ld + 16 * jj) * 8, a_rc[i][jj][r4] - acc[i][jj][r4]);
          }
      tail_publish(kind == 0 ? &t.upre_flag[r - t.rt0] : &t.part_flag[c - t.rt0], t.epoch);
      TAIL_STAMP(c - t.rt0, kind == 0 ? 10 : 11);
      continue;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int m = wm0 + i * 16 + lk + 4 * r4, n = wn0 + jj * 16 + li;
          sV[m * TS + n] = a_rc[i][jj][r4] - acc[i][jj][r4];
        }
    if (!tail_wait(t, &t.diag_flag[r - t.rt0], nullptr, slot)) return;       // (its barrier also publishes sV to the other waves)
    const unsigned long long h_diag = HELP_NOW();
    double rdr[2][4];
    {
      // invL_r (K-major, [q][p]) -> sAB as a 64 x TS tile; 1 / d_r of this lane's rows
      const __amdgpu_buffer_rsrc_t ri = tail_rsrc(t.invLt + (size_t)r * kInner * kInner);
      const __amdgpu_buffer_rsrc_t rd = tail_rsrc(t.dvec + (size_t)r * kInner);
      const int rw = 16 * wv + (lane >> 5), cw = 2 * (lane & 31);             // full 512-byte rows per half wave
      v2f64_t u[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = tail_ld2(ri, (rw * kInner + cw) * 8, 2 * k * kInner * 8);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) rdr[i][r4] = tail_ld1(rd, (wm0 + lk) * 8, (16 * i + 4 * r4) * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) { sAB[(rw + 2 * k) * TS + cw] = u[k].x; sAB[(rw + 2 * k) * TS + cw + 1] = u[k].y; }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) rdr[i][r4] = 1.0 / rdr[i][r4];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) acc[i][jj] = (v4f64){0.0, 0.0, 0.0, 0.0};
    tile_mma_lds(acc, sAB, sV);                  // X[p][n] = sum_q invLt[q][p] U[q][n]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          tail_st1(rt, acc_voff, ((16 * i + 4 * r4) * ld + 16 * jj) * 8, acc[i][jj][r4] * rdr[i][r4]);
    if (t.X && c >= t.x_c0) {
      // read by the bulk update, i.e. by a later launch: plain stores
      double* Xt = t.X + (size_t)(r - t.rt0) * kInner * t.ldx + (size_t)c * kInner;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            Xt[(size_t)(wm0 + i * 16 + lk + 4 * r4) * t.ldx + wn0 + jj * 16 + li] = acc[i][jj][r4];
    }
    tail_publish(&t.tile_flag[(size_t)(r - t.rt0) * t.ntc + c], t.epoch);
#ifdef CBA_TAILLOG
    if (g_helplog && tid == 0 && tk < (1 << 20)) {
      unsigned long long* e = g_helplog + (size_t)tk * 8;
      e[0] = h_start; e[1] = h_wait; e[2] = h_mma; e[3] = h_diag - h_kend; e[4] = wall_clock64(); e[5] = (unsigned long long)kind; e[6] = (unsigned long long)r; e[7] = (unsigned long long)c;
    }
#endif
  }
}

__global__ void __launch_bounds__(256, 2) k_ldlt_tail(TailArgs t) {
  __shared__ double smem[2 * kInner * TS];       // two 64 x TS tiles = 80 KB: two workgroups per CU
  volatile int* s_role = reinterpret_cast<volatile int*>(smem + kInner);   // padding of row 0 (4 more bytes of LDS would cost the second workgroup per CU)
  if (threadIdx.x == 0) *s_role = (int)atomicAdd(&t.ctrl[2], 1u);
  __syncthreads();
  const int role = *s_role;
  __syncthreads();
  if (role == 0) {
    __builtin_amdgcn_s_setprio(3);
    tail_chain(t, smem, smem + kInner * TS, t.rt0, t.nr, 0, &t.ctrl[3]);
  } else {
    tail_helper<false>(t, smem, smem + kInner * TS, role);
  }
}

// Block-sparse variant (grid-first elimination, gridfirst_plan.h): the block rows [0, nr) of F -- the grid unknowns of all cameras in
// strip / separator order -- with every column to the right, as ONE launch.  Roles 0 ... n_chains - 1 are pivot chains (one per
// strip, one per camera's separators: they run side by side), the next n_critical roles serve the chains' own tiles first, everybody
// else the border tiles of the row strips.  Same tile arithmetic, flags and bounded waits as k_ldlt_tail.
__global__ void __launch_bounds__(256, 2) k_ldlt_sparse(TailArgs t) {
  __shared__ double smem[2 * kInner * TS];
  volatile int* s_role = reinterpret_cast<volatile int*>(smem + kInner);
  if (threadIdx.x == 0) *s_role = (int)atomicAdd(&t.ctrl[2], 1u);
  __syncthreads();
  const int role = *s_role;
  __syncthreads();
  if (role < t.n_chains) {
    __builtin_amdgcn_s_setprio(3);
    const GfChain ch = t.chains[role];
    tail_chain(t, smem, smem + kInner * TS, __builtin_amdgcn_readfirstlane(ch.r0), __builtin_amdgcn_readfirstlane(ch.r1),
               __builtin_amdgcn_readfirstlane(ch.dep), &t.ctrl[kCtrlChainCu + role]);
  } else {
    tail_helper<true>(t, smem, smem + kInner * TS, role);
  }
}

// The engine's four HIP streams per device are created ONCE, as early as possible in the life of the process, and
// never destroyed.  Measured on MI355X / ROCm 7.2 (round 2): the same GEMM launch runs at 60 TFLOP/s on a stream that was created
// before the process launched its first kernel and at 52-53 TFLOP/s on a stream created afterwards (and that late stream also
// slows the older ones down).  cba_prepare_device() is the hook for hosts to call first thing; cba_create calls it as a fallback.
//   main  : everything on the critical path of a step, the whole factorisation included
//   chain / mid / far : side work (the stragglers of the Jacobian pass, memsets, the distributed solve's exchanges)
// No CU masks: a masked stream costs every launch on it 13-17 % (in-order dispatch, even over the shader engines: DESIGN.md).
struct DeviceStreams {
  hipStream_t main = nullptr, chain = nullptr, mid = nullptr, far = nullptr;
};
static std::mutex g_streams_mutex;
static std::map<int, DeviceStreams> g_streams;
static int device_streams(DeviceStreams* out) {
  int dev = 0;
  CBA_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_streams_mutex);
  auto it = g_streams.find(dev);
  if (it == g_streams.end()) {
    DeviceStreams d;
    int lo = 0, hi = 0;
    CBA_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CBA_HIP(hipStreamCreateWithFlags(&d.main, hipStreamNonBlocking));
    CBA_HIP(hipStreamCreateWithPriority(&d.chain, hipStreamNonBlocking, hi));
    CBA_HIP(hipStreamCreateWithFlags(&d.mid, hipStreamNonBlocking));
    CBA_HIP(hipStreamCreateWithFlags(&d.far, hipStreamNonBlocking));
    it = g_streams.emplace(dev, d).first;
  }
  *out = it->second;
  return CBA_OK;
}
int prepare_device_streams() { DeviceStreams d; return device_streams(&d); }

int make_main_stream(hipStream_t* s) {
  DeviceStreams d;
  int rc = device_streams(&d);
  if (rc != CBA_OK) return rc;
  *s = d.main;
  return CBA_OK;
}

static int super_width();
int ldlt_workspace_alloc(LdltWorkspace& w, int n_pad, int flag_rows_blocks) {
  ldlt_workspace_free(w);
  // X = D L of a super-panel's row strip: the K-major B operand of the bulk update.  A super-panel is at most super_width() + 512
  // rows wide (super_width_at), never wider than the matrix
  {
    // (the distributed schedule falls back to W = 2048 when the developer switch CBA_SUPER_W is not a multiple of its 512-column
    // groups: the panel buffer must hold that width as well)
    int x_rows = std::max(super_width(), (super_width() % 512) ? 2048 : 0) + 512;
    if (x_rows > kSuperMax) x_rows = kSuperMax;
    if (x_rows > n_pad) x_rows = n_pad;
    CBA_HIP(hipMalloc(&w.X, sizeof(double) * (size_t)x_rows * n_pad));
    w.x_rows = x_rows;
  }
  CBA_HIP(hipMalloc(&w.invLt, sizeof(double) * (size_t)(n_pad / kInner) * kInner * kInner));
  CBA_HIP(hipMalloc(&w.dvec, sizeof(double) * (size_t)n_pad));
  CBA_HIP(hipMalloc(&w.status, sizeof(int)));
  {
    DeviceStreams d;
    int rc = device_streams(&d);
    if (rc != CBA_OK) return rc;
    w.panel_stream = d.chain; w.mid_stream = d.mid; w.far_stream = d.far;     // shared, not owned
  }
  CBA_HIP(hipEventCreateWithFlags(&w.ev_strip, hipEventDisableTiming | hipEventDisableSystemFence));
  CBA_HIP(hipEventCreateWithFlags(&w.ev_mid, hipEventDisableTiming | hipEventDisableSystemFence));
  {
    const int ntc = n_pad / kInner;
    int rows = ntc < kTailMaxBlockRows ? ntc : kTailMaxBlockRows;
    if (flag_rows_blocks > rows) rows = flag_rows_blocks < ntc ? flag_rows_blocks : ntc;      // block-sparse launch: every grid block row has its flags
    const size_t words = (size_t)rows * ntc + 3 * (size_t)ntc;
    CBA_HIP(hipMalloc(&w.tail_flags, sizeof(unsigned) * words));
    CBA_HIP(hipMemset(w.tail_flags, 0, sizeof(unsigned) * words));
    CBA_HIP(hipMalloc(&w.tail_ctrl, sizeof(unsigned) * kCtrlWords));
    CBA_HIP(hipMemset(w.tail_ctrl, 0, sizeof(unsigned) * kCtrlWords));
    w.tail_rows_cap = rows * kInner;
    w.tail_epoch = 0;
    CBA_HIP(hipEventCreate(&w.tail_e0));
    CBA_HIP(hipEventCreate(&w.tail_e1));
    CBA_HIP(hipMalloc(&w.back_xe, sizeof(double) * 2 * (size_t)n_pad));
    CBA_HIP(hipMemset(w.back_xe, 0, sizeof(double) * 2 * (size_t)n_pad));
    w.back_epoch = 0;
  }
  w.n_alloc = n_pad;
  return CBA_OK;
}
void ldlt_workspace_free(LdltWorkspace& w) {
  if (w.X) hipFree(w.X);
  if (w.invLt) hipFree(w.invLt);
  if (w.dvec) hipFree(w.dvec);
  if (w.status) hipFree(w.status);
  if (w.ev_strip) hipEventDestroy(w.ev_strip);
  if (w.ev_mid) hipEventDestroy(w.ev_mid);
  for (auto& sp : w.spans) { hipEventDestroy(sp.e0); hipEventDestroy(sp.e1); }
  if (w.tail_flags) hipFree(w.tail_flags);
  if (w.tail_ctrl) hipFree(w.tail_ctrl);
  if (w.back_xe) hipFree(w.back_xe);
  if (w.tail_e0) hipEventDestroy(w.tail_e0);
  if (w.tail_e1) hipEventDestroy(w.tail_e1);
  w = LdltWorkspace();
}

static int span_begin(LdltWorkspace& w, hipStream_t s) {
  if (w.spans_used == (int)w.spans.size()) {
    LdltWorkspace::Span sp;
    CBA_HIP(hipEventCreate(&sp.e0)); CBA_HIP(hipEventCreate(&sp.e1));
    w.spans.push_back(sp);
  }
  CBA_HIP(hipEventRecord(w.spans[w.spans_used].e0, s));
  return CBA_OK;
}
static int span_end(LdltWorkspace& w, hipStream_t s, double flops) {
  CBA_HIP(hipEventRecord(w.spans[w.spans_used].e1, s));
  w.spans[w.spans_used].flops = flops;
  w.spans_used += 1;
  return CBA_OK;
}
int ldlt_collect_spans(LdltWorkspace& w, GemmStats* st) {
  for (int i = 0; i < w.spans_used; ++i) {
    CBA_HIP(hipEventSynchronize(w.spans[i].e1));
    float ms = 0;
    CBA_HIP(hipEventElapsedTime(&ms, w.spans[i].e0, w.spans[i].e1));
    if (st) { st->seconds += ms * 1e-3; st->flops += w.spans[i].flops; st->launches += 1; }
  }
  w.spans_used = 0;
  return CBA_OK;
}
// launch of the 128 x 128 GEMM bracketed by a timing span (only when the caller collects statistics)
static int timed_gemm128(const GemmArgs& g, hipStream_t s, LdltWorkspace& w, bool timed, double tiles) {
  int rc;
  if (timed && (rc = span_begin(w, s))) return rc;
  if ((rc = launch_gemm<128, 128, 64, 64, true>(g, s))) return rc;
  if (timed && (rc = span_end(w, s, tiles * 2.0 * 128 * 128 * g.K))) return rc;
  return CBA_OK;
}


// ---- dataflow launches: host side ----
// Width of the super-panels (rows factored by one dataflow launch in front of a bulk update); the bench harness overrides it
static int super_width() {
  static const char* e = CBA_GETENV("CBA_SUPER_W");        // developer switch (bench harness only)
  int v = e ? atoi(e) : 2048;
  if (v < 256) v = 256;
  if (v > kSuperMax) v = kSuperMax;
  return v / 128 * 128;
}
// Width of the super-panel that starts at row k0: near `sw`, chosen so that the bulk update behind it fills whole rounds of the
// chip.  The update has m (m + 1) / 2 equal tiles (m = trailing rows / 128) and 2 x CUs of them run at a time, all in step: at
// W = 2048 the three updates of cfg 2 have 6.81 / 4.45 / 2.59 rounds, i.e. 3 / 11 / 14 % of their last round is idle.
static int super_width_at(int n_pad, int k0, int sw) {
  static const bool fixed = CBA_GETENV("CBA_SUPER_FIXED") != nullptr;      // developer switch (bench harness only)
  if (fixed || sw < 1024) return sw;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const double slots = 2.0 * cus;
  int best = sw;
  double best_score = -1.0;
  for (int w = sw - 512; w <= sw + 512; w += 128) {
    if (w < 1024 || w > kSuperMax) continue;
    const long long m = (n_pad - (k0 + w)) / 128;
    if (m < 8) continue;
    const double tiles = (double)m * (m + 1) / 2, rounds = std::ceil(tiles / slots);
    // fill of the last round, minus a small penalty for leaving the nominal width (the strip's cost grows with w^2)
    const double score = tiles / (rounds * slots) - 0.01 * std::abs(w - sw) / 128.0;
    if (score > best_score) { best_score = score; best = w; }
  }
  return best;
}
// Rows left to the final dataflow launch (LdltWorkspace::tail_rows, cba_solver_options::factor_tail_rows), clamped to what the
// workspace has flags for: a final launch takes up to tail_rows + sw / 2 rows
// Default: 8192 on one GPU and for the replicated solve (measured optimum at the cfg-2 and cfg-3 sizes with the LDS-DMA helper
// loop).  In the distributed solve the final launch is work EVERY rank repeats while the bulk updates in front of it are split, so
// the optimum moves towards more super-panels: from the single-GPU component times 6144 for 2-3 ranks,
// 4096 from 4 ranks on.
int ldlt_tail_rows(const LdltWorkspace& w, int world) {
  static const char* e = CBA_GETENV("CBA_TAIL_ROWS");      // developer switch (bench harness only)
  int v = e ? atoi(e) : w.tail_rows;
  if (v <= 0) v = world >= 4 ? 4096 : world >= 2 ? 6144 : 8192;
  const int cap = w.tail_rows_cap - super_width() / 2;
  if (v > cap) v = cap;
  return v < 256 ? 256 : v;
}
// Clears the control words of the NEXT dataflow launch now (on stream s, which must be ordered in front of that launch): the
// first launch of a factorisation then starts without a memset between it and the Schur product.
int ldlt_clear_ctrl(LdltWorkspace& w, hipStream_t s) {
  CBA_HIP(hipMemsetAsync(w.tail_ctrl, 0, sizeof(unsigned) * kCtrlWords, s));
  w.tail_ctrl_clean = true;
  return CBA_OK;
}
double ldlt_tail_last_ms(LdltWorkspace& w) {
  if (!w.tail_timed) return 0.0;
  float ms = 0;
  if (hipEventSynchronize(w.tail_e1) != hipSuccess || hipEventElapsedTime(&ms, w.tail_e0, w.tail_e1) != hipSuccess) return 0.0;
  return ms;
}
// Factors rows [t0, n_fact) of S, whose trailing block [t0, n_pad)^2 carries every update of the rows above, with one launch
// on stream s.  t0 and n_fact are multiples of 64.
static int ldlt_tail(double* S, int n_fact, int ld, int t0, LdltWorkspace& w, hipStream_t s, GemmStats* st, double* X = nullptr,
                     int reserve_wgs = 0) {
  if (X && X == w.X && n_fact - t0 > w.x_rows) { set_error("ldlt_tail: super-panel wider than the panel buffer"); return CBA_ERR_STATE; }
  TailArgs t{};
  t.S = S; t.ld = ld;
  t.X = X; t.ldx = ld; t.x_c0 = n_fact / kInner;
  t.rt0 = t0 / kInner; t.nr = n_fact / kInner; t.ntc = ld / kInner;
  t.dvec = w.dvec; t.invLt = w.invLt; t.status = w.status;
  const int rows_cap = w.tail_rows_cap / kInner;
  t.tile_flag = w.tail_flags;
  t.diag_flag = w.tail_flags + (size_t)rows_cap * t.ntc;
  t.upre_flag = t.diag_flag + t.ntc;
  t.part_flag = t.upre_flag + t.ntc;
  t.ctrl = w.tail_ctrl;
  t.epoch = ++w.tail_epoch;
  long long ntasks = 0;
  t.pair = 0;
#ifdef CBA_DEV_SWITCHES
  {
    const char* pair_env = getenv("CBA_TAIL_PAIR");                        // bench harness only; read per call
    t.pair = (pair_env && atoi(pair_env)) && getenv("CBA_TAIL_XCD_LISTS") == nullptr ? 1 : 0;
  }
  if (t.pair) for (int r = t.rt0; r < t.nr; ++r) ntasks += tail_pair_row_count(t.rt0, t.nr, t.ntc, r);
#endif
  if (!t.pair) for (int r = t.rt0; r < t.nr; ++r) ntasks += (r + 1 < t.nr) ? t.ntc - r : t.ntc - t.nr;
  t.ntasks = (int)ntasks;
  static const bool no_evict = CBA_GETENV("CBA_TAIL_NO_EVICT") != nullptr;     // developer switches (bench harness only)
  static const bool one_list = CBA_GETENV("CBA_TAIL_XCD_LISTS") == nullptr;   // per-XCD lists measured: no gain (the helpers are not operand-bandwidth bound)
  t.evict = no_evict ? 0 : 1;
  t.xcd_lists = one_list ? 0 : 1;
  {
    const int nl = t.xcd_lists ? 8 : 1;
    for (int x = 0; x < 8; ++x) t.ntasks_x[x] = 0;
    for (int r = t.rt0; r < t.nr; ++r)
      for (int c = r + 1; c < t.ntc; ++c) t.ntasks_x[c % nl] += (c == r + 1 && r + 1 < t.nr) ? 2 : 1;
    if (t.pair) t.ntasks_x[0] = t.ntasks;
  }
  if (w.tail_ctrl_clean) w.tail_ctrl_clean = false;            // cleared ahead of time by the caller (ldlt_clear_ctrl)
  else CBA_HIP(hipMemsetAsync(w.tail_ctrl, 0, sizeof(unsigned) * kCtrlWords, s));
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  long long grid = ntasks + 1;
  if (grid > 2LL * cus - reserve_wgs) grid = 2LL * cus - reserve_wgs;        // two workgroups per CU are resident (80 KB of LDS each)
  if (grid < 2) grid = 2;
  // a helper that finds itself on the chain's CU leaves (the chain needs the CU's LDS bandwidth and matrix pipes); with a grid this
  // small the only helpers could all sit there and nobody would run the chain's PRE / PART tasks
  if (grid <= 3) t.evict = 0;
  // (the span of the launch itself is a harness statistic: two event records per launch are two bubbles on the critical stream)
#ifdef CBA_DEV_SWITCHES
  if (st) CBA_HIP(hipEventRecord(w.tail_e0, s));
#endif
  hipLaunchKernelGGL(k_ldlt_tail, dim3((unsigned)grid), dim3(256), 0, s, t);
  CBA_HIP(hipGetLastError());
  if (st) {
#ifdef CBA_DEV_SWITCHES
    CBA_HIP(hipEventRecord(w.tail_e1, s));
    w.tail_timed = true;
#endif
    const double R = (double)(n_fact - t0), C = (double)(ld - n_fact);
    st->flops += R * R * R / 3.0 + R * R * C;
  }
  return CBA_OK;
}

// Factor rows [0, n_fact) of the n_pad x n_pad matrix S (ld = n_pad).  Columns up to n_pad take part, so a right-hand side stored
// in a trailing column is forward-substituted and scaled on the fly (it ends up holding D^-1 L^-1 b).
//
// Two-level right-looking schedule on ONE stream: super-panels of ~2048 rows are factored -- diagonal part AND the whole row strip
// right of it -- by the dataflow launch (ldlt_tail with X output), each followed by ONE trailing update with K = the super-panel's
// width on the 128 x 128 MFMA GEMM, alone on the chip; the last tail_rows rows by one more dataflow launch.  No side streams, no
// look-ahead: the chain of a super-panel hides behind its own row-strip tiles, and the bulk update runs at its stand-alone rate.
// (Round 4 built two alternatives and dropped both, DESIGN.md section 3: the next super-panel's dataflow launch NEXT TO the bulk
// update -- its hand-offs through L2 take 5x as long under the GEMM's memory traffic -- and the far columns of a strip as one
// product with the explicit inverse of the super-panel's unit factor.)
int ldlt_factor(double* S, int n_fact, int ld, LdltWorkspace& w, hipStream_t s, GemmStats* st, int k_begin) {
  const int n_pad = ld;
  // (one stream: nothing here runs on the side streams -- their next users, the Jacobian pass and the distributed variant, order
  // themselves against the main stream with their own events; round 4 recorded an event and three stream waits here, a bubble in
  // front of the first dataflow launch)
  const int sw = super_width(), tail_rows = ldlt_tail_rows(w);
  int k0 = k_begin, rc;          // rows above k_begin are factored already and their update is applied (grid-first elimination)
  while (n_fact - k0 > tail_rows + sw / 2 && n_pad - (k0 + sw) >= 1024) {
    const int wk = super_width_at(n_pad, k0, sw);
    if ((rc = ldlt_tail(S, k0 + wk, ld, k0, w, s, st, w.X))) return rc;
    GemmArgs u{};
    u.A = S + (size_t)k0 * ld; u.lda = ld; u.B = w.X; u.ldb = n_pad; u.K = wk;
    u.C = S; u.ldc = ld; u.Cin = S; u.ldcin = ld; u.diag = 0; u.upper = 1;
    const int tl = (n_pad - (k0 + wk)) / 128;
    u.m_off = k0 + wk; u.m_tiles = tl; u.n_off = k0 + wk; u.n_tiles = tl;
    if ((rc = timed_gemm128(u, s, w, st != nullptr, (double)tl * (tl + 1) / 2))) return rc;
    if (st) { const double rows = (double)(n_pad - (k0 + wk)); st->flops += rows * rows * wk; st->launches += 1; }
    k0 += wk;
  }
  if ((rc = ldlt_tail(S, n_fact, ld, k0, w, s, st))) return rc;
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ---- grid-first elimination (gridfirst_plan.h) ----
// Block rows [0, nbg) of F -- the grid unknowns -- with every column to the right in ONE block-sparse dataflow launch; X = D L of
// the border columns goes to Xb (rows of the grid part x border columns, leading dimension ldxb, column 0 = column Gf of F).
static int ldlt_sparse(double* F, int ld, const GfDevice& g, LdltWorkspace& w, hipStream_t s, GemmStats* st, double* Xb, int ldxb) {
  TailArgs t{};
  t.S = F; t.ld = ld;
  t.X = Xb - (size_t)g.nbg * kInner; t.ldx = ldxb; t.x_c0 = g.nbg;
  t.rt0 = 0; t.nr = g.nbg; t.ntc = ld / kInner;
  t.dvec = w.dvec; t.invLt = w.invLt; t.status = w.status;
  const int rows_cap = w.tail_rows_cap / kInner;
  if (g.nbg > rows_cap || g.n_chains > kMaxChains) { set_error("ldlt_sparse: workspace too small for the plan"); return CBA_ERR_STATE; }
  t.tile_flag = w.tail_flags;
  t.diag_flag = w.tail_flags + (size_t)rows_cap * t.ntc;
  t.upre_flag = t.diag_flag + t.ntc;
  t.part_flag = t.upre_flag + t.ntc;
  t.ctrl = w.tail_ctrl;
  t.epoch = ++w.tail_epoch;
  t.tasks = g.tasks; t.ivals = g.ivals; t.chains = g.chains; t.n_chains = g.n_chains;
  t.act = g.act; t.act_words = g.act_words;
  for (int x = 0; x < 8; ++x) t.ntasks_x[x] = 0;
  t.ntasks_x[0] = g.n_tasks0; t.ntasks_x[1] = g.n_tasks1;
  t.ntasks = g.n_tasks0 + g.n_tasks1;
  t.evict = 1; t.xcd_lists = 0; t.pair = 0;
  if (w.tail_ctrl_clean) w.tail_ctrl_clean = false;
  else CBA_HIP(hipMemsetAsync(w.tail_ctrl, 0, sizeof(unsigned) * kCtrlWords, s));
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  long long grid = (long long)t.ntasks + g.n_chains;
  if (grid > 2LL * cus) grid = 2LL * cus;
  // Workgroups that serve list 0 first: per chain the tasks of about two block rows (PRE, PART and the band's tiles).  The chains
  // and these roles are the first workgroups dispatched; everything else starts with the border tiles.
  long long crit = (long long)g.n_chains * 16;
  if (crit > grid / 4) crit = grid / 4;
  if (crit < 1) crit = 1;
  if (grid < g.n_chains + crit + 1) grid = g.n_chains + crit + 1;
  t.n_critical = (int)crit;
  if (grid - g.n_chains <= 3) t.evict = 0;
#ifdef CBA_DEV_SWITCHES
  if (st) CBA_HIP(hipEventRecord(w.tail_e0, s));
#endif
  hipLaunchKernelGGL(k_ldlt_sparse, dim3((unsigned)grid), dim3(256), 0, s, t);
  CBA_HIP(hipGetLastError());
  if (st) {
#ifdef CBA_DEV_SWITCHES
    CBA_HIP(hipEventRecord(w.tail_e1, s));
    w.tail_timed = true;
#endif
    st->flops += g.flops_grid;
  }
  return CBA_OK;
}

// Factors rows [0, n_fact) of F = [grid | border] (ld = n_pad of the plan): block-sparse launch of the grid rows, ONE update of
// the border by the K = Gf product C -= L^T X on the 128 x 128 MFMA GEMM (optionally block-sparse in K: `kmask`, one bit per
// 128-column border tile and 16-row slab; `tile_list`: the (tm, tn) tiles of the update in the order they should be handed out,
// heaviest first -- a scheduling hint, any permutation of the upper tiles is correct), then the dense border by the two-level
// schedule of ldlt_factor.
int ldlt_factor_gridfirst(double* F, int n_fact, int ld, const GfDevice& g, double* Xb, int ldxb, LdltWorkspace& w, hipStream_t s,
                          GemmStats* st, const unsigned long long* kmask, int kmask_words, const int* tile_list, int tile_list_entries) {
  int rc;
  if ((rc = ldlt_sparse(F, ld, g, w, s, st, Xb, ldxb))) return rc;
  const int Gf = g.nbg * kInner;
  GemmArgs u{};
  u.A = F; u.lda = ld; u.B = Xb - Gf; u.ldb = ldxb; u.K = Gf;
  u.C = F; u.ldc = ld; u.Cin = F; u.ldcin = ld; u.diag = 0; u.upper = 1;
  const int tl = (ld - Gf) / 128;
  u.m_off = Gf; u.m_tiles = tl; u.n_off = Gf; u.n_tiles = tl;
  u.kmask = kmask; u.kmask_words = kmask_words; u.slab16 = 1; u.tile_list = reinterpret_cast<const int4*>(tile_list); u.tile_list_entries = tile_list_entries;
  if ((rc = timed_gemm128(u, s, w, st != nullptr, (double)tl * (tl + 1) / 2))) return rc;
  if (st && kmask && w.spans_used > 0) w.spans[w.spans_used - 1].masked_update = true;      // (the caller replaces the dense flop count by the executed one)
  if (st) { const double rows = (double)(ld - Gf); st->flops += rows * rows * Gf; st->launches += 1; }
  return ldlt_factor(F, n_fact, ld, w, s, st, Gf);
}

// ------------------------------------------------------------------------------------------------
// Distributed factorisation (cba_config.distributed_solve) -- the two-level schedule of ldlt_factor with
// the throughput-bound part (the K = W super-panel updates) split over the ranks and the latency-bound part (the dataflow
// launches) replicated:
//   ownership : 512-column groups of S, block-cyclic over the ranks (group g -> rank g % world)
//   on entry  : S holds THIS RANK'S PARTIAL reduced system (nothing has been summed over the ranks yet)
//   (1) rows [0, W) -- one contiguous block of S -- are summed in place (all-reduce: every rank factors them); the first
//       dataflow launch starts; underneath it the rows below are REDUCE-SCATTERED straight into their owners (rows [W, end of
//       the group) of each 512-column group travel to the group's owner only: the upper triangle once, no zeros);
//   (2) per super-panel [k0, k0 + W): every rank runs the dataflow launch on the complete row band (identical arithmetic on
//       identical data -> identical L, d, X on every rank), then updates ONLY ITS OWN column groups, the rows of the next
//       band first; as soon as those are done the next band is ALL-GATHERED from its owners (pack -> collective -> unpack on
//       a second stream) while the update of the rows below is still running on the main stream;
//   (3) the band in front of the final dataflow launch covers all remaining rows, so that the last launch is replicated too.
// After the last launch every rank holds the complete factor, d and the forward-substituted right-hand side: the back
// substitution runs replicated as in the single-GPU path.  Link volume per solve and rank: (world - 1) / world x the upper
// triangle for the reduce-scatter + the same for all the gathers together = what ONE all-reduce of the packed system moves.
// The collectives are blocking host calls (cba_collective_fn); they overlap with device work that was queued before them.
// ------------------------------------------------------------------------------------------------
constexpr int kOwnGroup = 512;
int launch_pack_upper(const double* S, int n_pad, double* P, int unpack, hipStream_t s);
struct RectArgs {
  double* S; int ld; int n_pad;
  int g_begin;          // first column group of the transfer
  int world;
  int R0;               // first row
  int nrows;            // > 0: every group sends rows [R0, R0 + nrows) (a band); 0: rows [R0, end of the group) (the triangle)
};
// i-th column group of rank q in this transfer: first column, width, number of rows, offset in q's block of the buffer
__host__ __device__ inline bool dist_rect(const RectArgs& a, int q, int i, int* col0, int* width, int* height, long long* off) {
  const int gq0 = a.g_begin + ((q - a.g_begin % a.world) % a.world + a.world) % a.world;
  const int g = gq0 + i * a.world;
  *col0 = g * kOwnGroup;
  if (*col0 >= a.n_pad) return false;
  *width = a.n_pad - *col0 < kOwnGroup ? a.n_pad - *col0 : kOwnGroup;
  if (a.nrows > 0) {
    *height = a.nrows;
    *off = (long long)i * a.nrows * kOwnGroup;
  } else {
    const long long h0 = (long long)(gq0 + 1) * kOwnGroup - a.R0;       // only the last group of the matrix can be narrower / shorter
    const int end = *col0 + kOwnGroup < a.n_pad ? *col0 + kOwnGroup : a.n_pad;
    *height = end - a.R0;
    *off = (long long)kOwnGroup * ((long long)i * h0 + (long long)a.world * kOwnGroup * ((long long)i * (i - 1) / 2));
  }
  return true;
}
static long long dist_count(const RectArgs& a, int q) {
  long long total = 0;
  for (int i = 0;; ++i) {
    int c0, wd, h; long long off;
    if (!dist_rect(a, q, i, &c0, &wd, &h, &off)) break;
    total = off + (long long)h * wd;
  }
  return total;
}
// buf <-> S for the groups of ranks q_first .. q_first + gridDim.z - 1 (blockIdx.y = group index, grid-stride over its entries)
__global__ void __launch_bounds__(256) k_dist_copy(RectArgs a, double* __restrict__ buf, long long rank_stride, int q_first, int unpack) {
  const int q = q_first + blockIdx.z;
  int col0, width, height; long long off;
  if (!dist_rect(a, q, blockIdx.y, &col0, &width, &height, &off)) return;
  double* b = buf + (long long)blockIdx.z * rank_stride + off;
  // a row of a group is <= 4 KB contiguous on both sides: rows over the x blocks, two doubles per lane (width is a multiple of 128)
  for (int r = blockIdx.x; r < height; r += gridDim.x) {
    double2* sp = reinterpret_cast<double2*>(a.S + (size_t)(a.R0 + r) * a.ld + col0);
    double2* bp = reinterpret_cast<double2*>(b + (long long)r * width);
    for (int c = threadIdx.x; c < width / 2; c += 256) {
      if (unpack) sp[c] = bp[c]; else bp[c] = sp[c];
    }
  }
}
static int dist_copy(const RectArgs& a, double* buf, long long rank_stride, int q_first, int q_count, int unpack, hipStream_t s) {
  const int groups = (a.n_pad / kOwnGroup - a.g_begin + a.world) / a.world + 1;
  if (groups <= 0 || q_count <= 0) return CBA_OK;
  hipLaunchKernelGGL(k_dist_copy, dim3(128, (unsigned)groups, (unsigned)q_count), dim3(256), 0, s, a, buf, rank_stride, q_first, unpack);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}
// Size of each of the two staging buffers: the largest transfer of the schedule, from the layout functions themselves -- the
// triangle below the first band (reduce-scatter), a band of W rows or everything that is left (all-gathers, for every possible
// number of super-panels), and the packed upper triangle of small systems.  (Round 3 reserved world x ceil(groups / world) x 512
// x n_pad doubles, about twice this: 2 x 14.7 GB more than needed at BASELINE configs[4].)
size_t ldlt_dist_buffer_doubles(int n_pad, int world) {
  if (world < 1) world = 1;
  int W = super_width();
  if (W <= 0 || W % kOwnGroup) W = 2048;
  long long need = (long long)n_pad * (n_pad / 128 + 1) * 64;                       // packed upper triangle
  auto transfer = [&](int g_begin, int R0, int nrows) {
    RectArgs a{nullptr, n_pad, n_pad, g_begin, world, R0, nrows};
    long long m = 0;
    for (int q = 0; q < world; ++q) m = std::max(m, dist_count(a, q));
    need = std::max(need, m * world);
  };
  if (n_pad > W) transfer(W / kOwnGroup, W, 0);
  for (int e0 = W; e0 < n_pad; e0 += W) {
    transfer(e0 / kOwnGroup, e0, std::min(W, n_pad - e0));
    transfer(e0 / kOwnGroup, e0, n_pad - e0);
  }
  return (size_t)need;
}
// The collectives through the caller's callback, or -- when only an all-reduce is available -- emulated with it (same
// results, more bytes: the tests with several ranks on one GPU and hosts that have not been moved to cba_collective_fn yet)
static int dist_collective(const DistComm& c, int op, double* send, double* recv, long long count, hipStream_t s2) {
  if (c.collective) return c.collective(op, send, recv, (int64_t)count, c.collective_user) == 0 ? CBA_OK : CBA_ERR_STATE;
  if (!c.allreduce) return CBA_ERR_STATE;
  if (op == CBA_COLL_ALLREDUCE_SUM) return c.allreduce(recv, (int64_t)count, c.allreduce_user) == 0 ? CBA_OK : CBA_ERR_STATE;
  if (op == CBA_COLL_REDUCE_SCATTER_SUM) {
    if (c.allreduce(send, (int64_t)count * c.world, c.allreduce_user) != 0) return CBA_ERR_STATE;
    CBA_HIP(hipMemcpyAsync(recv, send + (size_t)c.rank * count, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, s2));
    CBA_HIP(hipStreamSynchronize(s2));
    return CBA_OK;
  }
  CBA_HIP(hipMemsetAsync(recv, 0, sizeof(double) * (size_t)count * c.world, s2));
  CBA_HIP(hipMemcpyAsync(recv + (size_t)c.rank * count, send, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, s2));
  CBA_HIP(hipStreamSynchronize(s2));
  return c.allreduce(recv, (int64_t)count * c.world, c.allreduce_user) == 0 ? CBA_OK : CBA_ERR_STATE;
}
// trailing update of the column groups this rank owns, rows [r_begin, r_end): C -= L^T X with K = W (one launch)
static int dist_update(double* S, int ld, int k0, int W, int r_begin, int r_end, const DistComm& c, LdltWorkspace& w, hipStream_t s, GemmStats* st) {
  const int n_pad = ld;
  if (r_end <= r_begin) return CBA_OK;
  constexpr int G = kOwnGroup / 128;
  int g0 = r_begin / kOwnGroup;                           // first group that reaches past row r_begin
  while (g0 % c.world != c.rank) ++g0;
  int owned_tiles = 0;
  double tiles = 0;
  for (int gi = g0; gi * kOwnGroup < n_pad; gi += c.world) {
    const int hi = (gi + 1) * kOwnGroup < n_pad ? (gi + 1) * kOwnGroup : n_pad;
    owned_tiles += (hi - gi * kOwnGroup) / 128;
    for (int n0 = gi * kOwnGroup; n0 < hi; n0 += 128) {       // tiles at or above the diagonal (the others are skipped in the kernel)
      const int last_row = n0 + 127 < r_end - 1 ? n0 + 127 : r_end - 1;
      if (last_row >= r_begin) tiles += (last_row - r_begin) / 128 + 1;
    }
  }
  if (owned_tiles == 0) return CBA_OK;
  GemmArgs u{};
  u.A = S + (size_t)k0 * ld; u.lda = ld; u.B = w.X; u.ldb = n_pad; u.K = W;
  u.C = S; u.ldc = ld; u.Cin = S; u.ldcin = ld; u.diag = 0; u.upper = 0;
  u.m_off = r_begin; u.m_tiles = (r_end - r_begin) / 128; u.n_off = g0 * kOwnGroup; u.n_tiles = owned_tiles;
  u.col_group = G; u.col_stride = c.world;
  int rc = timed_gemm128(u, s, w, st != nullptr, tiles);
  if (rc) return rc;
  if (st) { st->flops += tiles * 2.0 * 128 * 128 * W; st->launches += 1; }
  return CBA_OK;
}
int ldlt_factor_distributed(double* S, int n_fact, int ld, LdltWorkspace& w, hipStream_t s, const DistComm& c, GemmStats* st) {
  const int n_pad = ld;
  int W = super_width();
  if (W <= 0 || W % kOwnGroup) W = 2048;
  if (c.world < 1 || c.rank < 0 || c.rank >= c.world || !c.send || !c.recv) return CBA_ERR_ARG;
  hipStream_t s2 = w.far_stream;
  int nsp = 0;
  for (int k0 = 0; n_fact - k0 > ldlt_tail_rows(w, c.world) + W / 2 && n_pad - (k0 + W) >= 1024; k0 += W) ++nsp;
  int rc;
  if (nsp == 0) {
    // small systems: one dataflow launch on everything -- sum the packed upper triangle, factor replicated
    if ((rc = launch_pack_upper(S, n_pad, c.send, 0, s))) return rc;
    CBA_HIP(hipStreamSynchronize(s));
    const long long packed = (long long)n_pad * (n_pad / 128 + 1) * 64;       // sum_i 128 (n_pad - 128 i)
    if ((rc = dist_collective(c, CBA_COLL_ALLREDUCE_SUM, nullptr, c.send, packed, s2))) return rc;
    if ((rc = launch_pack_upper(S, n_pad, c.send, 1, s))) return rc;
    return ldlt_factor(S, n_fact, ld, w, s, st);
  }
  // (1) first band: contiguous rows of S, summed in place; its dataflow launch starts; the rest goes to its owners underneath.
  //     The blocks of the reduce-scatter are packed first (second stream, next to the all-reduce of the band): queued behind
  //     the dataflow launch the copy kernel would get the few workgroup slots that launch leaves.
  CBA_HIP(hipStreamSynchronize(s));
  RectArgs a0{S, ld, n_pad, W / kOwnGroup, c.world, W, 0};
  long long count0 = 0;
  for (int q = 0; q < c.world; ++q) count0 = std::max(count0, dist_count(a0, q));
  if ((size_t)count0 * c.world > c.buf_doubles) return CBA_ERR_ARG;
  if (count0 > 0 && (rc = dist_copy(a0, c.send, count0, 0, c.world, 0, s2))) return rc;          // every destination's block
  if ((rc = dist_collective(c, CBA_COLL_ALLREDUCE_SUM, nullptr, S, (long long)W * ld, s2))) return rc;
  // the first dataflow launch leaves workgroup slots free for the kernels of the reduce-scatter that runs next to it: with all
  // 2 x CUs slots (and all LDS) taken by helpers, the collective's kernels start only when the helpers run out of tickets,
  // i.e. after the launch (measured with one rank: the 430 MB copy took 1.57 ms next to a full launch, 0.18 ms alone)
  static const int reserve = CBA_GETENV("CBA_DIST_RESERVE_WGS") ? atoi(CBA_GETENV("CBA_DIST_RESERVE_WGS")) : 64;
  if ((rc = ldlt_tail(S, W, ld, 0, w, s, st, w.X, (c.world > 1 || c.collective) ? reserve : 0))) return rc;
  {
    if (count0 > 0) {
      CBA_HIP(hipStreamSynchronize(s2));
      if ((rc = dist_collective(c, CBA_COLL_REDUCE_SCATTER_SUM, c.send, c.recv, count0, s2))) return rc;
      if ((rc = dist_copy(a0, c.recv, 0, c.rank, 1, 1, s2))) return rc;               // own groups back into S
    }
    CBA_HIP(hipEventRecord(w.ev_mid, s2));
    CBA_HIP(hipStreamWaitEvent(s, w.ev_mid, 0));
  }
  // (2) super-panels
  for (int k = 0; k < nsp; ++k) {
    const int k0 = k * W, e0 = k0 + W;
    const bool last = k == nsp - 1;
    const int e1 = last ? n_pad : e0 + W;                  // rows every rank needs next: the next band, or all that is left
    if (k > 0 && (rc = ldlt_tail(S, e0, ld, k0, w, s, st, w.X))) return rc;
    if ((rc = dist_update(S, ld, k0, W, e0, e1, c, w, s, st))) return rc;
    CBA_HIP(hipEventRecord(w.ev_strip, s));
    if (!last && (rc = dist_update(S, ld, k0, W, e1, n_pad, c, w, s, st))) return rc;
    // gather rows [e0, e1) from the owners of their columns, next to the update of the rows below
    CBA_HIP(hipStreamWaitEvent(s2, w.ev_strip, 0));
    RectArgs a{S, ld, n_pad, e0 / kOwnGroup, c.world, e0, e1 - e0};
    long long count = 0;
    for (int q = 0; q < c.world; ++q) count = std::max(count, dist_count(a, q));
    if ((size_t)count * c.world > c.buf_doubles) return CBA_ERR_ARG;
    if (c.world > 1 || c.collective) {
      if ((rc = dist_copy(a, c.send, 0, c.rank, 1, 0, s2))) return rc;
      CBA_HIP(hipStreamSynchronize(s2));
      if ((rc = dist_collective(c, CBA_COLL_ALLGATHER, c.send, c.recv, count, s2))) return rc;
      if ((rc = dist_copy(a, c.recv, count, 0, c.world, 1, s2))) return rc;
    }
    CBA_HIP(hipEventRecord(w.ev_mid, s2));
    CBA_HIP(hipStreamWaitEvent(s, w.ev_mid, 0));
  }
  // (3) the rest, replicated
  if ((rc = ldlt_tail(S, n_fact, ld, nsp * W, w, s, st))) return rc;
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// Backward substitution L^T x = z for the factored rows; z sits in column `zcol` of S.
//   x_j = z_j - sum_{i > j} L(i,j) x_i = z_j - sum_{i > j} S[j][i] x[i]
// Right-looking by panels of 256 rows: the panel's own triangle is solved by one workgroup (four
// 64-blocks, using the stored inverses of the unit-lower diagonal factors), then every earlier row
// subtracts its 256-column slice times the new x values (one wavefront per row, coalesced).
__global__ void k_gather_col(const double* __restrict__ S, int ld, int col, int n, double* __restrict__ x) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) x[j] = S[(size_t)j * ld + col];
}
// Panel triangle of the back substitution.  1024 lanes: row p = tid / 16 of the current 64-block, 16 lanes
// per row.  Every lane first loads ALL matrix entries it will need for the four 64-blocks (its slices of
// the rows of L right of each block and of the stored inverse blocks, 64 doubles) with independent loads
// -- one L2 round trip for the whole panel instead of two per block -- and the four dependent block
// solves then run out of registers and LDS.
__global__ void __launch_bounds__(1024) k_back_panel_diag(const double* __restrict__ S, int ld, int k0, int nb,
                                                          const double* __restrict__ invLt_all, double* __restrict__ x) {
  constexpr int NB = kPanel / kInner;          // 4 blocks
  constexpr int LPER = (kPanel - kInner) / 16; // 12 columns of L per lane and block (at most)
  constexpr int IPER = kInner / 16;            // 4 entries of the inverse per lane and block
  __shared__ double xs[kPanel];
  __shared__ double t[kInner];
  const int p = threadIdx.x >> 4, l = threadIdx.x & 15;
  const int nblk = nb / kInner;
  double Lr[NB][LPER], Ir[NB][IPER];
#pragma unroll
  for (int sub = 0; sub < NB; ++sub) {
    const int j0 = sub * kInner;
    const bool live = sub < nblk;
    const double* row = S + (size_t)(k0 + j0 + p) * ld + k0;
#pragma unroll
    for (int i = 0; i < LPER; ++i) {
      const int col = j0 + kInner + l + 16 * i;
      Lr[sub][i] = (live && col < nb) ? row[col] : 0.0;
    }
    const double* inv = invLt_all + (size_t)((k0 + j0) / kInner) * kInner * kInner + (size_t)p * kInner;
#pragma unroll
    for (int i = 0; i < IPER; ++i) Ir[sub][i] = live ? inv[l + 16 * i] : 0.0;
  }
  if (threadIdx.x < kPanel) xs[threadIdx.x] = (threadIdx.x < nb) ? x[k0 + threadIdx.x] : 0.0;
  __syncthreads();
#pragma unroll
  for (int sub = NB - 1; sub >= 0; --sub) {
    if (sub >= nblk) continue;   // uniform
    const int j0 = sub * kInner;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < LPER; ++i) {
      const int col = j0 + kInner + l + 16 * i;
      if (col < kPanel) acc += Lr[sub][i] * xs[col];
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 8, 64);
    if (l == 0) t[p] = xs[j0 + p] - acc;
    __syncthreads();
    // x[q] = sum_{p' >= q} invL(p',q) t[p'] = sum_{p'} invLt[q][p'] t[p']   (q = p here)
    double a2 = 0.0;
#pragma unroll
    for (int i = 0; i < IPER; ++i) a2 += Ir[sub][i] * t[l + 16 * i];
    a2 += __shfl_xor(a2, 1, 64); a2 += __shfl_xor(a2, 2, 64);
    a2 += __shfl_xor(a2, 4, 64); a2 += __shfl_xor(a2, 8, 64);
    if (l == 0) xs[j0 + p] = a2;
    __syncthreads();
  }
  if (threadIdx.x < nb) x[k0 + threadIdx.x] = xs[threadIdx.x];
}
__global__ void __launch_bounds__(256) k_back_panel_update(const double* __restrict__ S, int ld, int k0, int nb,
                                                           double* __restrict__ x) {
  __shared__ double xs[kPanel];
  for (int i = threadIdx.x; i < nb; i += 256) xs[i] = x[k0 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= k0) return;
  const double* row = S + (size_t)q * ld + k0;
  double acc = 0.0;
  for (int i = lane; i < nb; i += 64) acc += row[i] * xs[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) x[q] -= acc;
}
// ---- the same substitution as ONE dataflow launch (default) -------------------------------------------------------------
// Workgroup b owns block row r = nblk - 1 - b (64 rows): it walks its row strip from the last column block down to r + 1,
// subtracting S[r-rows][c-cols] x_c as the x_c become available, then solves its 64 x 64 unit triangle with the stored
// inverse and publishes x_r.  An entry of x travels as a 16-byte pair {value, tag = number of this call} written by ONE
// store instruction, so the consumer needs a single agent-scope load per entry to get value AND validity (a separate flag
// costs a second L2 round trip per block: the chain is 196 blocks long).  Workgroups are dispatched in index order and only
// wait for lower indices, so the launch cannot deadlock; every spin is bounded like in the tail launch.
// Chain per block: one load round trip + two 64 x 64 matrix-vector products out of registers ~ 1.5 us since round 4 (2.8 in
// round 3; the 98 launches of the panel version: 6 us per 64 rows).
struct BackArgs {
  const double* S; int ld; int n_fact; int zcol;
  const double* invLt;
  double* x;                  // n_fact doubles out
  double* xe;                 // 2 * n_pad doubles: {value, tag} pairs
  double tag;
  int* status;
  const unsigned long long* rowmask;   // optional [block row][mask_words]: bit c = tile (r, c) can be non-zero (grid-first order: the
  int mask_words;                      // grid x grid part of the factor is block-sparse); null = every tile
};
// Round 4: (1) a lane's 16 columns of a 64-column block are 8 jj + 2 q4 + {0, 1}, jj = 0 ... 7 -- one 16-byte load per jj, the four
// lanes of a row read 64 contiguous bytes per instruction; with 16 consecutive columns per lane a wavefront-load touched 64 cache
// lines and the strip loop, not the chain, set the pace: 0.56 -> 0.34 ms at BASELINE configs[1].  (2) x_c is double-buffered in
// LDS (one barrier per column block) and the strip entries of the next column block are in flight while x_c is polled.
// (Several 64-blocks per workgroup, handing x over through LDS instead of L2, were built and measured: 0.46 ms with two, 0.76 ms
// with four blocks -- the barriers of 8 / 16 wavefronts cost more per column block than the saved round trips,
// profiles/r04_back_substitution_blocks.txt.)
__global__ void __launch_bounds__(256) k_back_dataflow(BackArgs a) {
  __shared__ double s_x[2][kInner];
  __shared__ double s_t[kInner];
  __shared__ int s_ok;
  const int nblk = (a.n_fact + kInner - 1) / kInner;
  const int r = nblk - 1 - (int)blockIdx.x;
  const int tid = threadIdx.x, p = tid >> 2, q4 = tid & 3;     // row p of the block, lane q4 of the row's four
  const int j0 = r * kInner;
  const int rows = a.n_fact - j0 < kInner ? a.n_fact - j0 : kInner;
  const bool rlive = p < rows;
  auto ld16 = [&](const double* base, double (&v)[16], bool on, int cw) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int col = 8 * jj + 2 * q4;
      double2 t = make_double2(0.0, 0.0);
      if (on && col < cw) t = *reinterpret_cast<const double2*>(base + col);      // cw is even (n_fact is a multiple of 64)
      v[2 * jj] = t.x; v[2 * jj + 1] = t.y;
    }
  };
  auto dot16 = [&](const double (&v)[16], const double* xs) {
    double sum = 0.0;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) sum += v[2 * jj] * xs[8 * jj + 2 * q4] + v[2 * jj + 1] * xs[8 * jj + 2 * q4 + 1];
    return sum;
  };
  // this lane's slice of the stored inverse: x_r[p] = sum_{p'} invLt[p][p'] t[p'] over the lane's 16 columns p'
  double inv[16];
  ld16(a.invLt + (size_t)r * kInner * kInner + (size_t)p * kInner, inv, true, kInner);
  const double* row = a.S + (size_t)(j0 + (rlive ? p : 0)) * a.ld;
  const double z = rlive ? row[a.zcol] : 0.0;
  double acc = 0.0;
  const __amdgpu_buffer_rsrc_t rx = tail_rsrc(a.xe);
  double l[16], ln[16];
  // largest column block below c whose tile (r, .) can be non-zero (r itself when there is none)
  const unsigned long long* mrow = a.rowmask ? a.rowmask + (size_t)r * a.mask_words : nullptr;
  auto next_active = [&](int c) -> int {
    if (!mrow) return c - 1;
    int cc = c - 1;
    while (cc > r) {
      const unsigned long long wbits = mrow[cc >> 6] & (~0ull >> (63 - (cc & 63)));
      if (wbits) { const int hit = (cc & ~63) + 63 - __builtin_clzll(wbits); return hit > r ? hit : r; }
      cc = (cc & ~63) - 1;
    }
    return r;
  };
  int c = next_active(nblk);
  if (c > r) ld16(row + (size_t)c * kInner, l, rlive, a.n_fact - c * kInner < kInner ? a.n_fact - c * kInner : kInner);
  int par = 0;
  while (c > r) {
    // the strip's entries do not depend on x: those of the next column block are in flight while this one's x is polled
    const int cn = next_active(c);
    ld16(row + (size_t)(cn > r ? cn : c) * kInner, ln, rlive && cn > r, kInner);
    double* xs = s_x[par];
    par ^= 1;
    if (tid < kInner) {
      const unsigned long long t0 = wall_clock64();
      v2f64_t v;
      unsigned spins = 0;
      bool ok = true;
      for (;;) {
        v = tail_ld2(rx, (c * kInner + tid) * 16);
        if (__all(v.y == a.tag)) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0 && __builtin_amdgcn_readfirstlane((int)(wall_clock64() - t0 > kTailTimeoutTicks))) { ok = false; break; }
      }
      xs[tid] = v.x;
      if (tid == 0) { s_ok = ok ? 1 : 0; if (!ok) atomicExch(a.status, 3); }
    }
    __syncthreads();            // (the next write to this buffer is two column blocks away: behind the next barrier)
    if (!s_ok) return;
    acc += dot16(l, xs);
#pragma unroll
    for (int j = 0; j < 16; ++j) l[j] = ln[j];
    c = cn;
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  if (q4 == 0) s_t[p] = rlive ? z - acc : 0.0;
  __syncthreads();
  double xr = dot16(inv, s_t);
  xr += __shfl_xor(xr, 1, 64);
  xr += __shfl_xor(xr, 2, 64);
  if (q4 == 0 && rlive) {
    v2f64_t v; v.x = xr; v.y = a.tag;
    tail_st2(rx, (j0 + p) * 16, 0, v);
    a.x[j0 + p] = xr;
  }
}

int ldlt_back_solve(const double* S, int n_fact, int ld, int zcol, const LdltWorkspace& w, double* x, hipStream_t s,
                    const unsigned long long* rowmask, int mask_words) {
  static const bool no_df = CBA_GETENV("CBA_BACK_PANELS") != nullptr;       // developer switch (bench harness only)
  if (((w.back_dataflow && !no_df) || rowmask) && w.back_xe && n_fact % kInner == 0) {
    BackArgs a{};
    a.S = S; a.ld = ld; a.n_fact = n_fact; a.zcol = zcol; a.invLt = w.invLt; a.x = x; a.xe = w.back_xe; a.status = w.status;
    a.rowmask = rowmask; a.mask_words = mask_words;
    LdltWorkspace& wm = const_cast<LdltWorkspace&>(w);
    wm.back_epoch += 1;
    a.tag = (double)wm.back_epoch;
    const int nblk = (n_fact + kInner - 1) / kInner;
    hipLaunchKernelGGL(k_back_dataflow, dim3(nblk), dim3(256), 0, s, a);
    CBA_HIP(hipGetLastError());
    return CBA_OK;
  }

  hipLaunchKernelGGL(k_gather_col, dim3((n_fact + 255) / 256), dim3(256), 0, s, S, ld, zcol, n_fact, x);
  int last = ((n_fact - 1) / kPanel) * kPanel;
  for (int k0 = last; k0 >= 0; k0 -= kPanel) {
    int nb = (n_fact - k0 < kPanel) ? (n_fact - k0) : kPanel;
    hipLaunchKernelGGL(k_back_panel_diag, dim3(1), dim3(1024), 0, s, S, ld, k0, nb, w.invLt, x);
    if (k0 > 0) hipLaunchKernelGGL(k_back_panel_update, dim3((k0 + 3) / 4), dim3(256), 0, s, S, ld, k0, nb, x);
  }
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// Packed upper 128-row blocks of an n_pad x n_pad matrix: block i keeps rows [128 i, 128 i + 128) and
// columns [128 i, n_pad), rows contiguous.  This is what crosses ranks in the multi-GPU path.
int64_t packed_upper_doubles(int n_pad) {
  int64_t total = 0;
  for (int i = 0; i * 128 < n_pad; ++i) total += (int64_t)128 * (n_pad - 128 * i);
  return total;
}
__global__ void __launch_bounds__(256) k_pack_upper(double* __restrict__ S, int n_pad, double* __restrict__ P, int unpack) {
  const int row = blockIdx.x;                 // one workgroup per matrix row
  const int blk = row >> 7;
  const int c0 = blk << 7;
  const int width = n_pad - c0;
  // offset of block blk = sum_{j<blk} 128 (n_pad - 128 j) = 128 (blk n_pad - 64 blk (blk - 1))
  const size_t off = (size_t)128 * ((size_t)blk * n_pad - (size_t)64 * blk * (blk - 1)) + (size_t)(row - c0) * width;
  double* s = S + (size_t)row * n_pad + c0;
  double* p = P + off;
  for (int c = threadIdx.x * 2; c < width; c += 512) {
    if (unpack) *reinterpret_cast<double2*>(s + c) = *reinterpret_cast<const double2*>(p + c);
    else *reinterpret_cast<double2*>(p + c) = *reinterpret_cast<const double2*>(s + c);
  }
}
int launch_pack_upper(const double* S, int n_pad, double* P, int unpack, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_upper, dim3(n_pad), dim3(256), 0, s, const_cast<double*>(S), n_pad, P, unpack);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// diag(S) += lambda for real rows, = 1 for padding rows (multi-GPU path: after the all-reduce)
__global__ void k_finish_diag(double* __restrict__ S, int ld, int n_real, int n_pad, double lambda) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  if (i < n_real) S[(size_t)i * ld + i] += lambda;
  else S[(size_t)i * ld + i] = 1.0;
}
int launch_finish_diag(double* S, int ld, int n_real, int n_pad, double lambda, hipStream_t s) {
  hipLaunchKernelGGL(k_finish_diag, dim3((n_pad + 255) / 256), dim3(256), 0, s, S, ld, n_real, n_pad, lambda);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}
// out[0] = sum of all diagonal entries of the block-diagonal and dense parts (fixed-order reduction)
__global__ void __launch_bounds__(256) k_diag_sum(const double* __restrict__ Dblk, int bs, int nb, const double* __restrict__ Hdd,
                                                  int ld, int dd, double* __restrict__ out) {
  __shared__ double sh[256];
  double acc = 0.0;
  const int nblk = bs * nb;
  for (int i = threadIdx.x; i < nblk; i += 256) { int b = i / bs, k = i % bs; acc += Dblk[(size_t)b * bs * bs + k * bs + k]; }
  for (int i = threadIdx.x; i < dd; i += 256) acc += Hdd[(size_t)i * ld + i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = sh[0];
}
int launch_diag_sum(const double* Dblk, int bs, int nb, const double* Hdd, int ld, int dd, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_diag_sum, dim3(1), dim3(256), 0, s, Dblk, bs, nb, Hdd, ld, dd, out);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

}  // namespace cba
