// Grid-first elimination order (gridfirst_plan.h): the kernels around the factorisation --
//   k_gf_form     builds the full normal matrix F = [grid | rig | points | poses] (+ lambda on the diagonal, ones on padding rows,
//                 the right-hand side in the last column) from what the accumulation kernels wrote: the pose blocks D_i / b_i, the
//                 strips B (pose rows x dense columns) and the dense part H_dd / b_d (lm_optimizer_update_accumulator.h:108-155 is
//                 the reference's layout of the same four parts).  The factorisation works in place, so F is formed once per LM
//                 attempt (the pose-first path pays D^-1 B and the C read of the Schur product at the same point);
//   k_gf_scatter  takes the solution of F x = b back to the engine's x = [poses in slot order | dense columns].
// F keeps its upper triangle in row-major order like every symmetric matrix of the engine (kernels_linalg.hip).  Entries whose
// source is stored "the other way round" -- grid rows against rig / point / pose columns, rig / point rows against pose columns:
// H_dd and B are row-major with the EARLIER engine unknown as the row -- are transposed through LDS, so that both the reads and the
// writes of a tile move whole 512-byte rows.
#include "cba_internal.h"

namespace cba {

struct GfFormArgs {
  double* F; int ldf;
  int Gf, n_rp, n_border;
  const int* grid_of_f;          // [Gf] row of F -> engine grid index (dense column - n_rp), -1 = padding
  const double* Hdd; int ldh;    // engine dense order [rig | points | grid], upper, row-major
  const double* bd;
  const double* B;               // rows 6 slot + k, columns = engine dense order, row stride ldh
  const double* Dblk; const double* bblk;
  double lambda;
  const int2* tiles;             // 64 x 64 tiles (block row, block column) of F to form
  const unsigned long long* act; int act_words; int nbg;     // activity of the grid x border tiles (k_gf_touch); null = all
};

// class of a row / column of F: dense column (>= 0), pose row (-2 - row), padding (-1)
__device__ __forceinline__ int gf_class(const GfFormArgs& a, int f) {
  if (f < a.Gf) { const int e = a.grid_of_f[f]; return e < 0 ? -1 : a.n_rp + e; }
  const int b = f - a.Gf;
  if (b < a.n_rp) return b;
  if (b < a.n_border) return -2 - (b - a.n_rp);
  return -1;
}

__global__ void __launch_bounds__(256) k_gf_form(GfFormArgs a) {
  __shared__ double sT[64][65];
  __shared__ int s_ci[64], s_cj[64];
  const int2 tile = a.tiles[blockIdx.x];
  // a grid x border tile nothing touches in this pass (and no fill reaches): not formed, not read by anybody (every consumer
  // of the row strips honours the same activity bits)
  if (a.act && tile.x < a.nbg && tile.y >= a.nbg && !((a.act[(size_t)((tile.y - a.nbg) >> 1) * a.act_words + (tile.x >> 6)] >> (tile.x & 63)) & 1ull)) return;
  const int i0 = tile.x * 64, j0 = tile.y * 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < 64) s_ci[tid] = gf_class(a, i0 + tid);
  else if (tid < 128) s_cj[tid - 64] = gf_class(a, j0 + tid - 64);
  __syncthreads();
  const bool rhs_tile = j0 + 64 == a.ldf;
  // ---- transposed sources: one column of the tile per wavefront and step, lanes along the rows ----
  const int ci = s_ci[lane];
  for (int jj = wv; jj < 64; jj += 4) {
    const int cj = s_cj[jj];
    if (rhs_tile && jj == 63) continue;
    double v = 0.0;
    bool t = false;
    if (ci >= 0) {
      if (cj <= -2) { t = true; v = a.B[(size_t)(-2 - cj) * a.ldh + ci]; }                    // dense row x pose column
      else if (cj >= 0 && cj < ci) { t = true; v = a.Hdd[(size_t)cj * a.ldh + ci]; }            // grid row x rig / point column
    }
    if (t) sT[jj][lane] = v;
  }
  __syncthreads();
  // ---- rows of the tile: lanes along the columns ----
  const int fj = j0 + lane, cj = s_cj[lane];
  for (int ii = wv; ii < 64; ii += 4) {
    const int fi = i0 + ii, cr = s_ci[ii];
    double v = 0.0;
    if (fi > fj) v = 0.0;                                         // below the diagonal (diagonal tiles)
    else if (rhs_tile && lane == 63) {                            // right-hand side column (its own diagonal entry: 1)
      if (fi == fj) v = 1.0;
      else if (cr >= 0) v = a.bd[cr];
      else if (cr <= -2) v = a.bblk[-2 - cr];
    } else if (cr == -1 || cj == -1) v = (fi == fj) ? 1.0 : 0.0;  // padding rows / columns: identity
    else if (cr >= 0) {
      if (cj <= -2 || cj < cr) v = sT[lane][ii];
      else v = a.Hdd[(size_t)cr * a.ldh + cj] + (fi == fj ? a.lambda : 0.0);
    } else {                                                      // pose row: pose column of the same imageset, or zero
      const int pr = -2 - cr, pc = -2 - cj;
      if (cj <= -2 && pr / 6 == pc / 6) v = a.Dblk[(size_t)(pr / 6) * 36 + (pr % 6) * 6 + (pc % 6)] + (fi == fj ? a.lambda : 0.0);
    }
    a.F[(size_t)fi * a.ldf + fj] = v;
  }
}

int launch_gf_form(double* F, int ldf, int Gf, int n_rp, int n_border, const int* grid_of_f, const double* Hdd, int ldh, const double* bd,
                   const double* B, const double* Dblk, const double* bblk, double lambda, const int* tiles, int n_tiles,
                   const unsigned long long* act, int act_words, hipStream_t s) {
  if (n_tiles <= 0) return CBA_OK;
  GfFormArgs a{};
  a.act = act; a.act_words = act_words; a.nbg = Gf / 64;
  a.F = F; a.ldf = ldf; a.Gf = Gf; a.n_rp = n_rp; a.n_border = n_border; a.grid_of_f = grid_of_f;
  a.Hdd = Hdd; a.ldh = ldh; a.bd = bd; a.B = B; a.Dblk = Dblk; a.bblk = bblk; a.lambda = lambda;
  a.tiles = reinterpret_cast<const int2*>(tiles);
  hipLaunchKernelGGL(k_gf_form, dim3((unsigned)n_tiles), dim3(256), 0, s, a);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ---- activity of the row strips (per Jacobian pass) ----
// Which 64-row blocks of the grid part can be non-zero in a 128-column tile of the border: the blocks an observation of that
// tile's unknowns touches (its pose, its pattern point, its camera's rig pose x the 4 x 4 control patch under its PROJECTED pixel:
// state-dependent, hence per pass), closed under the fill of the grid x grid factor.  An imageset covers part of the image, so its
// pose columns are empty in the strips it does not reach -- about half of the border update's products at BASELINE configs[1].
// Granularity 64 rows x 128 columns: what the border update (128 x 128 tiles, 16-row K slabs) can skip.  The forming kernel, the
// block-sparse launch, the border update and the back substitution all read the SAME bits: a tile is formed, factored and used by
// all of them or by none (stale contents of unused tiles are never read).
struct GfTouchArgs {
  const int* obs_point; const int* obs_image; const int* obs_camera; const uint8_t* flags; const int* cells; const int64_t* img_start;
  const int* pose_slot; const CamDev* cams; const int* f_of_grid;
  int n_rp, rig_dof, n_tiles, words;
  unsigned long long* act;
};
__global__ void __launch_bounds__(256) k_gf_touch(GfTouchArgs a) {
  extern __shared__ unsigned long long sbits[];
  const int img = blockIdx.x, tid = threadIdx.x;
  const int nw = a.n_tiles * a.words;
  for (int i = tid; i < nw; i += 256) sbits[i] = 0ull;
  __syncthreads();
  const int64_t o0 = a.img_start[img], o1 = a.img_start[img + 1];
  const int slot = a.pose_slot ? a.pose_slot[img] : img;
  for (int64_t o = o0 + tid; o < o1; o += 256) {
    if (!(a.flags[o] & 2)) continue;                      // no Jacobian: nothing accumulated
    const int cam = a.obs_camera[o];
    const CamDev cd = a.cams[cam];
    const int ppg = cd.params_per_point;
    const int cx0 = a.cells[2 * o], cy0 = a.cells[2 * o + 1];
    int tl[6];
    tl[0] = (a.n_rp + 6 * slot) >> 7; tl[1] = (a.n_rp + 6 * slot + 5) >> 7;
    const int pb = a.rig_dof + 3 * a.obs_point[o];
    tl[2] = pb >> 7; tl[3] = (pb + 2) >> 7;
    tl[4] = a.rig_dof ? (6 * cam) >> 7 : tl[0]; tl[5] = a.rig_dof ? (6 * cam + 5) >> 7 : tl[0];
    // The block rows of the 4 x 4 control patch first -- a handful of bits in one to three words (a patch can straddle a strip, its
    // separator and the next strip) -- then ONE LDS atomic per tile and word: the lanes of a wavefront are consecutive observations
    // of one imageset, i.e. the same pose tile and neighbouring rows (one atomic per control point: 0.28 ms at BASELINE configs[1]).
    int wid[4] = {-1, -1, -1, -1};
    unsigned long long msk[4] = {0ull, 0ull, 0ull, 0ull};
    auto put = [&](int r) {
      const int w = r >> 6;
      const unsigned long long bit = 1ull << (r & 63);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (wid[i] == w || wid[i] < 0) { wid[i] = w; msk[i] |= bit; return; }
#pragma unroll
      for (int q = 0; q < 6; ++q) atomicOr(&sbits[tl[q] * a.words + w], bit);      // (a fifth word: not with 4 x 4 patches)
    };
    for (int k = 0; k < 16; ++k) {
      const int cx = cx0 + (k & 3), cy = cy0 + (k >> 2);
      if (cx < 0 || cy < 0 || cx >= cd.gw || cy >= cd.gh) continue;
      const int seq = cx + cy * cd.gw;
      const int e = cd.intr_offset - a.n_rp + ppg * (cd.gperm ? cd.gperm[seq] : seq);
      const int r0 = a.f_of_grid[e] >> 6, r1 = a.f_of_grid[e + ppg - 1] >> 6;
      put(r0);
      if (r1 != r0) put(r1);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      if (q > 0 && tl[q] == tl[q - 1]) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (wid[i] >= 0) atomicOr(&sbits[tl[q] * a.words + wid[i]], msk[i]);
    }
  }
  __syncthreads();
  for (int i = tid; i < nw; i += 256)
    if (sbits[i]) atomicOr(&a.act[i], sbits[i]);
}
// closure under the fill of the grid x grid factor: L_r'c != 0 if L_rc != 0 and L_rr' != 0 (r < r').  gridrow[r]: bits r' > r with
// tile (r, r') of the grid part in the plan's structure.  One lane per border tile; the last tile (right-hand side) is all rows.
__global__ void k_gf_close(unsigned long long* __restrict__ act, int n_tiles, int words, int nbg, const unsigned long long* __restrict__ gridrow) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tiles) return;
  unsigned long long w[16];
  for (int i = 0; i < words; ++i) w[i] = (t == n_tiles - 1) ? ~0ull : act[(size_t)t * words + i];
  for (int r = 0; r < nbg; ++r)
    if ((w[r >> 6] >> (r & 63)) & 1ull)
      for (int i = r >> 6; i < words; ++i) w[i] |= gridrow[(size_t)r * words + i];
  for (int i = 0; i < words; ++i) {
    unsigned long long v = w[i];
    if (64 * i + 64 > nbg) v &= (64 * i >= nbg) ? 0ull : (~0ull >> (64 - (nbg - 64 * i)));
    act[(size_t)t * words + i] = v;
  }
}
// K-slab mask of the border update (one bit per 16-row slab = a quarter of a block row; tile index = absolute column / 128) and
// the row masks of the back substitution (static structure of the plan with the border bits of the grid rows replaced)
__global__ void k_gf_masks(const unsigned long long* __restrict__ act, int n_tiles, int words, int nbg, int nbf, int tile0,
                           unsigned long long* __restrict__ kmask, int kwords, const unsigned long long* __restrict__ rowmask_static,
                           unsigned long long* __restrict__ rowmask, int mask_words) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_tiles * kwords) {
    const int t = i / kwords, kw = i - t * kwords;
    // slabs 64 kw ... 64 kw + 63 = block rows 16 kw ... 16 kw + 15
    const int r0 = 16 * kw;
    unsigned long long bits16 = 0;
    if (r0 < nbg) {
      bits16 = act[(size_t)t * words + (r0 >> 6)] >> (r0 & 63);          // r0 is a multiple of 16: the 16 bits sit in one word
      bits16 &= 0xffffull;
    }
    unsigned long long out = 0;
    for (int b = 0; b < 16; ++b)
      if ((bits16 >> b) & 1ull) out |= 0xfull << (4 * b);
    kmask[(size_t)(tile0 + t) * kwords + kw] = out;
  }
  const int j = i - n_tiles * kwords;
  if (j >= 0 && j < nbf * mask_words) {
    const int r = j / mask_words, w = j - r * mask_words;
    unsigned long long v = rowmask_static[j];
    if (r < nbg) {
      for (int b = 0; b < 64; ++b) {
        const int c = 64 * w + b;
        if (c < nbg || c >= nbf) continue;
        const unsigned long long on = (act[(size_t)((c - nbg) >> 1) * words + (r >> 6)] >> (r & 63)) & 1ull;
        v = (v & ~(1ull << b)) | (on << b);
      }
    }
    rowmask[j] = v;
  }
}
int launch_gf_activity(const PassArgs& pa, const uint8_t* flags, const int* cells, const int64_t* img_start, int n_images, const int* f_of_grid,
                       int n_rp, int rig_dof, int n_tiles, int words, int nbg, int nbf, const unsigned long long* gridrow, unsigned long long* act,
                       unsigned long long* kmask, int kwords, int tile0, const unsigned long long* rowmask_static, unsigned long long* rowmask,
                       int mask_words, hipStream_t s) {
  if (words > 16) { set_error("grid-first: more than 1024 grid block rows"); return CBA_ERR_UNSUPPORTED; }
  CBA_HIP(hipMemsetAsync(act, 0, sizeof(unsigned long long) * (size_t)n_tiles * words, s));
  if (pa.n_obs > 0 && n_images > 0) {
    GfTouchArgs a{};
    a.obs_point = pa.obs_point; a.obs_image = pa.obs_image; a.obs_camera = pa.obs_camera; a.flags = flags; a.cells = cells; a.img_start = img_start;
    a.pose_slot = pa.pose_slot; a.cams = pa.cams; a.f_of_grid = f_of_grid;
    a.n_rp = n_rp; a.rig_dof = rig_dof; a.n_tiles = n_tiles; a.words = words; a.act = act;
    hipLaunchKernelGGL(k_gf_touch, dim3((unsigned)n_images), dim3(256), sizeof(unsigned long long) * (size_t)n_tiles * words, s, a);
  }
  hipLaunchKernelGGL(k_gf_close, dim3((n_tiles + 63) / 64), dim3(64), 0, s, act, n_tiles, words, nbg, gridrow);
  const int total = n_tiles * kwords + nbf * mask_words;
  hipLaunchKernelGGL(k_gf_masks, dim3((total + 255) / 256), dim3(256), 0, s, act, n_tiles, words, nbg, nbf, tile0, kmask, kwords, rowmask_static, rowmask, mask_words);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// x (engine layout: 6 N pose entries in slot order, then the dense columns [rig | points | grid]) from xF (rows of F)
__global__ void __launch_bounds__(256) k_gf_scatter(const double* __restrict__ xF, int Gf, int n_rp, int block_dof, int G,
                                                    const int* __restrict__ f_of_grid, double* __restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = block_dof + n_rp + G;
  if (i >= total) return;
  int f;
  if (i < block_dof) f = Gf + n_rp + i;
  else if (i < block_dof + n_rp) f = Gf + (i - block_dof);
  else f = f_of_grid[i - block_dof - n_rp];
  x[i] = xF[f];
}
int launch_gf_scatter(const double* xF, int Gf, int n_rp, int block_dof, int G, const int* f_of_grid, double* x, hipStream_t s) {
  const int total = block_dof + n_rp + G;
  hipLaunchKernelGGL(k_gf_scatter, dim3((total + 255) / 256), dim3(256), 0, s, xF, Gf, n_rp, block_dof, G, f_of_grid, x);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

}  // namespace cba
