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
                   const double* B, const double* Dblk, const double* bblk, double lambda, const int* tiles, int n_tiles, hipStream_t s) {
  if (n_tiles <= 0) return CBA_OK;
  GfFormArgs a{};
  a.F = F; a.ldf = ldf; a.Gf = Gf; a.n_rp = n_rp; a.n_border = n_border; a.grid_of_f = grid_of_f;
  a.Hdd = Hdd; a.ldh = ldh; a.bd = bd; a.B = B; a.Dblk = Dblk; a.bblk = bblk; a.lambda = lambda;
  a.tiles = reinterpret_cast<const int2*>(tiles);
  hipLaunchKernelGGL(k_gf_form, dim3((unsigned)n_tiles), dim3(256), 0, s, a);
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
