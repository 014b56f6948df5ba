// Host-side plan of the GRID-FIRST elimination order (cba_solver_options.elimination).
//
// The reference eliminates the 6 x 6 pose blocks and factors the dense rest (LV/lm_optimizer.h:1247-1369,
// APP/bundle_adjustment/joint_optimization.cc:794-804).  (H + lambda I) x = b has one solution, so any exact elimination order
// gives the same x; this plan describes another one:
//
//   * an observation touches a 4 x 4 window of control points (APP/models/central_grid.h:199-209, noncentral_generic.h:224-283), so
//     two grid unknowns couple only if their control points are at most 3 apart in both grid directions: numbered along the SHORT
//     grid dimension the grid x grid block of J^T J is banded (half-bandwidth (3 short + 3) ppg + ppg - 1: 367 of 10 080 at
//     BASELINE configs[1]);
//   * the grid is eliminated FIRST by a block-sparse LDL^T, the dense border [rig | points | poses] (6 N + 3 P + 6 C unknowns: 5 445
//     against the 12 525 of the pose-first order) afterwards;
//   * a banded factorisation is one chain of n dependent pivots.  The long grid dimension is therefore cut into `strips` groups of
//     grid lines separated by 3-line separators; strips do not couple with each other, so every strip is a pivot chain of its own
//     (they run side by side in one dataflow launch), and the separators are eliminated after all strips of their camera.
//
// Everything here is STATIC structure (grid geometry only): which 64 x 64 tiles of the full normal matrix F, ordered
// [grid camera 0 | grid camera 1 | ... | rig | points | poses | right-hand side], can be non-zero in the factor, in what order the
// dataflow launch hands them out, and over which earlier block rows each tile accumulates.  Border columns are treated as dense.
// Pure host code, no device access: the CPU tests replay the task list with numpy (tests/test_gridfirst_plan.py).
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/cba.h"

namespace cba {

struct GfChain { int r0, r1, dep, pad; };            // block rows [r0, r1); dep: block r0 has predecessors (its diagonal tile comes from a PARTFULL task)
struct GfTask { int kind_n, r, c, iv0; };            // kind_n = kind | (number of K intervals << 8); kinds: 0 PRE(r): tile (r, r + 1); 1 PART(c): diagonal
                                                     // tile (c, c) less the rows below c - 1; 2 REG(r, c); 3 PARTFULL(c): diagonal tile (c, c), all rows;
                                                     // 4 REG2(r, c): the border tiles (r, c) and (r, c + 1) of one 128-column tile in one task
struct GfIval { int k0, k1; };                       // block rows [k0, k1)

struct GfPlan {
  int n_cameras = 0, n_images = 0, n_points = 0;
  int strips[16] = {};                               // per camera
  std::vector<std::vector<int>> gperm;               // per camera: control point (gx + gy * gw) -> rank in the engine's elimination order
  // layout of F
  int G = 0;            // grid unknowns of all cameras
  int Gf = 0;           // rows of the grid part of F: strip groups padded to 64, the whole part to 128
  int n_rp = 0;         // rig (6 C if C > 1) + point (3 P) unknowns
  int n_border = 0;     // n_rp + 6 N
  int n_fact = 0;       // factored rows (multiple of 64)
  int n_pad = 0;        // leading dimension (multiple of 128); the right-hand side is column n_pad - 1
  std::vector<int> f_of_grid;     // [G]  engine grid index (dense column - first grid column) -> row of F
  std::vector<int> grid_of_f;     // [Gf] row of F -> engine grid index, -1 = padding (identity row)
  // block structure (64 x 64 tiles)
  int nbg = 0;          // block rows of the grid part (Gf / 64)
  int nbf = 0;          // block rows factored (n_fact / 64)
  int ntc = 0;          // block columns (n_pad / 64)
  std::vector<GfChain> chains;
  std::vector<GfTask> tasks;        // list 0 (n_tasks0 entries: what the chains wait for), then list 1 (border tiles of the row strips)
  int n_tasks0 = 0;
  std::vector<GfIval> ivals;
  int mask_words = 0;
  std::vector<uint64_t> rowmask;  // [nbf][mask_words]: bit c = tile (r, c), c > r, can be non-zero in the factor
  int grid_words = 0;
  std::vector<uint64_t> gridrow;  // [nbg][grid_words]: the grid x grid part of rowmask (bit c, r < c < nbg): what the fill of a border
                                  // column follows (closure of the row strips' activity, k_gf_close / gf_order_imagesets)
  // 64 x 64 tiles of the grid x grid part that hold input data (the forming kernel copies these; fill-only tiles are zeroed)
  std::vector<int> grid_tiles;    // pairs (r, c), r <= c < nbg, every tile of rowmask (+ the diagonal tiles)
  double flops_grid = 0, flops_update = 0, flops_border = 0;   // model: dataflow launch of the grid rows, border update, border factorisation
  int half_bandwidth = 0;         // largest |i - j| of a structural non-zero inside a strip
};

// strips_override: 0 = automatic, >= 1 = that many strips per camera (clamped to what the grid allows).  Returns CBA_OK / CBA_ERR_ARG.
// single_tile_tasks: 1 = every border tile of the row strips is a task of its own (REG); 0 (default) = the two tiles of a 128-column border
// tile share one task (REG2: half as many workgroup slots wait at the chains' frontiers, the A strip is fetched once).
int gf_build_plan(const cba_camera* cams, int n_cameras, int n_images, int n_points, int strips_override, int single_tile_tasks, GfPlan* out);

// Order of the imagesets' pose columns in the border (slot of every imageset).  first_rows[i]: bit set over the grid block rows
// (plan.grid_words words) that imageset i touches -- from the measured pixels, a prediction of the per-pass activity.  The border
// update skips, per 128-column tile (~21 imagesets) and 16-row slab, what NO column of the tile reaches, so imagesets are grouped
// greedily: a tile is seeded with the unplaced imageset that starts earliest in the elimination order and filled with the
// imagesets that enlarge the tile's union (closed under the fill of the grid factor) least.  first_col: border index of slot 0's
// first column (tiles are counted in border columns).
void gf_order_imagesets(const GfPlan& plan, const std::vector<uint64_t>& touched_rows, int n_images, int first_col, std::vector<int>* slot_of_image);

// flop model of the two elimination orders (dense border columns): used by the automatic choice
void gf_flop_model(const cba_camera* cams, int n_cameras, int n_images, int n_points, double* pose_first, double* grid_first);

}  // namespace cba
