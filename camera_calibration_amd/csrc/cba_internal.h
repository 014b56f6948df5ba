// Internal declarations shared by the HIP translation units of libcalib_ba_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/cba.h"
#include "model.hip.h"

namespace cba {

void set_error(const std::string& msg);
#define CBA_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      ::cba::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                     \
      return CBA_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

// Variable ordering of JointOptimizationState (joint_optimization.cc:49-59, 142-170).
struct Layout {
  int n_cameras, n_images, n_points;
  int rig_in_state;
  int first_rig_tr_global, first_camera_tr_rig, first_points, first_intrinsics;
  int intr_offset[16];
  int total_dof, block_size, n_blocks, block_dof, dense_dof;
  int localize_only, eliminate_points;
};

constexpr int kMaxCameras = 16;
constexpr int kMaxGridCols = 80;                       // 5 * 16
constexpr int kMaxCols = 6 + 6 + 3 + kMaxGridCols;     // pose + rig + point + grid
// Jacobian record per observation (doubles): [res 2][weight 1][pose 2x6][rig 2x6][point 2x3][grid 2xKg]
constexpr int kRecHeader = 3 + 12 + 12 + 6;

struct DevState {
  double* rig_tr_global = nullptr;   // 7N
  double* camera_tr_rig = nullptr;   // 7C
  double* points = nullptr;          // 3P
  double* grids[kMaxCameras] = {};   // per camera
};

// Device-visible description of a pass over the observations.
struct PassArgs {
  int64_t n_obs;
  int n_cameras;
  const float* obs_xy;
  const int* obs_point;
  const int* obs_image;
  const int* obs_camera;
  double* last_projection;
  const double* points;
  const double* itg;        // [(img*C + cam)*16]: q(4) t(3) R(9)
  const CamDev* cams;       // device array [C]
  double fd_delta;
  const int* pose_slot;     // block position of each imageset (rows of B sorted by image footprint) or null
  // straggler split: observations whose base projection exceeds the iteration cap of the one-lane kernel are listed by
  // it and finished by the 16-lanes-per-observation kernel (kernels_obs.hip); in the Jacobian pass that launch and the
  // finite-difference tasks of the listed observations run on a side stream while the main launch skips them
  const int* obs_list;      // null = all observations in order
  const int* obs_count;     // device: entries in obs_list (clamped to obs_list_cap)
  int obs_list_cap;
  const uint8_t* skip;      // main launch: per-observation "handled by the list launch" (or null)
  // Jacobian records: the finite-difference tasks of the grid parameters write d pixel / d parameter straight into the grid
  // part of their observation's record (k_assemble fills the header); null outside the Jacobian pass
  double* jrec;
  int rec_doubles;
  // cost pass behind a solve whose status the host has not read yet (cba_step, one GPU): a non-zero word here means the solve broke
  // down (zero / NaN pivot, NaN update) -- the kernels that write the warm-start cache then leave at once, so that a rejected-by-NaN
  // attempt touches nothing, exactly as the reference, which skips the cost pass for a NaN update (LV/lm_optimizer.h:905-913)
  const int* guard;
};

// ---- kernels_obs.hip ----
int launch_compose_poses(const DevState& st, int N, int C, double* itg, hipStream_t s);
int launch_tangents(const double* dir_grid, double* tang, int G, hipStream_t s);
// base projection of every observation; lanes that exceed the iteration cap are appended to defer_list (and marked in
// defer_skip) for launch_base_project_slow, which takes the list through PassArgs::obs_list / obs_count / obs_list_cap
int launch_base_project(const PassArgs& a, int model_mask, double* cost_vec, double* pixels, uint8_t* flags, int* defer_list,
                        int* defer_count, int defer_cap, uint8_t* defer_skip, int outer_cap, const uint8_t* fd_slow, hipStream_t s);
int launch_base_project_slow(const PassArgs& a, int model_mask, double* cost_vec, double* pixels, uint8_t* flags, hipStream_t s);
// redo / redo_count: device work list (redo_cap entries / one int) for the tasks that leave their staged patch; tasks that find
// the list full are counted in *redo_overflow
// schedule: 0 = pooled (workgroup task pool, one LM attempt per trip), 1 = one task per lane; same expressions in the same order,
// flags identical, Jacobian entries agree to ~1e-12 (0.004 % of them differ: the compiler contracts one multiply-add of the damped
// 2 x 2 solve differently in the two kernels; include/cba.h: cba_set_fd_schedule, tests/test_gpu_stragglers.py)
int launch_fd_tasks(const PassArgs& a, int model_mask, int tasks_per_obs, int localize_only, const double* pixels,
                    const uint8_t* flags, double* fd_out, uint8_t* fd_ok, int64_t* redo, int* redo_count, int redo_cap, int* redo_overflow,
                    hipStream_t s, int schedule = 0);
int launch_assemble(const PassArgs& a, const Layout& L, const DevState& st, int tasks_per_obs, int rec_doubles,
                    const double* pixels, uint8_t* flags, const double* fd_out, const uint8_t* fd_ok, double* jrec,
                    int* cells, uint8_t* fd_slow, hipStream_t s);
struct AccumTargets {
  double* Dblk; double* bblk; double* B; double* Hdd; double* bd;
};
int launch_accumulate(const PassArgs& a, const Layout& L, int rec_doubles, const uint8_t* flags, const double* jrec,
                      const int* cells, const uint32_t* pair_tables, const int* pair_counts, AccumTargets t,
                      const double* det_scale, int points_separate, hipStream_t s);
// terms with a pattern-point column, bucketed by (camera, point) (poses eliminated); key lists: cba_set_observations
int launch_accumulate_points(const PassArgs& a, const Layout& L, const std::vector<cba_camera>& cams, int rec_doubles, const uint8_t* flags,
                             const double* jrec, const int* cells, const int* key_start, const int* key_obs, AccumTargets t,
                             const double* det_scale, hipStream_t s);
// deterministic mode (cba_config.deterministic): fixed-point scale of a pass, conversion of an accumulated array
int launch_det_scale(int64_t n, int rec_doubles, int used_doubles, const uint8_t* flags, const double* jrec, unsigned long long* bits,
                     double* scale, hipStream_t s);
int launch_det_convert(double* p, size_t n, const double* det_scale, hipStream_t s);
int launch_accumulate_strips(const PassArgs& a, const Layout& L, int n_images, int rec_doubles, const uint8_t* flags, const double* jrec,
                             const int* cells, unsigned long long* band_mask, const int64_t* img_start, double* B, int ld, const double* det_scale,
                             hipStream_t s);
int launch_accumulate_cells(const PassArgs& a, const std::vector<cba_camera>& cams, const std::vector<int>& cell_base_host,
                            int rec_doubles, int ld, const uint8_t* flags, const double* jrec, const int* cells,
                            const int* cell_base, int* count, int* start, int* fill, int* order, double* Hdd,
                            int rig_row_first, const double* det_scale, double* bd, hipStream_t s);
// 8 outputs: [0] sum ref (valid), [1] sum test (valid), [2] masked ref, [3] masked test, [4] count both valid,
// [5] n valid ref, [6] n valid test, [7] n jac dropped (flags)
int launch_reduce_costs(const double* ref, const double* test, const uint8_t* flags, int64_t n, double* partials,
                        double* out8, hipStream_t s);
int launch_apply_update(const Layout& L, const std::vector<cba_camera>& cams, const DevState& in, const double* x,
                        DevState& out, const int* pose_slot, int* const* gperm, hipStream_t s);
int launch_update_direction_grid(const double* in, const double* x, int G, double* out, hipStream_t s);
// ---- kernels_fit.hip (grid-only LM) ----
int launch_fit_pass(bool jac, int gw, int gh, const double* grid, const double* tang, int64_t n, const double* gp,
                    const double* dirs, double* cost_vec, double* rec, int* keys, int* status, hipStream_t s);
int launch_fit_accumulate(int gw, int gh, int64_t n, const double* rec, const int* keys, int* count, int* start, int* fill,
                          int* order, double* H, int ld, double* b, hipStream_t s);
int launch_fit_set_rhs(double* S, int ld, const double* b, int n, hipStream_t s);
int launch_fit_diag_sum(const double* H, int ld, int n, double* out, hipStream_t s);
int launch_project_points(const CamDev* cam_dev, int model, int64_t n, const double* local, const double* init,
                          double* pixels, uint8_t* ok, hipStream_t s);
int launch_unproject(const CamDev* cam_dev, int model, int64_t n, const double* pixels, double* lines, double* jac,
                     uint8_t* ok, hipStream_t s);

// ---- kernels_linalg.hip ----
// Inverse of the (bs x bs) diagonal blocks with lambda added, and Dinv*b.
int launch_block_inverse(const double* Dblk, const double* bblk, double lambda, int bs, int nb, double* Dinv,
                         double* dinvb, int* status, hipStream_t s);
// y[k] = base[k] - sum_j M[k][j] * v[j]   (row dot products) -- pose back-substitution
int launch_gemv_n(const double* M, int K, int n, int ld, const double* v, const double* base, double* y,
                  hipStream_t s);
struct GemmStats { double seconds = 0, flops = 0, bytes = 0; int launches = 0; };
// In-place blocked LDL^T of a symmetric matrix stored "upper in row-major" (= lower in column-major).
struct LdltWorkspace {
  double* X = nullptr;       // X = D L of a super-panel's row strip [x_rows][ld]
  int x_rows = 0;
  double* invLt = nullptr;   // per 64-block: transposed inverse of the unit factor [kInner][kInner]
  // scheduling options (cba_solver_options): rows left to the final dataflow launch; back substitution as one dataflow launch
  int tail_rows = 0;         // rows left to the final dataflow launch; 0 = the schedule's default (ldlt_tail_rows)
  bool back_dataflow = true;
  double* dvec = nullptr;    // n
  int* status = nullptr;
  // the device's side streams (shared, not owned): high-priority / two plain ones.  The factorisation itself runs on the caller's
  // stream; users: the exchanges of the distributed variant, the Jacobian pass' side work (cba_api.hip)
  hipStream_t panel_stream = nullptr, mid_stream = nullptr, far_stream = nullptr;
  hipEvent_t ev_strip = nullptr, ev_mid = nullptr;
  size_t n_alloc = 0;
  // kernel-only timing of the 128 x 128 GEMM launches of the factorisation (bulk and row-strip updates): event pairs
  // on the stream of each launch, read back by the caller after the step (ldlt_collect_spans)
  struct Span { hipEvent_t e0 = nullptr, e1 = nullptr; double flops = 0; bool masked_update = false; };
  std::vector<Span> spans;
  int spans_used = 0;
  // persistent tail launch (ldlt_tail): flags hold the number of the call that set them, nothing is cleared between calls
  unsigned* tail_flags = nullptr;    // tile flags [tail_rows_cap / 64][n / 64], then diag / upre / part flags [n / 64] each
  unsigned* tail_ctrl = nullptr;     // tickets, abort flag, role tickets, chain CU
  bool tail_ctrl_clean = false;      // the control words are already zero for the next dataflow launch (ldlt_clear_ctrl)
  unsigned tail_epoch = 0;
  int tail_rows_cap = 0;             // largest tail this workspace has flags for
  double* back_xe = nullptr;         // back substitution (k_back_dataflow): {value, tag} pairs, 2 * n doubles
  unsigned long long back_epoch = 0;
  hipEvent_t tail_e0 = nullptr, tail_e1 = nullptr;   // span of the last tail launch (statistics only)
  bool tail_timed = false;
};
// flag_rows_blocks: block rows a dataflow launch may cover (0 = the dense schedule's own maximum)
int ldlt_workspace_alloc(LdltWorkspace& w, int n, int flag_rows_blocks = 0);
// adds the GEMM launches timed since the last call to `st` (waits for them)
int ldlt_collect_spans(LdltWorkspace& w, GemmStats* st);
void ldlt_workspace_free(LdltWorkspace& w);
// k_begin: rows above it are factored already and their update is applied (the border of the grid-first order)
int ldlt_factor(double* S, int n, int ld, LdltWorkspace& w, hipStream_t s, GemmStats* trailing_stats, int k_begin = 0);
// x of L^T x = z (z = column zcol of S, forward-substituted by the factorisation); rowmask: optional block-sparsity of the factor's rows
int ldlt_back_solve(const double* S, int n_fact, int ld, int zcol, const LdltWorkspace& w, double* x, hipStream_t s,
                    const unsigned long long* rowmask = nullptr, int mask_words = 0);
// Grid-first elimination (gridfirst_plan.h): device copies of the plan's arrays
struct GfTask; struct GfIval; struct GfChain;
struct GfDevice {
  GfTask* tasks = nullptr; GfIval* ivals = nullptr; GfChain* chains = nullptr;
  int n_tasks0 = 0, n_tasks1 = 0, n_chains = 0;
  int nbg = 0, nbf = 0;
  unsigned long long* rowmask = nullptr; int mask_words = 0;      // static structure of every factored block row (the plan's)
  double flops_grid = 0;
  // per Jacobian pass (launch_gf_activity): activity of the grid x border tiles and what is derived from it
  unsigned long long* act = nullptr; int act_words = 0; int n_act_tiles = 0;
  unsigned long long* gridrow = nullptr;          // [nbg][act_words]: structure of the grid x grid factor's rows (closure of the activity)
  unsigned long long* kmask = nullptr; int kmask_words = 0;      // border update: [absolute 128-column tile][words], bit = 16-row K slab
  unsigned long long* rowmask_dyn = nullptr;      // back substitution: rowmask with the border bits of the grid rows from `act`
};
// F = [grid | border] in the plan's order, ld = its n_pad; Xb: (rows of the grid part) x (ld - Gf) panel buffer (zero outside the
// tiles the launch writes); kmask: optional block-sparsity of the border update (null = dense); tile_list: optional order of its
// upper tiles, tile_list_entries entries (tm, tn, s0, s1) relative to the border -- every upper tile once (s1 = 0) or as parts whose K-slab
// ranges [s0, s1) cover all slabs (added atomically), tm = -1 = an empty slot
int ldlt_factor_gridfirst(double* F, int n_fact, int ld, const GfDevice& g, double* Xb, int ldxb, LdltWorkspace& w, hipStream_t s,
                          GemmStats* st, const unsigned long long* kmask, int kmask_words, const int* tile_list, int tile_list_entries);
// Rows that the final dataflow launch factors (w.tail_rows clamped to the workspace's flag storage)
int ldlt_tail_rows(const LdltWorkspace& w, int world = 1);
int ldlt_clear_ctrl(LdltWorkspace& w, hipStream_t s);
// milliseconds of the last tail launch (waits for it); 0 when there was none
double ldlt_tail_last_ms(LdltWorkspace& w);
// Distributed variant (cba_config.distributed_solve): S holds this rank's PARTIAL reduced system on entry; the collectives are
// blocking host calls.  `send` / `recv`: device staging buffers of at least ldlt_dist_buffer_doubles(n_pad, world) doubles each.
struct DistComm {
  int rank = 0, world = 1;
  cba_collective_fn collective = nullptr; void* collective_user = nullptr;   // may be null: emulated with `allreduce`
  cba_allreduce_fn allreduce = nullptr; void* allreduce_user = nullptr;
  double* send = nullptr; double* recv = nullptr; size_t buf_doubles = 0;
};
size_t ldlt_dist_buffer_doubles(int n_pad, int world);
int ldlt_factor_distributed(double* S, int n, int ld, LdltWorkspace& w, hipStream_t s, const DistComm& c, GemmStats* trailing_stats);

}  // namespace cba
