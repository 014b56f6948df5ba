// C-ABI implementation (include/cba.h): device-resident problem, LM control flow
// (LMOptimizer::OptimizeImpl, libvis/src/libvis/lm_optimizer.h:629-991 in the reference tree) and the
// stateless model / solver entry points.  Everything numerical runs in the HIP kernels of
// kernels_obs.hip / kernels_linalg.hip; there is no CPU fallback.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "cba_internal.h"
#include "gridfirst_plan.h"

namespace cba {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

// implemented in kernels_linalg.hip
int launch_dinv_times_B_ld(const double* Dinv, const double* B, int bs, int nb, int dd, int ld, double* W, hipStream_t s);
int launch_gemv_t_partial(const double* M, int K, int n, int ld, const double* v, double* partial_ws, hipStream_t s);
int launch_gemv_t_final(int n, const double* base, double* y, int ystride, const double* partial_ws, int n_zero, hipStream_t s);
int launch_gemv_t_strided(const double* M, int K, int n, int ld, const double* v, const double* base, double* y,
                          int ystride, double* partial_ws, hipStream_t s);
int gemv_t_workspace_doubles(int n);
int schur_gemm(const double* A, const double* B, int Kpad, int ldab, const double* Cin, double* C, int n_pad, int ld,
               int n_real, int add_diag, double lambda, const unsigned long long* kmask, hipStream_t s, const int* chunk_order = nullptr,
               int keep_col = -1);
int schur_chunk_count(int n_pad);
void schur_chunk_order(const unsigned long long* mask_host, int n_pad, int Kpad, int* order);
int schur_mask_words(int Kpad);
int schur_slab_rows();
int launch_touch_mask(const double* B, int Kpad, int n_pad, int ld, unsigned long long* mask, hipStream_t s);
int launch_finish_diag(double* S, int ld, int n_real, int n_pad, double lambda, hipStream_t s);
int launch_diag_sum(const double* Dblk, int bs, int nb, const double* Hdd, int ld, int dd, double* out, hipStream_t s);
int make_main_stream(hipStream_t* s);
int prepare_device_streams();
int64_t packed_upper_doubles(int n_pad);
int launch_pack_upper(const double* S, int n_pad, double* P, int unpack, hipStream_t s);

int launch_gf_form(double* F, int ldf, int Gf, int n_rp, int n_border, const int* grid_of_f, const double* Hdd, int ldh, const double* bd,
                   const double* B, const double* Dblk, const double* bblk, double lambda, const int* tiles, int n_tiles,
                   const unsigned long long* act, int act_words, hipStream_t s);
int launch_gf_activity(const PassArgs& pa, const uint8_t* flags, const int* cells, const int64_t* img_start, int n_images, const int* f_of_grid,
                       int n_rp, int rig_dof, int n_tiles, int words, int nbg, int nbf, const unsigned long long* gridrow, unsigned long long* act,
                       unsigned long long* kmask, int kwords, int tile0, const unsigned long long* rowmask_static, unsigned long long* rowmask,
                       int mask_words, hipStream_t s);
int launch_gf_scatter(const double* xF, int Gf, int n_rp, int block_dof, int G, const int* f_of_grid, double* x, hipStream_t s);

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void make_layout(const cba_config& cfg, Layout& L) {
  L.n_cameras = cfg.n_cameras; L.n_images = cfg.n_images; L.n_points = cfg.n_points;
  L.localize_only = cfg.localize_only ? 1 : 0;
  L.eliminate_points = cfg.eliminate_points ? 1 : 0;
  const int N = cfg.n_images, C = cfg.n_cameras, P = cfg.n_points;
  L.rig_in_state = C > 1;
  const int rig_dof = L.rig_in_state ? 6 * C : 0;
  L.first_rig_tr_global = L.eliminate_points ? 3 * P : 0;
  L.first_camera_tr_rig = L.first_rig_tr_global + 6 * N;
  L.first_points = L.eliminate_points ? 0 : L.first_camera_tr_rig + rig_dof;
  L.first_intrinsics = L.eliminate_points ? (L.first_camera_tr_rig + rig_dof) : (L.first_points + 3 * P);
  if (L.eliminate_points) { L.block_size = 3; L.n_blocks = P; } else { L.block_size = 6; L.n_blocks = N; }
  L.block_dof = L.block_size * L.n_blocks;
  int off = L.first_intrinsics;
  for (int c = 0; c < C; ++c) {
    L.intr_offset[c] = off - L.block_dof;  // dense column
    off += (cfg.cameras[c].model_type == CBA_CENTRAL_GENERIC ? 2 : 5) * cfg.cameras[c].grid_w * cfg.cameras[c].grid_h;
  }
  L.total_dof = L.localize_only ? L.first_intrinsics : off;
  L.dense_dof = L.total_dof - L.block_dof;
}

static void padded_dims(int dd, int* n_pad, int* n_fact) {
  int nf = round_up(dd, 64);
  int np = round_up(dd + 1, 128);
  if (nf >= np) np += 128;  // keep the right-hand-side column (n_pad - 1) outside the factored rows
  *n_pad = np; *n_fact = nf;
}

static CamDev make_camdev(const cba_camera& c, const double* grid, const double* tangents, int intr_offset,
                          const int* gperm = nullptr) {
  CamDev d;
  d.model_type = c.model_type;
  d.gw = c.grid_w; d.gh = c.grid_h;
  d.min_x = c.calib_min_x; d.min_y = c.calib_min_y; d.max_x = c.calib_max_x; d.max_y = c.calib_max_y;
  d.gsx = (double)((float)c.grid_w - 3.f); d.gsy = (double)((float)c.grid_h - 3.f);
  d.span_x = (double)(c.calib_max_x + 1 - c.calib_min_x);
  d.span_y = (double)(c.calib_max_y + 1 - c.calib_min_y);
  d.jscale_x = (double)(((float)c.grid_w - 3.f) / (float)(c.calib_max_x + 1 - c.calib_min_x));
  d.jscale_y = (double)(((float)c.grid_h - 3.f) / (float)(c.calib_max_y + 1 - c.calib_min_y));
  d.grid = grid; d.tangents = tangents; d.intr_offset = intr_offset; d.gperm = gperm;
  d.params_per_point = c.model_type == CBA_CENTRAL_GENERIC ? 2 : 5;
  return d;
}

static bool camera_ok(const cba_camera& c) {
  return (c.model_type == CBA_CENTRAL_GENERIC || c.model_type == CBA_NONCENTRAL_GENERIC) && c.grid_w >= 4 && c.grid_h >= 4 &&
         c.calib_max_x >= c.calib_min_x && c.calib_max_y >= c.calib_min_y;
}

}  // namespace cba

using namespace cba;

// Stage timers: HIP events on the stream the kernels run on, read back only at the end of the step (a wait on the
// host in the middle of a step would keep the next stage's launches from being queued behind the running one).
struct KernelTimer {
  struct Span { hipEvent_t e0 = nullptr, e1 = nullptr; };
  std::vector<Span> spans;     // event pairs, reused from step to step
  int used = 0;                // spans recorded since the last collect
  double seconds = 0, flops = 0, bytes = 0;
  int launches = 0;
};

constexpr int kSlowCapMin = 16384;   // capacity of the straggler list: max(this, n_obs / 8), set with the observations

struct cba_problem {
  cba_config cfg{};
  std::vector<cba_camera> cams;
  Layout L{};
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t n_obs = 0;
  bool have_obs = false, have_state = false, have_system = false;
  int model_mask = 0;
  int tasks_per_obs = 0, rec_doubles = 0;
  // observations
  float* obs_xy = nullptr; int* obs_point = nullptr; int* obs_image = nullptr; int* obs_camera = nullptr;
  double* last_projection = nullptr;
  // state (double buffered)
  DevState st[2];
  int cur = 0;
  double* itg = nullptr;
  double* tangents[kMaxCameras] = {};
  CamDev* cams_dev[2] = {nullptr, nullptr};
  // pass outputs
  double* cost_ref = nullptr; double* cost_test = nullptr; double* pixels = nullptr; uint8_t* flags = nullptr;
  double* fd_out = nullptr; uint8_t* fd_ok = nullptr; double* jrec = nullptr; int* cells = nullptr;
  uint32_t* pair_tables = nullptr; int* pair_counts = nullptr;
  int* pt_start = nullptr; int* pt_obs = nullptr;   // observations bucketed by (camera, pattern point): k_accumulate_points
  std::vector<int> cell_base_host; int* cell_base = nullptr; int* cell_count = nullptr; int* cell_start = nullptr; int* cell_fill = nullptr;
  int* cell_order = nullptr;
  // imageset -> position of its 6x6 block / rows of B.  Imagesets are sorted along a Z-order curve of the
  // centre of their observations so that the 16-row K slabs of the Schur product touch few grid tiles.
  std::vector<int> pose_slot_host; int* pose_slot = nullptr;
  // straggler split of the Jacobian pass (see PassArgs)
  uint8_t* slow_skip = nullptr; uint8_t* fd_slow = nullptr; int* slow_list = nullptr; int* slow_count = nullptr;
  int slow_cap = kSlowCapMin;
  int straggler_threshold = 8;    // outer projection iterations before an observation goes to the straggler kernel
  int fd_schedule = -1;           // finite-difference kernel: -1 = automatic (default), 0 = pooled tasks, 1 = one task per lane (cba_set_fd_schedule)
  int64_t* img_start = nullptr;          // first observation of every imageset (+ end), for the strip accumulation
  unsigned long long* band_mask = nullptr;   // per observation: column bands of B it touches
  // the side stream is the factorisation's far stream (idle during the Jacobian pass): the process must stay
  // within four HIP streams -- a fifth shares a hardware queue with another one and serialises the LDL^T streams
  // (measured twice, also with GPU_MAX_HW_QUEUES=8)
  hipEvent_t ev_aux0 = nullptr, ev_aux1 = nullptr, ev_aux2 = nullptr, ev_clear = nullptr, ev_mask = nullptr;
  // control point -> rank in the engine's tiled order of the grid unknowns, per camera (see build_grid_order)
  int* gperm[kMaxCameras] = {};
  std::vector<int> dense_perm_host;   // reference dense column -> engine dense column (identity outside the grids)
  double* red_partials = nullptr; double* red8 = nullptr;
  // system
  int n_pad = 0, n_fact = 0, Kpad = 0;
  double* Dblk = nullptr; double* bblk = nullptr; double* B = nullptr; double* Hdd = nullptr; double* bd = nullptr;
  double* Dinv = nullptr; double* dinvb = nullptr; double* W = nullptr; double* S = nullptr; bool S_owned = true; double* P = nullptr; bool P_owned = true; double* P2 = nullptr; size_t dist_buf_doubles = 0;
  double* x = nullptr; double* scal = nullptr; double* gemv_ws = nullptr;
  unsigned long long* kmask = nullptr;   // block-sparsity of B per (column tile, K slab), rebuilt after every accumulation
  unsigned long long* kmask_host = nullptr; size_t kmask_host_words = 0;   // pinned copy (flop count of the Schur product)
  // chunk order of the Schur launch (heaviest first) from the masks of the PREVIOUS solve: the sparsity of B only changes with
  // the validity flags, and the order is a scheduling hint (any permutation is correct)
  int* chunk_order = nullptr; int* chunk_order_host = nullptr; bool chunk_order_valid = false; unsigned chunk_order_age = 0;
  int* status = nullptr;
  LdltWorkspace ldlt;
  KernelTimer timers[8];     // 0 ... 4: see cba_kernel_stats; 5: Jacobian pass, 6: solves, 7: cost passes queued behind a solve (cba_report.t_jac / t_solve / t_cost)
  // deterministic mode (cba_config.deterministic): fixed-point scale of the current pass
  unsigned long long* det_bits = nullptr; double* det_scale = nullptr;
  // finite-difference kernel: work lists of the tasks that leave their staged patch (main launch / side-stream launch)
  int64_t* fd_redo[2] = {nullptr, nullptr}; int* fd_redo_count = nullptr;   // counts: [0] main list, [1] side-stream list, [2] tasks that found a list full
  int fd_redo_cap = 0;
  double last_lambda = 0;
  bool mask_pending = false;      // a touch-mask launch of the last Jacobian pass may still read B on the side stream
  double* pin_status = nullptr;   // pinned host memory: {status, ldlt status, x[0]} of the last solve
  double* pin_cost = nullptr;     // pinned host memory: the 8 reduced scalars of the Jacobian pass when their read is deferred
  double last_x0 = 0;     // x[0] of the last solve (read back with the status words: the NaN test of lm_optimizer.h:905 needs no second wait)
  // grid-first elimination order (cba_solver_options.elimination; gridfirst_plan.h): the full normal matrix F = [grid | rig | points |
  // poses] is formed from Dblk / B / Hdd per LM attempt and factored in place; S, W, Dinv and the touch masks are not allocated
  bool gridfirst = false;
  GfPlan gf;
  GfDevice gfd;
  double* F = nullptr;            // gf.n_pad x gf.n_pad, upper triangle, row-major
  double* Xb = nullptr;           // gf.Gf x (gf.n_pad - gf.Gf): X = D L of the border columns (B operand of the border update)
  double* xF = nullptr;           // gf.n_fact: solution in the order of F
  int* gf_tiles = nullptr; int n_gf_tiles = 0;        // tiles of F the forming kernel writes
  int* gf_grid_of_f = nullptr; int* gf_f_of_grid = nullptr;
  unsigned long long* gf_kmask_host = nullptr;    // pinned copy of the border update's K-slab masks (executed flops of the launch)
  double gf_update_flops = 0;                     // executed flops of the border update with the masks of the last pass
  // order of the border update's tiles, heaviest first, from the masks of the PREVIOUS solve (a scheduling hint: any permutation is
  // correct, and the activity hardly moves from pass to pass)
  int* gf_tile_list = nullptr; int* gf_tile_list_host = nullptr; bool gf_tile_list_valid = false; unsigned gf_tile_list_age = 0;
  int gf_tile_list_entries = 0;     // slots of the launch (eight interleaved per-XCD lists, padded)
  bool gf_tile_list_dirty = false;  // the host copy was rebuilt since the last upload
  size_t gf_tile_list_capacity = 0; // ints
};

namespace cba {


static int timer_begin(cba_problem* p, int which, hipStream_t s = nullptr) {
  KernelTimer& t = p->timers[which];
  if (t.used == (int)t.spans.size()) {
    KernelTimer::Span sp;
    CBA_HIP(hipEventCreate(&sp.e0)); CBA_HIP(hipEventCreate(&sp.e1));
    t.spans.push_back(sp);
  }
  CBA_HIP(hipEventRecord(t.spans[t.used].e0, s ? s : p->stream));
  return CBA_OK;
}
static int timer_end(cba_problem* p, int which, double flops, double bytes, int launches, hipStream_t s = nullptr) {
  KernelTimer& t = p->timers[which];
  CBA_HIP(hipEventRecord(t.spans[t.used].e1, s ? s : p->stream));
  t.used += 1;
  t.flops += flops; t.bytes += bytes; t.launches += launches;
  return CBA_OK;
}
// adds the elapsed times of the spans recorded since the last call (waits for them)
static int timers_collect(cba_problem* p) {
  for (KernelTimer& t : p->timers) {
    for (int i = 0; i < t.used; ++i) {
      CBA_HIP(hipEventSynchronize(t.spans[i].e1));
      float ms = 0;
      CBA_HIP(hipEventElapsedTime(&ms, t.spans[i].e0, t.spans[i].e1));
      t.seconds += ms * 1e-3;
    }
    t.used = 0;
  }
  GemmStats gs;                       // kernel-only spans of the factorisation's 128 x 128 GEMM launches
  { int rc = ldlt_collect_spans(p->ldlt, &gs); if (rc != CBA_OK) return rc; }
  p->timers[4].seconds += gs.seconds; p->timers[4].flops += gs.flops; p->timers[4].launches += gs.launches;
  return CBA_OK;
}

template <typename T>
static int dev_alloc(T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  CBA_HIP(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  return CBA_OK;
}
#define CBA_TRY(expr) do { int _rc = (expr); if (_rc != CBA_OK) return _rc; } while (0)

// device buffers of one stateless call: freed on every exit path
struct DevPool {
  std::vector<void*> ptrs;
  ~DevPool() { for (void* q : ptrs) if (q) hipFree(q); }
  template <typename T> int alloc(T** p, size_t n) {
    int rc = dev_alloc(p, n);
    if (rc == CBA_OK) ptrs.push_back(*p);
    return rc;
  }
};
struct LdltGuard { LdltWorkspace* w; ~LdltGuard() { if (w) ldlt_workspace_free(*w); } };
// cba_solver_options -> the workspace of a problem / call (zero fields keep the defaults)
static void apply_solver_options(LdltWorkspace& w, const cba_solver_options* o) {
  if (!o) return;
  if (o->factor_tail_rows > 0) w.tail_rows = o->factor_tail_rows;
  w.back_dataflow = o->back_substitution == 0;
}

static int alloc_state(cba_problem* p, DevState& s) {
  CBA_TRY(dev_alloc(&s.rig_tr_global, 7 * (size_t)p->L.n_images));
  CBA_TRY(dev_alloc(&s.camera_tr_rig, 7 * (size_t)p->L.n_cameras));
  CBA_TRY(dev_alloc(&s.points, 3 * (size_t)p->L.n_points));
  for (int c = 0; c < p->L.n_cameras; ++c) {
    size_t G = (size_t)p->cams[c].grid_w * p->cams[c].grid_h;
    CBA_TRY(dev_alloc(&s.grids[c], (p->cams[c].model_type == CBA_CENTRAL_GENERIC ? 3 : 6) * G));
  }
  return CBA_OK;
}

// Engine-internal order of a camera's grid unknowns: control points are numbered tile by tile (8x8
// points for the central model = 128 columns, 5x5 for the non-central one = 125 columns) instead of
// row by row.  An imageset's rows of B are non-zero on the control points under its footprint, a 2-D
// region of the grid; with 2-D tiles that region intersects about half as many 128-column tiles of
// the Schur product as with 1-D runs of a grid row, and the block-sparse K loop skips the rest.
// The order is internal: cba_debug_dump / cba_get_state present everything in the reference order.
static int build_grid_order(cba_problem* p) {
  const Layout& L = p->L;
  p->dense_perm_host.resize(L.dense_dof);
  for (int i = 0; i < L.dense_dof; ++i) p->dense_perm_host[i] = i;
  if (L.localize_only) return CBA_OK;
  for (int c = 0; c < L.n_cameras; ++c) {
    const int gw = p->cams[c].grid_w, gh = p->cams[c].grid_h;
    const int per = p->cams[c].model_type == CBA_CENTRAL_GENERIC ? 2 : 5;
    const int tile = per == 2 ? 8 : 5;
    std::vector<int> perm((size_t)gw * gh);
    int rank = 0;
    if (p->gridfirst) perm = p->gf.gperm[c];          // elimination order of the grid-first plan (strips, then separators)
    else
    for (int ty = 0; ty < gh; ty += tile)
      for (int tx = 0; tx < gw; tx += tile)
        for (int y = ty; y < std::min(gh, ty + tile); ++y)
          for (int x = tx; x < std::min(gw, tx + tile); ++x) perm[x + (size_t)y * gw] = rank++;
    CBA_TRY(dev_alloc(&p->gperm[c], perm.size()));
    CBA_HIP(hipMemcpy(p->gperm[c], perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice));
    for (size_t g = 0; g < perm.size(); ++g)
      for (int d = 0; d < per; ++d)
        p->dense_perm_host[L.intr_offset[c] + per * g + d] = L.intr_offset[c] + per * perm[g] + d;
  }
  return CBA_OK;
}

static int upload_camdevs(cba_problem* p, int which) {
  std::vector<CamDev> h(p->L.n_cameras);
  for (int c = 0; c < p->L.n_cameras; ++c)
    h[c] = make_camdev(p->cams[c], p->st[which].grids[c], p->tangents[c], p->L.intr_offset[c], p->gperm[c]);
  CBA_HIP(hipMemcpyAsync(p->cams_dev[which], h.data(), sizeof(CamDev) * h.size(), hipMemcpyHostToDevice, p->stream));
  CBA_HIP(hipStreamSynchronize(p->stream));
  return CBA_OK;
}

// -1 (default): pooled wherever the projections of one wavefront differ in length -- the non-central model (83 tasks per observation) and
// rigs: 9 - 11 % faster in the bench trajectories of BASELINE configs[3] / [2] -- and one task per lane for a single central-generic
// camera, where after the first iteration every task of an observation takes the same two outer iterations and the pool's bookkeeping
// costs 4 % (configs[1]; in the FIRST iteration from the perturbed state the pool wins there too, 1.46 -> 1.29 ms).
// profiles/r05_fd_schedules.txt, r05_fd_schedules_bench.txt
static int fd_schedule_of(const cba_problem* p) {
  if (p->fd_schedule >= 0) return p->fd_schedule;
  return (p->L.n_cameras == 1 && p->model_mask == 1) ? 1 : 0;
}
static PassArgs pass_args(cba_problem* p, int which) {
  PassArgs a;
  a.n_obs = p->n_obs; a.n_cameras = p->L.n_cameras;
  a.obs_xy = p->obs_xy; a.obs_point = p->obs_point; a.obs_image = p->obs_image; a.obs_camera = p->obs_camera;
  a.last_projection = p->last_projection;
  a.points = p->st[which].points; a.itg = p->itg; a.cams = p->cams_dev[which];
  a.fd_delta = p->cfg.numerical_diff_delta;
  a.pose_slot = p->pose_slot;
  a.obs_list = nullptr; a.obs_count = nullptr; a.obs_list_cap = 0; a.skip = nullptr;
  a.jrec = p->jrec; a.rec_doubles = p->rec_doubles;
  a.guard = nullptr;
  return a;
}

static int read_scalars(cba_problem* p, const double* dev, double* host, int n) {
  CBA_HIP(hipMemcpyAsync(host, dev, sizeof(double) * n, hipMemcpyDeviceToHost, p->stream));
  CBA_HIP(hipStreamSynchronize(p->stream));
  return CBA_OK;
}

static int allreduce(cba_problem* p, double* dev, int64_t count) {
  if (!p->cfg.allreduce) return CBA_OK;
  CBA_HIP(hipStreamSynchronize(p->stream));
  int rc = p->cfg.allreduce(dev, count, p->cfg.allreduce_user);
  if (rc != 0) { set_error("allreduce callback failed"); return CBA_ERR_STATE; }
  return CBA_OK;
}

// residual pass on state `which`; fills cost vector `cost_vec` and reduces to out8 (host)
static int residual_pass(cba_problem* p, int which, double* cost_vec, const int* guard = nullptr) {
  CBA_TRY(launch_compose_poses(p->st[which], p->L.n_images, p->L.n_cameras, p->itg, p->stream));
  PassArgs a = pass_args(p, which);
  a.guard = guard;
  CBA_TRY(launch_base_project(a, p->model_mask, cost_vec, p->pixels, p->flags, p->slow_list, p->slow_count, p->slow_cap, p->slow_skip, p->straggler_threshold, nullptr, p->stream));
  PassArgs as = a;
  as.obs_list = p->slow_list; as.obs_count = p->slow_count; as.obs_list_cap = p->slow_cap;
  CBA_TRY(launch_base_project_slow(as, p->model_mask, cost_vec, p->pixels, p->flags, p->stream));
  return CBA_OK;
}

static int jacobian_pass_and_accumulate(cba_problem* p, double* t_acc) {
  const Layout& L = p->L;
  const int w = p->cur;
  for (int c = 0; c < L.n_cameras; ++c)
    CBA_TRY(launch_tangents(p->st[w].grids[c], p->tangents[c], p->cams[c].grid_w * p->cams[c].grid_h, p->stream));
  CBA_TRY(launch_compose_poses(p->st[w], L.n_images, L.n_cameras, p->itg, p->stream));
  PassArgs a = pass_args(p, w);
  PassArgs as = a;
  as.obs_list = p->slow_list; as.obs_count = p->slow_count; as.obs_list_cap = p->slow_cap;
  a.skip = p->slow_skip;
  hipStream_t aux = p->ldlt.far_stream, clr = p->ldlt.mid_stream;
  if (p->mask_pending) {      // (two passes without a solve in between: the previous pass's mask launch reads the B this one rewrites)
    CBA_HIP(hipStreamWaitEvent(p->stream, p->ev_mask, 0));
    p->mask_pending = false;
  }
  const size_t bs = L.block_size, nb = L.n_blocks;
  CBA_HIP(hipMemsetAsync(p->fd_redo_count + 2, 0, sizeof(int), p->stream));      // tasks that found a follow-up list full, this pass
  CBA_TRY(launch_base_project(a, p->model_mask, p->cost_ref, p->pixels, p->flags, p->slow_list, p->slow_count, p->slow_cap, p->slow_skip, p->straggler_threshold, p->fd_slow, p->stream));
  // ... and the stragglers of the base projection (long projection chains, see k_base_project_slow) are finished there,
  // followed by their finite-difference tasks, underneath the main finite-difference launch
  CBA_HIP(hipEventRecord(p->ev_aux2, p->stream));
  CBA_HIP(hipStreamWaitEvent(aux, p->ev_aux2, 0));
  CBA_TRY(launch_base_project_slow(as, p->model_mask, p->cost_ref, p->pixels, p->flags, aux));
  CBA_TRY(launch_fd_tasks(as, p->model_mask, p->tasks_per_obs, L.localize_only, p->pixels, p->flags, p->fd_out, p->fd_ok, p->fd_redo[1], p->fd_redo_count + 1, p->fd_redo_cap,
                          p->fd_redo_count + 2, aux, fd_schedule_of(p)));
  CBA_HIP(hipEventRecord(p->ev_aux1, aux));
  CBA_TRY(timer_begin(p, 3));
  CBA_TRY(launch_fd_tasks(a, p->model_mask, p->tasks_per_obs, L.localize_only, p->pixels, p->flags, p->fd_out, p->fd_ok, p->fd_redo[0], p->fd_redo_count, p->fd_redo_cap,
                          p->fd_redo_count + 2, p->stream, fd_schedule_of(p)));
  CBA_TRY(timer_end(p, 3, 0, 0, 1));
  // Third stream: the accumulation targets are cleared (1.3 GB for H_dd at cfg 2) underneath the finite-difference launch.  Round 3
  // issued the memsets first, on the side stream: the 0.2 ms fill of H_dd then held the chip before the base projection of the pass
  // got a workgroup slot (profiles/r04_v2_step_timeline_cfg2.txt: base projection 0.25 ms after the tangents); queued behind the
  // VALU-bound FD kernel the fill's workgroups take slots as they come free.
  CBA_HIP(hipStreamWaitEvent(clr, p->ev_aux2, 0));            // behind the base projection of this pass
  CBA_HIP(hipMemsetAsync(p->Dblk, 0, sizeof(double) * nb * bs * bs, clr));
  CBA_HIP(hipMemsetAsync(p->bblk, 0, sizeof(double) * nb * bs, clr));
  if (L.eliminate_points)
    CBA_HIP(hipMemsetAsync(p->B, 0, sizeof(double) * (size_t)p->Kpad * p->n_pad, clr));
  else if (p->Kpad > L.block_dof)     // padding rows of B (the strips below overwrite everything else, zeros included)
    CBA_HIP(hipMemsetAsync(p->B + (size_t)L.block_dof * p->n_pad, 0, sizeof(double) * (size_t)(p->Kpad - L.block_dof) * p->n_pad, clr));
  CBA_HIP(hipMemsetAsync(p->Hdd, 0, sizeof(double) * (size_t)p->n_pad * p->n_pad, clr));
  CBA_HIP(hipMemsetAsync(p->bd, 0, sizeof(double) * (size_t)p->n_pad, clr));
  CBA_HIP(hipEventRecord(p->ev_clear, clr));
  CBA_HIP(hipStreamWaitEvent(p->stream, p->ev_aux1, 0));
  CBA_HIP(hipStreamWaitEvent(p->stream, p->ev_clear, 0));
  a.skip = nullptr;
  const bool side = !L.localize_only;      // the per-cell accumulation runs on the side stream
  // (Running the assembly / accumulation of one chunk of imagesets next to the finite-difference launches of the
  // next chunk was measured and gained nothing: the two share the same CUs and the sum stayed the same.)
  CBA_TRY(launch_assemble(a, L, p->st[w], p->tasks_per_obs, p->rec_doubles, p->pixels, p->flags, p->fd_out, p->fd_ok,
                          p->jrec, p->cells, p->fd_slow, p->stream));
  double t0 = now_s();
  CBA_TRY(timer_begin(p, 2));
  AccumTargets T{p->Dblk, p->bblk, p->B, p->Hdd, p->bd};
  Layout Lp = L;
  Lp.dense_dof = p->n_pad;  // Hdd / B use the padded leading dimension as row stride
  const double* det = p->cfg.deterministic ? p->det_scale : nullptr;
  if (det) CBA_TRY(launch_det_scale(p->n_obs, p->rec_doubles, p->rec_doubles, p->flags, p->jrec, p->det_bits, p->det_scale, p->stream));
  const int points_separate = (!L.eliminate_points && p->pt_start) ? 1 : 0;
  // The four accumulation kernels write disjoint parts of the system (or add atomically).  The per-cell kernel runs on the side
  // stream next to the others; the per-point kernel (120 KB of LDS per workgroup, one per CU) goes FIRST on the main stream, alone:
  // next to the strips kernel its workgroups rarely find a CU with that much LDS free and the launch takes 3.9 ms instead of
  // ~0.6 at cfg 3 (measured, profiles/r03_v4_bench_cfg3_kernel_stats.txt).
  if (side) {
    CBA_HIP(hipEventRecord(p->ev_aux0, p->stream));
    CBA_HIP(hipStreamWaitEvent(aux, p->ev_aux0, 0));
    CBA_TRY(launch_accumulate_cells(a, p->cams, p->cell_base_host, p->rec_doubles, p->n_pad, p->flags, p->jrec, p->cells, p->cell_base,
                                    p->cell_count, p->cell_start, p->cell_fill, p->cell_order, p->Hdd,
                                    (!L.eliminate_points && L.rig_in_state) ? L.first_camera_tr_rig - L.block_dof : -1, det, p->bd, aux));
    if (p->gridfirst) {
      // Grid-first order: which grid block rows each 128-column tile of the border can reach in THIS pass (the control patch of an
      // observation sits under its projected pixel), closed under the fill of the grid factor, and the masks derived from it -- on
      // the side stream behind the per-cell accumulation, underneath the strips kernel of the main stream (waited for at the end of the pass)
      const GfPlan& g = p->gf;
      GfDevice& d = p->gfd;
      CBA_TRY(launch_gf_activity(a, p->flags, p->cells, p->img_start, L.n_images, p->gf_f_of_grid, g.n_rp, L.rig_in_state ? 6 * L.n_cameras : 0,
                                 d.n_act_tiles, d.act_words, g.nbg, g.nbf, d.gridrow, d.act, d.kmask, d.kmask_words, g.Gf / 128, d.rowmask, d.rowmask_dyn,
                                 d.mask_words, aux));
      CBA_HIP(hipMemcpyAsync(p->gf_kmask_host, d.kmask, sizeof(unsigned long long) * (size_t)(g.n_pad / 128) * d.kmask_words, hipMemcpyDeviceToHost, aux));
    }
    CBA_HIP(hipEventRecord(p->ev_aux1, aux));
  }
  if (points_separate)
    CBA_TRY(launch_accumulate_points(a, Lp, p->cams, p->rec_doubles, p->flags, p->jrec, p->cells, p->pt_start, p->pt_obs, T, det, p->stream));
  if (!L.eliminate_points)   // B strips (plain stores), the remaining terms are added on top atomically
    CBA_TRY(launch_accumulate_strips(a, Lp, L.n_images, p->rec_doubles, p->flags, p->jrec, p->cells, p->band_mask, p->img_start, p->B,
                                     p->n_pad, det, p->stream));
  CBA_TRY(launch_accumulate(a, Lp, p->rec_doubles, p->flags, p->jrec, p->cells, p->pair_tables, p->pair_counts, T, det, points_separate, p->stream));
  if (side) CBA_HIP(hipStreamWaitEvent(p->stream, p->ev_aux1, 0));
  if (det) {   // fixed point -> fp64, in place
    CBA_TRY(launch_det_convert(p->Dblk, nb * bs * bs, det, p->stream));
    CBA_TRY(launch_det_convert(p->bblk, nb * bs, det + 1, p->stream));      // J^T r: second scale
    CBA_TRY(launch_det_convert(p->Hdd, (size_t)L.dense_dof * p->n_pad, det, p->stream));
    CBA_TRY(launch_det_convert(p->bd, (size_t)p->n_pad, det + 1, p->stream));
    CBA_TRY(launch_det_convert(p->B, (size_t)L.block_dof * p->n_pad, det, p->stream));   // strips store integers, the pose x rig atomics add to them
  }
  CBA_TRY(timer_end(p, 2, 0, 0, 1));
  if (t_acc) *t_acc += now_s() - t0;
  // The block-sparsity mask of B is only read by the Schur product: it is built on the side stream, underneath the cost reduction,
  // the block inverses and W = D^-1 B of the solve that follows (solve_system waits for that stream in front of the product).
  if (!p->gridfirst) {
    CBA_HIP(hipEventRecord(p->ev_aux2, p->stream));
    CBA_HIP(hipStreamWaitEvent(aux, p->ev_aux2, 0));
    CBA_TRY(launch_touch_mask(p->B, p->Kpad, p->n_pad, p->n_pad, p->kmask, aux));
    CBA_HIP(hipEventRecord(p->ev_mask, aux));
    p->mask_pending = true;
  }
  p->have_system = true;
  return CBA_OK;
}

// out (pinned host memory): the two status words and x[0]; guard (device): non-zero when the solve broke down or x[0] is NaN (the
// reference's NaN test, lm_optimizer.h:905) -- read by the cost pass queued behind this launch (PassArgs::guard)
__global__ void k_solve_status(const int* __restrict__ s0, const int* __restrict__ s1, const double* __restrict__ x, double* __restrict__ out,
                               int* __restrict__ guard) {
  if (threadIdx.x == 0) {
    const double x0 = x[0];
    out[0] = (double)*s0; out[1] = (double)*s1; out[2] = x0;
    *guard = (*s0 != 0 || *s1 != 0 || x0 != x0) ? 1 : 0;
  }
}
// Order in which the border update hands out its 128 x 128 tiles (GemmArgs::tile_list): workgroup b of the launch runs on XCD b % 8 and
// the dispatcher hands workgroups out strictly in order, so (a) the list as a whole is sorted by executed K slabs, heaviest first --
// list scheduling: the light tiles fill the gaps behind the heavy ones, and every XCD (every eighth entry) sees the same sequence of
// weights, which keeps the in-order dispatcher from waiting for one XCD -- and (b) inside a run of tiles of about the same weight (7 %
// buckets) the tiles are dealt so that one XCD walks a CONTIGUOUS piece of the run in row-major order: its tiles in flight share an A
// panel and neighbouring B panels in that XCD's L2 instead of 64 unrelated pairs (FETCH_SIZE of the launch:
// profiles/r06_update_tile_order.txt).  Short lists are padded with (-1, -1) (the workgroup leaves).  weight(tm, tn): executed K slabs
// (or anything proportional) of upper tile (tm, tn), tm <= tn < nt.  Host work; the list is uploaded by the next solve.
// A tile with all K slabs runs for most of the launch (2.7 of 3.5 ms at BASELINE configs[1]); tiles heavier than a third of the heaviest
// are therefore handed out in PARTS (K ranges with equal shares of the executed slabs, GemmArgs::tile_list) that add to C atomically --
// not in the deterministic mode, where the additions must keep one order.  split(tm, tn, target): the K slab in front of which
// `target` units of the tile's weight lie; n_slabs: K slabs of the launch.
template <class Weight, class Split>
static void gf_build_tile_list(cba_problem* p, int nt, int n_slabs, Weight weight, Split split) {
  struct Tile { int w, tm, tn, s0, s1; };
  std::vector<Tile> all;
  all.reserve((size_t)nt * (nt + 1));
  int w_max = 0;
  for (int tm = 0; tm < nt; ++tm)
    for (int tn = tm; tn < nt; ++tn) { const int w = weight(tm, tn); all.push_back(Tile{w, tm, tn, 0, 0}); w_max = std::max(w_max, w); }
  if (!p->cfg.deterministic) {
    // unit: a third of the heaviest tile (measured at BASELINE configs[1] / [2] / [3], launch ms with units of 1/2, 1/3, 1/4, 1/6:
    // 3.31 / 3.31 / 3.26 / 3.26, 12.17 / 12.03 / 12.09 / 12.18, 5.40 / 5.27 / 5.37 / 5.41; whole tiles: 3.50 / 12.34 / 5.53)
    const int unit = std::max(8, w_max / 3);
    const size_t n0 = all.size();
    for (size_t i = 0; i < n0; ++i) {
      const int w = all[i].w, parts = (w + unit - 1) / unit;
      if (parts < 2) continue;
      int prev = 0, done = 0;
      bool ok = true;
      std::vector<Tile> add;
      for (int q = 1; q < parts && ok; ++q) {
        const int target = (int)((long long)w * q / parts);
        const int sq = split(all[i].tm, all[i].tn, target);
        if (sq <= prev || sq >= n_slabs) { ok = false; break; }
        add.push_back(Tile{target - done, all[i].tm, all[i].tn, prev, sq});
        prev = sq; done = target;
      }
      if (!ok) continue;
      add.push_back(Tile{w - done, all[i].tm, all[i].tn, prev, n_slabs});
      all[i] = add[0];
      for (size_t q = 1; q < add.size(); ++q) all.push_back(add[q]);
    }
  }
  std::stable_sort(all.begin(), all.end(), [](const Tile& u, const Tile& v) { return u.w > v.w; });      // row-major among equals
  std::vector<Tile> lists[8];
  size_t i0 = 0;
  while (i0 < all.size()) {
    size_t i1 = i0 + 1;
    while (i1 < all.size() && (double)all[i1].w >= 0.93 * all[i0].w) ++i1;                               // one bucket
    std::stable_sort(all.begin() + i0, all.begin() + i1, [](const Tile& u, const Tile& v) { return u.tm != v.tm ? u.tm < v.tm : (u.tn != v.tn ? u.tn < v.tn : u.s0 < v.s0); });
    const size_t L = i1 - i0;
    int start = 0;
    for (int x = 1; x < 8; ++x) if (lists[x].size() < lists[start].size()) start = x;
    size_t pos = i0;
    for (int k = 0; k < 8; ++k) {
      const size_t len = L / 8 + ((size_t)k < L % 8 ? 1 : 0);
      std::vector<Tile>& dst = lists[(start + k) % 8];
      dst.insert(dst.end(), all.begin() + pos, all.begin() + pos + len);
      pos += len;
    }
    i0 = i1;
  }
  size_t longest = 0;
  for (int x = 0; x < 8; ++x) longest = std::max(longest, lists[x].size());
  if (4 * 8 * longest > p->gf_tile_list_capacity) return;      // (cannot happen with the sizing of cba_create; the previous list stays)
  p->gf_tile_list_entries = (int)(8 * longest);
  for (size_t i = 0; i < longest; ++i)
    for (int x = 0; x < 8; ++x) {
      const bool have = i < lists[x].size();
      int* e = p->gf_tile_list_host + 4 * (8 * i + x);
      e[0] = have ? lists[x][i].tm : -1; e[1] = have ? lists[x][i].tn : -1;
      e[2] = have ? lists[x][i].s0 : 0; e[3] = have ? lists[x][i].s1 : 0;
    }
  p->gf_tile_list_valid = true;
  p->gf_tile_list_dirty = true;
}
// Builds S (+ right-hand side in its last column) for `lambda`, factors and solves; x (device) = full update.
static int solve_finish(cba_problem* p);
// solve_enqueue queues the whole solve on the stream (no host wait; the status words, x[0] and the guard word are written by its
// last launch); solve_finish waits for the stream and turns the status into a return code.  cba_step queues the attempt's cost
// pass BETWEEN the two on one GPU (PassArgs::guard keeps that pass from running behind a broken solve).
static int solve_enqueue(cba_problem* p, double lambda, cba_report* rep);
static int solve_system(cba_problem* p, double lambda, cba_report* rep) {
  CBA_TRY(solve_enqueue(p, lambda, rep));
  return solve_finish(p);
}
// Grid-first order: F formed from the accumulated parts, block-sparse launch of the grid rows, border update, dense border,
// masked back substitution, x back in the engine's layout.  Timers: 0 = the border update (the K = Gf product), 1 = the whole
// factorisation, 6 = the solve.
static int solve_enqueue_gridfirst(cba_problem* p, double lambda) {
  const Layout& L = p->L;
  const GfPlan& g = p->gf;
  const int ld = g.n_pad;
  CBA_TRY(timer_begin(p, 6));
  CBA_HIP(hipMemsetAsync(p->status, 0, sizeof(int), p->stream));
  CBA_HIP(hipMemsetAsync(p->ldlt.status, 0, sizeof(int), p->stream));
  CBA_TRY(ldlt_clear_ctrl(p->ldlt, p->stream));
  const GfDevice& d = p->gfd;
  CBA_TRY(launch_gf_form(p->F, ld, g.Gf, g.n_rp, g.n_border, p->gf_grid_of_f, p->Hdd, p->n_pad, p->bd, p->B, p->Dblk, p->bblk, lambda,
                         p->gf_tiles, p->n_gf_tiles, d.act, d.act_words, p->stream));
  GemmStats gs;
  CBA_TRY(timer_begin(p, 1));
  const int* tile_list = nullptr;
  if (p->gf_tile_list_valid) {
    if (p->gf_tile_list_dirty) {      // (rebuilt by solve_finish behind a stream wait: nothing in flight reads the device copy)
      CBA_HIP(hipMemcpyAsync(p->gf_tile_list, p->gf_tile_list_host, sizeof(int) * 4 * (size_t)p->gf_tile_list_entries, hipMemcpyHostToDevice, p->stream));
      p->gf_tile_list_dirty = false;
    }
    tile_list = p->gf_tile_list;
  }
  CBA_TRY(ldlt_factor_gridfirst(p->F, g.n_fact, ld, d, p->Xb, ld - g.Gf, p->ldlt, p->stream, &gs, d.kmask, d.kmask_words, tile_list,
                                tile_list ? p->gf_tile_list_entries : 0));
  CBA_TRY(timer_end(p, 1, gs.flops, 0, gs.launches));
  CBA_TRY(ldlt_back_solve(p->F, g.n_fact, ld, ld - 1, p->ldlt, p->xF, p->stream, d.rowmask_dyn, d.mask_words));
  CBA_TRY(launch_gf_scatter(p->xF, g.Gf, g.n_rp, L.block_dof, g.G, p->gf_f_of_grid, p->x, p->stream));
  if (!p->pin_status) CBA_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->pin_status), 4 * sizeof(double)));
  hipLaunchKernelGGL(k_solve_status, dim3(1), dim3(64), 0, p->stream, p->status, p->ldlt.status, p->x, p->pin_status, p->status + 1);
  CBA_HIP(hipGetLastError());
  CBA_TRY(timer_end(p, 6, 0, 0, 1));
  return CBA_OK;
}
static int solve_enqueue(cba_problem* p, double lambda, cba_report* rep) {
  if (p->gridfirst) return solve_enqueue_gridfirst(p, lambda);
  const Layout& L = p->L;
  const int bs = L.block_size, nb = L.n_blocks, dd = L.dense_dof, ld = p->n_pad;
  const bool multi = p->cfg.allreduce != nullptr;
  CBA_TRY(timer_begin(p, 6));
  CBA_HIP(hipMemsetAsync(p->status, 0, sizeof(int), p->stream));
  CBA_HIP(hipMemsetAsync(p->ldlt.status, 0, sizeof(int), p->stream));
  CBA_TRY(launch_block_inverse(p->Dblk, p->bblk, lambda, bs, nb, p->Dinv, p->dinvb, p->status, p->stream));
  // Side stream, next to W = D^-1 B and the Schur product (MFMA-bound, bandwidth to spare): the partial sums of the right-hand
  // side B^T D^-1 b (one pass over B), the touch masks on their way to the host, and the control words of the factorisation's first
  // dataflow launch.  Round 3 had all three between the Schur product and the factorisation: 0.15 ms of small launches and gaps.
  const int mask_tiles = p->n_pad / 128, mask_words = schur_mask_words(p->Kpad);
  if (p->kmask_host_words < (size_t)mask_tiles * mask_words) {
    if (p->kmask_host) CBA_HIP(hipHostFree(p->kmask_host));
    p->kmask_host_words = (size_t)mask_tiles * mask_words;
    CBA_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->kmask_host), p->kmask_host_words * sizeof(unsigned long long)));
  }
  {
    hipStream_t side = p->ldlt.far_stream;
    CBA_HIP(hipEventRecord(p->ev_aux0, p->stream));
    CBA_HIP(hipStreamWaitEvent(side, p->ev_aux0, 0));
    // right-hand side: S[j][n_pad-1] = bd[j] - sum_k B[k][j] dinvb[k]; the Schur launch leaves that column alone (keep_col)
    CBA_TRY(launch_gemv_t_partial(p->B, L.block_dof, dd, ld, p->dinvb, p->gemv_ws, side));
    CBA_TRY(launch_gemv_t_final(dd, p->bd, p->S + (ld - 1), ld, p->gemv_ws, p->n_pad, side));      // padding rows of the column: zero
    // (algorithmic flops of the Schur launch = K slabs actually multiplied: counted on the host after the solve)
    CBA_HIP(hipMemcpyAsync(p->kmask_host, p->kmask, (size_t)mask_tiles * mask_words * sizeof(unsigned long long), hipMemcpyDeviceToHost, side));
    CBA_TRY(ldlt_clear_ctrl(p->ldlt, side));
    CBA_HIP(hipEventRecord(p->ev_aux1, side));
  }
  CBA_TRY(launch_dinv_times_B_ld(p->Dinv, p->B, bs, nb, dd, ld, p->W, p->stream));
  CBA_HIP(hipStreamWaitEvent(p->stream, p->ev_aux1, 0));      // side stream: the touch masks (Jacobian pass), the right-hand side column, the control words
  p->mask_pending = false;
  CBA_TRY(timer_begin(p, 0));
  // lambda on the diagonal / ones on the padding diagonal: single GPU: in the product; replicated multi-GPU solve: after the
  // all-reduce; distributed solve: rank 0 adds them to its partial system, the reduction carries them to the owners
  const bool dist = multi && p->cfg.distributed_solve && p->cfg.world_size >= 1;
  const int* chunk_order = nullptr;
  if (p->chunk_order && p->chunk_order_valid) {
    CBA_HIP(hipMemcpyAsync(p->chunk_order, p->chunk_order_host, sizeof(int) * schur_chunk_count(p->n_pad), hipMemcpyHostToDevice, p->stream));
    chunk_order = p->chunk_order;
  }
  CBA_TRY(schur_gemm(p->B, p->W, p->Kpad, ld, p->Hdd, p->S, p->n_pad, ld, dd, (!multi || (dist && p->cfg.rank == 0)) ? 1 : 0, lambda, p->kmask, p->stream, chunk_order, ld - 1));
  CBA_TRY(timer_end(p, 0, 0, 0, 1));

  if (multi && !dist) {
    CBA_TRY(launch_pack_upper(p->S, p->n_pad, p->P, 0, p->stream));
    CBA_TRY(allreduce(p, p->P, packed_upper_doubles(p->n_pad)));
    CBA_TRY(launch_pack_upper(p->S, p->n_pad, p->P, 1, p->stream));
    CBA_TRY(launch_finish_diag(p->S, ld, dd, p->n_pad, lambda, p->stream));
  }
  GemmStats gs;
  CBA_TRY(timer_begin(p, 1));
  if (dist) {
    DistComm c;
    c.rank = p->cfg.rank; c.world = p->cfg.world_size;
    c.collective = p->cfg.collective; c.collective_user = p->cfg.collective_user;
    c.allreduce = p->cfg.allreduce; c.allreduce_user = p->cfg.allreduce_user;
    c.send = p->P; c.recv = p->P2; c.buf_doubles = p->dist_buf_doubles;
    int rc = ldlt_factor_distributed(p->S, p->n_fact, ld, p->ldlt, p->stream, c, &gs);
    if (rc == CBA_ERR_STATE) set_error("distributed solve: a collective callback failed");
    CBA_TRY(rc);
  } else {
    CBA_TRY(ldlt_factor(p->S, p->n_fact, ld, p->ldlt, p->stream, &gs));
  }
  CBA_TRY(timer_end(p, 1, gs.flops, 0, gs.launches));
  CBA_TRY(ldlt_back_solve(p->S, p->n_fact, ld, ld - 1, p->ldlt, p->x + L.block_dof, p->stream));
  // block part: x_b = D^-1 b - W x_d      (lm_optimizer.h:1366-1367)
  CBA_TRY(launch_gemv_n(p->W, L.block_dof, dd, ld, p->x + L.block_dof, p->dinvb, p->x, p->stream));
  // the two status words and x[0] reach the host through ONE launch that writes pinned host memory (three device-to-host copies in
  // a row cost 20 us each in front of the host's decision)
  if (!p->pin_status) CBA_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->pin_status), 4 * sizeof(double)));
  hipLaunchKernelGGL(k_solve_status, dim3(1), dim3(64), 0, p->stream, p->status, p->ldlt.status, p->x, p->pin_status, p->status + 1);
  CBA_HIP(hipGetLastError());
  CBA_TRY(timer_end(p, 6, 0, 0, 1));
  (void)rep;
  return CBA_OK;
}
static int solve_finish(cba_problem* p) {
  const int mask_tiles = p->n_pad / 128, mask_words = schur_mask_words(p->Kpad);
  CBA_HIP(hipStreamSynchronize(p->stream));
  const int st[2] = {(int)p->pin_status[0], (int)p->pin_status[1]};
  p->last_x0 = p->pin_status[2];
  if (p->gridfirst) {
    // executed K slabs of the border update (masks of this pass, copied on the side stream during the pass): the launch's flops
    const GfPlan& g = p->gf;
    const int kw = p->gfd.kmask_words, t0 = g.Gf / 128, nt = (g.n_pad - g.Gf) / 128;
    double slabs = 0;
    for (int tm = 0; tm < nt; ++tm)
      for (int tn = tm; tn < nt; ++tn)
        for (int w = 0; w < kw; ++w)
          slabs += __builtin_popcountll(p->gf_kmask_host[(size_t)(t0 + tm) * kw + w] & p->gf_kmask_host[(size_t)(t0 + tn) * kw + w]);
    p->gf_update_flops = slabs * 2.0 * 128 * 128 * 16;
    {
      LdltWorkspace& w = p->ldlt;
      for (int i = 0; i < w.spans_used; ++i)
        if (w.spans[i].masked_update) { w.spans[i].flops = p->gf_update_flops; w.spans[i].masked_update = false; }
    }
    // Tile order of the NEXT border updates (gf_build_tile_list; host work while the device idles)
    // (cba_set_observations leaves a first list predicted from the measured pixels; the first solve's masks replace it, then every 8th)
    ++p->gf_tile_list_age;
    if (!p->gf_tile_list_valid || p->gf_tile_list_age == 1 || (p->gf_tile_list_age & 7) == 0)
      gf_build_tile_list(p, nt, g.Gf / 16, [&](int tm, int tn) {
        int sl = 0;
        for (int w = 0; w < kw; ++w) sl += __builtin_popcountll(p->gf_kmask_host[(size_t)(t0 + tm) * kw + w] & p->gf_kmask_host[(size_t)(t0 + tn) * kw + w]);
        return sl;
      }, [&](int tm, int tn, int target) {
        int seen = 0;
        for (int w = 0; w < kw; ++w) {
          unsigned long long bits = p->gf_kmask_host[(size_t)(t0 + tm) * kw + w] & p->gf_kmask_host[(size_t)(t0 + tn) * kw + w];
          const int c = __builtin_popcountll(bits);
          if (seen + c < target) { seen += c; continue; }
          for (int b = 0; b < 64; ++b)
            if ((bits >> b) & 1ull) { if (seen == target) return 64 * w + b; ++seen; }
          return 64 * (w + 1);
        }
        return 0;
      });
  }
  if (!p->gridfirst) {
    double slabs = 0;
    for (int tm = 0; tm < mask_tiles; ++tm)
      for (int tn = tm; tn < mask_tiles; ++tn)
        for (int w = 0; w < mask_words; ++w)
          slabs += __builtin_popcountll(p->kmask_host[(size_t)tm * mask_words + w] & p->kmask_host[(size_t)tn * mask_words + w]);
    const double tiles = mask_tiles * (mask_tiles + 1) / 2.0;
    // (host work while the device idles: only for the first solve and then every 16th -- the pattern hardly moves)
    if (p->chunk_order_host && (!p->chunk_order_valid || (++p->chunk_order_age & 15) == 0)) {
      schur_chunk_order(p->kmask_host, p->n_pad, p->Kpad, p->chunk_order_host);
      p->chunk_order_valid = true;
    }
    p->timers[0].flops += slabs * 2.0 * 128 * 128 * schur_slab_rows();
    p->timers[0].bytes += tiles * 2.0 * 128 * 128 * 8 + slabs * 2.0 * schur_slab_rows() * 128 * 8;
  }
  if (st[1] == 3) { set_error("reduced solve: a dataflow launch timed out waiting for another workgroup"); return CBA_ERR_TIMEOUT; }
  if (st[0] || st[1]) return CBA_ERR_NUMERIC;
  return CBA_OK;
}

}  // namespace cba

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

const char* cba_last_error(void) { return g_error.c_str(); }
const char* cba_version(void) { return "camera_calibration_amd 0.1 (gfx950)"; }

int32_t cba_elimination_order(const cba_problem* p, int32_t out[4]) {
  if (!p) return 0;
  if (out) {
    out[0] = p->gridfirst ? p->gf.strips[0] : 0;
    out[1] = p->gridfirst ? p->gf.n_border : p->L.dense_dof;
    out[2] = p->gridfirst ? p->gf.Gf : 0;
    out[3] = p->gridfirst ? (int32_t)p->gf.chains.size() : 1;
  }
  return p->gridfirst ? 2 : 1;
}
int32_t cba_total_dof(const cba_problem* p) { return p ? p->L.total_dof : 0; }
int32_t cba_dense_dof(const cba_problem* p) { return p ? p->L.dense_dof : 0; }
int32_t cba_jacobian_record_doubles(const cba_problem* p) { return p ? p->rec_doubles : 0; }

int64_t cba_reduce_buffer_doubles(const cba_config* config) {
  if (!config || !config->cameras) return 0;
  Layout L; make_layout(*config, L);
  int n_pad, n_fact; padded_dims(L.dense_dof, &n_pad, &n_fact);
  if (config->distributed_solve) return (int64_t)ldlt_dist_buffer_doubles(n_pad, config->world_size);
  return packed_upper_doubles(n_pad);
}

int cba_prepare_device(int32_t device) {
  setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);   // read by the HIP runtime when it initialises (see cba_problem)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device available (the engine has no CPU fallback)"); return CBA_ERR_HIP; }
  if (device < 0 || device >= ndev) { set_error("cba_prepare_device: bad device ordinal"); return CBA_ERR_ARG; }
  CBA_HIP(hipSetDevice(device));
  return prepare_device_streams();
}

int cba_create(const cba_config* config, cba_problem** out) {
  if (!config || !out || !config->cameras || config->n_cameras < 1 || config->n_cameras > kMaxCameras ||
      config->n_images < 0 || config->n_points < 0 || !(config->numerical_diff_delta > 0)) {
    set_error("cba_create: bad config"); return CBA_ERR_ARG;
  }
  for (int c = 0; c < config->n_cameras; ++c)
    if (!camera_ok(config->cameras[c])) { set_error("cba_create: bad camera"); return CBA_ERR_ARG; }
  if (config->allreduce && config->eliminate_points) { set_error("image sharding requires eliminate_points = 0"); return CBA_ERR_UNSUPPORTED; }
  if (config->distributed_solve) {
    // the distributed factorisation walks the column groups by rank: an invalid rank would loop for ever or leave groups unowned
    if (!config->allreduce) { set_error("cba_create: distributed_solve needs an allreduce callback"); return CBA_ERR_ARG; }
    if (config->world_size < 1 || config->rank < 0 || config->rank >= config->world_size) {
      set_error("cba_create: distributed_solve needs 0 <= rank < world_size"); return CBA_ERR_ARG;
    }
  }
  setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);   // read by the HIP runtime when it initialises (see cba_problem)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device available (the engine has no CPU fallback)"); return CBA_ERR_HIP; }
  if (config->device < 0 || config->device >= ndev) { set_error("cba_create: bad device ordinal"); return CBA_ERR_ARG; }
  CBA_HIP(hipSetDevice(config->device));
  cba_problem* p = new cba_problem();
  // every early return below destroys the half-built problem (device memory, events); released on success
  struct Guard { cba_problem* q; ~Guard() { if (q) cba_destroy(q); } } guard{p};
  p->cfg = *config;
  p->cams.assign(config->cameras, config->cameras + config->n_cameras);
  p->cfg.cameras = p->cams.data();
  p->device = config->device;
  make_layout(p->cfg, p->L);
  const Layout& L = p->L;
  if (L.total_dof <= 0) { set_error("empty problem"); return CBA_ERR_ARG; }
  CBA_TRY(make_main_stream(&p->stream));
  CBA_HIP(hipEventCreateWithFlags(&p->ev_aux0, hipEventDisableTiming));
  CBA_HIP(hipEventCreateWithFlags(&p->ev_aux1, hipEventDisableTiming));
  CBA_HIP(hipEventCreateWithFlags(&p->ev_aux2, hipEventDisableTiming));
  CBA_HIP(hipEventCreateWithFlags(&p->ev_clear, hipEventDisableTiming));
  CBA_HIP(hipEventCreateWithFlags(&p->ev_mask, hipEventDisableTiming));
  CBA_TRY(dev_alloc(&p->slow_count, 1));
  CBA_HIP(hipMemset(p->slow_count, 0, sizeof(int)));
  for (int c = 0; c < L.n_cameras; ++c) p->model_mask |= (p->cams[c].model_type == CBA_CENTRAL_GENERIC) ? 1 : 2;
  const int maxKg = L.localize_only ? 0 : ((p->model_mask & 2) ? 80 : 32);
  p->tasks_per_obs = 3 + maxKg;
  p->rec_doubles = kRecHeader + 2 * maxKg;
  CBA_TRY(alloc_state(p, p->st[0]));
  CBA_TRY(alloc_state(p, p->st[1]));
  CBA_TRY(dev_alloc(&p->itg, 16 * (size_t)L.n_images * L.n_cameras));
  for (int c = 0; c < L.n_cameras; ++c) CBA_TRY(dev_alloc(&p->tangents[c], 6 * (size_t)p->cams[c].grid_w * p->cams[c].grid_h));
  {
    // elimination order (cba_solver_options.elimination)
    const bool eligible = !L.eliminate_points && !L.localize_only && !config->allreduce && L.n_images > 0;
    const int want = config->solver.elimination;
    if (want == 2 && !eligible) { set_error("cba_create: the grid-first elimination order needs eliminate_points = 0, localize_only = 0 and one rank"); return CBA_ERR_UNSUPPORTED; }
    if (want < 0 || want > 2 || config->solver.grid_strips < 0) { set_error("cba_create: bad solver options"); return CBA_ERR_ARG; }
    bool use = want == 2;
    if (want == 0 && eligible) {
      double pf = 0, gfl = 0;
      gf_flop_model(p->cams.data(), L.n_cameras, L.n_images, L.n_points, &pf, &gfl);
      int G = 0;
      for (int c = 0; c < L.n_cameras; ++c) G += (p->cams[c].model_type == CBA_CENTRAL_GENERIC ? 2 : 5) * p->cams[c].grid_w * p->cams[c].grid_h;
      use = G >= 2048 && gfl < 0.85 * pf;
    }
    if (use) {
      if (gf_build_plan(p->cams.data(), L.n_cameras, L.n_images, L.n_points, config->solver.grid_strips, config->solver.grid_single_tile_tasks, &p->gf) != CBA_OK) { set_error("cba_create: grid-first plan failed"); return CBA_ERR_ARG; }
      p->gridfirst = true;
    }
  }
  CBA_TRY(build_grid_order(p));
  CBA_TRY(dev_alloc(&p->cams_dev[0], L.n_cameras));
  CBA_TRY(dev_alloc(&p->cams_dev[1], L.n_cameras));
  CBA_TRY(upload_camdevs(p, 0));
  CBA_TRY(upload_camdevs(p, 1));
  // pair tables (upper-triangle (row,col) pairs) for the two column-count classes
  {
    const size_t stride = (size_t)kMaxCols * (kMaxCols + 1) / 2;
    std::vector<uint32_t> tab(2 * stride, 0);
    int counts[2] = {0, 0};
    for (int slot = 0; slot < 2; ++slot) {
      int Kg = L.localize_only ? 0 : (slot == 0 ? 32 : 80);
      int K = 6 + (L.rig_in_state ? 6 : 0) + 3 + Kg;
      // pairs with both columns in the pose/rig ("hot") range are accumulated in registers by the kernel
      const int nh = 6 + (L.rig_in_state ? 6 : 0), h0 = L.eliminate_points ? 3 : 0;
      int e = 0;
      for (int i = 0; i < K; ++i)
        for (int k = i; k < K; ++k) {
          const bool hot = i >= h0 && i < h0 + nh && k >= h0 && k < h0 + nh;
          const bool grid_grid = i >= K - Kg && k >= K - Kg;   // summed per grid cell by k_accumulate_cells
          // pose x (point | grid) goes through k_accumulate_strips when the poses are the Schur blocks
          const bool strip = !L.eliminate_points && i < 6 && k >= nh;
          // rig pose x grid is summed per grid cell by k_accumulate_cells as well (several cameras, poses eliminated)
          const bool rig_grid = !L.eliminate_points && L.rig_in_state && i >= 6 && i < 12 && k >= K - Kg;
          // every pair with a point column is summed per pattern point by k_accumulate_points (poses eliminated)
          const bool point = !L.eliminate_points && ((i >= nh && i < nh + 3) || (k >= nh && k < nh + 3));
          if (!hot && !grid_grid && !strip && !rig_grid && !point) tab[slot * stride + e++] = ((uint32_t)i << 16) | (uint32_t)k;
        }
      counts[slot] = e;
    }
    CBA_TRY(dev_alloc(&p->pair_tables, tab.size()));
    CBA_TRY(dev_alloc(&p->pair_counts, 2));
    CBA_HIP(hipMemcpy(p->pair_tables, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    CBA_HIP(hipMemcpy(p->pair_counts, counts, sizeof(counts), hipMemcpyHostToDevice));
  }
  {
    p->cell_base_host.assign(L.n_cameras + 1, 0);
    for (int c = 0; c < L.n_cameras; ++c) p->cell_base_host[c + 1] = p->cell_base_host[c] + p->cams[c].grid_w * p->cams[c].grid_h;
    const size_t nk = (size_t)p->cell_base_host.back();
    CBA_TRY(dev_alloc(&p->cell_base, p->cell_base_host.size()));
    CBA_HIP(hipMemcpy(p->cell_base, p->cell_base_host.data(), sizeof(int) * p->cell_base_host.size(), hipMemcpyHostToDevice));
    CBA_TRY(dev_alloc(&p->cell_count, nk + 1)); CBA_TRY(dev_alloc(&p->cell_start, nk + 1)); CBA_TRY(dev_alloc(&p->cell_fill, nk + 1));
  }
  CBA_TRY(dev_alloc(&p->det_bits, 2)); CBA_TRY(dev_alloc(&p->det_scale, 2));
  CBA_TRY(dev_alloc(&p->fd_redo_count, 4));
  CBA_HIP(hipMemset(p->fd_redo_count, 0, sizeof(int) * 4));
  CBA_TRY(dev_alloc(&p->red_partials, 256 * 8));
  CBA_TRY(dev_alloc(&p->red8, 16));
  // normal equations
  padded_dims(L.dense_dof, &p->n_pad, &p->n_fact);
  p->Kpad = round_up(L.block_dof > 0 ? L.block_dof : 1, 48);      // a multiple of the dense K slab (16) and of the block-sparse one (12)
  const size_t bs = L.block_size, nb = L.n_blocks;
  CBA_TRY(dev_alloc(&p->Dblk, nb * bs * bs));
  CBA_TRY(dev_alloc(&p->bblk, nb * bs));
  CBA_TRY(dev_alloc(&p->Dinv, nb * bs * bs));
  CBA_TRY(dev_alloc(&p->dinvb, (size_t)p->Kpad));
  CBA_TRY(dev_alloc(&p->B, (size_t)p->Kpad * p->n_pad));
  CBA_TRY(dev_alloc(&p->Hdd, (size_t)p->n_pad * p->n_pad));
  CBA_TRY(dev_alloc(&p->bd, (size_t)p->n_pad));
  if (p->gridfirst) {
    const GfPlan& g = p->gf;
    const size_t nf = (size_t)g.n_pad, wb = (size_t)(g.n_pad - g.Gf);
    CBA_TRY(dev_alloc(&p->F, nf * nf));
    CBA_HIP(hipMemset(p->F, 0, sizeof(double) * nf * nf));          // tiles outside the plan's structure stay zero for ever
    CBA_TRY(dev_alloc(&p->Xb, (size_t)g.Gf * wb));
    CBA_HIP(hipMemset(p->Xb, 0, sizeof(double) * (size_t)g.Gf * wb));
    CBA_TRY(dev_alloc(&p->xF, nf));
    CBA_TRY(dev_alloc(&p->gfd.tasks, g.tasks.size()));
    CBA_TRY(dev_alloc(&p->gfd.ivals, g.ivals.size()));
    CBA_TRY(dev_alloc(&p->gfd.chains, g.chains.size()));
    CBA_TRY(dev_alloc(&p->gfd.rowmask, g.rowmask.size()));
    CBA_HIP(hipMemcpy(p->gfd.tasks, g.tasks.data(), sizeof(GfTask) * g.tasks.size(), hipMemcpyHostToDevice));
    CBA_HIP(hipMemcpy(p->gfd.ivals, g.ivals.data(), sizeof(GfIval) * g.ivals.size(), hipMemcpyHostToDevice));
    CBA_HIP(hipMemcpy(p->gfd.chains, g.chains.data(), sizeof(GfChain) * g.chains.size(), hipMemcpyHostToDevice));
    CBA_HIP(hipMemcpy(p->gfd.rowmask, g.rowmask.data(), sizeof(uint64_t) * g.rowmask.size(), hipMemcpyHostToDevice));
    p->gfd.n_tasks0 = g.n_tasks0; p->gfd.n_tasks1 = (int)g.tasks.size() - g.n_tasks0; p->gfd.n_chains = (int)g.chains.size();
    p->gfd.nbg = g.nbg; p->gfd.nbf = g.nbf; p->gfd.mask_words = g.mask_words; p->gfd.flops_grid = g.flops_grid;
    // tiles the forming kernel writes per attempt: the structural tiles of the grid x grid part, the row strips of the grid rows
    // (every border column block + the right-hand side's), the upper triangle of the border
    // Rule: every tile that any launch of a solve WRITES is formed again for the next attempt (a broken solve -- zero pivot, NaN --
    // must not leave anything behind: tests/test_gpu_gridfirst.py).  The dense border launch and the border update also write the
    // padding-only block columns and the block rows behind the factored ones.
    std::vector<int> tiles(g.grid_tiles);
    for (int r = 0; r < g.nbg; ++r) {
      for (int c = g.nbg; c < g.nbf; ++c) { tiles.push_back(r); tiles.push_back(c); }
      tiles.push_back(r); tiles.push_back(g.ntc - 1);
    }
    for (int r = g.nbg; r < g.ntc; ++r)
      for (int c = r; c < g.ntc; ++c) { tiles.push_back(r); tiles.push_back(c); }
    p->n_gf_tiles = (int)(tiles.size() / 2);
    CBA_TRY(dev_alloc(&p->gf_tiles, tiles.size()));
    CBA_HIP(hipMemcpy(p->gf_tiles, tiles.data(), sizeof(int) * tiles.size(), hipMemcpyHostToDevice));
    CBA_TRY(dev_alloc(&p->gf_grid_of_f, g.grid_of_f.size()));
    CBA_HIP(hipMemcpy(p->gf_grid_of_f, g.grid_of_f.data(), sizeof(int) * g.grid_of_f.size(), hipMemcpyHostToDevice));
    CBA_TRY(dev_alloc(&p->gf_f_of_grid, g.f_of_grid.size()));
    CBA_HIP(hipMemcpy(p->gf_f_of_grid, g.f_of_grid.data(), sizeof(int) * g.f_of_grid.size(), hipMemcpyHostToDevice));
    // activity of the row strips (per pass): bit sets over the grid block rows per 128-column border tile, and what is derived
    {
      GfDevice& d = p->gfd;
      d.act_words = (g.nbg + 63) / 64;
      d.n_act_tiles = (g.n_pad - g.Gf) / 128;
      d.kmask_words = (g.Gf / 16 + 63) / 64;
      CBA_TRY(dev_alloc(&d.act, (size_t)d.n_act_tiles * d.act_words));
      CBA_TRY(dev_alloc(&d.kmask, (size_t)(g.n_pad / 128) * d.kmask_words));
      CBA_HIP(hipMemset(d.kmask, 0, sizeof(unsigned long long) * (size_t)(g.n_pad / 128) * d.kmask_words));
      CBA_TRY(dev_alloc(&d.rowmask_dyn, g.rowmask.size()));
      CBA_TRY(dev_alloc(&d.gridrow, g.gridrow.size()));
      CBA_HIP(hipMemcpy(d.gridrow, g.gridrow.data(), sizeof(uint64_t) * g.gridrow.size(), hipMemcpyHostToDevice));
      {
        const size_t nt = (size_t)d.n_act_tiles;
        // entries of four ints (tm, tn, K-slab range), eight interleaved per-XCD lists padded to the longest: 4 x 8 x (tiles in up to three
        // parts each, dealt evenly) is a quarter of this
        p->gf_tile_list_capacity = (size_t)32 * nt * (nt + 1) + 4096;
        CBA_TRY(dev_alloc(&p->gf_tile_list, p->gf_tile_list_capacity));
        CBA_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->gf_tile_list_host), sizeof(int) * p->gf_tile_list_capacity));
      }
      CBA_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->gf_kmask_host), sizeof(unsigned long long) * (size_t)(g.n_pad / 128) * d.kmask_words));
      std::memset(p->gf_kmask_host, 0, sizeof(unsigned long long) * (size_t)(g.n_pad / 128) * d.kmask_words);
    }
  } else {
    CBA_TRY(dev_alloc(&p->W, (size_t)p->Kpad * p->n_pad));
    CBA_TRY(dev_alloc(&p->S, (size_t)p->n_pad * p->n_pad));
  }
  if (config->allreduce) {
    // the reduced system crosses ranks as its upper 128-row blocks only (half the all-reduce volume)
    int64_t need = packed_upper_doubles(p->n_pad);
    if (config->distributed_solve) {       // staging of the reduce-scatter / all-gathers (send: P, receive: P2)
      need = (int64_t)ldlt_dist_buffer_doubles(p->n_pad, config->world_size);
      p->dist_buf_doubles = (size_t)need;
      CBA_TRY(dev_alloc(&p->P2, (size_t)need));
    }
    if (config->reduce_buffer) {
      if (config->reduce_buffer_doubles < need) { set_error("reduce_buffer too small"); return CBA_ERR_ARG; }
      p->P = static_cast<double*>(config->reduce_buffer); p->P_owned = false;
    } else {
      CBA_TRY(dev_alloc(&p->P, (size_t)need));
    }
  }
  if (p->S) CBA_HIP(hipMemset(p->S, 0, sizeof(double) * (size_t)p->n_pad * p->n_pad));
  if (p->W) CBA_HIP(hipMemset(p->W, 0, sizeof(double) * (size_t)p->Kpad * p->n_pad));
  CBA_HIP(hipMemset(p->dinvb, 0, sizeof(double) * (size_t)p->Kpad));
  CBA_TRY(dev_alloc(&p->x, (size_t)L.block_dof + p->n_pad));
  CBA_HIP(hipMemset(p->x, 0, sizeof(double) * ((size_t)L.block_dof + p->n_pad)));
  CBA_TRY(dev_alloc(&p->scal, 16));
  CBA_TRY(dev_alloc(&p->kmask, (size_t)(p->n_pad / 128) * schur_mask_words(p->Kpad)));
  if (schur_chunk_count(p->n_pad) > 0) {
    CBA_TRY(dev_alloc(&p->chunk_order, (size_t)schur_chunk_count(p->n_pad)));
    CBA_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->chunk_order_host), sizeof(int) * (size_t)schur_chunk_count(p->n_pad)));
  }
  CBA_TRY(dev_alloc(&p->gemv_ws, (size_t)gemv_t_workspace_doubles(p->n_pad)));
  CBA_TRY(dev_alloc(&p->status, 2));      // [0] block-inverse status, [1] guard word of the solve (k_solve_status)
  if (p->gridfirst) CBA_TRY(ldlt_workspace_alloc(p->ldlt, p->gf.n_pad, p->gf.nbg));
  else CBA_TRY(ldlt_workspace_alloc(p->ldlt, p->n_pad));
  apply_solver_options(p->ldlt, &config->solver);
  guard.q = nullptr;
  *out = p;
  return CBA_OK;
}

void cba_destroy(cba_problem* p) {
  if (!p) return;
  hipSetDevice(p->device);
  if (p->stream) hipStreamSynchronize(p->stream);
  auto F = [](void* q) { if (q) hipFree(q); };
  F(p->obs_xy); F(p->obs_point); F(p->obs_image); F(p->obs_camera); F(p->last_projection);
  for (int s = 0; s < 2; ++s) {
    F(p->st[s].rig_tr_global); F(p->st[s].camera_tr_rig); F(p->st[s].points);
    for (int c = 0; c < kMaxCameras; ++c) F(p->st[s].grids[c]);
    F(p->cams_dev[s]);
  }
  F(p->itg);
  for (int c = 0; c < kMaxCameras; ++c) { F(p->tangents[c]); F(p->gperm[c]); }
  F(p->cost_ref); F(p->cost_test); F(p->pixels); F(p->flags); F(p->fd_out); F(p->fd_ok); F(p->jrec); F(p->cells);
  F(p->pair_tables); F(p->pair_counts); F(p->pt_start); F(p->pt_obs); F(p->red_partials); F(p->red8);
  F(p->cell_base); F(p->cell_count); F(p->cell_start); F(p->cell_fill); F(p->cell_order); F(p->pose_slot);
  F(p->Dblk); F(p->bblk); F(p->B); F(p->Hdd); F(p->bd); F(p->Dinv); F(p->dinvb); F(p->W);
  if (p->S_owned) F(p->S);
  if (p->P_owned) F(p->P);
  F(p->P2);
  F(p->x); F(p->scal); F(p->status); F(p->gemv_ws); F(p->kmask);
  F(p->F); F(p->Xb); F(p->xF); F(p->gf_tiles); F(p->gf_grid_of_f); F(p->gf_f_of_grid);
  F(p->gfd.tasks); F(p->gfd.ivals); F(p->gfd.chains); F(p->gfd.rowmask);
  F(p->gfd.act); F(p->gfd.gridrow); F(p->gfd.kmask); F(p->gfd.rowmask_dyn);
  if (p->gf_kmask_host) hipHostFree(p->gf_kmask_host);
  F(p->gf_tile_list);
  if (p->gf_tile_list_host) hipHostFree(p->gf_tile_list_host);
  ldlt_workspace_free(p->ldlt);
  for (auto& t : p->timers) for (auto& sp : t.spans) { hipEventDestroy(sp.e0); hipEventDestroy(sp.e1); }
  if (p->kmask_host) hipHostFree(p->kmask_host);
  if (p->chunk_order_host) hipHostFree(p->chunk_order_host);
  if (p->chunk_order) hipFree(p->chunk_order);
  F(p->slow_skip); F(p->fd_slow); F(p->slow_list); F(p->slow_count); F(p->img_start); F(p->band_mask); F(p->det_bits); F(p->det_scale); F(p->fd_redo[0]); F(p->fd_redo[1]); F(p->fd_redo_count);
  if (p->ev_aux0) hipEventDestroy(p->ev_aux0);
  if (p->ev_aux1) hipEventDestroy(p->ev_aux1);
  if (p->ev_aux2) hipEventDestroy(p->ev_aux2);
  if (p->ev_clear) hipEventDestroy(p->ev_clear);
  if (p->ev_mask) hipEventDestroy(p->ev_mask);
  if (p->pin_status) hipHostFree(p->pin_status);
  if (p->pin_cost) hipHostFree(p->pin_cost);
  delete p;
}

int cba_set_observations(cba_problem* p, int64_t n, const float* xy, const int32_t* point_index,
                         const int32_t* image_index, const int32_t* camera_index, const double* last_projection) {
  if (!p || n < 0 || (n > 0 && (!xy || !point_index || !image_index || !camera_index))) { set_error("cba_set_observations: bad argument"); return CBA_ERR_ARG; }
  const Layout& L = p->L;
  int64_t prev = -1;
  for (int64_t i = 0; i < n; ++i) {
    if (point_index[i] < 0 || point_index[i] >= L.n_points || image_index[i] < 0 || image_index[i] >= L.n_images ||
        camera_index[i] < 0 || camera_index[i] >= L.n_cameras) { set_error("cba_set_observations: index out of range"); return CBA_ERR_ARG; }
    int64_t key = (int64_t)image_index[i] * L.n_cameras + camera_index[i];
    if (key < prev) { set_error("cba_set_observations: observations must be sorted image-major, then camera"); return CBA_ERR_ARG; }
    prev = key;
  }
  CBA_HIP(hipSetDevice(p->device));
  auto F = [](void* q) { if (q) hipFree(q); };
  F(p->obs_xy); F(p->obs_point); F(p->obs_image); F(p->obs_camera); F(p->last_projection);
  F(p->cost_ref); F(p->cost_test); F(p->pixels); F(p->flags); F(p->fd_out); F(p->fd_ok); F(p->jrec); F(p->cells); F(p->cell_order);
  p->n_obs = n;
  CBA_TRY(dev_alloc(&p->obs_xy, 2 * (size_t)n)); CBA_TRY(dev_alloc(&p->obs_point, (size_t)n));
  CBA_TRY(dev_alloc(&p->obs_image, (size_t)n)); CBA_TRY(dev_alloc(&p->obs_camera, (size_t)n));
  CBA_TRY(dev_alloc(&p->last_projection, 2 * (size_t)n));
  CBA_TRY(dev_alloc(&p->cost_ref, (size_t)n)); CBA_TRY(dev_alloc(&p->cost_test, (size_t)n));
  CBA_TRY(dev_alloc(&p->pixels, 2 * (size_t)n)); CBA_TRY(dev_alloc(&p->flags, (size_t)n));
  CBA_TRY(dev_alloc(&p->fd_out, 2 * (size_t)n * p->tasks_per_obs)); CBA_TRY(dev_alloc(&p->fd_ok, (size_t)n * p->tasks_per_obs));
  CBA_TRY(dev_alloc(&p->jrec, (size_t)n * p->rec_doubles)); CBA_TRY(dev_alloc(&p->cells, 2 * (size_t)n));
  {
    // gather-path follow-up lists of the finite-difference kernel: a quarter of all tasks (the share of tasks whose iterate
    // crosses a cell boundary is a few per cent) + 65 536
    const size_t cap = (size_t)n * p->tasks_per_obs / 4 + 65536;
    p->fd_redo_cap = (int)(cap > 0x7fffff00u ? 0x7fffff00u : cap);
    for (int i = 0; i < 2; ++i) { if (p->fd_redo[i]) hipFree(p->fd_redo[i]); p->fd_redo[i] = nullptr; CBA_TRY(dev_alloc(&p->fd_redo[i], (size_t)p->fd_redo_cap)); }
  }
  CBA_HIP(hipMemset(p->jrec, 0, sizeof(double) * (size_t)(n > 0 ? n : 1) * p->rec_doubles));   // records of mixed-model problems have unused tails
  CBA_TRY(dev_alloc(&p->cell_order, (size_t)n));
  F(p->img_start); p->img_start = nullptr;
  F(p->band_mask); p->band_mask = nullptr;
  CBA_TRY(dev_alloc(&p->band_mask, (size_t)(n > 0 ? n : 1)));
  {
    std::vector<int64_t> is((size_t)L.n_images + 1, 0);
    for (int64_t i = 0; i < n; ++i) is[image_index[i] + 1] += 1;
    for (int i = 0; i < L.n_images; ++i) is[i + 1] += is[i];
    CBA_TRY(dev_alloc(&p->img_start, is.size()));
    CBA_HIP(hipMemcpy(p->img_start, is.data(), sizeof(int64_t) * is.size(), hipMemcpyHostToDevice));
  }
  // observations of each (camera, pattern point), for the per-point accumulation (static: the point of an observation is data)
  F(p->pt_start); p->pt_start = nullptr; F(p->pt_obs); p->pt_obs = nullptr;
  if (!L.eliminate_points && n > 0 && n < 0x7fffffff && (int64_t)L.n_cameras * L.n_points < 0x7fffffff) {
    const size_t nk = (size_t)L.n_cameras * L.n_points;
    std::vector<int> ks(nk + 1, 0), ko((size_t)n);
    for (int64_t i = 0; i < n; ++i) ks[(size_t)camera_index[i] * L.n_points + point_index[i] + 1] += 1;
    for (size_t k = 0; k < nk; ++k) ks[k + 1] += ks[k];
    std::vector<int> fill(ks.begin(), ks.end() - 1);
    for (int64_t i = 0; i < n; ++i) ko[(size_t)fill[(size_t)camera_index[i] * L.n_points + point_index[i]]++] = (int)i;
    CBA_TRY(dev_alloc(&p->pt_start, ks.size())); CBA_TRY(dev_alloc(&p->pt_obs, ko.size()));
    CBA_HIP(hipMemcpy(p->pt_start, ks.data(), sizeof(int) * ks.size(), hipMemcpyHostToDevice));
    CBA_HIP(hipMemcpy(p->pt_obs, ko.data(), sizeof(int) * ko.size(), hipMemcpyHostToDevice));
  }
  F(p->slow_list); p->slow_list = nullptr;
  p->slow_cap = (int)std::min<int64_t>(std::max<int64_t>(kSlowCapMin, n / 8), 1 << 24);
  CBA_TRY(dev_alloc(&p->slow_list, (size_t)p->slow_cap));
  F(p->slow_skip); p->slow_skip = nullptr;
  CBA_TRY(dev_alloc(&p->slow_skip, (size_t)(n > 0 ? n : 1)));
  CBA_HIP(hipMemset(p->slow_skip, 0, (size_t)(n > 0 ? n : 1)));
  F(p->fd_slow); p->fd_slow = nullptr;
  CBA_TRY(dev_alloc(&p->fd_slow, (size_t)(n > 0 ? n : 1)));
  CBA_HIP(hipMemset(p->fd_slow, 0, (size_t)(n > 0 ? n : 1)));
  CBA_HIP(hipMemset(p->slow_count, 0, sizeof(int)));
  if (n > 0) {
    CBA_HIP(hipMemcpy(p->obs_xy, xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice));
    CBA_HIP(hipMemcpy(p->obs_point, point_index, sizeof(int) * n, hipMemcpyHostToDevice));
    CBA_HIP(hipMemcpy(p->obs_image, image_index, sizeof(int) * n, hipMemcpyHostToDevice));
    CBA_HIP(hipMemcpy(p->obs_camera, camera_index, sizeof(int) * n, hipMemcpyHostToDevice));
    if (last_projection) CBA_HIP(hipMemcpy(p->last_projection, last_projection, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    else CBA_HIP(hipMemset(p->last_projection, 0, sizeof(double) * 2 * n));
    CBA_HIP(hipMemset(p->flags, 0, (size_t)n));
  }
  // block order of the imagesets (eliminate_points = 0 only: the pose blocks are the Schur blocks)
  if (p->pose_slot) { hipFree(p->pose_slot); p->pose_slot = nullptr; }
  p->pose_slot_host.clear();
  if (!L.eliminate_points && L.n_images > 0 && n > 0) {
    // sort key: Z-order (Morton) index of the centre of the imageset's observations on an 8x8 raster of
    // the image, then the vertical centre -- neighbours in the order have overlapping 2-D footprints
    std::vector<double> sx(L.n_images, 0.0), sy(L.n_images, 0.0); std::vector<int> cnt(L.n_images, 0);
    double max_x = 1.0, max_y = 1.0;
    for (int64_t i = 0; i < n; ++i) {
      sx[image_index[i]] += xy[2 * i]; sy[image_index[i]] += xy[2 * i + 1]; cnt[image_index[i]] += 1;
      max_x = std::max(max_x, (double)xy[2 * i]); max_y = std::max(max_y, (double)xy[2 * i + 1]);
    }
    std::vector<int> order(L.n_images), key(L.n_images, 0);
    for (int i = 0; i < L.n_images; ++i) {
      order[i] = i;
      sx[i] = cnt[i] ? sx[i] / cnt[i] : 0.0; sy[i] = cnt[i] ? sy[i] / cnt[i] : 0.0;
      const int qx = std::min(7, (int)(8.0 * sx[i] / (max_x + 1.0))), qy = std::min(7, (int)(8.0 * sy[i] / (max_y + 1.0)));
      int k = 0;
      for (int b = 0; b < 3; ++b) k |= (((qx >> b) & 1) << (2 * b)) | (((qy >> b) & 1) << (2 * b + 1));
      key[i] = k;
    }
    std::stable_sort(order.begin(), order.end(), [&](int u, int v) { return key[u] != key[v] ? key[u] < key[v] : sy[u] < sy[v]; });
    // Round 5: refine that order into a nearest-neighbour chain on the imagesets' FOOTPRINTS in the Schur product's own units.  What the
    // block-sparse K loop of the product executes is, per pair of 128-column tiles, the 16-row slabs (2.7 imagesets) whose rows are
    // non-zero in both tiles -- so the cost of an order is how much the tile sets of neighbouring imagesets differ, and the Z-order of the
    // footprint CENTRES only approximates that (footprints differ in size and shape).  Tile set of an imageset = the tiles of its points'
    // columns and of the 4 x 4 control patches under its measured pixels (the engine's tiled grid order, build_grid_order); chain: start
    // at the head of the Z-order, always append the unplaced imageset whose tile set has the smallest Hamming distance to the last one.
    // Executed slabs of the product, modelled from the observation lists: x 0.86 (cfg 2), 0.85 (cfg 4), 0.83 (cfg 3) against the
    // Z-order; measured: profiles/r05_schur_row_order.txt.  The order is internal (x and the dumps are un-permuted); the sum over the
    // pose blocks is taken in another order, which moves S by rounding only.  O(N^2 T / 64): skipped above 8192 imagesets.
    if (p->gridfirst) {
      // Grid-first order: the pose columns of F are empty in the grid block rows the imageset does not reach (per-pass activity,
      // k_gf_touch), per 128-column tile = ~21 imagesets.  Imagesets are ordered by the first row of F their control patches touch
      // (under the MEASURED pixels: a heuristic, the activity itself comes from the projected ones), so that the imagesets of a tile
      // start at about the same place of the elimination order and their union stays small.
      const GfPlan& g = p->gf;
      const int W = g.grid_words;
      std::vector<uint64_t> touched((size_t)L.n_images * W, 0ull);
      for (int64_t i = 0; i < n; ++i) {
        const int img = image_index[i], cam = camera_index[i];
        const cba_camera& cm = p->cams[cam];
        if (!std::isfinite(xy[2 * i]) || !std::isfinite(xy[2 * i + 1])) continue;
        const int per = cm.model_type == CBA_CENTRAL_GENERIC ? 2 : 5;
        const double gx = 1.0 + (cm.grid_w - 3.0) * (xy[2 * i] - cm.calib_min_x) / (cm.calib_max_x + 1.0 - cm.calib_min_x);      // central_grid.h:150-154
        const double gy = 1.0 + (cm.grid_h - 3.0) * (xy[2 * i + 1] - cm.calib_min_y) / (cm.calib_max_y + 1.0 - cm.calib_min_y);
        const int fx = (int)std::floor(gx + 2) - 3, fy = (int)std::floor(gy + 2) - 3;
        for (int r = 0; r < 4; ++r)
          for (int q = 0; q < 4; ++q) {
            const int cx = fx + q, cy = fy + r;
            if (cx < 0 || cy < 0 || cx >= cm.grid_w || cy >= cm.grid_h) continue;
            const int e = L.intr_offset[cam] - g.n_rp + per * g.gperm[cam][cx + (size_t)cy * cm.grid_w];
            const int r0 = g.f_of_grid[e] >> 6, r1 = g.f_of_grid[e + per - 1] >> 6;
            touched[(size_t)img * W + (r0 >> 6)] |= 1ull << (r0 & 63);
            touched[(size_t)img * W + (r1 >> 6)] |= 1ull << (r1 & 63);
          }
      }
      std::vector<int> slot_of;
      gf_order_imagesets(g, touched, L.n_images, g.n_rp, &slot_of);
      for (int i = 0; i < L.n_images; ++i) order[slot_of[i]] = i;
      // First tile order of the border update, predicted from the same rows (the first solve would otherwise run its tiles in
      // row-major order: 4.7 instead of 3.5 ms at BASELINE configs[1]): per 128-column tile of the border the union of its imagesets'
      // rows, closed under the fill of the grid factor; rig / point tiles and the right-hand side's tile reach every row.
      if (p->gf_tile_list_host) {
        const int nt = (g.n_pad - g.Gf) / 128;
        std::vector<uint64_t> tact((size_t)nt * W, 0ull);
        auto all_rows = [&](int t) { for (int w = 0; w < W; ++w) tact[(size_t)t * W + w] = ~0ull; };
        for (int t = 0; t < nt && 128 * t < g.n_rp; ++t) all_rows(t);
        all_rows(nt - 1);
        for (int i = 0; i < L.n_images; ++i)
          for (int t : {(g.n_rp + 6 * slot_of[i]) >> 7, (g.n_rp + 6 * slot_of[i] + 5) >> 7})
            for (int w = 0; w < W; ++w) tact[(size_t)t * W + w] |= touched[(size_t)i * W + w];
        for (int t = 0; t < nt; ++t) {
          uint64_t* a = &tact[(size_t)t * W];
          for (int r = 0; r < g.nbg; ++r)
            if ((a[r >> 6] >> (r & 63)) & 1ull)
              for (int w = r >> 6; w < W; ++w) a[w] |= g.gridrow[(size_t)r * W + w];
          for (int w = 0; w < W; ++w)
            if (64 * w + 64 > g.nbg) a[w] &= (64 * w >= g.nbg) ? 0ull : (~0ull >> (64 - (g.nbg - 64 * w)));
        }
        gf_build_tile_list(p, nt, g.Gf / 16, [&](int tm, int tn) {
          int rows = 0;
          for (int w = 0; w < W; ++w) rows += __builtin_popcountll(tact[(size_t)tm * W + w] & tact[(size_t)tn * W + w]);
          return 4 * rows;                                   // K slabs of 16 rows: four per block row
        }, [&](int tm, int tn, int target) {
          int seen = 0;
          for (int r = 0; r < g.nbg; ++r)
            if (((tact[(size_t)tm * W + (r >> 6)] & tact[(size_t)tn * W + (r >> 6)]) >> (r & 63)) & 1ull) {
              if (seen + 4 > target) return 4 * r;
              seen += 4;
            }
          return 0;
        });
        p->gf_tile_list_age = 0;
      }
    } else
    if (!L.localize_only && L.n_images >= 4 && L.n_images <= 8192 && !p->dense_perm_host.empty()) {
      const int T = (L.dense_dof + 127) / 128, W = (T + 63) / 64;
      std::vector<unsigned long long> mask((size_t)L.n_images * W, 0ull);
      auto set_col = [&](int img, int col) { if (col >= 0 && col < L.dense_dof) { const int t = col >> 7; mask[(size_t)img * W + (t >> 6)] |= 1ull << (t & 63); } };
      for (int64_t i = 0; i < n; ++i) {
        const int img = image_index[i], cam = camera_index[i];
        const int pc = L.first_points - L.block_dof + 3 * point_index[i];
        set_col(img, pc); set_col(img, pc + 2);
        if (!std::isfinite(xy[2 * i]) || !std::isfinite(xy[2 * i + 1])) continue;      // (the cast below is undefined for a non-finite pixel)
        const cba_camera& cm = p->cams[cam];
        const int per = cm.model_type == CBA_CENTRAL_GENERIC ? 2 : 5;
        const double gx = 1.0 + (cm.grid_w - 3.0) * (xy[2 * i] - cm.calib_min_x) / (cm.calib_max_x + 1.0 - cm.calib_min_x);      // central_grid.h:150-154
        const double gy = 1.0 + (cm.grid_h - 3.0) * (xy[2 * i + 1] - cm.calib_min_y) / (cm.calib_max_y + 1.0 - cm.calib_min_y);
        const int fx = (int)std::floor(gx + 2) - 3, fy = (int)std::floor(gy + 2) - 3;
        for (int r = 0; r < 4; ++r)
          for (int q = 0; q < 4; ++q) {
            const int cx = fx + q, cy = fy + r;
            if (cx < 0 || cy < 0 || cx >= cm.grid_w || cy >= cm.grid_h) continue;
            const int first = L.intr_offset[cam] + per * (cx + cy * cm.grid_w);
            set_col(img, p->dense_perm_host[first]); set_col(img, p->dense_perm_host[first + per - 1]);
          }
      }
      std::vector<int> chain; chain.reserve(L.n_images);
      std::vector<char> placed(L.n_images, 0);
      int cur = order[0];
      chain.push_back(cur); placed[cur] = 1;
      for (int step = 1; step < L.n_images; ++step) {
        const unsigned long long* mc = &mask[(size_t)cur * W];
        int best = -1, best_d = 0x7fffffff;
        for (int r = 0; r < L.n_images; ++r) {                 // candidates in Z-order: ties go to the Z-order neighbour
          const int v = order[r];
          if (placed[v]) continue;
          const unsigned long long* mv = &mask[(size_t)v * W];
          int d = 0;
          for (int w = 0; w < W; ++w) d += __builtin_popcountll(mc[w] ^ mv[w]);
          if (d < best_d) { best_d = d; best = v; }
        }
        cur = best; chain.push_back(cur); placed[cur] = 1;
      }
      order.swap(chain);
    }
    p->pose_slot_host.assign(L.n_images, 0);
    for (int r = 0; r < L.n_images; ++r) p->pose_slot_host[order[r]] = r;
    CBA_TRY(dev_alloc(&p->pose_slot, (size_t)L.n_images));
    CBA_HIP(hipMemcpy(p->pose_slot, p->pose_slot_host.data(), sizeof(int) * L.n_images, hipMemcpyHostToDevice));
  }
  p->have_obs = true; p->have_system = false;
  return CBA_OK;
}

int cba_set_state(cba_problem* p, const double* rig_tr_global, const double* camera_tr_rig, const double* points,
                  const double* const* grids) {
  if (!p || !camera_tr_rig || !grids || (p->L.n_images > 0 && !rig_tr_global) || (p->L.n_points > 0 && !points)) { set_error("cba_set_state: bad argument"); return CBA_ERR_ARG; }
  CBA_HIP(hipSetDevice(p->device));
  DevState& s = p->st[p->cur];
  const Layout& L = p->L;
  if (L.n_images) CBA_HIP(hipMemcpy(s.rig_tr_global, rig_tr_global, sizeof(double) * 7 * L.n_images, hipMemcpyHostToDevice));
  CBA_HIP(hipMemcpy(s.camera_tr_rig, camera_tr_rig, sizeof(double) * 7 * L.n_cameras, hipMemcpyHostToDevice));
  if (L.n_points) CBA_HIP(hipMemcpy(s.points, points, sizeof(double) * 3 * L.n_points, hipMemcpyHostToDevice));
  for (int c = 0; c < L.n_cameras; ++c) {
    if (!grids[c]) { set_error("cba_set_state: null grid"); return CBA_ERR_ARG; }
    size_t G = (size_t)p->cams[c].grid_w * p->cams[c].grid_h;
    CBA_HIP(hipMemcpy(s.grids[c], grids[c], sizeof(double) * (p->cams[c].model_type == CBA_CENTRAL_GENERIC ? 3 : 6) * G, hipMemcpyHostToDevice));
  }
  p->have_state = true; p->have_system = false;
  return CBA_OK;
}

int cba_get_state(cba_problem* p, double* rig_tr_global, double* camera_tr_rig, double* points, double* const* grids) {
  if (!p || !p->have_state) { set_error("cba_get_state: no state"); return CBA_ERR_STATE; }
  CBA_HIP(hipSetDevice(p->device));
  CBA_HIP(hipStreamSynchronize(p->stream));
  const DevState& s = p->st[p->cur];
  const Layout& L = p->L;
  if (rig_tr_global && L.n_images) CBA_HIP(hipMemcpy(rig_tr_global, s.rig_tr_global, sizeof(double) * 7 * L.n_images, hipMemcpyDeviceToHost));
  if (camera_tr_rig) CBA_HIP(hipMemcpy(camera_tr_rig, s.camera_tr_rig, sizeof(double) * 7 * L.n_cameras, hipMemcpyDeviceToHost));
  if (points && L.n_points) CBA_HIP(hipMemcpy(points, s.points, sizeof(double) * 3 * L.n_points, hipMemcpyDeviceToHost));
  if (grids)
    for (int c = 0; c < L.n_cameras; ++c) {
      if (!grids[c]) continue;
      size_t G = (size_t)p->cams[c].grid_w * p->cams[c].grid_h;
      CBA_HIP(hipMemcpy(grids[c], s.grids[c], sizeof(double) * (p->cams[c].model_type == CBA_CENTRAL_GENERIC ? 3 : 6) * G, hipMemcpyDeviceToHost));
    }
  return CBA_OK;
}

int cba_get_last_projection(cba_problem* p, double* out) {
  if (!p || !out || !p->have_obs) { set_error("cba_get_last_projection: bad argument"); return CBA_ERR_ARG; }
  CBA_HIP(hipSetDevice(p->device));
  CBA_HIP(hipStreamSynchronize(p->stream));
  if (p->n_obs) CBA_HIP(hipMemcpy(out, p->last_projection, sizeof(double) * 2 * p->n_obs, hipMemcpyDeviceToHost));
  return CBA_OK;
}

int cba_cost(cba_problem* p, double* cost, int64_t* n_valid, double* cost_vector) {
  if (!p || !p->have_obs || !p->have_state) { set_error("cba_cost: observations/state missing"); return CBA_ERR_STATE; }
  CBA_HIP(hipSetDevice(p->device));
  // (the device-side camera descriptions hold pointers and constants only: uploaded once, by cba_create)
  CBA_TRY(residual_pass(p, p->cur, p->cost_test));
  CBA_TRY(launch_reduce_costs(nullptr, p->cost_test, nullptr, p->n_obs, p->red_partials, p->red8, p->stream));
  CBA_TRY(allreduce(p, p->red8, 8));
  double h[8];
  CBA_TRY(read_scalars(p, p->red8, h, 8));
  if (cost) *cost = h[1];
  if (n_valid) *n_valid = (int64_t)h[6];
  if (cost_vector && p->n_obs) CBA_HIP(hipMemcpy(cost_vector, p->cost_test, sizeof(double) * p->n_obs, hipMemcpyDeviceToHost));
  return CBA_OK;
}

int cba_set_fd_schedule(cba_problem* p, int32_t schedule) {
  if (!p || schedule < -1 || schedule > 1) { set_error("cba_set_fd_schedule: bad argument"); return CBA_ERR_ARG; }
  p->fd_schedule = schedule;
  return CBA_OK;
}
int cba_set_straggler_threshold(cba_problem* p, int32_t outer_iterations) {
  if (!p || outer_iterations < 0) { set_error("cba_set_straggler_threshold: bad argument"); return CBA_ERR_ARG; }
  p->straggler_threshold = outer_iterations > 100 ? 100 : outer_iterations;
  return CBA_OK;
}

int cba_debug_accumulate(cba_problem* p, double* cost) {
  if (!p || !p->have_obs || !p->have_state) { set_error("cba_debug_accumulate: observations/state missing"); return CBA_ERR_STATE; }
  CBA_HIP(hipSetDevice(p->device));
  // (the device-side camera descriptions hold pointers and constants only: uploaded once, by cba_create)
  CBA_TRY(jacobian_pass_and_accumulate(p, nullptr));
  CBA_TRY(launch_reduce_costs(p->cost_ref, nullptr, p->flags, p->n_obs, p->red_partials, p->red8, p->stream));
  double h[8];
  CBA_TRY(read_scalars(p, p->red8, h, 8));
  if (cost) *cost = h[0];
  return CBA_OK;
}

int cba_debug_solve(cba_problem* p, double lambda) {
  if (!p || !p->have_system) { set_error("cba_debug_solve: no accumulated system"); return CBA_ERR_STATE; }
  CBA_HIP(hipSetDevice(p->device));
  return solve_system(p, lambda, nullptr);
}

int cba_debug_apply_update(cba_problem* p, const double* x) {
  if (!p || !x || !p->have_state) { set_error("cba_debug_apply_update: bad argument"); return CBA_ERR_STATE; }
  CBA_HIP(hipSetDevice(p->device));
  {
    std::vector<double> xp(x, x + p->L.total_dof);
    if (!p->pose_slot_host.empty())
      for (int i = 0; i < p->L.n_images; ++i)
        for (int k = 0; k < 6; ++k) xp[p->L.first_rig_tr_global + 6 * p->pose_slot_host[i] + k] = x[p->L.first_rig_tr_global + 6 * i + k];
    for (int i = 0; i < p->L.dense_dof; ++i) xp[p->L.block_dof + p->dense_perm_host[i]] = x[p->L.block_dof + i];
    CBA_HIP(hipMemcpy(p->x, xp.data(), sizeof(double) * p->L.total_dof, hipMemcpyHostToDevice));
  }
  CBA_TRY(launch_apply_update(p->L, p->cams, p->st[p->cur], p->x, p->st[p->cur ^ 1], p->pose_slot, p->gperm, p->stream));
  CBA_HIP(hipStreamSynchronize(p->stream));
  p->cur ^= 1;
  p->have_system = false;
  return CBA_OK;
}

int cba_step(cba_problem* p, double init_lambda, int32_t max_lm_attempts, double init_lambda_factor, cba_report* report) {
  if (!p || !report || max_lm_attempts < 1 || init_lambda_factor < 0) { set_error("cba_step: bad argument"); return CBA_ERR_ARG; }
  if (!p->have_obs || !p->have_state) { set_error("cba_step: observations/state missing"); return CBA_ERR_STATE; }
  CBA_HIP(hipSetDevice(p->device));
  std::memset(report, 0, sizeof(*report));
  CBA_TRY(timers_collect(p));
  for (auto& t : p->timers) { t.seconds = t.flops = t.bytes = 0; t.launches = 0; }
  const Layout& L = p->L;
  const bool multi = p->cfg.allreduce != nullptr;
  double h[8];
  // ---- residual + Jacobian pass, accumulation (lm_optimizer.h:706-720) ----
  double t0 = now_s();
  // (the device-side camera descriptions hold pointers and constants only: uploaded once, by cba_create)
  CBA_TRY(timer_begin(p, 5));
  CBA_TRY(jacobian_pass_and_accumulate(p, &report->t_accumulate));
  CBA_TRY(launch_reduce_costs(p->cost_ref, nullptr, p->flags, p->n_obs, p->red_partials, p->red8, p->stream));
  CBA_TRY(timer_end(p, 5, 0, 0, 1));
  // One GPU and a given lambda: nothing the host does before the first solve depends on the pass's scalars, so their read rides
  // on the solve's own wait (one host wait per LM attempt fewer: the device does not idle between the pass and the solve).  The
  // "cost is already zero" exit of lm_optimizer.h:755-760 is then taken after that solve, whose result is discarded.
  const bool defer_cost_read = !multi && init_lambda >= 0;
  double last_cost = 0;
  double lambda = p->last_lambda;
  auto take_pass_scalars = [&]() {
    last_cost = h[0];
    report->initial_cost = last_cost;
    report->n_residuals_valid = (int64_t)h[5];
    report->n_jacobians_dropped = (int64_t)h[7];
    report->final_cost = last_cost;
  };
  if (defer_cost_read) {
    if (!p->pin_cost) CBA_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->pin_cost), 16 * sizeof(double)));
    CBA_HIP(hipMemcpyAsync(p->pin_cost, p->red8, 8 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  } else {
    CBA_TRY(allreduce(p, p->red8, 8));
    CBA_TRY(read_scalars(p, p->red8, h, 8));
    take_pass_scalars();
    if (last_cost == 0) {                                             // lm_optimizer.h:755-760
      report->lambda = lambda;
      CBA_TRY(timers_collect(p));
      report->t_jac = p->timers[5].seconds;
      return CBA_OK;
    }
  }
  (void)t0;
  if (init_lambda >= 0) {
    lambda = init_lambda;
  } else {  // lm_optimizer.h:766-781
    CBA_TRY(launch_diag_sum(p->Dblk, L.block_size, L.n_blocks, p->Hdd, p->n_pad, L.dense_dof, p->scal, p->stream));
    CBA_TRY(allreduce(p, p->scal, 1));
    double sum;
    CBA_TRY(read_scalars(p, p->scal, &sum, 1));
    const int n_img_global = (multi && p->cfg.n_images_global > 0) ? p->cfg.n_images_global : L.n_images;
    const double dof_global = (double)L.total_dof + 6.0 * (n_img_global - L.n_images);
    lambda = init_lambda_factor * sum / dof_global;
  }
  // ---- LM attempts (lm_optimizer.h:802-965) ----
  // One GPU: the attempt's state update, cost-only pass and cost reduction are queued BEHIND the solve before the host looks at the
  // solve's status -- one host wait per attempt instead of two (the device no longer idles for a host round trip between the back
  // substitution and the cost pass).  The kernels of that pass that write the warm-start cache read the solve's guard word and do
  // nothing behind a broken solve (PassArgs::guard), so a NaN / zero-pivot attempt leaves no trace, as in the reference, which
  // skips the cost pass for a NaN update (lm_optimizer.h:905-913).  Decisions are unchanged: the same numbers reach the same tests.
  const bool fused = !multi;
  if (fused && !p->pin_cost) CBA_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->pin_cost), 16 * sizeof(double)));
  for (int lm = 0; lm < max_lm_attempts; ++lm) {
    report->lm_attempts += 1;
    t0 = now_s();
    const int cand = p->cur ^ 1;
    int rc;
    if (fused) {
      rc = solve_enqueue(p, lambda, report);
      if (rc == CBA_OK) {
        CBA_TRY(timer_begin(p, 7));
        CBA_TRY(launch_apply_update(L, p->cams, p->st[p->cur], p->x, p->st[cand], p->pose_slot, p->gperm, p->stream));
        CBA_TRY(residual_pass(p, cand, p->cost_test, p->status + 1));
        CBA_TRY(launch_reduce_costs(p->cost_ref, p->cost_test, nullptr, p->n_obs, p->red_partials, p->red8, p->stream));
        CBA_HIP(hipMemcpyAsync(p->pin_cost + 8, p->red8, 8 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        CBA_TRY(timer_end(p, 7, 0, 0, 1));
        rc = solve_finish(p);            // the attempt's one host wait
      }
    } else {
      rc = solve_system(p, lambda, report);
    }
    if (rc != CBA_OK && rc != CBA_ERR_NUMERIC) return rc;      // (before pin_cost is consumed: a solve that failed early never synchronised)
    if (defer_cost_read && lm == 0) {      // the solve has waited for the stream: the pass's scalars are in pinned memory
      CBA_HIP(hipStreamSynchronize(p->stream));                // (a no-op after a completed solve; a numeric failure may return before its wait)
      for (int i = 0; i < 8; ++i) h[i] = p->pin_cost[i];
      take_pass_scalars();
      if (last_cost == 0) {                // lm_optimizer.h:755-760 (the solve above is discarded; its queued cost pass has rewritten the warm-start
                                           // cache from the candidate state, which for a zero cost is the same pixels: x solves H x = 0 there)
        report->lm_attempts = 0;
        report->lambda = p->last_lambda;
        CBA_TRY(timers_collect(p));
        report->t_jac = p->timers[5].seconds;
        return CBA_OK;
      }
    }
    const double x0 = rc == CBA_OK ? p->last_x0 : NAN;
    bool failed = rc == CBA_ERR_NUMERIC || std::isnan(x0);
    if (multi) {
      // Image sharding: the pose-block inverses and x[0] are rank-local, the factorisation is replicated.  Every rank must
      // take the same accept / reject / NaN branch -- otherwise one rank would enter the next solve's all-reduce of the
      // packed system while the others wait in the 8-double cost all-reduce -- so the failure flag is summed over ranks.
      const double f = failed ? 1.0 : 0.0;
      CBA_HIP(hipMemcpyAsync(p->scal + 8, &f, sizeof(double), hipMemcpyHostToDevice, p->stream));
      CBA_TRY(allreduce(p, p->scal + 8, 1));
      double fs = 0.0;
      CBA_TRY(read_scalars(p, p->scal + 8, &fs, 1));
      failed = fs != 0.0;
    }
    if (failed) {   // NaN update -> lambda *= 2 (lm_optimizer.h:905-913)
      lambda = 2.f * lambda;
      continue;
    }
    if (fused) {
      for (int i = 0; i < 8; ++i) h[i] = p->pin_cost[8 + i];
    } else {
      t0 = now_s();
      CBA_TRY(launch_apply_update(L, p->cams, p->st[p->cur], p->x, p->st[cand], p->pose_slot, p->gperm, p->stream));
      CBA_TRY(residual_pass(p, cand, p->cost_test));
      CBA_TRY(launch_reduce_costs(p->cost_ref, p->cost_test, nullptr, p->n_obs, p->red_partials, p->red8, p->stream));
      CBA_TRY(allreduce(p, p->red8, 8));
      CBA_TRY(read_scalars(p, p->red8, h, 8));
      report->t_cost += now_s() - t0;
    }
    // CostIsSmallerThan (lm_optimizer.h:993-1011): only residuals valid in both passes
    const bool smaller = h[4] > 0 && h[3] < h[2];
    if (smaller) {
      p->cur = cand;
      lambda = 0.5f * lambda;
      report->accepted = 1;
      last_cost = h[1];
      break;
    } else {
      lambda = 2.f * lambda;
    }
  }
  report->final_cost = last_cost;
  report->lambda = lambda;
  p->last_lambda = lambda;
  p->have_system = true;
  // stage times from the device-side spans (HIP events on the streams the kernels ran on)
  CBA_TRY(timers_collect(p));
  report->t_gemm = p->timers[0].seconds;
  report->t_factor = p->timers[1].seconds;
  report->t_accumulate = p->timers[2].seconds;
  report->t_jac = p->timers[5].seconds;       // device-side spans (HIP events), like the other stage times
  report->t_solve = p->timers[6].seconds;
  if (fused) report->t_cost = p->timers[7].seconds;       // cost passes queued behind the solves: device-side spans
  return CBA_OK;
}

int cba_kernel_stats(cba_problem* p, int32_t which, double* seconds, double* flops, double* bytes, int32_t* launches) {
  if (!p || which < 0 || which > 4) { set_error("cba_kernel_stats: bad argument"); return CBA_ERR_ARG; }
  CBA_HIP(hipSetDevice(p->device));
  CBA_TRY(timers_collect(p));
  const KernelTimer& t = p->timers[which];
  if (seconds) *seconds = t.seconds;
  if (flops) *flops = t.flops;
  if (bytes) *bytes = t.bytes;
  if (launches) *launches = t.launches;
  return CBA_OK;
}

int cba_debug_dump(cba_problem* p, int32_t what, void* out, size_t bytes) {
  if (!p || !out) { set_error("cba_debug_dump: bad argument"); return CBA_ERR_ARG; }
  CBA_HIP(hipSetDevice(p->device));
  CBA_HIP(hipStreamSynchronize(p->stream));
  const Layout& L = p->L;
  const size_t n = (size_t)p->n_obs, bs = L.block_size, nb = L.n_blocks, dd = L.dense_dof, ld = p->n_pad;
  auto copy = [&](const void* src, size_t need) -> int {
    if (bytes < need) { set_error("cba_debug_dump: buffer too small"); return CBA_ERR_ARG; }
    if (need) CBA_HIP(hipMemcpy(out, src, need, hipMemcpyDeviceToHost));
    return CBA_OK;
  };
  auto copy2d = [&](const double* src, size_t rows, size_t cols) -> int {
    if (bytes < rows * cols * sizeof(double)) { set_error("cba_debug_dump: buffer too small"); return CBA_ERR_ARG; }
    if (rows && cols)
      CBA_HIP(hipMemcpy2D(out, cols * sizeof(double), src, ld * sizeof(double), cols * sizeof(double), rows, hipMemcpyDeviceToHost));
    return CBA_OK;
  };
  // block-part items are stored in slot order on the device; present them in imageset order
  auto unpermute_rows = [&](size_t row_doubles) {
    if (p->pose_slot_host.empty()) return;
    std::vector<double> tmp((size_t)nb * bs * row_doubles);
    std::memcpy(tmp.data(), out, tmp.size() * sizeof(double));
    double* o = static_cast<double*>(out);
    for (size_t i = 0; i < nb; ++i)
      std::memcpy(o + i * bs * row_doubles, tmp.data() + (size_t)p->pose_slot_host[i] * bs * row_doubles, bs * row_doubles * sizeof(double));
  };
  // dense-part items use the engine's tiled grid order on the device; present them in the reference order
  const std::vector<int>& dperm = p->dense_perm_host;
  auto unpermute_cols = [&](size_t rows, size_t row_offset_in_out) {   // out[r][j] = tmp[r][dperm[j]]
    double* o = static_cast<double*>(out) + row_offset_in_out;
    std::vector<double> tmp(dd);
    for (size_t r = 0; r < rows; ++r) {
      std::memcpy(tmp.data(), o + r * dd, dd * sizeof(double));
      for (size_t j = 0; j < dd; ++j) o[r * dd + j] = tmp[dperm[j]];
    }
  };
  switch (what) {
    case CBA_DUMP_COST_VECTOR: return copy(p->cost_ref, n * sizeof(double));
    case CBA_DUMP_TEST_COST_VECTOR: return copy(p->cost_test, n * sizeof(double));
    case CBA_DUMP_PIXELS: return copy(p->pixels, 2 * n * sizeof(double));
    case CBA_DUMP_FLAGS: return copy(p->flags, n);
    case CBA_DUMP_JACOBIANS: return copy(p->jrec, n * p->rec_doubles * sizeof(double));
    case CBA_DUMP_BLOCK_DIAG_H: { int rc = copy(p->Dblk, nb * bs * bs * sizeof(double)); if (rc == CBA_OK) unpermute_rows(bs); return rc; }
    case CBA_DUMP_BLOCK_DIAG_B: { int rc = copy(p->bblk, nb * bs * sizeof(double)); if (rc == CBA_OK) unpermute_rows(1); return rc; }
    case CBA_DUMP_OFF_DIAG_H: {
      int rc = copy2d(p->B, nb * bs, dd);
      if (rc == CBA_OK) { unpermute_rows(dd); unpermute_cols(nb * bs, 0); }
      return rc;
    }
    case CBA_DUMP_DENSE_H: {
      int rc = copy2d(p->Hdd, dd, dd);
      if (rc != CBA_OK) return rc;
      // upper triangle in the engine order -> upper triangle in the reference order
      std::vector<double> tmp(dd * dd);
      std::memcpy(tmp.data(), out, tmp.size() * sizeof(double));
      double* o = static_cast<double*>(out);
      for (size_t i = 0; i < dd; ++i)
        for (size_t j = 0; j < dd; ++j) {
          if (j < i) { o[i * dd + j] = 0.0; continue; }
          const size_t a = dperm[i], b = dperm[j];
          o[i * dd + j] = a <= b ? tmp[a * dd + b] : tmp[b * dd + a];
        }
      return CBA_OK;
    }
    case CBA_DUMP_DENSE_B: { int rc = copy(p->bd, dd * sizeof(double)); if (rc == CBA_OK) unpermute_cols(1, 0); return rc; }
    case CBA_DUMP_X: {
      if (bytes < (size_t)L.total_dof * sizeof(double)) { set_error("cba_debug_dump: buffer too small"); return CBA_ERR_ARG; }
      CBA_HIP(hipMemcpy(out, p->x, (size_t)L.total_dof * sizeof(double), hipMemcpyDeviceToHost));
      if (!p->pose_slot_host.empty()) unpermute_rows(1);   // the block part comes first in x (eliminate_points = 0)
      unpermute_cols(1, L.block_dof);
      return CBA_OK;
    }
    default: set_error("cba_debug_dump: unknown item"); return CBA_ERR_ARG;
  }
}

// ---- model-level entry points: device-resident camera model -----------------------------------------
struct cba_model {
  cba_camera cam{};
  int device = 0;
  double* d_grid = nullptr; CamDev* d_cam = nullptr;
  // scratch, grown on demand
  int64_t cap = 0;
  double *d_a = nullptr, *d_b = nullptr, *d_c = nullptr, *d_j = nullptr; uint8_t* d_ok = nullptr;
};
static int model_reserve(cba_model* m, int64_t n) {
  if (n <= m->cap) return CBA_OK;
  auto F = [](void* q) { if (q) hipFree(q); };
  F(m->d_a); F(m->d_b); F(m->d_c); F(m->d_j); F(m->d_ok);
  m->d_a = m->d_b = m->d_c = m->d_j = nullptr; m->d_ok = nullptr; m->cap = 0;
  const int64_t cap = n < 256 ? 256 : n;
  CBA_TRY(dev_alloc(&m->d_a, 3 * (size_t)cap));     // local points / pixels (in)
  CBA_TRY(dev_alloc(&m->d_b, 6 * (size_t)cap));     // pixels / lines (out)
  CBA_TRY(dev_alloc(&m->d_c, 2 * (size_t)cap));     // initial pixels
  CBA_TRY(dev_alloc(&m->d_j, 12 * (size_t)cap));    // un-projection Jacobians
  CBA_TRY(dev_alloc(&m->d_ok, (size_t)cap));
  m->cap = cap;
  return CBA_OK;
}
void cba_model_destroy(cba_model* m) {
  if (!m) return;
  hipSetDevice(m->device);
  auto F = [](void* q) { if (q) hipFree(q); };
  F(m->d_grid); F(m->d_cam); F(m->d_a); F(m->d_b); F(m->d_c); F(m->d_j); F(m->d_ok);
  delete m;
}
int64_t cba_fd_redo_overflow(cba_problem* p) {
  if (!p || !p->fd_redo_count) return -1;
  int v = 0;
  if (hipSetDevice(p->device) != hipSuccess || hipStreamSynchronize(p->stream) != hipSuccess ||
      hipMemcpy(&v, p->fd_redo_count + 2, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return v;
}
int cba_debug_fd_redo_counts(cba_problem* p, int64_t out[3]) {
  if (!p || !out || !p->fd_redo_count) { set_error("cba_debug_fd_redo_counts: bad argument"); return CBA_ERR_ARG; }
  int v[3] = {0, 0, 0};
  CBA_HIP(hipSetDevice(p->device));
  CBA_HIP(hipStreamSynchronize(p->stream));
  CBA_HIP(hipMemcpy(v, p->fd_redo_count, sizeof(v), hipMemcpyDeviceToHost));
  for (int i = 0; i < 3; ++i) out[i] = v[i];
  return CBA_OK;
}

int cba_model_set_grid(cba_model* m, const double* grid) {
  if (!m || !grid) { set_error("cba_model_set_grid: bad argument"); return CBA_ERR_ARG; }
  CBA_HIP(hipSetDevice(m->device));
  const size_t G = (size_t)m->cam.grid_w * m->cam.grid_h;
  CBA_HIP(hipMemcpy(m->d_grid, grid, (m->cam.model_type == CBA_CENTRAL_GENERIC ? 3 : 6) * G * sizeof(double), hipMemcpyHostToDevice));
  return CBA_OK;
}
int cba_model_create(const cba_camera* camera, const double* grid, int32_t device, cba_model** out) {
  if (!camera || !grid || !out || !camera_ok(*camera)) { set_error("bad camera / grid"); return CBA_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device available (the engine has no CPU fallback)"); return CBA_ERR_HIP; }
  if (device < 0 || device >= ndev) { set_error("bad device ordinal"); return CBA_ERR_ARG; }
  CBA_HIP(hipSetDevice(device));
  cba_model* m = new cba_model();
  struct Guard { cba_model* q; ~Guard() { if (q) cba_model_destroy(q); } } guard{m};
  m->cam = *camera; m->device = device;
  const size_t G = (size_t)camera->grid_w * camera->grid_h;
  CBA_TRY(dev_alloc(&m->d_grid, (camera->model_type == CBA_CENTRAL_GENERIC ? 3 : 6) * G));
  CBA_TRY(cba_model_set_grid(m, grid));
  CamDev h = make_camdev(*camera, m->d_grid, nullptr, 0);
  CBA_TRY(dev_alloc(&m->d_cam, 1));
  CBA_HIP(hipMemcpy(m->d_cam, &h, sizeof(CamDev), hipMemcpyHostToDevice));
  guard.q = nullptr;
  *out = m;
  return CBA_OK;
}
int cba_model_project(cba_model* m, int64_t n, const double* local_points, const double* init_pixels, double* pixels, uint8_t* ok) {
  if (!m || n < 0 || (n > 0 && (!local_points || !pixels || !ok))) { set_error("cba_model_project: bad argument"); return CBA_ERR_ARG; }
  if (n == 0) return CBA_OK;
  CBA_HIP(hipSetDevice(m->device));
  CBA_TRY(model_reserve(m, n));
  CBA_HIP(hipMemcpy(m->d_a, local_points, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
  if (init_pixels) CBA_HIP(hipMemcpy(m->d_c, init_pixels, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
  CBA_TRY(launch_project_points(m->d_cam, m->cam.model_type, n, m->d_a, init_pixels ? m->d_c : nullptr, m->d_b, m->d_ok, nullptr));
  CBA_HIP(hipMemcpy(pixels, m->d_b, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
  CBA_HIP(hipMemcpy(ok, m->d_ok, (size_t)n, hipMemcpyDeviceToHost));
  return CBA_OK;
}
int cba_model_unproject(cba_model* m, int64_t n, const double* pixels, double* lines, double* jacobians, uint8_t* ok) {
  if (!m || n < 0 || (n > 0 && (!pixels || !lines || !ok))) { set_error("cba_model_unproject: bad argument"); return CBA_ERR_ARG; }
  if (n == 0) return CBA_OK;
  CBA_HIP(hipSetDevice(m->device));
  CBA_TRY(model_reserve(m, n));
  CBA_HIP(hipMemcpy(m->d_a, pixels, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
  CBA_TRY(launch_unproject(m->d_cam, m->cam.model_type, n, m->d_a, m->d_b, jacobians ? m->d_j : nullptr, m->d_ok, nullptr));
  CBA_HIP(hipMemcpy(lines, m->d_b, sizeof(double) * 6 * n, hipMemcpyDeviceToHost));
  CBA_HIP(hipMemcpy(ok, m->d_ok, (size_t)n, hipMemcpyDeviceToHost));
  if (jacobians) CBA_HIP(hipMemcpy(jacobians, m->d_j, sizeof(double) * 12 * n, hipMemcpyDeviceToHost));
  return CBA_OK;
}

// ---- stateless entry points (one-shot wrappers) ------------------------------------------------------
int cba_project(const cba_camera* camera, const double* grid, int64_t n, const double* local_points,
                const double* init_pixels, double* pixels, uint8_t* ok, int32_t device) {
  if (n < 0 || (n > 0 && (!local_points || !pixels || !ok))) { set_error("cba_project: bad argument"); return CBA_ERR_ARG; }
  cba_model* m = nullptr;
  CBA_TRY(cba_model_create(camera, grid, device, &m));
  const int rc = cba_model_project(m, n, local_points, init_pixels, pixels, ok);
  cba_model_destroy(m);
  return rc;
}

int cba_unproject(const cba_camera* camera, const double* grid, int64_t n, const double* pixels, double* lines,
                  double* jacobians, uint8_t* ok, int32_t device) {
  if (n < 0 || (n > 0 && (!pixels || !lines || !ok))) { set_error("cba_unproject: bad argument"); return CBA_ERR_ARG; }
  cba_model* m = nullptr;
  CBA_TRY(cba_model_create(camera, grid, device, &m));
  const int rc = cba_model_unproject(m, n, pixels, lines, jacobians, ok);
  cba_model_destroy(m);
  return rc;
}

int cba_schur_solve(int32_t block_size, int32_t n_blocks, int32_t dense_dof, const double* block_diag_H,
                    const double* off_diag_H, const double* dense_H, const double* block_diag_b,
                    const double* dense_b, double* x, int32_t device) {
  return cba_schur_solve_opt(block_size, n_blocks, dense_dof, block_diag_H, off_diag_H, dense_H, block_diag_b, dense_b, x, nullptr, device);
}
int cba_schur_solve_opt(int32_t block_size, int32_t n_blocks, int32_t dense_dof, const double* block_diag_H,
                        const double* off_diag_H, const double* dense_H, const double* block_diag_b,
                        const double* dense_b, double* x, const cba_solver_options* options, int32_t device) {
  if (block_size < 1 || block_size > 6 || n_blocks < 1 || dense_dof < 1 || !block_diag_H || !off_diag_H || !dense_H ||
      !block_diag_b || !dense_b || !x) { set_error("cba_schur_solve: bad argument"); return CBA_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device available (the engine has no CPU fallback)"); return CBA_ERR_HIP; }
  if (device < 0 || device >= ndev) { set_error("bad device ordinal"); return CBA_ERR_ARG; }
  CBA_HIP(hipSetDevice(device));
  const int bs = block_size, nb = n_blocks, dd = dense_dof, bdof = bs * nb;
  int n_pad, n_fact; padded_dims(dd, &n_pad, &n_fact);
  const int ld = n_pad, Kpad = round_up(bdof, 16);
  double *Dblk, *bblk, *Dinv, *dinvb, *B, *W, *Hdd, *bd, *S, *xd, *gws; int* status;
  DevPool pool;
  CBA_TRY(pool.alloc(&gws, (size_t)gemv_t_workspace_doubles(dd)));
  CBA_TRY(pool.alloc(&Dblk, (size_t)nb * bs * bs)); CBA_TRY(pool.alloc(&bblk, (size_t)bdof)); CBA_TRY(pool.alloc(&Dinv, (size_t)nb * bs * bs));
  CBA_TRY(pool.alloc(&dinvb, (size_t)Kpad)); CBA_TRY(pool.alloc(&B, (size_t)Kpad * ld)); CBA_TRY(pool.alloc(&W, (size_t)Kpad * ld));
  CBA_TRY(pool.alloc(&Hdd, (size_t)ld * ld)); CBA_TRY(pool.alloc(&bd, (size_t)ld)); CBA_TRY(pool.alloc(&S, (size_t)ld * ld));
  CBA_TRY(pool.alloc(&xd, (size_t)bdof + ld)); CBA_TRY(pool.alloc(&status, 1));
  CBA_HIP(hipMemset(B, 0, sizeof(double) * (size_t)Kpad * ld)); CBA_HIP(hipMemset(W, 0, sizeof(double) * (size_t)Kpad * ld));
  CBA_HIP(hipMemset(Hdd, 0, sizeof(double) * (size_t)ld * ld)); CBA_HIP(hipMemset(S, 0, sizeof(double) * (size_t)ld * ld));
  CBA_HIP(hipMemset(bd, 0, sizeof(double) * ld)); CBA_HIP(hipMemset(status, 0, sizeof(int))); CBA_HIP(hipMemset(dinvb, 0, sizeof(double) * Kpad));
  CBA_HIP(hipMemset(xd, 0, sizeof(double) * ((size_t)bdof + ld)));
  // upper triangles only: the reference fills lower triangles with NaN in its golden test, so copy and scrub
  std::vector<double> hD(block_diag_H, block_diag_H + (size_t)nb * bs * bs), hH((size_t)dd * dd);
  for (int b = 0; b < nb; ++b)
    for (int r = 0; r < bs; ++r)
      for (int c = 0; c < r; ++c) hD[(size_t)b * bs * bs + r * bs + c] = 0.0;
  for (int r = 0; r < dd; ++r)
    for (int c = 0; c < dd; ++c) hH[(size_t)r * dd + c] = (c >= r) ? dense_H[(size_t)r * dd + c] : 0.0;
  CBA_HIP(hipMemcpy(Dblk, hD.data(), sizeof(double) * hD.size(), hipMemcpyHostToDevice));
  CBA_HIP(hipMemcpy(bblk, block_diag_b, sizeof(double) * bdof, hipMemcpyHostToDevice));
  CBA_HIP(hipMemcpy2D(B, ld * sizeof(double), off_diag_H, dd * sizeof(double), dd * sizeof(double), bdof, hipMemcpyHostToDevice));
  CBA_HIP(hipMemcpy2D(Hdd, ld * sizeof(double), hH.data(), dd * sizeof(double), dd * sizeof(double), dd, hipMemcpyHostToDevice));
  CBA_HIP(hipMemcpy(bd, dense_b, sizeof(double) * dd, hipMemcpyHostToDevice));
  LdltWorkspace w;
  LdltGuard wguard{&w};
  CBA_TRY(ldlt_workspace_alloc(w, n_pad));
  apply_solver_options(w, options);
  CBA_HIP(hipMemset(w.status, 0, sizeof(int)));
  hipStream_t s = nullptr;
  CBA_TRY(launch_block_inverse(Dblk, bblk, 0.0, bs, nb, Dinv, dinvb, status, s));
  CBA_TRY(launch_dinv_times_B_ld(Dinv, B, bs, nb, dd, ld, W, s));
  CBA_TRY(schur_gemm(B, W, Kpad, ld, Hdd, S, n_pad, ld, dd, 1, 0.0, nullptr, s));
  CBA_TRY(launch_gemv_t_strided(B, bdof, dd, ld, dinvb, bd, S + (ld - 1), ld, gws, s));
  CBA_TRY(ldlt_factor(S, n_fact, ld, w, s, nullptr));
  CBA_TRY(ldlt_back_solve(S, n_fact, ld, ld - 1, w, xd + bdof, s));
  CBA_TRY(launch_gemv_n(W, bdof, dd, ld, xd + bdof, dinvb, xd, s));
  CBA_HIP(hipDeviceSynchronize());
  int st[2];
  CBA_HIP(hipMemcpy(&st[0], status, sizeof(int), hipMemcpyDeviceToHost));
  CBA_HIP(hipMemcpy(&st[1], w.status, sizeof(int), hipMemcpyDeviceToHost));
  CBA_HIP(hipMemcpy(x, xd, sizeof(double) * (bdof + dd), hipMemcpyDeviceToHost));
  if (st[1] == 3) { set_error("cba_schur_solve: a dataflow launch of the factorisation timed out"); return CBA_ERR_TIMEOUT; }
  if (st[0] || st[1]) { set_error("cba_schur_solve: zero pivot"); return CBA_ERR_NUMERIC; }
  return CBA_OK;
}

int64_t cba_gridfirst_plan_query(const cba_camera* cameras, int32_t n_cameras, int32_t n_images, int32_t n_points, int32_t strips,
                                 int32_t single_tile_tasks, int32_t what, void* out, int64_t capacity_bytes) {
  GfPlan pl;
  int rc = gf_build_plan(cameras, n_cameras, n_images, n_points, strips, single_tile_tasks, &pl);
  if (rc != CBA_OK) { set_error("cba_gridfirst_plan_query: bad argument"); return rc; }
  auto give = [&](const void* src, size_t bytes) -> int64_t {
    if (out && capacity_bytes >= (int64_t)bytes && bytes) std::memcpy(out, src, bytes);
    return (int64_t)bytes;
  };
  switch (what) {
    case 0: {
      const int32_t h[16] = {pl.G, pl.Gf, pl.n_rp, pl.n_border, pl.n_fact, pl.n_pad, pl.nbg, pl.nbf, pl.ntc, (int32_t)pl.chains.size(),
                             (int32_t)pl.tasks.size(), pl.n_tasks0, (int32_t)pl.ivals.size(), pl.mask_words, pl.half_bandwidth, pl.strips[0]};
      return give(h, sizeof(h));
    }
    case 1: return give(pl.f_of_grid.data(), pl.f_of_grid.size() * sizeof(int));
    case 2: return give(pl.chains.data(), pl.chains.size() * sizeof(GfChain));
    case 3: return give(pl.tasks.data(), pl.tasks.size() * sizeof(GfTask));
    case 4: return give(pl.ivals.data(), pl.ivals.size() * sizeof(GfIval));
    case 5: return give(pl.rowmask.data(), pl.rowmask.size() * sizeof(uint64_t));
    case 6: { const double f[3] = {pl.flops_grid, pl.flops_update, pl.flops_border}; return give(f, sizeof(f)); }
    default:
      if (what >= 16 && what < 16 + n_cameras) return give(pl.gperm[what - 16].data(), pl.gperm[what - 16].size() * sizeof(int));
      set_error("cba_gridfirst_plan_query: unknown item");
      return CBA_ERR_ARG;
  }
}

int cba_fit_grid_to_directions(const cba_camera* camera, double* grid, int64_t n, const double* grid_points,
                               const double* directions, int32_t max_iteration_count, cba_fit_report* report, int32_t device) {
  if (!camera || !grid || n < 0 || (n > 0 && (!grid_points || !directions)) || max_iteration_count < 0 || !camera_ok(*camera) ||
      camera->model_type != CBA_CENTRAL_GENERIC) { set_error("cba_fit_grid_to_directions: bad argument"); return CBA_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device available (the engine has no CPU fallback)"); return CBA_ERR_HIP; }
  if (device < 0 || device >= ndev) { set_error("bad device ordinal"); return CBA_ERR_ARG; }
  CBA_HIP(hipSetDevice(device));
  const int gw = camera->grid_w, gh = camera->grid_h, G = gw * gh, dof = 2 * G;
  int n_pad, n_fact; padded_dims(dof, &n_pad, &n_fact);
  const int ld = n_pad;
  cba_fit_report rep{};
  double *g[2] = {nullptr, nullptr}, *tang, *gp, *dirs, *cost_ref, *cost_test, *rec, *H, *b, *S, *x, *partials, *red8, *scal;
  int *keys, *count, *start, *fill, *order, *status;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  DevPool pool;
  CBA_TRY(pool.alloc(&g[0], 3 * (size_t)G)); CBA_TRY(pool.alloc(&g[1], 3 * (size_t)G)); CBA_TRY(pool.alloc(&tang, 6 * (size_t)G));
  CBA_TRY(pool.alloc(&gp, 2 * nn)); CBA_TRY(pool.alloc(&dirs, 3 * nn)); CBA_TRY(pool.alloc(&cost_ref, 3 * nn)); CBA_TRY(pool.alloc(&cost_test, 3 * nn));
  CBA_TRY(pool.alloc(&rec, nn * 99)); CBA_TRY(pool.alloc(&keys, nn)); CBA_TRY(pool.alloc(&order, nn));
  CBA_TRY(pool.alloc(&count, (size_t)G + 1)); CBA_TRY(pool.alloc(&start, (size_t)G + 1)); CBA_TRY(pool.alloc(&fill, (size_t)G + 1));
  CBA_TRY(pool.alloc(&H, (size_t)ld * ld)); CBA_TRY(pool.alloc(&b, (size_t)ld)); CBA_TRY(pool.alloc(&S, (size_t)ld * ld)); CBA_TRY(pool.alloc(&x, (size_t)ld));
  CBA_TRY(pool.alloc(&partials, 256 * 8)); CBA_TRY(pool.alloc(&red8, 8)); CBA_TRY(pool.alloc(&scal, 8)); CBA_TRY(pool.alloc(&status, 1));
  CBA_HIP(hipMemcpy(g[0], grid, sizeof(double) * 3 * G, hipMemcpyHostToDevice));
  if (n) {
    CBA_HIP(hipMemcpy(gp, grid_points, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    CBA_HIP(hipMemcpy(dirs, directions, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
  }
  CBA_HIP(hipMemset(status, 0, sizeof(int)));
  LdltWorkspace w;
  LdltGuard wguard{&w};
  CBA_TRY(ldlt_workspace_alloc(w, n_pad));
  hipStream_t s = nullptr;
  CBA_TRY(make_main_stream(&s));
  auto read = [&](const double* dev, double* host, int k) -> int {
    CBA_HIP(hipMemcpyAsync(host, dev, sizeof(double) * k, hipMemcpyDeviceToHost, s));
    CBA_HIP(hipStreamSynchronize(s));
    return CBA_OK;
  };
  int cur = 0, rc = CBA_OK;
  double lambda = -1.0, last_cost = 0.0;
  const double init_lambda_factor = (double)0.001f;
  for (int iteration = 0; iteration < max_iteration_count && rc == CBA_OK; ++iteration) {
    double t0 = now_s();
    if ((rc = launch_tangents(g[cur], tang, G, s))) break;
    if ((rc = launch_fit_pass(true, gw, gh, g[cur], tang, n, gp, dirs, cost_ref, rec, keys, status, s))) break;
    CBA_HIP(hipMemsetAsync(H, 0, sizeof(double) * (size_t)ld * ld, s));
    CBA_HIP(hipMemsetAsync(b, 0, sizeof(double) * (size_t)ld, s));
    if ((rc = launch_fit_accumulate(gw, gh, n, rec, keys, count, start, fill, order, H, ld, b, s))) break;
    if ((rc = launch_reduce_costs(cost_ref, nullptr, nullptr, 3 * n, partials, red8, s))) break;
    if ((rc = launch_fit_diag_sum(H, ld, dof, scal, s))) break;
    double h8[8], hsum = 0; int st = 0;
    if ((rc = read(red8, h8, 8)) || (rc = read(scal, &hsum, 1))) break;
    CBA_HIP(hipMemcpy(&st, status, sizeof(int), hipMemcpyDeviceToHost));
    rep.t_pass += now_s() - t0;
    if (st == 3) { set_error("cba_fit_grid_to_directions: a grid point lies outside the grid's 4x4 patches"); rc = CBA_ERR_ARG; break; }
    last_cost = h8[0];
    if (iteration == 0) { rep.initial_cost = last_cost; lambda = init_lambda_factor * hsum / dof; }
    if (last_cost == 0) break;
    bool applied = false;
    for (int lm = 0; lm < 10 && rc == CBA_OK; ++lm) {
      rep.lm_attempts += 1;
      t0 = now_s();
      CBA_HIP(hipMemcpyAsync(S, H, sizeof(double) * (size_t)ld * ld, hipMemcpyDeviceToDevice, s));
      if ((rc = launch_finish_diag(S, ld, dof, n_pad, lambda, s))) break;
      if ((rc = launch_fit_set_rhs(S, ld, b, dof, s))) break;
      CBA_HIP(hipMemsetAsync(w.status, 0, sizeof(int), s));
      if ((rc = ldlt_factor(S, n_fact, ld, w, s, nullptr))) break;
      if ((rc = ldlt_back_solve(S, n_fact, ld, ld - 1, w, x, s))) break;
      CBA_HIP(hipMemcpyAsync(&st, w.status, sizeof(int), hipMemcpyDeviceToHost, s));
      CBA_HIP(hipStreamSynchronize(s));
      rep.t_solve += now_s() - t0;
      if (st != 0) { lambda = 2.f * lambda; continue; }     // zero pivot: treated like the reference's NaN update
      t0 = now_s();
      if ((rc = launch_update_direction_grid(g[cur], x, G, g[cur ^ 1], s))) break;
      if ((rc = launch_fit_pass(false, gw, gh, g[cur ^ 1], tang, n, gp, dirs, cost_test, nullptr, nullptr, status, s))) break;
      if ((rc = launch_reduce_costs(cost_ref, cost_test, nullptr, 3 * n, partials, red8, s))) break;
      if ((rc = read(red8, h8, 8))) break;
      rep.t_pass += now_s() - t0;
      if (h8[4] > 0 && h8[3] < h8[2]) {                       // CostIsSmallerThan
        cur ^= 1;
        lambda = 0.5f * lambda;
        applied = true;
        rep.iterations_performed += 1;
        last_cost = h8[1];
        break;
      }
      lambda = 2.f * lambda;
    }
    if (!applied || last_cost == 0) break;
  }
  if (rc == CBA_OK) {
    rep.final_cost = last_cost; rep.lambda = lambda;
    if (hipMemcpy(grid, g[cur], sizeof(double) * 3 * G, hipMemcpyDeviceToHost) != hipSuccess) { set_error("cba_fit_grid_to_directions: copy back failed"); rc = CBA_ERR_HIP; }
    if (report) *report = rep;
  }
  return rc;
}

}  // extern "C"
