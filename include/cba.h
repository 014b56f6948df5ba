/*
 * cba.h -- C-ABI of the MI355X-native bundle-adjustment engine (libcalib_ba_hip.so).
 *
 * Drop-in boundary for the reference's JointOptimization path.  The reference has no FFI layer;
 * the seam is the free function
 *
 *   double vis::OptimizeJointly(Dataset&, BAState*, int max_iteration_count, double init_lambda,
 *                               double numerical_diff_delta, double regularization_weight,
 *                               bool localize_only, bool eliminate_points, SchurMode schur_mode,
 *                               double* final_lambda, bool* performed_an_iteration, ...)
 *   -- applications/camera_calibration/src/camera_calibration/bundle_adjustment/joint_optimization.h:53-70
 *
 * A host adapter with exactly that signature (camera_calibration_amd/host/joint_optimization_hip.cc)
 * marshals Dataset/BAState into the packed arrays below and calls cba_step once per outer iteration
 * (the reference runs optimizer.Optimize(max_iteration_count = 1) per outer iteration,
 * joint_optimization.cc:906-940).  See INTEGRATION.md for the reference-side binding.
 *
 * Conventions: plain pointers and sizes only; all arrays are HOST pointers unless stated; fp64
 * unless stated; every call returns CBA_OK (0) or a negative error code and never aborts.
 * File:line citations are relative to the reference tree; APP = applications/camera_calibration/
 * src/camera_calibration, LV = libvis/src/libvis.
 */
#ifndef CBA_H_
#define CBA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  CBA_OK = 0,
  CBA_ERR_ARG = -1,        /* bad argument (replaces the reference's CHECK() aborts) */
  CBA_ERR_HIP = -2,        /* HIP runtime error; cba_last_error() has the text */
  CBA_ERR_STATE = -3,      /* call sequence error (e.g. step before set_state) */
  CBA_ERR_NUMERIC = -4,    /* factorisation broke down (zero pivot) */
  CBA_ERR_UNSUPPORTED = -5,
  CBA_ERR_TIMEOUT = -6     /* a dataflow launch of the reduced solve gave up waiting for another workgroup (3 s): device fault or a
                              launch that could not become resident; nothing was written to the state */
};

/* CameraModel::Type of the two generic models (APP/models/camera_model.h:44-52) */
enum { CBA_CENTRAL_GENERIC = 0, CBA_NONCENTRAL_GENERIC = 1 };

/* Calibrated rectangle and grid resolution of one camera: the constructor arguments of
 * CentralGenericModel / NoncentralGenericModel (APP/models/central_generic.h:54-58). */
typedef struct {
  int32_t model_type;
  int32_t width, height;
  int32_t calib_min_x, calib_min_y, calib_max_x, calib_max_y;
  int32_t grid_w, grid_h;
} cba_camera;

/* Cross-rank sum of a DEVICE fp64 buffer (image sharding).  Called on the host with the
 * engine's stream idle; must return after the reduced values are visible on the device. */
typedef int (*cba_allreduce_fn)(void* device_ptr, int64_t count, void* user);

/* Collectives of the distributed reduced solve (optional, cba_config.collective).  Blocking host call on DEVICE fp64
 * buffers; the buffers' producers have completed when it is called, other engine work may be running on the device (the
 * call is what that work overlaps with), and the results must be visible on the device when it returns.  `count` is the
 * per-rank block size in doubles:
 *   CBA_COLL_ALLREDUCE_SUM       recvbuf[count] summed over the ranks in place (sendbuf unused)
 *   CBA_COLL_REDUCE_SCATTER_SUM  sendbuf[world * count], block r summed over the ranks into rank r's recvbuf[count]   (ncclReduceScatter)
 *   CBA_COLL_ALLGATHER           sendbuf[count] of rank r into block r of every rank's recvbuf[world * count]        (ncclAllGather)
 * Returns 0 on success. */
enum { CBA_COLL_ALLREDUCE_SUM = 0, CBA_COLL_REDUCE_SCATTER_SUM = 1, CBA_COLL_ALLGATHER = 2 };
typedef int (*cba_collective_fn)(int32_t op, void* sendbuf, void* recvbuf, int64_t count, void* user);

/* Scheduling options of the reduced solve (LV/lm_optimizer.h:1361, Eigen's LDLT there).  They select between equivalent schedules:
 * results change only in the last bits.  All zero = defaults.  Per problem / per call -- nothing here is process-wide. */
typedef struct {
  int32_t factor_tail_rows;   /* rows left to the FINAL dataflow launch of the two-level factorisation;
                                 0 = default (8192; in the distributed solve, where every rank repeats the final launch, 6144 with
                                 2-3 ranks and 4096 from 4 ranks on), clamped to what the launch has flags for.  Smaller values
                                 give small systems the super-panel structure of large ones (the tests use that) */
  int32_t back_substitution;  /* 0 = one dataflow launch (default); 1 = panels of 256 rows (98 launches at BASELINE configs[1]) */
  int32_t elimination;        /* order in which the unknowns are eliminated.  The reference eliminates the 6 x 6 pose blocks and factors
                                 the dense rest of D = 3 P + 6 C + (grid unknowns) rows (APP/bundle_adjustment/joint_optimization.cc:794-804,
                                 LV/lm_optimizer.h:1247-1369); (H + lambda I) x = b has one solution, so every exact order gives the same x
                                 up to rounding.  1 = that order (pose-first).  2 = grid-first: an observation touches a 4 x 4 window of
                                 control points (APP/models/central_grid.h:199-209), so the grid x grid block is banded; the grid is
                                 eliminated by a block-sparse LDL^T and the dense border [rig | points | poses] of 6 N + 3 P + 6 C rows is
                                 factored instead (5 445 instead of 12 525 rows at BASELINE configs[1]: 0.39 instead of 0.77 TFLOP per
                                 solve).  0 = automatic: grid-first on one GPU when a flop model of the two orders favours it (poses the
                                 Schur blocks, intrinsics optimised, at least 2048 grid unknowns), pose-first otherwise (always with image
                                 sharding: the reduced system that crosses the ranks is the pose-first one).  DESIGN.md section 3a */
  int32_t grid_strips;        /* grid-first order: independent strips the long grid dimension is cut into (separated by 3-line
                                 separators that are eliminated after the strips): one pivot chain per strip instead of one chain of
                                 all grid unknowns.  0 = automatic (at most 4) */
  int32_t grid_single_tile_tasks; /* grid-first order: 1 = one task per 64-column border tile in the block-sparse launch (the first version
                                 of round 6); 0 = default: the two tiles of a 128-column border tile share a task (half as many workgroup
                                 slots wait at the pivot chains' frontiers) */
} cba_solver_options;

typedef struct {
  int32_t n_cameras;
  const cba_camera* cameras;
  int32_t n_images;          /* used imagesets (BAState::image_used already applied, joint_optimization.cc:80-90) */
  int32_t n_points;
  double numerical_diff_delta;   /* OptimizeJointly argument */
  int32_t localize_only;         /* OptimizeJointly argument */
  int32_t eliminate_points;      /* OptimizeJointly argument (0 = eliminate imageset poses, the CLI's mode) */
  int32_t device;                /* HIP device ordinal */
  /* multi-GPU (optional): this rank owns a contiguous range of the imagesets; dense-part blocks are
   * summed over ranks with `allreduce` once per Gauss-Newton step. */
  cba_allreduce_fn allreduce;
  void* allreduce_user;
  int32_t n_images_global;       /* total imagesets over all ranks (0 = n_images) */
  void* reduce_buffer;           /* optional caller-owned DEVICE buffer for the packed reduced system (all-reduced in place) */
  int64_t reduce_buffer_doubles; /* its size; must be >= cba_reduce_buffer_doubles() */
  /* 1 = run-to-run reproducible normal equations: JtJ / Jtr terms are accumulated in 64-bit fixed point (integer
   * atomics are order-independent) instead of fp64 atomics; two runs on the same input then give bit-identical H, b,
   * x, costs and states.  The reference is single-threaded and therefore reproducible; this is the mode that matches
   * that property.  Resolution: one power-of-two quantum per pass for JtJ (2^62 / (n_obs x the largest single contribution))
   * and a finer one for Jtr -- ABSOLUTE, ~19 digits below the largest POSSIBLE sum (n_obs contributions of the largest size;
   * the largest actual entry is orders of magnitude below that because an entry collects ~1e2..1e3 contributions, not n_obs).
   * Measured against the fp64-atomic mode at BASELINE configs[1]: diagonal entries within 1e-8 of the largest one agree to
   * 2e-6 relative, smaller ones (control points that a handful of observations touch) to 3e-14 of the largest entry in absolute
   * terms; H, B, b to 1e-11 of their maxima -- below the truncation error of the forward-difference Jacobians (delta = 1e-4).  Costs about 1 ms per LM iteration at BASELINE configs[1]
   * (DESIGN.md section 4a).  0 = fp64 atomics. */
  int32_t deterministic;
  /* Distributed reduced solve (multi-GPU, optional; needs `allreduce`).  0: the reduced system is all-reduced and its
   * factorisation replicated on every rank (14 ms at BASELINE configs[1]).  1: the two-level factorisation is split -- the
   * 512-column groups of the reduced system are owned block-cyclically by the ranks; the partial systems are reduce-scattered
   * straight into the owners (only the first 2048 rows are all-reduced); per super-panel of 2048 rows every rank runs the
   * latency-bound dataflow launch on the complete row band (identical arithmetic on identical data) and applies the K = 2048
   * trailing update only to its own column groups, next band first, which is then all-gathered from its owners while the
   * rest of the update is still running.  Link volume per solve = that of the one all-reduce it replaces.  For reduced
   * systems like BASELINE configs[4] (D = 42 789: 0.45 s of replicated factorisation against 35 ms of sharded work per step).
   * `rank` / `world_size` describe the communicator behind `allreduce` / `collective`. */
  int32_t distributed_solve;
  int32_t rank;
  int32_t world_size;
  /* reduce-scatter / all-gather of the distributed solve (optional: without it they are emulated with `allreduce`, same
   * results, 2-world x the bytes) */
  cba_collective_fn collective;
  void* collective_user;
  cba_solver_options solver;     /* scheduling options of the reduced solve (all zero = defaults) */
} cba_config;

/* OptimizationReport (LV/lm_optimizer.h:55-77) + what OptimizeJointly returns through pointers */
typedef struct {
  double initial_cost;     /* cost of the residual+Jacobian pass */
  double final_cost;       /* report.final_cost */
  double lambda;           /* optimizer.lambda() after the call (-> *final_lambda) */
  int32_t accepted;        /* report.num_iterations_performed (0/1) (-> *performed_an_iteration) */
  int32_t lm_attempts;
  int64_t n_residuals_valid;   /* residuals with cost >= 0 in the Jacobian pass (global) */
  int64_t n_jacobians_dropped; /* valid residuals added without Jacobian (joint_optimization.cc:373-376, 446-448) */
  double t_jac;            /* cost_and_jacobian_evaluation_time of the Jacobian pass [s]: device-side span (HIP events) */
  double t_solve;          /* solve_time [s], all LM attempts: device-side spans (HIP events) */
  double t_cost;           /* cost-only passes [s] */
  double t_accumulate;     /* part of t_jac spent in the JtJ accumulation kernel [s] */
  double t_gemm;           /* part of t_solve: Schur complement GEMM [s] */
  double t_factor;         /* part of t_solve: reduced-system factorisation [s] */
} cba_report;

typedef struct cba_problem cba_problem; /* opaque, device-resident */

/* Creates the engine's HIP streams for `device` (four per device, shared by every problem on it, alive until the
 * process exits).  Call it as early as possible: on ROCm 7.2 / MI355X a stream created before the process has
 * launched its first kernel runs the dominant GEMM ~15 % faster than one created later.  cba_create() calls it
 * itself, so this is an optimisation hook, not a requirement.  (No reference counterpart: the reference has no
 * device streams on this path.) */
int cba_prepare_device(int32_t device);

const char* cba_last_error(void);
const char* cba_version(void);

int cba_create(const cba_config* config, cba_problem** out);
void cba_destroy(cba_problem* p);

/* Dataset -> packed observations, sorted image-major, then camera, then feature order (the loop
 * order of JointOptimizationCostFunction::Compute, joint_optimization.cc:273-291).
 * xy: 2n fp32 PointFeature::xy; point_index: PointFeature::index; image_index: sequential index of
 * the (used) imageset on this rank; last_projection: 2n PointFeature::last_projection or NULL (zeros). */
int cba_set_observations(cba_problem* p, int64_t n, const float* xy, const int32_t* point_index,
                         const int32_t* image_index, const int32_t* camera_index,
                         const double* last_projection);

/* BAState -> device. rig_tr_global 7N (qw qx qy qz tx ty tz), camera_tr_rig 7C, points 3P,
 * grids[c]: 3G doubles row-major (index gx + gy*grid_w, Image<Vec3d>); non-central: direction grid
 * followed by the point grid. */
int cba_set_state(cba_problem* p, const double* rig_tr_global, const double* camera_tr_rig,
                  const double* points, const double* const* grids);
int cba_get_state(cba_problem* p, double* rig_tr_global, double* camera_tr_rig, double* points,
                  double* const* grids);
int cba_get_last_projection(cba_problem* p, double* out /* 2n */);

/* One optimizer.Optimize(max_iteration_count = 1) call of OptimizeJointly's loop
 * (joint_optimization.cc:916-925, LV/lm_optimizer.h:629-991): residual+Jacobian pass, JtJ/Jtr
 * accumulation, then up to max_lm_attempts x { Schur solve, state update, cost-only pass,
 * CostIsSmallerThan }.  init_lambda < 0 selects the automatic init_lambda_factor * mean(diag H). */
int cba_step(cba_problem* p, double init_lambda, int32_t max_lm_attempts, double init_lambda_factor,
             cba_report* report);

/* Compute<false> on the current state (VerifyCost, joint_optimization.cc:866-877; also the report
 * statistics F4).  cost_vector (n, host, may be NULL) gets the per-residual Huber cost or -1. */
int cba_cost(cba_problem* p, double* cost, int64_t* n_valid, double* cost_vector);

/* Finite-difference tasks of the last Jacobian pass whose iterate left the staged control patch AND found the gather-path
 * follow-up list full (the list holds a quarter of all tasks + 65 536): such a task loses its Jacobian like a failed
 * projection (the residual is kept, joint_optimization.cc:373-376) -- counted here so that it is never silent.  Expected: 0.
 * -1 on error.  (No reference counterpart.) */
int64_t cba_fd_redo_overflow(cba_problem* p);
/* Diagnostics: out[0] / out[1] = tasks of the last Jacobian pass that went to the gather-path follow-up list of the main /
 * side-stream finite-difference launch, out[2] = the overflow count above. */
int cba_debug_fd_redo_counts(cba_problem* p, int64_t out[3]);
/* ---- stateless model-level entry points (CameraModel API) ---- */
/* CameraModel::Project / ProjectWithInitialEstimate for n local points (APP/models/central_grid.h:79-97,
 * central_generic.cc:433-519, noncentral_generic.cc:156-264). init_pixels NULL = start at the centre
 * of the calibrated area. ok[i] = return value. */
int cba_project(const cba_camera* camera, const double* grid, int64_t n, const double* local_points,
                const double* init_pixels, double* pixels, uint8_t* ok, int32_t device);
/* CameraModel::Unproject / UnprojectWithJacobian (central_generic.h:97-105, central_generic.cc:521-549,
 * noncentral twins). lines: 6n (direction, origin); jacobians: 12n (6x2 row-major) or NULL. */
int cba_unproject(const cba_camera* camera, const double* grid, int64_t n, const double* pixels,
                  double* lines, double* jacobians, uint8_t* ok, int32_t device);

/* The same two calls on a DEVICE-RESIDENT camera model: the grid is uploaded once by cba_model_create and scratch buffers
 * are kept between calls, so a caller that projects point by point (CameraModel::Project in a loop, e.g.
 * APP/calibration_report.cc:101-148) pays one small launch per call instead of a grid upload and six allocations.
 * cba_model_set_grid replaces the grid (after an optimisation step changed the intrinsics). */
typedef struct cba_model cba_model;
int cba_model_create(const cba_camera* camera, const double* grid, int32_t device, cba_model** out);
void cba_model_destroy(cba_model* m);
int cba_model_set_grid(cba_model* m, const double* grid);
int cba_model_project(cba_model* m, int64_t n, const double* local_points, const double* init_pixels, double* pixels, uint8_t* ok);
int cba_model_unproject(cba_model* m, int64_t n, const double* pixels, double* lines, double* jacobians, uint8_t* ok);

/* ---- solver-level entry point ---- */
/* LMOptimizer::SolveWithSchurComplementDenseOffDiag (LV/lm_optimizer.h:1247-1369) on host arrays in
 * the reference's layout (symmetric parts: upper triangles only are read).  x = [block part; dense]. */
int cba_schur_solve(int32_t block_size, int32_t n_blocks, int32_t dense_dof, const double* block_diag_H,
                    const double* off_diag_H, const double* dense_H, const double* block_diag_b,
                    const double* dense_b, double* x, int32_t device);
/* the same with explicit scheduling options (NULL = defaults) */
int cba_schur_solve_opt(int32_t block_size, int32_t n_blocks, int32_t dense_dof, const double* block_diag_H,
                        const double* off_diag_H, const double* dense_H, const double* block_diag_b,
                        const double* dense_b, double* x, const cba_solver_options* options, int32_t device);

/* ---- grid-only LM ---- */
/* OptimizationReport of the fit + the optimizer's final lambda */
typedef struct {
  double initial_cost;          /* cost of the first residual+Jacobian pass */
  double final_cost;            /* report.final_cost */
  double lambda;                /* lambda after the last attempt */
  int32_t iterations_performed; /* report.num_iterations_performed */
  int32_t lm_attempts;          /* dense solves */
  double t_pass;                /* residual / Jacobian / cost passes incl. accumulation [s] */
  double t_solve;               /* dense solves [s] */
} cba_fit_report;
/* CentralGenericModel::FitToPixelDirectionsImpl (APP/models/central_generic.cc:551-568): LM over the 2G local grid
 * updates minimising sum_i 0.5 |normalize(spline(grid_point_i)) - direction_i|^2 (cost function :153-225, residual
 * and Jacobian :86-150, state update :65-80), LMOptimizer::Optimize(max_iteration_count, max_lm_attempts = 10,
 * init_lambda = -1, init_lambda_factor = 0.001f), dense solve (LV/lm_optimizer.h).  grid: 3G doubles, row-major,
 * unit directions, updated in place.  grid_points: 2n, in grid coordinates (PixelCornerConvToGridPoint already
 * applied, as FitToPixelDirections / FitToDenseModel do, central_generic.cc:413-431); they must address a full 4x4
 * patch (CBA_ERR_ARG otherwise, the reference CHECK-aborts).  Central-generic model only. */
int cba_fit_grid_to_directions(const cba_camera* camera, double* grid, int64_t n, const double* grid_points,
                               const double* directions, int32_t max_iteration_count, cba_fit_report* report, int32_t device);

/* ---- parity/debug access (read-only views of the last cba_step / cba_debug_* call) ---- */
enum {
  CBA_DUMP_COST_VECTOR = 1,      /* n doubles: Jacobian-pass residual costs (-1 invalid) */
  CBA_DUMP_PIXELS = 2,           /* 2n doubles: projected pixels of the Jacobian pass */
  CBA_DUMP_FLAGS = 3,            /* n bytes: bit0 valid, bit1 has_jacobian */
  CBA_DUMP_JACOBIANS = 4,        /* n * cba_jacobian_record_doubles() doubles */
  CBA_DUMP_BLOCK_DIAG_H = 5,     /* n_blocks*bs*bs, upper triangles, WITHOUT lambda */
  CBA_DUMP_BLOCK_DIAG_B = 6,
  CBA_DUMP_OFF_DIAG_H = 7,       /* (n_blocks*bs) x dense_dof row-major */
  CBA_DUMP_DENSE_H = 8,          /* dense_dof x dense_dof row-major, upper triangle, WITHOUT lambda */
  CBA_DUMP_DENSE_B = 9,
  CBA_DUMP_X = 10,               /* total_dof doubles: last update vector */
  CBA_DUMP_TEST_COST_VECTOR = 11 /* n doubles: cost vector of the last cost-only pass */
};
int cba_debug_dump(cba_problem* p, int32_t what, void* out, size_t bytes);
/* Tuning knob of the base projection (AddReprojectionResidual's ProjectWithInitialEstimate, joint_optimization.cc:325-343).
 * The one-lane-per-observation kernel hands an observation to the straggler kernel (16 lanes per observation, the same
 * 100 x 10-iteration procedure evaluated speculatively) after this many outer iterations; default 8.  0 sends every
 * observation there (the tests use it to compare the two kernels), >= 100 disables the hand-over.  Results do not depend
 * on the value. */
int cba_set_straggler_threshold(cba_problem* p, int32_t outer_iterations);
/* Scheduling knob of the finite-difference re-projections (joint_optimization.cc:357-372, APP/models/central_grid.h:187-245,
 * noncentral_generic.h:224-283): 0 = a workgroup takes a pool of tasks and every lane runs ONE damping attempt of its current
 * projection per loop trip, fetching the next task when it is done (a wavefront does not wait for its slowest projection);
 * 1 = one task per lane (the rounds 2-4 kernel); -1 (default) = automatic: pooled for the non-central model and for rigs (9 - 11 %
 * faster at BASELINE configs[3] / [2]), one task per lane for a single central-generic camera (4 % faster at configs[1]).
 * Both evaluate the same expressions in the same order for every task; validity / has-Jacobian flags are identical, Jacobian
 * entries agree to ~1e-12 of a record's largest entry (the compiler fuses a multiply-add differently in the two kernels: 0.004 %
 * of the entries differ, by an ulp of a pixel in one projection). */
int cba_set_fd_schedule(cba_problem* p, int32_t schedule);
/* Runs only the residual+Jacobian pass + accumulation on the current state (no solve). */
int cba_debug_accumulate(cba_problem* p, double* cost);
/* Solves the accumulated system for the given lambda (no state update); x via CBA_DUMP_X. */
int cba_debug_solve(cba_problem* p, double lambda);
/* state -= x for a caller-provided x (JointOptimizationState::operator-=, joint_optimization.cc:172-214);
 * the result becomes the current state. */
int cba_debug_apply_update(cba_problem* p, const double* x);

/* Host-only view of the static plan of the GRID-FIRST elimination order (cba_solver_options.elimination; camera_calibration_amd/csrc/
 * gridfirst_plan.h): no device is touched, the CPU tests replay the task list with numpy.  `what`: 0 = header (int32: G, Gf, n_rp,
 * n_border, n_fact, n_pad, nbg, nbf, ntc, chains, tasks, tasks of list 0, intervals, mask words, half-bandwidth, strips of camera 0),
 * 1 = row of F of every grid unknown in the engine's order (int32 x G), 2 = chains (int32 x 4: r0, r1, dep, 0), 3 = tasks (int32 x 4:
 * kind | intervals << 8, r, c, first interval; kinds: gridfirst_plan.h), 4 = K intervals (int32 x 2), 5 = row masks (uint64 x nbf x mask words), 6 = flop model
 * (double x 3: dataflow launch of the grid rows, border update, border factorisation), 16 + c = control point -> elimination rank of
 * camera c (int32 x grid_w grid_h).  Returns the number of bytes of the item (written if capacity_bytes suffices) or a negative
 * error code.  (No reference counterpart: LV/lm_optimizer.h:1247-1369 has one elimination order.) */
int64_t cba_gridfirst_plan_query(const cba_camera* cameras, int32_t n_cameras, int32_t n_images, int32_t n_points, int32_t strips,
                                 int32_t single_tile_tasks, int32_t what, void* out, int64_t capacity_bytes);

/* Elimination order the problem uses (cba_solver_options.elimination resolved): 1 = pose-first, 2 = grid-first; out[0..3] (optional,
 * may be NULL) = strips of camera 0, rows of the border system that is factored densely, rows of the grid part, pivot chains. */
int32_t cba_elimination_order(const cba_problem* p, int32_t out[4]);
int32_t cba_total_dof(const cba_problem* p);
int32_t cba_dense_dof(const cba_problem* p);
/* layout of one CBA_DUMP_JACOBIANS record: [res 2][weight 1][pose 2x6][rig 2x6][point 2x3][grid 2xK] */
int32_t cba_jacobian_record_doubles(const cba_problem* p);
int64_t cba_reduce_buffer_doubles(const cba_config* config);
/* device-side event timing of the dominant kernels of the last cba_step (bench roofline):
 * which: 0 = Schur GEMM launch, 1 = the whole factorisation (flops = its trailing updates), 2 = accumulation,
 * 3 = finite-difference projection kernel, 4 = the 128x128 GEMM launches inside the factorisation, kernel time only
 * (events on the stream of each launch). Returns seconds and flop/byte counts. */
int cba_kernel_stats(cba_problem* p, int32_t which, double* seconds, double* flops, double* bytes,
                     int32_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* CBA_H_ */
