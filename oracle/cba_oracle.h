/*
 * cba_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C fp64 restatement of the reference's JointOptimization hot path
 * (puzzlepaint/camera_calibration).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (camera_calibration_amd/, include/cba.h) never does.
 *
 * Parity pinning: checked against the reference's own golden vectors / known-answer
 * tests (see tests/test_oracle_golden.py): LMOptimizer.SchurComplement2 (H,b -> x),
 * BSpline.SlowFastAlgorithmConsistency, CentralGenericBSpline.ProjectUnproject,
 * NoncentralGenericBSpline.OrthogonalCameraProjectionAndUnprojection, HuberLoss identities,
 * TestOptimizeJointly convergence (cost <= 1e-6 * cameras), and -- round 2 -- against the reference's OWN CODE where
 * it compiles from its own source files (oracle/_ref, built by `make ref` from /root/reference against the Eigen /
 * libvis stand-ins in oracle/ref_shim): generic_models/src (Project / Unproject / UnprojectWithJacobian of both generic
 * models, the reference's self-test and its 17 x 13 calibrated camera), the generated Jacobian code
 * (joint_optimization_jacobians.h, *_generic_jacobians.cc), the local parametrisations, b_spline.h and HuberLoss;
 * tests/test_oracle_vs_ref.py holds the comparison (1e-14 class) and tests/golden/ref_vectors.npz the reference-computed
 * vectors.  Round 5: the reference's ENTIRE CPU path -- joint_optimization.cc (state, per-observation driver, OptimizeJointly),
 * central_generic.cc / noncentral_generic.cc, lm_optimizer.h and the outer-loop / report functions -- compiles whole against the
 * run-time-sized Eigen / Sophus stand-ins of oracle/ref_shim_lm (oracle/_ref/libcalibref_ba.so, libcalibref_lm.so,
 * libcalibref_f14.so) and runs next to this restatement: normal equations 1e-10, lambda 1e-11, identical decisions
 * (tests/test_oracle_vs_ref_whole_path.py).  Not the reference's or Eigen's code in those builds: Eigen's LDLT (this file's
 * term-by-term restatement) and the small fixed-size Eigen / Sophus operations of the stand-ins --
 * see DESIGN.md "Oracle".
 *
 * Citations are relative to /root/reference:
 *   APP = applications/camera_calibration/src/camera_calibration, LV = libvis/src/libvis
 */
#ifndef CBA_ORACLE_H_
#define CBA_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_CENTRAL_GENERIC = 0, ORC_NONCENTRAL_GENERIC = 1 };

/* One camera: calibrated rectangle + grid resolution (APP/models/camera_model.h:42-204). */
typedef struct {
  int32_t model_type;
  int32_t width, height;
  int32_t calib_min_x, calib_min_y, calib_max_x, calib_max_y;
  int32_t grid_w, grid_h;
} orc_camera;

/* Problem description: arrays are owned by the caller (numpy). */
typedef struct {
  int32_t n_cameras, n_images, n_points;
  int64_t n_obs;
  const orc_camera* cams;
  /* observations sorted image-major, then camera, then feature order
   * (the reference's loop order, APP/bundle_adjustment/joint_optimization.cc:273-291) */
  const float* obs_xy;       /* 2*n_obs, PointFeature::xy (fp32, APP/dataset.h:67) */
  const int32_t* obs_point;  /* PointFeature::index */
  const int32_t* obs_image;  /* sequential (used-image) index */
  const int32_t* obs_camera;
  double* last_projection;   /* 2*n_obs, PointFeature::last_projection (in/out) */
  double fd_delta;           /* numerical_diff_delta */
  int32_t localize_only;
  int32_t eliminate_points;
} orc_problem;

/* State: rig_tr_global 7N (qw qx qy qz tx ty tz), camera_tr_rig 7C, points 3P,
 * grids[c]: central 3G doubles (index gx + gy*gw); noncentral: direction grid 3G then point grid 3G. */
typedef struct {
  double* rig_tr_global;
  double* camera_tr_rig;
  double* points;
  double** grids;
} orc_state;

/* Normal equations in the reference's dense layout (LV/lm_optimizer.h:660-685).
 * All symmetric parts hold only their upper triangle. */
typedef struct {
  int32_t block_size, n_blocks, dense_dof;
  double* block_diag_H; /* n_blocks * bs*bs, row-major */
  double* off_diag_H;   /* (n_blocks*bs) x dense_dof, row-major */
  double* dense_H;      /* dense_dof x dense_dof, row-major */
  double* block_diag_b; /* n_blocks*bs */
  double* dense_b;      /* dense_dof */
} orc_system;

/* Per-observation debug record (kernel-level parity). */
typedef struct {
  int32_t valid;        /* projection succeeded */
  int32_t has_jacobian; /* 0 = residual added without Jacobian */
  double pixel[2];
  double residual[2];
  double cost;          /* Huber cost or -1 */
  double weight;
  double pose_jac[12];  /* 2x6 row-major */
  double rig_jac[12];   /* 2x6 */
  double point_jac[6];  /* 2x3 */
  int32_t grid_indices[80];   /* local to the camera's intrinsics block */
  double grid_jac[160];       /* 2xK row-major, K = 32 or 80 */
} orc_obs_record;

/* ---- host threads ----
 * n <= 0 selects all hardware threads.  Default 1 (the reference path is single-threaded).  Results do not depend on
 * the thread count (bit-identical): threads split independent per-observation work and cache tiles only. */
void orc_set_num_threads(int n);
/* developer aid: per-observation count of B-spline evaluations of the base projection in the following passes (NULL = off) */
void orc_debug_set_eval_trace(int32_t* per_observation);
int orc_get_num_threads(void);
void orc_debug_set_state_trace(double* buf300);   /* single-threaded debugging only */
int orc_debug_state_trace_count(void);
int orc_hardware_threads(void);

/* ---- model level ---- */
void orc_bspline_surface(const double* ctrl, int w, int h, int dim, double x, double y, double* out);
void orc_bspline_surface_slow(const double* ctrl, int w, int h, int dim, double x, double y, double* out);
int orc_unproject(const orc_camera* cam, const double* grid, double x, double y, double* line6);
int orc_unproject_with_jacobian(const orc_camera* cam, const double* grid, double x, double y,
                                double* line6, double* jac12);
int orc_project_with_initial_estimate(const orc_camera* cam, const double* grid,
                                      const double* local_point, double* pixel);
int orc_project_direction_with_initial_estimate(const orc_camera* cam, const double* grid, const double* direction, double* pixel);
int orc_project(const orc_camera* cam, const double* grid, const double* local_point, double* pixel);
void orc_grid_point_to_pixel(const orc_camera* cam, double gx, double gy, double* px);
void orc_pixel_to_grid_point(const orc_camera* cam, double x, double y, double* gp);

/* ---- parametrisations / small math ---- */
void orc_tangents(const double* dir, double* t1, double* t2);
void orc_apply_quaternion_update(const double* q_wxyz, const double* update3, double* out_wxyz);
void orc_se3_mul(const double* a7, const double* b7, double* out7);
void orc_se3_exp(const double* tangent6, double* out7);
double orc_huber_cost_sq(double sq, double k);
double orc_huber_weight_sq(double sq, double k);
void orc_compute_jacobian(const double* q_wxyz, const double* p, double* jac30);
void orc_compute_rig_jacobian(const double* cq, const double* p, const double* rq, const double* rt,
                              double* jac51);

/* ---- problem level ---- */
int32_t orc_intrinsics_param_count(const orc_camera* cam);
/* test hook: ProjectionJacobianWrtIntrinsics (M5 / N3) on one point; J is 2 x K row-major, K = 32 / 80; returns 1 ok, 0 a
 * perturbed projection failed, -1 the 4 x 4 patch leaves the grid (a CHECK abort in the reference) */
int orc_debug_projection_jacobian_wrt_intrinsics(const orc_camera* cam, const double* grid, const double* local_point,
                                                 const double* pixel, double delta, int32_t* indices, double* J);
int32_t orc_dense_dof(const orc_problem* pb);
int32_t orc_total_dof(const orc_problem* pb);
/* Compute<false>: cost-only pass.  cost_vec has n_obs entries (-1 = invalid). */
double orc_cost_pass(const orc_problem* pb, const orc_state* st, double* cost_vec);
double orc_cost_pass_records(const orc_problem* pb, const orc_state* st, double* cost_vec, orc_obs_record* records);
/* Compute<true>: residual+Jacobian pass accumulating into sys (zeroed first).
 * records may be NULL; image range [img_begin,img_end) restricts the pass (cpu_baseline sampling). */
double orc_jacobian_pass(const orc_problem* pb, const orc_state* st, orc_system* sys,
                         double* cost_vec, orc_obs_record* records,
                         int32_t img_begin, int32_t img_end);
/* SolveWithSchurComplementDenseOffDiag; adds nothing to the diagonal itself. x = [block part; dense part]. */
void orc_schur_solve(const orc_system* sys, double* x);
/* Pivoted LDLT solve of a dense symmetric system given by its upper triangle (row-major). */
void orc_ldlt_solve_upper(const double* A_upper, int n, const double* b, double* x);
void orc_ldlt_solve_upper_unblocked(const double* A_upper, int n, const double* b, double* x);
/* state -= x  (JointOptimizationState::operator-=). */
void orc_apply_update(const orc_problem* pb, const orc_state* st_in, const double* x, orc_state* st_out);
/* One OptimizeJointly call (max_iteration_count outer iterations). Returns final cost. */
double orc_optimize_jointly(orc_problem* pb, orc_state* st, int max_iteration_count,
                            double init_lambda, double* final_lambda, int32_t* performed_an_iteration,
                            double* timings3 /* t_jac, t_solve, t_cost; may be NULL */,
                            int32_t* lm_attempts /* may be NULL */);

/* ---- grid-only LM (CentralGenericModel::FitToPixelDirections) ---- */
double orc_fit_grid_pass(int32_t gw, int32_t gh, const double* grid, int64_t n, const double* grid_points,
                         const double* directions, double* H /*dof x dof or NULL*/, double* b, double* cost_vec);
void orc_fit_grid_apply_update(int32_t gw, int32_t gh, const double* grid_in, const double* x, double* grid_out);
void orc_fit_grid_to_points(int32_t gw, int32_t gh, double* grid, int64_t n, const double* grid_points,
                            const double* directions, int32_t max_iteration_count, double* report4);

#ifdef __cplusplus
}
#endif
#endif
