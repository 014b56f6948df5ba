/*
 * cba_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See cba_oracle.h.
 *
 * Each function cites the reference file:line it restates
 * (APP = applications/camera_calibration/src/camera_calibration, LV = libvis/src/libvis).
 * Eigen 3.3.7 / Sophus arithmetic that the reference calls into is restated from its
 * published algorithms (quaternion product, toRotationMatrix, _transformVector, LDLT with
 * diagonal pivoting); the call sites are listed next to each restatement.
 */
#include "cba_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXK 80

/* Number of host threads the oracle may use (default 1: the reference path is single-threaded).
 * Threads only split work whose per-entry arithmetic and summation order do not depend on the split: results are
 * bit-identical for every thread count (tests/test_oracle_golden.py::test_threaded_oracle_is_bit_identical). */
static int g_threads = 1;
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n <= 0) n = omp_get_num_procs();
  g_threads = n < 1 ? 1 : n;
#else
  (void)n; g_threads = 1;
#endif
}
int orc_get_num_threads(void) { return g_threads; }
int orc_hardware_threads(void) {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * small vector helpers
 * ---------------------------------------------------------------------------------------- */
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
/* Eigen normalized(): v / sqrt(v.v) when the squared norm is > 0. */
static inline void normalize3(const double* v, double* o) {
  double z = dot3(v, v);
  if (z > 0) {
    double n = sqrt(z);
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
  } else {
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
  }
}
static double now_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------
 * M1: pixel <-> grid coordinates (APP/models/central_grid.h:127-161, noncentral_generic.h:152-182).
 * The literals 1.f / 3.f are float but exactly representable; the *Scale* helpers divide two
 * floats, i.e. the scale factor is rounded to fp32 (restated as such).
 * ---------------------------------------------------------------------------------------- */
void orc_pixel_to_grid_point(const orc_camera* cam, double x, double y, double* gp) {
  gp[0] = 1.0 + (double)((float)cam->grid_w - 3.f) * (x - cam->calib_min_x) / (cam->calib_max_x + 1 - cam->calib_min_x);
  gp[1] = 1.0 + (double)((float)cam->grid_h - 3.f) * (y - cam->calib_min_y) / (cam->calib_max_y + 1 - cam->calib_min_y);
}
void orc_grid_point_to_pixel(const orc_camera* cam, double gx, double gy, double* px) {
  if (cam->model_type == ORC_CENTRAL_GENERIC && gx == floor(gx) && gy == floor(gy)) {
    /* central overload takes int x,y: the whole expression is evaluated in float (central_grid.h:127-131) */
    float fx = (float)cam->calib_min_x + (((float)gx - 1.f) / ((float)cam->grid_w - 3.f)) * (float)(cam->calib_max_x + 1 - cam->calib_min_x);
    float fy = (float)cam->calib_min_y + (((float)gy - 1.f) / ((float)cam->grid_h - 3.f)) * (float)(cam->calib_max_y + 1 - cam->calib_min_y);
    px[0] = fx; px[1] = fy;
  } else {
    /* non-central overload takes double x,y (noncentral_generic.h:152-156) */
    px[0] = cam->calib_min_x + ((gx - 1.f) / (double)((float)cam->grid_w - 3.f)) * (cam->calib_max_x + 1 - cam->calib_min_x);
    px[1] = cam->calib_min_y + ((gy - 1.f) / (double)((float)cam->grid_h - 3.f)) * (cam->calib_max_y + 1 - cam->calib_min_y);
  }
}
static inline double pixel_scale_to_grid_scale_x(const orc_camera* cam, double len) {
  float f = ((float)cam->grid_w - 3.f) / (float)(cam->calib_max_x + 1 - cam->calib_min_x);
  return len * f;
}
static inline double pixel_scale_to_grid_scale_y(const orc_camera* cam, double len) {
  float f = ((float)cam->grid_h - 3.f) / (float)(cam->calib_max_y + 1 - cam->calib_min_y);
  return len * f;
}
/* APP/models/camera_model.h:159-162 */
static inline int in_calibrated_area(const orc_camera* cam, double x, double y) {
  return x >= cam->calib_min_x && y >= cam->calib_min_y && x < cam->calib_max_x + 1 && y < cam->calib_max_y + 1;
}
static inline void center_of_calibrated_area(const orc_camera* cam, double* px) {
  px[0] = 0.5 * (cam->calib_min_x + cam->calib_max_x + 1);
  px[1] = 0.5 * (cam->calib_min_y + cam->calib_max_y + 1);
}

/* ------------------------------------------------------------------------------------------
 * M2: uniform cubic B-spline (APP/b_spline.h:33-104, 168-186)
 * ---------------------------------------------------------------------------------------- */
static void cubic_weights_exact(double f, double* w) {
  /* b_spline.h:49-60; f in [3,4) */
  double fd = f - 3;
  w[3] = 1. / 6. * fd * fd * fd;
  w[2] = -1. / 2. * f * f * f + 5 * f * f - 16 * f + 50. / 3.;
  w[1] = 1. / 2. * f * f * f - 11. / 2. * f * f + (39. / 2.) * f - 131. / 6.;
  w[0] = -1. / 6. * (f - 4) * (f - 4) * (f - 4);
}
void orc_bspline_surface(const double* ctrl, int w, int h, int dim, double x, double y, double* out) {
  (void)h;
  x += 2; y += 2;
  int ix = (int)x, iy = (int)y;
  double wx[4], wy[4];
  cubic_weights_exact(x - (ix - 3), wx);
  cubic_weights_exact(y - (iy - 3), wy);
  double rows[4][8];
  for (int r = 0; r < 4; ++r) {
    int ky = iy - 3 + r;
    for (int d = 0; d < dim; ++d) {
      const double* base = ctrl + ((size_t)ky * w + (ix - 3)) * dim + d;
      rows[r][d] = wx[0] * base[0] + wx[1] * base[dim] + wx[2] * base[2 * dim] + wx[3] * base[3 * dim];
    }
  }
  for (int d = 0; d < dim; ++d)
    out[d] = wy[0] * rows[0][d] + wy[1] * rows[1][d] + wy[2] * rows[2][d] + wy[3] * rows[3][d];
}
/* Cox-de Boor definition, b_spline.h:33-43 */
static double basis_fn(int i, int order, double x) {
  if (order == 0) return (x >= i && x < i + 1) ? 1 : 0;
  return (x - i) / order * basis_fn(i, order - 1, x) + (i + order + 1 - x) / order * basis_fn(i + 1, order - 1, x);
}
void orc_bspline_surface_slow(const double* ctrl, int w, int h, int dim, double x, double y, double* out) {
  (void)h;
  x += 2; y += 2;
  int ix = (int)x, iy = (int)y;
  for (int d = 0; d < dim; ++d) out[d] = 0;
  for (int ky = iy - 3; ky <= iy; ++ky)
    for (int kx = ix - 3; kx <= ix; ++kx) {
      double b = basis_fn(kx, 3, x) * basis_fn(ky, 3, y);
      for (int d = 0; d < dim; ++d) out[d] += ctrl[((size_t)ky * w + kx) * dim + d] * b;
    }
}

/* ------------------------------------------------------------------------------------------
 * P1/P2: tangents and local direction / line updates
 * (APP/local_parametrizations/line_parametrization.h:54-60, 107-120; direction_parametrization.h:45-55)
 * ---------------------------------------------------------------------------------------- */
void orc_tangents(const double* dir, double* t1, double* t2) {
  double axis[3] = {0, 0, 0};
  if (fabs(dir[0]) > (double)0.9f) axis[1] = 1; else axis[0] = 1;
  double c[3];
  cross3(dir, axis, c);
  normalize3(c, t1);
  cross3(dir, t1, t2);
}
static void apply_local_update_to_direction(double* dir, const double* t1, const double* t2, double o1, double o2) {
  double v[3];
  for (int i = 0; i < 3; ++i) v[i] = dir[i] + o1 * t1[i] + o2 * t2[i];
  normalize3(v, dir);
}
static void apply_local_update_to_line(double* origin, double* dir, const double* t1, const double* t2,
                                       double o1, double o2, double o3, double o4, double o5) {
  for (int i = 0; i < 3; ++i) origin[i] = origin[i] + o3 * t1[i] + o4 * t2[i] + o5 * dir[i];
  apply_local_update_to_direction(dir, t1, t2, o1, o2);
}

/* ------------------------------------------------------------------------------------------
 * M3 / N1: unprojection with Jacobian wrt. the pixel.
 * Restates CentralGenericBSpline_Unproject_ComputeResidualAndJacobian
 * (APP/models/central_generic_jacobians.cc:320-448) and the non-central twin
 * (noncentral_generic_jacobians.cc:31-205): same factorisation of the cubic weights and the
 * same 15-digit decimal literals; the non-central version normalises with sqrtf.
 * patch: 16 control vectors [row][col][dim], dim = 3 (central) or 6 (dir, origin).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  double a5, a3, b, c, d8, d7;   /* value weights: a = a5*a3 (applied as a3*(p*a5)), d = d8*d7 */
  double da_s, db, dc, dd_s, a84, d85;  /* derivative pieces */
} axis_weights;

static void axis_weights_generated(double f, axis_weights* w) {
  double t4 = 0.166666666666667 * f;
  w->a5 = -t4 + 0.666666666666667;
  w->a3 = (f - 4) * (f - 4);
  w->d8 = t4 - 0.5;
  w->d7 = (f - 3) * (f - 3);
  double f2 = f * f;
  double h = 0.5 * f * f2;
  w->b = 19.5 * f - 5.5 * f2 + h - 21.8333333333333;
  w->c = -16 * f + 5 * f2 - h + 16.6666666666667;
  double t80 = 1.5 * f2;
  w->da_s = 0.166666666666667 * w->a3;
  w->dd_s = 0.166666666666667 * w->d7;
  w->db = -11.0 * f + t80 + 19.5;
  w->dc = 10 * f - t80 - 16;
  w->a84 = 2 * f - 8;
  w->d85 = 2 * f - 6;
}

/* value and d/dfrac_x, d/dfrac_y of the (un-normalised) spline for one scalar channel */
static void spline_channel(const double* p /*16, stride given*/, int stride, const axis_weights* wx,
                           const axis_weights* wy, double* val, double* dval_dx, double* dval_dy) {
  double R[4], dR[4];
  for (int r = 0; r < 4; ++r) {
    double p0 = p[(r * 4 + 0) * stride], p1 = p[(r * 4 + 1) * stride];
    double p2 = p[(r * 4 + 2) * stride], p3 = p[(r * 4 + 3) * stride];
    double p0a = p0 * wx->a5;
    double p3d = p3 * wx->d8;
    R[r] = p1 * wx->b + p2 * wx->c + wx->a3 * p0a + wx->d7 * p3d;
    dR[r] = -p0 * wx->da_s + p1 * wx->db + p2 * wx->dc + p3 * wx->dd_s + p0a * wx->a84 + p3d * wx->d85;
  }
  double r0s = R[0] * wy->a3;   /* term15 */
  double r3s = wy->d7 * R[3];   /* term22 */
  *val = wy->a5 * r0s + wy->d8 * r3s + wy->c * R[2] + wy->b * R[1];
  double wya = wy->a5 * wy->a3; /* term77 */
  double wyd = wy->d8 * wy->d7; /* term86 */
  *dval_dx = wy->c * dR[2] + wy->b * dR[1] + wya * dR[0] + wyd * dR[3];
  double t101 = wy->a5 * wy->a84;
  double t103 = wy->d8 * wy->d85;
  *dval_dy = t101 * R[0] + t103 * R[3] - 0.166666666666667 * r0s + 0.166666666666667 * r3s + R[2] * wy->dc + R[1] * wy->db;
}

static void gather_patch(const orc_camera* cam, const double* grid, int ix, int iy, double* patch /*16*6*/) {
  int dim = cam->model_type == ORC_CENTRAL_GENERIC ? 3 : 6;
  size_t G = (size_t)cam->grid_w * cam->grid_h;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      size_t gi = (size_t)(iy - 3 + r) * cam->grid_w + (ix - 3 + c);
      double* o = patch + (r * 4 + c) * dim;
      o[0] = grid[3 * gi + 0]; o[1] = grid[3 * gi + 1]; o[2] = grid[3 * gi + 2];
      if (dim == 6) {
        o[3] = grid[3 * G + 3 * gi + 0]; o[4] = grid[3 * G + 3 * gi + 1]; o[5] = grid[3 * G + 3 * gi + 2];
      }
    }
}

/* line6 = [direction(3), origin(3)] (origin = 0 for central). jac12 = 6x2 row-major, direction rows first.
 * APP/models/central_generic.cc:521-549, noncentral_generic.cc:266-293 */
int orc_unproject_with_jacobian(const orc_camera* cam, const double* grid, double x, double y,
                                double* line6, double* jac12) {
  if (!in_calibrated_area(cam, x, y)) return 0;
  double gp[2];
  orc_pixel_to_grid_point(cam, x, y, gp);
  gp[0] += 2; gp[1] += 2;
  int ix = (int)floor(gp[0]), iy = (int)floor(gp[1]);
  double frac_x = gp[0] - (ix - 3), frac_y = gp[1] - (iy - 3);
  int dim = cam->model_type == ORC_CENTRAL_GENERIC ? 3 : 6;
  double patch[16 * 6];
  gather_patch(cam, grid, ix, iy, patch);
  axis_weights wx, wy;
  axis_weights_generated(frac_x, &wx);
  axis_weights_generated(frac_y, &wy);
  double v[6], dvx[6], dvy[6];
  for (int d = 0; d < dim; ++d) spline_channel(patch + d, dim, &wx, &wy, &v[d], &dvx[d], &dvy[d]);
  double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  double inv, inv3;
  if (cam->model_type == ORC_CENTRAL_GENERIC) {
    inv = 1. / sqrt(sq);                /* central_generic_jacobians.cc:399 */
    double invb = 1. / sqrt(sq);        /* :411-412 */
    inv3 = invb * invb * invb;
  } else {
    /* noncentral_generic_jacobians.cc:110 is `1 / sqrtf(term75)`: int / float, i.e. the DIVISION is rounded to fp32 as
     * well (found by running the reference's code, oracle/_ref, tests/test_oracle_vs_ref.py) */
    inv = (double)(1.0f / sqrtf((float)sq));
    double tmp = (double)sqrtf((float)sq); /* :158-159 */
    inv3 = 1 / (tmp * tmp * tmp);
  }
  /* term94 = 2*inv3*(dvx.(v/2)), term113/149 likewise for y */
  double sx = 2 * inv3 * (dvx[0] * (0.5 * v[0]) + dvx[1] * (0.5 * v[1]) + dvx[2] * (0.5 * v[2]));
  double sy = inv3 * ((0.5 * v[0]) * (2 * dvy[0]) + (0.5 * v[1]) * (2 * dvy[1]) + (0.5 * v[2]) * (2 * dvy[2]));
  for (int d = 0; d < 3; ++d) {
    line6[d] = v[d] * inv;
    jac12[d * 2 + 0] = -v[d] * sx + inv * dvx[d];
    jac12[d * 2 + 1] = -sy * v[d] + inv * dvy[d];
  }
  for (int d = 3; d < 6; ++d) {
    if (dim == 6) {
      line6[d] = v[d];
      jac12[d * 2 + 0] = dvx[d];
      jac12[d * 2 + 1] = dvy[d];
    } else {
      line6[d] = 0; jac12[d * 2 + 0] = 0; jac12[d * 2 + 1] = 0;
    }
  }
  for (int d = 0; d < 6; ++d) {
    jac12[d * 2 + 0] = pixel_scale_to_grid_scale_x(cam, jac12[d * 2 + 0]);
    jac12[d * 2 + 1] = pixel_scale_to_grid_scale_y(cam, jac12[d * 2 + 1]);
  }
  return 1;
}

/* Unproject (APP/models/central_generic.h:97-105, noncentral_generic.h:100-115): exact-fraction
 * weights of b_spline.h, then normalise the direction. */
int orc_unproject(const orc_camera* cam, const double* grid, double x, double y, double* line6) {
  if (!in_calibrated_area(cam, x, y)) return 0;
  double gp[2];
  orc_pixel_to_grid_point(cam, x, y, gp);
  size_t G = (size_t)cam->grid_w * cam->grid_h;
  double d[3];
  orc_bspline_surface(grid, cam->grid_w, cam->grid_h, 3, gp[0], gp[1], d);
  normalize3(d, line6);
  if (cam->model_type == ORC_NONCENTRAL_GENERIC) {
    orc_bspline_surface(grid + 3 * G, cam->grid_w, cam->grid_h, 3, gp[0], gp[1], line6 + 3);
  } else {
    line6[3] = line6[4] = line6[5] = 0;
  }
  return 1;
}

/* TangentsJacobianWrtLineDirection, APP/local_parametrizations/line_parametrization.h:62-105. 6x3 row-major */
static void tangents_jacobian_wrt_direction(const double* d, double* J) {
  memset(J, 0, 18 * sizeof(double));
  if (fabs(d[0]) > (double)0.9f) {
    double t0 = d[0] * d[0], t1 = d[2] * d[2], t2 = t0 + t1;
    double t7 = 1. / sqrt(t2), t3 = t7 * t7 * t7;
    double t4 = d[0] * d[2] * t3, t5 = t0 * t3, t6 = t1 * t3, t8 = d[0] * t7, t9 = -d[1] * t4, t10 = d[2] * t7;
    J[0] = t4; J[2] = -t5;
    J[6] = t6; J[8] = -t4;
    J[9] = d[1] * t6; J[10] = t8; J[11] = t9;
    J[12] = -t8; J[14] = -t10;
    J[15] = t9; J[16] = t10; J[17] = d[1] * t5;
  } else {
    double t0 = d[1] * d[1], t1 = d[2] * d[2], t2 = t0 + t1;
    double t7 = 1. / sqrt(t2), t3 = t7 * t7 * t7;
    double t4 = d[1] * d[2] * t3, t5 = t0 * t3, t6 = t1 * t3, t8 = d[1] * t7, t9 = d[2] * t7, t10 = -d[0] * t4;
    J[4] = -t4; J[5] = t5;
    J[7] = -t6; J[8] = t4;
    J[10] = -t8; J[11] = -t9;
    J[12] = t8; J[13] = d[0] * t6; J[14] = t10;
    J[15] = t9; J[16] = t10; J[17] = d[0] * t5;
  }
}

/* M4 / N2: iterative projection.
 * central: CentralGenericModel::ProjectDirectionWithInitialEstimate (APP/models/central_generic.cc:433-519),
 *          reached through ProjectWithInitialEstimate = normalise + call (central_grid.h:86-88);
 * non-central: NoncentralGenericModel::ProjectWithInitialEstimate (noncentral_generic.cc:156-264).
 * target: unit direction (central) or the local point (non-central). Returns 1 on convergence,
 * 0 on failure, -1 where the reference CHECK()-aborts. */
/* developer aid (tools/projection_iteration_histogram.py): B-spline evaluations spent by this thread's projections */
static _Thread_local long g_eval_count = 0;
static int32_t* g_eval_trace = NULL;            /* per observation: evaluations of its base projection */
void orc_debug_set_eval_trace(int32_t* per_observation) { g_eval_trace = per_observation; }

/* debug (tools/projection_orbits.py): the loop state (pixel, lambda) at the top of every outer iteration of ONE projection */
static double* g_state_trace = NULL;
static int g_state_trace_n = 0;
void orc_debug_set_state_trace(double* buf300) { g_state_trace = buf300; g_state_trace_n = 0; }
int orc_debug_state_trace_count(void) { return g_state_trace_n; }
static int project_target(const orc_camera* cam, const double* grid, const double* target, double* result) {
  const double kEpsilon = 1e-12;
  double lambda = -1;
  for (int it = 0; it < 100; ++it) {
    double line[6], J[12];
    ++g_eval_count;
    if (g_state_trace && g_state_trace_n < 100) {
      g_state_trace[3 * g_state_trace_n] = result[0]; g_state_trace[3 * g_state_trace_n + 1] = result[1];
      g_state_trace[3 * g_state_trace_n + 2] = lambda; ++g_state_trace_n;
    }
    if (!orc_unproject_with_jacobian(cam, grid, result[0], result[1], line, J)) return -1;
    double cost, H00, H01, H11, b0, b1;
    if (cam->model_type == ORC_CENTRAL_GENERIC) {
      double dx = line[0] - target[0], dy = line[1] - target[1], dz = line[2] - target[2];
      cost = dx * dx + dy * dy + dz * dz;
      H00 = J[0] * J[0] + J[2] * J[2] + J[4] * J[4];
      H01 = J[0] * J[1] + J[2] * J[3] + J[4] * J[5];
      H11 = J[1] * J[1] + J[3] * J[3] + J[5] * J[5];
      b0 = dx * J[0] + dy * J[2] + dz * J[4];
      b1 = dx * J[1] + dy * J[3] + dz * J[5];
    } else {
      double t1[3], t2[3];
      orc_tangents(line, t1, t2);
      double pto[3] = {line[3] - target[0], line[4] - target[1], line[5] - target[2]};
      double d1 = dot3(t1, pto), d2 = dot3(t2, pto);
      double TJ[18];
      tangents_jacobian_wrt_direction(line, TJ);
      /* t1_t2_origin_wrt_xy [9x2] */
      double M[18];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 2; ++c)
          M[r * 2 + c] = TJ[r * 3 + 0] * J[0 * 2 + c] + TJ[r * 3 + 1] * J[1 * 2 + c] + TJ[r * 3 + 2] * J[2 * 2 + c];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 2; ++c) M[(6 + r) * 2 + c] = J[(3 + r) * 2 + c];
      /* residuals_wrt_xy = d_wrt_t1_t2_origin [2x9] * M */
      double R[4];
      for (int c = 0; c < 2; ++c) {
        R[0 * 2 + c] = pto[0] * M[0 * 2 + c] + pto[1] * M[1 * 2 + c] + pto[2] * M[2 * 2 + c] + 0 * M[3 * 2 + c] + 0 * M[4 * 2 + c] + 0 * M[5 * 2 + c] +
                       t1[0] * M[6 * 2 + c] + t1[1] * M[7 * 2 + c] + t1[2] * M[8 * 2 + c];
        R[1 * 2 + c] = 0 * M[0 * 2 + c] + 0 * M[1 * 2 + c] + 0 * M[2 * 2 + c] + pto[0] * M[3 * 2 + c] + pto[1] * M[4 * 2 + c] + pto[2] * M[5 * 2 + c] +
                       t2[0] * M[6 * 2 + c] + t2[1] * M[7 * 2 + c] + t2[2] * M[8 * 2 + c];
      }
      cost = d1 * d1 + d2 * d2;
      H00 = R[0] * R[0] + R[2] * R[2];
      H01 = R[0] * R[1] + R[2] * R[3];
      H11 = R[1] * R[1] + R[3] * R[3];
      b0 = d1 * R[0] + d2 * R[2];
      b1 = d1 * R[1] + d2 * R[3];
    }
    if (lambda < 0) lambda = 0.01 * 0.5 * (H00 + H11);
    int accepted = 0;
    for (int lm = 0; lm < 10; ++lm) {
      double H00lm = H00 + lambda, H11lm = H11 + lambda;
      double x1 = (b1 - H01 / H00lm * b0) / (H11lm - H01 * H01 / H00lm);
      double x0 = (b0 - H01 * x1) / H00lm;
      /* std::max<double>(min, std::min(max + 0.999, v)):  std::min(a,b) = (b<a)?b:a, std::max(a,b) = (a<b)?b:a */
      double tx, ty;
      {
        double a = cam->calib_max_x + 0.999, b = result[0] - x0;
        double m = (b < a) ? b : a;
        tx = ((double)cam->calib_min_x < m) ? m : (double)cam->calib_min_x;
        a = cam->calib_max_y + 0.999; b = result[1] - x1;
        m = (b < a) ? b : a;
        ty = ((double)cam->calib_min_y < m) ? m : (double)cam->calib_min_y;
      }
      double test_cost = INFINITY;
      double tl[6];
      ++g_eval_count;
      if (orc_unproject(cam, grid, tx, ty, tl)) {
        if (cam->model_type == ORC_CENTRAL_GENERIC) {
          double ex = tl[0] - target[0], ey = tl[1] - target[1], ez = tl[2] - target[2];
          test_cost = ex * ex + ey * ey + ez * ez;
        } else {
          double t1[3], t2[3];
          orc_tangents(tl, t1, t2);
          double pto[3] = {tl[3] - target[0], tl[4] - target[1], tl[5] - target[2]};
          double e1 = dot3(t1, pto), e2 = dot3(t2, pto);
          test_cost = e1 * e1 + e2 * e2;
        }
      }
      if (test_cost < cost) {
        lambda *= 0.5;
        result[0] = tx; result[1] = ty;
        accepted = 1;
        break;
      } else {
        lambda *= 2;
      }
    }
    if (!accepted) return cost < kEpsilon;
    if (cost < kEpsilon) return 1;
  }
  return 0;
}

int orc_project_with_initial_estimate(const orc_camera* cam, const double* grid,
                                      const double* local_point, double* pixel) {
  int r;
  if (cam->model_type == ORC_CENTRAL_GENERIC) {
    double dir[3];
    normalize3(local_point, dir);
    r = project_target(cam, grid, dir, pixel);
  } else {
    r = project_target(cam, grid, local_point, pixel);
  }
  return r > 0;
}
/* CentralGenericModel::ProjectDirectionWithInitialEstimate (APP/models/central_generic.cc:137-224): the direction is taken as given
 * (CentralGridModel::Project normalises before the call; a second normalisation could move it by an ulp) */
int orc_project_direction_with_initial_estimate(const orc_camera* cam, const double* grid, const double* direction, double* pixel) {
  return project_target(cam, grid, direction, pixel) > 0;
}
/* CameraModel::Project: start from the centre of the calibrated area (central_grid.h:79-97) */
int orc_project(const orc_camera* cam, const double* grid, const double* local_point, double* pixel) {
  center_of_calibrated_area(cam, pixel);
  return orc_project_with_initial_estimate(cam, grid, local_point, pixel);
}

/* ------------------------------------------------------------------------------------------
 * Sophus / Eigen quaternion semantics (external: Eigen 3.3.7 Quaternion.h; vendored Sophus
 * libvis/third_party/sophus/sophus/so3.hpp:159-168, 215-232, se3.hpp:183-207).
 * Poses are stored (qw,qx,qy,qz,tx,ty,tz).
 * ---------------------------------------------------------------------------------------- */
static void quat_mul(const double* a, const double* b, double* o) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static void quat_normalize(double* q) { /* so3.hpp:159-168 */
  double len = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= len; q[1] /= len; q[2] /= len; q[3] /= len;
}
static void quat_rotate(const double* q, const double* v, double* o) { /* Eigen _transformVector */
  double uv[3];
  cross3(q + 1, v, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  double c[3];
  cross3(q + 1, uv, c);
  o[0] = v[0] + q[0] * uv[0] + c[0];
  o[1] = v[1] + q[0] * uv[1] + c[1];
  o[2] = v[2] + q[0] * uv[2] + c[2];
}
static void quat_to_matrix(const double* q, double* R) { /* Eigen toRotationMatrix */
  double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
  double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
  double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
  double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
void orc_se3_mul(const double* a, const double* b, double* o) { /* se3.hpp:203-207, so3.hpp:215-232 */
  double t[3];
  quat_rotate(a, b + 4, t);
  double q[4];
  quat_mul(a, b, q);
  double sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (sn != 1.0) {
    double s = 2.0 / (1.0 + sn);
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
  }
  o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
  o[4] = a[4] + t[0]; o[5] = a[5] + t[1]; o[6] = a[6] + t[2];
}
/* SE3::exp (tangent = [upsilon(3), omega(3)]); used only by the synthetic generators
 * (APP/test/util.h:302,338,381).  Standard closed form. */
void orc_se3_exp(const double* a, double* o) {
  const double* u = a; const double* w = a + 3;
  double th2 = dot3(w, w), th = sqrt(th2);
  double q[4];
  double half = 0.5 * th;
  double imag, real;
  if (th < 1e-10) {
    double th4 = th2 * th2;
    imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
    real = 1.0 - th2 / 8.0 + th4 / 384.0;
  } else {
    imag = sin(half) / th;
    real = cos(half);
  }
  q[0] = real; q[1] = imag * w[0]; q[2] = imag * w[1]; q[3] = imag * w[2];
  quat_normalize(q);
  /* V = I + (1-cos)/th^2 * W + (th - sin)/th^3 * W^2 */
  double A, B;
  if (th < 1e-10) { A = 0.5; B = 1.0 / 6.0; }
  else { A = (1 - cos(th)) / th2; B = (th - sin(th)) / (th2 * th); }
  double wu[3], wwu[3];
  cross3(w, u, wu);
  cross3(w, wu, wwu);
  o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
  for (int i = 0; i < 3; ++i) o[4 + i] = u[i] + A * wu[i] + B * wwu[i];
}

/* P3: ApplyLocalUpdateToQuaternion (APP/local_parametrizations/quaternion_parametrization.h:39-61)
 * including the float-typed norm / sinc, then SE3d(q, t) normalises (so3.hpp:536-541). */
void orc_apply_quaternion_update(const double* q, const double* u, double* out) {
  const float norm_update = (float)sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  if (norm_update == 0) {
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  } else {
    const float s = sinf(norm_update) / norm_update;
    double uq[4] = {(double)cosf(norm_update), s * u[0], s * u[1], s * u[2]};
    quat_mul(uq, q, out);
  }
  quat_normalize(out);
}

/* B1: HuberLoss (LV/loss_functions.h:93-131) on the squared residual norm */
double orc_huber_cost_sq(double sq, double k) {
  if (sq < k * k) return 0.5 * sq;
  return k * (sqrt(sq) - 0.5 * k);
}
double orc_huber_weight_sq(double sq, double k) { return (sq < k * k) ? 1 : (k / sqrt(sq)); }

/* ------------------------------------------------------------------------------------------
 * A4: d(R(q) p + t)/d(q,t,p) with the un-normalised polynomial R(q)
 * (APP/bundle_adjustment/joint_optimization_jacobians.h:40-118 [3x10] and :121-343 [3x17]).
 * Written from the closed form dR/dq_k rather than the generated common-subexpression code.
 * Layouts: ComputeJacobian: [q(w,x,y,z) | t | p];
 * ComputeRigJacobian (as consumed by the caller, joint_optimization.cc:405-425):
 *   [rig_q_global(4) | rig_t_global(3) | camera_q_rig(4) | camera_t_rig(3) | point(3)].
 * ---------------------------------------------------------------------------------------- */
static void poly_rotation(const double* q, double* R) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * y * y - 2 * z * z; R[1] = 2 * x * y - 2 * w * z; R[2] = 2 * x * z + 2 * w * y;
  R[3] = 2 * x * y + 2 * w * z; R[4] = 1 - 2 * x * x - 2 * z * z; R[5] = 2 * y * z - 2 * w * x;
  R[6] = 2 * x * z - 2 * w * y; R[7] = 2 * y * z + 2 * w * x; R[8] = 1 - 2 * x * x - 2 * y * y;
}
/* d(R(q) v)/dq, 3x4 row-major (columns w,x,y,z) */
static void drot_dq(const double* q, const double* v, double* D) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double a = v[0], b = v[1], c = v[2];
  D[0] = 2 * y * c - 2 * z * b;  D[1] = 2 * y * b + 2 * z * c;               D[2] = -4 * y * a + 2 * x * b + 2 * w * c;  D[3] = -4 * z * a - 2 * w * b + 2 * x * c;
  D[4] = 2 * z * a - 2 * x * c;  D[5] = 2 * y * a - 4 * x * b - 2 * w * c;   D[6] = 2 * x * a + 2 * z * c;               D[7] = 2 * w * a - 4 * z * b + 2 * y * c;
  D[8] = -2 * y * a + 2 * x * b; D[9] = 2 * z * a + 2 * w * b - 4 * x * c;   D[10] = -2 * w * a + 2 * z * b - 4 * y * c; D[11] = 2 * x * a + 2 * y * b;
}
void orc_compute_jacobian(const double* q, const double* p, double* J /*3x10*/) {
  double D[12], R[9];
  drot_dq(q, p, D);
  poly_rotation(q, R);
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 4; ++k) J[r * 10 + k] = D[r * 4 + k];
    for (int k = 0; k < 3; ++k) J[r * 10 + 4 + k] = (r == k) ? 1 : 0;
    for (int k = 0; k < 3; ++k) J[r * 10 + 7 + k] = R[r * 3 + k];
  }
}
void orc_compute_rig_jacobian(const double* cq, const double* p, const double* rq, const double* rt, double* J /*3x17*/) {
  double Rc[9], Rr[9], Dr[12], Dc[12];
  poly_rotation(cq, Rc);
  poly_rotation(rq, Rr);
  drot_dq(rq, p, Dr);
  double v[3];
  for (int r = 0; r < 3; ++r) v[r] = Rr[r * 3 + 0] * p[0] + Rr[r * 3 + 1] * p[1] + Rr[r * 3 + 2] * p[2] + rt[r];
  drot_dq(cq, v, Dc);
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 4; ++k)
      J[r * 17 + k] = Rc[r * 3 + 0] * Dr[0 * 4 + k] + Rc[r * 3 + 1] * Dr[1 * 4 + k] + Rc[r * 3 + 2] * Dr[2 * 4 + k];
    for (int k = 0; k < 3; ++k) J[r * 17 + 4 + k] = Rc[r * 3 + k];
    for (int k = 0; k < 4; ++k) J[r * 17 + 7 + k] = Dc[r * 4 + k];
    for (int k = 0; k < 3; ++k) J[r * 17 + 11 + k] = (r == k) ? 1 : 0;
    for (int k = 0; k < 3; ++k)
      J[r * 17 + 14 + k] = Rc[r * 3 + 0] * Rr[0 * 3 + k] + Rc[r * 3 + 1] * Rr[1 * 3 + k] + Rc[r * 3 + 2] * Rr[2 * 3 + k];
  }
}
/* QuaternionJacobianWrtLocalUpdate (quaternion_parametrization.h:63-72): 4x3, rows (w,x,y,z) */
static void quat_jac_wrt_update(const double* q, double* Q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  Q[0] = -x; Q[1] = -y; Q[2] = -z;
  Q[3] = w;  Q[4] = z;  Q[5] = -y;
  Q[6] = -z; Q[7] = w;  Q[8] = x;
  Q[9] = y;  Q[10] = -x; Q[11] = w;
}

/* ------------------------------------------------------------------------------------------
 * A1: variable ordering (APP/bundle_adjustment/joint_optimization.cc:49-59, 142-170)
 * ---------------------------------------------------------------------------------------- */
int32_t orc_intrinsics_param_count(const orc_camera* cam) {
  return (cam->model_type == ORC_CENTRAL_GENERIC ? 2 : 5) * cam->grid_w * cam->grid_h;
}
typedef struct {
  int rig_in_state;
  int first_rig_tr_global, first_camera_tr_rig, first_points, first_intrinsics;
  int intr_offset[64];
  int total_dof, block_dof, block_size, n_blocks;
} layout_t;
static void make_layout(const orc_problem* pb, layout_t* L) {
  int N = pb->n_images, C = pb->n_cameras, P = pb->n_points;
  L->rig_in_state = C > 1;
  int rig_dof = L->rig_in_state ? 6 * C : 0;
  L->first_rig_tr_global = pb->eliminate_points ? 3 * P : 0;
  L->first_camera_tr_rig = L->first_rig_tr_global + 6 * N;
  L->first_points = pb->eliminate_points ? 0 : L->first_camera_tr_rig + rig_dof;
  L->first_intrinsics = pb->eliminate_points ? (L->first_camera_tr_rig + rig_dof) : (L->first_points + 3 * P);
  int off = L->first_intrinsics;
  for (int c = 0; c < C; ++c) {
    L->intr_offset[c] = off;
    off += orc_intrinsics_param_count(&pb->cams[c]);
  }
  L->total_dof = pb->localize_only ? L->first_intrinsics : off;
  if (pb->eliminate_points) { L->block_size = 3; L->n_blocks = P; }
  else { L->block_size = 6; L->n_blocks = N; }
  L->block_dof = L->block_size * L->n_blocks;
}
int32_t orc_total_dof(const orc_problem* pb) { layout_t L; make_layout(pb, &L); return L.total_dof; }
int32_t orc_dense_dof(const orc_problem* pb) { layout_t L; make_layout(pb, &L); return L.total_dof - L.block_dof; }

/* ------------------------------------------------------------------------------------------
 * B2/B3: JtJ accumulation (LV/lm_optimizer_jtj_accumulator_base.h:287-401,
 * lm_optimizer_update_accumulator.h:181-322, 478-505).  The overloads reduce to: for every
 * position pair i <= k of the (ascending) index list add Jw[:,i].J[:,k] to H(idx i, idx k);
 * b(idx i) += r.Jw[:,i].  Only upper triangles are written.
 * ---------------------------------------------------------------------------------------- */
static inline void add_H(orc_system* s, int block_dof, int row, int col, double v) {
  if (row < block_dof) {
    if (col < block_dof) {
      int blk = row / s->block_size;
      int base = blk * s->block_size;
      s->block_diag_H[(size_t)blk * s->block_size * s->block_size + (row - base) * s->block_size + (col - base)] += v;
    } else {
      s->off_diag_H[(size_t)row * s->dense_dof + (col - block_dof)] += v;
    }
  } else {
    s->dense_H[(size_t)(row - block_dof) * s->dense_dof + (col - block_dof)] += v;
  }
}
static inline void add_b(orc_system* s, int block_dof, int row, double v) {
  if (row < block_dof) s->block_diag_b[row] += v;
  else s->dense_b[row - block_dof] += v;
}
static void accumulate(orc_system* s, int block_dof, const double* r, double weight, int K, const int* idx, const double* J /*2xK*/) {
  for (int i = 0; i < K; ++i) {
    double jw0 = weight * J[i], jw1 = weight * J[K + i];
    for (int k = i; k < K; ++k) add_H(s, block_dof, idx[i], idx[k], jw0 * J[k] + jw1 * J[K + k]);
    add_b(s, block_dof, idx[i], r[0] * jw0 + r[1] * jw1);
  }
}

/* ------------------------------------------------------------------------------------------
 * A2/A3/A5: per-observation residual and Jacobian
 * (APP/bundle_adjustment/joint_optimization.cc:240-593; M5 = central_grid.h:187-245,
 *  N3 = noncentral_generic.h:224-283)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const orc_problem* pb;
  const orc_state* st;
  layout_t L;
  double** tangents;     /* per camera: 6 doubles per grid point (t1,t2) */
  double** work_grids;   /* mutable copies for the in-place perturbation of M5/N3 */
} pass_ctx;

static int projection_jacobian_wrt_intrinsics(const orc_camera* cam, double* grid, const double* tang,
                                              const double* local_point, const double* pixel, double delta,
                                              int* indices, double* J /*2xK*/) {
  int central = cam->model_type == ORC_CENTRAL_GENERIC;
  int per = central ? 2 : 5;
  int K = per * 16;
  size_t G = (size_t)cam->grid_w * cam->grid_h;
  double dir[3];
  if (central) normalize3(local_point, dir);
  double gp[2];
  orc_pixel_to_grid_point(cam, pixel[0], pixel[1], gp);
  int ix = (int)floor(gp[0]), iy = (int)floor(gp[1]);
  int li = 0;
  for (int y = 0; y < 4; ++y) {
    int gy = iy + y - 1;
    for (int x = 0; x < 4; ++x) {
      int gx = ix + x - 1;
      if (gy < 0 || gy >= cam->grid_h || gx < 0 || gx >= cam->grid_w) return -1; /* CHECK abort in the reference */
      int seq = gx + gy * cam->grid_w;
      for (int i = 0; i < per; ++i) indices[li + i] = per * seq + i;
      const double* t1 = tang + 6 * (size_t)seq;
      const double* t2 = t1 + 3;
      double* gd = grid + 3 * (size_t)seq;
      double od[3] = {gd[0], gd[1], gd[2]};
      if (central) {
        for (int d = 0; d < 2; ++d) {
          apply_local_update_to_direction(gd, t1, t2, d == 0 ? delta : 0, d == 1 ? delta : 0);
          double tp[2] = {pixel[0], pixel[1]};
          int ok = project_target(cam, grid, dir, tp);
          gd[0] = od[0]; gd[1] = od[1]; gd[2] = od[2];
          if (ok <= 0) return 0;
          J[li + d] = (tp[0] - pixel[0]) / delta;
          J[K + li + d] = (tp[1] - pixel[1]) / delta;
        }
      } else {
        double* go = grid + 3 * G + 3 * (size_t)seq;
        double oo[3] = {go[0], go[1], go[2]};
        for (int d = 0; d < 5; ++d) {
          double dl[5] = {0, 0, 0, 0, 0};
          dl[d] = delta;
          double to[3] = {oo[0], oo[1], oo[2]}, td[3] = {od[0], od[1], od[2]};
          apply_local_update_to_line(to, td, t1, t2, dl[0], dl[1], dl[2], dl[3], dl[4]);
          go[0] = to[0]; go[1] = to[1]; go[2] = to[2];
          gd[0] = td[0]; gd[1] = td[1]; gd[2] = td[2];
          double tp[2] = {pixel[0], pixel[1]};
          int ok = project_target(cam, grid, local_point, tp);
          if (ok <= 0) {
            go[0] = oo[0]; go[1] = oo[1]; go[2] = oo[2];
            gd[0] = od[0]; gd[1] = od[1]; gd[2] = od[2];
            return 0;
          }
          J[li + d] = (tp[0] - pixel[0]) / delta;
          J[K + li + d] = (tp[1] - pixel[1]) / delta;
        }
        go[0] = oo[0]; go[1] = oo[1]; go[2] = oo[2];
        gd[0] = od[0]; gd[1] = od[1]; gd[2] = od[2];
      }
      li += per;
    }
  }
  return 1;
}

/* test hook: M5 / N3 on a caller-supplied grid (copied: the routine perturbs it in place); tangents are computed here */
int orc_debug_projection_jacobian_wrt_intrinsics(const orc_camera* cam, const double* grid, const double* local_point,
                                                 const double* pixel, double delta, int32_t* indices, double* J) {
  size_t G = (size_t)cam->grid_w * cam->grid_h;
  size_t n = (cam->model_type == ORC_CENTRAL_GENERIC ? 3 : 6) * G;
  double* g = (double*)malloc(n * sizeof(double));
  double* tang = (double*)malloc(6 * G * sizeof(double));
  memcpy(g, grid, n * sizeof(double));
  for (size_t i = 0; i < G; ++i) orc_tangents(g + 3 * i, tang + 6 * i, tang + 6 * i + 3);
  int idx[MAXK];
  int ok = projection_jacobian_wrt_intrinsics(cam, g, tang, local_point, pixel, delta, idx, J);
  int K = cam->model_type == ORC_CENTRAL_GENERIC ? 32 : 80;
  for (int k = 0; k < K; ++k) indices[k] = idx[k];
  free(g); free(tang);
  return ok;
}

/* what one observation adds to the normal equations (K = 0: nothing) */
typedef struct {
  int K;
  double res[2], weight;
  int idx[6 + 6 + 3 + MAXK];
  double J[2 * (6 + 6 + 3 + MAXK)];
} acc_item;

/* returns cost (>=0) or -1 for an invalid residual; fills rec if non-NULL; if want_acc, `item` receives the
 * observation's contribution to the normal equations (accumulated by the caller, in observation order) */
static double add_reprojection_residual(pass_ctx* ctx, int64_t o, const double* image_tr_global /*7*/,
                                        const double* R /*9*/, int compute_jacobians, int want_acc, acc_item* item,
                                        orc_obs_record* rec) {
  if (item) item->K = 0;
  const orc_problem* pb = ctx->pb;
  const orc_state* st = ctx->st;
  int cam_i = pb->obs_camera[o];
  const orc_camera* cam = &pb->cams[cam_i];
  double* grid = ctx->work_grids[cam_i];
  int central = cam->model_type == ORC_CENTRAL_GENERIC;
  const double* point = st->points + 3 * (size_t)pb->obs_point[o];
  double local[3];
  for (int r = 0; r < 3; ++r) local[r] = R[r * 3 + 0] * point[0] + R[r * 3 + 1] * point[1] + R[r * 3 + 2] * point[2] + image_tr_global[4 + r];

  if (rec) memset(rec, 0, sizeof(*rec));
  double pixel[2] = {pb->last_projection[2 * o], pb->last_projection[2 * o + 1]};
  if (!in_calibrated_area(cam, pixel[0], pixel[1]) || pixel[0] != pixel[0] || pixel[1] != pixel[1])
    center_of_calibrated_area(cam, pixel);
  const long evals0 = g_eval_count;
  if (!orc_project_with_initial_estimate(cam, grid, local, pixel)) {
    center_of_calibrated_area(cam, pixel);
    if (!orc_project_with_initial_estimate(cam, grid, local, pixel)) {
      if (g_eval_trace) g_eval_trace[o] = (int32_t)(g_eval_count - evals0);
      if (rec) rec->cost = -1;
      return -1;
    }
  }
  if (g_eval_trace) g_eval_trace[o] = (int32_t)(g_eval_count - evals0);
  pb->last_projection[2 * o] = pixel[0];
  pb->last_projection[2 * o + 1] = pixel[1];

  double res[2] = {pixel[0] - (double)pb->obs_xy[2 * o], pixel[1] - (double)pb->obs_xy[2 * o + 1]};
  double sq = res[0] * res[0] + res[1] * res[1];
  double cost = orc_huber_cost_sq(sq, 1.0);
  if (rec) {
    rec->valid = 1; rec->pixel[0] = pixel[0]; rec->pixel[1] = pixel[1];
    rec->residual[0] = res[0]; rec->residual[1] = res[1]; rec->cost = cost;
    rec->weight = orc_huber_weight_sq(sq, 1.0);
  }
  if (!compute_jacobians) return cost;

  /* numerical d pixel / d local_point (joint_optimization.cc:357-376) */
  double pwl[6];
  const double kDelta = pb->fd_delta * (central ? sqrt(dot3(local, local)) : 0.1);
  for (int dim = 0; dim < 3; ++dim) {
    double op[3] = {local[0], local[1], local[2]};
    op[dim] += kDelta;
    double opx[2] = {pixel[0], pixel[1]};
    if (!orc_project_with_initial_estimate(cam, grid, op, opx)) return cost; /* residual without Jacobian */
    pwl[0 * 3 + dim] = (opx[0] - pixel[0]) / kDelta;
    pwl[1 * 3 + dim] = (opx[1] - pixel[1]) / kDelta;
  }

  double pose_jac[12], rig_jac[12], point_jac[6];
  int img = pb->obs_image[o];
  if (ctx->L.rig_in_state) {
    const double* ctr = st->camera_tr_rig + 7 * (size_t)cam_i;
    const double* rtg = st->rig_tr_global + 7 * (size_t)img;
    double J[51];
    orc_compute_rig_jacobian(ctr, point, rtg, rtg + 4, J);
    double Qc[12], Qr[12];
    quat_jac_wrt_update(ctr, Qc);
    quat_jac_wrt_update(rtg, Qr);
    double A[8]; /* pwl * J[:,0:4]  (2x4) */
    for (int r = 0; r < 2; ++r)
      for (int k = 0; k < 4; ++k) A[r * 4 + k] = pwl[r * 3 + 0] * J[0 * 17 + k] + pwl[r * 3 + 1] * J[1 * 17 + k] + pwl[r * 3 + 2] * J[2 * 17 + k];
    for (int r = 0; r < 2; ++r)
      for (int k = 0; k < 3; ++k) {
        pose_jac[r * 6 + k] = A[r * 4 + 0] * Qr[0 * 3 + k] + A[r * 4 + 1] * Qr[1 * 3 + k] + A[r * 4 + 2] * Qr[2 * 3 + k] + A[r * 4 + 3] * Qr[3 * 3 + k];
        pose_jac[r * 6 + 3 + k] = pwl[r * 3 + 0] * J[0 * 17 + 4 + k] + pwl[r * 3 + 1] * J[1 * 17 + 4 + k] + pwl[r * 3 + 2] * J[2 * 17 + 4 + k];
      }
    for (int r = 0; r < 2; ++r)
      for (int k = 0; k < 4; ++k) A[r * 4 + k] = pwl[r * 3 + 0] * J[0 * 17 + 7 + k] + pwl[r * 3 + 1] * J[1 * 17 + 7 + k] + pwl[r * 3 + 2] * J[2 * 17 + 7 + k];
    for (int r = 0; r < 2; ++r)
      for (int k = 0; k < 3; ++k) {
        rig_jac[r * 6 + k] = A[r * 4 + 0] * Qc[0 * 3 + k] + A[r * 4 + 1] * Qc[1 * 3 + k] + A[r * 4 + 2] * Qc[2 * 3 + k] + A[r * 4 + 3] * Qc[3 * 3 + k];
        rig_jac[r * 6 + 3 + k] = pwl[r * 3 + 0] * J[0 * 17 + 11 + k] + pwl[r * 3 + 1] * J[1 * 17 + 11 + k] + pwl[r * 3 + 2] * J[2 * 17 + 11 + k];
        point_jac[r * 3 + k] = pwl[r * 3 + 0] * J[0 * 17 + 14 + k] + pwl[r * 3 + 1] * J[1 * 17 + 14 + k] + pwl[r * 3 + 2] * J[2 * 17 + 14 + k];
      }
  } else {
    double J[30];
    orc_compute_jacobian(image_tr_global, point, J);
    double Q[12];
    quat_jac_wrt_update(image_tr_global, Q);
    double B[9]; /* J[:,0:4] * Q (3x3) */
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) B[r * 3 + k] = J[r * 10 + 0] * Q[0 * 3 + k] + J[r * 10 + 1] * Q[1 * 3 + k] + J[r * 10 + 2] * Q[2 * 3 + k] + J[r * 10 + 3] * Q[3 * 3 + k];
    for (int r = 0; r < 2; ++r)
      for (int k = 0; k < 3; ++k) {
        pose_jac[r * 6 + k] = pwl[r * 3 + 0] * B[0 * 3 + k] + pwl[r * 3 + 1] * B[1 * 3 + k] + pwl[r * 3 + 2] * B[2 * 3 + k];
        pose_jac[r * 6 + 3 + k] = pwl[r * 3 + 0] * J[0 * 10 + 4 + k] + pwl[r * 3 + 1] * J[1 * 10 + 4 + k] + pwl[r * 3 + 2] * J[2 * 10 + 4 + k];
        point_jac[r * 3 + k] = pwl[r * 3 + 0] * J[0 * 10 + 7 + k] + pwl[r * 3 + 1] * J[1 * 10 + 7 + k] + pwl[r * 3 + 2] * J[2 * 10 + 7 + k];
      }
    memset(rig_jac, 0, sizeof(rig_jac));
  }

  int gidx[MAXK];
  double gjac[2 * MAXK];
  int Kg = 0;
  if (!pb->localize_only) {
    Kg = central ? 32 : 80;
    int ok = projection_jacobian_wrt_intrinsics(cam, grid, ctx->tangents[cam_i], local, pixel, pb->fd_delta, gidx, gjac);
    if (ok <= 0) return cost; /* residual without Jacobian */
  }
  if (rec) {
    rec->has_jacobian = 1;
    memcpy(rec->pose_jac, pose_jac, sizeof(pose_jac));
    memcpy(rec->rig_jac, rig_jac, sizeof(rig_jac));
    memcpy(rec->point_jac, point_jac, sizeof(point_jac));
    for (int k = 0; k < Kg; ++k) { rec->grid_indices[k] = gidx[k]; rec->grid_jac[k] = gjac[k]; rec->grid_jac[Kg + k] = gjac[Kg + k]; }
  }
  if (!want_acc) return cost;

  /* assemble the ascending index list / 2xK Jacobian (joint_optimization.cc:479-590) */
  int* idx = item->idx;
  double* Jall = item->J;
  int K = 6 + (ctx->L.rig_in_state ? 6 : 0) + 3 + Kg;
  int pos = 0;
  int pose_idx = ctx->L.first_rig_tr_global + 6 * img;
  int rig_idx = ctx->L.first_camera_tr_rig + 6 * cam_i;
  int point_idx = ctx->L.first_points + 3 * pb->obs_point[o];
#define PUT(base, n, src, stride) \
  for (int k = 0; k < (n); ++k) { idx[pos] = (base) + k; Jall[pos] = (src)[k]; Jall[K + pos] = (src)[(stride) + k]; ++pos; }
  if (pb->eliminate_points) {
    PUT(point_idx, 3, point_jac, 3);
    PUT(pose_idx, 6, pose_jac, 6);
    if (ctx->L.rig_in_state) PUT(rig_idx, 6, rig_jac, 6);
  } else {
    PUT(pose_idx, 6, pose_jac, 6);
    if (ctx->L.rig_in_state) PUT(rig_idx, 6, rig_jac, 6);
    PUT(point_idx, 3, point_jac, 3);
  }
#undef PUT
  for (int k = 0; k < Kg; ++k) { idx[pos] = ctx->L.intr_offset[cam_i] + gidx[k]; Jall[pos] = gjac[k]; Jall[K + pos] = gjac[Kg + k]; ++pos; }
  item->K = K; item->res[0] = res[0]; item->res[1] = res[1]; item->weight = orc_huber_weight_sq(sq, 1.0);
  return cost;
}

static void ctx_init(pass_ctx* ctx, const orc_problem* pb, const orc_state* st, int with_tangents) {
  ctx->pb = pb; ctx->st = st;
  make_layout(pb, &ctx->L);
  int C = pb->n_cameras;
  ctx->tangents = (double**)calloc(C, sizeof(double*));
  ctx->work_grids = (double**)calloc(C, sizeof(double*));
  for (int c = 0; c < C; ++c) {
    const orc_camera* cam = &pb->cams[c];
    size_t G = (size_t)cam->grid_w * cam->grid_h;
    size_t n = (cam->model_type == ORC_CENTRAL_GENERIC ? 3 : 6) * G;
    ctx->work_grids[c] = (double*)malloc(n * sizeof(double));
    memcpy(ctx->work_grids[c], st->grids[c], n * sizeof(double));
    if (with_tangents) { /* ComputeTangentsImage, joint_optimization.cc:229-238, 254-270 */
      ctx->tangents[c] = (double*)malloc(6 * G * sizeof(double));
      for (size_t g = 0; g < G; ++g) orc_tangents(st->grids[c] + 3 * g, ctx->tangents[c] + 6 * g, ctx->tangents[c] + 6 * g + 3);
    }
  }
}
static void ctx_free(pass_ctx* ctx) {
  for (int c = 0; c < ctx->pb->n_cameras; ++c) { free(ctx->tangents[c]); free(ctx->work_grids[c]); }
  free(ctx->tangents); free(ctx->work_grids);
}

/* Observations are processed in chunks: the per-observation work (projections, finite differences, Jacobian chain)
 * of a chunk may run on several threads -- each with its own scratch copy of the grids, which M5 / N3 perturb in
 * place -- and the chunk's contributions are then added to the normal equations and to the cost by ONE thread in
 * observation order, so sums are bit-identical to the single-threaded loop of the reference
 * (joint_optimization.cc:273-291). */
static double run_pass(const orc_problem* pb, const orc_state* st, int compute_jacobians, orc_system* sys,
                       double* cost_vec, orc_obs_record* records, int img_begin, int img_end) {
  const int nthreads = g_threads;
  pass_ctx* ctxs = (pass_ctx*)malloc(nthreads * sizeof(pass_ctx));
  for (int t = 0; t < nthreads; ++t) ctx_init(&ctxs[t], pb, st, compute_jacobians);
  double cost = 0;
  int64_t o_begin = 0;
  while (o_begin < pb->n_obs && pb->obs_image[o_begin] < img_begin) ++o_begin;
  int64_t o_end = o_begin;
  while (o_end < pb->n_obs && pb->obs_image[o_end] < img_end) ++o_end;
  const int64_t chunk = nthreads > 1 ? 256 * (int64_t)nthreads : 1;
  acc_item* items = sys ? (acc_item*)malloc((size_t)chunk * sizeof(acc_item)) : NULL;
  double* costs = (double*)malloc((size_t)chunk * sizeof(double));
  const int block_dof = ctxs[0].L.block_dof;
  for (int64_t c0 = o_begin; c0 < o_end; c0 += chunk) {
    const int64_t c1 = c0 + chunk < o_end ? c0 + chunk : o_end;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads) if (nthreads > 1)
    for (int64_t o = c0; o < c1; ++o) {
#ifdef _OPENMP
      pass_ctx* ctx = &ctxs[omp_get_thread_num()];
#else
      pass_ctx* ctx = &ctxs[0];
#endif
      const int img = pb->obs_image[o], cam_i = pb->obs_camera[o];
      double itg[7], R[9];
      orc_se3_mul(st->camera_tr_rig + 7 * (size_t)cam_i, st->rig_tr_global + 7 * (size_t)img, itg); /* :277 */
      quat_to_matrix(itg, R);
      costs[o - c0] = add_reprojection_residual(ctx, o, itg, R, compute_jacobians, sys != NULL, items ? &items[o - c0] : NULL,
                                                records ? &records[o] : NULL);
    }
    for (int64_t o = c0; o < c1; ++o) {
      const double c = costs[o - c0];
      if (sys && items[o - c0].K > 0)
        accumulate(sys, block_dof, items[o - c0].res, items[o - c0].weight, items[o - c0].K, items[o - c0].idx, items[o - c0].J);
      if (cost_vec) cost_vec[o] = c;
      if (c >= 0) cost += c;
    }
  }
  free(items); free(costs);
  for (int t = 0; t < nthreads; ++t) ctx_free(&ctxs[t]);
  free(ctxs);
  return cost;
}

double orc_cost_pass(const orc_problem* pb, const orc_state* st, double* cost_vec) {
  return run_pass(pb, st, 0, NULL, cost_vec, NULL, 0, pb->n_images);
}

/* Compute<false> with the per-observation records filled (valid, pixel, residual, cost): what a caller needs to drive an
 * accumulator of its own through the cost-only pass (oracle/ref_lmopt.cc feeds the reference's LMOptimizer with them). */
double orc_cost_pass_records(const orc_problem* pb, const orc_state* st, double* cost_vec, orc_obs_record* records) {
  return run_pass(pb, st, 0, NULL, cost_vec, records, 0, pb->n_images);
}

double orc_jacobian_pass(const orc_problem* pb, const orc_state* st, orc_system* sys, double* cost_vec,
                         orc_obs_record* records, int32_t img_begin, int32_t img_end) {
  if (sys) {
    size_t bs = sys->block_size, nb = sys->n_blocks, dd = sys->dense_dof;
    memset(sys->block_diag_H, 0, nb * bs * bs * sizeof(double));
    memset(sys->off_diag_H, 0, nb * bs * dd * sizeof(double));
    memset(sys->dense_H, 0, dd * dd * sizeof(double));
    memset(sys->block_diag_b, 0, nb * bs * sizeof(double));
    memset(sys->dense_b, 0, dd * sizeof(double));
  }
  return run_pass(pb, st, 1, sys, cost_vec, records, img_begin, img_end);
}

/* ------------------------------------------------------------------------------------------
 * Eigen 3.3.7 LDLT (external dependency; call sites LV/lm_optimizer.h:1289, 1361):
 * in-place, lower storage, pivot = largest |diagonal| among the not-yet-processed *stored*
 * diagonal entries (Eigen's unblocked kernel is left-looking, so those are the original values),
 * solve with zero tolerance on D.
 * ---------------------------------------------------------------------------------------- */
typedef struct { int n; double* m; /* n x n column-major, lower */ int* transp; } ldlt_t;
#define LM(i, j) m[(size_t)(j) * n + (i)]
static void ldlt_compute_unblocked(ldlt_t* f) {
  int n = f->n; double* m = f->m;
  double* temp = (double*)malloc(n * sizeof(double));
  if (n <= 1) { if (n == 1) f->transp[0] = 0; free(temp); return; }
  for (int k = 0; k < n; ++k) {
    int piv = k; double best = -1;
    for (int i = k; i < n; ++i) { double a = fabs(LM(i, i)); if (a > best) { best = a; piv = i; } }
    f->transp[k] = piv;
    if (k != piv) {
      int s = n - piv - 1;
      for (int j = 0; j < k; ++j) { double t = LM(k, j); LM(k, j) = LM(piv, j); LM(piv, j) = t; }
      for (int i = 0; i < s; ++i) { double t = LM(piv + 1 + i, k); LM(piv + 1 + i, k) = LM(piv + 1 + i, piv); LM(piv + 1 + i, piv) = t; }
      { double t = LM(k, k); LM(k, k) = LM(piv, piv); LM(piv, piv) = t; }
      for (int i = k + 1; i < piv; ++i) { double t = LM(i, k); LM(i, k) = LM(piv, i); LM(piv, i) = t; }
    }
    int rs = n - k - 1;
    if (k > 0) {
      double acc = 0;
      for (int j = 0; j < k; ++j) { temp[j] = LM(j, j) * LM(k, j); acc += LM(k, j) * temp[j]; }
      LM(k, k) -= acc;
      if (rs > 0) {
        for (int j = 0; j < k; ++j) {
          double tj = temp[j];
          const double* col = &LM(k + 1, j);
          double* dst = &LM(k + 1, k);
          for (int i = 0; i < rs; ++i) dst[i] -= col[i] * tj;
        }
      }
    }
    double akk = LM(k, k);
    int valid = fabs(akk) > 0;
    if (k == 0 && !valid) { for (int j = 0; j < n; ++j) f->transp[j] = j; break; }
    if (rs > 0 && valid) { double* dst = &LM(k + 1, k); for (int i = 0; i < rs; ++i) dst[i] /= akk; }
  }
  free(temp);
}
/* The same factorisation, cache-blocked and (optionally) multi-threaded, with bit-identical results.
 * Two facts make that possible: (1) the unblocked kernel above is left-looking, so at step k the diagonal entries it
 * compares for the pivot are still the ORIGINAL ones -- the whole transposition sequence follows from the original
 * diagonal and the symmetric permutation can be applied up front; (2) every entry is then computed as
 *   L(i,k) = (A(i,k) - l(i,0) t(k,0) - l(i,1) t(k,1) - ...) / d_k,   t(k,j) = d_j l(k,j),  subtractions in j order,
 * and the diagonal as A(k,k) - (sum_j l(k,j) t(k,j)), which a panel algorithm can reproduce term by term.
 * Panel [k0,k1): (a) the terms j < k0 for all panel columns, tiled over rows and j (GEMM-like, parallel over row
 * blocks); (b) the NB x NB diagonal block, serial; (c) the terms k0 <= j < k for the rows below, parallel over row
 * blocks.  Used for n >= 256; smaller systems and a zero first pivot take the unblocked kernel. */
#define LDLT_NB 64
#define LDLT_RB 128
static void ldlt_compute(ldlt_t* f) {
  const int n = f->n; double* m = f->m;
  if (n < 256) { ldlt_compute_unblocked(f); return; }
  /* pivot sequence from the original diagonal (selection with "first largest wins", as in the kernel above) */
  double* dg = (double*)malloc(n * sizeof(double));
  int* perm = (int*)malloc(n * sizeof(int));       /* perm[new] = old */
  for (int i = 0; i < n; ++i) { dg[i] = fabs(LM(i, i)); perm[i] = i; }
  for (int k = 0; k < n; ++k) {
    int piv = k; double best = -1;
    for (int i = k; i < n; ++i) if (dg[i] > best) { best = dg[i]; piv = i; }
    f->transp[k] = piv;
    if (piv != k) { double t = dg[k]; dg[k] = dg[piv]; dg[piv] = t; int q = perm[k]; perm[k] = perm[piv]; perm[piv] = q; }
  }
  if (!(dg[0] > 0)) { free(dg); free(perm); ldlt_compute_unblocked(f); return; }   /* zero matrix: kernel's early exit */
  free(dg);
  /* symmetric permutation of the lower triangle */
  {
    double* w = (double*)malloc((size_t)n * n * sizeof(double));
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int j = 0; j < n; ++j)
      for (int i = j; i < n; ++i) {
        int r = perm[i], c = perm[j];
        w[(size_t)j * n + i] = r >= c ? LM(r, c) : LM(c, r);
      }
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int j = 0; j < n; ++j) memcpy(&LM(j, j), &w[(size_t)j * n + j], (size_t)(n - j) * sizeof(double));
    free(w);
  }
  free(perm);
  double* T = (double*)malloc((size_t)LDLT_NB * n * sizeof(double));     /* T[kk][j] = d_j l(k0+kk, j) */
  double acc[LDLT_NB];
  for (int k0 = 0; k0 < n; k0 += LDLT_NB) {
    const int k1 = k0 + LDLT_NB < n ? k0 + LDLT_NB : n, nbk = k1 - k0;
    /* (a) terms j < k0 */
    for (int kk = 0; kk < nbk; ++kk) {
      double a = 0;
      double* Tk = T + (size_t)kk * n;
      for (int j = 0; j < k0; ++j) { Tk[j] = LM(j, j) * LM(k0 + kk, j); a += LM(k0 + kk, j) * Tk[j]; }
      acc[kk] = a;
    }
    if (k0 > 0) {
      const int nrb = (n - k0 + LDLT_RB - 1) / LDLT_RB;
      const int team_a = g_threads < nrb ? g_threads : nrb;      /* no more threads than row blocks: idle ones only spin */
#pragma omp parallel for schedule(dynamic, 1) num_threads(team_a)
      for (int rb = 0; rb < nrb; ++rb) {
        const int i0 = k0 + rb * LDLT_RB, i1 = i0 + LDLT_RB < n ? i0 + LDLT_RB : n;
        /* the block L(i0:i1, j0:j1) is packed into a padded thread-local buffer once and reused by all panel
         * columns (column strides of n doubles alias in the L1 sets for many n) */
        double Lp[64][LDLT_RB + 8];
        for (int j0 = 0; j0 < k0; j0 += 64) {
          const int j1 = j0 + 64 < k0 ? j0 + 64 : k0;
          for (int j = j0; j < j1; ++j) memcpy(Lp[j - j0], &LM(i0, j), (size_t)(i1 - i0) * sizeof(double));
          for (int kk = 0; kk < nbk; ++kk) {
            const int k = k0 + kk;
            const int ib = i0 > k + 1 ? i0 : k + 1;
            if (ib >= i1) continue;
            double* restrict dst = &LM(0, k);
            const double* Tk = T + (size_t)kk * n;
            int i = ib;
            for (; i + 16 <= i1; i += 16) {          /* 16 rows stay in registers while j runs (same order per entry) */
              double d[16];
              for (int t = 0; t < 16; ++t) d[t] = dst[i + t];
              for (int j = j0; j < j1; ++j) {
                const double tj = Tk[j];
                const double* restrict col = &Lp[j - j0][i - i0];
                for (int t = 0; t < 16; ++t) d[t] -= col[t] * tj;
              }
              for (int t = 0; t < 16; ++t) dst[i + t] = d[t];
            }
            for (; i < i1; ++i) {
              double d = dst[i];
              for (int j = j0; j < j1; ++j) d -= Lp[j - j0][i - i0] * Tk[j];
              dst[i] = d;
            }
          }
        }
      }
    }
    /* (b) diagonal block: rows and columns of the panel */
    for (int kk = 0; kk < nbk; ++kk) {
      const int k = k0 + kk;
      double* Tk = T + (size_t)kk * n;
      double a = acc[kk];
      for (int j = k0; j < k; ++j) { Tk[j] = LM(j, j) * LM(k, j); a += LM(k, j) * Tk[j]; }
      if (k > 0) LM(k, k) -= a;
      for (int j = k0; j < k; ++j) {
        const double tj = Tk[j];
        for (int i = k + 1; i < k1; ++i) LM(i, k) -= LM(i, j) * tj;
      }
      const double akk = LM(k, k);
      if (fabs(akk) > 0) for (int i = k + 1; i < k1; ++i) LM(i, k) /= akk;
    }
    /* (c) rows below the panel */
    if (k1 < n) {
      const int nrb = (n - k1 + LDLT_RB - 1) / LDLT_RB;
      const int team_c = g_threads < nrb ? g_threads : nrb;
#pragma omp parallel for schedule(dynamic, 1) num_threads(team_c)
      for (int rb = 0; rb < nrb; ++rb) {
        const int i0 = k1 + rb * LDLT_RB, i1 = i0 + LDLT_RB < n ? i0 + LDLT_RB : n;
        for (int kk = 0; kk < nbk; ++kk) {
          const int k = k0 + kk;
          double* restrict dst = &LM(0, k);
          const double* Tk = T + (size_t)kk * n;
          const double akk = LM(k, k);
          const int valid = fabs(akk) > 0;
          int i = i0;
          for (; i + 16 <= i1; i += 16) {
            double d[16];
            for (int t = 0; t < 16; ++t) d[t] = dst[i + t];
            for (int j = k0; j < k; ++j) {
              const double tj = Tk[j];
              const double* restrict col = &LM(i, j);
              for (int t = 0; t < 16; ++t) d[t] -= col[t] * tj;
            }
            if (valid) for (int t = 0; t < 16; ++t) d[t] /= akk;
            for (int t = 0; t < 16; ++t) dst[i + t] = d[t];
          }
          for (; i < i1; ++i) {
            double d = dst[i];
            for (int j = k0; j < k; ++j) d -= LM(i, j) * Tk[j];
            dst[i] = valid ? d / akk : d;
          }
        }
      }
    }
  }
  free(T);
}
static void ldlt_solve(const ldlt_t* f, double* x /* in: b, out: x */) {
  int n = f->n; const double* m = f->m;
  for (int k = 0; k < n; ++k) { int p = f->transp[k]; if (p != k) { double t = x[k]; x[k] = x[p]; x[p] = t; } }
  for (int j = 0; j < n; ++j) { double xj = x[j]; if (xj != 0) for (int i = j + 1; i < n; ++i) x[i] -= LM(i, j) * xj; }
  const double tol = 1.0 / 1.7976931348623157e308;
  for (int i = 0; i < n; ++i) { if (fabs(LM(i, i)) > tol) x[i] /= LM(i, i); else x[i] = 0; }
  for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int j = i + 1; j < n; ++j) s -= LM(j, i) * x[j]; x[i] = s; }
  for (int k = n - 1; k >= 0; --k) { int p = f->transp[k]; if (p != k) { double t = x[k]; x[k] = x[p]; x[p] = t; } }
}
#undef LM
void orc_ldlt_solve_upper(const double* A, int n, const double* b, double* x) {
  ldlt_t f; f.n = n;
  f.m = (double*)malloc((size_t)n * n * sizeof(double));
  f.transp = (int*)malloc(n * sizeof(int));
  /* selfadjointView<Upper>: lower(i,j) of the column-major work matrix = A_upper(j,i), i >= j */
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) f.m[(size_t)j * n + i] = A[(size_t)j * n + i];
  ldlt_compute(&f);
  if (x != b) memcpy(x, b, n * sizeof(double));
  ldlt_solve(&f, x);
  free(f.m); free(f.transp);
}

/* the textbook kernel only (tests compare the blocked kernel with it bit for bit) */
void orc_ldlt_solve_upper_unblocked(const double* A, int n, const double* b, double* x) {
  ldlt_t f; f.n = n;
  f.m = (double*)malloc((size_t)n * n * sizeof(double));
  f.transp = (int*)malloc(n * sizeof(int));
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) f.m[(size_t)j * n + i] = A[(size_t)j * n + i];
  ldlt_compute_unblocked(&f);
  if (x != b) memcpy(x, b, n * sizeof(double));
  ldlt_solve(&f, x);
  free(f.m); free(f.transp);
}

/* B6: SolveWithSchurComplementDenseOffDiag (LV/lm_optimizer.h:1247-1369) */
void orc_schur_solve(const orc_system* s, double* x) {
  int bs = s->block_size, nb = s->n_blocks, dd = s->dense_dof;
  size_t bd = (size_t)bs * nb;
  double* D_inv_B = (double*)malloc(bd * dd * sizeof(double));
  double* D_inv_b1 = (double*)malloc(bd * sizeof(double));
  double Hinv[36], e[6];
  for (int blk = 0; blk < nb; ++blk) {
    size_t base = (size_t)blk * bs;
    const double* Hb = s->block_diag_H + (size_t)blk * bs * bs;
    for (int c = 0; c < bs; ++c) { /* ldlt().solve(I), column by column */
      for (int r = 0; r < bs; ++r) e[r] = (r == c) ? 1 : 0;
      double col[6];
      orc_ldlt_solve_upper(Hb, bs, e, col);
      for (int r = 0; r < bs; ++r) Hinv[r * bs + c] = col[r];
    }
    for (int row = 0; row < bs; ++row) {
      double r = 0;
      for (int k = 0; k < bs; ++k) r += Hinv[row * bs + k] * s->block_diag_b[base + k];
      D_inv_b1[base + row] = r;
      double* out = D_inv_B + (base + row) * dd;
      for (int col = 0; col < dd; ++col) out[col] = 0;
      for (int k = 0; k < bs; ++k) {
        double h = Hinv[row * bs + k];
        const double* src = s->off_diag_H + (base + k) * dd;
        for (int col = 0; col < dd; ++col) out[col] += h * src[col];
      }
    }
  }
  double* schur_b = (double*)malloc(dd * sizeof(double));
  double* schur_M = (double*)calloc((size_t)dd * dd, sizeof(double));
  for (int i = 0; i < dd; ++i) schur_b[i] = 0;
  for (size_t k = 0; k < bd; ++k) {
    const double* Brow = s->off_diag_H + k * dd;
    double v = D_inv_b1[k];
    if (v != 0) for (int i = 0; i < dd; ++i) schur_b[i] += Brow[i] * v;
  }
  for (int i = 0; i < dd; ++i) schur_b[i] = s->dense_b[i] - schur_b[i];
  /* B^T D^-1 B, upper triangle: every entry accumulates its rank-1 terms in block-row order k = 0, 1, ...
   * (lm_optimizer.h:1325-1329 leaves the order to Eigen's product kernel).  Tiled over (i, j) so that a tile of M
   * stays in cache while k runs; the per-entry order does not depend on the tiling or on the thread count. */
  {
    const int TI = 32, TJ = 512;
    const int nti = (dd + TI - 1) / TI, ntj = (dd + TJ - 1) / TJ;
#pragma omp parallel for collapse(2) schedule(dynamic, 1) num_threads(g_threads)
    for (int ti = 0; ti < nti; ++ti)
      for (int tj = 0; tj < ntj; ++tj) {
        const int i0 = ti * TI, i1 = i0 + TI < dd ? i0 + TI : dd;
        const int j0 = tj * TJ, j1 = j0 + TJ < dd ? j0 + TJ : dd;
        if (j1 <= i0) continue;                      /* tile entirely below the diagonal */
        for (size_t k = 0; k < bd; ++k) {
          const double* Brow = s->off_diag_H + k * dd;
          const double* restrict Wrow = D_inv_B + k * dd;
          for (int i = i0; i < i1; ++i) {
            const double bi = Brow[i];
            if (bi == 0) continue;
            double* restrict Mrow = schur_M + (size_t)i * dd;
            for (int j = (i > j0 ? i : j0); j < j1; ++j) Mrow[j] += bi * Wrow[j];
          }
        }
      }
  }
  for (int i = 0; i < dd; ++i) {
    double* Mrow = schur_M + (size_t)i * dd;
    const double* Hrow = s->dense_H + (size_t)i * dd;
    for (int j = i; j < dd; ++j) Mrow[j] = Hrow[j] - Mrow[j];
  }
  double* xd = x + bd;
  orc_ldlt_solve_upper(schur_M, dd, schur_b, xd);
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (long long k = 0; k < (long long)bd; ++k) {
    const double* Wrow = D_inv_B + (size_t)k * dd;
    double acc = 0;
    for (int i = 0; i < dd; ++i) acc += Wrow[i] * xd[i];
    x[k] = D_inv_b1[k] - acc;
  }
  free(D_inv_B); free(D_inv_b1); free(schur_b); free(schur_M);
}

/* A1: JointOptimizationState::operator-= (joint_optimization.cc:172-214),
 * M6 SubtractDelta (central_grid.h:168-184), N3 (noncentral_generic.h:195-219) */
void orc_apply_update(const orc_problem* pb, const orc_state* in, const double* x, orc_state* out) {
  layout_t L; make_layout(pb, &L);
  int N = pb->n_images, C = pb->n_cameras, P = pb->n_points;
  for (int i = 0; i < N; ++i) {
    const double* d = x + L.first_rig_tr_global + 6 * i;
    double nd[3] = {-d[0], -d[1], -d[2]};
    orc_apply_quaternion_update(in->rig_tr_global + 7 * i, nd, out->rig_tr_global + 7 * i);
    for (int k = 0; k < 3; ++k) out->rig_tr_global[7 * i + 4 + k] = in->rig_tr_global[7 * i + 4 + k] - d[3 + k];
  }
  for (int c = 0; c < C; ++c) {
    if (L.rig_in_state) {
      const double* d = x + L.first_camera_tr_rig + 6 * c;
      double nd[3] = {-d[0], -d[1], -d[2]};
      orc_apply_quaternion_update(in->camera_tr_rig + 7 * c, nd, out->camera_tr_rig + 7 * c);
      for (int k = 0; k < 3; ++k) out->camera_tr_rig[7 * c + 4 + k] = in->camera_tr_rig[7 * c + 4 + k] - d[3 + k];
    } else {
      memcpy(out->camera_tr_rig + 7 * c, in->camera_tr_rig + 7 * c, 7 * sizeof(double));
    }
  }
  for (int p = 0; p < 3 * P; ++p) out->points[p] = in->points[p] - x[L.first_points + p];
  for (int c = 0; c < C; ++c) {
    const orc_camera* cam = &pb->cams[c];
    size_t G = (size_t)cam->grid_w * cam->grid_h;
    int central = cam->model_type == ORC_CENTRAL_GENERIC;
    size_t n = (central ? 3 : 6) * G;
    if (out->grids[c] != in->grids[c]) memcpy(out->grids[c], in->grids[c], n * sizeof(double));
    if (pb->localize_only) continue;
    const double* d = x + L.intr_offset[c];
    double* g = out->grids[c];
    for (size_t i = 0; i < G; ++i) {
      double t1[3], t2[3];
      orc_tangents(g + 3 * i, t1, t2);
      if (central) {
        apply_local_update_to_direction(g + 3 * i, t1, t2, -d[2 * i], -d[2 * i + 1]);
      } else {
        apply_local_update_to_line(g + 3 * G + 3 * i, g + 3 * i, t1, t2, -d[5 * i], -d[5 * i + 1], -d[5 * i + 2], -d[5 * i + 3], -d[5 * i + 4]);
      }
    }
  }
}

/* B5: CostIsSmallerThan (LV/lm_optimizer.h:993-1011) */
static int cost_is_smaller_than(const double* left, const double* right, int64_t n) {
  double ls = 0, rs = 0; int64_t count = 0;
  for (int64_t i = 0; i < n; ++i)
    if (left[i] >= 0 && right[i] >= 0) { ls += left[i]; rs += right[i]; ++count; }
  return count > 0 && ls < rs;
}

static void state_alloc_like(const orc_problem* pb, orc_state* s) {
  s->rig_tr_global = (double*)malloc(7 * (size_t)pb->n_images * sizeof(double));
  s->camera_tr_rig = (double*)malloc(7 * (size_t)pb->n_cameras * sizeof(double));
  s->points = (double*)malloc(3 * (size_t)pb->n_points * sizeof(double));
  s->grids = (double**)malloc(pb->n_cameras * sizeof(double*));
  for (int c = 0; c < pb->n_cameras; ++c) {
    size_t G = (size_t)pb->cams[c].grid_w * pb->cams[c].grid_h;
    s->grids[c] = (double*)malloc((pb->cams[c].model_type == ORC_CENTRAL_GENERIC ? 3 : 6) * G * sizeof(double));
  }
}
static void state_copy(const orc_problem* pb, const orc_state* src, orc_state* dst) {
  memcpy(dst->rig_tr_global, src->rig_tr_global, 7 * (size_t)pb->n_images * sizeof(double));
  memcpy(dst->camera_tr_rig, src->camera_tr_rig, 7 * (size_t)pb->n_cameras * sizeof(double));
  memcpy(dst->points, src->points, 3 * (size_t)pb->n_points * sizeof(double));
  for (int c = 0; c < pb->n_cameras; ++c) {
    size_t G = (size_t)pb->cams[c].grid_w * pb->cams[c].grid_h;
    memcpy(dst->grids[c], src->grids[c], (pb->cams[c].model_type == ORC_CENTRAL_GENERIC ? 3 : 6) * G * sizeof(double));
  }
}
static void state_free(const orc_problem* pb, orc_state* s) {
  free(s->rig_tr_global); free(s->camera_tr_rig); free(s->points);
  for (int c = 0; c < pb->n_cameras; ++c) free(s->grids[c]);
  free(s->grids);
}

/* A6 + B4: OptimizeJointly / LMOptimizer::OptimizeImpl with max_iteration_count=1 per Optimize call
 * (APP/bundle_adjustment/joint_optimization.cc:757-953, LV/lm_optimizer.h:629-991).
 * A fresh LMOptimizer is used per outer iteration with init_lambda carried over (:916-925). */
double orc_optimize_jointly(orc_problem* pb, orc_state* st, int max_iteration_count, double init_lambda,
                            double* final_lambda, int32_t* performed_an_iteration, double* timings3,
                            int32_t* lm_attempts) {
  layout_t L; make_layout(pb, &L);
  if (performed_an_iteration) *performed_an_iteration = 0;
  if (timings3) timings3[0] = timings3[1] = timings3[2] = 0;
  if (lm_attempts) *lm_attempts = 0;
  const int max_lm_attempts = 50;
  const double init_lambda_factor = 0.00001;
  int dof = L.total_dof, bs = L.block_size, nb = L.n_blocks, dd = dof - L.block_dof;
  orc_system sys;
  sys.block_size = bs; sys.n_blocks = nb; sys.dense_dof = dd;
  sys.block_diag_H = (double*)malloc((size_t)nb * bs * bs * sizeof(double));
  sys.off_diag_H = (double*)malloc((size_t)nb * bs * dd * sizeof(double));
  sys.dense_H = (double*)malloc((size_t)dd * dd * sizeof(double));
  sys.block_diag_b = (double*)malloc((size_t)nb * bs * sizeof(double));
  sys.dense_b = (double*)malloc((size_t)dd * sizeof(double));
  double* cost_vec = (double*)malloc(pb->n_obs * sizeof(double));
  double* test_vec = (double*)malloc(pb->n_obs * sizeof(double));
  double* orig_diag = (double*)malloc(dof * sizeof(double));
  double* x = (double*)malloc(dof * sizeof(double));
  orc_state upd; state_alloc_like(pb, &upd);
  double final_cost = -1;
  double lambda = 0;

  for (int iteration = 0; iteration < max_iteration_count; ++iteration) {
    /* one optimizer.Optimize(max_iteration_count = 1) call */
    double t0 = now_seconds();
    double last_cost = orc_jacobian_pass(pb, st, &sys, cost_vec, NULL, 0, pb->n_images);
    if (timings3) timings3[0] += now_seconds() - t0;
    int num_iterations_performed = 0;
    if (last_cost == 0) { final_cost = last_cost; if (final_lambda) *final_lambda = lambda; break; }
    if (init_lambda >= 0) {
      lambda = init_lambda;
    } else {
      lambda = 0;
      for (int b = 0; b < nb; ++b) for (int k = 0; k < bs; ++k) lambda += sys.block_diag_H[(size_t)b * bs * bs + k * bs + k];
      for (int i = 0; i < dd; ++i) lambda += sys.dense_H[(size_t)i * dd + i];
      lambda = init_lambda_factor * lambda / dof;
    }
    int di = 0;
    for (int b = 0; b < nb; ++b) for (int k = 0; k < bs; ++k) orig_diag[di++] = sys.block_diag_H[(size_t)b * bs * bs + k * bs + k];
    for (int i = 0; i < dd; ++i) orig_diag[di++] = sys.dense_H[(size_t)i * dd + i];

    int applied = 0;
    for (int lm = 0; lm < max_lm_attempts; ++lm) {
      if (lm_attempts) *lm_attempts += 1;
      t0 = now_seconds();
      di = 0;
      for (int b = 0; b < nb; ++b) for (int k = 0; k < bs; ++k) sys.block_diag_H[(size_t)b * bs * bs + k * bs + k] = orig_diag[di++] + lambda;
      for (int i = 0; i < dd; ++i) sys.dense_H[(size_t)i * dd + i] = orig_diag[di++] + lambda;
      orc_schur_solve(&sys, x);
      if (timings3) timings3[1] += now_seconds() - t0;
      if (x[0] != x[0]) { lambda = 2.f * lambda; continue; }
      orc_apply_update(pb, st, x, &upd);
      t0 = now_seconds();
      double test_cost = orc_cost_pass(pb, &upd, test_vec);
      if (timings3) timings3[2] += now_seconds() - t0;
      if (cost_is_smaller_than(test_vec, cost_vec, pb->n_obs)) {
        state_copy(pb, &upd, st);
        lambda = 0.5f * lambda;
        applied = 1;
        num_iterations_performed += 1;
        last_cost = test_cost;
        break;
      } else {
        lambda = 2.f * lambda;
      }
    }
    final_cost = last_cost;
    init_lambda = lambda;
    if (final_lambda) *final_lambda = lambda;
    if (num_iterations_performed == 0) break;
    if (performed_an_iteration) *performed_an_iteration = 1;
    (void)applied;
  }
  state_free(pb, &upd);
  free(sys.block_diag_H); free(sys.off_diag_H); free(sys.dense_H); free(sys.block_diag_b); free(sys.dense_b);
  free(cost_vec); free(test_vec); free(orig_diag); free(x);
  return final_cost;
}


/* =============================================================================================
 * grid-only LM, CentralGenericModel::FitToPixelDirections
 * (APP/models/central_generic.cc:44-83 state, :86-150 residual + Jacobian w.r.t. the local grid
 * updates, :153-225 cost function, :551-568 driver).  Residual of sample i (3 scalar residuals with
 * QuadraticLoss, LV/loss_functions.h:68-89): r = normalize(sum_c w_c P_c) - measurement; the
 * Jacobian pass evaluates the weights with the generated code's 15-digit literals
 * (central_generic_jacobians.cc:34-57), the cost-only pass goes through UnprojectFromGrid
 * (exact fractions, b_spline.h:49-60).  The derivative is restated analytically instead of through
 * the generated common-subexpression code:
 *     d r / d P_c = w_c (I - d d^T) / |v| ,    d P_c / d(update) = [t1_c t2_c]
 * (DirectionJacobianWrtLocalUpdate, direction_parametrization.h:57-70).
 * ============================================================================================= */
static void fit_residual_and_jacobian(int gw, const double* grid, const double* tang /*6 per grid point*/,
                                      double gpx, double gpy, const double* meas, double* r, int* idx /*32*/,
                                      double* J /*3 x 32 row-major*/) {
  int ix = (int)floor(gpx + 2), iy = (int)floor(gpy + 2);
  double fx = gpx + 2 - (ix - 3), fy = gpy + 2 - (iy - 3);
  axis_weights ax, ay;
  axis_weights_generated(fx, &ax);
  axis_weights_generated(fy, &ay);
  double wx[4] = {ax.a5 * ax.a3, ax.b, ax.c, ax.d8 * ax.d7};
  double wy[4] = {ay.a5 * ay.a3, ay.b, ay.c, ay.d8 * ay.d7};
  double v[3] = {0, 0, 0};
  double w[16];
  for (int y = 0; y < 4; ++y) {
    double row[3] = {0, 0, 0};
    for (int x = 0; x < 4; ++x) {
      int seq = (ix - 3 + x) + (iy - 3 + y) * gw;
      const double* P = grid + 3 * (size_t)seq;
      row[0] += wx[x] * P[0]; row[1] += wx[x] * P[1]; row[2] += wx[x] * P[2];
      w[x + 4 * y] = wx[x] * wy[y];
      idx[2 * (x + 4 * y)] = 2 * seq; idx[2 * (x + 4 * y) + 1] = 2 * seq + 1;
    }
    v[0] += wy[y] * row[0]; v[1] += wy[y] * row[1]; v[2] += wy[y] * row[2];
  }
  double inv = 1. / sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  double d[3] = {v[0] * inv, v[1] * inv, v[2] * inv};
  r[0] = d[0] - meas[0]; r[1] = d[1] - meas[1]; r[2] = d[2] - meas[2];
  for (int c = 0; c < 16; ++c) {
    int seq = idx[2 * c] / 2;
    const double* t1 = tang + 6 * (size_t)seq;
    const double* t2 = t1 + 3;
    double s = w[c] * inv;
    double dt1 = dot3(d, t1), dt2 = dot3(d, t2);
    for (int a = 0; a < 3; ++a) {
      J[a * 32 + 2 * c] = s * (t1[a] - d[a] * dt1);
      J[a * 32 + 2 * c + 1] = s * (t2[a] - d[a] * dt2);
    }
  }
}

/* Compute<compute_jacobians>: returns the cost; cost_vec (3n) gets 0.5 r^2 per scalar residual;
 * H (dof x dof row-major, upper triangle written) and b (dof) are zeroed and filled when H != NULL. */
double orc_fit_grid_pass(int32_t gw, int32_t gh, const double* grid, int64_t n, const double* grid_points,
                         const double* directions, double* H, double* b, double* cost_vec) {
  const int dof = 2 * gw * gh;
  double cost = 0;
  double* tang = NULL;
  if (H) {
    memset(H, 0, (size_t)dof * dof * sizeof(double));
    memset(b, 0, (size_t)dof * sizeof(double));
    tang = (double*)malloc(6 * (size_t)gw * gh * sizeof(double));
    for (int g = 0; g < gw * gh; ++g) orc_tangents(grid + 3 * (size_t)g, tang + 6 * (size_t)g, tang + 6 * (size_t)g + 3);
  }
  for (int64_t i = 0; i < n; ++i) {
    double r[3];
    if (H) {
      int idx[32]; double J[96];
      fit_residual_and_jacobian(gw, grid, tang, grid_points[2 * i], grid_points[2 * i + 1], directions + 3 * i, r, idx, J);
      for (int a = 0; a < 3; ++a) {   /* AddResidualWithJacobian(scalar residual, indices, row): H += J^T J (upper), b += J^T r */
        const double* Ja = J + a * 32;
        for (int p = 0; p < 32; ++p) {
          b[idx[p]] += Ja[p] * r[a];
          for (int q = p; q < 32; ++q) H[(size_t)idx[p] * dof + idx[q]] += Ja[p] * Ja[q];
        }
      }
    } else {
      double d[3];
      orc_bspline_surface(grid, gw, gh, 3, grid_points[2 * i], grid_points[2 * i + 1], d);   /* UnprojectFromGrid */
      double nrm = sqrt(dot3(d, d));
      for (int a = 0; a < 3; ++a) r[a] = d[a] / nrm - directions[3 * i + a];
    }
    for (int a = 0; a < 3; ++a) { double c = 0.5 * r[a] * r[a]; cost += c; if (cost_vec) cost_vec[3 * i + a] = c; }
  }
  free(tang);
  return cost;
}

/* DirectionGridStateWithLocalUpdates::operator-= (central_generic.cc:65-80) */
void orc_fit_grid_apply_update(int32_t gw, int32_t gh, const double* grid_in, const double* x, double* grid_out) {
  for (int g = 0; g < gw * gh; ++g) {
    double t1[3], t2[3], d[3] = {grid_in[3 * g], grid_in[3 * g + 1], grid_in[3 * g + 2]};
    orc_tangents(d, t1, t2);
    apply_local_update_to_direction(d, t1, t2, -x[2 * g], -x[2 * g + 1]);
    grid_out[3 * g] = d[0]; grid_out[3 * g + 1] = d[1]; grid_out[3 * g + 2] = d[2];
  }
}

/* FitToPixelDirectionsImpl (:551-568): LMOptimizer::Optimize(max_iteration_count, max_lm_attempts = 10,
 * init_lambda = -1, init_lambda_factor = 0.001f) on the dense system (LV/lm_optimizer.h:629-991, dense
 * solve x = H.selfadjointView<Upper>().ldlt().solve(b)).  report4 = {initial cost, final cost,
 * iterations performed, final lambda}. */
void orc_fit_grid_to_points(int32_t gw, int32_t gh, double* grid, int64_t n, const double* grid_points,
                            const double* directions, int32_t max_iteration_count, double* report4) {
  const int dof = 2 * gw * gh, max_lm_attempts = 10;
  const double init_lambda_factor = (double)0.001f;
  double* H = (double*)malloc((size_t)dof * dof * sizeof(double));
  double* b = (double*)malloc((size_t)dof * sizeof(double));
  double* x = (double*)malloc((size_t)dof * sizeof(double));
  double* diag = (double*)malloc((size_t)dof * sizeof(double));
  double* cv = (double*)malloc(3 * (size_t)n * sizeof(double));
  double* tv = (double*)malloc(3 * (size_t)n * sizeof(double));
  double* test_grid = (double*)malloc(3 * (size_t)gw * gh * sizeof(double));
  double lambda = -1, last_cost = 0, initial = 0;
  int performed = 0;
  for (int iteration = 0; iteration < max_iteration_count; ++iteration) {
    last_cost = orc_fit_grid_pass(gw, gh, grid, n, grid_points, directions, H, b, cv);
    if (iteration == 0) initial = last_cost;
    if (last_cost == 0) break;
    if (iteration == 0) {
      double s = 0;
      for (int i = 0; i < dof; ++i) s += H[(size_t)i * dof + i];
      lambda = init_lambda_factor * s / dof;
    }
    for (int i = 0; i < dof; ++i) diag[i] = H[(size_t)i * dof + i];
    int applied = 0;
    for (int lm = 0; lm < max_lm_attempts; ++lm) {
      for (int i = 0; i < dof; ++i) H[(size_t)i * dof + i] = diag[i] + lambda;
      orc_ldlt_solve_upper(H, dof, b, x);
      if (x[0] != x[0]) { lambda = 2.f * lambda; continue; }
      orc_fit_grid_apply_update(gw, gh, grid, x, test_grid);
      double test_cost = orc_fit_grid_pass(gw, gh, test_grid, n, grid_points, directions, NULL, NULL, tv);
      if (cost_is_smaller_than(tv, cv, 3 * n)) {
        memcpy(grid, test_grid, 3 * (size_t)gw * gh * sizeof(double));
        lambda = 0.5f * lambda;
        applied = 1; performed += 1; last_cost = test_cost;
        break;
      }
      lambda = 2.f * lambda;
    }
    if (!applied || last_cost == 0) break;
  }
  if (report4) { report4[0] = initial; report4[1] = last_cost; report4[2] = performed; report4[3] = lambda; }
  free(H); free(b); free(x); free(diag); free(cv); free(tv); free(test_grid);
}
