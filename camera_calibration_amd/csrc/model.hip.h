// Device-side generic camera models for gfx950: cubic B-spline unprojection (value and pixel
// Jacobian) and the iterative 2-unknown Levenberg-Marquardt projection.
//
// What is computed follows the reference (file:line relative to the reference tree,
// APP = applications/camera_calibration/src/camera_calibration):
//   Unproject                 APP/models/central_generic.h:97-105, noncentral_generic.h:100-115, b_spline.h:45-104
//   UnprojectWithJacobian     APP/models/central_generic.cc:521-549, central_generic_jacobians.cc:320-448,
//                             noncentral_generic.cc:266-293, noncentral_generic_jacobians.cc:31-205
//   ProjectWithInitialEstimate APP/models/central_generic.cc:433-519, noncentral_generic.cc:156-264
// How it is computed is written for one lane = one projection: the 4x4 control patch is gathered
// through L1/L2 (neighbouring lanes work on the same observation, hence the same cache lines), all
// math is FMA-contracted fp64, and a "substituted" control point implements the reference's
// in-place perturbation of one grid vector (central_grid.h:219-230) without touching the grid.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cba {

constexpr int kCentral = 0;
constexpr int kNoncentral = 1;

struct CamDev {
  int model_type;
  int gw, gh;
  int min_x, min_y, max_x, max_y;
  double gsx, gsy;        // grid_w - 3, grid_h - 3
  double span_x, span_y;  // max + 1 - min
  double jscale_x, jscale_y;  // PixelScaleToGridScale: (gw-3.f)/span evaluated in fp32 (central_grid.h:156-161)
  const double* grid;     // 3G direction grid [+ 3G point grid for the non-central model]
  const double* tangents; // 6G (t1,t2) of the direction grid (Jacobian pass only)
  int intr_offset;        // first dense column of this camera's intrinsics
  int params_per_point;   // 2 or 5
  const int* gperm;       // control point (gx + gy*gw) -> rank in the engine's tiled unknown order (null = identity)
};

// Dense column of parameter d of control point `seq` of this camera (engine-internal order).
__device__ __forceinline__ int grid_column(const CamDev& c, int seq, int d) {
  return c.intr_offset + c.params_per_point * (c.gperm ? c.gperm[seq] : seq) + d;
}

// A control point replaced by a perturbed copy (index < 0: none).
struct Subst {
  int index;
  double d[3];
  double o[3];
};

__device__ __forceinline__ bool in_calibrated_area(const CamDev& c, double x, double y) {
  return x >= c.min_x && y >= c.min_y && x < c.max_x + 1 && y < c.max_y + 1;  // camera_model.h:159-162
}
__device__ __forceinline__ void pixel_to_grid(const CamDev& c, double x, double y, double& gx, double& gy) {
  gx = 1.0 + c.gsx * (x - c.min_x) / c.span_x;  // central_grid.h:150-154
  gy = 1.0 + c.gsy * (y - c.min_y) / c.span_y;
}
__device__ __forceinline__ void normalize3(double& x, double& y, double& z) {
  double s = x * x + y * y + z * z;
  if (s > 0) {
    double n = sqrt(s);
    x /= n; y /= n; z /= n;
  }
}

// ComputeTangentsForDirectionOrLine, APP/local_parametrizations/line_parametrization.h:54-60
__device__ __forceinline__ void tangents_of(const double* d, double* t1, double* t2) {
  double cx, cy, cz;
  if (fabs(d[0]) > (double)0.9f) {  // d x e_y
    cx = -d[2]; cy = 0.0; cz = d[0];
  } else {                           // d x e_x
    cx = 0.0; cy = d[2]; cz = -d[1];
  }
  normalize3(cx, cy, cz);
  t1[0] = cx; t1[1] = cy; t1[2] = cz;
  t2[0] = d[1] * cz - d[2] * cy;
  t2[1] = d[2] * cx - d[0] * cz;
  t2[2] = d[0] * cy - d[1] * cx;
}

// exact-fraction cubic weights of b_spline.h:49-60 (f in [3,4))
__device__ __forceinline__ void weights_value(double f, double* w) {
  double fd = f - 3.0, fa = f - 4.0;
  w[3] = 1. / 6. * fd * fd * fd;
  w[2] = -1. / 2. * f * f * f + 5 * f * f - 16 * f + 50. / 3.;
  w[1] = 1. / 2. * f * f * f - 11. / 2. * f * f + (39. / 2.) * f - 131. / 6.;
  w[0] = -1. / 6. * fa * fa * fa;
}
// value / derivative weights in the factorisation of the generated Jacobian code
// (central_generic_jacobians.cc:323-336, 401-408) incl. its 15-digit decimal literals
__device__ __forceinline__ void weights_jac(double f, double* w, double* dw) {
  double t4 = 0.166666666666667 * f;
  double a5 = -t4 + 0.666666666666667, a3 = (f - 4) * (f - 4);
  double d8 = t4 - 0.5, d7 = (f - 3) * (f - 3);
  double f2 = f * f, h = 0.5 * f * f2;
  w[0] = a5 * a3;
  w[1] = 19.5 * f - 5.5 * f2 + h - 21.8333333333333;
  w[2] = -16 * f + 5 * f2 - h + 16.6666666666667;
  w[3] = d8 * d7;
  double t80 = 1.5 * f2;
  dw[0] = -0.166666666666667 * a3 + a5 * (2 * f - 8);
  dw[1] = -11.0 * f + t80 + 19.5;
  dw[2] = 10 * f - t80 - 16;
  dw[3] = 0.166666666666667 * d7 + d8 * (2 * f - 6);
}

template <int MODEL>
struct Line {
  double d[3];
  double o[3];
};

// Gathers control point (gx,gy) honouring the substitution.
template <int MODEL>
__device__ __forceinline__ void load_ctrl(const CamDev& c, const Subst& s, int seq, double* d, double* o) {
  if (seq == s.index) {
    d[0] = s.d[0]; d[1] = s.d[1]; d[2] = s.d[2];
    if (MODEL == kNoncentral) { o[0] = s.o[0]; o[1] = s.o[1]; o[2] = s.o[2]; }
    return;
  }
  const double* g = c.grid + 3 * (size_t)seq;
  d[0] = g[0]; d[1] = g[1]; d[2] = g[2];
  if (MODEL == kNoncentral) {
    const double* p = c.grid + 3 * (size_t)c.gw * c.gh + 3 * (size_t)seq;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
  }
}

// One 4x4 control patch staged in LDS (finite-difference kernel: the 35 / 83 task lanes of an observation all evaluate the
// spline on the observation's own patch, a few times each): p[(r * 4 + q) * DIM + k], DIM = 3 (direction) or 6
// (direction, origin); (fx, fy) = grid coordinates of p[0].  An evaluation whose patch is a different one (the pixel
// moved across a cell boundary) gathers from global memory as before.
typedef const __attribute__((address_space(3))) double* lds_cdouble_ptr;   // explicit LDS pointer: ds_read, never a flat load
template <int MODEL>
struct StagedPatch {
  lds_cdouble_ptr p;      // the observation's patch
  lds_cdouble_ptr sub;    // this lane's substituted control point (DIM doubles), also in LDS: the substitution is a select
  int fx, fy;             // on the 32-bit LDS ADDRESS followed by unconditional ds_reads (a select on the loaded values
};                        // was turned into 16 branches per evaluation by the compiler)
template <int MODEL, bool LDS>
__device__ __forceinline__ void ctrl_point(const CamDev& c, const Subst& s, lds_cdouble_ptr stage, lds_cdouble_ptr sub_p, int r, int q, int seq,
                                           double* d, double* o) {
  if (LDS) {
    constexpr int DIM = (MODEL == kCentral) ? 3 : 6;
    lds_cdouble_ptr g = (seq == s.index) ? sub_p : stage + (r * 4 + q) * DIM;
    d[0] = g[0]; d[1] = g[1]; d[2] = g[2];
    if (MODEL == kNoncentral) { o[0] = g[3]; o[1] = g[4]; o[2] = g[5]; }
  } else {
    load_ctrl<MODEL>(c, s, seq, d, o);
  }
}

// Unproject: exact-fraction weights, direction normalised in fp64. Returns false outside the rectangle.
template <int MODEL, bool LDS>
__device__ __forceinline__ void unproject_eval(const CamDev& c, const Subst& s, lds_cdouble_ptr stage, lds_cdouble_ptr sub_p, int ix, int iy, double gx, double gy,
                                               double* dir, double* org) {
  double wx[4], wy[4];
  weights_value(gx - (ix - 3), wx);
  weights_value(gy - (iy - 3), wy);
  double vd[3] = {0, 0, 0}, vo[3] = {0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double rd[3] = {0, 0, 0}, ro[3] = {0, 0, 0};
    int rowbase = (iy - 3 + r) * c.gw + (ix - 3);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double d[3], o[3];
      ctrl_point<MODEL, LDS>(c, s, stage, sub_p, r, q, rowbase + q, d, o);
      rd[0] += wx[q] * d[0]; rd[1] += wx[q] * d[1]; rd[2] += wx[q] * d[2];
      if (MODEL == kNoncentral) { ro[0] += wx[q] * o[0]; ro[1] += wx[q] * o[1]; ro[2] += wx[q] * o[2]; }
    }
    vd[0] += wy[r] * rd[0]; vd[1] += wy[r] * rd[1]; vd[2] += wy[r] * rd[2];
    if (MODEL == kNoncentral) { vo[0] += wy[r] * ro[0]; vo[1] += wy[r] * ro[1]; vo[2] += wy[r] * ro[2]; }
  }
  normalize3(vd[0], vd[1], vd[2]);
  dir[0] = vd[0]; dir[1] = vd[1]; dir[2] = vd[2];
  if (MODEL == kNoncentral) { org[0] = vo[0]; org[1] = vo[1]; org[2] = vo[2]; }
}
template <int MODEL>
__device__ bool unproject(const CamDev& c, const Subst& s, double x, double y, double* dir, double* org) {
  if (!in_calibrated_area(c, x, y)) return false;
  double gx, gy;
  pixel_to_grid(c, x, y, gx, gy);
  gx += 2; gy += 2;
  int ix = (int)gx, iy = (int)gy;
  unproject_eval<MODEL, false>(c, s, (lds_cdouble_ptr)0, (lds_cdouble_ptr)0, ix, iy, gx, gy, dir, org);
  return true;
}
template <int MODEL>
__device__ __forceinline__ bool unproject_staged(const CamDev& c, const Subst& s, const StagedPatch<MODEL>& st, double x, double y,
                                                 double* dir, double* org, bool& miss) {
  if (!in_calibrated_area(c, x, y)) return false;
  double gx, gy;
  pixel_to_grid(c, x, y, gx, gy);
  gx += 2; gy += 2;
  int ix = (int)gx, iy = (int)gy;
  if (ix - 3 != st.fx || iy - 3 != st.fy) { miss = true; return false; }
  unproject_eval<MODEL, true>(c, s, st.p, st.sub, ix, iy, gx, gy, dir, org);
  return true;
}

// UnprojectWithJacobian. jd = d direction / d pixel (3x2), jo = d origin / d pixel (3x2, non-central).
template <int MODEL, bool LDS>
__device__ __forceinline__ void unproject_jac_eval(const CamDev& c, const Subst& s, lds_cdouble_ptr stage, lds_cdouble_ptr sub_p, int ix, int iy, double gx, double gy,
                                                   double* dir, double* org, double* jd, double* jo) {
  double wx[4], dwx[4], wy[4], dwy[4];
  weights_jac(gx - (ix - 3), wx, dwx);
  weights_jac(gy - (iy - 3), wy, dwy);
  double v[3] = {0, 0, 0}, vx[3] = {0, 0, 0}, vy[3] = {0, 0, 0};
  double u[3] = {0, 0, 0}, ux[3] = {0, 0, 0}, uy[3] = {0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double R[3] = {0, 0, 0}, dR[3] = {0, 0, 0}, Q[3] = {0, 0, 0}, dQ[3] = {0, 0, 0};
    int rowbase = (iy - 3 + r) * c.gw + (ix - 3);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double d[3], o[3];
      ctrl_point<MODEL, LDS>(c, s, stage, sub_p, r, q, rowbase + q, d, o);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        R[k] += wx[q] * d[k];
        dR[k] += dwx[q] * d[k];
        if (MODEL == kNoncentral) { Q[k] += wx[q] * o[k]; dQ[k] += dwx[q] * o[k]; }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v[k] += wy[r] * R[k];
      vx[k] += wy[r] * dR[k];
      vy[k] += dwy[r] * R[k];
      if (MODEL == kNoncentral) { u[k] += wy[r] * Q[k]; ux[k] += wy[r] * dQ[k]; uy[k] += dwy[r] * Q[k]; }
    }
  }
  double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  double inv, inv3;
  if (MODEL == kCentral) {
    inv = 1.0 / sqrt(sq);
    inv3 = inv * inv * inv;
  } else {
    // the generated non-central code normalises with `1 / sqrtf(term75)` (noncentral_generic_jacobians.cc:110): square
    // root AND division in fp32 (int / float); :158-159 cubes the fp32 root in fp64
    // sqrtf and the fp32 `/` are IEEE correctly rounded in HIP's default mode; the __fsqrt_rn / __fdiv_rn intrinsics map to the
    // approximate native instructions in this toolchain and differed from the reference in 3.5 % of the pixels
    float sf = sqrtf((float)sq);
    inv = (double)(1.0f / sf);
    double t = (double)sf;
    inv3 = 1.0 / (t * t * t);
  }
  double sx = inv3 * (v[0] * vx[0] + v[1] * vx[1] + v[2] * vx[2]);
  double sy = inv3 * (v[0] * vy[0] + v[1] * vy[1] + v[2] * vy[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dir[k] = v[k] * inv;
    jd[2 * k + 0] = (inv * vx[k] - v[k] * sx) * c.jscale_x;
    jd[2 * k + 1] = (inv * vy[k] - v[k] * sy) * c.jscale_y;
    if (MODEL == kNoncentral) {
      org[k] = u[k];
      jo[2 * k + 0] = ux[k] * c.jscale_x;
      jo[2 * k + 1] = uy[k] * c.jscale_y;
    }
  }
}
template <int MODEL>
__device__ bool unproject_jac(const CamDev& c, const Subst& s, double x, double y, double* dir, double* org,
                              double* jd, double* jo) {
  if (!in_calibrated_area(c, x, y)) return false;
  double gx, gy;
  pixel_to_grid(c, x, y, gx, gy);
  gx += 2; gy += 2;
  int ix = (int)floor(gx), iy = (int)floor(gy);
  unproject_jac_eval<MODEL, false>(c, s, (lds_cdouble_ptr)0, (lds_cdouble_ptr)0, ix, iy, gx, gy, dir, org, jd, jo);
  return true;
}
template <int MODEL>
__device__ __forceinline__ bool unproject_jac_staged(const CamDev& c, const Subst& s, const StagedPatch<MODEL>& st, double x, double y,
                                                     double* dir, double* org, double* jd, double* jo, bool& miss) {
  if (!in_calibrated_area(c, x, y)) return false;
  double gx, gy;
  pixel_to_grid(c, x, y, gx, gy);
  gx += 2; gy += 2;
  int ix = (int)floor(gx), iy = (int)floor(gy);
  if (ix - 3 != st.fx || iy - 3 != st.fy) { miss = true; return false; }
  unproject_jac_eval<MODEL, true>(c, s, st.p, st.sub, ix, iy, gx, gy, dir, org, jd, jo);
  return true;
}

// TangentsJacobianWrtLineDirection (line_parametrization.h:62-105), applied to the 3x2 direction
// Jacobian: returns d t1 / d pixel and d t2 / d pixel (each 3x2).
__device__ __forceinline__ void tangent_derivs(const double* d, const double* jd, double* dt1, double* dt2) {
  double TJ[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) TJ[i] = 0.0;
  if (fabs(d[0]) > (double)0.9f) {
    double t0 = d[0] * d[0], t1 = d[2] * d[2], t7 = 1.0 / sqrt(t0 + t1), t3 = t7 * t7 * t7;
    double t4 = d[0] * d[2] * t3, t5 = t0 * t3, t6 = t1 * t3, t8 = d[0] * t7, t9 = -d[1] * t4, t10 = d[2] * t7;
    TJ[0] = t4; TJ[2] = -t5; TJ[6] = t6; TJ[8] = -t4;
    TJ[9] = d[1] * t6; TJ[10] = t8; TJ[11] = t9; TJ[12] = -t8; TJ[14] = -t10;
    TJ[15] = t9; TJ[16] = t10; TJ[17] = d[1] * t5;
  } else {
    double t0 = d[1] * d[1], t1 = d[2] * d[2], t7 = 1.0 / sqrt(t0 + t1), t3 = t7 * t7 * t7;
    double t4 = d[1] * d[2] * t3, t5 = t0 * t3, t6 = t1 * t3, t8 = d[1] * t7, t9 = d[2] * t7, t10 = -d[0] * t4;
    TJ[4] = -t4; TJ[5] = t5; TJ[7] = -t6; TJ[8] = t4;
    TJ[10] = -t8; TJ[11] = -t9; TJ[12] = t8; TJ[13] = d[0] * t6; TJ[14] = t10;
    TJ[15] = t9; TJ[16] = t10; TJ[17] = d[0] * t5;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      dt1[2 * r + q] = TJ[r * 3 + 0] * jd[0 + q] + TJ[r * 3 + 1] * jd[2 + q] + TJ[r * 3 + 2] * jd[4 + q];
      dt2[2 * r + q] = TJ[(3 + r) * 3 + 0] * jd[0 + q] + TJ[(3 + r) * 3 + 1] * jd[2 + q] + TJ[(3 + r) * 3 + 2] * jd[4 + q];
    }
}

// ---- pieces of the iterative projection, shared by the one-lane loop (project_target) and the 16-lanes-per-observation
// ---- straggler kernel (kernels_obs.hip: k_base_project_slow), so that both evaluate the same expressions ----
// cost and 2 x 2 normal equations of one LM iteration from UnprojectWithJacobian's outputs at the current pixel
template <int MODEL>
__device__ __forceinline__ void projection_normal_equations(const double* dir, const double* org, const double* jd, const double* jo,
                                                            const double* target, double& cost, double& H00, double& H01,
                                                            double& H11, double& b0, double& b1) {
  if (MODEL == kCentral) {
    double dx = dir[0] - target[0], dy = dir[1] - target[1], dz = dir[2] - target[2];
    cost = dx * dx + dy * dy + dz * dz;
    H00 = jd[0] * jd[0] + jd[2] * jd[2] + jd[4] * jd[4];
    H01 = jd[0] * jd[1] + jd[2] * jd[3] + jd[4] * jd[5];
    H11 = jd[1] * jd[1] + jd[3] * jd[3] + jd[5] * jd[5];
    b0 = dx * jd[0] + dy * jd[2] + dz * jd[4];
    b1 = dx * jd[1] + dy * jd[3] + dz * jd[5];
  } else {
    double t1[3], t2[3];
    tangents_of(dir, t1, t2);
    double p[3] = {org[0] - target[0], org[1] - target[1], org[2] - target[2]};
    double d1 = t1[0] * p[0] + t1[1] * p[1] + t1[2] * p[2];
    double d2 = t2[0] * p[0] + t2[1] * p[1] + t2[2] * p[2];
    double dt1[6], dt2[6];
    tangent_derivs(dir, jd, dt1, dt2);
    double R[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      R[q] = p[0] * dt1[q] + p[1] * dt1[2 + q] + p[2] * dt1[4 + q] + t1[0] * jo[q] + t1[1] * jo[2 + q] + t1[2] * jo[4 + q];
      R[2 + q] = p[0] * dt2[q] + p[1] * dt2[2 + q] + p[2] * dt2[4 + q] + t2[0] * jo[q] + t2[1] * jo[2 + q] + t2[2] * jo[4 + q];
    }
    cost = d1 * d1 + d2 * d2;
    H00 = R[0] * R[0] + R[2] * R[2];
    H01 = R[0] * R[1] + R[2] * R[3];
    H11 = R[1] * R[1] + R[3] * R[3];
    b0 = d1 * R[0] + d2 * R[2];
    b1 = d1 * R[1] + d2 * R[3];
  }
}
// the damped 2 x 2 solve and the clamped candidate pixel (central_generic.cc:470-481)
__device__ __forceinline__ void projection_candidate(const CamDev& c, double H00, double H01, double H11, double b0, double b1,
                                                     double lambda, double px, double py, double& tx, double& ty) {
  const double lo_x = (double)c.min_x, hi_x = c.max_x + 0.999, lo_y = (double)c.min_y, hi_y = c.max_y + 0.999;
  double H00l = H00 + lambda, H11l = H11 + lambda;
  double x1 = (b1 - H01 / H00l * b0) / (H11l - H01 * H01 / H00l);
  double x0 = (b0 - H01 * x1) / H00l;
  double cx = px - x0, cy = py - x1;
  double mx = (cx < hi_x) ? cx : hi_x;  // std::min(hi, v)
  tx = (lo_x < mx) ? mx : lo_x;         // std::max(lo, .)
  double my = (cy < hi_y) ? cy : hi_y;
  ty = (lo_y < my) ? my : lo_y;
}
// cost of a candidate from Unproject's outputs
template <int MODEL>
__device__ __forceinline__ double projection_test_cost(const double* td, const double* to, const double* target) {
  if (MODEL == kCentral) {
    double ex = td[0] - target[0], ey = td[1] - target[1], ez = td[2] - target[2];
    return ex * ex + ey * ey + ez * ez;
  } else {
    double t1[3], t2[3];
    tangents_of(td, t1, t2);
    double p[3] = {to[0] - target[0], to[1] - target[1], to[2] - target[2]};
    double e1 = t1[0] * p[0] + t1[1] * p[1] + t1[2] * p[2];
    double e2 = t2[0] * p[0] + t2[1] * p[1] + t2[2] * p[2];
    return e1 * e1 + e2 * e2;
  }
}

// Iterative projection. target = unit direction (central) or local point (non-central).
// px,py: in = initial estimate (must lie in the calibrated area), out = result.
// Returns true iff converged (squared error < 1e-12), as the reference does.
// STG: the spline is evaluated on the patch staged in LDS; if an iterate needs another patch, *miss is set and the call
// returns false -- the caller then repeats the whole projection on the gather path (rare: a pixel within the last LM
// step of a cell boundary).
// max_outer < 100: give up after that many outer iterations with *capped = true (the caller hands the observation to the
// straggler kernel, which runs the complete loop); nothing of the reference's 100-iteration semantics is cut short.
template <int MODEL, bool STG = false>
__device__ bool project_target(const CamDev& c, const Subst& s, const double* target, double& px, double& py,
                               const StagedPatch<MODEL>* st = nullptr, bool* miss = nullptr, int max_outer = 100,
                               bool* capped = nullptr) {
  constexpr double kEpsilon = 1e-12;
  double lambda = -1.0;
  for (int it = 0; it < max_outer; ++it) {
    double dir[3], org[3], jd[6], jo[6];
    bool inside;
    if (STG) inside = unproject_jac_staged<MODEL>(c, s, *st, px, py, dir, org, jd, jo, *miss);
    else inside = unproject_jac<MODEL>(c, s, px, py, dir, org, jd, jo);
    if (!inside) return false;  // CHECK() in the reference
    double cost, H00, H01, H11, b0, b1;
    projection_normal_equations<MODEL>(dir, org, jd, jo, target, cost, H00, H01, H11, b0, b1);
    if (lambda < 0) lambda = 0.01 * 0.5 * (H00 + H11);
    bool accepted = false;
    for (int lm = 0; lm < 10; ++lm) {
      double tx, ty;
      projection_candidate(c, H00, H01, H11, b0, b1, lambda, px, py, tx, ty);
      double test_cost = INFINITY;
      double td[3], to[3];
      bool tin;
      if (STG) { tin = unproject_staged<MODEL>(c, s, *st, tx, ty, td, to, *miss); if (*miss) return false; }
      else tin = unproject<MODEL>(c, s, tx, ty, td, to);
      if (tin) test_cost = projection_test_cost<MODEL>(td, to, target);
      if (test_cost < cost) {
        lambda *= 0.5;
        px = tx; py = ty;
        accepted = true;
        break;
      } else {
        lambda *= 2.0;
      }
    }
    if (!accepted) return cost < kEpsilon;
    if (cost < kEpsilon) return true;
  }
  if (max_outer < 100 && capped) *capped = true;
  return false;
}

// ProjectWithInitialEstimate(local_point): the central model normalises first (central_grid.h:86-88).
template <int MODEL, bool STG = false>
__device__ __forceinline__ bool project_point(const CamDev& c, const Subst& s, const double* local, double& px, double& py,
                                              const StagedPatch<MODEL>* st = nullptr, bool* miss = nullptr, int max_outer = 100,
                                              bool* capped = nullptr) {
  if (MODEL == kCentral) {
    double d[3] = {local[0], local[1], local[2]};
    normalize3(d[0], d[1], d[2]);
    return project_target<MODEL, STG>(c, s, d, px, py, st, miss, max_outer, capped);
  } else {
    return project_target<MODEL, STG>(c, s, local, px, py, st, miss, max_outer, capped);
  }
}

__device__ __forceinline__ void center_pixel(const CamDev& c, double& px, double& py) {
  px = 0.5 * (c.min_x + c.max_x + 1);
  py = 0.5 * (c.min_y + c.max_y + 1);
}

}  // namespace cba
