// Per-observation kernels of the bundle-adjustment engine (gfx950).
//
//  k_compose_poses   image_tr_global = camera_tr_rig[c] * rig_tr_global[i]   (joint_optimization.cc:277-280)
//  k_tangents        ComputeTangentsImage                                      (joint_optimization.cc:229-238)
//  k_base_project    AddReprojectionResidual, residual part                    (joint_optimization.cc:321-347)
//  k_base_project_slow  the same for the observations whose projection runs long (failing projections: the reference's
//                    whole 100 x 10-iteration budget, twice), 16 lanes per observation
//  k_fd_tasks        the 3 + K finite-difference re-projections                (joint_optimization.cc:357-372,
//                                                                               central_grid.h:187-245,
//                                                                               noncentral_generic.h:224-283)
//  k_assemble        analytic chain to pose / rig / point Jacobians            (joint_optimization.cc:379-438)
//  k_accumulate      AddResidualWithJacobian -> block-sparse JtJ / Jtr         (lm_optimizer_jtj_accumulator_base.h:287-401,
//                                                                               lm_optimizer_update_accumulator.h:181-322)
//  k_reduce_costs    cost sums and CostIsSmallerThan                           (lm_optimizer.h:993-1011)
//  k_update_*        JointOptimizationState::operator-=                        (joint_optimization.cc:172-214)
//
//  k_accumulate_strips / k_accumulate_cells   the pose x dense strips and the grid x grid block of JtJ (+ the grid part of
//                    Jtr), summed per (imageset, column band) / per control-patch cell before they reach HBM
//
// Parallel decomposition (MI355X-first, not the reference's single loop): the packed observation
// array is streamed coalesced; every finite-difference projection is its own lane (35 or 83 lanes
// per observation, the workgroup stages the observations' 4x4 control patches in LDS); the
// outer product of one observation is spread over the 64 lanes of a wavefront and lands in HBM with
// hardware fp64 atomics (global_atomic_add_f64), except for the terms that many observations share
// (strips, cells), which are summed on chip first.
#include "cba_internal.h"
#include <algorithm>
#include <mutex>

namespace cba {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void quat_mul(const double* a, const double* b, double* o) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}
// Eigen's quaternion * vector (v + w*uv + q x uv with uv = 2 q x v)
__device__ __forceinline__ void quat_rotate(const double* q, const double* v, double* o) {
  double ux = 2 * (q[2] * v[2] - q[3] * v[1]);
  double uy = 2 * (q[3] * v[0] - q[1] * v[2]);
  double uz = 2 * (q[1] * v[1] - q[2] * v[0]);
  o[0] = v[0] + q[0] * ux + (q[2] * uz - q[3] * uy);
  o[1] = v[1] + q[0] * uy + (q[3] * ux - q[1] * uz);
  o[2] = v[2] + q[0] * uz + (q[1] * uy - q[2] * ux);
}
// rotation matrix of a unit quaternion (Eigen toRotationMatrix form)
__device__ __forceinline__ void quat_to_matrix(const double* q, double* R) {
  double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
  double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
  double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
  double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// un-normalised polynomial rotation R(q) differentiated by the analytic Jacobians
// (joint_optimization_jacobians.h:40-118)
__device__ __forceinline__ void poly_rotation(const double* q, double* R) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * y * y - 2 * z * z; R[1] = 2 * x * y - 2 * w * z; R[2] = 2 * x * z + 2 * w * y;
  R[3] = 2 * x * y + 2 * w * z; R[4] = 1 - 2 * x * x - 2 * z * z; R[5] = 2 * y * z - 2 * w * x;
  R[6] = 2 * x * z - 2 * w * y; R[7] = 2 * y * z + 2 * w * x; R[8] = 1 - 2 * x * x - 2 * y * y;
}
// d(R(q) v)/dq folded with QuaternionJacobianWrtLocalUpdate (quaternion_parametrization.h:63-72):
// M = d(R(q) v)/dq [3x4] * dq/dupdate [4x3]  -> 3x3
__device__ __forceinline__ void rotated_point_wrt_update(const double* q, const double* v, double* M) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double a = v[0], b = v[1], c = v[2];
  double D[12];
  D[0] = 2 * y * c - 2 * z * b;  D[1] = 2 * y * b + 2 * z * c;             D[2] = -4 * y * a + 2 * x * b + 2 * w * c;   D[3] = -4 * z * a - 2 * w * b + 2 * x * c;
  D[4] = 2 * z * a - 2 * x * c;  D[5] = 2 * y * a - 4 * x * b - 2 * w * c; D[6] = 2 * x * a + 2 * z * c;                D[7] = 2 * w * a - 4 * z * b + 2 * y * c;
  D[8] = -2 * y * a + 2 * x * b; D[9] = 2 * z * a + 2 * w * b - 4 * x * c; D[10] = -2 * w * a + 2 * z * b - 4 * y * c;  D[11] = 2 * x * a + 2 * y * b;
  // Q rows (w,x,y,z): [-x -y -z; w z -y; -z w x; y -x w]
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double* d = D + 4 * r;
    M[3 * r + 0] = -d[0] * x + d[1] * w - d[2] * z + d[3] * y;
    M[3 * r + 1] = -d[0] * y + d[1] * z + d[2] * w - d[3] * x;
    M[3 * r + 2] = -d[0] * z - d[1] * y + d[2] * x + d[3] * w;
  }
}
__device__ __forceinline__ double huber_cost_sq(double sq) { return sq < 1.0 ? 0.5 * sq : (sqrt(sq) - 0.5); }
__device__ __forceinline__ double huber_weight_sq(double sq) { return sq < 1.0 ? 1.0 : 1.0 / sqrt(sq); }

// ------------------------------------------------------------------------------------------------
__global__ void k_compose_poses(const double* __restrict__ rig, const double* __restrict__ camrig, int N, int C,
                                double* __restrict__ itg) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * C) return;
  int i = t / C, c = t % C;
  const double* a = camrig + 7 * c;   // camera_tr_rig[c]
  const double* b = rig + 7 * (size_t)i;  // rig_tr_global[i]
  // Sophus SE3 product (se3.hpp:203-207) + renormalisation (so3.hpp:215-232)
  double q[4], tr[3];
  quat_rotate(a, b + 4, tr);
  quat_mul(a, b, q);
  double sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (sn != 1.0) {
    double s = 2.0 / (1.0 + sn);
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
  }
  double* o = itg + 16 * (size_t)t;
  o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
  o[4] = a[4] + tr[0]; o[5] = a[5] + tr[1]; o[6] = a[6] + tr[2];
  quat_to_matrix(q, o + 7);
}
int launch_compose_poses(const DevState& st, int N, int C, double* itg, hipStream_t s) {
  int n = N * C;
  hipLaunchKernelGGL(k_compose_poses, dim3((n + 255) / 256), dim3(256), 0, s, st.rig_tr_global, st.camera_tr_rig, N, C, itg);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

__global__ void k_tangents(const double* __restrict__ grid, double* __restrict__ tang, int G) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  double d[3] = {grid[3 * g], grid[3 * g + 1], grid[3 * g + 2]};
  double t1[3], t2[3];
  tangents_of(d, t1, t2);
  double* o = tang + 6 * (size_t)g;
  o[0] = t1[0]; o[1] = t1[1]; o[2] = t1[2]; o[3] = t2[0]; o[4] = t2[1]; o[5] = t2[2];
}
int launch_tangents(const double* dir_grid, double* tang, int G, hipStream_t s) {
  hipLaunchKernelGGL(k_tangents, dim3((G + 255) / 256), dim3(256), 0, s, dir_grid, tang, G);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ------------------------------------------------------------------------------------------------
// residual pass: one lane per observation
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void local_point_of(const PassArgs& a, int64_t o, int cam, double* local) {
  const double* T = a.itg + 16 * ((size_t)a.obs_image[o] * a.n_cameras + cam);
  const double* p = a.points + 3 * (size_t)a.obs_point[o];
  double px = p[0], py = p[1], pz = p[2];
  local[0] = T[7] * px + T[8] * py + T[9] * pz + T[4];
  local[1] = T[10] * px + T[11] * py + T[12] * pz + T[5];
  local[2] = T[13] * px + T[14] * py + T[15] * pz + T[6];
}

// One lane per observation runs AddReprojectionResidual's projection: warm start, retry from the centre.  Almost every
// lane is done after 1-3 outer iterations, but a projection that FAILS runs the reference's whole budget first -- 100 outer
// iterations with up to 10 damping attempts each, twice (warm start and centre): ~600 B-spline evaluations, 1.5 ms for a
// single lane, and a pass cannot end before its slowest lane (0.4 % of the observations of the BASELINE configs fail from
// the perturbed initial state: points whose projection is pinned at the border of the calibrated area, or that settle in a
// local minimum next to it).  So a lane gives up after `outer_cap` (default 8) outer iterations of either attempt and puts its
// observation on the straggler list; k_base_project_slow then runs the COMPLETE procedure for the list with 16 lanes per
// observation.  Nothing of the 100 x 10 semantics is cut short, the long chains are only evaluated faster.

template <int MODEL>
__device__ __forceinline__ bool base_projection(const PassArgs& a, const CamDev& c, int64_t o, const double* local, int max_outer,
                                                bool& capped, double& px, double& py) {
  Subst none; none.index = -1;
  px = a.last_projection[2 * o]; py = a.last_projection[2 * o + 1];
  if (!in_calibrated_area(c, px, py) || px != px || py != py) center_pixel(c, px, py);
  capped = false;
  bool ok = project_point<MODEL>(c, none, local, px, py, nullptr, nullptr, max_outer, &capped);
  if (!ok && !capped) {
    center_pixel(c, px, py);
    ok = project_point<MODEL>(c, none, local, px, py, nullptr, nullptr, max_outer, &capped);
  }
  return ok;
}
__device__ __forceinline__ void store_base_projection(const PassArgs& a, int64_t o, bool ok, double px, double py,
                                                      double* __restrict__ cost_vec, double* __restrict__ pixels,
                                                      uint8_t* __restrict__ flags) {
  if (!ok) {
    cost_vec[o] = -1.0;   // AddInvalidResidual (lm_optimizer_update_accumulator.h:158-160)
    flags[o] = 0;
    return;
  }
  a.last_projection[2 * o] = px;
  a.last_projection[2 * o + 1] = py;
  pixels[2 * o] = px;
  pixels[2 * o + 1] = py;
  double rx = px - (double)a.obs_xy[2 * o], ry = py - (double)a.obs_xy[2 * o + 1];
  cost_vec[o] = huber_cost_sq(rx * rx + ry * ry);
  flags[o] = 1;
}

// defer_*: straggler list (device), its fill count, its capacity, and the per-observation "on the list" byte that the
// finite-difference launch of the same pass reads through PassArgs::skip.
template <int MODEL>
__global__ void __launch_bounds__(256) k_base_project(PassArgs a, double* __restrict__ cost_vec,
                                                      double* __restrict__ pixels, uint8_t* __restrict__ flags,
                                                      int* __restrict__ defer_list, int* __restrict__ defer_count, int defer_cap,
                                                      uint8_t* __restrict__ defer_skip, int outer_cap,
                                                      const uint8_t* __restrict__ fd_slow) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= a.n_obs) return;
  if (a.guard && *a.guard != 0) return;                     // the solve in front of this cost pass broke down: touch nothing
  const int cam = a.obs_camera[o];
  const CamDev c = a.cams[cam];
  if (c.model_type != MODEL) return;
  double local[3];
  local_point_of(a, o, cam, local);
  double px, py;
  bool capped;
  bool ok = base_projection<MODEL>(a, c, o, local, outer_cap, capped, px, py);
  // fd_slow (Jacobian pass only): a finite-difference projection of this observation failed in the previous Jacobian pass;
  // the whole observation goes to the list so that its tasks run on the side stream
  if (capped || (fd_slow && fd_slow[o])) {
    const int idx = atomicAdd(defer_count, 1);
    if (idx < defer_cap) { defer_list[idx] = (int)o; defer_skip[o] = 1; return; }
    if (capped) ok = base_projection<MODEL>(a, c, o, local, 100, capped, px, py);     // list full: the one-lane path, to the end
  }
  defer_skip[o] = 0;
  store_base_projection(a, o, ok, px, py, cost_vec, pixels, flags);
}

// The straggler kernel: 16 lanes per listed observation evaluate the SAME procedure speculatively.
//   lanes 0-7: the warm-start attempt, lanes 8-15: the attempt from the centre of the calibrated area -- the second attempt
//     does not depend on the first (same target, fixed start), the reference merely skips it when the first succeeds;
//   within an attempt, lane k (k = 0..7) takes damping attempt lm = base + k of the current round (lambda * 2^k) and
//     evaluates BOTH Unproject at the candidate (the test cost) and UnprojectWithJacobian at the same candidate (what the
//     NEXT outer iteration needs if this candidate is the first accepted one) -- two independent instruction streams in
//     one lane, which the scheduler interleaves.  The first accepted candidate in reference order (lowest lm) wins and
//     broadcasts pixel and evaluation to the group.
// An outer iteration thus costs one evaluation latency instead of 1 + (attempts until acceptance), and both attempts run
// side by side: ~100 evaluation latencies instead of ~600.  All arithmetic goes through the same device functions as the
// one-lane loop (project_target), evaluated on identical inputs.
// Exact shortcut: the loop state is (pixel, lambda); an iteration that maps it to itself (bitwise) will do so 100 times and
// end in `return false` -- the pinned-at-the-border lanes -- so the attempt stops there with that result.
template <int MODEL>
__global__ void __launch_bounds__(256) k_base_project_slow(PassArgs a, double* __restrict__ cost_vec, double* __restrict__ pixels,
                                                           uint8_t* __restrict__ flags) {
  constexpr double kEpsilon = 1e-12;
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (a.guard && *a.guard != 0) return;                     // (wave-uniform) the solve in front of this cost pass broke down
  const int cnt = min(*a.obs_count, a.obs_list_cap);
  if (((tid & ~63) >> 4) >= cnt) return;                    // wave-uniform
  const int g = tid >> 4;
  const int64_t o = a.obs_list[g < cnt ? g : cnt - 1];      // idle groups shadow the last entry; they never evaluate or store
  const int cam = a.obs_camera[o];
  const CamDev c = a.cams[cam];
  const bool live = g < cnt && c.model_type == MODEL;
  const int attempt = (lane >> 3) & 1, cand = lane & 7, gbase = lane & ~7;
  double target[3];
  local_point_of(a, o, cam, target);
  if (MODEL == kCentral) normalize3(target[0], target[1], target[2]);
  double px = a.last_projection[2 * o], py = a.last_projection[2 * o + 1];
  if (attempt == 1 || !in_calibrated_area(c, px, py) || px != px || py != py) center_pixel(c, px, py);
  Subst none; none.index = -1;
  double dir[3] = {0, 0, 0}, org[3] = {0, 0, 0}, jd[6] = {0, 0, 0, 0, 0, 0}, jo[6] = {0, 0, 0, 0, 0, 0};
  bool cur_in = false, active = live, result = false;
  if (active) cur_in = unproject_jac<MODEL>(c, none, px, py, dir, org, jd, jo);
  double lambda = -1.0;
  long long prev_px = -1, prev_py = -1, prev_lambda = -1;   // bit patterns of the previous iteration's state (-1 = NaN pattern: none)
  for (int it = 0; it < 100; ++it) {
    if (!__any(active)) break;
    if (active && !cur_in) { result = false; active = false; }          // CHECK() in the reference
    double cost = 0, H00 = 0, H01 = 0, H11 = 0, b0 = 0, b1 = 0;
    if (active) {
      projection_normal_equations<MODEL>(dir, org, jd, jo, target, cost, H00, H01, H11, b0, b1);
      if (lambda < 0) lambda = 0.01 * 0.5 * (H00 + H11);
      const long long bx = __double_as_longlong(px), by = __double_as_longlong(py), bl = __double_as_longlong(lambda);
      if (bx == prev_px && by == prev_py && bl == prev_lambda) { result = false; active = false; }   // fixed point
      prev_px = bx; prev_py = by; prev_lambda = bl;
    }
    bool accepted = false;
#pragma unroll 1
    for (int base = 0; base < 10; base += 8) {
      const int lm = base + cand;
      const bool mine = active && !accepted && lm < 10;
      double lam_c = lambda;
      for (int k = 0; k < cand; ++k) lam_c *= 2.0;             // the rejected attempts before this one
      double tx = px, ty = py, tc = INFINITY;
      double ndir[3] = {0, 0, 0}, norg[3] = {0, 0, 0}, njd[6] = {0, 0, 0, 0, 0, 0}, njo[6] = {0, 0, 0, 0, 0, 0};
      bool nin = false;
      if (mine) {
        projection_candidate(c, H00, H01, H11, b0, b1, lam_c, px, py, tx, ty);
        // the clamped candidate lies inside the calibrated area, so Unproject / UnprojectWithJacobian reduce to their
        // evaluation parts (model.hip.h: unproject, unproject_jac) -- straight-line code for both
        if (in_calibrated_area(c, tx, ty)) {
          double gx, gy;
          pixel_to_grid(c, tx, ty, gx, gy);
          gx += 2; gy += 2;
          double td[3], to[3];
          unproject_eval<MODEL, false>(c, none, (lds_cdouble_ptr)0, (lds_cdouble_ptr)0, (int)gx, (int)gy, gx, gy, td, to);
          unproject_jac_eval<MODEL, false>(c, none, (lds_cdouble_ptr)0, (lds_cdouble_ptr)0, (int)floor(gx), (int)floor(gy), gx, gy, ndir, norg, njd, njo);
          tc = projection_test_cost<MODEL>(td, to, target);
          nin = true;
        }
      }
      const bool acc_c = mine && (tc < cost);
      const unsigned m = (unsigned)((__ballot(acc_c) >> gbase) & 0xffull);
      const int w = m ? (__ffs(m) - 1) : -1;
      const int src = gbase + (w < 0 ? 0 : w);
      const double wtx = __shfl(tx, src, 64), wty = __shfl(ty, src, 64);
      const int w_in = __shfl((int)nin, src, 64);
      double wdir[3], worg[3], wjd[6], wjo[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) wdir[k] = __shfl(ndir[k], src, 64);
#pragma unroll
      for (int k = 0; k < 6; ++k) wjd[k] = __shfl(njd[k], src, 64);
      if (MODEL != kCentral) {
#pragma unroll
        for (int k = 0; k < 3; ++k) worg[k] = __shfl(norg[k], src, 64);
#pragma unroll
        for (int k = 0; k < 6; ++k) wjo[k] = __shfl(njo[k], src, 64);
      }
      if (active && !accepted) {
        if (w >= 0) {
          double l = lambda;
          for (int k = 0; k < w; ++k) l *= 2.0;              // the rejected attempts before the accepted one
          lambda = l * 0.5;
          px = wtx; py = wty;
#pragma unroll
          for (int k = 0; k < 3; ++k) dir[k] = wdir[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) jd[k] = wjd[k];
          if (MODEL != kCentral) {
#pragma unroll
            for (int k = 0; k < 3; ++k) org[k] = worg[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) jo[k] = wjo[k];
          }
          cur_in = w_in != 0;
          accepted = true;
        } else {
          const int tried = 10 - base < 8 ? 10 - base : 8;
          for (int k = 0; k < tried; ++k) lambda *= 2.0;
        }
      }
    }
    if (active) {
      if (!accepted) { result = cost < kEpsilon; active = false; }
      else if (cost < kEpsilon) { result = true; active = false; }
    }
  }
  // still active after 100 outer iterations: not converged (result stays false)
  const int first = lane & ~15;
  const int ok0 = __shfl((int)result, first, 64), ok1 = __shfl((int)result, first + 8, 64);
  const double px0 = __shfl(px, first, 64), py0 = __shfl(py, first, 64);
  const double px1 = __shfl(px, first + 8, 64), py1 = __shfl(py, first + 8, 64);
  if (live && (lane & 15) == 0)
    store_base_projection(a, o, ok0 || ok1, ok0 ? px0 : px1, ok0 ? py0 : py1, cost_vec, pixels, flags);
}

// Main launch (one lane per observation, stragglers deferred) followed by the straggler launch on `s_slow` (the same stream
// in a cost pass; the Jacobian pass passes its side stream and orders it with `ev_main_done`).
int launch_base_project(const PassArgs& a, int model_mask, double* cost_vec, double* pixels, uint8_t* flags, int* defer_list,
                        int* defer_count, int defer_cap, uint8_t* defer_skip, int outer_cap, const uint8_t* fd_slow, hipStream_t s) {
  if (a.n_obs == 0) return CBA_OK;
  CBA_HIP(hipMemsetAsync(defer_count, 0, sizeof(int), s));
  dim3 grid((unsigned)((a.n_obs + 255) / 256)), block(256);
  if (model_mask & 1) hipLaunchKernelGGL(k_base_project<kCentral>, grid, block, 0, s, a, cost_vec, pixels, flags, defer_list, defer_count, defer_cap, defer_skip, outer_cap, fd_slow);
  if (model_mask & 2) hipLaunchKernelGGL(k_base_project<kNoncentral>, grid, block, 0, s, a, cost_vec, pixels, flags, defer_list, defer_count, defer_cap, defer_skip, outer_cap, fd_slow);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}
// `a.obs_list / obs_count / obs_list_cap` = the straggler list filled by launch_base_project
int launch_base_project_slow(const PassArgs& a, int model_mask, double* cost_vec, double* pixels, uint8_t* flags, hipStream_t s) {
  if (a.n_obs == 0) return CBA_OK;
  dim3 grid((unsigned)(((int64_t)a.obs_list_cap * 16 + 255) / 256)), block(256);
  if (model_mask & 1) hipLaunchKernelGGL(k_base_project_slow<kCentral>, grid, block, 0, s, a, cost_vec, pixels, flags);
  if (model_mask & 2) hipLaunchKernelGGL(k_base_project_slow<kNoncentral>, grid, block, 0, s, a, cost_vec, pixels, flags);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ------------------------------------------------------------------------------------------------
// finite-difference tasks: one lane per (observation, task)
//   task 0..2      : local point component += kDelta                (joint_optimization.cc:357-372)
//   task 3..3+K-1  : grid parameter (cell, d) += delta in its local parametrisation
// ------------------------------------------------------------------------------------------------
// The task lanes of an observation (35 central / 83 non-central) evaluate the B-spline on ONE 4x4 control patch, a few
// times each (about two LM iterations of one UnprojectWithJacobian + one Unproject).  The workgroup stages the patches of
// the observations it covers (at most 256 / tasks + 2) in LDS once; every evaluation then reads its 16 control points
// with ds_read instead of 48 / 96 gathers through L1, and the kernel is built for 4 (central) / 3 (non-central)
// wavefronts per SIMD instead of 2 / 1 (the gathers of the whole patch in flight cost ~100 / ~190 VGPRs).  A lane whose
// pixel crosses into a neighbouring cell repeats its projection on the gather path after the staged attempt.
constexpr int kFdMaxObsPerBlock = 10;
// Row stride of a staged control patch in LDS: 16 control points + 2 doubles of padding.  Without the padding a row is 96 (central) /
// 192 (non-central) dwords, i.e. every observation's patch starts on the same bank, and the compiler reads a control point with
// ds_read2_b64 / ds_read_b128 (bank modulus 32 / 64 dwords, lane groups of 16): whenever the lanes of a group belong to different
// observations -- every group that straddles an observation boundary in k_fd_tasks, most groups in k_fd_pool once the lanes have
// drifted apart -- their reads collide (SQ_LDS_BANK_CONFLICT 18 % of the LDS cycles of k_fd_tasks, profiles/r04_pmc_valu_lds.txt).
// Four dwords of padding move neighbouring observations onto neighbouring 16-byte slots.  Layout only: same values, same arithmetic.
// Measured (profiles/r05_fd_patch_padding.txt): non-central cfg 4 FD kernel 8.0 -> 7.4 ms; cfg 2 / cfg 3 unchanged; 1 double of padding
// instead of 2: the same; padding the per-lane substitution slots (sSub) to 7 doubles: 7.7 ms (the 16-byte reads split).
#ifndef CBA_FD_PATCH_PAD
#define CBA_FD_PATCH_PAD 2
#endif
template <int DIM> struct FdPatchRow { static constexpr int kStride = 16 * DIM + CBA_FD_PATCH_PAD; };
#ifndef CBA_FD_WAVES_CENTRAL
#define CBA_FD_WAVES_CENTRAL 3      // wavefronts per SIMD the kernel is register-allocated for
#endif
#ifndef CBA_FD_WAVES_NONCENTRAL
#define CBA_FD_WAVES_NONCENTRAL 2
#endif
// The gather-path follow-up list is sized with the problem (launch_fd_tasks' redo_cap = a quarter of all tasks + 65 536; a miss
// needs a pixel within one LM step of a cell boundary).  A task that still finds the list full is counted in redo_count[2]
// (cba_fd_redo_overflow) and loses its Jacobian like a failed projection -- visible, never silent.

// One finite-difference task: (observation o, task k) -> fd_out / fd_ok at index t = o * tasks_per_obs + k.
// fd_task_setup: the perturbed input of the task (local point or substituted control point) and the finite-difference step;
// false = the control point lies outside the grid (CHECK() in the reference; cannot happen inside the rectangle): fd_ok[t] = 0.
template <int MODEL, bool STG>
__device__ __forceinline__ bool fd_task_setup(const PassArgs& a, const CamDev& c, int cam, int64_t o, int k, double bx, double by,
                                              double* local, double& delta, Subst& sub, double* sub_slot) {
  constexpr int PER = (MODEL == kCentral) ? 2 : 5;
  local_point_of(a, o, cam, local);
  sub.index = -1;
  if (k < 3) {
    delta = a.fd_delta * (MODEL == kCentral ? sqrt(local[0] * local[0] + local[1] * local[1] + local[2] * local[2]) : 0.1);
    local[k] += delta;
    return true;
  }
  delta = a.fd_delta;
  int g = k - 3;
  int cell = g / PER, d = g - cell * PER;
  double gx, gy;
  pixel_to_grid(c, bx, by, gx, gy);
  int ix = (int)floor(gx), iy = (int)floor(gy);
  int cx = ix + (cell & 3) - 1, cy = iy + (cell >> 2) - 1;
  if (cx < 0 || cy < 0 || cx >= c.gw || cy >= c.gh) return false;
  int seq = cx + cy * c.gw;
  sub.index = seq;
  const double* gd = c.grid + 3 * (size_t)seq;
  const double* tg = c.tangents + 6 * (size_t)seq;
  double o1 = (d == 0) ? delta : 0.0, o2 = (d == 1) ? delta : 0.0;
  // ApplyLocalUpdateToDirection / ApplyLocalUpdateToLine (direction_parametrization.h:45-55,
  // line_parametrization.h:107-120): always renormalises the direction
  double nd[3] = {gd[0] + o1 * tg[0] + o2 * tg[3], gd[1] + o1 * tg[1] + o2 * tg[4], gd[2] + o1 * tg[2] + o2 * tg[5]};
  normalize3(nd[0], nd[1], nd[2]);
  sub.d[0] = nd[0]; sub.d[1] = nd[1]; sub.d[2] = nd[2];
  if (MODEL == kNoncentral) {
    const double* go = c.grid + 3 * (size_t)c.gw * c.gh + 3 * (size_t)seq;
    double o3 = (d == 2) ? delta : 0.0, o4 = (d == 3) ? delta : 0.0, o5 = (d == 4) ? delta : 0.0;
    sub.o[0] = go[0] + o3 * tg[0] + o4 * tg[3] + o5 * gd[0];
    sub.o[1] = go[1] + o3 * tg[1] + o4 * tg[4] + o5 * gd[1];
    sub.o[2] = go[2] + o3 * tg[2] + o4 * tg[5] + o5 * gd[2];
  }
  if (STG) {
    sub_slot[0] = sub.d[0]; sub_slot[1] = sub.d[1]; sub_slot[2] = sub.d[2];
    if (MODEL == kNoncentral) { sub_slot[3] = sub.o[0]; sub_slot[4] = sub.o[1]; sub_slot[5] = sub.o[2]; }
  }
  return true;
}
// fd_task_store: the difference quotient of the task
__device__ __forceinline__ void fd_task_store(const PassArgs& a, const CamDev& c, int64_t o, int k, int64_t t, bool ok, double px, double py,
                                              double bx, double by, double delta, double* __restrict__ fd_out, uint8_t* __restrict__ fd_ok) {
  if (k >= 3 && a.jrec) {          // grid parameter: straight into the record (rows 0 / 1 of the 2 x K_g block)
    double* g = a.jrec + (size_t)o * a.rec_doubles + kRecHeader;
    const int Kg = c.params_per_point * 16;
    g[k - 3] = (px - bx) / delta;
    g[Kg + k - 3] = (py - by) / delta;
  } else {
    fd_out[2 * t] = (px - bx) / delta;
    fd_out[2 * t + 1] = (py - by) / delta;
  }
  fd_ok[t] = ok ? 1 : 0;
}
// STG: spline evaluated on the staged patch `st`; returns false if an iterate left that patch (nothing is written then).
template <int MODEL, bool STG>
__device__ __forceinline__ bool fd_task(const PassArgs& a, const CamDev& c, int cam, int64_t o, int k, int64_t t, const double* __restrict__ pixels,
                                        double* __restrict__ fd_out, uint8_t* __restrict__ fd_ok, StagedPatch<MODEL>* st, double* sub_slot) {
  double local[3];
  const double bx = pixels[2 * o], by = pixels[2 * o + 1];
  double px = bx, py = by;
  double delta;
  Subst sub;
  if (!fd_task_setup<MODEL, STG>(a, c, cam, o, k, bx, by, local, delta, sub, sub_slot)) { fd_ok[t] = 0; return true; }
  bool miss = false;
  const bool ok = project_point<MODEL, STG>(c, sub, local, px, py, st, &miss);
  if (STG && miss) return false;
  fd_task_store(a, c, o, k, t, ok, px, py, bx, by, delta, fd_out, fd_ok);
  return true;
}

// The task lanes of an observation (35 central / 83 non-central) evaluate the B-spline on ONE 4x4 control patch, a few
// times each (about two LM iterations of one UnprojectWithJacobian + one Unproject).  The workgroup stages the patches of
// the observations it covers (at most 256 / tasks + 2) in LDS once; every evaluation then reads its 16 control points
// with ds_read instead of 48 / 96 gathers through L1, which also takes the ~100 / ~190 VGPRs of a whole patch in flight
// out of the kernel.  A lane whose pixel crosses into a neighbouring cell appends its task to `redo` and the follow-up
// launch k_fd_redo repeats it on the gather path (the same arithmetic on the same control points).
template <int MODEL>
__global__ void __launch_bounds__(256, MODEL == kCentral ? CBA_FD_WAVES_CENTRAL : CBA_FD_WAVES_NONCENTRAL)
k_fd_tasks(PassArgs a, int tasks_per_obs, int localize_only, const double* __restrict__ pixels, const uint8_t* __restrict__ flags,
           double* __restrict__ fd_out, uint8_t* __restrict__ fd_ok, int64_t* __restrict__ redo, int* __restrict__ redo_count, int redo_cap,
           int* __restrict__ redo_overflow) {
  constexpr int PER = (MODEL == kCentral) ? 2 : 5;
  constexpr int DIM = (MODEL == kCentral) ? 3 : 6;
  __shared__ double sPatch[kFdMaxObsPerBlock][FdPatchRow<DIM>::kStride];
  __shared__ double sSub[256][DIM];             // per lane: its substituted control point
  __shared__ int sOrigin[kFdMaxObsPerBlock][2];
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x;
  const int64_t slot_first = t0 / tasks_per_obs;              // first observation slot of this workgroup
  // ---- stage the patches ----
  {
    const int64_t slot_last = (t0 + blockDim.x - 1) / tasks_per_obs;
    const int n_slots = (int)(slot_last - slot_first) + 1;    // <= 256 / 35 + 2 = 9 <= kFdMaxObsPerBlock
    const int64_t limit = a.obs_list ? (int64_t)min(*a.obs_count, a.obs_list_cap) : a.n_obs;
    for (int e = threadIdx.x; e < n_slots * 16; e += blockDim.x) {
      const int j = e >> 4, pt = e & 15;
      const int64_t slot = slot_first + j;
      bool live = slot < limit;
      int64_t o = 0;
      if (live) { o = a.obs_list ? (int64_t)a.obs_list[slot] : slot; live = (flags[o] & 1) != 0; }
      int fx = -(1 << 20), fy = -(1 << 20);
      if (live) {
        const CamDev c = a.cams[a.obs_camera[o]];
        if (c.model_type == MODEL) {
          double gx, gy;
          pixel_to_grid(c, pixels[2 * o], pixels[2 * o + 1], gx, gy);
          fx = (int)floor(gx + 2) - 3; fy = (int)floor(gy + 2) - 3;      // as unproject_jac places its patch
          const int cx = fx + (pt & 3), cy = fy + (pt >> 2);
          if (cx >= 0 && cy >= 0 && cx < c.gw && cy < c.gh) {
            const double* g = c.grid + 3 * ((size_t)cx + (size_t)cy * c.gw);
            sPatch[j][pt * DIM + 0] = g[0]; sPatch[j][pt * DIM + 1] = g[1]; sPatch[j][pt * DIM + 2] = g[2];
            if (MODEL == kNoncentral) {
              const double* p = g + 3 * (size_t)c.gw * c.gh;
              sPatch[j][pt * DIM + 3] = p[0]; sPatch[j][pt * DIM + 4] = p[1]; sPatch[j][pt * DIM + 5] = p[2];
            }
          } else {
            fx = -(1 << 20);                                             // never matches: such a patch is never evaluated
          }
        }
      }
      if (pt == 0) { sOrigin[j][0] = fx; sOrigin[j][1] = fy; }
    }
    __syncthreads();
  }
  int64_t t = t0 + threadIdx.x;
  int64_t o = t / tasks_per_obs;
  int k = (int)(t - o * tasks_per_obs);
  const int j = (int)(o - slot_first);
  if (a.obs_list) {
    const int cnt = min(*a.obs_count, a.obs_list_cap);
    if (o >= cnt) return;
    o = a.obs_list[o];
    t = o * tasks_per_obs + k;          // results are indexed by (observation, task)
  } else {
    if (o >= a.n_obs) return;
    if (a.skip && a.skip[o]) return;
  }
  if (!(flags[o] & 1)) return;
  int cam = a.obs_camera[o];
  const CamDev c = a.cams[cam];
  if (c.model_type != MODEL) return;
  const int n_tasks = 3 + (localize_only ? 0 : PER * 16);
  if (k >= n_tasks) return;
  StagedPatch<MODEL> st;
  st.p = (lds_cdouble_ptr)&sPatch[j][0];
  st.sub = (lds_cdouble_ptr)&sSub[threadIdx.x][0];
  st.fx = sOrigin[j][0]; st.fy = sOrigin[j][1];
  if (!fd_task<MODEL, true>(a, c, cam, o, k, t, pixels, fd_out, fd_ok, &st, &sSub[threadIdx.x][0])) {
    const int slot = atomicAdd(redo_count, 1);
    if (slot < redo_cap) redo[slot] = t;
    else { fd_ok[t] = 0; atomicAdd(redo_overflow, 1); }     // list full: dropped Jacobian, as a failed projection, and counted
  }
}
// ---- pooled schedule (round 5): lanes take tasks from a workgroup pool, one LM attempt per trip ----
// With one task per lane (k_fd_tasks above) a wavefront runs as long as its SLOWEST lane: projections need 1-3 outer iterations, and
// about one in a hundred ends with ten rejected damping attempts (the iterate is converged to rounding and no candidate improves
// it: ten Unproject evaluations, ten 2 x 2 solves) -- almost every second wavefront has such a lane and pays for it 64-fold
// (SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU), profiles/r05_fd_lane_utilisation.txt).  Here a workgroup owns a POOL of
// consecutive tasks (eight per lane), stages the control patches of all their observations once, and every lane runs a small state
// machine: take the next task of the pool -> [UnprojectWithJacobian + normal equations when an outer iteration starts] -> ONE damping
// attempt (candidate, Unproject, accept / reject) per trip of the loop -> store -> next task.  A lane stuck in rejected attempts
// just takes fewer tasks.  Every task evaluates exactly the expressions of project_target (model.hip.h) in the same order --
// the same device functions on the same inputs.  Flags and decisions are identical to the one-task-per-lane kernel; 0.004 % of the
// Jacobian entries differ by an ulp of a pixel in one projection, because the compiler contracts a multiply-add of the damped 2 x 2
// solve differently in the two kernels (tests/test_gpu_stragglers.py allows 1e-11 relative and 5e-4 differing entries; include/cba.h:
// cba_set_fd_schedule).  The default schedule is chosen per configuration, so runs with different camera setups are not bit-comparable.
#ifndef CBA_FD_POOL_WAVES_CENTRAL
#define CBA_FD_POOL_WAVES_CENTRAL 3     // 4 (128 VGPRs, 16 spilled) measured: cfg 2 the same, cfg 3 4 % slower
#endif
constexpr int kFdPoolFactor = 8;
template <int MODEL> struct FdPool { static constexpr int kMaxObs = (MODEL == kCentral) ? 64 : 32; };    // patches staged per workgroup (24.6 KB)
template <int MODEL>
__global__ void __launch_bounds__(256, MODEL == kCentral ? CBA_FD_POOL_WAVES_CENTRAL : CBA_FD_WAVES_NONCENTRAL)
k_fd_pool(PassArgs a, int tasks_per_obs, int pool, const double* __restrict__ pixels, const uint8_t* __restrict__ flags,
          double* __restrict__ fd_out, uint8_t* __restrict__ fd_ok, int64_t* __restrict__ redo, int* __restrict__ redo_count, int redo_cap,
          int* __restrict__ redo_overflow) {
  constexpr int PER = (MODEL == kCentral) ? 2 : 5;
  constexpr int DIM = (MODEL == kCentral) ? 3 : 6;
  constexpr int kMaxObs = FdPool<MODEL>::kMaxObs;
  constexpr double kEpsilon = 1e-12;
  __shared__ double sPatch[kMaxObs][FdPatchRow<DIM>::kStride];
  __shared__ double sSub[256][DIM];             // per lane: the substituted control point of its current task
  __shared__ int sOrigin[kMaxObs][2];
  __shared__ int sNext;
  const int64_t n_slots_all = a.obs_list ? (int64_t)min(*a.obs_count, a.obs_list_cap) : a.n_obs;
  const int64_t total = n_slots_all * tasks_per_obs;
  const int64_t t0 = (int64_t)blockIdx.x * pool;
  if (t0 >= total) return;                                      // (uniform) list launches are sized by the list's capacity
  const int n_pool = (int)((total - t0 < pool) ? total - t0 : pool);
  const int64_t slot_first = t0 / tasks_per_obs;
  {
    // ---- stage the patches of every observation the pool touches ----
    const int n_slots = (int)((t0 + n_pool - 1) / tasks_per_obs - slot_first) + 1;      // <= kMaxObs (launch_fd_tasks sizes the pool)
    for (int e = threadIdx.x; e < n_slots * 16; e += blockDim.x) {
      const int j = e >> 4, pt = e & 15;
      const int64_t slot = slot_first + j;
      const int64_t o = a.obs_list ? (int64_t)a.obs_list[slot] : slot;
      bool live = (flags[o] & 1) != 0;
      int fx = -(1 << 20), fy = -(1 << 20);
      if (live) {
        const CamDev c = a.cams[a.obs_camera[o]];
        if (c.model_type == MODEL) {
          double gx, gy;
          pixel_to_grid(c, pixels[2 * o], pixels[2 * o + 1], gx, gy);
          fx = (int)floor(gx + 2) - 3; fy = (int)floor(gy + 2) - 3;      // as unproject_jac places its patch
          const int cx = fx + (pt & 3), cy = fy + (pt >> 2);
          if (cx >= 0 && cy >= 0 && cx < c.gw && cy < c.gh) {
            const double* g = c.grid + 3 * ((size_t)cx + (size_t)cy * c.gw);
            sPatch[j][pt * DIM + 0] = g[0]; sPatch[j][pt * DIM + 1] = g[1]; sPatch[j][pt * DIM + 2] = g[2];
            if (MODEL == kNoncentral) {
              const double* p = g + 3 * (size_t)c.gw * c.gh;
              sPatch[j][pt * DIM + 3] = p[0]; sPatch[j][pt * DIM + 4] = p[1]; sPatch[j][pt * DIM + 5] = p[2];
            }
          } else {
            fx = -(1 << 20);                                             // never matches: such a patch is never evaluated
          }
        }
      }
      if (pt == 0) { sOrigin[j][0] = fx; sOrigin[j][1] = fy; }
    }
    if (threadIdx.x == 0) sNext = 0;
    __syncthreads();
  }
  const int n_tasks = 3 + PER * 16;
  const int first_k = (int)(t0 - slot_first * tasks_per_obs);          // task index of the pool's first task inside its observation
  // ---- per-lane state machine ----
  enum { kIdle = 0, kIterate = 1, kAttempt = 2, kStore = 3 };
  int state = kIdle, k = 0, it = 0, lm = 0;
  bool exhausted = false, ok = false;
  int64_t o = 0, t = 0;
  CamDev c = a.cams[0];
  Subst sub; sub.index = -1;
  StagedPatch<MODEL> st;
  st.sub = (lds_cdouble_ptr)&sSub[threadIdx.x][0];
  st.p = (lds_cdouble_ptr)&sPatch[0][0]; st.fx = st.fy = -(1 << 20);
  double target[3] = {0, 0, 0}, bx = 0, by = 0, px = 0, py = 0, delta = 1, lambda = -1;
  double cost = 0, H00 = 0, H01 = 0, H11 = 0, b0 = 0, b1 = 0;
  auto to_redo = [&]() {                     // an iterate left the staged patch: the whole task is repeated on the gather path
    const int slot = atomicAdd(redo_count, 1);
    if (slot < redo_cap) redo[slot] = t;
    else { fd_ok[t] = 0; atomicAdd(redo_overflow, 1); }     // list full: dropped Jacobian, as a failed projection, and counted
    state = kIdle;
  };
  for (;;) {
    if (state == kIdle && !exhausted) {
      for (int tries = 0; tries < 8; ++tries) {              // (invalid observations / other cameras: skip their task slots quickly)
        const int idx = atomicAdd(&sNext, 1);
        if (idx >= n_pool) { exhausted = true; break; }
        const int loc = idx + first_k;                       // 32-bit decode inside the pool (a 64-bit division per task is ~100 instructions)
        const int js = loc / tasks_per_obs;
        k = loc - js * tasks_per_obs;
        const int64_t slot = slot_first + js;
        o = a.obs_list ? (int64_t)a.obs_list[slot] : slot;
        if (!a.obs_list && a.skip && a.skip[o]) continue;
        if (!(flags[o] & 1) || k >= n_tasks) continue;
        const int cam = a.obs_camera[o];
        c = a.cams[cam];
        if (c.model_type != MODEL) continue;
        t = o * tasks_per_obs + k;            // results are indexed by (observation, task)
        const int j = js;
        st.p = (lds_cdouble_ptr)&sPatch[j][0];
        st.fx = sOrigin[j][0]; st.fy = sOrigin[j][1];
        bx = pixels[2 * o]; by = pixels[2 * o + 1];
        px = bx; py = by;
        if (!fd_task_setup<MODEL, true>(a, c, cam, o, k, bx, by, target, delta, sub, &sSub[threadIdx.x][0])) { fd_ok[t] = 0; continue; }
        if (MODEL == kCentral) normalize3(target[0], target[1], target[2]);      // project_point: the central model projects directions
        lambda = -1.0; it = 0;
        state = kIterate;
        break;
      }
    }
    if (__all(state == kIdle && exhausted)) break;
    if (state == kIterate) {                 // project_target: top of an outer iteration
      double dir[3], org[3], jd[6], jo[6];
      bool miss = false;
      const bool inside = unproject_jac_staged<MODEL>(c, sub, st, px, py, dir, org, jd, jo, miss);
      if (miss) to_redo();
      else if (!inside) { ok = false; state = kStore; }                 // CHECK() in the reference
      else {
        projection_normal_equations<MODEL>(dir, org, jd, jo, target, cost, H00, H01, H11, b0, b1);
        if (lambda < 0) lambda = 0.01 * 0.5 * (H00 + H11);
        lm = 0;
        state = kAttempt;
      }
    }
    if (state == kAttempt) {                 // one damping attempt
      double tx, ty;
      projection_candidate(c, H00, H01, H11, b0, b1, lambda, px, py, tx, ty);
      double test_cost = INFINITY;
      double td[3], to[3];
      bool miss = false;
      const bool tin = unproject_staged<MODEL>(c, sub, st, tx, ty, td, to, miss);
      if (miss) to_redo();
      else {
        if (tin) test_cost = projection_test_cost<MODEL>(td, to, target);
        if (test_cost < cost) {
          lambda *= 0.5;
          px = tx; py = ty;
          if (cost < kEpsilon) { ok = true; state = kStore; }
          else if (++it >= 100) { ok = false; state = kStore; }         // not converged after 100 outer iterations
          else state = kIterate;
        } else {
          lambda *= 2.0;
          if (++lm >= 10) { ok = cost < kEpsilon; state = kStore; }     // no candidate accepted
        }
      }
    }
    if (state == kStore) {
      fd_task_store(a, c, o, k, t, ok, px, py, bx, by, delta, fd_out, fd_ok);
      state = kIdle;
    }
  }
}

// localize_only (3 tasks per observation, no grid tasks): a workgroup would cover 86 observations -- nothing to share, the
// plain lane-per-task gather kernel
template <int MODEL>
__global__ void __launch_bounds__(256) k_fd_tasks_gather(PassArgs a, int tasks_per_obs, int n_tasks, const double* __restrict__ pixels,
                                                         const uint8_t* __restrict__ flags, double* __restrict__ fd_out, uint8_t* __restrict__ fd_ok) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t o = t / tasks_per_obs;
  const int k = (int)(t - o * tasks_per_obs);
  if (a.obs_list) {
    const int cnt = min(*a.obs_count, a.obs_list_cap);
    if (o >= cnt) return;
    o = a.obs_list[o];
    t = o * tasks_per_obs + k;
  } else {
    if (o >= a.n_obs) return;
    if (a.skip && a.skip[o]) return;
  }
  if (!(flags[o] & 1)) return;
  const int cam = a.obs_camera[o];
  const CamDev c = a.cams[cam];
  if (c.model_type != MODEL || k >= n_tasks) return;
  fd_task<MODEL, false>(a, c, cam, o, k, t, pixels, fd_out, fd_ok, nullptr, nullptr);
}
// follow-up: the tasks whose iterates left the staged patch, on the gather path
template <int MODEL>
__global__ void __launch_bounds__(256) k_fd_redo(PassArgs a, int tasks_per_obs, const double* __restrict__ pixels, double* __restrict__ fd_out,
                                                 uint8_t* __restrict__ fd_ok, const int64_t* __restrict__ redo, const int* __restrict__ redo_count, int redo_cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(*redo_count, redo_cap);
  if (i >= n) return;
  const int64_t t = redo[i];
  const int64_t o = t / tasks_per_obs;
  const int k = (int)(t - o * tasks_per_obs);
  const int cam = a.obs_camera[o];
  const CamDev c = a.cams[cam];
  if (c.model_type != MODEL) return;
  fd_task<MODEL, false>(a, c, cam, o, k, t, pixels, fd_out, fd_ok, nullptr, nullptr);
}
int launch_fd_tasks(const PassArgs& a, int model_mask, int tasks_per_obs, int localize_only, const double* pixels,
                    const uint8_t* flags, double* fd_out, uint8_t* fd_ok, int64_t* redo, int* redo_count, int redo_cap, int* redo_overflow,
                    hipStream_t s, int schedule) {
  if (a.n_obs == 0) return CBA_OK;
  int64_t total = (a.obs_list ? (int64_t)a.obs_list_cap : a.n_obs) * tasks_per_obs;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (256 / tasks_per_obs + 2 > kFdMaxObsPerBlock) {      // localize_only: 3 tasks per observation
    if (model_mask & 1) hipLaunchKernelGGL(k_fd_tasks_gather<kCentral>, grid, block, 0, s, a, tasks_per_obs, 3, pixels, flags, fd_out, fd_ok);
    if (model_mask & 2) hipLaunchKernelGGL(k_fd_tasks_gather<kNoncentral>, grid, block, 0, s, a, tasks_per_obs, 3, pixels, flags, fd_out, fd_ok);
    CBA_HIP(hipGetLastError());
    return CBA_OK;
  }
  CBA_HIP(hipMemsetAsync(redo_count, 0, sizeof(int), s));
  if (schedule == 0) {
    // pooled schedule (default): a workgroup takes `pool` consecutive tasks; the pool is capped by the patches a workgroup can stage
    auto launch_pool = [&](auto kernel, int max_obs) {
      int pool = kFdPoolFactor * 256;
      const int cap = (max_obs - 2) * tasks_per_obs;
      if (pool > cap) pool = cap;
      const dim3 pgrid((unsigned)((total + pool - 1) / pool));
      hipLaunchKernelGGL(kernel, pgrid, block, 0, s, a, tasks_per_obs, pool, pixels, flags, fd_out, fd_ok, redo, redo_count, redo_cap, redo_overflow);
    };
    if (model_mask & 1) launch_pool(k_fd_pool<kCentral>, FdPool<kCentral>::kMaxObs);
    if (model_mask & 2) launch_pool(k_fd_pool<kNoncentral>, FdPool<kNoncentral>::kMaxObs);
  } else {
    if (model_mask & 1)
      hipLaunchKernelGGL(k_fd_tasks<kCentral>, grid, block, 0, s, a, tasks_per_obs, localize_only, pixels, flags, fd_out, fd_ok, redo, redo_count, redo_cap, redo_overflow);
    if (model_mask & 2)
      hipLaunchKernelGGL(k_fd_tasks<kNoncentral>, grid, block, 0, s, a, tasks_per_obs, localize_only, pixels, flags, fd_out, fd_ok, redo, redo_count, redo_cap,
                         redo_overflow);
  }
  // gather-path follow-up for the (rare) tasks that left their staged patch: fixed grid over the list capacity, the count
  // stays on the device (workgroups past it exit at once)
  const dim3 rgrid((unsigned)((redo_cap + 255) / 256));
  if (model_mask & 1) hipLaunchKernelGGL(k_fd_redo<kCentral>, rgrid, block, 0, s, a, tasks_per_obs, pixels, fd_out, fd_ok, redo, redo_count, redo_cap);
  if (model_mask & 2) hipLaunchKernelGGL(k_fd_redo<kNoncentral>, rgrid, block, 0, s, a, tasks_per_obs, pixels, fd_out, fd_ok, redo, redo_count, redo_cap);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ------------------------------------------------------------------------------------------------
// assemble the Jacobian record of one observation (one lane per observation)
// ------------------------------------------------------------------------------------------------
// returns true if the observation keeps its Jacobian (the caller then copies the grid part of the record)
__device__ __forceinline__ bool assemble_header(const PassArgs& a, int64_t o, int rig_in_state, int localize_only,
                                                     const double* __restrict__ rig7, const double* __restrict__ camrig7,
                                                     int tasks_per_obs, int rec_doubles, const double* __restrict__ pixels,
                                                     uint8_t* __restrict__ flags, const double* __restrict__ fd_out,
                                                     const uint8_t* __restrict__ fd_ok, double* __restrict__ jrec,
                                                     int* __restrict__ cells) {
  if (o >= a.n_obs) return false;
  uint8_t f = flags[o];
  if (!(f & 1)) return false;
  int cam = a.obs_camera[o];
  const CamDev c = a.cams[cam];
  const int per = c.params_per_point;
  const int Kg = localize_only ? 0 : per * 16;
  const int n_tasks = 3 + Kg;
  double* rec = jrec + (size_t)o * rec_doubles;
  double px = pixels[2 * o], py = pixels[2 * o + 1];
  double rx = px - (double)a.obs_xy[2 * o], ry = py - (double)a.obs_xy[2 * o + 1];
  rec[0] = rx; rec[1] = ry;
  rec[2] = huber_weight_sq(rx * rx + ry * ry);
  const uint8_t* okp = fd_ok + (size_t)o * tasks_per_obs;
  bool all_ok = true;
  for (int k = 0; k < n_tasks; ++k) all_ok = all_ok && (okp[k] != 0);
  if (!all_ok) {  // residual is kept, Jacobian dropped (joint_optimization.cc:373-376, 446-448)
    flags[o] = 1;
    return false;
  }
  const double* fd = fd_out + 2 * (size_t)o * tasks_per_obs;
  double pwl[6];  // d pixel / d local point, 2x3
#pragma unroll
  for (int k = 0; k < 3; ++k) { pwl[k] = fd[2 * k]; pwl[3 + k] = fd[2 * k + 1]; }
  const double* p = a.points + 3 * (size_t)a.obs_point[o];
  const int img = a.obs_image[o];
  double* Jpose = rec + 3;
  double* Jrig = rec + 15;
  double* Jpt = rec + 27;
  if (rig_in_state) {
    // local = R(qc) (R(qr) p + tr) + tc            (ComputeRigJacobian, joint_optimization_jacobians.h:121-343)
    const double* qc = camrig7 + 7 * (size_t)cam;
    const double* qr = rig7 + 7 * (size_t)img;
    double Rc[9], Rr[9], Mr[9], Mc[9];
    poly_rotation(qc, Rc);
    poly_rotation(qr, Rr);
    rotated_point_wrt_update(qr, p, Mr);
    double v[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) v[r] = Rr[3 * r] * p[0] + Rr[3 * r + 1] * p[1] + Rr[3 * r + 2] * p[2] + qr[4 + r];
    rotated_point_wrt_update(qc, v, Mc);
    double A[6];  // pwl * Rc  (2x3)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) A[3 * r + k] = pwl[3 * r] * Rc[k] + pwl[3 * r + 1] * Rc[3 + k] + pwl[3 * r + 2] * Rc[6 + k];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        Jpose[6 * r + k] = A[3 * r] * Mr[k] + A[3 * r + 1] * Mr[3 + k] + A[3 * r + 2] * Mr[6 + k];
        Jpose[6 * r + 3 + k] = A[3 * r + k];
        Jrig[6 * r + k] = pwl[3 * r] * Mc[k] + pwl[3 * r + 1] * Mc[3 + k] + pwl[3 * r + 2] * Mc[6 + k];
        Jrig[6 * r + 3 + k] = pwl[3 * r + k];
        Jpt[3 * r + k] = A[3 * r] * Rr[k] + A[3 * r + 1] * Rr[3 + k] + A[3 * r + 2] * Rr[6 + k];
      }
  } else {
    // single camera: Jacobian wrt. a left update of image_q_global (joint_optimization.cc:392-397, 431-437)
    const double* q = a.itg + 16 * ((size_t)img * a.n_cameras + cam);
    double M[9], R[9];
    rotated_point_wrt_update(q, p, M);
    poly_rotation(q, R);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        Jpose[6 * r + k] = pwl[3 * r] * M[k] + pwl[3 * r + 1] * M[3 + k] + pwl[3 * r + 2] * M[6 + k];
        Jpose[6 * r + 3 + k] = pwl[3 * r + k];
        Jrig[6 * r + k] = 0.0; Jrig[6 * r + 3 + k] = 0.0;
        Jpt[3 * r + k] = pwl[3 * r] * R[k] + pwl[3 * r + 1] * R[3 + k] + pwl[3 * r + 2] * R[6 + k];
      }
  }
  double gx, gy;
  pixel_to_grid(c, px, py, gx, gy);
  cells[2 * o] = (int)floor(gx) - 1;
  cells[2 * o + 1] = (int)floor(gy) - 1;
  flags[o] = 3;
  return !localize_only;
}
__global__ void __launch_bounds__(256) k_assemble(PassArgs a, int rig_in_state, int localize_only,
                                                  const double* __restrict__ rig7, const double* __restrict__ camrig7,
                                                  int tasks_per_obs, int rec_doubles, const double* __restrict__ pixels,
                                                  uint8_t* __restrict__ flags, const double* __restrict__ fd_out,
                                                  const uint8_t* __restrict__ fd_ok, double* __restrict__ jrec,
                                                  int* __restrict__ cells, uint8_t* __restrict__ fd_slow) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // the header of a record is one lane's work
  const bool with_jacobian = assemble_header(a, o, rig_in_state, localize_only, rig7, camrig7, tasks_per_obs, rec_doubles, pixels, flags,
                                             fd_out, fd_ok, jrec, cells);
  // a residual that lost its Jacobian has a finite-difference projection that FAILED, i.e. ran its whole iteration budget
  // (5-8 ms for one lane at the non-central config): next time its tasks run on the side stream (k_base_project)
  if (o < a.n_obs) fd_slow[o] = flags[o] == 1;
  (void)with_jacobian;     // the 2 x K_g grid part of the record was written by the finite-difference tasks themselves (fd_task)
}
int launch_assemble(const PassArgs& a, const Layout& L, const DevState& st, int tasks_per_obs, int rec_doubles,
                    const double* pixels, uint8_t* flags, const double* fd_out, const uint8_t* fd_ok, double* jrec,
                    int* cells, uint8_t* fd_slow, hipStream_t s) {
  if (a.n_obs == 0) return CBA_OK;
  hipLaunchKernelGGL(k_assemble, dim3((unsigned)((a.n_obs + 255) / 256)), dim3(256), 0, s, a, L.rig_in_state,
                     L.localize_only, st.rig_tr_global, st.camera_tr_rig, tasks_per_obs, rec_doubles, pixels, flags,
                     fd_out, fd_ok, jrec, cells, fd_slow);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ------------------------------------------------------------------------------------------------
// JtJ / Jtr accumulation: one wavefront per observation.  The K columns of the observation's
// Jacobian (ascending global variable index) are staged in LDS; the K(K+1)/2 upper-triangle
// products are dealt to the 64 lanes through a (row,col) pair table and added with hardware fp64
// atomics to the block-diagonal / off-diagonal / dense parts (GetPartOfHAndB,
// lm_optimizer_update_accumulator.h:478-505).  Only upper triangles are written.
// ------------------------------------------------------------------------------------------------
struct AccumLayout {
  int rig_in_state, eliminate_points, localize_only;
  int first_rig_tr_global, first_camera_tr_rig, first_points;
  int block_dof, block_size, dense_dof;
};

// One wavefront walks kAccChunk consecutive observations.  Entries whose row AND column belong to the
// imageset pose / rig pose ("hot": every observation of the same image and camera hits the same few
// addresses, and the rig blocks are hit by every observation of the camera) are summed in registers
// over the chunk and flushed with one atomic per entry when the (image, camera) key changes; all other
// entries go straight to HBM with hardware fp64 atomics.
constexpr int kAccChunk = 8;
// with the point terms gone (k_accumulate_points) an observation costs ~90 products here, and what limits the kernel is the flush:
// every wavefront adds its rig-pose sums to the SAME 27 entries per camera (2.4 ms at cfg 3 with chunks of 8: 93 k atomics per
// address); 64 observations per wavefront = 8x fewer flushes
constexpr int kAccChunkHot = 64;
constexpr int kHotMax = 12;                       // pose 6 + rig 6
constexpr int kHotPairs = kHotMax * (kHotMax + 1) / 2;   // 78

// Accumulator policy.  DET = false: hardware fp64 atomics -- sums depend on the order in which wavefronts arrive (last
// bits differ from run to run).  DET = true (cba_config.deterministic): every contribution is converted to 64-bit fixed
// point with ONE power-of-two scale per pass (k_det_scale: 2^62 / (n_obs * largest possible |contribution|), so no sum
// can overflow) and added with integer atomics; integer addition is associative, so registers, LDS and HBM sums are
// bit-identical for every execution order -- no sorting, no second data layout.  The targets hold the integers during
// the pass (same 8-byte slots) and k_det_convert turns them into doubles afterwards.  Resolution: ABSOLUTE, about 19 decimal
// digits below the largest POSSIBLE sum (n_obs x the largest contribution; actual entries collect 1e2..1e3 contributions, so the
// largest actual entry sits orders of magnitude below that bound: measured, diagonal entries within 1e-8 of the largest one agree
// with the fp64-atomic mode to 2e-6 relative, tests/test_gpu_deterministic.py); the n_obs bound assumes at most one contribution per entry and observation,
// which holds for every target (an observation touches an entry of H / b once).  DESIGN.md section 4a.
template <bool DET> struct Acc;
template <> struct Acc<false> {
  typedef double T;
  static __device__ __forceinline__ T from(double v, double) { return v; }
  static __device__ __forceinline__ void add(double* p, T v) { unsafeAtomicAdd(p, v); }
  static __device__ __forceinline__ void add_lds(double* p, T v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  static __device__ __forceinline__ double to_double(T v, double) { return v; }
};
template <> struct Acc<true> {
  typedef long long T;
  static __device__ __forceinline__ T from(double v, double scale) { return __double2ll_rn(v * scale); }
  static __device__ __forceinline__ void add(double* p, T v) { atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); }
  static __device__ __forceinline__ void add_lds(double* p, T v) {
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  static __device__ __forceinline__ double to_double(T v, double scale) { return (double)v / scale; }
};

template <bool DET>
__device__ __forceinline__ void acc_add_H(const AccumLayout& L, const AccumTargets& T, int row, int col, typename Acc<DET>::T v) {
  if (row < L.block_dof) {
    if (col < L.block_dof) {
      int blk = row / L.block_size;
      int base = blk * L.block_size;
      Acc<DET>::add(T.Dblk + (size_t)blk * L.block_size * L.block_size + (row - base) * L.block_size + (col - base), v);
    } else {
      Acc<DET>::add(T.B + (size_t)row * L.dense_dof + (col - L.block_dof), v);
    }
  } else {
    Acc<DET>::add(T.Hdd + (size_t)(row - L.block_dof) * L.dense_dof + (col - L.block_dof), v);
  }
}
template <bool DET>
__device__ __forceinline__ void acc_add_b(const AccumLayout& L, const AccumTargets& T, int row, typename Acc<DET>::T v) {
  if (row < L.block_dof) Acc<DET>::add(T.bblk + row, v);
  else Acc<DET>::add(T.bd + (row - L.block_dof), v);
}

template <bool DET>
__global__ void __launch_bounds__(256) k_accumulate(PassArgs a, AccumLayout L, int rec_doubles,
                                                    const uint8_t* __restrict__ flags, const double* __restrict__ jrec,
                                                    const int* __restrict__ cells, const uint32_t* __restrict__ pair_tables,
                                                    const int* __restrict__ pair_counts, AccumTargets T,
                                                    const double* __restrict__ det_scale, int points_separate) {
  typedef typename Acc<DET>::T acc_t;
  const double scale = DET ? det_scale[0] : 1.0;
  const double scale_b = DET ? det_scale[1] : 1.0;     // the J^T r sums have their own (finer) fixed-point scale
  __shared__ double sJ0[4][kMaxCols];
  __shared__ double sJ1[4][kMaxCols];
  __shared__ double sW0[4][kMaxCols];
  __shared__ double sW1[4][kMaxCols];
  __shared__ int sIdx[4][kMaxCols];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int chunk = points_separate ? kAccChunkHot : kAccChunk;
  const int64_t o_begin = ((int64_t)blockIdx.x * 4 + wv) * chunk;
  const int nrig = L.rig_in_state ? 6 : 0;
  const int nh = 6 + nrig;                       // hot columns
  const int h0 = L.eliminate_points ? 3 : 0;     // their first position in the ascending column list
  const int nhp = nh * (nh + 1) / 2;             // hot pairs
  // hot slots of this lane: slot s < nhp is pair (hi, hk); slot nhp + i is b entry i.  Two slots per lane.
  int hs_i[2], hs_k[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int sidx = lane + 64 * t;
    hs_i[t] = -1; hs_k[t] = -1;
    if (sidx < nhp) {
      int rem = sidx, i = 0;
      while (rem >= nh - i) { rem -= nh - i; ++i; }
      hs_i[t] = i; hs_k[t] = i + rem;
    } else if (sidx < nhp + nh) {
      hs_i[t] = sidx - nhp; hs_k[t] = -2;        // b entry
    }
  }
  acc_t hot[2] = {0, 0};
  int cur_pose = -1, cur_rig = -1;
  auto flush = [&]() {
    if (cur_pose < 0) return;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (hs_i[t] < 0) continue;
      int row = hs_i[t] < 6 ? cur_pose + hs_i[t] : cur_rig + hs_i[t] - 6;
      if (hs_k[t] == -2) acc_add_b<DET>(L, T, row, hot[t]);
      else acc_add_H<DET>(L, T, row, hs_k[t] < 6 ? cur_pose + hs_k[t] : cur_rig + hs_k[t] - 6, hot[t]);
      hot[t] = 0;
    }
  };
  for (int c = 0; c < chunk; ++c) {
    const int64_t o = o_begin + c;
    if (o >= a.n_obs) break;
    if (flags[o] != 3) continue;  // wave-uniform
    const int cam = a.obs_camera[o];
    const CamDev cd = a.cams[cam];
    const int per = cd.params_per_point;
    const int Kg = L.localize_only ? 0 : per * 16;
    const int K = 6 + nrig + 3 + Kg;
    const double* rec = jrec + (size_t)o * rec_doubles;
    const double w = rec[2];
    const int pose_idx = L.first_rig_tr_global + 6 * (a.pose_slot ? a.pose_slot[a.obs_image[o]] : a.obs_image[o]);
    const int rig_idx = L.first_camera_tr_rig + 6 * cam;
    const int point_idx = L.first_points + 3 * a.obs_point[o];
    if (pose_idx != cur_pose || rig_idx != cur_rig) {
      flush();
      cur_pose = pose_idx; cur_rig = rig_idx;
    }
    const int cx0 = cells[2 * o], cy0 = cells[2 * o + 1];
    __builtin_amdgcn_wave_barrier();   // previous iteration's LDS reads are done before overwriting
    // points_separate (poses eliminated): every term with a point column is summed per pattern point by
    // k_accumulate_points; only the pose / rig-pose ("hot") columns are left here
    const int Ks = points_separate ? nh : K;
    for (int k = lane; k < Ks; k += 64) {
      int idx; double j0, j1;
      int kk = k;
      // ascending index order: [point] pose [rig] [point] grid  (joint_optimization.cc:490-590)
      if (L.eliminate_points) {
        if (kk < 3) { idx = point_idx + kk; j0 = rec[27 + kk]; j1 = rec[30 + kk]; goto done; }
        kk -= 3;
      }
      if (kk < 6) { idx = pose_idx + kk; j0 = rec[3 + kk]; j1 = rec[9 + kk]; goto done; }
      kk -= 6;
      if (nrig) {
        if (kk < 6) { idx = rig_idx + kk; j0 = rec[15 + kk]; j1 = rec[21 + kk]; goto done; }
        kk -= 6;
      }
      if (!L.eliminate_points) {
        if (kk < 3) { idx = point_idx + kk; j0 = rec[27 + kk]; j1 = rec[30 + kk]; goto done; }
        kk -= 3;
      }
      {
        int cell = kk / per, d = kk - cell * per;
        int seq = (cx0 + (cell & 3)) + (cy0 + (cell >> 2)) * cd.gw;
        idx = L.block_dof + grid_column(cd, seq, d);
        j0 = rec[kRecHeader + kk]; j1 = rec[kRecHeader + Kg + kk];
      }
    done:
      sIdx[wv][k] = idx;
      sJ0[wv][k] = j0; sJ1[wv][k] = j1;
      sW0[wv][k] = w * j0; sW1[wv][k] = w * j1;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const double r0 = rec[0], r1 = rec[1];
    // hot entries -> registers
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (hs_i[t] < 0) continue;
      const int i = h0 + hs_i[t];
      if (hs_k[t] == -2) hot[t] += Acc<DET>::from(r0 * sW0[wv][i] + r1 * sW1[wv][i], scale_b);
      else { const int k = h0 + hs_k[t]; hot[t] += Acc<DET>::from(sW0[wv][i] * sJ0[wv][k] + sW1[wv][i] * sJ1[wv][k], scale); }
    }
    if (points_separate) continue;                 // wave-uniform
    // b += Jw^T r (non-hot positions)
    for (int k = lane; k < K - Kg; k += 64) {      // the grid entries are summed per cell by k_accumulate_cells
      if (k >= h0 && k < h0 + nh) continue;
      acc_add_b<DET>(L, T, sIdx[wv][k], Acc<DET>::from(r0 * sW0[wv][k] + r1 * sW1[wv][k], scale_b));
    }
    // remaining upper-triangle products (the pair tables exclude hot-hot pairs)
    const int slot = (per == 2) ? 0 : 1;
    const int npairs = pair_counts[slot];
    const uint32_t* table = pair_tables + (size_t)slot * (kMaxCols * (kMaxCols + 1) / 2);
    for (int e = lane; e < npairs; e += 64) {
      uint32_t pr = table[e];
      int i = pr >> 16, k = pr & 0xffff;
      double v = sW0[wv][i] * sJ0[wv][k] + sW1[wv][i] * sJ1[wv][k];
      acc_add_H<DET>(L, T, sIdx[wv][i], sIdx[wv][k], Acc<DET>::from(v, scale));
    }
  }
  flush();
}
// ------------------------------------------------------------------------------------------------
// Terms with a pattern-point column, grouped by point (poses eliminated; lm_optimizer_jtj_accumulator_base.h:359-400 adds
// them observation by observation): point x point, J^T r of the point, rig pose x point and point x grid.  Every one of
// the ~n_obs / n_points observations of a point adds to the SAME 9 (+18) entries and to the same three rows of H_dd, so
// the observations are bucketed by (camera, point) ONCE (the point of an observation never changes: cba_set_observations)
// and one workgroup per bucket and column chunk
//   * sums the 6 + 3 (+ 18) dense entries in registers -> one atomic per entry (several cameras share a point),
//   * sums the three point rows over the camera's grid columns in LDS (3 x chunk_cols doubles, LDS atomics) and writes
//     them out with plain coalesced stores: (point rows) x (grid columns of this camera) belong to this bucket alone.
// That replaces 9 + 3 K_g global atomics per observation (105 / 249 of them, 456-way contended on the point entries at
// BASELINE configs[1]) by K_g LDS atomics and 3 x (grid columns) stores per point.
// ------------------------------------------------------------------------------------------------
constexpr int kPointChunkColsMax = 5120;        // 3 rows x 5120 doubles = 120 KB of LDS
constexpr int kPointThreads = 1024;             // one workgroup per CU (LDS): 16 wavefronts keep the record reads in flight
constexpr int kPointUnroll = 4;                 // items per lane and step, loads of all four issued before the first use
template <bool DET>
__global__ void __launch_bounds__(kPointThreads) k_accumulate_points(PassArgs a, AccumLayout L, int rec_doubles, const uint8_t* __restrict__ flags,
                                                                     const double* __restrict__ jrec, const int* __restrict__ cells,
                                                                     const int* __restrict__ key_start, const int* __restrict__ key_obs,
                                                                     int n_points, int nchunks, int chunk_cols, AccumTargets T,
                                                                     const double* __restrict__ det_scale) {
  typedef typename Acc<DET>::T acc_t;
  constexpr int NT = kPointThreads, NW = kPointThreads / 64, U = kPointUnroll;
  extern __shared__ double s_rows[];            // [3][chunk_cols]; fixed point in deterministic mode
  __shared__ acc_t s_red[NW][27];
  const double scale = DET ? det_scale[0] : 1.0;
  const double scale_b = DET ? det_scale[1] : 1.0;
  const int key = blockIdx.x / nchunks, chunk = blockIdx.x - key * nchunks;
  const int cam = key / n_points, pt = key - cam * n_points;
  const CamDev& cd = a.cams[cam];
  const int per = cd.params_per_point, gw = cd.gw;
  const int* __restrict__ gperm = cd.gperm;
  const int Kg = L.localize_only ? 0 : per * 16;
  const int ncols = Kg ? per * gw * cd.gh : 0;
  const int c0 = chunk * chunk_cols;
  const int o_begin = key_start[key], n = key_start[key + 1] - o_begin;
  if (n == 0 || (chunk > 0 && c0 >= ncols)) return;      // H_dd is zero-filled before the accumulation
  const int c1 = c0 + chunk_cols < ncols ? c0 + chunk_cols : ncols;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nrig = L.rig_in_state ? 6 : 0;
  const int point_idx = L.first_points + 3 * pt;
  if (c0 < c1) {
    for (int i = tid; i < 3 * chunk_cols; i += NT) s_rows[i] = 0.0;
    __syncthreads();
    // one (observation, grid column) item per lane and slot; three dependent load levels (list -> record / cell -> order
    // of the control point), each issued for all U slots before anything is used
    const int items = n * Kg;
    for (int it0 = tid; it0 < items; it0 += NT * U) {
      int64_t o[U]; int kk[U]; bool live[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it = it0 + u * NT;
        live[u] = it < items;
        const int oi = live[u] ? it / Kg : 0;
        kk[u] = live[u] ? it - oi * Kg : 0;
        o[u] = key_obs[o_begin + oi];
      }
      int cx[U], cy[U]; uint8_t fl[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { fl[u] = flags[o[u]]; cx[u] = cells[2 * o[u]]; cy[u] = cells[2 * o[u] + 1]; }
      int col[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cell = kk[u] / per, d = kk[u] - cell * per;
        int seq = (cx[u] + (cell & 3)) + (cy[u] + (cell >> 2)) * gw;
        live[u] = live[u] && fl[u] == 3;
        if (!live[u]) seq = 0;                                       // cells of an invalid observation are not defined
        col[u] = per * (gperm ? gperm[seq] : seq) + d;               // camera-local column (grid_column - intr_offset)
        live[u] = live[u] && col[u] >= c0 && col[u] < c1;
      }
      // the record is read only by the chunk its columns fall into (a 4 x 4 patch nearly always lies in one chunk: half the
      // record traffic of a two-chunk launch)
      double w[U], g0[U], g1[U], q0[U][3], q1[U][3];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!live[u]) continue;
        const double* rec = jrec + (size_t)o[u] * rec_doubles;
        w[u] = rec[2]; g0[u] = rec[kRecHeader + kk[u]]; g1[u] = rec[kRecHeader + Kg + kk[u]];
#pragma unroll
        for (int r = 0; r < 3; ++r) { q0[u][r] = rec[27 + r]; q1[u][r] = rec[30 + r]; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!live[u]) continue;
#pragma unroll
        for (int r = 0; r < 3; ++r)
          Acc<DET>::add_lds(&s_rows[r * chunk_cols + (col[u] - c0)], Acc<DET>::from((w[u] * q0[u][r]) * g0[u] + (w[u] * q1[u][r]) * g1[u], scale));
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double* out = T.Hdd + (size_t)(point_idx - L.block_dof + r) * L.dense_dof + cd.intr_offset + c0;
      for (int c = tid; c < c1 - c0; c += NT) out[c] = s_rows[r * chunk_cols + c];
    }
  }
  if (chunk != 0) return;
  // dense entries of the bucket: [0..5] point x point (upper), [6..8] J^T r, [9..26] rig pose x point
  acc_t acc[27];
#pragma unroll
  for (int e = 0; e < 27; ++e) acc[e] = 0;
  for (int i = tid; i < n; i += NT) {
    const int64_t o = key_obs[o_begin + i];
    if (flags[o] != 3) continue;
    const double* rec = jrec + (size_t)o * rec_doubles;
    const double r0 = rec[0], r1 = rec[1], w = rec[2];
    double p0[3], p1[3], w0[3], w1[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { p0[r] = rec[27 + r]; p1[r] = rec[30 + r]; w0[r] = w * p0[r]; w1[r] = w * p1[r]; }
    int e = 0;
#pragma unroll
    for (int i2 = 0; i2 < 3; ++i2)
#pragma unroll
      for (int k2 = i2; k2 < 3; ++k2) acc[e++] += Acc<DET>::from(w0[i2] * p0[k2] + w1[i2] * p1[k2], scale);
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[6 + r] += Acc<DET>::from(r0 * w0[r] + r1 * w1[r], scale_b);
    if (nrig) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const double wq0 = w * rec[15 + q], wq1 = w * rec[21 + q];
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[9 + 3 * q + r] += Acc<DET>::from(wq0 * p0[r] + wq1 * p1[r], scale);
      }
    }
  }
  const int ne = nrig ? 27 : 9;
#pragma unroll
  for (int e = 0; e < 27; ++e) {
    if (e >= ne) break;
    acc_t v = acc[e];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) s_red[wv][e] = v;
  }
  __syncthreads();
  if (tid < ne) {
    acc_t v = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) v += s_red[i][tid];
    if (tid < 6) {
      int i2 = 0, rem = tid;
      while (rem >= 3 - i2) { rem -= 3 - i2; ++i2; }
      acc_add_H<DET>(L, T, point_idx + i2, point_idx + i2 + rem, v);
    } else if (tid < 9) {
      acc_add_b<DET>(L, T, point_idx + tid - 6, v);
    } else {
      const int q = (tid - 9) / 3, r = (tid - 9) - 3 * q;
      acc_add_H<DET>(L, T, L.first_camera_tr_rig + 6 * cam + q, point_idx + r, v);
    }
  }
}
int point_chunks(const std::vector<cba_camera>& cams, int localize_only, int* chunk_cols) {
  int maxcols = 0;
  if (!localize_only)
    for (const cba_camera& c : cams) maxcols = std::max(maxcols, (c.model_type == CBA_CENTRAL_GENERIC ? 2 : 5) * c.grid_w * c.grid_h);
  const int nchunks = std::max(1, (maxcols + kPointChunkColsMax - 1) / kPointChunkColsMax);
  *chunk_cols = std::max(8, ((maxcols + nchunks - 1) / nchunks + 7) / 8 * 8);
  return nchunks;
}
int launch_accumulate_points(const PassArgs& a, const Layout& L, const std::vector<cba_camera>& cams, int rec_doubles, const uint8_t* flags,
                             const double* jrec, const int* cells, const int* key_start, const int* key_obs, AccumTargets t,
                             const double* det_scale, hipStream_t s) {
  if (a.n_obs == 0 || L.n_points == 0) return CBA_OK;
  AccumLayout al;
  al.rig_in_state = L.rig_in_state; al.eliminate_points = L.eliminate_points; al.localize_only = L.localize_only;
  al.first_rig_tr_global = L.first_rig_tr_global; al.first_camera_tr_rig = L.first_camera_tr_rig;
  al.first_points = L.first_points; al.block_dof = L.block_dof; al.block_size = L.block_size; al.dense_dof = L.dense_dof;
  int chunk_cols = 0;
  const int nchunks = point_chunks(cams, L.localize_only, &chunk_cols);
  const size_t lds = sizeof(double) * 3 * (size_t)chunk_cols;
  const dim3 grid((unsigned)((size_t)L.n_cameras * L.n_points * nchunks));
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_accumulate_points<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * kPointChunkColsMax * 8);
    if (attr_rc == hipSuccess)
      attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_accumulate_points<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * kPointChunkColsMax * 8);
  });
  CBA_HIP(attr_rc);
  if (det_scale)
    hipLaunchKernelGGL(k_accumulate_points<true>, grid, dim3(kPointThreads), lds, s, a, al, rec_doubles, flags, jrec, cells, key_start, key_obs, L.n_points, nchunks,
                       chunk_cols, t, det_scale);
  else
    hipLaunchKernelGGL(k_accumulate_points<false>, grid, dim3(kPointThreads), lds, s, a, al, rec_doubles, flags, jrec, cells, key_start, key_obs, L.n_points, nchunks,
                       chunk_cols, t, det_scale);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ------------------------------------------------------------------------------------------------
// grid x grid block of JtJ, grouped by grid cell.  All observations whose 4x4 control patch starts at
// the same cell add their K_g x K_g products to the same entries of H_dd, so they are first bucketed by
// (camera, cell) with a counting sort and then one workgroup per cell sums its bucket in registers
// (528 entries for the central model, 3240 for the non-central one) and issues ONE atomic per entry.
// This removes 58 % (central) / 81 % (non-central) of the atomics of the per-observation scatter.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cell_count(PassArgs a, const uint8_t* __restrict__ flags, const int* __restrict__ cells,
                                                    const int* __restrict__ cell_base, int* __restrict__ count) {
  int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= a.n_obs || flags[o] != 3) return;
  const int cam = a.obs_camera[o];
  const int gw = a.cams[cam].gw;
  atomicAdd(count + cell_base[cam] + cells[2 * o + 1] * gw + cells[2 * o], 1);
}
__global__ void __launch_bounds__(1024) k_cell_scan(const int* __restrict__ count, int n, int* __restrict__ start) {
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + threadIdx.x;
    int v = (i < n) ? count[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      int t = (threadIdx.x >= off) ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < n) start[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) start[n] = carry;
}
__global__ void __launch_bounds__(256) k_cell_fill(PassArgs a, const uint8_t* __restrict__ flags, const int* __restrict__ cells,
                                                   const int* __restrict__ cell_base, const int* __restrict__ start,
                                                   int* __restrict__ fill, int* __restrict__ order) {
  int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= a.n_obs || flags[o] != 3) return;
  const int cam = a.obs_camera[o];
  const int key = cell_base[cam] + cells[2 * o + 1] * a.cams[cam].gw + cells[2 * o];
  order[start[key] + atomicAdd(fill + key, 1)] = (int)o;
}
// rig_row0 >= 0 (several cameras, poses eliminated): the bucket also sums the camera's rig-pose x grid block
// (6 x K_g entries of H_dd that EVERY observation of the camera would otherwise hit with atomics).
// One workgroup per cell.  The K_g (K_g + 1) / 2 pair sums are dealt to the 256 lanes (3 per lane central, 13 non-central),
// the records of the bucket are staged in LDS eight at a time (all loads of a stage in flight together), and every entry
// is summed in bucket order and added to H_dd with ONE atomic.  (The first version gave a cell to one wavefront: 51 pair
// sums per lane = 441 registers, one wavefront per SIMD walking ~200 records with a dependent global load each -- 3.1 ms
// at BASELINE configs[3].)
template <int PER, bool DET, bool RIG>
__global__ void __launch_bounds__(256) k_accumulate_cells(PassArgs a, int cam, int key0, int n_cells, int rec_doubles, int ld,
                                                          const double* __restrict__ jrec, const int* __restrict__ start,
                                                          const int* __restrict__ order, double* __restrict__ Hdd, int rig_row0,
                                                          const double* __restrict__ det_scale, double* __restrict__ bd) {
  typedef typename Acc<DET>::T acc_t;
  const double scale = DET ? det_scale[0] : 1.0;
  const double scale_b = DET ? det_scale[1] : 1.0;
  constexpr int KG = PER * 16;
  // Every lane owns one TZ x TZ tile of the upper triangle of the K_g x K_g block (4 x 4 of 80 x 80: 210 tiles, 2 x 2 of
  // 32 x 32: 136 tiles): per record 4 TZ LDS reads feed TZ^2 products, instead of four reads per product with the pairs
  // dealt out one by one (1.34 -> see DESIGN.md at BASELINE configs[3]).  Diagonal tiles compute their lower half too and
  // do not store it.
  constexpr int TZ = PER == 5 ? 4 : 2;
  constexpr int NT = KG / TZ;
  constexpr int NTILE = NT * (NT + 1) / 2;
  static_assert(NTILE <= 256 && KG % TZ == 0, "one tile per lane");
  constexpr int NR = (6 * KG + 255) / 256;
  constexpr int RB = 8;                          // records per stage
  __shared__ double sJ0[RB][KG];
  __shared__ double sJ1[RB][KG];
  __shared__ double sRig[RB][12];
  __shared__ double sW[RB];
  __shared__ double sRes[RB][2];
  const int tid = threadIdx.x;
  const int cell = blockIdx.x;
  if (cell >= n_cells) return;
  const int o_begin = start[key0 + cell], o_end = start[key0 + cell + 1];
  if (o_begin == o_end) return;                  // workgroup-uniform
  // The grid part of J^T r of the bucket as well (lanes 0 .. K_g - 1, one atomic per entry and cell: the per-observation
  // kernel used to issue K_g atomics per observation for it).
  acc_t bacc = 0;
  // this lane's tile (ti <= tk), enumerated row by row
  int ti = 0, tk = 0;
  const bool has_tile = tid < NTILE;
  if (has_tile) { int rem = tid; while (rem >= NT - ti) { rem -= NT - ti; ++ti; } tk = ti + rem; }
  const int i0 = ti * TZ, k0 = tk * TZ;
  acc_t acc[TZ][TZ], racc[NR];
#pragma unroll
  for (int x = 0; x < TZ; ++x)
#pragma unroll
    for (int y = 0; y < TZ; ++y) acc[x][y] = 0;
#pragma unroll
  for (int t = 0; t < NR; ++t) racc[t] = 0;
  constexpr bool rig = RIG;
  for (int base = o_begin; base < o_end; base += RB) {
    const int nrec = o_end - base < RB ? o_end - base : RB;
    __syncthreads();                             // the previous stage has been consumed
    for (int e = tid; e < nrec * KG; e += 256) {
      const int r = e / KG, k = e - r * KG;
      const double* rec = jrec + (size_t)order[base + r] * rec_doubles;
      sJ0[r][k] = rec[kRecHeader + k]; sJ1[r][k] = rec[kRecHeader + KG + k];
    }
    if (tid < nrec) sW[tid] = jrec[(size_t)order[base + tid] * rec_doubles + 2];
    if (tid >= 32 && tid < 32 + 2 * nrec) sRes[(tid - 32) >> 1][(tid - 32) & 1] = jrec[(size_t)order[base + ((tid - 32) >> 1)] * rec_doubles + ((tid - 32) & 1)];
    if (rig && tid >= 64 && tid < 64 + nrec * 12) {
      const int r = (tid - 64) / 12, q = (tid - 64) - r * 12;
      sRig[r][q] = jrec[(size_t)order[base + r] * rec_doubles + 15 + q];
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < nrec; ++r) {
      const double w = sW[r];
      if (tid < KG) bacc += Acc<DET>::from(sRes[r][0] * (w * sJ0[r][tid]) + sRes[r][1] * (w * sJ1[r][tid]), scale_b);
      if (has_tile) {
        double a0[TZ], a1[TZ], b0[TZ], b1[TZ];
#pragma unroll
        for (int x = 0; x < TZ; ++x) { a0[x] = sJ0[r][i0 + x]; a1[x] = sJ1[r][i0 + x]; b0[x] = sJ0[r][k0 + x]; b1[x] = sJ1[r][k0 + x]; }
#pragma unroll
        for (int x = 0; x < TZ; ++x)
#pragma unroll
          for (int y = 0; y < TZ; ++y) acc[x][y] += Acc<DET>::from(w * (a0[x] * b0[y] + a1[x] * b1[y]), scale);
      }
      if (rig) {
#pragma unroll
        for (int t = 0; t < NR; ++t) {
          const int e = tid + 256 * t;
          if (e < 6 * KG) { const int q = e / KG, k = e - q * KG; racc[t] += Acc<DET>::from(w * (sRig[r][q] * sJ0[r][k] + sRig[r][6 + q] * sJ1[r][k]), scale); }
        }
      }
    }
  }
  const CamDev cd = a.cams[cam];
  const int cy0 = cell / cd.gw, cx0 = cell - cy0 * cd.gw;
  if (tid < KG) {
    const int ck = tid / PER, dk = tid - ck * PER;
    Acc<DET>::add(bd + grid_column(cd, (cx0 + (ck & 3)) + (cy0 + (ck >> 2)) * cd.gw, dk), bacc);
  }
  if (has_tile) {
#pragma unroll
    for (int x = 0; x < TZ; ++x)
#pragma unroll
      for (int y = 0; y < TZ; ++y) {
        const int i = i0 + x, k = k0 + y;
        if (i > k) continue;                       // lower half of a diagonal tile
        const int ci = i / PER, di = i - ci * PER, ck = k / PER, dk = k - ck * PER;
        int row = grid_column(cd, (cx0 + (ci & 3)) + (cy0 + (ci >> 2)) * cd.gw, di);
        int col = grid_column(cd, (cx0 + (ck & 3)) + (cy0 + (ck >> 2)) * cd.gw, dk);
        if (row > col) { const int t2 = row; row = col; col = t2; }   // tiled order is not monotone in the patch order
        Acc<DET>::add(Hdd + (size_t)row * ld + col, acc[x][y]);
      }
  }
  if (!rig) return;
#pragma unroll
  for (int t = 0; t < NR; ++t) {
    const int e = tid + 256 * t;
    if (e >= 6 * KG) continue;
    const int q = e / KG, k = e - q * KG, ck = k / PER, dk = k - ck * PER;
    const int col = grid_column(cd, (cx0 + (ck & 3)) + (cy0 + (ck >> 2)) * cd.gw, dk);
    Acc<DET>::add(Hdd + (size_t)(rig_row0 + q) * ld + col, racc[t]);      // rig rows precede the grid columns
  }
}
int launch_accumulate_cells(const PassArgs& a, const std::vector<cba_camera>& cams, const std::vector<int>& cell_base_host,
                            int rec_doubles, int ld, const uint8_t* flags, const double* jrec, const int* cells,
                            const int* cell_base, int* count, int* start, int* fill, int* order, double* Hdd,
                            int rig_row_first /* dense row of camera 0's rig block, or -1 */, const double* det_scale, double* bd,
                            hipStream_t s) {
  if (a.n_obs == 0) return CBA_OK;
  const int n_keys = cell_base_host.back();
  CBA_HIP(hipMemsetAsync(count, 0, sizeof(int) * (size_t)n_keys, s));
  CBA_HIP(hipMemsetAsync(fill, 0, sizeof(int) * (size_t)n_keys, s));
  dim3 grid((unsigned)((a.n_obs + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_cell_count, grid, block, 0, s, a, flags, cells, cell_base, count);
  hipLaunchKernelGGL(k_cell_scan, dim3(1), dim3(1024), 0, s, count, n_keys, start);
  hipLaunchKernelGGL(k_cell_fill, grid, block, 0, s, a, flags, cells, cell_base, start, fill, order);
  for (size_t c = 0; c < cams.size(); ++c) {
    const int n_cells = cams[c].grid_w * cams[c].grid_h;
    dim3 g2((unsigned)n_cells);
    const int rr = rig_row_first >= 0 ? rig_row_first + 6 * (int)c : -1;
    const bool central = cams[c].model_type == CBA_CENTRAL_GENERIC;
#define CBA_CELLS2(PER_, DET_, RIG_) hipLaunchKernelGGL((k_accumulate_cells<PER_, DET_, RIG_>), g2, block, 0, s, a, (int)c, cell_base_host[c], \
                                                        n_cells, rec_doubles, ld, jrec, start, order, Hdd, rr, det_scale, bd)
#define CBA_CELLS(PER_, DET_) do { if (rr >= 0) CBA_CELLS2(PER_, DET_, true); else CBA_CELLS2(PER_, DET_, false); } while (0)
    if (central) { if (det_scale) CBA_CELLS(2, true); else CBA_CELLS(2, false); }
    else { if (det_scale) CBA_CELLS(5, true); else CBA_CELLS(5, false); }
#undef CBA_CELLS2
#undef CBA_CELLS
  }
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ------------------------------------------------------------------------------------------------
// Off-diagonal strips B_i (pose block i x dense columns), eliminate_points = 0.  Every entry of B_i gets
// ~8 contributions from different observations of imageset i (neighbouring pattern points share grid
// cells), far apart in the observation order, so per-observation atomics cannot merge them.  Here one
// workgroup owns (imageset, band of kStripBand dense columns): it scans the imageset's observations,
// sums the 6 x (3 + K_cell) products that fall into its band in LDS (ds_add_f64) and writes the band of
// the six rows with plain coalesced stores -- zeros included, so B needs no memset and receives no
// global atomics from these terms (210 of the ~350 per observation).
// ------------------------------------------------------------------------------------------------
constexpr int kStripBand = 1024;
// bit t of band_mask[o]: observation o couples its pose to a dense column of band t (<= 64 bands = 65 536 columns -- the
// 4-camera rig of BASELINE configs[4] has 42; wider systems fall back to "all bands")
__global__ void __launch_bounds__(256) k_strip_band_mask(PassArgs a, AccumLayout L, const uint8_t* __restrict__ flags,
                                                         const int* __restrict__ cells, unsigned long long* __restrict__ band_mask, int n_bands) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= a.n_obs) return;
  if (flags[o] != 3) { band_mask[o] = 0ull; return; }
  if (n_bands > 64) { band_mask[o] = ~0ull; return; }
  const CamDev cd = a.cams[a.obs_camera[o]];
  unsigned long long m = 0ull;
  const int pc = L.first_points - L.block_dof + 3 * a.obs_point[o];
  m |= 1ull << (pc / kStripBand); m |= 1ull << ((pc + 2) / kStripBand);
  if (!L.localize_only) {
    const int cx0 = cells[2 * o], cy0 = cells[2 * o + 1];
    for (int c = 0; c < 16; ++c) {
      const int col = grid_column(cd, (cx0 + (c & 3)) + (cy0 + (c >> 2)) * cd.gw, 0);
      m |= 1ull << (col / kStripBand); m |= 1ull << ((col + cd.params_per_point - 1) / kStripBand);
    }
  }
  band_mask[o] = m;
}
constexpr int kStripWaves = 8;
template <bool DET>
__global__ void __launch_bounds__(64 * kStripWaves) k_accumulate_strips(PassArgs a, AccumLayout L, int rec_doubles, const uint8_t* __restrict__ flags,
                                                            const double* __restrict__ jrec, const int* __restrict__ cells,
                                                            const unsigned long long* __restrict__ band_mask,
                                                            const int64_t* __restrict__ img_start, double* __restrict__ B, int ld,
                                                            const double* __restrict__ det_scale) {
  __shared__ double acc[6][kStripBand];      // DET: the same 8-byte slots hold fixed-point integers (zero bits = 0 in both)
  const double scale = DET ? *det_scale : 1.0;
  const int img = blockIdx.x, band = blockIdx.y;
  const int col_lo = band * kStripBand;
  const int col_hi = min(col_lo + kStripBand, ld);
  for (int i = threadIdx.x; i < 6 * kStripBand; i += 64 * kStripWaves) (&acc[0][0])[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long bit = 1ull << (band & 63);
  const int64_t o_begin = img_start[img], o_end = img_start[img + 1];
  // kStripWaves wavefronts.  Pass = 64 * kStripWaves consecutive observations: every wavefront loads the band masks of one group of 64
  // (one coalesced load instead of a chain of dependent L2 round trips) and publishes its ballot; then the matching
  // observations of the whole pass are dealt round-robin to the wavefronts (the matches of one band are neighbours
  // in the observation order, so whole groups would leave most wavefronts idle), all 64 lanes on an observation's
  // columns.
  __shared__ unsigned long long gmask[kStripWaves];
  for (int64_t p0 = o_begin; p0 < o_end; p0 += 64 * kStripWaves) {
    const int64_t mine = p0 + 64 * wv + lane;
    const bool hit = mine < o_end && (band_mask[mine] & bit);
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) gmask[wv] = bal;
    __syncthreads();
    int seen = 0;
    for (int gidx = 0; gidx < kStripWaves; ++gidx) {
      unsigned long long todo = gmask[gidx];
      while (todo) {
        const int idx = __builtin_ctzll(todo);
        todo &= todo - 1;
        if (((seen++) % kStripWaves) != wv) continue;
        const int64_t o = p0 + 64 * gidx + idx;
        const CamDev cd = a.cams[a.obs_camera[o]];
        const int per = cd.params_per_point;
        const int Kg = L.localize_only ? 0 : per * 16;
        const int nc = 3 + Kg;                              // dense columns of this observation coupled to the pose
        const double* rec = jrec + (size_t)o * rec_doubles;
        const int cx0 = cells[2 * o], cy0 = cells[2 * o + 1];
        const int point_col = L.first_points - L.block_dof + 3 * a.obs_point[o];
        const double w = rec[2];
        for (int c = lane; c < nc; c += 64) {
          int col; double j0, j1;
          if (c < 3) { col = point_col + c; j0 = rec[27 + c]; j1 = rec[30 + c]; }
          else {
            const int g = c - 3, cell = g / per, d = g - cell * per;
            col = grid_column(cd, (cx0 + (cell & 3)) + (cy0 + (cell >> 2)) * cd.gw, d);
            j0 = rec[kRecHeader + g]; j1 = rec[kRecHeader + Kg + g];
          }
          if (col < col_lo || col >= col_hi) continue;
          const double w0 = w * j0, w1 = w * j1;
#pragma unroll
          for (int k = 0; k < 6; ++k) Acc<DET>::add(&acc[k][col - col_lo], Acc<DET>::from(rec[3 + k] * w0 + rec[9 + k] * w1, scale));
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  const int slot = a.pose_slot ? a.pose_slot[img] : img;
  for (int k = 0; k < 6; ++k) {
    double* row = B + (size_t)(6 * slot + k) * ld + col_lo;
    // DET: the fixed-point integers as they are -- k_accumulate adds integer atomics on top (pose x rig-pose entries) and
    // the whole of B is converted afterwards (launch_det_convert)
    for (int c = threadIdx.x; c < col_hi - col_lo; c += 64 * kStripWaves) row[c] = acc[k][c];
  }
}
int launch_accumulate_strips(const PassArgs& a, const Layout& L, int n_images, int rec_doubles, const uint8_t* flags, const double* jrec,
                             const int* cells, unsigned long long* band_mask, const int64_t* img_start, double* B, int ld, const double* det_scale,
                             hipStream_t s) {
  if (n_images == 0) return CBA_OK;
  AccumLayout al;
  al.rig_in_state = L.rig_in_state; al.eliminate_points = L.eliminate_points; al.localize_only = L.localize_only;
  al.first_rig_tr_global = L.first_rig_tr_global; al.first_camera_tr_rig = L.first_camera_tr_rig;
  al.first_points = L.first_points; al.block_dof = L.block_dof; al.block_size = L.block_size; al.dense_dof = L.dense_dof;
  const int bands = (ld + kStripBand - 1) / kStripBand;
  if (a.n_obs > 0)
    hipLaunchKernelGGL(k_strip_band_mask, dim3((unsigned)((a.n_obs + 255) / 256)), dim3(256), 0, s, a, al, flags, cells, band_mask, bands);
  if (det_scale)
    hipLaunchKernelGGL(k_accumulate_strips<true>, dim3((unsigned)n_images, (unsigned)bands), dim3(64 * kStripWaves), 0, s, a, al, rec_doubles, flags,
                       jrec, cells, band_mask, img_start, B, ld, det_scale);
  else
    hipLaunchKernelGGL(k_accumulate_strips<false>, dim3((unsigned)n_images, (unsigned)bands), dim3(64 * kStripWaves), 0, s, a, al, rec_doubles, flags,
                       jrec, cells, band_mask, img_start, B, ld, det_scale);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

int launch_accumulate(const PassArgs& a, const Layout& L, int rec_doubles, const uint8_t* flags, const double* jrec,
                      const int* cells, const uint32_t* pair_tables, const int* pair_counts, AccumTargets t,
                      const double* det_scale, int points_separate, hipStream_t s) {
  if (a.n_obs == 0) return CBA_OK;
  AccumLayout al;
  al.rig_in_state = L.rig_in_state; al.eliminate_points = L.eliminate_points; al.localize_only = L.localize_only;
  al.first_rig_tr_global = L.first_rig_tr_global; al.first_camera_tr_rig = L.first_camera_tr_rig;
  al.first_points = L.first_points; al.block_dof = L.block_dof; al.block_size = L.block_size; al.dense_dof = L.dense_dof;
  const int chunk = points_separate ? kAccChunkHot : kAccChunk;
  const dim3 grid((unsigned)((a.n_obs + 4 * chunk - 1) / (4 * chunk)));
  if (det_scale)
    hipLaunchKernelGGL(k_accumulate<true>, grid, dim3(256), 0, s, a, al, rec_doubles, flags, jrec, cells, pair_tables, pair_counts, t, det_scale, points_separate);
  else
    hipLaunchKernelGGL(k_accumulate<false>, grid, dim3(256), 0, s, a, al, rec_doubles, flags, jrec, cells, pair_tables, pair_counts, t, det_scale, points_separate);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ---- deterministic mode: scale of the fixed-point accumulation, and the conversion back to doubles ----
// out_bits[0]: bit pattern of max over the observations with a Jacobian of  2 w |J|max^2  >= every |contribution| to H,
// out_bits[1]: the same for  2 w |J|max |r|max  >= every |contribution| to b  (the residuals are orders of magnitude below
// the Jacobian entries, so b gets its own, finer scale).  Positive doubles order like their bit patterns; max is order-independent.
__global__ void __launch_bounds__(256) k_det_bound(int64_t n, int rec_doubles, int used_doubles, const uint8_t* __restrict__ flags,
                                                   const double* __restrict__ jrec, unsigned long long* __restrict__ out_bits) {
  const int lane = threadIdx.x & 63;
  double m = 0.0, mb = 0.0;
  for (int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); o < n; o += (int64_t)gridDim.x * 4) {
    if (flags[o] != 3) continue;
    const double* rec = jrec + (size_t)o * rec_doubles;
    double jm = 0.0;
    for (int k = lane; k < used_doubles; k += 64) if (k > 2) jm = fmax(jm, fabs(rec[k]));
    for (int off = 32; off > 0; off >>= 1) jm = fmax(jm, __shfl_xor(jm, off, 64));
    m = fmax(m, 2.0 * rec[2] * jm * jm);
    mb = fmax(mb, 2.0 * rec[2] * jm * fmax(fabs(rec[0]), fabs(rec[1])));
  }
  if (lane == 0 && m > 0.0) atomicMax(out_bits, (unsigned long long)__double_as_longlong(m));
  if (lane == 0 && mb > 0.0) atomicMax(out_bits + 1, (unsigned long long)__double_as_longlong(mb));
}
__global__ void k_det_scale(const unsigned long long* __restrict__ bits, int64_t n_obs, double* __restrict__ scale) {
  for (int i = 0; i < 2; ++i) {
    const double m = __longlong_as_double((long long)bits[i]);
    const double bound = m * (double)(n_obs > 0 ? n_obs : 1);   // no entry receives more than n_obs contributions
    int e = 0;
    if (bound > 0.0) frexp(bound, &e);                          // bound < 2^e
    scale[i] = ldexp(1.0, 62 - e);                              // bound * scale < 2^62
  }
}
int launch_det_scale(int64_t n, int rec_doubles, int used_doubles, const uint8_t* flags, const double* jrec, unsigned long long* bits,
                     double* scale, hipStream_t s) {
  CBA_HIP(hipMemsetAsync(bits, 0, 2 * sizeof(unsigned long long), s));
  if (n > 0) hipLaunchKernelGGL(k_det_bound, dim3(1024), dim3(256), 0, s, n, rec_doubles, used_doubles, flags, jrec, bits);
  hipLaunchKernelGGL(k_det_scale, dim3(1), dim3(1), 0, s, bits, n, scale);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}
__global__ void __launch_bounds__(256) k_det_convert(double* __restrict__ p, size_t n, const double* __restrict__ det_scale) {
  const double inv = 1.0 / *det_scale;                          // power of two: exact
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    p[i] = (double)(*reinterpret_cast<const long long*>(p + i)) * inv;
}
int launch_det_convert(double* p, size_t n, const double* det_scale, hipStream_t s) {
  if (n == 0) return CBA_OK;
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_det_convert, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, p, n, det_scale);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ------------------------------------------------------------------------------------------------
// deterministic cost reductions (fixed assignment of observations to lanes, fixed trees)
// ------------------------------------------------------------------------------------------------
constexpr int kRedBlocks = 256;
__global__ void __launch_bounds__(256) k_reduce_costs_partial(const double* __restrict__ ref, const double* __restrict__ test,
                                                              const uint8_t* __restrict__ flags, int64_t n,
                                                              double* __restrict__ partials) {
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)kRedBlocks * 256) {
    double r = ref ? ref[i] : -1.0, t = test ? test[i] : -1.0;
    if (r >= 0) { acc[0] += r; acc[5] += 1; }
    if (t >= 0) { acc[1] += t; acc[6] += 1; }
    if (r >= 0 && t >= 0) { acc[2] += r; acc[3] += t; acc[4] += 1; }
    if (flags && flags[i] == 1) acc[7] += 1;
  }
  __shared__ double sh[8][256];
#pragma unroll
  for (int k = 0; k < 8; ++k) sh[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
#pragma unroll
      for (int k = 0; k < 8; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 8) partials[blockIdx.x * 8 + threadIdx.x] = sh[threadIdx.x][0];
}
__global__ void k_reduce_costs_final(const double* __restrict__ partials, double* __restrict__ out8) {
  int k = threadIdx.x;
  if (k >= 8) return;
  double s = 0;
  for (int b = 0; b < kRedBlocks; ++b) s += partials[b * 8 + k];
  out8[k] = s;
}
int launch_reduce_costs(const double* ref, const double* test, const uint8_t* flags, int64_t n, double* partials,
                        double* out8, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_costs_partial, dim3(kRedBlocks), dim3(256), 0, s, ref, test, flags, n, partials);
  hipLaunchKernelGGL(k_reduce_costs_final, dim3(1), dim3(64), 0, s, partials, out8);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ------------------------------------------------------------------------------------------------
// state update: state_out = state_in - x
// ------------------------------------------------------------------------------------------------
// ApplyLocalUpdateToQuaternion incl. the fp32-typed norm / sinc (quaternion_parametrization.h:39-61),
// then SE3d(q, t) normalises (so3.hpp:536-541).
__device__ __forceinline__ void pose_minus(const double* in, const double* d, double* out) {
  double u[3] = {-d[0], -d[1], -d[2]};
  const float n = (float)sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  double q[4];
  if (n == 0.0f) {
    q[0] = in[0]; q[1] = in[1]; q[2] = in[2]; q[3] = in[3];
  } else {
    // fp32 sin/cos evaluated via fp64 and rounded once (faithfully rounded fp32 result)
    const float sn = (float)sin((double)n), cs = (float)cos((double)n);
    const float sbu = sn / n;
    double uq[4] = {(double)cs, (double)sbu * u[0], (double)sbu * u[1], (double)sbu * u[2]};
    quat_mul(uq, in, q);
  }
  double len = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  out[0] = q[0] / len; out[1] = q[1] / len; out[2] = q[2] / len; out[3] = q[3] / len;
  out[4] = in[4] - d[3]; out[5] = in[5] - d[4]; out[6] = in[6] - d[5];
}
__global__ void k_update_poses(const double* __restrict__ in, const double* __restrict__ x, int n, double* __restrict__ out,
                               int apply, const int* __restrict__ slot) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (apply) {
    pose_minus(in + 7 * (size_t)i, x + 6 * (size_t)(slot ? slot[i] : i), out + 7 * (size_t)i);
  } else {
    for (int k = 0; k < 7; ++k) out[7 * (size_t)i + k] = in[7 * (size_t)i + k];
  }
}
__global__ void k_update_points(const double* __restrict__ in, const double* __restrict__ x, int n, double* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = in[i] - x[i];
}
// SubtractDelta: central_grid.h:168-184 / noncentral_generic.h:195-219 (tangents recomputed from the
// current direction, full renormalisation)
__global__ void k_update_grid(const double* __restrict__ in, const double* __restrict__ x, int G, int per, int apply,
                              const int* __restrict__ gperm, double* __restrict__ out) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  double d[3] = {in[3 * g], in[3 * g + 1], in[3 * g + 2]};
  if (!apply) {
    out[3 * g] = d[0]; out[3 * g + 1] = d[1]; out[3 * g + 2] = d[2];
    if (per == 5) for (int k = 0; k < 3; ++k) out[3 * (size_t)G + 3 * g + k] = in[3 * (size_t)G + 3 * g + k];
    return;
  }
  double t1[3], t2[3];
  tangents_of(d, t1, t2);
  const double* dx = x + (size_t)per * (gperm ? gperm[g] : g);
  double o1 = -dx[0], o2 = -dx[1];
  double nd[3] = {d[0] + o1 * t1[0] + o2 * t2[0], d[1] + o1 * t1[1] + o2 * t2[1], d[2] + o1 * t1[2] + o2 * t2[2]};
  normalize3(nd[0], nd[1], nd[2]);
  out[3 * g] = nd[0]; out[3 * g + 1] = nd[1]; out[3 * g + 2] = nd[2];
  if (per == 5) {
    double o3 = -dx[2], o4 = -dx[3], o5 = -dx[4];
    const double* oi = in + 3 * (size_t)G + 3 * g;
    double* oo = out + 3 * (size_t)G + 3 * g;
    for (int k = 0; k < 3; ++k) oo[k] = oi[k] + o3 * t1[k] + o4 * t2[k] + o5 * d[k];
  }
}
int launch_apply_update(const Layout& L, const std::vector<cba_camera>& cams, const DevState& in, const double* x,
                        DevState& out, const int* pose_slot, int* const* gperm, hipStream_t s) {
  int N = L.n_images, C = L.n_cameras, P = L.n_points;
  if (N > 0)
    hipLaunchKernelGGL(k_update_poses, dim3((N + 255) / 256), dim3(256), 0, s, in.rig_tr_global,
                       x + L.first_rig_tr_global, N, out.rig_tr_global, 1, pose_slot);
  hipLaunchKernelGGL(k_update_poses, dim3((C + 255) / 256), dim3(256), 0, s, in.camera_tr_rig,
                     x + (L.rig_in_state ? L.first_camera_tr_rig : 0), C, out.camera_tr_rig, L.rig_in_state, (const int*)nullptr);
  if (P > 0)     // a problem without pattern points (n_points = 0 is accepted by cba_create) must not launch an empty grid
    hipLaunchKernelGGL(k_update_points, dim3((3 * P + 255) / 256), dim3(256), 0, s, in.points, x + L.first_points, 3 * P,
                       out.points);
  for (int c = 0; c < C; ++c) {
    int G = cams[c].grid_w * cams[c].grid_h;
    int per = cams[c].model_type == CBA_CENTRAL_GENERIC ? 2 : 5;
    hipLaunchKernelGGL(k_update_grid, dim3((G + 255) / 256), dim3(256), 0, s, in.grids[c],
                       x + (L.localize_only ? 0 : L.block_dof + L.intr_offset[c]), G, per, L.localize_only ? 0 : 1,
                       gperm ? gperm[c] : nullptr, out.grids[c]);
  }
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// direction grid -= x in its local parametrisation (DirectionGridStateWithLocalUpdates::operator-=,
// central_generic.cc:65-80: the same tangent-plane update as SubtractDelta)
int launch_update_direction_grid(const double* in, const double* x, int G, double* out, hipStream_t s) {
  if (G == 0) return CBA_OK;
  hipLaunchKernelGGL(k_update_grid, dim3((G + 255) / 256), dim3(256), 0, s, in, x, G, 2, 1, (const int*)nullptr, out);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

// ------------------------------------------------------------------------------------------------
// stateless model-level kernels (cba_project / cba_unproject)
// ------------------------------------------------------------------------------------------------
template <int MODEL>
__global__ void __launch_bounds__(256) k_project_points(const CamDev* __restrict__ camp, int64_t n,
                                                        const double* __restrict__ local, const double* __restrict__ init,
                                                        double* __restrict__ pixels, uint8_t* __restrict__ ok) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const CamDev c = *camp;
  Subst none; none.index = -1;
  double px, py;
  if (init) { px = init[2 * i]; py = init[2 * i + 1]; }
  else center_pixel(c, px, py);
  double lp[3] = {local[3 * i], local[3 * i + 1], local[3 * i + 2]};
  bool r = in_calibrated_area(c, px, py) && project_point<MODEL>(c, none, lp, px, py);
  pixels[2 * i] = px; pixels[2 * i + 1] = py;
  ok[i] = r ? 1 : 0;
}
int launch_project_points(const CamDev* cam_dev, int model, int64_t n, const double* local, const double* init,
                          double* pixels, uint8_t* ok, hipStream_t s) {
  if (n == 0) return CBA_OK;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (model == kCentral) hipLaunchKernelGGL(k_project_points<kCentral>, grid, block, 0, s, cam_dev, n, local, init, pixels, ok);
  else hipLaunchKernelGGL(k_project_points<kNoncentral>, grid, block, 0, s, cam_dev, n, local, init, pixels, ok);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

template <int MODEL>
__global__ void __launch_bounds__(256) k_unproject(const CamDev* __restrict__ camp, int64_t n, const double* __restrict__ pixels,
                                                   double* __restrict__ lines, double* __restrict__ jac,
                                                   uint8_t* __restrict__ ok) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const CamDev c = *camp;
  Subst none; none.index = -1;
  double d[3] = {0, 0, 0}, o[3] = {0, 0, 0}, jd[6] = {0, 0, 0, 0, 0, 0}, jo[6] = {0, 0, 0, 0, 0, 0};
  bool r;
  if (jac) r = unproject_jac<MODEL>(c, none, pixels[2 * i], pixels[2 * i + 1], d, o, jd, jo);
  else r = unproject<MODEL>(c, none, pixels[2 * i], pixels[2 * i + 1], d, o);
  for (int k = 0; k < 3; ++k) { lines[6 * i + k] = d[k]; lines[6 * i + 3 + k] = (MODEL == kNoncentral) ? o[k] : 0.0; }
  if (jac)
    for (int k = 0; k < 6; ++k) { jac[12 * i + k] = jd[k]; jac[12 * i + 6 + k] = (MODEL == kNoncentral) ? jo[k] : 0.0; }
  ok[i] = r ? 1 : 0;
}
int launch_unproject(const CamDev* cam_dev, int model, int64_t n, const double* pixels, double* lines, double* jac,
                     uint8_t* ok, hipStream_t s) {
  if (n == 0) return CBA_OK;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (model == kCentral) hipLaunchKernelGGL(k_unproject<kCentral>, grid, block, 0, s, cam_dev, n, pixels, lines, jac, ok);
  else hipLaunchKernelGGL(k_unproject<kNoncentral>, grid, block, 0, s, cam_dev, n, pixels, lines, jac, ok);
  CBA_HIP(hipGetLastError());
  return CBA_OK;
}

}  // namespace cba
