// oracle/_ref, part 2 -- TEST INFRASTRUCTURE ONLY.
//
// C entry points over pieces of the reference APPLICATION's hot path that compile from their own files
// (APP = /root/reference/applications/camera_calibration/src/camera_calibration, LV = /root/reference/libvis/src/libvis):
//   APP/bundle_adjustment/joint_optimization_jacobians.h   ComputeJacobian, ComputeRigJacobian      (row A4)
//   APP/models/central_generic_jacobians.cc                CentralGenericBSpline_Unproject_...      (row M3)
//   APP/models/noncentral_generic_jacobians.cc             NoncentralGenericBSpline_Unproject_...   (row N2)
//   APP/local_parametrizations/{line,direction,quaternion}_parametrization.h                        (rows P1-P3)
//   APP/b_spline.h                                         EvalUniformCubicBSplineSurface (+ slow)  (row M1)
//   LV/loss_functions.h                                    HuberLoss                                (row B1)
// The files are #included from where they lie (nothing is copied); Eigen, libvis and cuda_runtime.h are replaced by
// the stand-ins in oracle/ref_shim.  The rest of the path (joint_optimization.cc, lm_optimizer.h, central_generic.cc)
// needs the real Eigen (LDLT, dynamic matrices), Sophus, Qt and glog and is unbuildable here.
#include <cuda_runtime.h>
#include <libvis/eigen.h>
#include <libvis/image.h>
#include <libvis/libvis.h>

#include "camera_calibration/b_spline.h"
#include "camera_calibration/bundle_adjustment/joint_optimization_jacobians.h"
#include "camera_calibration/io/io_util.h"
#include "camera_calibration/local_parametrizations/direction_parametrization.h"
#include "camera_calibration/local_parametrizations/line_parametrization.h"
#include "camera_calibration/local_parametrizations/quaternion_parametrization.h"
#include "camera_calibration/models/central_generic_jacobians.cc"
#include "camera_calibration/models/noncentral_generic_jacobians.cc"
#include "libvis/loss_functions.h"

using namespace vis;

extern "C" {

// ComputeJacobian: rows are d(local x|y|z) / d(q w x y z, t x y z, p x y z), 3 x 10 row-major
void ref_compute_jacobian(const double* q_wxyz, const double* p, double* jac30) {
  ComputeJacobian<double>(q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3], p[0], p[1], p[2], jac30, jac30 + 10, jac30 + 20);
}
// ComputeRigJacobian: 3 x 17 row-major: d / d(ctr q4 t3, rtg q4 t3, p3)
void ref_compute_rig_jacobian(const double* ctr_q, const double* p, const double* rtg_q, const double* rtg_t, double* jac51) {
  ComputeRigJacobian<double>(ctr_q[0], ctr_q[1], ctr_q[2], ctr_q[3], p[0], p[1], p[2], rtg_q[0], rtg_q[1], rtg_q[2], rtg_q[3],
                             rtg_t[0], rtg_t[1], rtg_t[2], jac51, jac51 + 17, jac51 + 34);
}
// patch: 16 control points, [y][x][3]
void ref_central_unproject_patch(double frac_x, double frac_y, const double* patch48, double* dir3, double* jac6) {
  Vec3d p[4][4];
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x) p[y][x] = Vec3d(patch48[3 * (4 * y + x)], patch48[3 * (4 * y + x) + 1], patch48[3 * (4 * y + x) + 2]);
  Matrix<double, 3, 1> r;
  Matrix<double, 3, 2> J;
  CentralGenericBSpline_Unproject_ComputeResidualAndJacobian<double>(frac_x, frac_y, p, &r, &J);
  for (int i = 0; i < 3; ++i) { dir3[i] = r(i); jac6[2 * i] = J(i, 0); jac6[2 * i + 1] = J(i, 1); }
}
// patch: 16 lines, [y][x][6] = direction(3), origin(3)
void ref_noncentral_unproject_patch(double frac_x, double frac_y, const double* patch96, double* line6, double* jac12) {
  Matrix<double, 6, 1> l[4][4];
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x)
      for (int d = 0; d < 6; ++d) l[y][x](d) = patch96[6 * (4 * y + x) + d];
  ParametrizedLine<double, 3> r;
  Matrix<double, 6, 2> J;
  NoncentralGenericBSpline_Unproject_ComputeResidualAndJacobian<double>(frac_x, frac_y, l, &r, &J);
  for (int i = 0; i < 3; ++i) { line6[i] = r.direction()(i); line6[3 + i] = r.origin()(i); }
  for (int i = 0; i < 6; ++i) { jac12[2 * i] = J(i, 0); jac12[2 * i + 1] = J(i, 1); }
}
void ref_tangents(const double* dir, double* t1, double* t2) {
  LineTangents t;
  ComputeTangentsForDirectionOrLine(Vec3d(dir[0], dir[1], dir[2]), &t);
  for (int i = 0; i < 3; ++i) { t1[i] = t.t1(i); t2[i] = t.t2(i); }
}
void ref_tangents_jacobian(const double* dir, double* jac18) {   // 6 x 3 row-major
  Matrix<double, 6, 3> J;
  TangentsJacobianWrtLineDirection(Vec3d(dir[0], dir[1], dir[2]), &J);
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) jac18[3 * r + c] = J(r, c);
}
void ref_apply_direction_update(const double* dir, double o1, double o2, double* out) {
  Vec3d d(dir[0], dir[1], dir[2]);
  DirectionTangents t;
  ComputeTangentsForDirectionOrLine(d, &t);
  ApplyLocalUpdateToDirection(&d, t, o1, o2);
  for (int i = 0; i < 3; ++i) out[i] = d(i);
}
void ref_apply_line_update(const double* line6 /* direction, origin */, const double* o5, double* out6) {
  ParametrizedLine<double, 3> l(Vec3d(line6[3], line6[4], line6[5]), Vec3d(line6[0], line6[1], line6[2]));
  LineTangents t;
  ComputeTangentsForDirectionOrLine(l.direction(), &t);
  ApplyLocalUpdateToLine(&l, t, o5[0], o5[1], o5[2], o5[3], o5[4]);
  for (int i = 0; i < 3; ++i) { out6[i] = l.direction()(i); out6[3 + i] = l.origin()(i); }
}
void ref_local_update_jacobian_wrt_direction(const double* dir, double* jac6) {   // 2 x 3 row-major
  Vec3d d(dir[0], dir[1], dir[2]);
  DirectionTangents t;
  ComputeTangentsForDirectionOrLine(d, &t);
  Matrix<double, 2, 3> J;
  LocalUpdateJacobianWrtDirection(d, t, &J);
  for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) jac6[3 * r + c] = J(r, c);
}
void ref_convert_direction_to_local_update(const double* base, const double* target, double* o2) {
  Vec3d b(base[0], base[1], base[2]);
  DirectionTangents t;
  ComputeTangentsForDirectionOrLine(b, &t);
  ConvertDirectionToLocalUpdate(b, Vec3d(target[0], target[1], target[2]), t, &o2[0], &o2[1]);
}
// ApplyLocalUpdateToQuaternion with Eigen's (un-normalised) Hamilton product; Sophus' renormalisation on top of it
// (libvis/third_party/sophus/sophus/so3.hpp) is outside this function in the reference as well.
void ref_apply_quaternion_update(const double* q_wxyz, const double* u3, double* out_wxyz) {
  Quaterniond q(q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3]);
  Quaterniond r = ApplyLocalUpdateToQuaternion(q, Vec3d(u3[0], u3[1], u3[2]));
  out_wxyz[0] = r.w(); out_wxyz[1] = r.x(); out_wxyz[2] = r.y(); out_wxyz[3] = r.z();
}
void ref_quaternion_jacobian(const double* q_wxyz, double* jac12) {   // 4 x 3 row-major
  Matrix<double, 4, 3> J;
  QuaternionJacobianWrtLocalUpdate(Quaterniond(q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3]), &J);
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) jac12[3 * r + c] = J(r, c);
}
// control net: w x h points of `dim` (2 or 3) floats/doubles given as doubles, row-major
void ref_bspline_surface(const double* ctrl, int w, int h, int dim, double x, double y, int slow, double* out) {
  if (dim == 3) {
    Image<Vec3d> img(w, h);
    for (int yy = 0; yy < h; ++yy) for (int xx = 0; xx < w; ++xx) img(xx, yy) = Vec3d(ctrl[3 * (xx + yy * w)], ctrl[3 * (xx + yy * w) + 1], ctrl[3 * (xx + yy * w) + 2]);
    Vec3d r = slow ? EvalUniformCubicBSplineSurfaceGenericSlow(img, x, y) : EvalUniformCubicBSplineSurface(img, x, y);
    for (int i = 0; i < 3; ++i) out[i] = r(i);
  } else {
    Image<Vec2d> img(w, h);
    for (int yy = 0; yy < h; ++yy) for (int xx = 0; xx < w; ++xx) img(xx, yy) = Vec2d(ctrl[2 * (xx + yy * w)], ctrl[2 * (xx + yy * w) + 1]);
    Vec2d r = slow ? EvalUniformCubicBSplineSurfaceGenericSlow(img, x, y) : EvalUniformCubicBSplineSurface(img, x, y);
    for (int i = 0; i < 2; ++i) out[i] = r(i);
  }
}
// the reference's BSpline.SlowFastAlgorithmConsistency net is Vec2f: float arithmetic on the control points
void ref_bspline_surface_f32(const float* ctrl, int w, int h, double x, double y, int slow, float* out2) {
  Image<Vec2f> img(w, h);
  for (int yy = 0; yy < h; ++yy) for (int xx = 0; xx < w; ++xx) img(xx, yy) = Vec2f(ctrl[2 * (xx + yy * w)], ctrl[2 * (xx + yy * w) + 1]);
  Vec2f r = slow ? EvalUniformCubicBSplineSurfaceGenericSlow(img, x, y) : EvalUniformCubicBSplineSurface(img, x, y);
  out2[0] = r(0); out2[1] = r(1);
}
double ref_huber_cost_sq(double sq, double k) { return HuberLoss<double>(k).ComputeCostFromSquaredResidual(sq); }
double ref_huber_weight_sq(double sq, double k) { return HuberLoss<double>(k).ComputeWeightFromSquaredResidual(sq); }
double ref_huber_cost(double r, double k) { return HuberLoss<double>(k).ComputeCost(r); }
double ref_huber_weight(double r, double k) { return HuberLoss<double>(k).ComputeWeight(r); }

// The reference's binary serialisation primitives (APP/io/io_util.h:37-125: integers in network byte order, floats raw).
// `kind`: 0 = u32, 1 = i32, 2 = float.  Appends to `path`; used by the tests to assemble a dataset.bin with the reference's
// own writers in the order of SaveDataset (APP/io/calibration_io.cc:51-137) and to read one back.
int ref_io_append(const char* path, int kind, double value) {
  FILE* f = fopen(path, "ab");
  if (!f) return -1;
  if (kind == 0) { u32 v = (u32)value; write_one(&v, f); }
  else if (kind == 1) { i32 v = (i32)value; write_one(&v, f); }
  else { float v = (float)value; write_one(&v, f); }
  fclose(f);
  return 0;
}
int ref_io_read_at(const char* path, long offset, int kind, double* value) {
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  fseek(f, offset, SEEK_SET);
  if (kind == 0) { u32 v = 0; read_one(&v, f); *value = v; }
  else if (kind == 1) { i32 v = 0; read_one(&v, f); *value = v; }
  else { float v = 0; read_one(&v, f); *value = v; }
  fclose(f);
  return 0;
}

}  // extern "C"
