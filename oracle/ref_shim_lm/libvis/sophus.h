// Stand-in for <libvis/sophus.h> next to ref_shim_lm/Eigen -- TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Sophus::SE3 as vendored by
// the reference (libvis/third_party/sophus/sophus/se3.hpp, so3.hpp): unit quaternion + translation; group product = quaternion product +
// rotated translation; SE3(rotation matrix, translation) converts the matrix with Eigen's matrix -> quaternion rule.
#ifndef CBA_REF_SHIM_LM_LIBVIS_SOPHUS_
#define CBA_REF_SHIM_LM_LIBVIS_SOPHUS_
#include <Eigen/Geometry>
#include "libvis/libvis.h"
namespace Sophus {
template <class T>
class SE3 {
 public:
  typedef Eigen::Quaternion<T> Quat;
  typedef Eigen::Matrix<T, 3, 1> Vec3;
  typedef Eigen::Matrix<T, 3, 3> Mat3;
  static constexpr int DoF = 6;
  SE3() : q_(1, 0, 0, 0) {}
  SE3(const Quat& q, const Vec3& t) : q_(q), t_(t) { q_.normalize(); }
  SE3(const Mat3& R, const Vec3& t) : q_(R), t_(t) {}
  void setRotationMatrix(const Mat3& R) { q_ = Quat(R); }
  const Quat& unit_quaternion() const { return q_; }
  const Vec3& translation() const { return t_; }
  Vec3& translation() { return t_; }
  Mat3 rotationMatrix() const { return q_.toRotationMatrix(); }
  Vec3 operator*(const Vec3& p) const { return Vec3(rotationMatrix() * p + t_); }
  // se3.hpp: inverse = (R^-1, R^-1 * (t * -1)); so3.hpp: the inverse of a unit quaternion is its conjugate
  SE3 inverse() const {
    SE3 r; r.q_ = Quat(q_.w(), -q_.x(), -q_.y(), -q_.z());
    r.t_ = Vec3(r.q_.toRotationMatrix() * Vec3(t_ * T(-1)));
    return r;
  }
  // so3.hpp:215-232: quaternion product, renormalised by the first-order rule when the squared norm left 1
  SE3 operator*(const SE3& b) const {
    Quat q = q_ * b.q_;
    const T sn = q.squaredNorm();
    if (sn != T(1)) { const T s = T(2) / (T(1) + sn); q = Quat(q.w() * s, q.x() * s, q.y() * s, q.z() * s); }
    SE3 r; r.q_ = q; r.t_ = (*this) * b.t_;
    return r;
  }
 private:
  Quat q_;
  Vec3 t_;
};
typedef SE3<double> SE3d;
typedef SE3<float> SE3f;
}  // namespace Sophus
namespace vis {
using Sophus::SE3d;
using Sophus::SE3f;
}
#endif
