// Stand-in for <libvis/sophus.h> -- TEST INFRASTRUCTURE ONLY.  The reference header pulls in Sophus (absent from this image) and
// exports SE3d & friends into vis.  The boundary compile proof (tests/test_integration_patch.py: the adapter of
// integration/reference.patch against the reference's REAL dataset.h / ba_state.h / camera_model.h) needs the members those
// headers and the adapter touch: construction from (quaternion, translation), unit_quaternion(), translation(),
// rotationMatrix(), composition, inverse -- written over this directory's Eigen stand-in with Sophus' conventions
// (T = [R(q) | t], (a * b)(x) = a(b(x))).
#ifndef CBA_REF_SHIM_LIBVIS_SOPHUS_
#define CBA_REF_SHIM_LIBVIS_SOPHUS_
#include <Eigen/Geometry>
#include "libvis/libvis.h"
namespace Sophus {
template <class T>
class SE3 {
 public:
  typedef Eigen::Quaternion<T> Quat;
  typedef Eigen::Matrix<T, 3, 1> Vec3;
  typedef Eigen::Matrix<T, 3, 3> Mat3;
  SE3() : q_(1, 0, 0, 0) { t_(0) = 0; t_(1) = 0; t_(2) = 0; }
  SE3(const Quat& q, const Vec3& t) : q_(q), t_(t) {}
  const Quat& unit_quaternion() const { return q_; }
  const Vec3& translation() const { return t_; }
  Vec3& translation() { return t_; }
  Mat3 rotationMatrix() const {
    const T w = q_.w(), x = q_.x(), y = q_.y(), z = q_.z();
    Mat3 R;
    R(0, 0) = 1 - 2 * (y * y + z * z); R(0, 1) = 2 * (x * y - z * w);     R(0, 2) = 2 * (x * z + y * w);
    R(1, 0) = 2 * (x * y + z * w);     R(1, 1) = 1 - 2 * (x * x + z * z); R(1, 2) = 2 * (y * z - x * w);
    R(2, 0) = 2 * (x * z - y * w);     R(2, 1) = 2 * (y * z + x * w);     R(2, 2) = 1 - 2 * (x * x + y * y);
    return R;
  }
  Vec3 operator*(const Vec3& p) const {
    const Mat3 R = rotationMatrix();
    Vec3 r;
    for (int i = 0; i < 3; ++i) r(i) = R(i, 0) * p(0) + R(i, 1) * p(1) + R(i, 2) * p(2) + t_(i);
    return r;
  }
  SE3 operator*(const SE3& b) const { return SE3(q_ * b.q_, (*this) * b.t_); }
  SE3 inverse() const {
    const Quat qi(q_.w(), -q_.x(), -q_.y(), -q_.z());
    SE3 r(qi, Vec3());
    const Mat3 R = r.rotationMatrix();
    for (int i = 0; i < 3; ++i) r.t_(i) = -(R(i, 0) * t_(0) + R(i, 1) * t_(1) + R(i, 2) * t_(2));
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
