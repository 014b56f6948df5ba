// Minimal value types of the reference's boundary (Eigen / Sophus are not available here): just enough
// of Vec2f/Vec2d/Vec3d, Quaterniond and SE3d for Dataset / BAState / CameraModel to keep their shape.
// Semantics follow the vendored Sophus (libvis/third_party/sophus/sophus/se3.hpp:183-207, so3.hpp:159-232
// in the reference tree): unit quaternion + translation, normalised on construction.
#pragma once
#include <cmath>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace vis {

typedef unsigned int u32;
typedef size_t usize;

struct Vec2f { float v[2] = {0, 0}; float& x() { return v[0]; } float& y() { return v[1]; } float x() const { return v[0]; } float y() const { return v[1]; } Vec2f() {} Vec2f(float a, float b) { v[0] = a; v[1] = b; } };
struct Vec2i { int v[2] = {0, 0}; int& x() { return v[0]; } int& y() { return v[1]; } int x() const { return v[0]; } int y() const { return v[1]; } Vec2i() {} Vec2i(int a, int b) { v[0] = a; v[1] = b; } };
struct Vec2d {
  double v[2] = {0, 0};
  Vec2d() {}
  Vec2d(double a, double b) { v[0] = a; v[1] = b; }
  double& x() { return v[0]; } double& y() { return v[1]; }
  double x() const { return v[0]; } double y() const { return v[1]; }
  static Vec2d Zero() { return Vec2d(0, 0); }
  bool hasNaN() const { return v[0] != v[0] || v[1] != v[1]; }
};
struct Vec3d {
  double v[3] = {0, 0, 0};
  Vec3d() {}
  Vec3d(double a, double b, double c) { v[0] = a; v[1] = b; v[2] = c; }
  double& x() { return v[0]; } double& y() { return v[1]; } double& z() { return v[2]; }
  double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; }
  double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; }
  static Vec3d Zero() { return Vec3d(0, 0, 0); }
  Vec3d operator+(const Vec3d& o) const { return Vec3d(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  Vec3d operator-(const Vec3d& o) const { return Vec3d(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
  Vec3d operator*(double s) const { return Vec3d(v[0] * s, v[1] * s, v[2] * s); }
  double dot(const Vec3d& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  double norm() const { return std::sqrt(dot(*this)); }
  Vec3d normalized() const { double n = norm(); return n > 0 ? Vec3d(v[0] / n, v[1] / n, v[2] / n) : *this; }
};
struct Line3d {  // ParametrizedLine<double,3>
  Vec3d o, d;
  Vec3d& origin() { return o; } Vec3d& direction() { return d; }
  const Vec3d& origin() const { return o; } const Vec3d& direction() const { return d; }
};

struct Quaterniond {
  double w_ = 1, x_ = 0, y_ = 0, z_ = 0;
  Quaterniond() {}
  Quaterniond(double w, double x, double y, double z) : w_(w), x_(x), y_(y), z_(z) {}  // w first, as Eigen
  double w() const { return w_; } double x() const { return x_; } double y() const { return y_; } double z() const { return z_; }
  Quaterniond operator*(const Quaterniond& b) const {
    return Quaterniond(w_ * b.w_ - x_ * b.x_ - y_ * b.y_ - z_ * b.z_, w_ * b.x_ + x_ * b.w_ + y_ * b.z_ - z_ * b.y_,
                       w_ * b.y_ + y_ * b.w_ + z_ * b.x_ - x_ * b.z_, w_ * b.z_ + z_ * b.w_ + x_ * b.y_ - y_ * b.x_);
  }
  Vec3d rotate(const Vec3d& p) const {  // Eigen _transformVector
    Vec3d q(x_, y_, z_);
    Vec3d uv(q.y() * p.z() - q.z() * p.y(), q.z() * p.x() - q.x() * p.z(), q.x() * p.y() - q.y() * p.x());
    uv = uv * 2.0;
    Vec3d c(q.y() * uv.z() - q.z() * uv.y(), q.z() * uv.x() - q.x() * uv.z(), q.x() * uv.y() - q.y() * uv.x());
    return p + uv * w_ + c;
  }
};

class SE3d {
 public:
  static constexpr int DoF = 6;
  SE3d() {}
  SE3d(const Quaterniond& q, const Vec3d& t) : q_(q), t_(t) {
    double n = std::sqrt(q.w() * q.w() + q.x() * q.x() + q.y() * q.y() + q.z() * q.z());
    q_ = Quaterniond(q.w() / n, q.x() / n, q.y() / n, q.z() / n);
  }
  const Quaterniond& unit_quaternion() const { return q_; }
  const Vec3d& translation() const { return t_; }
  Vec3d& translation() { return t_; }
  SE3d operator*(const SE3d& o) const {
    SE3d r;
    r.t_ = t_ + q_.rotate(o.t_);
    Quaterniond q = q_ * o.q_;
    double sn = q.w() * q.w() + q.x() * q.x() + q.y() * q.y() + q.z() * q.z();
    if (sn != 1.0) { double s = 2.0 / (1.0 + sn); q = Quaterniond(q.w() * s, q.x() * s, q.y() * s, q.z() * s); }
    r.q_ = q;
    return r;
  }
  Vec3d operator*(const Vec3d& p) const { return q_.rotate(p) + t_; }
  // se3.hpp: (R^-1, R^-1 * (t * -1)); the inverse of a unit quaternion is its conjugate
  SE3d inverse() const {
    SE3d r;
    r.q_ = Quaterniond(q_.w(), -q_.x(), -q_.y(), -q_.z());
    r.t_ = r.q_.rotate(Vec3d(-t_.x(), -t_.y(), -t_.z()));
    return r;
  }
 private:
  Quaterniond q_;
  Vec3d t_;
};

// Image<Vec3d> stand-in: row-major, index x + y*width (libvis/src/libvis/image.h)
template <typename T>
class Image {
 public:
  Image() {}
  Image(u32 w, u32 h) { SetSize(w, h); }
  void SetSize(u32 w, u32 h) { w_ = w; h_ = h; d_.assign((size_t)w * h, T()); }
  u32 width() const { return w_; } u32 height() const { return h_; }
  T& operator()(u32 x, u32 y) { return d_[x + (size_t)y * w_]; }
  const T& operator()(u32 x, u32 y) const { return d_[x + (size_t)y * w_]; }
  T& at(u32 x, u32 y) { return (*this)(x, y); }
  const T& at(u32 x, u32 y) const { return (*this)(x, y); }
  T* data() { return d_.data(); } const T* data() const { return d_.data(); }
 private:
  u32 w_ = 0, h_ = 0;
  std::vector<T> d_;
};

}  // namespace vis
