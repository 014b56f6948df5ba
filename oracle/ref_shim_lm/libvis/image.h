// Stand-in for <libvis/image.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref build): storage, element access, SetTo; Write() writes nothing
#ifndef CBA_REF_SHIM_LM_LIBVIS_IMAGE_
#define CBA_REF_SHIM_LM_LIBVIS_IMAGE_
#include <string>
#include <vector>
#include <Eigen/Core>
#include "libvis/libvis.h"
namespace vis {
template <class T>
class Image {
 public:
  Image() : w_(0), h_(0) {}
  Image(int w, int h) : w_(w), h_(h), d_((std::size_t)w * h) {}
  template <class D> explicit Image(const Eigen::MatrixBase<D>& size) : w_(size(0)), h_(size(1)), d_((std::size_t)w_ * h_) {}
  void SetSize(int w, int h) { w_ = w; h_ = h; d_.assign((std::size_t)w * h, T()); }
  template <class D> void SetSize(const Eigen::MatrixBase<D>& size) { SetSize((int)size(0), (int)size(1)); }
  Eigen::Matrix<unsigned, 2, 1> size() const { return Eigen::Matrix<unsigned, 2, 1>(w_, h_); }
  template <class D> const T& operator()(const Eigen::MatrixBase<D>& p) const { return d_[p(0) + (std::size_t)p(1) * w_]; }
  template <class D> T& operator()(const Eigen::MatrixBase<D>& p) { return d_[p(0) + (std::size_t)p(1) * w_]; }
  template <class V> void SetTo(const V& v) { for (auto& e : d_) e = T(v); }
  bool Write(const std::string&) const { return true; }
  const T* data() const { return d_.data(); }
  T* data() { return d_.data(); }
  unsigned width() const { return w_; }
  unsigned height() const { return h_; }
  // libvis image.h:129-149 (InterpolateImageBilinear): integer part by truncation, the fractions held in FLOAT, four products summed in
  // the order top-left, top-right, bottom-left, bottom-right
  template <class R, class D> R InterpolateBilinear(const Eigen::MatrixBase<D>& position) const {
    const int ix = static_cast<int>(position(0)), iy = static_cast<int>(position(1));
    const float fx = position(0) - ix, fy = position(1) - iy;
    const float fx_inv = 1.f - fx, fy_inv = 1.f - fy;
    const T& tl = d_[ix + (std::size_t)iy * w_]; const T& tr = d_[ix + 1 + (std::size_t)iy * w_];
    const T& bl = d_[ix + (std::size_t)(iy + 1) * w_]; const T& br = d_[ix + 1 + (std::size_t)(iy + 1) * w_];
    return R(R(R((double)(fx_inv * fy_inv) * R(tl) + (double)(fx * fy_inv) * R(tr)) + (double)(fx_inv * fy) * R(bl)) + (double)(fx * fy) * R(br));
  }
  const T& at(int x, int y) const { return d_[x + (std::size_t)y * w_]; }
  T& at(int x, int y) { return d_[x + (std::size_t)y * w_]; }
  const T& operator()(int x, int y) const { return d_[x + (std::size_t)y * w_]; }
  T& operator()(int x, int y) { return d_[x + (std::size_t)y * w_]; }
 private:
  int w_, h_;
  std::vector<T> d_;
};
}
#endif
