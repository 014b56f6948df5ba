// Stand-in for <libvis/image.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref build): a row-major owning image with the
// (x, y) element access APP/b_spline.h uses (the reference class stores pixels contiguously with index x + y * width
// for unpadded images, LV/image.h).
#ifndef CBA_REF_SHIM_LIBVIS_IMAGE_
#define CBA_REF_SHIM_LIBVIS_IMAGE_
#include <vector>
#include "libvis/libvis.h"
namespace vis {
template <class T>
class Image {
 public:
  Image() : w_(0), h_(0) {}
  Image(int w, int h) : w_(w), h_(h), d_((std::size_t)w * h) {}
  Image(int w, int h, const T* data) : w_(w), h_(h), d_(data, data + (std::size_t)w * h) {}
  void SetSize(int w, int h) { w_ = w; h_ = h; d_.assign((std::size_t)w * h, T()); }
  const T* data() const { return d_.data(); }
  T* data() { return d_.data(); }
  const T& at(int x, int y) const { return d_[x + (std::size_t)y * w_]; }
  T& at(int x, int y) { return d_[x + (std::size_t)y * w_]; }
  int width() const { return w_; }
  int height() const { return h_; }
  const T& operator()(int x, int y) const { return d_[x + (std::size_t)y * w_]; }
  T& operator()(int x, int y) { return d_[x + (std::size_t)y * w_]; }
 private:
  int w_, h_;
  std::vector<T> d_;
};
}
#endif
