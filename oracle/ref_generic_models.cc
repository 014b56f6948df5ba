// oracle/_ref, part 1 -- TEST INFRASTRUCTURE ONLY.
//
// C entry points over the REFERENCE's own stand-alone generic camera models, compiled straight from
//   /root/reference/applications/camera_calibration/generic_models/src/{central_generic.h, noncentral_generic.h,
//   *_unprojection_jacobian.h, util.h}
// (Eigen-only code; built against the Eigen stand-in in oracle/ref_shim because Eigen is not installed here).
// Those headers implement the same projection LM / B-spline un-projection as the application's models
// (APP/models/central_generic.cc:433-549, noncentral_generic.cc:156-330), so they pin rows M1-M4 / N1-N3 of the
// oracle against code the reference authors wrote.  No reference source is copied: the headers are included from
// where they lie.  Output: oracle/_ref/libcalibref.so (git-ignored).
#include <cstdio>
#include <string>

#include "central_generic.h"
#include "noncentral_generic.h"

extern "C" {

// ---- central-generic ----
void* ref_central_create(int width, int height, int min_x, int min_y, int max_x, int max_y, int gw, int gh, const double* grid) {
  auto* cam = new CentralGenericCamera<double>(width, height, min_x, min_y, max_x, max_y, gw, gh);
  for (int y = 0; y < gh; ++y)
    for (int x = 0; x < gw; ++x) {
      const double* g = grid + 3 * (x + (size_t)y * gw);
      cam->grid_value(x, y) = Eigen::Vector3d(g[0], g[1], g[2]);
    }
  return cam;
}
// CentralGenericCamera::Read (the reference's own YAML reader); params: width height min_x min_y max_x max_y gw gh
void* ref_central_read(const char* yaml_path, int* params8) {
  auto* cam = new CentralGenericCamera<double>();
  std::string why;
  if (!cam->Read(yaml_path, &why)) { std::fprintf(stderr, "ref_central_read: %s\n", why.c_str()); delete cam; return nullptr; }
  params8[0] = cam->width(); params8[1] = cam->height();
  params8[2] = cam->calibration_min_x(); params8[3] = cam->calibration_min_y();
  params8[4] = cam->calibration_max_x(); params8[5] = cam->calibration_max_y();
  params8[6] = cam->grid_width(); params8[7] = cam->grid_height();
  return cam;
}
void ref_central_get_grid(void* h, double* grid) {
  auto* cam = static_cast<CentralGenericCamera<double>*>(h);
  for (int y = 0; y < cam->grid_height(); ++y)
    for (int x = 0; x < cam->grid_width(); ++x)
      for (int d = 0; d < 3; ++d) grid[3 * (x + (size_t)y * cam->grid_width()) + d] = cam->grid_value(x, y)(d);
}
void ref_central_destroy(void* h) { delete static_cast<CentralGenericCamera<double>*>(h); }
int ref_central_project(void* h, const double* p, double* px) {
  Eigen::Vector2d out;
  bool ok = static_cast<CentralGenericCamera<double>*>(h)->Project(Eigen::Vector3d(p[0], p[1], p[2]), &out);
  px[0] = out.x(); px[1] = out.y();
  return ok;
}
int ref_central_project_init(void* h, const double* p, double* px_inout) {
  Eigen::Vector2d out(px_inout[0], px_inout[1]);
  bool ok = static_cast<CentralGenericCamera<double>*>(h)->ProjectWithInitialEstimate(Eigen::Vector3d(p[0], p[1], p[2]), &out);
  px_inout[0] = out.x(); px_inout[1] = out.y();
  return ok;
}
int ref_central_project_jacobian(void* h, const double* p, double* px, double* jac6, double delta) {
  Eigen::Vector2d out;
  Eigen::Matrix<double, 2, 3> J;
  bool ok = static_cast<CentralGenericCamera<double>*>(h)->ProjectWithJacobian(Eigen::Vector3d(p[0], p[1], p[2]), &out, &J, delta);
  px[0] = out.x(); px[1] = out.y();
  for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) jac6[3 * r + c] = J(r, c);
  return ok;
}
int ref_central_unproject(void* h, const double* px, double* dir) {
  Eigen::Vector3d d;
  bool ok = static_cast<CentralGenericCamera<double>*>(h)->Unproject(Eigen::Vector2d(px[0], px[1]), &d);
  for (int i = 0; i < 3; ++i) dir[i] = d(i);
  return ok;
}
int ref_central_unproject_jacobian(void* h, const double* px, double* dir, double* jac6) {
  Eigen::Vector3d d;
  Eigen::Matrix<double, 3, 2> J;
  bool ok = static_cast<CentralGenericCamera<double>*>(h)->UnprojectWithJacobian(Eigen::Vector2d(px[0], px[1]), &d, &J);
  for (int i = 0; i < 3; ++i) { dir[i] = d(i); jac6[2 * i] = J(i, 0); jac6[2 * i + 1] = J(i, 1); }
  return ok;
}

// ---- non-central generic ----  grids: direction grid 3G, then point grid 3G (the layout of include/cba.h)
void* ref_noncentral_create(int width, int height, int min_x, int min_y, int max_x, int max_y, int gw, int gh, const double* grids) {
  auto* cam = new NoncentralGenericCamera<double>(width, height, min_x, min_y, max_x, max_y, gw, gh);
  const size_t G = (size_t)gw * gh;
  for (int y = 0; y < gh; ++y)
    for (int x = 0; x < gw; ++x) {
      const double* d = grids + 3 * (x + (size_t)y * gw);
      const double* o = grids + 3 * G + 3 * (x + (size_t)y * gw);
      cam->direction_grid_value(x, y) = Eigen::Vector3d(d[0], d[1], d[2]);
      cam->point_grid_value(x, y) = Eigen::Vector3d(o[0], o[1], o[2]);
    }
  return cam;
}
void ref_noncentral_destroy(void* h) { delete static_cast<NoncentralGenericCamera<double>*>(h); }
int ref_noncentral_project(void* h, const double* p, double* px) {
  Eigen::Vector2d out;
  bool ok = static_cast<NoncentralGenericCamera<double>*>(h)->Project(Eigen::Vector3d(p[0], p[1], p[2]), &out);
  px[0] = out.x(); px[1] = out.y();
  return ok;
}
int ref_noncentral_project_init(void* h, const double* p, double* px_inout) {
  Eigen::Vector2d out(px_inout[0], px_inout[1]);
  bool ok = static_cast<NoncentralGenericCamera<double>*>(h)->ProjectWithInitialEstimate(Eigen::Vector3d(p[0], p[1], p[2]), &out);
  px_inout[0] = out.x(); px_inout[1] = out.y();
  return ok;
}
int ref_noncentral_unproject(void* h, const double* px, double* line6) {   // direction, origin
  Eigen::ParametrizedLine<double, 3> l;
  bool ok = static_cast<NoncentralGenericCamera<double>*>(h)->Unproject(Eigen::Vector2d(px[0], px[1]), &l);
  for (int i = 0; i < 3; ++i) { line6[i] = l.direction()(i); line6[3 + i] = l.origin()(i); }
  return ok;
}
int ref_noncentral_unproject_jacobian(void* h, const double* px, double* line6, double* jac12) {
  Eigen::ParametrizedLine<double, 3> l;
  Eigen::Matrix<double, 6, 2> J;
  bool ok = static_cast<NoncentralGenericCamera<double>*>(h)->UnprojectWithJacobian(Eigen::Vector2d(px[0], px[1]), &l, &J);
  for (int i = 0; i < 3; ++i) { line6[i] = l.direction()(i); line6[3 + i] = l.origin()(i); }
  for (int i = 0; i < 6; ++i) { jac12[2 * i] = J(i, 0); jac12[2 * i + 1] = J(i, 1); }
  return ok;
}

}  // extern "C"
