// Report statistics over the HIP projection kernel (see calibration_report.h).
#include "calibration_report.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

#include "../../include/cba.h"

namespace vis {

void ComputeAllReprojectionErrors(int camera_index, const Dataset& dataset, const BAState& calibration,
                                  usize* reprojection_error_count, double* reprojection_error_sum,
                                  double* reprojection_error_max, std::vector<Vec2d>* reprojection_errors,
                                  std::vector<Vec2f>* reprojection_features) {
  *reprojection_error_count = 0;
  *reprojection_error_sum = 0.;
  *reprojection_error_max = 0;
  reprojection_errors->clear();
  reprojection_features->clear();

  // gather the local points of every feature in the reference's traversal order (imagesets, then features)
  std::vector<double> local;
  std::vector<Vec2f> xy;
  for (int imageset_index = 0; imageset_index < dataset.ImagesetCount(); ++imageset_index) {
    if (!calibration.image_used[imageset_index]) continue;
    const SE3d image_tr_global = calibration.image_tr_global(camera_index, imageset_index);
    for (const PointFeature& feature : dataset.GetImageset(imageset_index)->FeaturesOfCamera(camera_index)) {
      const Vec3d p = image_tr_global * calibration.points[feature.index];
      local.push_back(p.x()); local.push_back(p.y()); local.push_back(p.z());
      xy.push_back(feature.xy);
    }
  }
  const int64_t n = (int64_t)xy.size();
  if (n == 0) return;
  const CameraModel* cam = calibration.intrinsics[camera_index].get();
  const cba_camera abi = cam->abi_camera();
  const std::vector<double> grid = cam->abi_grid();
  std::vector<double> pixels(2 * (size_t)n);
  std::vector<uint8_t> ok((size_t)n);
  // CameraModel::Project: start from the centre of the calibrated area (init_pixels = NULL)
  if (cba_project(&abi, grid.data(), n, local.data(), nullptr, pixels.data(), ok.data(), 0) != CBA_OK) {
    std::fprintf(stderr, "ComputeAllReprojectionErrors: %s\n", cba_last_error());
    return;   // no CPU fallback
  }
  for (int64_t i = 0; i < n; ++i) {
    if (!ok[i]) continue;
    ++*reprojection_error_count;
    const Vec2d e(pixels[2 * i] - (double)xy[i].x(), pixels[2 * i + 1] - (double)xy[i].y());
    reprojection_errors->push_back(e);
    reprojection_features->push_back(xy[i]);
    const double m = std::sqrt(e.x() * e.x() + e.y() * e.y());
    *reprojection_error_sum += m;
    *reprojection_error_max = std::max(*reprojection_error_max, m);
  }
}

void ComputeReprojectionErrorHistogram(int resolution, double extent_in_px, const std::vector<Vec2d>& reprojection_errors,
                                       Image<double>* hist_image) {
  hist_image->SetSize(resolution, resolution);
  for (const Vec2d& e : reprojection_errors) {
    const double hx_f = resolution * 0.5f * ((e.x() / extent_in_px) + 1.f);
    const int hx = static_cast<int>(hx_f) - ((hx_f < 0) ? 1 : 0);
    const double hy_f = resolution * 0.5f * ((e.y() / extent_in_px) + 1.f);
    const int hy = static_cast<int>(hy_f) - ((hy_f < 0) ? 1 : 0);
    if (hx >= 0 && hy >= 0 && hx < resolution && hy < resolution) hist_image->data()[(size_t)hy * resolution + hx] += 1.0;
  }
}

}  // namespace vis
