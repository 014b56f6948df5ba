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

// Projects every feature of `camera_index` in the used imagesets (reference traversal order); magnitude < 0 = failed.
static bool ProjectAllFeatures(int camera_index, const Dataset& dataset, const BAState& state, std::vector<double>* magnitudes) {
  std::vector<double> local;
  std::vector<Vec2f> xy;
  for (int i = 0; i < dataset.ImagesetCount(); ++i) {
    if (!state.image_used.at(i)) continue;
    const SE3d image_tr_global = state.image_tr_global(camera_index, i);
    for (const PointFeature& f : dataset.GetImageset(i)->FeaturesOfCamera(camera_index)) {
      const Vec3d p = image_tr_global * state.points[f.index];
      local.push_back(p.x()); local.push_back(p.y()); local.push_back(p.z());
      xy.push_back(f.xy);
    }
  }
  const int64_t n = (int64_t)xy.size();
  magnitudes->assign((size_t)n, -1.0);
  if (n == 0) return true;
  const CameraModel* cam = state.intrinsics[camera_index].get();
  const cba_camera abi = cam->abi_camera();
  const std::vector<double> grid = cam->abi_grid();
  std::vector<double> pixels(2 * (size_t)n);
  std::vector<uint8_t> ok((size_t)n);
  if (cba_project(&abi, grid.data(), n, local.data(), nullptr, pixels.data(), ok.data(), 0) != CBA_OK) {
    std::fprintf(stderr, "DeleteOutlierFeatures: %s\n", cba_last_error());
    return false;
  }
  for (int64_t i = 0; i < n; ++i) {
    if (!ok[i]) continue;
    const double ex = pixels[2 * i] - (double)xy[i].x(), ey = pixels[2 * i + 1] - (double)xy[i].y();
    (*magnitudes)[i] = std::sqrt(ex * ex + ey * ey);
  }
  return true;
}

void DeleteOutlierFeatures(int camera_index, Dataset* dataset, BAState* state, float outlier_removal_factor,
                           CalibrationWindow* /*calibration_window*/, bool /*step_by_step*/,
                           const char* /*outlier_visualization_path*/) {
  std::vector<double> magnitudes;
  if (!ProjectAllFeatures(camera_index, *dataset, *state, &magnitudes)) return;
  std::vector<double> reprojection_errors;
  for (double m : magnitudes) if (m >= 0) reprojection_errors.push_back(m);
  if (reprojection_errors.size() < 8) return;   // arbitrary threshold (calibration.cc:97)
  std::sort(reprojection_errors.begin(), reprojection_errors.end());
  const double first_quartile_error = reprojection_errors[0.25f * reprojection_errors.size() + 0.5f];
  const double third_quartile_error = reprojection_errors[0.75f * reprojection_errors.size() + 0.5f];
  const double outlier_threshold = third_quartile_error + outlier_removal_factor * (third_quartile_error - first_quartile_error);
  size_t cursor = 0;
  for (int i = 0; i < dataset->ImagesetCount(); ++i) {
    if (!state->image_used.at(i)) continue;
    std::vector<PointFeature>& features = dataset->GetImageset(i)->FeaturesOfCamera(camera_index);
    std::vector<PointFeature> kept;
    for (const PointFeature& f : features) {
      const double m = magnitudes[cursor++];
      if (m < 0 || m > outlier_threshold) continue;   // does not project / above the threshold
      kept.push_back(f);
    }
    features.swap(kept);
    if (features.size() < 3) state->image_used.at(i) = false;
  }
}

}  // namespace vis
