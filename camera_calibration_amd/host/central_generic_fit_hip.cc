// Grid-only fitting of the central-generic model on the HIP engine.
// Mirrors APP/models/central_generic.cc:267-431 (FitToDenseModel, FitToPixelDirections) and APP/calibration.cc:373-529
// (ResampleModel, central-generic source and target); the LM fit itself (FitToPixelDirectionsImpl, :551-568) is
// cba_fit_grid_to_directions.  No CPU fallback: a failed engine call leaves the grid as initialised and reports it.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>

#include "calibration_fit.h"

namespace vis {
namespace {

bool IsNan(const Vec3d& v) { return std::isnan(v.x()) || std::isnan(v.y()) || std::isnan(v.z()); }

bool RunFit(CentralGenericModel* model, const std::vector<Vec2d>& grid_points, const std::vector<Vec3d>& directions, int max_iteration_count) {
  const cba_camera abi = model->abi_camera();
  std::vector<double> grid = model->abi_grid();
  std::vector<double> gp(2 * grid_points.size()), dir(3 * directions.size());
  for (size_t i = 0; i < grid_points.size(); ++i) { gp[2 * i] = grid_points[i].x(); gp[2 * i + 1] = grid_points[i].y(); }
  for (size_t i = 0; i < directions.size(); ++i) { dir[3 * i] = directions[i].x(); dir[3 * i + 1] = directions[i].y(); dir[3 * i + 2] = directions[i].z(); }
  cba_fit_report rep;
  if (cba_fit_grid_to_directions(&abi, grid.data(), (int64_t)grid_points.size(), gp.data(), dir.data(), max_iteration_count, &rep, model->device) != CBA_OK) {
    std::fprintf(stderr, "CentralGenericModel fit: %s\n", cba_last_error());
    return false;
  }
  model->set_abi_grid(grid.data());
  return true;
}

}  // namespace

void CentralGenericModel::FitToPixelDirections(const std::vector<Vec2d>& pixels, const std::vector<Vec3d>& directions, int max_iteration_count) {
  std::vector<Vec2d> grid_points(pixels.size());
  for (usize i = 0; i < pixels.size(); ++i) grid_points[i] = PixelCornerConvToGridPoint(pixels[i].x(), pixels[i].y());
  RunFit(this, grid_points, directions, max_iteration_count);
}

bool CentralGenericModel::FitToDenseModel(const Image<Vec3d>& dense_model, int subsample_step, int max_iteration_count) {
  const int dw = (int)dense_model.width(), dh = (int)dense_model.height();
  auto dense = [&](int x, int y) -> const Vec3d& { return dense_model.data()[x + (size_t)y * dw]; };
  auto grid = [&](int x, int y) -> Vec3d& { return m_grid.data()[x + (size_t)y * m_grid.width()]; };
  const double scale_x = dw / static_cast<double>(m_width), scale_y = dh / static_cast<double>(m_height);
  const Vec3d nan3(std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN());
  bool have_nan = false;
  for (int gy = 0; gy < (int)m_grid.height(); ++gy)
    for (int gx = 0; gx < (int)m_grid.width(); ++gx) {
      const Vec2d p = GridPointToPixelCornerConv(gx, gy);
      const int cx = (int)(scale_x * p.x()), cy = (int)(scale_y * p.y());
      if (cx < 0 || cy < 0 || cx >= dw || cy >= dh) { grid(gx, gy) = nan3; have_nan = true; continue; }
      if (!std::isnan(dense(cx, cy).x())) { grid(gx, gy) = dense(cx, cy); continue; }
      bool found = false;
      for (int radius = 1; radius < 5 && !found; ++radius) {
        const int x0 = cx - radius, x1 = cx + radius, y0 = cy - radius, y1 = cy + radius;
        for (int x = std::max(0, x0); x <= std::min(dw - 1, x1); ++x) {           // top and bottom
          if (y0 >= 0 && !std::isnan(dense(x, y0).x())) { grid(gx, gy) = dense(x, y0); found = true; break; }
          if (y1 < dh && !std::isnan(dense(x, y1).x())) { grid(gx, gy) = dense(x, y1); found = true; break; }
        }
        if (found) break;
        for (int y = std::max(0, y0); y <= std::min(dh - 1, y1); ++y) {           // left and right
          if (x0 >= 0 && !std::isnan(dense(x0, y).x())) { grid(gx, gy) = dense(x0, y); found = true; break; }
          if (x1 < dw && !std::isnan(dense(x1, y).x())) { grid(gx, gy) = dense(x1, y); found = true; break; }
        }
      }
      if (!found) { grid(gx, gy) = nan3; have_nan = true; }
    }
  for (int iteration = 0; have_nan && iteration < dw + dh; ++iteration) {          // linear steps from the neighbours
    have_nan = false;
    for (int gy = 0; gy < (int)m_grid.height(); ++gy)
      for (int gx = 0; gx < (int)m_grid.width(); ++gx) {
        if (!IsNan(grid(gx, gy))) continue;
        Vec3d sum = Vec3d::Zero();
        int count = 0;
        const int directions[4][2] = {{0, 1}, {0, -1}, {1, 0}, {-1, 0}};
        for (int d = 0; d < 4; ++d) {
          const int nx1 = gx + directions[d][0], ny1 = gy + directions[d][1], nx2 = gx + 2 * directions[d][0], ny2 = gy + 2 * directions[d][1];
          if (nx2 < 0 || ny2 < 0 || nx2 >= (int)m_grid.width() || ny2 >= (int)m_grid.height()) continue;
          const Vec3d v1 = grid(nx1, ny1), v2 = grid(nx2, ny2);
          if (IsNan(v1) || IsNan(v2)) continue;
          sum = sum + v1 + (v1 - v2);
          ++count;
        }
        if (count > 0) { const double n = sum.norm(); grid(gx, gy) = Vec3d(sum.x() / n, sum.y() / n, sum.z() / n); }
        else have_nan = true;
      }
  }
  if (have_nan) return false;
  const double model_to_camera_x = static_cast<double>(m_width) / dw, model_to_camera_y = static_cast<double>(m_height) / dh;
  std::vector<Vec2d> grid_points;
  std::vector<Vec3d> directions;
  for (u32 y = m_calibration_min_y; y <= (u32)m_calibration_max_y; y += subsample_step)
    for (u32 x = m_calibration_min_x; x <= (u32)m_calibration_max_x; x += subsample_step) {
      const int mx = scale_x * x, my = scale_y * y;
      const Vec3d& measurement = dense(mx, my);
      if (IsNan(measurement)) continue;
      grid_points.push_back(PixelCornerConvToGridPoint(model_to_camera_x * (mx + 0.5f), model_to_camera_y * (my + 0.5f)));
      directions.push_back(measurement);
    }
  return RunFit(this, grid_points, directions, max_iteration_count);
}

bool ResampleModel(std::shared_ptr<CameraModel>& model_to_optimize, SE3d* /*camera_tr_rig*/, int calibration_min_x, int calibration_min_y,
                   int calibration_max_x, int calibration_max_y, CameraModel::Type model_type, int target_resolution_x, int target_resolution_y) {
  // Special case non-central -> non-central (calibration.cc:386-425): both grids re-gridded bilinearly (host only; the result
  // is an initial state for the next bundle adjustment, not a fit)
  if (model_to_optimize->type() == CameraModel::Type::NoncentralGeneric && model_type == CameraModel::Type::NoncentralGeneric) {
    NoncentralGenericModel* old = static_cast<NoncentralGenericModel*>(model_to_optimize.get());
    Image<Vec3d> new_point_grid(target_resolution_x, target_resolution_y), new_direction_grid(target_resolution_x, target_resolution_y);
    const int ow = old->point_grid().width(), oh = old->point_grid().height();
    // libvis Image::InterpolateBilinear for Vec3d pixels (LV/image.h:152-176): the fractions are floats
    auto bilinear = [](const Image<Vec3d>& img, double x, double y) {
      const int ix = (int)x, iy = (int)y;
      const float fx = (float)(x - ix), fy = (float)(y - iy), fxi = 1.f - fx, fyi = 1.f - fy;
      const Vec3d &a = img(ix, iy), &b = img(ix + 1, iy), &c = img(ix, iy + 1), &d = img(ix + 1, iy + 1);
      Vec3d r;
      for (int k = 0; k < 3; ++k) r.v[k] = (double)(fxi * fyi) * a.v[k] + (double)(fx * fyi) * b.v[k] + (double)(fxi * fy) * c.v[k] + (double)(fx * fy) * d.v[k];
      return r;
    };
    for (int y = 0; y < target_resolution_y; ++y)
      for (int x = 0; x < target_resolution_x; ++x) {
        // static GridPointToPixelCornerConv of the NEW grid (central_grid.h:132-140, evaluated in float)
        const double px = calibration_min_x + ((x - 1.f) / (target_resolution_x - 3.f)) * (calibration_max_x + 1 - calibration_min_x);
        const double py = calibration_min_y + ((y - 1.f) / (target_resolution_y - 3.f)) * (calibration_max_y + 1 - calibration_min_y);
        Vec2d og = old->PixelCornerConvToGridPoint(px, py);
        const double ogx = std::min(std::max(og.x(), 0.0), ow - 1.001), ogy = std::min(std::max(og.y(), 0.0), oh - 1.001);
        new_point_grid(x, y) = bilinear(old->point_grid(), ogx, ogy);
        new_direction_grid(x, y) = bilinear(old->direction_grid(), ogx, ogy);
      }
    auto* fresh = new NoncentralGenericModel(target_resolution_x, target_resolution_y, calibration_min_x, calibration_min_y,
                                             calibration_max_x, calibration_max_y, model_to_optimize->width(), model_to_optimize->height());
    fresh->SetPointGrid(new_point_grid);
    fresh->SetDirectionGrid(new_direction_grid);
    model_to_optimize.reset(fresh);
    return true;
  }
  if (model_to_optimize->type() != CameraModel::Type::CentralGeneric || model_type != CameraModel::Type::CentralGeneric) {
    std::fprintf(stderr, "ResampleModel: built for central-generic -> central-generic and non-central -> non-central (the reference has no other non-central source case either, calibration.cc:427-430)\n");
    return false;
  }
  const int w = model_to_optimize->width(), h = model_to_optimize->height();
  // dense direction model of the old camera: every pixel centre un-projected in one batch
  std::vector<double> pixels(2 * (size_t)w * h), lines(6 * (size_t)w * h);
  std::vector<uint8_t> ok((size_t)w * h);
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) { pixels[2 * ((size_t)y * w + x)] = x + 0.5; pixels[2 * ((size_t)y * w + x) + 1] = y + 0.5; }
  const cba_camera abi = model_to_optimize->abi_camera();
  const std::vector<double> grid = model_to_optimize->abi_grid();
  if (cba_unproject(&abi, grid.data(), (int64_t)w * h, pixels.data(), lines.data(), nullptr, ok.data(), model_to_optimize->device) != CBA_OK) {
    std::fprintf(stderr, "ResampleModel: %s\n", cba_last_error());
    return false;
  }
  Image<Vec3d> dense_model(w, h);
  const double nan = std::numeric_limits<double>::quiet_NaN();
  for (size_t i = 0; i < (size_t)w * h; ++i) dense_model.data()[i] = ok[i] ? Vec3d(lines[6 * i], lines[6 * i + 1], lines[6 * i + 2]) : Vec3d(nan, nan, nan);
  const int area_w = calibration_max_x - calibration_min_x + 1, area_h = calibration_max_y - calibration_min_y + 1;
  auto* fresh = new CentralGenericModel(target_resolution_x, target_resolution_y, calibration_min_x, calibration_min_y,
                                        calibration_max_x, calibration_max_y, w, h);
  const int subsample_step = std::max<int>(1, std::min(std::round(area_w / 300), std::round(area_h / 300)));
  if (!fresh->FitToDenseModel(dense_model, subsample_step, 3)) { delete fresh; return false; }
  model_to_optimize.reset(fresh);
  return true;
}

}  // namespace vis
