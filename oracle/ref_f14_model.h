// oracle/_ref, part 5 (camera model) -- TEST INFRASTRUCTURE ONLY.
// A concrete CameraModel for the outer-loop / report functions: the reference's CentralGridModel (APP/models/central_grid.h: Project,
// Unproject, Rotate, the grid itself); the iterative projection it delegates to its derived class (CentralGenericModel::
// ProjectDirectionWithInitialEstimate, APP/models/central_generic.cc:137-224, which needs the rest of that file) is the oracle's
// orc_project_direction_with_initial_estimate on the LIVE grid -- pinned separately against the reference authors' stand-alone implementation in
// tests/test_oracle_vs_ref.py.  ChooseNiceCameraOrientation is declared here and DEFINED by the reference's own text: oracle/Makefile
// pipes APP/models/central_generic.cc:570-621 into the compiler with `CentralGenericModel` renamed to this class.
#pragma once
#include "camera_calibration/models/central_grid.h"
#include <vector>
#include "cba_oracle.h"
namespace vis {
class RefOrientedModel : public CentralGridModel<RefOrientedModel> {
 public:
  RefOrientedModel(int gw, int gh, int min_x, int min_y, int max_x, int max_y, int width, int height)
      : CentralGridModel<RefOrientedModel>(CameraModel::Type::CentralGeneric, gw, gh, min_x, min_y, max_x, max_y, width, height) {}
  CameraModel* duplicate() override { return new RefOrientedModel(*this); }
  bool ProjectDirectionWithInitialEstimate(const Vec3d& local_direction, Vec2d* result) const {
    const int gw = grid().width(), gh = grid().height();
    orc_camera cam{ORC_CENTRAL_GENERIC, width(), height(), calibration_min_x(), calibration_min_y(), calibration_max_x(), calibration_max_y(), gw, gh};
    std::vector<double> g(3 * (size_t)gw * gh);
    for (int y = 0; y < gh; ++y)
      for (int x = 0; x < gw; ++x)
        for (int k = 0; k < 3; ++k) g[3 * (x + (size_t)y * gw) + k] = grid()(x, y)(k);
    const double dir[3] = {local_direction.x(), local_direction.y(), local_direction.z()};
    double px[2] = {result->x(), result->y()};
    const bool ok = orc_project_direction_with_initial_estimate(&cam, g.data(), dir, px) != 0;
    if (ok) *result = Vec2d(px[0], px[1]);
    return ok;
  }
  using CentralGridModel<RefOrientedModel>::Unproject;
  bool Unproject(double x, double y, Vec3d* result) const override {
    if (!IsInCalibratedArea(x, y)) return false;
    const Vec2d gp = PixelCornerConvToGridPoint(x, y);
    *result = UnprojectFromGrid(gp.x(), gp.y());
    return true;
  }
  Mat3d ChooseNiceCameraOrientation() override;      // body: APP/models/central_generic.cc:570-621, piped in by oracle/Makefile
};
}  // namespace vis
