// Host mirror of the grid-only fitting step: the CentralGenericModel members FitToDenseModel /
// FitToPixelDirections are declared in camera_model.h; this header adds ResampleModel (APP/calibration.cc:373-529,
// same signature; only the central-generic -> central-generic case is built, camera_tr_rig is untouched there).
#pragma once
#include <memory>
#include "camera_model.h"
#include "vis_types.h"

namespace vis {

bool ResampleModel(std::shared_ptr<CameraModel>& model_to_optimize, SE3d* camera_tr_rig, int calibration_min_x, int calibration_min_y,
                   int calibration_max_x, int calibration_max_y, CameraModel::Type model_type, int target_resolution_x, int target_resolution_y);

}  // namespace vis
