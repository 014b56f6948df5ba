// Host mirror of the outer calibration loop: same names and argument meaning as APP/calibration.cc
// (APP = applications/camera_calibration/src/camera_calibration): RunBundleAdjustment (:187-304), ScaleToMetric
// (:307-370), and CentralGenericModel::ChooseNiceCameraOrientation (APP/models/central_generic.cc:570-621) as a free
// function because the mirror's model class has no Mat3d type.  Window / step-by-step / state-output arguments of the
// reference are accepted where they exist and ignored (no UI here).
#pragma once
#include "dataset.h"
#include "joint_optimization.h"

namespace vis {

struct Mat3d { double m[3][3]; };   // row-major 3x3

// rotation matrix -> quaternion as SE3d(rotation, translation) does (Eigen's conversion)
Quaterniond MatrixToQuat(const Mat3d& r);

// Rotates the model's grid in place and returns the rotation (central-generic models only; identity otherwise).
Mat3d ChooseNiceCameraOrientation(CameraModel* model);

class CalibrationWindow;
void RunBundleAdjustment(bool use_cuda, SchurMode schur_mode, int max_iteration_count, double cost_reduction_threshold,
                         Dataset* dataset, BAState* state, double regularization_weight, bool localize_only,
                         CalibrationWindow* calibration_window = nullptr, bool step_by_step = false,
                         const char* state_output_path = nullptr);

void ScaleToMetric(Dataset* dataset, BAState* state);

}  // namespace vis
