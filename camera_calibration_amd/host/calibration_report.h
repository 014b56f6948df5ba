// Host mirror of the reference's report statistics -- same names and argument meaning as
// APP/calibration_report.cc:101-168 (APP = applications/camera_calibration/src/camera_calibration).
// The projections are batched through the C-ABI (cba_project, HIP); the reductions stay on the host.
#pragma once
#include <vector>
#include "dataset.h"

namespace vis {

// APP/calibration_report.cc:101-148.  Features whose projection fails are skipped (they contribute to
// neither the count nor the vectors), exactly as in the reference.
void ComputeAllReprojectionErrors(int camera_index, const Dataset& dataset, const BAState& calibration,
                                  usize* reprojection_error_count, double* reprojection_error_sum,
                                  double* reprojection_error_max, std::vector<Vec2d>* reprojection_errors,
                                  std::vector<Vec2f>* reprojection_features);

// APP/calibration_report.cc:151-168
void ComputeReprojectionErrorHistogram(int resolution, double extent_in_px, const std::vector<Vec2d>& reprojection_errors,
                                       Image<double>* hist_image);

// APP/calibration.cc:62-184.  The window / visualisation arguments of the reference are
// accepted and ignored (calibration_window must be null: there is no UI here).
class CalibrationWindow;
void DeleteOutlierFeatures(int camera_index, Dataset* dataset, BAState* state, float outlier_removal_factor,
                           CalibrationWindow* calibration_window = nullptr, bool step_by_step = false,
                           const char* outlier_visualization_path = nullptr);

}  // namespace vis
