// Host mirror of the reference's on-disk formats: same names and argument meaning as
// APP/io/calibration_io.h / calibration_io.cc:51-247, 432-985 (APP = applications/camera_calibration/src/
// camera_calibration).  dataset.bin is byte-compatible; the YAML files are written in the reference's line
// format (std::setprecision(14)) and read with a small parser for exactly that subset (the reference links
// yaml-cpp, which this image does not have).  The reference's convenience .obj side files (camera centres, pattern points) are written
// next to the pose and point files as the reference does.
#pragma once
#include <memory>
#include <vector>
#include "dataset.h"

namespace vis {

bool SaveDataset(const char* path, const Dataset& dataset);          // calibration_io.cc:51-136
bool LoadDataset(const char* path, Dataset* dataset);                // :138-247

bool SaveBAState(const char* base_path, const BAState& state);       // :432-466
bool LoadBAState(const char* base_path, BAState* state, Dataset* dataset);   // :468-524

bool SaveCameraModel(const CameraModel& model, const char* path);    // :527-651 (generic models)
std::shared_ptr<CameraModel> LoadCameraModel(const char* path);      // :653-783

bool SavePoses(const std::vector<bool>& image_used, const std::vector<SE3d>& image_tr_pattern, const char* path);   // :785-839
bool LoadPoses(std::vector<bool>* image_used, std::vector<SE3d>* image_tr_pattern, const char* path);               // :841-888

bool SavePointsAndIndexMapping(const BAState& calibration, const char* path);   // :890-937
bool LoadPointsAndIndexMapping(std::vector<Vec3d>* optimized_geometry, std::unordered_map<int, int>* feature_id_to_points_index,
                               const char* path);                                // :939-985

}  // namespace vis
