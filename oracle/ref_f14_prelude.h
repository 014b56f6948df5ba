// oracle/_ref, part 5 (prelude) -- TEST INFRASTRUCTURE ONLY.
//
// What the reference's outer-loop and report functions need around them to compile OUTSIDE their files: the reference's real Dataset /
// BAState / CameraModel / joint_optimization.h headers (included from where they lie), and stand-ins for the Qt window, key input, file
// system and visualisation calls those functions make on the side.  The functions themselves are NOT in this repository: oracle/Makefile
// pipes their line ranges out of /root/reference into the compiler (see the rule of _ref/libcalibref_f14.so) behind this prelude.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <memory>
#include <ostream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>
#include "cba_quiet_log.h"

#include <libvis/eigen.h>
#include <libvis/image.h>
#include <libvis/libvis.h>
#include <libvis/logging.h>
#include <libvis/sophus.h>
#include <libvis/util.h>                                                   // the reference's own erase_if

#include "camera_calibration/dataset.h"                                    // APP/dataset.h:57-212
#include "camera_calibration/hash_vec2i.h"
#include "camera_calibration/models/camera_model.h"
#include "camera_calibration/bundle_adjustment/ba_state.h"                 // APP/bundle_adjustment/ba_state.h:46-97
#include "camera_calibration/bundle_adjustment/joint_optimization.h"       // SchurMode, OptimizeJointly (defined by ref_f14_glue.cc)
#ifdef CBA_REF_REAL_BA
#include <libvis/lm_optimizer.h>                                           // libcalibref_ba.so: the real OptimizationReport
#endif

namespace vis {
#ifndef CBA_REF_REAL_BA
// LV/lm_optimizer.h:55-77 (only final_cost is read by RunBundleAdjustment's CUDA branch, which the tests never take)
struct OptimizationReport { double initial_cost, final_cost; int num_iterations_performed; double cost_and_jacobian_evaluation_time, solve_time; };
#endif
inline OptimizationReport CudaOptimizeJointly(Dataset&, BAState*, int, int, double, double, double, double*) { std::abort(); }
inline bool SaveBAState(const char*, const BAState&) { return true; }
// the Qt window of the application: every call is a no-op here (the tests pass a null pointer anyway)
class CalibrationWindow {
 public:
  void UpdateRemovedOutliers(int, const Image<Vec3u8>&) {}
  void SetCurrentCameraIndex(int) {}
  void UpdateObservationDirections(int, const Image<Vec3u8>&) {}
  void UpdateErrorHistogram(int, const Image<u8>&) {}
  void UpdateReprojectionErrors(int, const Image<Vec3u8>&, Dataset*, BAState*) {}
  void UpdateErrorDirections(int, const Image<Vec3u8>&) {}
};
char GetKeyInput();              // APP/util.h:42, :45 -- defined by ref_f14_glue.cc: no key is ever pressed
int PollKeyInput();
inline void VisualizeModelDirections(const CameraModel&, Image<Vec3u8>*) {}
inline void CreateReprojectionErrorHistogram(int, const Dataset&, const BAState&, Image<u8>*) {}
inline void CreateReprojectionErrorMagnitudeVisualization(const Dataset&, int, const BAState&, float, Image<Vec3u8>*) {}
inline void CreateReprojectionErrorDirectionVisualization(const Dataset&, int, const BAState&, Image<Vec3u8>*) {}
}  // namespace vis
// QFileInfo(path).dir().mkpath(".") / QDir(path).mkpath("."): "create the containing folder" -- done with mkdir(2), component by component
#include <sys/stat.h>
struct QDir {
  std::string d;
  QDir() {}
  explicit QDir(const char* p) : d(p) {}
  bool mkpath(const char*) {
    for (size_t i = 1; i <= d.size(); ++i)
      if (i == d.size() || d[i] == '/') ::mkdir(d.substr(0, i).c_str(), 0777);
    return true;
  }
};
struct QFileInfo {
  std::string p;
  explicit QFileInfo(const char* path) : p(path) {}
  QDir dir() const { const size_t k = p.rfind('/'); return QDir(k == std::string::npos ? "." : p.substr(0, k).c_str()); }
};
