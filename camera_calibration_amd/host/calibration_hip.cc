// Outer calibration loop over the HIP engine (see calibration.h).
#include "calibration.h"

#include <cmath>
#include <cstdio>
#include <limits>
#include <map>

#include "calibration_io.h"

namespace vis {
namespace {

Mat3d Mul(const Mat3d& a, const Mat3d& b) {
  Mat3d r;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { r.m[i][j] = 0; for (int k = 0; k < 3; ++k) r.m[i][j] += a.m[i][k] * b.m[k][j]; }
  return r;
}
Vec3d Apply(const Mat3d& a, const Vec3d& v) {
  return Vec3d(a.m[0][0] * v.x() + a.m[0][1] * v.y() + a.m[0][2] * v.z(), a.m[1][0] * v.x() + a.m[1][1] * v.y() + a.m[1][2] * v.z(),
               a.m[2][0] * v.x() + a.m[2][1] * v.y() + a.m[2][2] * v.z());
}
Mat3d QuatToMatrix(double w, double x, double y, double z) {
  return Mat3d{{{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}}};
}
}  // namespace

// Eigen matrix -> quaternion (Shepperd's branches)
Quaterniond MatrixToQuat(const Mat3d& r) {
  const double t = r.m[0][0] + r.m[1][1] + r.m[2][2];
  double q[4];
  if (t > 0) {
    const double s = std::sqrt(t + 1.0) * 2;
    q[0] = 0.25 * s; q[1] = (r.m[2][1] - r.m[1][2]) / s; q[2] = (r.m[0][2] - r.m[2][0]) / s; q[3] = (r.m[1][0] - r.m[0][1]) / s;
  } else {
    int i = 0;
    if (r.m[1][1] > r.m[0][0]) i = 1;
    if (r.m[2][2] > r.m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    const double s = std::sqrt(r.m[i][i] - r.m[j][j] - r.m[k][k] + 1.0) * 2;
    q[0] = (r.m[k][j] - r.m[j][k]) / s; q[1 + i] = 0.25 * s; q[1 + j] = (r.m[j][i] + r.m[i][j]) / s; q[1 + k] = (r.m[k][i] + r.m[i][k]) / s;
  }
  return Quaterniond(q[0], q[1], q[2], q[3]);
}

Mat3d ChooseNiceCameraOrientation(CameraModel* model) {
  Mat3d identity{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
  auto* cg = dynamic_cast<CentralGenericModel*>(model);
  if (!cg) return identity;
  const int w = model->width(), h = model->height();
  const int right_min_x = std::min<int>(w - 1, w / 2 + 11), right_max_x = w - 1;
  const int right_min_y = std::max<int>(0, h / 2 - 10), right_max_y = std::min<int>(h - 1, h / 2 + 10);
  std::vector<double> px;
  px.push_back(0.5f * w); px.push_back(0.5f * h);
  for (int y = right_min_y; y <= right_max_y; ++y)
    for (int x = right_min_x; x <= right_max_x; ++x) { px.push_back(x + 0.5f); px.push_back(y + 0.5f); }
  const int64_t n = (int64_t)px.size() / 2;
  std::vector<double> lines(6 * (size_t)n);
  std::vector<uint8_t> ok((size_t)n);
  const cba_camera abi = model->abi_camera();
  const std::vector<double> grid = model->abi_grid();
  if (cba_unproject(&abi, grid.data(), n, px.data(), lines.data(), nullptr, ok.data(), model->device) != CBA_OK) {
    std::fprintf(stderr, "ChooseNiceCameraOrientation: %s\n", cba_last_error());
    return identity;
  }
  Vec3d forward = ok[0] ? Vec3d(lines[0], lines[1], lines[2]) : Vec3d(0, 0, 1);
  // Quaterniond::FromTwoVectors(forward, (0, 0, 1)).toRotationMatrix()
  const double fn = forward.norm();
  const Vec3d v0(forward.x() / fn, forward.y() / fn, forward.z() / fn);
  const double c = v0.z();
  const double s = std::sqrt((1 + c) * 2);
  const Mat3d forward_rotation = QuatToMatrix(s * 0.5, v0.y() / s, -v0.x() / s, 0.0);   // axis = v0 x (0,0,1) = (v0.y, -v0.x, 0)
  Vec3d right_sum = Vec3d::Zero();
  u32 right_count = 0;
  for (int64_t i = 1; i < n; ++i) {
    if (!ok[i]) continue;
    right_sum = right_sum + Vec3d(lines[6 * i], lines[6 * i + 1], lines[6 * i + 2]);
    ++right_count;
  }
  Mat3d right_rotation = identity;
  if (right_count > 0) {
    const Vec3d fr = Apply(forward_rotation, Vec3d(right_sum.x() / right_count, right_sum.y() / right_count, right_sum.z() / right_count));
    const double angle = std::atan2(-fr.y(), fr.x());
    right_rotation = Mat3d{{{std::cos(angle), -std::sin(angle), 0}, {std::sin(angle), std::cos(angle), 0}, {0, 0, 1}}};
  }
  const Mat3d rotation = Mul(right_rotation, forward_rotation);
  Image<Vec3d>& g = cg->grid();                    // Rotate(rotation), central_grid.h:70-76
  for (size_t i = 0; i < (size_t)g.width() * g.height(); ++i) g.data()[i] = Apply(rotation, g.data()[i]);
  return rotation;
}

void RunBundleAdjustment(bool /*use_cuda*/, SchurMode schur_mode, int max_iteration_count, double cost_reduction_threshold,
                         Dataset* dataset, BAState* state, double regularization_weight, bool localize_only,
                         CalibrationWindow* /*calibration_window*/, bool /*step_by_step*/, const char* state_output_path) {
  const double numerical_diff_delta = 1e-4;        // numerical_diff_delta_range = {1e-4}, calibration.cc:201
  double lambda = -1;
  double last_cost = std::numeric_limits<double>::infinity();
  // ONE device-resident problem for the whole loop: the reference calls OptimizeJointly(max_iteration_count = 1) per
  // iteration (calibration.cc:227-237), which on this backend would re-create the device problem (3 GB of buffers at
  // BASELINE configs[1]) and re-upload the observations every time.  Per iteration only the state crosses the bus:
  // down for SaveBAState / the orientation beautification, up again because that beautification edits it.
  JointOptimizationSession session(*dataset, state, numerical_diff_delta, localize_only, /*eliminate_points*/ false, schur_mode);
  if (regularization_weight > 0)
    std::fprintf(stderr, "OptimizeJointly: Regularization is disabled at the moment since it is untested with the current version.\n");
  for (int iteration = 0; iteration < max_iteration_count; ++iteration) {
    const double cost = session.Optimize(/*max_iteration_count*/ 1, lambda, &lambda, nullptr, /*print_progress*/ false);
    session.ReadBackState();
    if (state_output_path) SaveBAState(state_output_path, *state);
    if (!localize_only) {                          // beautify all camera orientations (:248-254)
      for (int c = 0; c < state->num_cameras(); ++c) {
        const Mat3d rotation = ChooseNiceCameraOrientation(state->intrinsics[c].get());
        const SE3d rotation_transform(MatrixToQuat(rotation), Vec3d::Zero());
        state->camera_tr_rig[c] = rotation_transform * state->camera_tr_rig[c];
      }
      session.UploadState();
    }
    if (cost >= last_cost - cost_reduction_threshold) break;
    last_cost = cost;
  }
  session.ReadBackLastProjections();               // PointFeature::last_projection, mutated in place by the reference
}

void ScaleToMetric(Dataset* dataset, BAState* state) {
  double scaling_log_sum = 0;
  int scaling_count = 0;
  for (int k = 0; k < dataset->KnownGeometriesCount(); ++k) {
    const KnownGeometry& geometry = dataset->GetKnownGeometry(k);
    std::map<std::pair<int, int>, int> corner_position_to_index;
    for (const auto& item : geometry.feature_id_to_position) {
      auto it = state->feature_id_to_points_index.find(item.first);
      if (it != state->feature_id_to_points_index.end()) corner_position_to_index[{item.second.x(), item.second.y()}] = it->second;
    }
    if (corner_position_to_index.empty()) continue;
    for (const auto& item : geometry.feature_id_to_position) {
      auto it = corner_position_to_index.find({item.second.x(), item.second.y()});
      if (it == corner_position_to_index.end()) continue;
      const int index = it->second;
      const int kNeighbors[2][2] = {{1, 0}, {0, 1}};
      for (int n = 0; n < 2; ++n) {
        auto cit = corner_position_to_index.find({item.second.x() + kNeighbors[n][0], item.second.y() + kNeighbors[n][1]});
        if (cit == corner_position_to_index.end()) continue;
        const double ideal_distance = geometry.cell_length_in_meters;
        const double actual_distance = (state->points[index] - state->points[cit->second]).norm();
        scaling_log_sum += std::log(ideal_distance / actual_distance);
        ++scaling_count;
      }
    }
  }
  const double f = std::exp(scaling_log_sum / scaling_count);
  // BAState::ScaleState, ba_state.cc:60-76
  for (SE3d& p : state->camera_tr_rig) p.translation() = p.translation() * f;
  for (SE3d& p : state->rig_tr_global) p.translation() = p.translation() * f;
  for (Vec3d& p : state->points) p = p * f;
  for (auto& m : state->intrinsics)
    if (auto* nc = dynamic_cast<NoncentralGenericModel*>(m.get()))
      for (size_t i = 0; i < (size_t)nc->point_grid().width() * nc->point_grid().height(); ++i) nc->point_grid().data()[i] = nc->point_grid().data()[i] * f;
}

}  // namespace vis
