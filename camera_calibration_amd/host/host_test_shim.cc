// extern "C" shim used by the tests to drive the C++ host adapter from Python: builds Dataset / BAState /
// CameraModel objects from packed arrays, calls vis::OptimizeJointly with the reference's signature
// (as APP/test/util.h:452-469 does) and copies the result back.
#include "joint_optimization.h"
#include "calibration_report.h"
#include "calibration_io.h"
#include "calibration_fit.h"
#include "calibration.h"

#include <chrono>
#include <cmath>
#include <limits>

using namespace vis;

namespace {
// Dataset / BAState from packed arrays, and back
struct PackedProblem {
  Dataset dataset;
  BAState state;
  int n_cameras, n_imagesets_total, n_points;
  int64_t n_obs;
  const int32_t* imageset_index; const int32_t* camera_index;
  PackedProblem(int n_cameras_, const cba_camera* cams, const double* const* grids_in, int n_imagesets_total_, const uint8_t* image_used,
                const double* rig_tr_global, const double* camera_tr_rig, int n_points_, const double* points, int64_t n_obs_, const float* xy,
                const int32_t* point_index, const int32_t* imageset_index_, const int32_t* camera_index_)
      : dataset(n_cameras_), n_cameras(n_cameras_), n_imagesets_total(n_imagesets_total_), n_points(n_points_), n_obs(n_obs_),
        imageset_index(imageset_index_), camera_index(camera_index_) {
    for (int c = 0; c < n_cameras; ++c) {
      const cba_camera& k = cams[c];
      dataset.SetImageSize(c, Vec2i(k.width, k.height));
      std::shared_ptr<CameraModel> m;
      if (k.model_type == CBA_CENTRAL_GENERIC)
        m.reset(new CentralGenericModel(k.grid_w, k.grid_h, k.calib_min_x, k.calib_min_y, k.calib_max_x, k.calib_max_y, k.width, k.height));
      else
        m.reset(new NoncentralGenericModel(k.grid_w, k.grid_h, k.calib_min_x, k.calib_min_y, k.calib_max_x, k.calib_max_y, k.width, k.height));
      m->set_abi_grid(grids_in[c]);
      state.intrinsics.push_back(m);
      const double* p = camera_tr_rig + 7 * c;
      state.camera_tr_rig.push_back(SE3d(Quaterniond(p[0], p[1], p[2], p[3]), Vec3d(p[4], p[5], p[6])));
    }
    for (int i = 0; i < n_imagesets_total; ++i) {
      dataset.NewImageset();
      const double* p = rig_tr_global + 7 * (size_t)i;
      state.rig_tr_global.push_back(SE3d(Quaterniond(p[0], p[1], p[2], p[3]), Vec3d(p[4], p[5], p[6])));
      state.image_used.push_back(image_used[i] != 0);
    }
    for (int p = 0; p < n_points; ++p) {
      state.points.push_back(Vec3d(points[3 * p], points[3 * p + 1], points[3 * p + 2]));
      state.feature_id_to_points_index[1000 + p] = p;   // feature id -> point index
    }
    for (int64_t o = 0; o < n_obs; ++o) {
      auto& feats = dataset.GetImageset(imageset_index[o])->FeaturesOfCamera(camera_index[o]);
      feats.emplace_back(Vec2f(xy[2 * o], xy[2 * o + 1]), 1000 + point_index[o]);
    }
    state.ComputeFeatureIdToPointsIndex(&dataset);
  }
  void unpack(double* const* grids_out, double* rig_tr_global, double* camera_tr_rig, double* points, double* last_projection_out) {
    for (int c = 0; c < n_cameras; ++c) {
      std::vector<double> g = state.intrinsics[c]->abi_grid();
      for (size_t i = 0; i < g.size(); ++i) grids_out[c][i] = g[i];
      const Quaterniond& q = state.camera_tr_rig[c].unit_quaternion();
      double* p = camera_tr_rig + 7 * c;
      p[0] = q.w(); p[1] = q.x(); p[2] = q.y(); p[3] = q.z();
      for (int k = 0; k < 3; ++k) p[4 + k] = state.camera_tr_rig[c].translation().v[k];
    }
    for (int i = 0; i < n_imagesets_total; ++i) {
      const Quaterniond& q = state.rig_tr_global[i].unit_quaternion();
      double* p = rig_tr_global + 7 * (size_t)i;
      p[0] = q.w(); p[1] = q.x(); p[2] = q.y(); p[3] = q.z();
      for (int k = 0; k < 3; ++k) p[4 + k] = state.rig_tr_global[i].translation().v[k];
    }
    for (int p = 0; p < n_points; ++p) for (int k = 0; k < 3; ++k) points[3 * p + k] = state.points[p].v[k];
    if (last_projection_out) {
      // same traversal order as the input arrays were appended in (per imageset/camera feature vectors keep input order)
      std::vector<size_t> cursor((size_t)n_imagesets_total * n_cameras, 0);
      for (int64_t o = 0; o < n_obs; ++o) {
        size_t key = (size_t)imageset_index[o] * n_cameras + camera_index[o];
        const PointFeature& f = dataset.GetImageset(imageset_index[o])->FeaturesOfCamera(camera_index[o])[cursor[key]++];
        last_projection_out[2 * o] = f.last_projection.x(); last_projection_out[2 * o + 1] = f.last_projection.y();
      }
    }
  }
};
}  // namespace

extern "C" int cba_host_optimize_jointly(
    int n_cameras, const cba_camera* cams, const double* const* grids_in, double* const* grids_out,
    int n_imagesets_total, const uint8_t* image_used, double* rig_tr_global /*7 per imageset (all, used or not)*/,
    double* camera_tr_rig, int n_points, double* points,
    int64_t n_obs, const float* xy, const int32_t* point_index, const int32_t* imageset_index /*original*/, const int32_t* camera_index,
    int max_iteration_count, double init_lambda, double numerical_diff_delta, int localize_only, int eliminate_points,
    double* final_cost, double* final_lambda, int* performed_an_iteration, double* last_projection_out) {
  PackedProblem pp(n_cameras, cams, grids_in, n_imagesets_total, image_used, rig_tr_global, camera_tr_rig, n_points, points, n_obs, xy,
                   point_index, imageset_index, camera_index);
  bool performed = false;
  double lam = init_lambda;
  double cost = OptimizeJointly(pp.dataset, &pp.state, max_iteration_count, init_lambda, numerical_diff_delta, /*regularization_weight*/ 0,
                                localize_only != 0, eliminate_points != 0, SchurMode::Dense, &lam, &performed,
                                /*debug_verify_cost*/ true, false, false, false, false, /*print_progress*/ false);
  *final_cost = cost; *final_lambda = lam; *performed_an_iteration = performed ? 1 : 0;
  pp.unpack(grids_out, rig_tr_global, camera_tr_rig, points, last_projection_out);
  // a model-level call through the mirrored CameraModel API
  Vec2d px;
  Vec3d probe = pp.state.camera_tr_rig[0] * (pp.state.rig_tr_global[0] * pp.state.points[0]);
  (void)pp.state.intrinsics[0]->Project(probe, &px);
  return 0;
}

// vis::RunBundleAdjustment (host/calibration.h) from packed arrays.  mode 0: the shipped implementation (ONE device-resident
// JointOptimizationSession for the whole loop); mode 1: the reference's loop literally -- OptimizeJointly(max_iteration_count = 1)
// per iteration (APP/calibration.cc:227-237), i.e. cba_create + observation upload + cba_destroy every iteration -- to
// measure what the session saves.  seconds_out: wall time of the loop.
extern "C" int cba_host_run_bundle_adjustment(
    int n_cameras, const cba_camera* cams, const double* const* grids_in, double* const* grids_out,
    int n_imagesets_total, const uint8_t* image_used, double* rig_tr_global, double* camera_tr_rig, int n_points, double* points,
    int64_t n_obs, const float* xy, const int32_t* point_index, const int32_t* imageset_index, const int32_t* camera_index,
    int max_iteration_count, double cost_reduction_threshold, int mode, int* iterations_out, double* seconds_out, double* last_projection_out) {
  PackedProblem pp(n_cameras, cams, grids_in, n_imagesets_total, image_used, rig_tr_global, camera_tr_rig, n_points, points, n_obs, xy,
                   point_index, imageset_index, camera_index);
  const auto t0 = std::chrono::steady_clock::now();
  int iterations = 0;
  if (mode == 0) {
    RunBundleAdjustment(false, SchurMode::Dense, max_iteration_count, cost_reduction_threshold, &pp.dataset, &pp.state, 0.0, false);
    iterations = -1;
  } else {
    double lambda = -1, last_cost = std::numeric_limits<double>::infinity();
    for (int iteration = 0; iteration < max_iteration_count; ++iteration) {
      const double cost = OptimizeJointly(pp.dataset, &pp.state, 1, lambda, 1e-4, 0.0, false, false, SchurMode::Dense, &lambda, nullptr, false,
                                          false, false, false, false, false);
      ++iterations;
      for (int c = 0; c < pp.state.num_cameras(); ++c) {
        const Mat3d rotation = ChooseNiceCameraOrientation(pp.state.intrinsics[c].get());
        pp.state.camera_tr_rig[c] = SE3d(MatrixToQuat(rotation), Vec3d::Zero()) * pp.state.camera_tr_rig[c];
      }
      if (cost >= last_cost - cost_reduction_threshold) break;
      last_cost = cost;
    }
  }
  *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  *iterations_out = iterations;
  pp.unpack(grids_out, rig_tr_global, camera_tr_rig, points, last_projection_out);
  return 0;
}


// Drives vis::ComputeAllReprojectionErrors / ComputeReprojectionErrorHistogram (calibration_report.h) from packed
// arrays.  errors_out: 2 per observation of `camera` (capacity n_obs), returns the count through *count.
extern "C" int cba_host_reprojection_report(
    int n_cameras, const cba_camera* cams, const double* const* grids, int camera,
    int n_imagesets, const double* rig_tr_global, const double* camera_tr_rig, int n_points, const double* points,
    int64_t n_obs, const float* xy, const int32_t* point_index, const int32_t* imageset_index, const int32_t* camera_index,
    int64_t* count, double* sum, double* max, double* errors_out, float* features_out,
    int hist_resolution, double hist_extent, double* hist_out,
    float outlier_removal_factor, uint8_t* keep_out /*n_obs or NULL*/, uint8_t* used_out /*n_imagesets*/) {
  Dataset dataset(n_cameras);
  BAState state;
  for (int c = 0; c < n_cameras; ++c) {
    const cba_camera& k = cams[c];
    std::shared_ptr<CameraModel> m;
    if (k.model_type == CBA_CENTRAL_GENERIC)
      m.reset(new CentralGenericModel(k.grid_w, k.grid_h, k.calib_min_x, k.calib_min_y, k.calib_max_x, k.calib_max_y, k.width, k.height));
    else
      m.reset(new NoncentralGenericModel(k.grid_w, k.grid_h, k.calib_min_x, k.calib_min_y, k.calib_max_x, k.calib_max_y, k.width, k.height));
    m->set_abi_grid(grids[c]);
    state.intrinsics.push_back(m);
    const double* p = camera_tr_rig + 7 * c;
    state.camera_tr_rig.push_back(SE3d(Quaterniond(p[0], p[1], p[2], p[3]), Vec3d(p[4], p[5], p[6])));
  }
  for (int i = 0; i < n_imagesets; ++i) {
    dataset.NewImageset();
    const double* p = rig_tr_global + 7 * (size_t)i;
    state.rig_tr_global.push_back(SE3d(Quaterniond(p[0], p[1], p[2], p[3]), Vec3d(p[4], p[5], p[6])));
    state.image_used.push_back(true);
  }
  for (int p = 0; p < n_points; ++p) state.points.push_back(Vec3d(points[3 * p], points[3 * p + 1], points[3 * p + 2]));
  for (int64_t o = 0; o < n_obs; ++o) {
    auto& feats = dataset.GetImageset(imageset_index[o])->FeaturesOfCamera(camera_index[o]);
    feats.emplace_back(Vec2f(xy[2 * o], xy[2 * o + 1]), point_index[o]);
    feats.back().index = point_index[o];
  }
  usize n = 0; double s = 0, mx = 0;
  std::vector<Vec2d> errors; std::vector<Vec2f> feats;
  ComputeAllReprojectionErrors(camera, dataset, state, &n, &s, &mx, &errors, &feats);
  *count = (int64_t)n; *sum = s; *max = mx;
  for (size_t i = 0; i < errors.size(); ++i) {
    errors_out[2 * i] = errors[i].x(); errors_out[2 * i + 1] = errors[i].y();
    features_out[2 * i] = feats[i].x(); features_out[2 * i + 1] = feats[i].y();
  }
  Image<double> hist;
  ComputeReprojectionErrorHistogram(hist_resolution, hist_extent, errors, &hist);
  for (int i = 0; i < hist_resolution * hist_resolution; ++i) hist_out[i] = hist.data()[i];
  // DeleteOutlierFeatures on the same objects: report which observations survive and which imagesets stay used
  if (keep_out) {
    for (int64_t o = 0; o < n_obs; ++o) keep_out[o] = 0;
    // tag every feature with its observation index through the (otherwise unused) id field
    std::vector<size_t> cursor((size_t)n_imagesets * n_cameras, 0);
    for (int64_t o = 0; o < n_obs; ++o) {
      size_t key = (size_t)imageset_index[o] * n_cameras + camera_index[o];
      dataset.GetImageset(imageset_index[o])->FeaturesOfCamera(camera_index[o])[cursor[key]++].id = (int)o;
    }
    DeleteOutlierFeatures(camera, &dataset, &state, outlier_removal_factor);
    for (int i = 0; i < n_imagesets; ++i) {
      used_out[i] = state.image_used[i] ? 1 : 0;
      for (int c = 0; c < n_cameras; ++c)
        for (const PointFeature& f : dataset.GetImageset(i)->FeaturesOfCamera(c)) keep_out[f.id] = 1;
    }
  }
  return 0;
}


// F2 round trip through the C++ mirror: load dataset.bin + BAState directory, write both back elsewhere.
extern "C" int cba_host_io_roundtrip(const char* dataset_in, const char* state_in, const char* dataset_out, const char* state_out,
                                     int* n_imagesets, int* n_cameras, int64_t* n_features, int* n_points, int* n_unindexed) {
  Dataset dataset;
  if (!LoadDataset(dataset_in, &dataset)) return -1;
  BAState state;
  if (!LoadBAState(state_in, &state, &dataset)) return -2;
  *n_imagesets = dataset.ImagesetCount(); *n_cameras = dataset.num_cameras(); *n_points = (int)state.points.size();
  int64_t nf = 0; int bad = 0;
  for (int i = 0; i < dataset.ImagesetCount(); ++i)
    for (int c = 0; c < dataset.num_cameras(); ++c)
      for (const PointFeature& f : dataset.GetImageset(i)->FeaturesOfCamera(c)) { ++nf; if (f.index < 0) ++bad; }
  *n_features = nf; *n_unindexed = bad;
  if (!SaveDataset(dataset_out, dataset)) return -3;
  if (!SaveBAState(state_out, state)) return -4;
  return 0;
}


// F3 through the C++ mirror: FitToDenseModel on a dense direction image, then ResampleModel to a finer grid.
extern "C" int cba_host_fit_and_resample(const cba_camera* cam, int dense_w, int dense_h, const double* dense /*3 per pixel, NaN = invalid*/,
                                         int subsample_step, int max_iteration_count, double* grid_out /*3G*/,
                                         int target_w, int target_h, double* resampled_out /*3 * target_w * target_h*/) {
  CentralGenericModel model(cam->grid_w, cam->grid_h, cam->calib_min_x, cam->calib_min_y, cam->calib_max_x, cam->calib_max_y, cam->width, cam->height);
  Image<Vec3d> dm(dense_w, dense_h);
  for (size_t i = 0; i < (size_t)dense_w * dense_h; ++i) dm.data()[i] = Vec3d(dense[3 * i], dense[3 * i + 1], dense[3 * i + 2]);
  if (!model.FitToDenseModel(dm, subsample_step, max_iteration_count)) return -1;
  std::vector<double> g = model.abi_grid();
  for (size_t i = 0; i < g.size(); ++i) grid_out[i] = g[i];
  if (target_w > 0) {
    std::shared_ptr<CameraModel> m(new CentralGenericModel(model));
    SE3d dummy;
    if (!ResampleModel(m, &dummy, cam->calib_min_x, cam->calib_min_y, cam->calib_max_x, cam->calib_max_y, CameraModel::Type::CentralGeneric, target_w, target_h)) return -2;
    std::vector<double> g2 = m->abi_grid();
    for (size_t i = 0; i < g2.size(); ++i) resampled_out[i] = g2[i];
  }
  return 0;
}


// F3, non-central helpers: InitializeFromCentralGenericModel (+ Scale of the point grid) and ResampleModel non-central ->
// non-central.  grids_in / out: direction grid then point grid, 3 doubles per control point (the C-ABI layout).
extern "C" int cba_host_noncentral_init_and_resample(const cba_camera* central_cam, const double* central_grid /*3G*/, const double* point_grid /*3G or null*/,
                                                     double scale, double* init_out /*6G*/, int target_w, int target_h, double* resampled_out /*6 tw th*/) {
  CentralGenericModel central(central_cam->grid_w, central_cam->grid_h, central_cam->calib_min_x, central_cam->calib_min_y, central_cam->calib_max_x,
                              central_cam->calib_max_y, central_cam->width, central_cam->height);
  central.set_abi_grid(central_grid);
  NoncentralGenericModel* nc = new NoncentralGenericModel(4, 4, 0, 0, 1, 1, 2, 2);       // everything is overwritten by the initialisation
  std::shared_ptr<CameraModel> m(nc);
  nc->InitializeFromCentralGenericModel(central);
  const size_t G = (size_t)central_cam->grid_w * central_cam->grid_h;
  if (nc->width() != central_cam->width || nc->calibration_max_x() != central_cam->calib_max_x || nc->point_grid().width() != (u32)central_cam->grid_w) return -1;
  if (point_grid) {
    for (size_t i = 0; i < G; ++i) nc->point_grid().data()[i] = Vec3d(point_grid[3 * i], point_grid[3 * i + 1], point_grid[3 * i + 2]);
    nc->Scale(scale);
  }
  std::vector<double> g = nc->abi_grid();
  for (size_t i = 0; i < g.size(); ++i) init_out[i] = g[i];
  SE3d dummy;
  if (!ResampleModel(m, &dummy, central_cam->calib_min_x, central_cam->calib_min_y, central_cam->calib_max_x, central_cam->calib_max_y,
                     CameraModel::Type::NoncentralGeneric, target_w, target_h)) return -2;
  std::vector<double> g2 = m->abi_grid();
  for (size_t i = 0; i < g2.size(); ++i) resampled_out[i] = g2[i];
  return 0;
}

// F1 through the C++ mirror: ChooseNiceCameraOrientation on a central-generic grid (rotation 9 doubles row-major,
// grid rotated in place) and ScaleToMetric on a one-geometry lattice (returns the scaled points).
extern "C" int cba_host_nice_orientation(const cba_camera* cam, double* grid /*3G in/out*/, double* rotation9) {
  CentralGenericModel model(cam->grid_w, cam->grid_h, cam->calib_min_x, cam->calib_min_y, cam->calib_max_x, cam->calib_max_y, cam->width, cam->height);
  model.set_abi_grid(grid);
  const Mat3d r = ChooseNiceCameraOrientation(&model);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) rotation9[3 * i + j] = r.m[i][j];
  std::vector<double> g = model.abi_grid();
  for (size_t i = 0; i < g.size(); ++i) grid[i] = g[i];
  return 0;
}
extern "C" int cba_host_scale_to_metric(float cell_length, int n, const int32_t* feature_id, const int32_t* pos_xy, double* points /*3n in/out*/,
                                        double* pose7 /*one rig pose, in/out*/) {
  Dataset dataset(1);
  dataset.SetKnownGeometriesCount(1);
  KnownGeometry& g = dataset.GetKnownGeometry(0);
  g.cell_length_in_meters = cell_length;
  BAState state;
  for (int i = 0; i < n; ++i) {
    g.feature_id_to_position[feature_id[i]] = Vec2i(pos_xy[2 * i], pos_xy[2 * i + 1]);
    state.feature_id_to_points_index[feature_id[i]] = i;
    state.points.push_back(Vec3d(points[3 * i], points[3 * i + 1], points[3 * i + 2]));
  }
  state.rig_tr_global.push_back(SE3d(Quaterniond(pose7[0], pose7[1], pose7[2], pose7[3]), Vec3d(pose7[4], pose7[5], pose7[6])));
  ScaleToMetric(&dataset, &state);
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) points[3 * i + k] = state.points[i].v[k];
  for (int k = 0; k < 3; ++k) pose7[4 + k] = state.rig_tr_global[0].translation().v[k];
  return 0;
}
