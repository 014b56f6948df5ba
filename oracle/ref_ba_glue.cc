// oracle/_ref, part 6 (glue of libcalibref_ba.so) -- TEST INFRASTRUCTURE ONLY.
//
// The reference's ENTIRE CPU bundle-adjustment path, compiled whole from where it lies:
//   APP/bundle_adjustment/joint_optimization.cc   JointOptimizationState, JointOptimizationCostFunction (the per-observation driver
//                                                 :240-593 that rounds 1-4 could only restate), OptimizeJointly :757-953
//   APP/models/central_generic.cc (+ the generated central_generic_jacobians.cc), APP/models/central_grid.h
//                                                 projection LM, finite-difference Jacobians w.r.t. the grid, FitToPixelDirections
//   APP/models/noncentral_generic.cc (+ generated Jacobians)
//   LV/lm_optimizer.h, lm_optimizer_update_accumulator.h, APP/dataset.cc, APP/bundle_adjustment/ba_state.cc
// against the stand-ins of oracle/ref_shim_lm (Eigen incl. Geometry, Sophus::SE3, Image, no-op CUDA / display headers) and a stand-in
// all_models.h that dispatches over the two generic models only.  Eigen's LDLT is the one numerical piece that is not the reference's or
// Eigen's code: the stand-in's LDLT::solve hands the system to the oracle's term-by-term restatement (orc_ldlt_solve_upper_unblocked).
// Nothing of the reference is copied into this repository.
//
// This file is appended to ref_f14_glue.cc's translation unit (CBA_REF_REAL_BA defined): the F1 / F4 entry points of that file run here
// on the reference's real CentralGenericModel and around the reference's real OptimizeJointly; ref_ba_optimize_jointly below is the
// hot path itself on packed arrays.
namespace vis {
int g_ref_ba_optimize_calls = 0;
double g_ref_ba_fd_delta_seen = 0;
}

namespace {
// cam9 per camera: model_type (0 central, 1 non-central), width height min_x min_y max_x max_y gw gh
std::shared_ptr<CameraModel> make_any_model(const int* p9, const double* grid) {
  const int gw = p9[7], gh = p9[8];
  if (p9[0] == 0) return make_model(p9 + 1, grid);
  auto m = std::make_shared<NoncentralGenericModel>(gw, gh, p9[3], p9[4], p9[5], p9[6], p9[1], p9[2]);
  Image<Vec3d> dir(gw, gh), pt(gw, gh);
  const size_t G = (size_t)gw * gh;
  for (int y = 0; y < gh; ++y)
    for (int x = 0; x < gw; ++x) {
      const double* d = grid + 3 * (x + (size_t)y * gw);
      dir(x, y) = Vec3d(d[0], d[1], d[2]);
      pt(x, y) = Vec3d(d[3 * G], d[3 * G + 1], d[3 * G + 2]);
    }
  m->SetDirectionGrid(dir);
  m->SetPointGrid(pt);
  return m;
}
void store_any_grid(const CameraModel* cm, double* grid) {
  if (cm->type() == CameraModel::Type::CentralGeneric) { store_grid(cm, grid); return; }
  const NoncentralGenericModel* m = static_cast<const NoncentralGenericModel*>(cm);
  const int gw = m->direction_grid().width(), gh = m->direction_grid().height();
  const size_t G = (size_t)gw * gh;
  for (int y = 0; y < gh; ++y)
    for (int x = 0; x < gw; ++x)
      for (int k = 0; k < 3; ++k) {
        grid[3 * (x + (size_t)y * gw) + k] = m->direction_grid()(x, y)(k);
        grid[3 * G + 3 * (x + (size_t)y * gw) + k] = m->point_grid()(x, y)(k);
      }
}
}  // namespace

namespace {
struct Marshalled {
  Dataset ds; BAState st; std::vector<PointFeature*> feats;
};
void marshal(int n_cameras, int n_images, int n_points, const int* cam9, int64_t n_obs, const float* obs_xy, const int* obs_point,
             const int* obs_image, const int* obs_camera, const double* rig_tr_global, const double* camera_tr_rig, const double* points,
             double* const* grids, const double* last_projection, Marshalled* m) {
  Dataset& ds = m->ds; BAState& st = m->st;
  ds.Reset(n_cameras);
  for (int c = 0; c < n_cameras; ++c) ds.SetImageSize(c, Vec2i(cam9[9 * c + 1], cam9[9 * c + 2]));
  for (int i = 0; i < n_images; ++i) ds.NewImageset();
  for (int64_t o = 0; o < n_obs; ++o) {
    PointFeature f(Vec2f(obs_xy[2 * o], obs_xy[2 * o + 1]), obs_point[o]);
    f.index = obs_point[o];
    f.last_projection = Vec2d(last_projection[2 * o], last_projection[2 * o + 1]);
    ds.GetImageset(obs_image[o])->FeaturesOfCamera(obs_camera[o]).push_back(f);
  }
  // pointers after all push_backs: the packed order is image-major, camera, feature order
  std::vector<size_t> cursor((size_t)n_images * n_cameras, 0);
  for (int64_t o = 0; o < n_obs; ++o) {
    auto& v = ds.GetImageset(obs_image[o])->FeaturesOfCamera(obs_camera[o]);
    m->feats.push_back(&v[cursor[(size_t)obs_image[o] * n_cameras + obs_camera[o]]++]);
  }
  st.image_used.assign(n_images, true);
  for (int c = 0; c < n_cameras; ++c) st.camera_tr_rig.push_back(pose_of(camera_tr_rig + 7 * c));
  for (int i = 0; i < n_images; ++i) st.rig_tr_global.push_back(pose_of(rig_tr_global + 7 * i));
  for (int p = 0; p < n_points; ++p) { st.points.push_back(Vec3d(points[3 * p], points[3 * p + 1], points[3 * p + 2])); st.feature_id_to_points_index[p] = p; }
  for (int c = 0; c < n_cameras; ++c) st.intrinsics.push_back(make_any_model(cam9 + 9 * c, grids[c]));
}
}  // namespace

CBA_EXPORT double ref_ba_optimize_jointly_mode(int, int, int, const int*, int64_t, const float*, const int*, const int*, const int*, double*, double*, double*,
                                               double* const*, double*, int, double, double, int, int, double*, int*, int);

// vis::OptimizeJointly (APP/bundle_adjustment/joint_optimization.cc:757-953), SchurMode::Dense, on a packed problem.  State and the
// warm-start cache (PointFeature::last_projection, packed observation order) in / out.  Returns the final cost.
CBA_EXPORT double ref_ba_optimize_jointly(int n_cameras, int n_images, int n_points, const int* cam9, int64_t n_obs, const float* obs_xy,
                                          const int* obs_point, const int* obs_image, const int* obs_camera, double* rig_tr_global,
                                          double* camera_tr_rig, double* points, double* const* grids, double* last_projection,
                                          int max_iteration_count, double init_lambda, double numerical_diff_delta, int localize_only,
                                          int eliminate_points, double* final_lambda, int* performed_an_iteration) {
  return ref_ba_optimize_jointly_mode(n_cameras, n_images, n_points, cam9, n_obs, obs_xy, obs_point, obs_image, obs_camera, rig_tr_global, camera_tr_rig,
                                      points, grids, last_projection, max_iteration_count, init_lambda, numerical_diff_delta, localize_only,
                                      eliminate_points, final_lambda, performed_an_iteration, (int)SchurMode::Dense);
}
// The same with the SchurMode chosen by the caller: 0 = Dense (the CPU path); in the library built from the PATCHED tree (`make patched`:
// integration/reference.patch) 5 = SchurMode::HIP, the reference's own OptimizeJointly handing the LM iteration to the MI355X through the
// adapter the patch adds and include/cba.h.
CBA_EXPORT double ref_ba_optimize_jointly_mode(int n_cameras, int n_images, int n_points, const int* cam9, int64_t n_obs, const float* obs_xy,
                                               const int* obs_point, const int* obs_image, const int* obs_camera, double* rig_tr_global,
                                               double* camera_tr_rig, double* points, double* const* grids, double* last_projection,
                                               int max_iteration_count, double init_lambda, double numerical_diff_delta, int localize_only,
                                               int eliminate_points, double* final_lambda, int* performed_an_iteration, int schur_mode) {
  Marshalled m;
  marshal(n_cameras, n_images, n_points, cam9, n_obs, obs_xy, obs_point, obs_image, obs_camera, rig_tr_global, camera_tr_rig, points, grids,
          last_projection, &m);
  BAState& st = m.st;
  double lam = 0; bool performed = false;
  const double cost = OptimizeJointly(m.ds, &st, max_iteration_count, init_lambda, numerical_diff_delta, /*regularization_weight*/ 0.0,
                                      localize_only != 0, eliminate_points != 0, static_cast<SchurMode>(schur_mode), &lam, &performed, false, false, false, false,
                                      false, /*print_progress*/ false);
  if (final_lambda) *final_lambda = lam;
  if (performed_an_iteration) *performed_an_iteration = performed ? 1 : 0;
  for (int i = 0; i < n_images; ++i) store_pose(st.rig_tr_global[i], rig_tr_global + 7 * i);
  for (int c = 0; c < n_cameras; ++c) store_pose(st.camera_tr_rig[c], camera_tr_rig + 7 * c);
  for (int p = 0; p < n_points; ++p) for (int k = 0; k < 3; ++k) points[3 * p + k] = st.points[p](k);
  for (int c = 0; c < n_cameras; ++c) store_any_grid(st.intrinsics[c].get(), grids[c]);
  for (int64_t o = 0; o < n_obs; ++o) { last_projection[2 * o] = m.feats[o]->last_projection.x(); last_projection[2 * o + 1] = m.feats[o]->last_projection.y(); }
  return cost;
}

// One JointOptimizationCostFunction::Compute<true> (joint_optimization.cc:240-593: THE per-observation driver) into the reference's own
// UpdateEquationAccumulator, set up as LMOptimizer::OptimizeImpl sets it up for SchurMode::Dense (LV/lm_optimizer.h:668-720, lambda 0):
// the normal equations of one Jacobian pass in the layout of orc_system (upper triangles, row-major), the per-residual cost vector
// (n_obs entries expected; -1 marks an invalid residual as the accumulator stores it) and the warm-start cache after the pass.
// Returns the cost.
CBA_EXPORT double ref_ba_system(int n_cameras, int n_images, int n_points, const int* cam9, int64_t n_obs, const float* obs_xy,
                                const int* obs_point, const int* obs_image, const int* obs_camera, const double* rig_tr_global,
                                const double* camera_tr_rig, const double* points, double* const* grids, double* last_projection,
                                double numerical_diff_delta, int localize_only, int eliminate_points, double* block_diag_H, double* off_diag_H,
                                double* dense_H, double* block_diag_b, double* dense_b, double* cost_vector, int64_t* n_costs) {
  Marshalled m;
  marshal(n_cameras, n_images, n_points, cam9, n_obs, obs_xy, obs_point, obs_image, obs_camera, rig_tr_global, camera_tr_rig, points, grids,
          last_projection, &m);
  JointOptimizationCostFunction cost_function;                       // as OptimizeJointly, :776-782
  cost_function.dataset = &m.ds;
  cost_function.numerical_diff_delta = numerical_diff_delta;
  cost_function.regularization_weight = 0.0;
  cost_function.localize_only = localize_only != 0;
  cost_function.eliminate_points = eliminate_points != 0;
  cost_function.on_the_fly_block_processing = false;
  std::vector<int> original_to_seq_index;
  JointOptimizationState opt_state(m.st, &original_to_seq_index, &cost_function.seq_to_original_index, localize_only != 0, eliminate_points != 0);
  const int bs = eliminate_points ? 3 : SE3d::DoF;
  const int nb = eliminate_points ? (int)opt_state.points.size() : (int)opt_state.rig_tr_global.size();
  const int block_dof = bs * nb, dd = opt_state.degrees_of_freedom() - block_dof;
  Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic> H_dense, H_off;
  std::vector<Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic>> H_blocks(nb);
  Eigen::Matrix<double, Eigen::Dynamic, 1> b_dense, b_block;
  H_dense.resize(dd, dd); H_off.resize(block_dof, dd); b_dense.resize(dd); b_block.resize(block_dof);
  for (auto& B : H_blocks) B.resize(bs, bs);
  std::vector<double> costs;
  UpdateEquationAccumulator<double> update_eq(block_dof, &H_dense, &H_off, nullptr, &H_blocks, &b_dense, &b_block, &costs, 0, 0.0);
  cost_function.template Compute<true>(opt_state, &update_eq);
  update_eq.AccumulateFinishedBlocks();
  for (int b = 0; b < nb; ++b) for (int r = 0; r < bs; ++r) for (int c = 0; c < bs; ++c) block_diag_H[((size_t)b * bs + r) * bs + c] = H_blocks[b](r, c);
  for (int r = 0; r < block_dof; ++r) for (int c = 0; c < dd; ++c) off_diag_H[(size_t)r * dd + c] = H_off(r, c);
  for (int r = 0; r < dd; ++r) for (int c = 0; c < dd; ++c) dense_H[(size_t)r * dd + c] = H_dense(r, c);
  for (int r = 0; r < block_dof; ++r) block_diag_b[r] = b_block(r);
  for (int r = 0; r < dd; ++r) dense_b[r] = b_dense(r);
  *n_costs = (int64_t)costs.size();
  for (size_t i = 0; i < costs.size() && (int64_t)i < n_obs; ++i) cost_vector[i] = costs[i];
  for (int64_t o = 0; o < n_obs; ++o) { last_projection[2 * o] = m.feats[o]->last_projection.x(); last_projection[2 * o + 1] = m.feats[o]->last_projection.y(); }
  return update_eq.cost();
}

// ---- the grid-only LM of the central-generic model, the reference's own (APP/models/central_generic.cc compiled whole) ----
// CentralGenericModel::FitToPixelDirections (:419-431 -> FitToPixelDirectionsImpl: LMOptimizer on DirectionGridFit...): grid in / out.
CBA_EXPORT void ref_f3_fit_to_pixel_directions(const int* cam8, double* grid, int64_t n, const double* pixels, const double* directions,
                                               int max_iteration_count) {
  auto m = make_model(cam8, grid);
  std::vector<Vec2d> px; std::vector<Vec3d> dirs;
  px.reserve(n); dirs.reserve(n);
  for (int64_t i = 0; i < n; ++i) {
    px.push_back(Vec2d(pixels[2 * i], pixels[2 * i + 1]));
    dirs.push_back(Vec3d(directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]));
  }
  m->FitToPixelDirections(px, dirs, max_iteration_count);
  store_grid(m.get(), grid);
}
// CentralGenericModel::FitToDenseModel (:267-417): dense_model is (height, width, 3) row-major with NaN for invalid pixels.  Returns 1 and
// the fitted grid, or 0 when the initialisation left grid points undefined (the reference returns false).  max_iteration_count = 0 stops
// after the initialisation + sample selection (the LM loop runs zero iterations): the initial grid.
CBA_EXPORT int ref_f3_fit_to_dense_model(const int* cam8, int dense_width, int dense_height, const double* dense_model, int subsample_step,
                                         int max_iteration_count, double* grid_out) {
  std::vector<double> zeros(3 * (size_t)cam8[6] * cam8[7], 0.0);
  auto m = make_model(cam8, zeros.data());
  Image<Vec3d> dense(dense_width, dense_height);
  for (int y = 0; y < dense_height; ++y)
    for (int x = 0; x < dense_width; ++x) {
      const double* d = dense_model + 3 * (x + (size_t)y * dense_width);
      dense(x, y) = Vec3d(d[0], d[1], d[2]);
    }
  const bool ok = m->FitToDenseModel(dense, subsample_step, max_iteration_count);
  store_grid(m.get(), grid_out);
  return ok ? 1 : 0;
}

// ---- the reference's own on-disk writers and its dataset reader (APP/io/calibration_io.cc, piped by oracle/Makefile) ----
// LoadDataset(path_in) followed by SaveDataset(path_out): a file in the reference's format is a fixed point of this.  Returns 0 on failure,
// else 1; counts[0..3] = cameras, imagesets, features, known geometries of what was loaded.
CBA_EXPORT int ref_f2_dataset_load_and_save(const char* path_in, const char* path_out, int64_t* counts) {
  Dataset ds;
  if (!LoadDataset(path_in, &ds)) return 0;
  int64_t features = 0;
  for (int i = 0; i < ds.ImagesetCount(); ++i)
    for (int c = 0; c < ds.num_cameras(); ++c) features += (int64_t)ds.GetImageset(i)->FeaturesOfCamera(c).size();
  counts[0] = ds.num_cameras(); counts[1] = ds.ImagesetCount(); counts[2] = features; counts[3] = ds.KnownGeometriesCount();
  return SaveDataset(path_out, ds) ? 1 : 0;
}
CBA_EXPORT int ref_f2_save_camera_model(const int* cam9, const double* grid, const char* path) {
  return SaveCameraModel(*make_any_model(cam9, grid), path) ? 1 : 0;
}
CBA_EXPORT int ref_f2_save_poses(int n, const uint8_t* image_used, const double* poses7, const char* path) {
  std::vector<bool> used(n); std::vector<SE3d> poses;
  for (int i = 0; i < n; ++i) { used[i] = image_used[i] != 0; poses.push_back(pose_of(poses7 + 7 * i)); }
  return SavePoses(used, poses, path) ? 1 : 0;
}
CBA_EXPORT int ref_f2_save_points(int n_points, const double* points, int n_map, const int* feature_ids, const int* point_index, const char* path) {
  BAState st;
  for (int p = 0; p < n_points; ++p) st.points.push_back(Vec3d(points[3 * p], points[3 * p + 1], points[3 * p + 2]));
  for (int i = 0; i < n_map; ++i) st.feature_id_to_points_index[feature_ids[i]] = point_index[i];
  return SavePointsAndIndexMapping(st, path) ? 1 : 0;
}

// ResampleModel (APP/calibration.cc:373-528), generic source and target models: cam9 / grid of the model to resample, the target type
// (0 central, 1 non-central) and resolution.  Returns 1 and the new grid(s) in the layout of make_any_model (central: G x 3; non-central:
// directions then points), or 0 where the reference returns false.
CBA_EXPORT int ref_f3_resample_model(const int* cam9, const double* grid, int target_type, int target_gw, int target_gh, double* grid_out) {
  std::shared_ptr<CameraModel> model = make_any_model(cam9, grid);
  SE3d camera_tr_rig;
  const bool ok = ResampleModel(model, &camera_tr_rig, cam9[3], cam9[4], cam9[5], cam9[6],
                                target_type == 0 ? CameraModel::Type::CentralGeneric : CameraModel::Type::NoncentralGeneric, target_gw, target_gh);
  if (!ok) return 0;
  store_any_grid(model.get(), grid_out);
  return 1;
}

// ---- The refinement stage of Calibrate() (APP/calibration.cc:1030-1142, compiled as the body of RefCalibrateRefinementStage): feature -> point
// indexing, full grid resolutions, the pyramid levels (RunBundleAdjustment(10, 1e-4), RunBundleAdjustment(50, 1), ResampleModel to the next
// level), the optional outlier stage (RunBundleAdjustment, DeleteOutlierFeatures per camera), the main RunBundleAdjustment(100, 1e-4) and
// ScaleToMetric -- reference code all the way down (libcalibref_ba.so).  The known geometry: one pattern with cell_length and integer positions
// (2 per point, feature id = point index).  cam9 / grids in: the models of the coarsest pyramid level; out: the models of the full resolution
// (cam9_out receives their grid sizes; grids_out must hold the full-resolution grids).  keep[o] / image_used[i]: what the outlier stage left.
// trace[0] = OptimizeJointly calls made.  Returns 0 where the reference returns false.
CBA_EXPORT int ref_f1_calibrate_refinement_stage(int num_pyramid_levels, int approx_pixels_per_cell, float outlier_removal_factor, int localize_only,
                                                 float cell_length, const int* positions_xy, int n_cameras, int n_images, int n_points,
                                                 const int* cam9, int64_t n_obs, const float* obs_xy, const int* obs_point, const int* obs_image,
                                                 const int* obs_camera, double* rig_tr_global, double* camera_tr_rig, double* points,
                                                 double* const* grids_in, int* cam9_out, double* const* grids_out, uint8_t* image_used,
                                                 uint8_t* keep, double* trace) {
  std::vector<double> lp(2 * (size_t)n_obs, 0.0);
  Marshalled m;
  marshal(n_cameras, n_images, n_points, cam9, n_obs, obs_xy, obs_point, obs_image, obs_camera, rig_tr_global, camera_tr_rig, points, grids_in,
          lp.data(), &m);
  m.ds.SetKnownGeometriesCount(1);
  m.ds.GetKnownGeometry(0).cell_length_in_meters = cell_length;
  for (int p = 0; p < n_points; ++p) m.ds.GetKnownGeometry(0).feature_id_to_position[p] = Vec2i(positions_xy[2 * p], positions_xy[2 * p + 1]);
  // tag the features with their packed index: PointFeature::last_projection.x is free for that BEFORE the first OptimizeJointly call only, so
  // the tags live in a side table keyed by (imageset, camera, feature id) instead -- a point is seen at most once per image and camera
  std::vector<std::unordered_map<int, int64_t>> tag((size_t)n_images * n_cameras);
  for (int64_t o = 0; o < n_obs; ++o) tag[(size_t)obs_image[o] * n_cameras + obs_camera[o]][obs_point[o]] = o;
  g_ref_ba_optimize_calls = 0;
  const CameraModel::Type type = cam9[0] == 0 ? CameraModel::Type::CentralGeneric : CameraModel::Type::NoncentralGeneric;
  const bool ok = RefCalibrateRefinementStage(&m.ds, nullptr, /*use_cuda*/ false, SchurMode::Dense, num_pyramid_levels, type, approx_pixels_per_cell,
                                              /*regularization_weight*/ 0.0, outlier_removal_factor, localize_only != 0, nullptr, &m.st, nullptr, nullptr);
  trace[0] = g_ref_ba_optimize_calls;
  if (!ok) return 0;
  BAState& st = m.st;
  for (int i = 0; i < n_images; ++i) { store_pose(st.rig_tr_global[i], rig_tr_global + 7 * i); image_used[i] = st.image_used[i] ? 1 : 0; }
  for (int c = 0; c < n_cameras; ++c) store_pose(st.camera_tr_rig[c], camera_tr_rig + 7 * c);
  for (int p = 0; p < n_points; ++p) for (int k = 0; k < 3; ++k) points[3 * p + k] = st.points[p](k);
  for (int c = 0; c < n_cameras; ++c) {
    int gw = 0, gh = 0;
    st.intrinsics[c]->GetGridResolution(&gw, &gh);
    for (int k = 0; k < 9; ++k) cam9_out[9 * c + k] = cam9[9 * c + k];
    cam9_out[9 * c] = st.intrinsics[c]->type() == CameraModel::Type::CentralGeneric ? 0 : 1;
    cam9_out[9 * c + 7] = gw; cam9_out[9 * c + 8] = gh;
    store_any_grid(st.intrinsics[c].get(), grids_out[c]);
  }
  for (int64_t o = 0; o < n_obs; ++o) keep[o] = 0;
  for (int i = 0; i < n_images; ++i)
    for (int c = 0; c < n_cameras; ++c)
      for (const PointFeature& f : m.ds.GetImageset(i)->FeaturesOfCamera(c)) keep[tag[(size_t)i * n_cameras + c][f.id]] = 1;
  return 1;
}
