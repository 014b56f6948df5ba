// vis::OptimizeJointly on the MI355X engine: host adapter between the reference's C++ boundary types
// and the C-ABI (include/cba.h).  Restates the marshalling parts of
// APP/bundle_adjustment/joint_optimization.cc:757-953 (sequential image indexing :80-90, read-back
// :942-950); the LM iteration itself is cba_step.
#include "joint_optimization.h"

#include <cstdio>
#include <cstdlib>

namespace vis {

static int g_device = -1;
static int hip_device() {
  if (g_device >= 0) return g_device;
  const char* e = std::getenv("CBA_DEVICE");
  return e ? std::atoi(e) : 0;
}
void SetHipDevice(int device) { g_device = device; }

// ---- CameraModel calls -> cba_project / cba_unproject ------------------------------------------------
bool CameraModel::ProjectWithInitialEstimate(const Vec3d& local_point, Vec2d* result) const {
  cba_camera cam = abi_camera();
  std::vector<double> grid = abi_grid();
  double px[2], init[2] = {result->x(), result->y()};
  uint8_t ok = 0;
  int rc = cba_project(&cam, grid.data(), 1, local_point.v, init, px, &ok, hip_device());
  if (rc != CBA_OK) { std::fprintf(stderr, "CameraModel::ProjectWithInitialEstimate: %s\n", cba_last_error()); return false; }
  if (ok) *result = Vec2d(px[0], px[1]);
  return ok != 0;
}
bool CameraModel::Project(const Vec3d& local_point, Vec2d* result) const {
  *result = CenterOfCalibratedArea();
  return ProjectWithInitialEstimate(local_point, result);
}
bool CameraModel::Unproject(double x, double y, Line3d* result) const {
  cba_camera cam = abi_camera();
  std::vector<double> grid = abi_grid();
  double px[2] = {x, y}, line[6];
  uint8_t ok = 0;
  int rc = cba_unproject(&cam, grid.data(), 1, px, line, nullptr, &ok, hip_device());
  if (rc != CBA_OK) { std::fprintf(stderr, "CameraModel::Unproject: %s\n", cba_last_error()); return false; }
  if (ok) { result->direction() = Vec3d(line[0], line[1], line[2]); result->origin() = Vec3d(line[3], line[4], line[5]); }
  return ok != 0;
}

static void pack_pose(const SE3d& T, double* o) {
  const Quaterniond& q = T.unit_quaternion();
  o[0] = q.w(); o[1] = q.x(); o[2] = q.y(); o[3] = q.z();
  o[4] = T.translation().x(); o[5] = T.translation().y(); o[6] = T.translation().z();
}
static SE3d unpack_pose(const double* o) { return SE3d(Quaterniond(o[0], o[1], o[2], o[3]), Vec3d(o[4], o[5], o[6])); }

double OptimizeJointly(Dataset& dataset, BAState* state, int max_iteration_count, double init_lambda,
                       double numerical_diff_delta, double regularization_weight, bool localize_only,
                       bool eliminate_points, SchurMode /*schur_mode*/, double* final_lambda,
                       bool* performed_an_iteration, bool debug_verify_cost, bool debug_fix_points,
                       bool debug_fix_poses, bool debug_fix_rig_poses, bool debug_fix_intrinsics, bool print_progress) {
  if (performed_an_iteration) *performed_an_iteration = false;
  if (regularization_weight > 0)  // joint_optimization.cc:299-305: disabled in the reference as well
    std::fprintf(stderr, "OptimizeJointly: Regularization is disabled at the moment since it is untested with the current version.\n");
  if (debug_fix_points || debug_fix_poses || debug_fix_rig_poses || debug_fix_intrinsics) {
    std::fprintf(stderr, "OptimizeJointly(HIP): debug_fix_* is not supported by this backend\n");
    std::abort();  // the reference signals programmer errors with CHECK() aborts
  }
  const int C = state->num_cameras();
  // sequential indexing of the used imagesets (JointOptimizationState ctor, joint_optimization.cc:80-90)
  std::vector<int> seq_to_original;
  for (usize i = 0; i < state->rig_tr_global.size(); ++i)
    if (state->image_used[i]) seq_to_original.push_back((int)i);
  const int N = (int)seq_to_original.size(), P = (int)state->points.size();

  std::vector<cba_camera> cams(C);
  std::vector<std::vector<double>> grids(C);
  std::vector<const double*> grid_ptrs(C);
  for (int c = 0; c < C; ++c) { cams[c] = state->intrinsics[c]->abi_camera(); grids[c] = state->intrinsics[c]->abi_grid(); grid_ptrs[c] = grids[c].data(); }

  // observations, image-major then camera then feature order (the loop order of Compute, :273-291)
  std::vector<float> xy; std::vector<int32_t> pt, im, cm; std::vector<double> lastp;
  std::vector<PointFeature*> feature_ptrs;
  for (int s = 0; s < N; ++s)
    for (int c = 0; c < C; ++c)
      for (PointFeature& f : dataset.GetImageset(seq_to_original[s])->FeaturesOfCamera(c)) {
        xy.push_back(f.xy.x()); xy.push_back(f.xy.y());
        pt.push_back(f.index); im.push_back(s); cm.push_back(c);
        lastp.push_back(f.last_projection.x()); lastp.push_back(f.last_projection.y());
        feature_ptrs.push_back(&f);
      }
  const int64_t n_obs = (int64_t)pt.size();

  cba_config cfg{};
  cfg.n_cameras = C; cfg.cameras = cams.data(); cfg.n_images = N; cfg.n_points = P;
  cfg.numerical_diff_delta = numerical_diff_delta; cfg.localize_only = localize_only; cfg.eliminate_points = eliminate_points;
  cfg.device = hip_device();
  cba_problem* prob = nullptr;
  auto fail = [&](const char* what) -> double {
    std::fprintf(stderr, "OptimizeJointly(HIP): %s failed: %s\n", what, cba_last_error());
    if (prob) cba_destroy(prob);
    std::abort();
    return -1;
  };
  if (cba_create(&cfg, &prob) != CBA_OK) return fail("cba_create");
  if (cba_set_observations(prob, n_obs, xy.data(), pt.data(), im.data(), cm.data(), lastp.data()) != CBA_OK) return fail("cba_set_observations");
  std::vector<double> rig(7 * (size_t)N), camrig(7 * (size_t)C), points(3 * (size_t)P);
  for (int s = 0; s < N; ++s) pack_pose(state->rig_tr_global[seq_to_original[s]], &rig[7 * (size_t)s]);
  for (int c = 0; c < C; ++c) pack_pose(state->camera_tr_rig[c], &camrig[7 * (size_t)c]);
  for (int p = 0; p < P; ++p) for (int k = 0; k < 3; ++k) points[3 * (size_t)p + k] = state->points[p].v[k];
  if (cba_set_state(prob, rig.data(), camrig.data(), points.data(), grid_ptrs.data()) != CBA_OK) return fail("cba_set_state");

  if (debug_verify_cost) {  // joint_optimization.cc:866-877
    double c1 = 0, c2 = 0;
    if (cba_cost(prob, &c1, nullptr, nullptr) != CBA_OK || cba_cost(prob, &c2, nullptr, nullptr) != CBA_OK) return fail("cba_cost");
    if (!(std::fabs(c1 - c2) <= 1e-3f)) { std::fprintf(stderr, "OptimizeJointly(HIP): VerifyCost failed (%g vs %g)\n", c1, c2); std::abort(); }
  }

  double final_cost = -1;
  for (int iteration = 0; iteration < max_iteration_count; ++iteration) {  // joint_optimization.cc:906-940
    cba_report rep;
    if (cba_step(prob, init_lambda, /*max_lm_attempts*/ 50, /*init_lambda_factor*/ 0.00001, &rep) != CBA_OK) return fail("cba_step");
    final_cost = rep.final_cost;
    init_lambda = rep.lambda;
    if (final_lambda) *final_lambda = rep.lambda;
    if (print_progress)
      std::fprintf(stderr, "LMOptimizer: [%d] Initial cost: %.9g  Final cost: %.9g  lambda: %.4g  (attempts %d; jac %.1f ms, solve %.1f ms)\n",
                   iteration, rep.initial_cost, rep.final_cost, rep.lambda, rep.lm_attempts, rep.t_jac * 1e3, rep.t_solve * 1e3);
    if (!rep.accepted) break;
    if (performed_an_iteration) *performed_an_iteration = true;
  }

  // read back (joint_optimization.cc:942-950) + the warm-start cache the reference mutates in place
  std::vector<double*> grid_out(C);
  for (int c = 0; c < C; ++c) grid_out[c] = grids[c].data();
  if (cba_get_state(prob, rig.data(), camrig.data(), points.data(), grid_out.data()) != CBA_OK) return fail("cba_get_state");
  if (n_obs && cba_get_last_projection(prob, lastp.data()) != CBA_OK) return fail("cba_get_last_projection");
  for (int c = 0; c < C; ++c) state->camera_tr_rig[c] = unpack_pose(&camrig[7 * (size_t)c]);
  for (int s = 0; s < N; ++s) state->rig_tr_global[seq_to_original[s]] = unpack_pose(&rig[7 * (size_t)s]);
  for (int p = 0; p < P; ++p) state->points[p] = Vec3d(points[3 * (size_t)p], points[3 * (size_t)p + 1], points[3 * (size_t)p + 2]);
  for (int c = 0; c < C; ++c) {
    std::shared_ptr<CameraModel> dup(state->intrinsics[c]->duplicate());
    dup->set_abi_grid(grids[c].data());
    state->intrinsics[c] = dup;
  }
  for (int64_t o = 0; o < n_obs; ++o) feature_ptrs[o]->last_projection = Vec2d(lastp[2 * o], lastp[2 * o + 1]);
  cba_destroy(prob);
  return final_cost;
}

}  // namespace vis
