// vis::OptimizeJointly on the MI355X engine: host adapter between the reference's C++ boundary types
// and the C-ABI (include/cba.h).  Restates the marshalling parts of
// APP/bundle_adjustment/joint_optimization.cc:757-953 (sequential image indexing :80-90, read-back
// :942-950); the LM iteration itself is cba_step.
#include "joint_optimization.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace vis {

static int g_device = -1;
static int hip_device() {
  if (g_device >= 0) return g_device;
  const char* e = std::getenv("CBA_DEVICE");
  return e ? std::atoi(e) : 0;
}
void SetHipDevice(int device) { g_device = device; }

// ---- CameraModel calls -> the device-resident model (cba_model_*) -----------------------------------------
cba_model* CameraModel::device_model(int device_ordinal) const {
  if (m_dev && m_dev_device != device_ordinal) release_device_model();
  if (m_dev) {
    // the handle was created for one camera description: a SetGrid / operator= with another grid resolution or other
    // calibration bounds needs a new one (cba_model_set_grid would copy the OLD grid's length out of the new vector)
    const cba_camera now = abi_camera();
    if (std::memcmp(&now, &m_dev_camera, sizeof(cba_camera)) != 0) release_device_model();
  }
  if (!m_dev) {
    cba_camera cam = abi_camera();
    std::vector<double> grid = abi_grid();
    if (cba_model_create(&cam, grid.data(), device_ordinal, &m_dev) != CBA_OK) { m_dev = nullptr; return nullptr; }
    m_dev_camera = cam;
    m_dev_device = device_ordinal;
    m_dev_stale = false;
  } else if (m_dev_stale) {
    std::vector<double> grid = abi_grid();
    if (cba_model_set_grid(m_dev, grid.data()) != CBA_OK) return nullptr;
    m_dev_stale = false;
  }
  return m_dev;
}
void CameraModel::release_device_model() const {
  if (m_dev) cba_model_destroy(m_dev);
  m_dev = nullptr; m_dev_device = -1; m_dev_stale = true;
}
bool CameraModel::ProjectWithInitialEstimate(const Vec3d& local_point, Vec2d* result) const {
  cba_model* dm = device_model(hip_device());
  double px[2], init[2] = {result->x(), result->y()};
  uint8_t ok = 0;
  if (!dm || cba_model_project(dm, 1, local_point.v, init, px, &ok) != CBA_OK) {
    std::fprintf(stderr, "CameraModel::ProjectWithInitialEstimate: %s\n", cba_last_error());
    return false;
  }
  if (ok) *result = Vec2d(px[0], px[1]);
  return ok != 0;
}
bool CameraModel::Project(const Vec3d& local_point, Vec2d* result) const {
  *result = CenterOfCalibratedArea();
  return ProjectWithInitialEstimate(local_point, result);
}
bool CameraModel::Unproject(double x, double y, Line3d* result) const {
  cba_model* dm = device_model(hip_device());
  double px[2] = {x, y}, line[6];
  uint8_t ok = 0;
  if (!dm || cba_model_unproject(dm, 1, px, line, nullptr, &ok) != CBA_OK) {
    std::fprintf(stderr, "CameraModel::Unproject: %s\n", cba_last_error());
    return false;
  }
  if (ok) { result->direction() = Vec3d(line[0], line[1], line[2]); result->origin() = Vec3d(line[3], line[4], line[5]); }
  return ok != 0;
}

static void pack_pose(const SE3d& T, double* o) {
  const Quaterniond& q = T.unit_quaternion();
  o[0] = q.w(); o[1] = q.x(); o[2] = q.y(); o[3] = q.z();
  o[4] = T.translation().x(); o[5] = T.translation().y(); o[6] = T.translation().z();
}
static SE3d unpack_pose(const double* o) { return SE3d(Quaterniond(o[0], o[1], o[2], o[3]), Vec3d(o[4], o[5], o[6])); }

struct JointOptimizationSession::Impl {
  Dataset* dataset = nullptr;
  BAState* state = nullptr;
  cba_problem* prob = nullptr;
  SchurMode schur_mode = SchurMode::Dense;
  bool eliminate_points = false;
  int C = 0, N = 0, P = 0;
  int64_t n_obs = 0;
  std::vector<int> seq_to_original;
  std::vector<cba_camera> cams;
  std::vector<std::vector<double>> grids;
  std::vector<PointFeature*> feature_ptrs;
  std::vector<double> rig, camrig, points;
  [[noreturn]] void fail(const char* what) {
    std::fprintf(stderr, "OptimizeJointly(HIP): %s failed: %s\n", what, cba_last_error());
    if (prob) cba_destroy(prob);
    std::abort();   // the reference signals unrecoverable errors with CHECK() aborts as well
  }
  void pack_state() {
    for (int s = 0; s < N; ++s) pack_pose(state->rig_tr_global[seq_to_original[s]], &rig[7 * (size_t)s]);
    for (int c = 0; c < C; ++c) pack_pose(state->camera_tr_rig[c], &camrig[7 * (size_t)c]);
    for (int p = 0; p < P; ++p) for (int k = 0; k < 3; ++k) points[3 * (size_t)p + k] = state->points[p].v[k];
    for (int c = 0; c < C; ++c) grids[c] = state->intrinsics[c]->abi_grid();
  }
};

JointOptimizationSession::JointOptimizationSession(Dataset& dataset, BAState* state, double numerical_diff_delta, bool localize_only,
                                                   bool eliminate_points, SchurMode schur_mode)
    : m(new Impl()) {
  const auto t0 = std::chrono::steady_clock::now();
  m->dataset = &dataset; m->state = state; m->schur_mode = schur_mode; m->eliminate_points = eliminate_points;
  m->C = state->num_cameras();
  // sequential indexing of the used imagesets (JointOptimizationState ctor, joint_optimization.cc:80-90)
  for (usize i = 0; i < state->rig_tr_global.size(); ++i)
    if (state->image_used[i]) m->seq_to_original.push_back((int)i);
  m->N = (int)m->seq_to_original.size(); m->P = (int)state->points.size();
  m->cams.resize(m->C); m->grids.resize(m->C);
  for (int c = 0; c < m->C; ++c) m->cams[c] = state->intrinsics[c]->abi_camera();
  // observations, image-major then camera then feature order (the loop order of Compute, :273-291)
  std::vector<float> xy; std::vector<int32_t> pt, im, cm; std::vector<double> lastp;
  for (int s = 0; s < m->N; ++s)
    for (int c = 0; c < m->C; ++c)
      for (PointFeature& f : dataset.GetImageset(m->seq_to_original[s])->FeaturesOfCamera(c)) {
        xy.push_back(f.xy.x()); xy.push_back(f.xy.y());
        pt.push_back(f.index); im.push_back(s); cm.push_back(c);
        lastp.push_back(f.last_projection.x()); lastp.push_back(f.last_projection.y());
        m->feature_ptrs.push_back(&f);
      }
  m->n_obs = (int64_t)pt.size();
  cba_config cfg{};
  cfg.n_cameras = m->C; cfg.cameras = m->cams.data(); cfg.n_images = m->N; cfg.n_points = m->P;
  cfg.numerical_diff_delta = numerical_diff_delta; cfg.localize_only = localize_only; cfg.eliminate_points = eliminate_points;
  cfg.device = hip_device();
  if (cba_create(&cfg, &m->prob) != CBA_OK) m->fail("cba_create");
  if (cba_set_observations(m->prob, m->n_obs, xy.data(), pt.data(), im.data(), cm.data(), lastp.data()) != CBA_OK) m->fail("cba_set_observations");
  m->rig.resize(7 * (size_t)m->N); m->camrig.resize(7 * (size_t)m->C); m->points.resize(3 * (size_t)m->P);
  UploadState();
  m_setup_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

JointOptimizationSession::~JointOptimizationSession() {
  if (m->prob) cba_destroy(m->prob);
  delete m;
}

void JointOptimizationSession::UploadState() {
  m->pack_state();
  std::vector<const double*> grid_ptrs(m->C);
  for (int c = 0; c < m->C; ++c) grid_ptrs[c] = m->grids[c].data();
  if (cba_set_state(m->prob, m->rig.data(), m->camrig.data(), m->points.data(), grid_ptrs.data()) != CBA_OK) m->fail("cba_set_state");
}

void JointOptimizationSession::VerifyCost() {
  double c1 = 0, c2 = 0;
  if (cba_cost(m->prob, &c1, nullptr, nullptr) != CBA_OK || cba_cost(m->prob, &c2, nullptr, nullptr) != CBA_OK) m->fail("cba_cost");
  if (!(std::fabs(c1 - c2) <= 1e-3f)) { std::fprintf(stderr, "OptimizeJointly(HIP): VerifyCost failed (%g vs %g)\n", c1, c2); std::abort(); }
}

double JointOptimizationSession::Optimize(int max_iteration_count, double init_lambda, double* final_lambda, bool* performed_an_iteration,
                                          bool print_progress) {
  if (performed_an_iteration) *performed_an_iteration = false;
  // on-the-fly block processing starts from a fixed lambda (joint_optimization.cc:801-808); the HIP engine runs every
  // SchurMode on its dense path, but a drop-in caller must see the reference's lambda trajectory
  if (!m->eliminate_points && (m->schur_mode == SchurMode::DenseOnTheFly || m->schur_mode == SchurMode::SparseOnTheFly) && init_lambda < 0)
    init_lambda = 0.0001f;
  double final_cost = -1;
  for (int iteration = 0; iteration < max_iteration_count; ++iteration) {  // joint_optimization.cc:906-940
    cba_report rep;
    if (cba_step(m->prob, init_lambda, /*max_lm_attempts*/ 50, /*init_lambda_factor*/ 0.00001, &rep) != CBA_OK) m->fail("cba_step");
    final_cost = rep.final_cost;
    init_lambda = rep.lambda;
    if (final_lambda) *final_lambda = rep.lambda;
    if (print_progress)
      std::fprintf(stderr, "LMOptimizer: [%d] Initial cost: %.9g  Final cost: %.9g  lambda: %.4g  (attempts %d; jac %.1f ms, solve %.1f ms)\n",
                   iteration, rep.initial_cost, rep.final_cost, rep.lambda, rep.lm_attempts, rep.t_jac * 1e3, rep.t_solve * 1e3);
    if (!rep.accepted) break;
    if (performed_an_iteration) *performed_an_iteration = true;
  }
  return final_cost;
}

void JointOptimizationSession::ReadBackState() {   // joint_optimization.cc:942-950
  BAState* state = m->state;
  std::vector<double*> grid_out(m->C);
  for (int c = 0; c < m->C; ++c) grid_out[c] = m->grids[c].data();
  if (cba_get_state(m->prob, m->rig.data(), m->camrig.data(), m->points.data(), grid_out.data()) != CBA_OK) m->fail("cba_get_state");
  for (int c = 0; c < m->C; ++c) state->camera_tr_rig[c] = unpack_pose(&m->camrig[7 * (size_t)c]);
  for (int s = 0; s < m->N; ++s) state->rig_tr_global[m->seq_to_original[s]] = unpack_pose(&m->rig[7 * (size_t)s]);
  for (int p = 0; p < m->P; ++p) state->points[p] = Vec3d(m->points[3 * (size_t)p], m->points[3 * (size_t)p + 1], m->points[3 * (size_t)p + 2]);
  for (int c = 0; c < m->C; ++c) {
    std::shared_ptr<CameraModel> dup(state->intrinsics[c]->duplicate());
    dup->set_abi_grid(m->grids[c].data());
    state->intrinsics[c] = dup;
  }
}

void JointOptimizationSession::ReadBackLastProjections() {
  if (!m->n_obs) return;
  std::vector<double> lastp(2 * (size_t)m->n_obs);
  if (cba_get_last_projection(m->prob, lastp.data()) != CBA_OK) m->fail("cba_get_last_projection");
  for (int64_t o = 0; o < m->n_obs; ++o) m->feature_ptrs[o]->last_projection = Vec2d(lastp[2 * o], lastp[2 * o + 1]);
}

double OptimizeJointly(Dataset& dataset, BAState* state, int max_iteration_count, double init_lambda,
                       double numerical_diff_delta, double regularization_weight, bool localize_only,
                       bool eliminate_points, SchurMode schur_mode, double* final_lambda,
                       bool* performed_an_iteration, bool debug_verify_cost, bool debug_fix_points,
                       bool debug_fix_poses, bool debug_fix_rig_poses, bool debug_fix_intrinsics, bool print_progress) {
  if (performed_an_iteration) *performed_an_iteration = false;
  if (regularization_weight > 0)  // joint_optimization.cc:299-305: disabled in the reference as well
    std::fprintf(stderr, "OptimizeJointly: Regularization is disabled at the moment since it is untested with the current version.\n");
  if (debug_fix_points || debug_fix_poses || debug_fix_rig_poses || debug_fix_intrinsics) {
    std::fprintf(stderr, "OptimizeJointly(HIP): debug_fix_* is not supported by this backend\n");
    std::abort();  // the reference signals programmer errors with CHECK() aborts
  }
  JointOptimizationSession session(dataset, state, numerical_diff_delta, localize_only, eliminate_points, schur_mode);
  if (debug_verify_cost) session.VerifyCost();
  const double final_cost = session.Optimize(max_iteration_count, init_lambda, final_lambda, performed_an_iteration, print_progress);
  session.ReadBackState();
  session.ReadBackLastProjections();
  return final_cost;
}

}  // namespace vis
