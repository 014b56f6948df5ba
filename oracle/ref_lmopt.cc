// oracle/_ref, part 4 -- TEST INFRASTRUCTURE ONLY.
//
// The reference's Levenberg-Marquardt driver and its Schur-complement solve, compiled from where they lie:
//   LV/lm_optimizer.h:629-991     LMOptimizer<double>::OptimizeImpl   (lambda initialisation :766-781, LM attempts :802-965, NaN update
//                                 :905-913, accept / reject and the lambda x 0.5 / x 2 rule :943-977)                       row B4
//   LV/lm_optimizer.h:993-1011    CostIsSmallerThan                                                                         row B5
//   LV/lm_optimizer.h:1247-1369   SolveWithSchurComplementDenseOffDiag (D^-1 B, B^T D^-1 B, the reduced system, the back
//                                 substitution of the block part)                                                           row B6
//   LV/lm_optimizer_update_accumulator.h, lm_optimizer_jtj_accumulator_base.h (as in ref_lm.cc)                             rows B2, B3
// (LV = /root/reference/libvis/src/libvis, APP = /root/reference/applications/camera_calibration/src/camera_calibration.)
// Nothing is copied: the headers are #included and built against the run-time-sized Eigen stand-in of oracle/ref_shim_lm.
//
// What this file adds around them:
//   * OrcState / OrcCost: the State and CostFunction LMOptimizer::Optimize is instantiated with.  The per-observation numbers
//     (projection, residual, the analytic and finite-difference Jacobian blocks) come from the oracle's passes
//     (orc_jacobian_pass / orc_cost_pass_records, themselves pinned to reference code piece by piece: ref_app.cc, ref_lm.cc) and are
//     handed to the REFERENCE's accumulator with the call sequence of JointOptimizationCostFunction::Compute
//     (APP/bundle_adjustment/joint_optimization.cc:334-342 AddInvalidResidual, :345-347 / :373-376 / :446-448 AddResidual,
//     :479-590 AddResidualWithJacobian dispatch), so OptimizeImpl runs on the real bundle-adjustment problem;
//   * the calls of OptimizeJointly around Optimize (APP/bundle_adjustment/joint_optimization.cc:797-812, :916-940):
//     UseBlockDiagonalStructureForSchurComplement(block_size, num_blocks, dense, no on-the-fly, 1024, no CUDA), then per outer
//     iteration Optimize(max_iteration_count 1, max_lm_attempts 50, init_lambda, init_lambda_factor 0.00001) carrying lambda();
//   * Eigen's LDLT (lm_optimizer.h:1289, 1361) stays the oracle's restatement (ref_shim_lm/Eigen/Core: LDLT::solve calls
//     orc_ldlt_solve_upper_unblocked) -- the one piece of the solve that is NOT reference code here.
// tests/test_oracle_vs_ref.py compares the trajectory (accept decisions, LM attempts, lambda, costs, states) and the Schur algebra
// with the oracle's own restatement (orc_optimize_jointly, orc_schur_solve).
#include <ostream>
namespace cba_ref_shim {
struct NullLog {
  template <class T> NullLog& operator<<(const T&) { return *this; }
  NullLog& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
};
}
#define LOG(severity) ::cba_ref_shim::NullLog()

#include <cuda_runtime.h>
#include <libvis/eigen.h>
#include <libvis/libvis.h>

#include "libvis/lm_optimizer.h"

#include "../cba_oracle.h"

using namespace vis;

namespace {

// JointOptimizationState's variable ordering (joint_optimization.cc:49-59, 142-170), as oracle/ref.py: accumulate_records
struct Offsets {
  int first_rig, first_cam, first_pts;
  std::vector<int> intr;
  Offsets(const orc_problem* pb) {
    const int rig_dof = pb->n_cameras > 1 ? 6 * pb->n_cameras : 0;
    first_rig = 0;
    first_cam = 6 * pb->n_images;
    first_pts = first_cam + rig_dof;
    int at = first_pts + 3 * pb->n_points;
    for (int c = 0; c < pb->n_cameras; ++c) { intr.push_back(at); at += orc_intrinsics_param_count(&pb->cams[c]); }
    if (pb->eliminate_points) {
      first_pts = 0;
      first_rig = 3 * pb->n_points;
      first_cam = first_rig + 6 * pb->n_images;
    }
  }
};

size_t grid_doubles(const orc_camera& c) { return (size_t)(c.model_type == ORC_CENTRAL_GENERIC ? 3 : 6) * c.grid_w * c.grid_h; }

// The optimisation state: deep copies (is_reversible() is false for JointOptimizationState, joint_optimization.cc:135, so
// OptimizeImpl works on a copy per LM attempt and assigns it back on acceptance, lm_optimizer.h:915-917, :955-957)
struct OrcState {
  const orc_problem* pb = nullptr;
  std::vector<double> rig, cam, pts;
  std::vector<std::vector<double>> grids;
  mutable std::vector<double*> grid_ptrs;

  OrcState() {}
  OrcState(const orc_problem* p, const orc_state* st) : pb(p) {
    rig.assign(st->rig_tr_global, st->rig_tr_global + 7 * (size_t)p->n_images);
    cam.assign(st->camera_tr_rig, st->camera_tr_rig + 7 * (size_t)p->n_cameras);
    pts.assign(st->points, st->points + 3 * (size_t)p->n_points);
    for (int c = 0; c < p->n_cameras; ++c) grids.emplace_back(st->grids[c], st->grids[c] + grid_doubles(p->cams[c]));
  }
  orc_state view() const {
    grid_ptrs.resize(grids.size());
    for (size_t c = 0; c < grids.size(); ++c) grid_ptrs[c] = const_cast<double*>(grids[c].data());
    orc_state s;
    s.rig_tr_global = const_cast<double*>(rig.data());
    s.camera_tr_rig = const_cast<double*>(cam.data());
    s.points = const_cast<double*>(pts.data());
    s.grids = grid_ptrs.data();
    return s;
  }
  void store(orc_state* st) const {
    std::copy(rig.begin(), rig.end(), st->rig_tr_global);
    std::copy(cam.begin(), cam.end(), st->camera_tr_rig);
    std::copy(pts.begin(), pts.end(), st->points);
    for (size_t c = 0; c < grids.size(); ++c) std::copy(grids[c].begin(), grids[c].end(), st->grids[c]);
  }
  int degrees_of_freedom() const { return orc_total_dof(pb); }
  static constexpr bool is_reversible() { return false; }
  template <typename Derived>
  void operator-=(const MatrixBase<Derived>& delta) {           // JointOptimizationState::operator-= (joint_optimization.cc:172-214)
    std::vector<double> x(delta.rows());
    for (int i = 0; i < delta.rows(); ++i) x[i] = delta(i);
    OrcState out(*this);
    orc_state in_v = view(), out_v = out.view();
    orc_apply_update(pb, &in_v, x.data(), &out_v);
    *this = out;
  }
};

struct OrcCost {
  const orc_problem* pb;
  mutable int jacobian_passes = 0, cost_passes = 0;
  mutable std::vector<orc_obs_record> recs;

  template <int K, class Accumulator>
  void add_with_jacobian(Accumulator* accumulator, const Offsets& off, int o, const orc_obs_record& r) const {
    const Vec2d residual(r.residual[0], r.residual[1]);
    Matrix<double, 2, 6> pose_jacobian, rig_jacobian;
    Matrix<double, 2, 3> point_jacobian;
    for (int i = 0; i < 2; ++i) {
      for (int j = 0; j < 6; ++j) { pose_jacobian(i, j) = r.pose_jac[6 * i + j]; rig_jacobian(i, j) = r.rig_jac[6 * i + j]; }
      for (int j = 0; j < 3; ++j) point_jacobian(i, j) = r.point_jac[3 * i + j];
    }
    const int cam = pb->obs_camera[o];
    const usize pose_jac_index = off.first_rig + 6 * pb->obs_image[o];
    const usize rig_jac_index = off.first_cam + 6 * cam;
    const usize point_jac_index = off.first_pts + 3 * pb->obs_point[o];
    const bool rig_in_state = pb->n_cameras > 1, eliminate_points = pb->eliminate_points, localize_only = pb->localize_only;
    Matrix<int, K, 1> grid_update_indices;
    Matrix<double, 2, K, Eigen::RowMajor> pixel_wrt_grid_updates;
    if (!localize_only)
      for (int i = 0; i < K; ++i) {
        grid_update_indices(i) = off.intr[cam] + r.grid_indices[i];
        pixel_wrt_grid_updates(0, i) = r.grid_jac[i];
        pixel_wrt_grid_updates(1, i) = r.grid_jac[K + i];
      }
    // AccumulateModelJacobian's dispatch, joint_optimization.cc:479-590 (the same twelve calls as ref_lm.cc: add_observation)
    if (localize_only) {
      if (eliminate_points) {
        if (rig_in_state) accumulator->AddResidualWithJacobian(residual, point_jac_index, point_jacobian, pose_jac_index, pose_jacobian, rig_jac_index, rig_jacobian, true, true, true, HuberLoss<double>(1.0));
        else accumulator->AddResidualWithJacobian(residual, point_jac_index, point_jacobian, pose_jac_index, pose_jacobian, true, true, HuberLoss<double>(1.0));
      } else {
        if (rig_in_state) accumulator->AddResidualWithJacobian(residual, pose_jac_index, pose_jacobian, rig_jac_index, rig_jacobian, point_jac_index, point_jacobian, true, true, true, HuberLoss<double>(1.0));
        else accumulator->AddResidualWithJacobian(residual, pose_jac_index, pose_jacobian, point_jac_index, point_jacobian, true, true, HuberLoss<double>(1.0));
      }
    } else {
      if (eliminate_points) {
        if (rig_in_state) accumulator->AddResidualWithJacobian(residual, point_jac_index, point_jacobian, pose_jac_index, pose_jacobian, rig_jac_index, rig_jacobian, grid_update_indices, pixel_wrt_grid_updates, true, true, true, HuberLoss<double>(1.0));
        else accumulator->AddResidualWithJacobian(residual, point_jac_index, point_jacobian, pose_jac_index, pose_jacobian, grid_update_indices, pixel_wrt_grid_updates, true, true, HuberLoss<double>(1.0));
      } else {
        if (rig_in_state) accumulator->AddResidualWithJacobian(residual, pose_jac_index, pose_jacobian, rig_jac_index, rig_jacobian, point_jac_index, point_jacobian, grid_update_indices, pixel_wrt_grid_updates, true, true, true, HuberLoss<double>(1.0));
        else accumulator->AddResidualWithJacobian(residual, pose_jac_index, pose_jacobian, point_jac_index, point_jacobian, grid_update_indices, pixel_wrt_grid_updates, true, true, HuberLoss<double>(1.0));
      }
    }
  }

  // JointOptimizationCostFunction::Compute (joint_optimization.cc:240-449): loop order = observation order of the packed arrays
  template <bool compute_jacobians, class Accumulator>
  void Compute(const OrcState& state, Accumulator* accumulator) const {
    const Offsets off(pb);
    recs.resize((size_t)pb->n_obs);
    orc_state st = state.view();
    if (compute_jacobians) { orc_jacobian_pass(pb, &st, nullptr, nullptr, recs.data(), 0, pb->n_images); ++jacobian_passes; }
    else { orc_cost_pass_records(pb, &st, nullptr, recs.data()); ++cost_passes; }
    for (int64_t o = 0; o < pb->n_obs; ++o) {
      const orc_obs_record& r = recs[(size_t)o];
      if (!r.valid) { accumulator->AddInvalidResidual(); continue; }                                 // :334-342
      if (!compute_jacobians || !r.has_jacobian) {                                                  // :345-347, :373-376, :446-448
        accumulator->AddResidual(Vec2d(r.residual[0], r.residual[1]), HuberLoss<double>(1.0));
        continue;
      }
      if (pb->cams[pb->obs_camera[o]].model_type == ORC_CENTRAL_GENERIC) add_with_jacobian<32>(accumulator, off, (int)o, r);
      else add_with_jacobian<80>(accumulator, off, (int)o, r);
    }
  }
};

}  // namespace

// the friend the reference declares for its own test (lm_optimizer.h:1627; LV/test/lm_optimizer.cc:476-543): access to the
// private system and to SolveWithSchurComplementDenseOffDiag
namespace vis {
class LMOptimizerTestHelper {
 public:
  static void SchurSolve(int bs, int nb, int dd, const double* block_diag_H, const double* off_diag_H, const double* dense_H,
                         const double* block_diag_b, const double* dense_b, double* x) {
    LMOptimizer<double> o;
    o.m_use_block_diagonal_structure = true;
    o.m_block_size = bs;
    o.m_num_blocks = nb;
    o.m_block_diag_H.resize(nb);
    for (int b = 0; b < nb; ++b) {
      o.m_block_diag_H[b].resize(bs, bs);
      for (int r = 0; r < bs; ++r) for (int c = 0; c < bs; ++c) o.m_block_diag_H[b](r, c) = block_diag_H[((size_t)b * bs + r) * bs + c];
    }
    o.m_dense_H.resize(dd, dd);
    for (int r = 0; r < dd; ++r) for (int c = 0; c < dd; ++c) o.m_dense_H(r, c) = dense_H[(size_t)r * dd + c];
    o.m_off_diag_H.resize(bs * nb, dd);
    for (int r = 0; r < bs * nb; ++r) for (int c = 0; c < dd; ++c) o.m_off_diag_H(r, c) = off_diag_H[(size_t)r * dd + c];
    o.m_block_diag_b.resize(bs * nb);
    for (int r = 0; r < bs * nb; ++r) o.m_block_diag_b(r) = block_diag_b[r];
    o.m_dense_b.resize(dd);
    for (int r = 0; r < dd; ++r) o.m_dense_b(r) = dense_b[r];
    o.m_x.resize(bs * nb + dd);
    OrcCost dummy{nullptr};
    o.SolveWithSchurComplementDenseOffDiag<OrcState>(bs * nb, dd, nullptr, dummy);
    for (int r = 0; r < bs * nb + dd; ++r) x[r] = o.m_x(r);
  }
  static bool CostIsSmallerThan(const double* left, const double* right, int n) {
    LMOptimizer<double> o;
    return o.CostIsSmallerThan(std::vector<double>(left, left + n), std::vector<double>(right, right + n));
  }
};
}  // namespace vis

extern "C" {

__attribute__((visibility("default")))
void ref_lmopt_schur_solve(int bs, int nb, int dd, const double* block_diag_H, const double* off_diag_H, const double* dense_H,
                           const double* block_diag_b, const double* dense_b, double* x) {
  vis::LMOptimizerTestHelper::SchurSolve(bs, nb, dd, block_diag_H, off_diag_H, dense_H, block_diag_b, dense_b, x);
}

__attribute__((visibility("default")))
int ref_lmopt_cost_is_smaller_than(const double* left, const double* right, int n) {
  return vis::LMOptimizerTestHelper::CostIsSmallerThan(left, right, n) ? 1 : 0;
}

// OptimizeJointly's optimizer calls (joint_optimization.cc:797-812, :916-940) on an oracle problem; st is updated in place.
// trace (may be NULL): per outer iteration 6 doubles: initial_cost, final_cost, lambda after the call, num_iterations_performed,
// cost-only passes (= LM attempts that reached the cost test), Jacobian passes.
__attribute__((visibility("default")))
double ref_lmopt_optimize_jointly(orc_problem* pb, orc_state* st, int max_iteration_count, double init_lambda, double* final_lambda,
                                  int32_t* performed_an_iteration, double* trace) {
  if (performed_an_iteration) *performed_an_iteration = 0;
  OrcCost cost_function{pb};
  OrcState opt_state(pb, st);
  LMOptimizer<double> optimizer;
  const int block_size = pb->eliminate_points ? 3 : 6;
  const int num_blocks = pb->eliminate_points ? pb->n_points : pb->n_images;
  optimizer.UseBlockDiagonalStructureForSchurComplement(block_size, num_blocks, /*sparse_storage_for_off_diag_H*/ false,
                                                        /*on_the_fly_block_processing*/ false, /*block_batch_size*/ 1024,
                                                        /*compute_schur_complement_with_cuda*/ false);
  double final_cost = -1;
  for (int iteration = 0; iteration < max_iteration_count; ++iteration) {
    const int c0 = cost_function.cost_passes, j0 = cost_function.jacobian_passes;
    OptimizationReport report = optimizer.Optimize(&opt_state, cost_function, /*max_iteration_count*/ 1, /*max_lm_attempts*/ 50,
                                                   init_lambda, /*init_lambda_factor*/ 0.00001, /*print_progress*/ false);
    final_cost = report.final_cost;
    init_lambda = optimizer.lambda();
    if (final_lambda) *final_lambda = optimizer.lambda();
    if (trace) {
      double* t = trace + 6 * iteration;
      t[0] = report.initial_cost; t[1] = report.final_cost; t[2] = optimizer.lambda(); t[3] = report.num_iterations_performed;
      t[4] = cost_function.cost_passes - c0; t[5] = cost_function.jacobian_passes - j0;
    }
    if (report.num_iterations_performed == 0) break;
    else if (performed_an_iteration) *performed_an_iteration = 1;
  }
  opt_state.store(st);
  return final_cost;
}

}  // extern "C"
