// oracle/_ref, part 3 -- TEST INFRASTRUCTURE ONLY.
//
// C entry points over the pieces of the reference's Levenberg-Marquardt machinery and grid models that the round-2 review
// found pinned only by the builder's own reading (LV = /root/reference/libvis/src/libvis,
// APP = /root/reference/applications/camera_calibration/src/camera_calibration):
//   LV/lm_optimizer_jtj_accumulator_base.h    AddResidualWithJacobian overloads (weighting, which block pairs)      rows B2, A5
//   LV/lm_optimizer_update_accumulator.h      UpdateEquationAccumulator: AddJacobianBB / BI / II, GetPartOfHAndB,
//                                             upper-only writes, on_diagonal b accumulation, residual cost vector    row  B3
//   APP/models/central_grid.h                 CentralGridModel::ProjectionJacobianWrtIntrinsics, SubtractDelta       rows M5, M6
//   APP/models/noncentral_generic.h           NoncentralGenericModel::ProjectionJacobianWrtIntrinsics, SubtractDelta row  N3
// The headers are #included from where they lie (nothing is copied) and compiled against the Eigen / libvis stand-ins of
// oracle/ref_shim.  What this file itself adds is (a) the call sequence of AccumulateModelJacobian
// (APP/bundle_adjustment/joint_optimization.cc:479-590; round 3 could not compile that file -- Sophus, cublasXt -- and since round 5 it
// compiles whole against the run-time-sized stand-ins, libcalibref_ba.so / ref_ba_glue.cc, where the reference's own call sequence runs), restated call by call
// below, and (b) a CentralGridModel subclass whose projection is the reference authors' stand-alone implementation
// (generic_models/src/central_generic.h, pinned in ref_generic_models.cc) evaluated on the grid the base class mutates in place.
#include <cuda_runtime.h>
#include <libvis/eigen.h>
#include <libvis/image.h>
#include <libvis/libvis.h>

#include "libvis/lm_optimizer_update_accumulator.h"

#include "camera_calibration/models/central_grid.h"
#include "camera_calibration/models/noncentral_generic.h"
#include "central_generic.h"      // generic_models/src: CentralGenericCamera<double>
#include "noncentral_generic.h"   // generic_models/src: NoncentralGenericCamera<double>

using namespace vis;

// Out-of-line members of NoncentralGenericModel that the class declaration (APP/models/noncentral_generic.h) leaves to
// noncentral_generic.cc, which cannot be built here (yaml-cpp, the real Eigen).  Only what the vtable and the two inline
// templates under test need: the projection is the reference authors' stand-alone implementation on the LIVE grids
// (ProjectionJacobianWrtIntrinsics perturbs m_point_grid / m_direction_grid in place, noncentral_generic.h:240-268).
namespace vis {
NoncentralGenericModel::NoncentralGenericModel(int grid_resolution_x, int grid_resolution_y, int calibration_min_x, int calibration_min_y,
                                               int calibration_max_x, int calibration_max_y, int width, int height)
    : CameraModel(width, height, calibration_min_x, calibration_min_y, calibration_max_x, calibration_max_y, CameraModel::Type::NoncentralGeneric) {
  m_point_grid.SetSize(grid_resolution_x, grid_resolution_y);
  m_direction_grid.SetSize(grid_resolution_x, grid_resolution_y);
}
CameraModel* NoncentralGenericModel::duplicate() { return nullptr; }
void NoncentralGenericModel::Scale(double) {}
bool NoncentralGenericModel::ProjectWithInitialEstimate(const Vec3d& local_point, Vec2d* result) const {
  NoncentralGenericCamera<double> cam(width(), height(), calibration_min_x(), calibration_min_y(), calibration_max_x(),
                                      calibration_max_y(), m_point_grid.width(), m_point_grid.height());
  for (int y = 0; y < (int)m_point_grid.height(); ++y)
    for (int x = 0; x < (int)m_point_grid.width(); ++x) {
      cam.direction_grid_value(x, y) = m_direction_grid(x, y);
      cam.point_grid_value(x, y) = m_point_grid(x, y);
    }
  Eigen::Vector2d px(result->x(), result->y());
  const bool ok = cam.ProjectWithInitialEstimate(local_point, &px);
  *result = px;
  return ok;
}
}  // namespace vis

namespace {

struct RefAccum {
  int block_dof, block_size, dense_dof;
  Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic> dense_H, off_diag_H;
  std::vector<Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic>> block_diag_H;
  Eigen::Matrix<double, Eigen::Dynamic, 1> dense_b, block_diag_b;
  std::vector<double> cost_vector;
  UpdateEquationAccumulator<double>* acc = nullptr;
};

template <class M>
M load_rowmajor(const double* src) {
  M m;
  for (int r = 0; r < M::RowsAtCompileTime; ++r)
    for (int c = 0; c < M::ColsAtCompileTime; ++c) m(r, c) = src[r * M::ColsAtCompileTime + c];
  return m;
}

// AccumulateModelJacobian's dispatch (joint_optimization.cc:479-590), one call per observation.  K = IntrinsicsJacobianSize.
template <int K>
void add_observation(UpdateEquationAccumulator<double>* accumulator, bool rig_in_state, bool eliminate_points, bool localize_only,
                     const Vec2d& residual, usize pose_jac_index, const Matrix<double, 2, 6>& pose_jacobian, usize rig_jac_index,
                     const Matrix<double, 2, 6>& rig_jacobian, usize point_jac_index, const Matrix<double, 2, 3>& point_jacobian,
                     const int* grid_idx, const double* grid_jac) {
  Matrix<int, K, 1> grid_update_indices;
  Matrix<double, 2, K, Eigen::RowMajor> pixel_wrt_grid_updates;
  if (!localize_only) {
    for (int i = 0; i < K; ++i) {
      grid_update_indices(i) = grid_idx[i];
      pixel_wrt_grid_updates(0, i) = grid_jac[i];
      pixel_wrt_grid_updates(1, i) = grid_jac[K + i];
    }
  }
  if (localize_only) {
    if (eliminate_points) {
      if (rig_in_state) {
        accumulator->AddResidualWithJacobian(residual, point_jac_index, point_jacobian, pose_jac_index, pose_jacobian, rig_jac_index,
                                             rig_jacobian, true, true, true, HuberLoss<double>(1.0));   // :497-507
      } else {
        accumulator->AddResidualWithJacobian(residual, point_jac_index, point_jacobian, pose_jac_index, pose_jacobian, true, true,
                                             HuberLoss<double>(1.0));                                   // :509-516
      }
    } else {
      if (rig_in_state) {
        accumulator->AddResidualWithJacobian(residual, pose_jac_index, pose_jacobian, rig_jac_index, rig_jacobian, point_jac_index,
                                             point_jacobian, true, true, true, HuberLoss<double>(1.0));  // :520-530
      } else {
        accumulator->AddResidualWithJacobian(residual, pose_jac_index, pose_jacobian, point_jac_index, point_jacobian, true, true,
                                             HuberLoss<double>(1.0));                                   // :532-539
      }
    }
  } else {
    if (eliminate_points) {
      if (rig_in_state) {
        accumulator->AddResidualWithJacobian(residual, point_jac_index, point_jacobian, pose_jac_index, pose_jacobian, rig_jac_index,
                                             rig_jacobian, grid_update_indices, pixel_wrt_grid_updates, true, true, true,
                                             HuberLoss<double>(1.0));                                   // :545-557
      } else {
        accumulator->AddResidualWithJacobian(residual, point_jac_index, point_jacobian, pose_jac_index, pose_jacobian,
                                             grid_update_indices, pixel_wrt_grid_updates, true, true, HuberLoss<double>(1.0));  // :559-568
      }
    } else {
      if (rig_in_state) {
        accumulator->AddResidualWithJacobian(residual, pose_jac_index, pose_jacobian, rig_jac_index, rig_jacobian, point_jac_index,
                                             point_jacobian, grid_update_indices, pixel_wrt_grid_updates, true, true, true,
                                             HuberLoss<double>(1.0));                                   // :572-584
      } else {
        accumulator->AddResidualWithJacobian(residual, pose_jac_index, pose_jacobian, point_jac_index, point_jacobian,
                                             grid_update_indices, pixel_wrt_grid_updates, true, true, HuberLoss<double>(1.0));  // :586-595
      }
    }
  }
}

// CentralGridModel with the stand-alone reference projection on the LIVE grid (ProjectionJacobianWrtIntrinsics perturbs
// m_grid in place through grid().data(), central_grid.h:219-230)
class RefCentralGrid : public CentralGridModel<RefCentralGrid> {
 public:
  RefCentralGrid(int gw, int gh, int min_x, int min_y, int max_x, int max_y, int width, int height)
      : CentralGridModel<RefCentralGrid>(CameraModel::Type::CentralGeneric, gw, gh, min_x, min_y, max_x, max_y, width, height) {}
  CameraModel* duplicate() override { return new RefCentralGrid(*this); }
  bool ProjectDirectionWithInitialEstimate(const Vec3d& local_direction, Vec2d* result) const {
    CentralGenericCamera<double> cam(width(), height(), calibration_min_x(), calibration_min_y(), calibration_max_x(),
                                     calibration_max_y(), grid().width(), grid().height());
    for (int y = 0; y < (int)grid().height(); ++y)
      for (int x = 0; x < (int)grid().width(); ++x) cam.grid_value(x, y) = grid()(x, y);
    Eigen::Vector2d px(result->x(), result->y());
    const bool ok = cam.ProjectWithInitialEstimate(local_direction, &px);
    *result = px;
    return ok;
  }
  using CentralGridModel<RefCentralGrid>::Unproject;
  bool Unproject(double x, double y, Vec3d* result) const override {
    if (!IsInCalibratedArea(x, y)) return false;
    const Vec2d gp = PixelCornerConvToGridPoint(x, y);
    *result = UnprojectFromGrid(gp.x(), gp.y());
    return true;
  }
};

RefCentralGrid* make_grid_model(const int* p8, const double* grid) {
  // p8: width height min_x min_y max_x max_y gw gh
  auto* m = new RefCentralGrid(p8[6], p8[7], p8[2], p8[3], p8[4], p8[5], p8[0], p8[1]);
  for (int y = 0; y < p8[7]; ++y)
    for (int x = 0; x < p8[6]; ++x) {
      const double* g = grid + 3 * (x + (size_t)y * p8[6]);
      m->grid()(x, y) = Vec3d(g[0], g[1], g[2]);
    }
  return m;
}

// (the stand-in Eigen has no dynamic MatrixBase: the update vector is a fixed-size Matrix, for the grid sizes the tests use)
template <int N>
void subtract_delta_n(RefCentralGrid* m, const double* delta) {
  Matrix<double, N, 1> d;
  for (int i = 0; i < N; ++i) d(i) = delta[i];
  m->SubtractDelta(d);
}
NoncentralGenericModel* make_noncentral_model(const int* p8, const double* grids) {
  auto* m = new NoncentralGenericModel(p8[6], p8[7], p8[2], p8[3], p8[4], p8[5], p8[0], p8[1]);
  const size_t G = (size_t)p8[6] * p8[7];
  for (int y = 0; y < p8[7]; ++y)
    for (int x = 0; x < p8[6]; ++x) {
      const double* d = grids + 3 * (x + (size_t)y * p8[6]);
      const double* o = grids + 3 * G + 3 * (x + (size_t)y * p8[6]);
      m->direction_grid()(x, y) = Vec3d(d[0], d[1], d[2]);
      m->point_grid()(x, y) = Vec3d(o[0], o[1], o[2]);
    }
  return m;
}
template <int N>
void noncentral_subtract_delta_n(NoncentralGenericModel* m, const double* delta) {
  Matrix<double, N, 1> d;
  for (int i = 0; i < N; ++i) d(i) = delta[i];
  m->SubtractDelta(d);
}
}  // namespace

extern "C" {

void* ref_accum_create(int block_dof, int block_size, int dense_dof) {
  auto* a = new RefAccum();
  a->block_dof = block_dof; a->block_size = block_size; a->dense_dof = dense_dof;
  a->dense_H.resize(dense_dof, dense_dof);
  a->off_diag_H.resize(block_dof, dense_dof);
  a->block_diag_H.resize(block_dof / block_size);
  for (auto& m : a->block_diag_H) m.resize(block_size, block_size);
  a->dense_b.resize(dense_dof);
  a->block_diag_b.resize(block_dof);
  // dense off-diagonal storage, no on-the-fly block processing: what LMOptimizer::Optimize sets up for SchurMode::Dense
  // (LV/lm_optimizer.h:706-720)
  a->acc = new UpdateEquationAccumulator<double>(block_dof, &a->dense_H, &a->off_diag_H, nullptr, &a->block_diag_H, &a->dense_b,
                                                 &a->block_diag_b, &a->cost_vector, 0, 0.0);
  return a;
}
void ref_accum_destroy(void* h) { auto* a = static_cast<RefAccum*>(h); delete a->acc; delete a; }

// mode: bit 0 rig_in_state, bit 1 eliminate_points, bit 2 localize_only.  kg: 0 (localize_only), 32 or 80.
// Jacobians row-major: pose / rig 2 x 6, point 2 x 3, grid 2 x kg.
void ref_accum_add(void* h, int mode, int kg, const double* res2, int pose_idx, const double* pose_jac, int rig_idx,
                   const double* rig_jac, int point_idx, const double* point_jac, const int* grid_idx, const double* grid_jac) {
  auto* a = static_cast<RefAccum*>(h);
  const Vec2d residual(res2[0], res2[1]);
  const auto pj = load_rowmajor<Matrix<double, 2, 6>>(pose_jac);
  const auto rj = load_rowmajor<Matrix<double, 2, 6>>(rig_jac);
  const auto tj = load_rowmajor<Matrix<double, 2, 3>>(point_jac);
  const bool rig = mode & 1, elim = mode & 2, loc = mode & 4;
  if (kg == 80) add_observation<80>(a->acc, rig, elim, loc, residual, pose_idx, pj, rig_idx, rj, point_idx, tj, grid_idx, grid_jac);
  else add_observation<32>(a->acc, rig, elim, loc, residual, pose_idx, pj, rig_idx, rj, point_idx, tj, grid_idx, grid_jac);
}
// residual kept without Jacobian (joint_optimization.cc:476-478) / invalid residual (:334-342)
void ref_accum_add_residual(void* h, const double* res2) {
  static_cast<RefAccum*>(h)->acc->AddResidual(Vec2d(res2[0], res2[1]), HuberLoss<double>(1.0));
}
void ref_accum_add_invalid(void* h) { static_cast<RefAccum*>(h)->acc->AddInvalidResidual(); }

// row-major copies; block_diag_H: [block][bs][bs]
void ref_accum_get(void* h, double* block_diag_H, double* off_diag_H, double* dense_H, double* dense_b, double* block_diag_b,
                   double* cost, double* cost_vector, int* n_costs) {
  auto* a = static_cast<RefAccum*>(h);
  const int bs = a->block_size, nb = a->block_dof / bs, dd = a->dense_dof;
  for (int b = 0; b < nb; ++b)
    for (int r = 0; r < bs; ++r)
      for (int c = 0; c < bs; ++c) block_diag_H[((size_t)b * bs + r) * bs + c] = a->block_diag_H[b](r, c);
  for (int r = 0; r < a->block_dof; ++r)
    for (int c = 0; c < dd; ++c) off_diag_H[(size_t)r * dd + c] = a->off_diag_H(r, c);
  for (int r = 0; r < dd; ++r) {
    for (int c = 0; c < dd; ++c) dense_H[(size_t)r * dd + c] = a->dense_H(r, c);
    dense_b[r] = a->dense_b(r);
  }
  for (int r = 0; r < a->block_dof; ++r) block_diag_b[r] = a->block_diag_b(r);
  *cost = a->acc->cost();
  *n_costs = (int)a->cost_vector.size();
  for (size_t i = 0; i < a->cost_vector.size(); ++i) cost_vector[i] = a->cost_vector[i];
}

// CentralGridModel::ProjectionJacobianWrtIntrinsics (central_grid.h:187-245).  tangents: 6 doubles per grid point (t1, t2).
// Returns the function's bool; indices32 / jac64 (2 x 32 row-major) are filled on success.
int ref_central_grid_projection_jacobian(const int* p8, const double* grid, const double* tangents, const double* local_point,
                                         const double* pixel, double delta, int* indices32, double* jac64) {
  RefCentralGrid* m = make_grid_model(p8, grid);
  Image<vis::DirectionTangents> tang(p8[6], p8[7]);
  for (int i = 0; i < p8[6] * p8[7]; ++i) {
    tang.data()[i].t1 = Vec3d(tangents[6 * i], tangents[6 * i + 1], tangents[6 * i + 2]);
    tang.data()[i].t2 = Vec3d(tangents[6 * i + 3], tangents[6 * i + 4], tangents[6 * i + 5]);
  }
  Matrix<int, 32, 1> idx;
  Matrix<double, 2, 32, Eigen::RowMajor> J;
  const bool ok = m->ProjectionJacobianWrtIntrinsics(Vec3d(local_point[0], local_point[1], local_point[2]), Vec2d(pixel[0], pixel[1]),
                                                     tang, delta, &idx, &J);
  if (ok)
    for (int i = 0; i < 32; ++i) { indices32[i] = idx(i); jac64[i] = J(0, i); jac64[32 + i] = J(1, i); }
  delete m;
  return ok;
}
// CentralGridModel::SubtractDelta (central_grid.h:168-185): grid (3G, in place) -= delta (2G local updates)
int ref_central_grid_subtract_delta(const int* p8, double* grid, const double* delta) {
  RefCentralGrid* m = make_grid_model(p8, grid);
  const int G = p8[6] * p8[7];
  switch (2 * G) {
    case 442: subtract_delta_n<442>(m, delta); break;   // 17 x 13, the reference's own calibrated camera
    case 160: subtract_delta_n<160>(m, delta); break;   // 10 x 8
    case 96: subtract_delta_n<96>(m, delta); break;     // 8 x 6
    case 50: subtract_delta_n<50>(m, delta); break;     // 5 x 5 (the gtest-sized fixtures)
    default: delete m; return 0;
  }
  for (int y = 0; y < p8[7]; ++y)
    for (int x = 0; x < p8[6]; ++x)
      for (int k = 0; k < 3; ++k) grid[3 * (x + (size_t)y * p8[6]) + k] = m->grid()(x, y)(k);
  delete m;
  return 1;
}

// NoncentralGenericModel::ProjectionJacobianWrtIntrinsics (noncentral_generic.h:224-283).  grids: direction grid 3G, then point grid
// 3G; tangents: 6 doubles per grid point.  indices80 / jac160 (2 x 80 row-major) are filled on success.
int ref_noncentral_grid_projection_jacobian(const int* p8, const double* grids, const double* tangents, const double* local_point,
                                            const double* pixel, double delta, int* indices80, double* jac160) {
  NoncentralGenericModel* m = make_noncentral_model(p8, grids);
  Image<vis::LineTangents> tang(p8[6], p8[7]);
  for (int i = 0; i < p8[6] * p8[7]; ++i) {
    tang.data()[i].t1 = Vec3d(tangents[6 * i], tangents[6 * i + 1], tangents[6 * i + 2]);
    tang.data()[i].t2 = Vec3d(tangents[6 * i + 3], tangents[6 * i + 4], tangents[6 * i + 5]);
  }
  Matrix<int, 80, 1> idx;
  Matrix<double, 2, 80, Eigen::RowMajor> J;
  const bool ok = m->ProjectionJacobianWrtIntrinsics(Vec3d(local_point[0], local_point[1], local_point[2]), Vec2d(pixel[0], pixel[1]),
                                                     tang, delta, &idx, &J);
  if (ok)
    for (int i = 0; i < 80; ++i) { indices80[i] = idx(i); jac160[i] = J(0, i); jac160[80 + i] = J(1, i); }
  delete m;
  return ok;
}
// NoncentralGenericModel::SubtractDelta (noncentral_generic.h:195-222): grids (direction 3G, point 3G; in place) -= delta (5G)
int ref_noncentral_grid_subtract_delta(const int* p8, double* grids, const double* delta) {
  NoncentralGenericModel* m = make_noncentral_model(p8, grids);
  const int G = p8[6] * p8[7];
  switch (5 * G) {
    case 320: noncentral_subtract_delta_n<320>(m, delta); break;   // 8 x 8, the camera of the reference's self-test
    case 125: noncentral_subtract_delta_n<125>(m, delta); break;   // 5 x 5
    default: delete m; return 0;
  }
  for (int y = 0; y < p8[7]; ++y)
    for (int x = 0; x < p8[6]; ++x)
      for (int k = 0; k < 3; ++k) {
        grids[3 * (x + (size_t)y * p8[6]) + k] = m->direction_grid()(x, y)(k);
        grids[3 * G + 3 * (x + (size_t)y * p8[6]) + k] = m->point_grid()(x, y)(k);
      }
  delete m;
  return 1;
}

}  // extern "C"
