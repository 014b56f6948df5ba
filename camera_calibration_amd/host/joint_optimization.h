// Drop-in replacement header for APP/bundle_adjustment/joint_optimization.h:45-70 (reference tree):
// same enum, same free function, same argument meaning.  The implementation
// (joint_optimization_hip.cc) marshals Dataset / BAState into the packed arrays of include/cba.h and
// runs every outer iteration as one cba_step on the MI355X.
#pragma once
#include "dataset.h"

namespace vis {

enum class SchurMode {
  Dense = 0,
  DenseCUDA,        // accepted for source compatibility; every mode runs the HIP dense Schur path
  DenseOnTheFly,
  Sparse,
  SparseOnTheFly
};

/// Returns the final cost (-1 if no iteration ran).  See the reference header for the argument
/// documentation; debug_fix_* are not supported by the HIP backend (they force the reference onto its
/// slow unstructured path) and make the call fail loudly.
double OptimizeJointly(Dataset& dataset, BAState* state, int max_iteration_count, double init_lambda,
                       double numerical_diff_delta, double regularization_weight, bool localize_only,
                       bool eliminate_points, SchurMode schur_mode, double* final_lambda,
                       bool* performed_an_iteration = nullptr, bool debug_verify_cost = false,
                       bool debug_fix_points = false, bool debug_fix_poses = false, bool debug_fix_rig_poses = false,
                       bool debug_fix_intrinsics = false, bool print_progress = true);

/// HIP device ordinal used by OptimizeJointly and the CameraModel calls (default 0 / env CBA_DEVICE).
void SetHipDevice(int device);

}  // namespace vis
