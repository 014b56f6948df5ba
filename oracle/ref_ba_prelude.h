// oracle/_ref, part 6 (prelude of libcalibref_ba.so) -- TEST INFRASTRUCTURE ONLY.
// The reference's real generic camera models next to ref_f14_prelude.h, and a counting wrapper around the reference's real
// vis::OptimizeJointly: oracle/Makefile pipes RunBundleAdjustment (APP/calibration.cc:187-304) into the compiler with the name
// `OptimizeJointly` mapped to the wrapper, so that the tests can ask how many calls the loop made.
#pragma once
#include "ref_f14_prelude.h"
#include "camera_calibration/models/central_generic.h"
#include "camera_calibration/models/noncentral_generic.h"
namespace vis {
extern int g_ref_ba_optimize_calls;
extern double g_ref_ba_fd_delta_seen;
inline double RefCountedOptimizeJointly(Dataset& dataset, BAState* state, int max_iteration_count, double init_lambda, double numerical_diff_delta,
                                        double regularization_weight, bool localize_only, bool eliminate_points, SchurMode schur_mode,
                                        double* final_lambda) {
  ++g_ref_ba_optimize_calls;
  g_ref_ba_fd_delta_seen = numerical_diff_delta;
  return OptimizeJointly(dataset, state, max_iteration_count, init_lambda, numerical_diff_delta, regularization_weight, localize_only,
                         eliminate_points, schur_mode, final_lambda, nullptr, false, false, false, false, false, /*print_progress*/ false);
}
}
