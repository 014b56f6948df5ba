"""The oracle against the REFERENCE'S ENTIRE CPU bundle-adjustment path, compiled whole from /root/reference into
oracle/_ref/libcalibref_ba.so (oracle/ref_ba_glue.cc, oracle/Makefile): APP/bundle_adjustment/joint_optimization.cc (JointOptimizationState,
the per-observation driver JointOptimizationCostFunction::Compute :240-593, OptimizeJointly :757-953), APP/models/central_generic.cc and
noncentral_generic.cc with their generated Jacobians, APP/models/central_grid.h, LV/lm_optimizer.h and its accumulator, APP/dataset.cc,
APP/bundle_adjustment/ba_state.cc -- on stand-ins for Eigen / Sophus only (oracle/ref_shim_lm; Eigen's LDLT::solve = the oracle's
term-by-term restatement).  Until round 5 the per-observation driver was the one piece of the hot path that was pinned only through
its pieces; here it runs itself.

What the tolerances mean.  Both sides evaluate the same expressions, but not the same machine code: g++ contracts a*b+c into FMAs in
the C++ reference code while the C oracle is built without contraction, and the stand-in Eigen sums small products in its own order.
That moves a projected pixel by ~5e-12 px; the finite-difference Jacobians (delta 1e-4 relative, joint_optimization.cc:357-372,
central_grid.h:187-245) divide such differences by ~1e-4, hence ~1e-10 of the largest entry of H for the central model and ~1e-8 for the
non-central one (two grids, a longer projection chain).  Over several LM iterations the iterative projection's finite stopping
tolerance adds ~1e-7 px of noise (see tests/test_oracle_vs_ref_outer_loop.py)."""
import dataclasses

import numpy as np
import pytest

from camera_calibration_amd import synthetic as syn
from camera_calibration_amd.problem import NONCENTRAL_GENERIC
from oracle import oracle as orc
from oracle import ref

pytestmark = pytest.mark.skipif(not ref.ba_available(), reason="oracle/_ref/libcalibref_ba.so not built (needs /root/reference)")

_proj = lambda cam, grid, pts: orc.project(cam, grid, pts)         # noqa: E731

CASES = {
    "1cam": (dict(num_cameras=1), {}),
    "rig": (dict(num_cameras=2), {}),
    "eliminate_points": (dict(num_cameras=1), dict(eliminate_points=True)),
    "rig_eliminate_points": (dict(num_cameras=2), dict(eliminate_points=True)),
    "localize_only": (dict(num_cameras=2), dict(localize_only=True)),
    "noncentral": (dict(num_cameras=1, model_type=NONCENTRAL_GENERIC), {}),
    "noncentral_rig": (dict(num_cameras=2, model_type=NONCENTRAL_GENERIC), {}),
}


def _problem(case, seed=0, num_points=30, num_poses=5):
    kw, flags = CASES[case]
    kw = dict(kw)
    pb, st, _ = syn.reference_test_problem(kw.pop("num_cameras"), orc.project, seed=seed, num_points=num_points, num_poses=num_poses, **kw)
    return dataclasses.replace(pb, **flags), st


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("case", list(CASES))
def test_normal_equations_of_the_references_own_driver(case):
    """One Jacobian pass: JointOptimizationCostFunction::Compute<true> into the reference's UpdateEquationAccumulator against
    orc_jacobian_pass -- every entry of H and b, every residual's cost, the warm-start cache."""
    pb, st = _problem(case)
    op = orc.OracleProblem(pb)
    sys_o = op.new_system()
    cost_o, cv_o, _ = op.jacobian_pass(st, sys_o)
    r = ref.ba_system(pb, st)
    S = r["system"]
    tol = 1e-6 if "noncentral" in case else 1e-8
    assert r["n_costs"] == pb.n_obs
    assert abs(r["cost"] - cost_o) <= 1e-12 * cost_o
    np.testing.assert_allclose(r["cost_vector"], cv_o, rtol=0, atol=1e-9)
    np.testing.assert_allclose(r["last_projection"], op.last_projection, rtol=0, atol=1e-9)
    worst = max(_rel(np.triu(S.block_diag_H), np.triu(sys_o.block_diag_H)), _rel(S.off_diag_H, sys_o.off_diag_H),
                _rel(np.triu(S.dense_H), np.triu(sys_o.dense_H)), _rel(S.block_diag_b, sys_o.block_diag_b), _rel(S.dense_b, sys_o.dense_b))
    print(case, "H / b worst deviation relative to the largest entry:", worst)
    assert worst <= tol
    # the structure (which blocks an observation touches): where one side holds an exact zero the other holds one too, or a
    # finite-difference quotient of two projections that differ in the last bits (observed: 3 entries of 1e5, each 5e-12 of the block's largest)
    for A, B in ((S.off_diag_H, sys_o.off_diag_H), (np.triu(S.dense_H), np.triu(sys_o.dense_H))):
        differs = (A == 0) != (B == 0)
        assert differs.sum() <= 8 and np.abs(np.where(differs, A - B, 0.0)).max() <= 1e-10 * np.abs(B).max()


def test_normal_equations_on_a_fine_grid_with_invalid_observations():
    """BASELINE configs[0]'s grid (16 x 12) with six imagesets, three points moved behind the camera: their residuals are invalid
    on both sides (AddInvalidResidual, joint_optimization.cc:334-342) and contribute nothing."""
    pb, st, _ = syn.baseline_config(1, _proj, n_imagesets=6)
    st = st.copy()
    st.points[[0, 5, 9]] += np.array([0.0, 0.0, -80.0])
    op = orc.OracleProblem(pb)
    sys_o = op.new_system()
    cost_o, cv_o, _ = op.jacobian_pass(st, sys_o)
    r = ref.ba_system(pb, st)
    invalid = cv_o < 0
    assert invalid.sum() >= 3
    np.testing.assert_array_equal(r["cost_vector"] < 0, invalid)
    assert abs(r["cost"] - cost_o) <= 1e-12 * cost_o
    S = r["system"]
    assert max(_rel(np.triu(S.dense_H), np.triu(sys_o.dense_H)), _rel(S.off_diag_H, sys_o.off_diag_H), _rel(S.dense_b, sys_o.dense_b)) <= 1e-8


def _trajectory(pb, st, iterations, stop_rule=False):
    """vis::OptimizeJointly (reference, whole) against orc_optimize_jointly, call by call as RunBundleAdjustment makes them."""
    a, b = st.copy(), st.copy()
    op = orc.OracleProblem(pb)
    lp = np.zeros((pb.n_obs, 2))
    lam_a = lam_b = -1.0
    last = float("inf")
    worst = dict(cost=0.0, lam=0.0, state=0.0)
    attempts = []
    for _ in range(iterations):
        ra = ref.ba_optimize_jointly(pb, a, lp, 1, lam_a)
        rb = op.optimize_jointly(b, 1, lam_b)
        assert ra["performed"] == rb["performed"]
        # lambda after a call = lambda before * 0.5 * 2^(rejected attempts) (lm_optimizer.h:943-977): equal lambdas = equal decisions
        worst["lam"] = max(worst["lam"], abs(ra["final_lambda"] - rb["final_lambda"]) / rb["final_lambda"])
        worst["cost"] = max(worst["cost"], abs(ra["cost"] - rb["cost"]) / max(abs(rb["cost"]), 1e-300))
        worst["state"] = max(worst["state"], np.abs(a.points - b.points).max(), np.abs(a.rig_tr_global - b.rig_tr_global).max(),
                             np.abs(a.camera_tr_rig - b.camera_tr_rig).max(), max(np.abs(ga - gb).max() for ga, gb in zip(a.grids, b.grids)))
        lam_a, lam_b = ra["final_lambda"], rb["final_lambda"]
        attempts.append(rb["lm_attempts"])
        if stop_rule and (not rb["performed"] or rb["cost"] >= last - 1e-4):
            break
        last = rb["cost"]
    return worst, attempts


@pytest.mark.parametrize("case", list(CASES))
def test_optimize_jointly_trajectory_of_the_references_own_code(case):
    pb, st = _problem(case, seed=7, num_points=40, num_poses=8)
    worst, attempts = _trajectory(pb, st, 5)
    print(case, worst, attempts)
    # lambda: the first one is 0.001 * mean(diag H) (1e-10 as H); afterwards it only halves / doubles -- so it stays at the first
    # call's deviation exactly as long as the decisions are the same
    assert worst["lam"] <= (1e-6 if "noncentral" in case else 1e-8)
    assert worst["cost"] <= 1e-4 and worst["state"] <= 1e-6


def test_optimize_jointly_with_rejected_updates_to_the_stopping_rule():
    """A noisy problem run until RunBundleAdjustment's rule fires: the late calls reject updates (several LM attempts per call); the
    reference's own code and the oracle take the same decisions all the way."""
    pb, st, _ = syn.baseline_config(1, _proj, n_imagesets=6, grid_wh=(8, 6))
    worst, attempts = _trajectory(pb, st, 40, stop_rule=True)
    print(worst, attempts)
    assert max(attempts) >= 2 and len(attempts) >= 5
    assert worst["lam"] <= 1e-8 and worst["cost"] <= 1e-6 and worst["state"] <= 1e-5


def test_the_references_whole_calibration_loop():
    """RunBundleAdjustment's text around the reference's own OptimizeJointly on the reference's own CentralGenericModel (everything
    reference code) against the same loop text around the oracle (libcalibref_f14.so): same number of OptimizeJointly calls, same
    converged state; and ChooseNiceCameraOrientation as central_generic.cc compiles it equals the piped-function build."""
    pb, st0, _ = syn.reference_test_problem(1, orc.project, seed=0, num_points=30, num_poses=5)
    st_all, calls_all, delta = ref.ba_run_bundle_adjustment(pb, st0, 40, 1e-4)
    st_orc, calls_orc, _ = ref.f1_run_bundle_adjustment(pb, st0, 40, 1e-4)
    assert calls_all == calls_orc >= 3 and delta == 1e-4
    for x, y in ((st_all.points, st_orc.points), (st_all.rig_tr_global, st_orc.rig_tr_global), (st_all.grids[0], st_orc.grids[0])):
        np.testing.assert_allclose(x, y, rtol=0, atol=1e-6)
    g = st0.grids[0]
    R = np.zeros(9); g2 = np.ascontiguousarray(g, dtype=np.float64).reshape(-1, 3).copy()
    ref.ba_lib().ref_f1_choose_nice_camera_orientation(ref._ip(ref._cam_params8(pb.cameras[0])), ref._dp(g2), ref._dp(R))
    R_f14, g_f14 = ref.f1_choose_nice_camera_orientation(pb.cameras[0], g)
    np.testing.assert_allclose(R.reshape(3, 3), R_f14, rtol=0, atol=1e-15)
    np.testing.assert_allclose(g2, g_f14, rtol=0, atol=1e-15)


def test_the_references_whole_calibration_loop_at_baseline_config_1():
    """BASELINE configs[0] (30 imagesets, 16 x 12 grid, 10 008 observations, D = 1 413 -- the reference's own CPU-runnable case) run to
    convergence by reference code only (RunBundleAdjustment + OptimizeJointly + CentralGenericModel + LMOptimizer, ~40 s on the
    stand-in Eigen) and by the same loop text around the oracle: the stopping rule fires after the same number of calls and the
    converged calibrations agree to 1e-7 after gauge alignment (observed 5e-13 ... 9e-10) -- the converged-parity statement of
    BASELINE.json's north_star between the oracle and the reference itself."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import converged_parity as cp
    pb, st0, _ = syn.baseline_config(1, _proj)
    assert pb.n_images == 30 and pb.dense_dof == 1413
    st_all, calls_all, _ = ref.ba_run_bundle_adjustment(pb, st0, 100, 1e-4)
    orc.set_num_threads(0)
    try:
        st_orc, calls_orc, _ = ref.f1_run_bundle_adjustment(pb, st0, 100, 1e-4)
    finally:
        orc.set_num_threads(1)
    print("OptimizeJointly calls until the stopping rule fires:", calls_all, calls_orc)
    assert calls_all == calls_orc >= 5
    dev = cp.gauge_aligned_deviation(pb, st_orc, st_all)
    print({k: v for k, v in dev.items() if k != "gauge"})
    for name in ("points_aligned_rel", "grids_aligned_abs", "pose_rotation_aligned_abs", "pose_translation_aligned_rel"):
        assert dev[name] <= 1e-7, (name, dev[name])
    op = orc.OracleProblem(pb)
    c_all, c_orc = op.cost_pass(st_all)[0], op.cost_pass(st_orc)[0]
    assert abs(c_all - c_orc) <= 1e-6 * c_all
