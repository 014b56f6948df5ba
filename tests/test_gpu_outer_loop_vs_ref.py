"""SURVEY 8f rows F1 / F4 on the GPU against the REFERENCE'S OWN outer-loop and report code (oracle/_ref/libcalibref_f14.so, built in
the development container from /root/reference and shipped prebuilt; see tests/test_oracle_vs_ref_outer_loop.py for what it holds).

* RunBundleAdjustment: camera_calibration_amd.calibration.run_bundle_adjustment (HIP engine, device-resident state, orientation
  beautification through cba_unproject) next to the reference's loop text (APP/calibration.cc:187-304) driving the CPU oracle's
  OptimizeJointly, from the same perturbed start of BASELINE configs[0], threshold 1e-4, at most 100 iterations: the number of
  OptimizeJointly calls until the reference's stopping rule fires must be identical, the converged state equal to the converged-parity
  tolerance of tests/test_gpu_converged_parity.py (1e-7 after gauge alignment).
* DeleteOutlierFeatures / ComputeAllReprojectionErrors with the projections on the GPU (cba_project): keep masks and image_used
  identical, errors 1e-9 px."""
import os
import sys

import numpy as np
import pytest

from camera_calibration_amd import calibration as cal
from camera_calibration_amd import report as rp
from camera_calibration_amd import synthetic as syn
from oracle import oracle as orc
from oracle import ref
from parity_record import check, check_equal

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import converged_parity as cp  # noqa: E402

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref.f14_available(), reason="oracle/_ref/libcalibref_f14.so not shipped with this snapshot")]


def test_run_bundle_adjustment_on_the_engine_stops_where_the_references_loop_stops():
    pb, st0, _ = syn.baseline_config(1, lambda cam, grid, pts: orc.project(cam, grid, pts))
    orc.set_num_threads(0)
    try:
        st_ref, calls, delta = ref.f1_run_bundle_adjustment(pb, st0, 100, 1e-4)
    finally:
        orc.set_num_threads(1)
    st_eng, costs = cal.run_bundle_adjustment(pb, st0, 100, 1e-4)
    case = "RunBundleAdjustment (reference loop text + CPU oracle) vs run_bundle_adjustment (HIP engine), BASELINE configs[0]"
    print(case, calls, len(costs), costs[-1])
    assert calls >= 5 and delta == pb.fd_delta
    check_equal(case, "OptimizeJointly calls until cost >= last_cost - 1e-4 (engine - reference loop)", abs(len(costs) - calls))
    op = orc.OracleProblem(pb)
    c_ref, c_eng = op.cost_pass(st_ref)[0], op.cost_pass(st_eng)[0]
    check(case, "cost of the converged state rel (evaluated by the oracle on both)", abs(c_eng - c_ref) / c_ref, 1e-6,
          note="observed 1.1e-8.  Both costs come from a fresh cost pass (projections started at the centre of the calibrated area, not at "
               "the loop's warm-start cache); the iterative projection stops at a finite tolerance, which alone moves such a cost by up "
               "to 7e-8 relative between states that agree to 6e-16 (measured, tests/test_oracle_vs_ref_outer_loop.py)")
    dev = cp.gauge_aligned_deviation(pb, st_eng, st_ref)
    for name in ("points_aligned_rel", "grids_aligned_abs", "pose_rotation_aligned_abs", "pose_translation_aligned_rel"):
        check(case, f"converged state after gauge alignment: {name}", dev[name], 1e-7)
    if ref.ba_available():
        # the same loop run by reference code ONLY (libcalibref_ba.so: RunBundleAdjustment + OptimizeJointly + CentralGenericModel +
        # LMOptimizer compiled from the reference's sources; ~40 s of host time)
        st_all, calls_all, _ = ref.ba_run_bundle_adjustment(pb, st0, 100, 1e-4)
        case2 = "RunBundleAdjustment run by reference code only (compiled from its sources) vs run_bundle_adjustment (HIP engine), BASELINE configs[0]"
        check_equal(case2, "OptimizeJointly calls until cost >= last_cost - 1e-4 (engine - reference)", abs(len(costs) - calls_all))
        dev = cp.gauge_aligned_deviation(pb, st_eng, st_all)
        for name in ("points_aligned_rel", "grids_aligned_abs", "pose_rotation_aligned_abs", "pose_translation_aligned_rel"):
            check(case2, f"converged state after gauge alignment: {name}", dev[name], 1e-7)
        c_all = op.cost_pass(st_all)[0]
        check(case2, "cost of the converged state rel (evaluated by the oracle on both)", abs(c_eng - c_all) / c_all, 1e-6,
              note="fresh cost passes; see the note of the row above")
    # both sides leave the loop in the beautified orientation: image centre along +z
    cam = pb.cameras[0]
    for st in (st_ref, st_eng):
        centre = orc.unproject(cam, st.grids[0], np.array([[0.5 * cam.width, 0.5 * cam.height]]))[0][0, :3]
        np.testing.assert_allclose(centre, [0, 0, 1], atol=1e-9)


@pytest.mark.skipif(not ref.ba_available(), reason="oracle/_ref/libcalibref_ba.so not shipped with this snapshot")
@pytest.mark.parametrize("case", ["1cam", "rig", "noncentral", "eliminate_points"])
def test_engine_against_the_references_own_optimize_jointly(case):
    """The HIP engine next to the reference's ENTIRE CPU path compiled from the reference's sources (libcalibref_ba.so:
    joint_optimization.cc, the generic models, lm_optimizer.h; see tests/test_oracle_vs_ref_whole_path.py) -- no restatement between
    them except Eigen's LDLT: five OptimizeJointly(1) calls from the same start, lambda carried as RunBundleAdjustment carries it."""
    import dataclasses
    from camera_calibration_amd import engine as eng
    from camera_calibration_amd.problem import NONCENTRAL_GENERIC
    kw = dict(model_type=NONCENTRAL_GENERIC) if case == "noncentral" else {}
    pb, st0, _ = syn.reference_test_problem(2 if case == "rig" else 1, orc.project, seed=7, num_points=40, num_poses=8, **kw)
    if case == "eliminate_points":
        pb = dataclasses.replace(pb, eliminate_points=True)
    a = st0.copy()
    lp = np.zeros((pb.n_obs, 2))
    e = eng.Engine(pb)
    name = f"HIP engine vs the reference's own OptimizeJointly (compiled from its sources), {case}"
    try:
        e.set_state(st0)
        lam_a = lam_e = -1.0
        worst = dict(cost=0.0, lam=0.0)
        for _ in range(5):
            ra = ref.ba_optimize_jointly(pb, a, lp, 1, lam_a)
            re = e.step(lam_e)
            assert bool(re.accepted) == ra["performed"]
            worst["lam"] = max(worst["lam"], abs(re.final_lambda - ra["final_lambda"]) / ra["final_lambda"])
            worst["cost"] = max(worst["cost"], abs(re.final_cost - ra["cost"]) / ra["cost"])
            lam_a, lam_e = ra["final_lambda"], re.final_lambda
        b = e.get_state(st0)
    finally:
        e.close()
    state = max(np.abs(a.points - b.points).max(), np.abs(a.rig_tr_global - b.rig_tr_global).max(),
                max(np.abs(x - y).max() for x, y in zip(a.grids, b.grids)))
    print(name, worst, state)
    check(name, "lambda rel, max over five calls (equal lambdas = equal accept / reject decisions)", worst["lam"], 1e-6 if case == "noncentral" else 1e-8,
          note="the first lambda is 0.001 * mean(diag H); H carries the finite-difference noise of tests/test_oracle_vs_ref_whole_path.py")
    check(name, "cost rel, max over five calls", worst["cost"], 1e-4,
          note="the late costs are ~1e-5 absolute on these noiseless problems, where 1e-11 absolute is 1e-6 relative")
    check(name, "state after five calls, raw max abs", state, 1e-6, note="no gauge alignment; both sides take the same steps")


@pytest.mark.skipif(not ref.ba_available(), reason="oracle/_ref/libcalibref_ba.so not shipped with this snapshot")
@pytest.mark.parametrize("case", ["central 40x30 grid", "rig 20x16 grids", "non-central 16x12 grid"])
def test_normal_equations_on_the_gpu_against_the_references_own_driver(case):
    """One Jacobian pass of the HIP engine (k_base_project, k_fd_*, k_assemble, k_accumulate*) against ONE
    JointOptimizationCostFunction::Compute<true> of the reference into the reference's own UpdateEquationAccumulator, compiled from the
    reference's sources (ref_ba_system) -- every entry of H and b, every residual's cost, the warm-start cache; no restatement between
    the two sides.  Sizes the stand-in Eigen finishes in seconds: BASELINE-style problems with 12 imagesets."""
    from camera_calibration_amd import engine as eng
    cfg, grid = {"central 40x30 grid": (2, (40, 30)), "rig 20x16 grids": (3, (20, 16)), "non-central 16x12 grid": (4, (16, 12))}[case]
    pb, st, _ = syn.baseline_config(cfg, lambda cam, g, pts: orc.project(cam, g, pts), n_imagesets=12, grid_wh=grid)
    r = ref.ba_system(pb, st)
    S = r["system"]
    e = eng.Engine(pb)
    try:
        e.set_state(st)
        cost = e.debug_accumulate()
        vec = e.dump(eng.DUMP_COST_VECTOR)
        got = dict(block_diag_H=e.dump(eng.DUMP_BLOCK_DIAG_H), block_diag_b=e.dump(eng.DUMP_BLOCK_DIAG_B), off_diag_H=e.dump(eng.DUMP_OFF_DIAG_H),
                   dense_H=np.triu(e.dump(eng.DUMP_DENSE_H)), dense_b=e.dump(eng.DUMP_DENSE_B))
        lp = e.get_last_projection()
    finally:
        e.close()
    name = f"Jacobian pass on the GPU vs the reference's own driver (compiled from its sources), {case}: {pb.n_obs} observations, D = {pb.dense_dof}"
    want = dict(block_diag_H=np.stack([np.triu(b) for b in S.block_diag_H]), block_diag_b=S.block_diag_b, off_diag_H=S.off_diag_H,
                dense_H=np.triu(S.dense_H), dense_b=S.dense_b)
    got["block_diag_H"] = np.stack([np.triu(b) for b in got["block_diag_H"]])
    assert r["n_costs"] == pb.n_obs
    check_equal(name, "residuals valid on one side and invalid on the other", int(((vec < 0) != (r["cost_vector"] < 0)).sum()))
    check(name, "cost rel", abs(cost - r["cost"]) / r["cost"], 1e-11)
    check(name, "per-residual cost, max abs", float(np.abs(vec - r["cost_vector"]).max()), 1e-10, note="observed 2e-12")
    check(name, "warm-start cache (last_projection), max abs px", float(np.abs(lp - r["last_projection"]).max()), 1e-10, note="observed 2e-12")
    tol = 1e-9                                # observed 7e-14 ... 8e-12 (the GPU contracts to FMAs as g++ does in the reference code)
    for k in want:
        dev = float(np.abs(got[k] - want[k]).max() / np.abs(want[k]).max())
        check(name, f"{k}: max deviation relative to the largest entry", dev, tol,
              note="finite-difference Jacobians of projections that differ in the last bits (FMA contraction, summation order): "
                   "see tests/test_oracle_vs_ref_whole_path.py")


def test_outlier_removal_and_reprojection_statistics_on_the_gpu_are_the_references():
    pb, st, _ = syn.reference_test_problem(2, orc.project, seed=3, num_points=60, num_poses=12)
    case = "DeleteOutlierFeatures / ComputeAllReprojectionErrors, projections on the GPU, vs the reference's functions"
    for c in range(pb.n_cameras):
        r = ref.f4_compute_all_reprojection_errors(c, pb, st)
        h = rp.compute_all_reprojection_errors(c, pb, st)
        check_equal(case, f"camera {c}: reprojection_error_count", abs(h["count"] - r["count"]))
        check(case, f"camera {c}: reprojection errors, max abs (px)", float(np.abs(h["errors"] - r["errors"]).max()), 1e-9,
              note="the iterative projection stops at a finite tolerance; pixels of order 1e2")
        np.testing.assert_array_equal(h["features"], r["features"])
        h_ref = ref.f4_reprojection_error_histogram(50, 2.0, r["errors"])
        check_equal(case, f"camera {c}: histogram bins that differ", int((rp.reprojection_error_histogram(50, 2.0, h["errors"]) != h_ref).sum()))
        med = rp.reprojection_error_summary(h)["reprojection_error_median"]
        check(case, f"camera {c}: reprojection_error_median rel", abs(med - ref.f4_reprojection_error_median(r["errors"])) / med, 1e-10)
        for factor in (0.5, 1.5):
            k_ref, u_ref = ref.f1_delete_outlier_features(c, pb, st, factor)
            k, u, _ = rp.delete_outlier_features(c, pb, st, factor)
            check_equal(case, f"camera {c}, factor {factor}: keep-mask entries that differ", int((k != k_ref).sum()))
            check_equal(case, f"camera {c}, factor {factor}: image_used entries that differ", int((u != u_ref).sum()))
            assert (~k_ref).sum() >= 1


def test_calibrate_refinement_stage_on_the_gpu_follows_the_oracle_run_stage():
    """camera_calibration_amd.calibration.calibrate_refinement_stage (APP/calibration.cc:1030-1142: pyramid level, ResampleModel, outlier stage, main
    bundle adjustment, ScaleToMetric) with its GPU defaults -- HIP engine, cba_fit_grid_to_directions, cba_unproject, cba_project -- against the
    same orchestration on the CPU oracle (which tests/test_oracle_vs_ref_calibrate_stage.py compares with the reference's own stage): every
    RunBundleAdjustment run stops after the same number of iterations, the outlier masks are the same, the costs agree."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_oracle_vs_ref_calibrate_stage as stage
    from camera_calibration_amd import calibration_io as cio
    pb, st0, _ = syn.baseline_config(1, lambda cam, g, pts: orc.project(cam, g, pts), n_imagesets=8, grid_wh=(8, 6), lattice_xy=(8, 9))
    positions = syn.pattern_positions(8, 9)
    ds = cio.DatasetData(image_sizes=[(640, 480)], known_geometries=[cio.KnownGeometry(0.01188, {i: tuple(int(v) for v in p) for i, p in enumerate(positions)})])
    mapping = {i: i for i in range(pb.n_points)}
    g = cal.calibrate_refinement_stage(pb, st0, ds, mapping, 2, 80, 1.5, False)
    orc.set_num_threads(0)
    try:
        o = cal.calibrate_refinement_stage(pb, st0, ds, mapping, 2, 80, 1.5, False, run_ba_fn=stage._oracle_run_ba, resample_fn=stage._oracle_resample,
                                           delete_outliers_fn=lambda c, p, s, f, u: rp.delete_outlier_features(c, p, s, f, u, project_fn=stage._project))
    finally:
        orc.set_num_threads(1)
    case = "calibrate_refinement_stage (2 pyramid levels, outlier factor 1.5): HIP engine vs the oracle-run stage"
    print(case, [b["iterations"] for b in g["ba_runs"]], [b["iterations"] for b in o["ba_runs"]], int((~g["keep"]).sum()))
    assert [(c.grid_w, c.grid_h) for c in g["problem"].cameras] == [(10, 8)] and len(g["ba_runs"]) == 4
    check_equal(case, "RunBundleAdjustment runs whose iteration count differs", sum(a["iterations"] != b["iterations"] for a, b in zip(g["ba_runs"], o["ba_runs"])))
    check_equal(case, "outlier mask entries that differ", int((g["keep"] != o["keep"]).sum()))
    check_equal(case, "image_used entries that differ", int((g["image_used"] != o["image_used"]).sum()))
    check(case, "final cost of the last run, rel", abs(g["ba_runs"][-1]["final_cost"] - o["ba_runs"][-1]["final_cost"]) / o["ba_runs"][-1]["final_cost"], 1e-3,
          note="two trajectories that end on an absolute cost threshold of 1e-4; see tests/test_oracle_vs_ref_calibrate_stage.py")
    check(case, "metric scale factor, rel", abs(g["scale"] - o["scale"]) / o["scale"], 1e-4)


_PATCHED_SCRIPT = r"""
import json, sys
import numpy as np
from camera_calibration_amd import synthetic as syn
from camera_calibration_amd.problem import NONCENTRAL_GENERIC
from oracle import oracle as orc
from oracle import ref
import dataclasses
case = sys.argv[1]
kw = dict(model_type=NONCENTRAL_GENERIC) if case == "noncentral" else {}
pb, st0, _ = syn.reference_test_problem(2 if case.startswith("rig") else 1, orc.project, seed=7, num_points=40, num_poses=8, **kw)
pb = dataclasses.replace(pb, eliminate_points=case == "rig_eliminate_points", localize_only=case == "rig_localize_only")
out = {}
for name, mode in (("dense", ref.SCHUR_MODE_DENSE), ("hip", ref.SCHUR_MODE_HIP)):
    st = st0.copy(); lp = np.zeros((pb.n_obs, 2)); lam = -1.0; rows = []
    for _ in range(4):
        r = ref.patched_optimize_jointly(pb, st, lp, 1, lam, mode)
        lam = r["final_lambda"]; rows.append([r["cost"], lam, int(r["performed"])])
    out[name] = dict(rows=rows, points=st.points.tolist(), rig=st.rig_tr_global.tolist(), grid=np.asarray(st.grids[0]).reshape(-1).tolist())
print("RESULT " + json.dumps(out))
"""


@pytest.mark.parametrize("case", ["1cam", "rig", "noncentral", "rig_eliminate_points", "rig_localize_only"])
def test_the_patched_references_own_optimize_jointly_in_schur_mode_hip(case):
    """THE drop-in test: the reference's own vis::OptimizeJointly -- its Dataset, BAState, CentralGenericModel / NoncentralGenericModel, compiled
    from the reference's sources with integration/reference.patch applied (oracle/_ref/patched/libcalibref_ba.so, `make -C oracle patched`) --
    called with SchurMode::HIP (the patch's dispatch -> joint_optimization_hip.cc -> include/cba.h -> libcalib_ba_hip.so -> MI355X) and with
    SchurMode::Dense (the reference's CPU path, same library, same inputs): four calls each, lambda carried.  (Runs in a child process: a
    failed CHECK in the reference's code aborts.  Rounds 5 and 6 on MI355X: lambda <= 1e-8 / 1e-6, same accept decisions, state <= 1e-6.)
    The two rig cases with eliminate_points / localize_only go through the flags of APP/bundle_adjustment/joint_optimization.h:45-70."""
    import json
    import subprocess
    if not ref.patched_available():
        pytest.skip("oracle/_ref/patched/libcalibref_ba.so not shipped with this snapshot")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _PATCHED_SCRIPT, case], capture_output=True, text=True, timeout=240, cwd=root,
                       env=dict(os.environ, PYTHONPATH=root))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(line[0][7:])
    d, h = out["dense"], out["hip"]
    name = f"the patched reference's own OptimizeJointly: SchurMode::HIP vs SchurMode::Dense, {case}"
    for (cd, ld, pd), (ch, lh, ph) in zip(d["rows"], h["rows"]):
        assert pd == ph
        check_equal(name, "accept decisions that differ", int(pd != ph))
        check(name, "lambda rel per call (equal lambdas = equal decisions)", abs(lh - ld) / ld, 1e-6 if case == "noncentral" else 1e-8)
        check(name, "cost rel per call", abs(ch - cd) / cd, 1e-4)
    state = max(float(np.abs(np.array(d[k]) - np.array(h[k])).max()) for k in ("points", "rig", "grid"))
    check(name, "state after four calls, raw max abs", state, 1e-6)
