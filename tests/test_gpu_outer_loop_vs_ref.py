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
    # both sides leave the loop in the beautified orientation: image centre along +z
    cam = pb.cameras[0]
    for st in (st_ref, st_eng):
        centre = orc.unproject(cam, st.grids[0], np.array([[0.5 * cam.width, 0.5 * cam.height]]))[0][0, :3]
        np.testing.assert_allclose(centre, [0, 0, 1], atol=1e-9)


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
