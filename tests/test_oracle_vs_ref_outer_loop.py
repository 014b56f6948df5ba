"""SURVEY 8f rows F1 and F4 pinned to the REFERENCE'S OWN CODE: DeleteOutlierFeatures, RunBundleAdjustment, ScaleToMetric
(APP/calibration.cc:62-184, 187-304, 307-370), CentralGenericModel::ChooseNiceCameraOrientation (APP/models/central_generic.cc:570-621),
ComputeAllReprojectionErrors / ComputeReprojectionErrorHistogram (APP/calibration_report.cc:101-168) and the median rule of the
report (:686-692), compiled from /root/reference into oracle/_ref/libcalibref_f14.so (oracle/Makefile pipes the functions' line ranges
into the compiler; oracle/ref_f14_glue.cc marshals).  Each test compares three things on the same inputs: the reference's function,
the oracle's restatement (oracle/oracle.py) and the product's host logic (camera_calibration_amd.calibration / .report, with the
projection injected so that no GPU is needed).

Tolerances: decisions (keep masks, image_used, histogram bins, iteration counts) exact; floating-point values 1e-12 relative -- the
reference side runs on the run-time-sized Eigen stand-in of oracle/ref_shim_lm, whose 3x3 products sum in the same order as the
restatements but are not the same machine code."""
import dataclasses

import numpy as np
import pytest

from camera_calibration_amd import calibration as cal
from camera_calibration_amd import calibration_io as cio
from camera_calibration_amd import report as rp
from camera_calibration_amd import synthetic as syn
from camera_calibration_amd.problem import CENTRAL_GENERIC, Camera, Problem, State
from camera_calibration_amd.se3 import se3_mul
from oracle import oracle as orc
from oracle import ref

pytestmark = pytest.mark.skipif(not ref.f14_available(), reason="oracle/_ref/libcalibref_f14.so not built (needs /root/reference)")

_project = lambda cam, grid, pts: orc.project(cam, grid, pts)          # noqa: E731
_unproject = lambda cam, grid, px: orc.unproject(cam, grid, px)       # noqa: E731


def _subset(pb, mask):
    return Problem(pb.cameras, pb.n_images, pb.n_points, pb.obs_xy[mask], pb.obs_point[mask], pb.obs_image[mask], pb.obs_camera[mask],
                   pb.fd_delta, pb.localize_only, pb.eliminate_points)


def _tilted(seed, a=0.2, b=-0.15, **kw):
    pb, st, _ = syn.reference_test_problem(1, orc.project, seed=seed, num_points=20, num_poses=4, **kw)
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Rz = np.array([[np.cos(b), -np.sin(b), 0], [np.sin(b), np.cos(b), 0], [0, 0, 1]])
    return pb.cameras[0], st.grids[0] @ (Rz @ Rx).T


# ---- F1: ChooseNiceCameraOrientation ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,a,b", [(41, 0.2, -0.15), (7, -0.4, 0.3), (9, 0.0, 0.0), (11, 0.05, 1.2)])
def test_choose_nice_camera_orientation_is_the_references(seed, a, b):
    cam, g = _tilted(seed, a, b)
    R_ref, g_ref = ref.f1_choose_nice_camera_orientation(cam, g)
    R_orc, g_orc = orc.choose_nice_camera_orientation(cam, g)
    R_host, g_host = cal.choose_nice_camera_orientation(cam, g, unproject_fn=_unproject)
    for R, gg in ((R_orc, g_orc), (R_host, g_host)):
        np.testing.assert_allclose(R, R_ref, rtol=0, atol=1e-14)
        np.testing.assert_allclose(np.asarray(gg).reshape(-1, 3), g_ref, rtol=0, atol=1e-14)
    np.testing.assert_allclose(R_ref @ R_ref.T, np.eye(3), atol=1e-14)


def test_choose_nice_camera_orientation_fallbacks_are_the_references():
    """Image centre outside the calibrated area -> forward = (0, 0, 1) (central_generic.cc:583-585); no pixel of the right band inside
    it -> right_rotation = identity (:614-616)."""
    base, g = _tilted(13, 0.1, 0.2)
    w, h = base.width, base.height
    left_only = Camera(CENTRAL_GENERIC, w, h, base.calib_min_x, base.calib_min_y, w // 2 - 5, base.calib_max_y, base.grid_w, base.grid_h)
    centre_in_band_out = Camera(CENTRAL_GENERIC, w, h, base.calib_min_x, base.calib_min_y, w // 2 + 5, base.calib_max_y, base.grid_w, base.grid_h)
    for cam in (left_only, centre_in_band_out):
        R_ref, g_ref = ref.f1_choose_nice_camera_orientation(cam, g)
        R_orc, g_orc = orc.choose_nice_camera_orientation(cam, g)
        R_host, g_host = cal.choose_nice_camera_orientation(cam, g, unproject_fn=_unproject)
        np.testing.assert_allclose(R_orc, R_ref, rtol=0, atol=1e-14)
        np.testing.assert_allclose(R_host, R_ref, rtol=0, atol=1e-14)
        np.testing.assert_allclose(np.asarray(g_host).reshape(-1, 3), g_ref, rtol=0, atol=1e-14)
    R_ref, _ = ref.f1_choose_nice_camera_orientation(left_only, g)
    np.testing.assert_array_equal(R_ref, np.eye(3))                      # both fallbacks at once: FromTwoVectors(z, z) and identity


# ---- F1: ScaleToMetric -----------------------------------------------------------------------------------------------------
def _lattice(rng, nx, ny, pitch, scale, missing=()):
    ids, pos, pts = {}, {}, []
    for y in range(ny):
        for x in range(nx):
            fid = 100 + y * nx + x
            pos[fid] = (x, y)
            if fid in missing:
                continue
            ids[fid] = len(pts)
            pts.append(np.array([x, y, 0.0]) * pitch * scale + rng.normal(0, 1e-4, 3))
    return ids, pos, np.array(pts)


@pytest.mark.parametrize("missing", [(), (101, 105, 110)])
def test_scale_to_metric_is_the_references(missing):
    rng = np.random.default_rng(5)
    ids, pos, pts = _lattice(rng, 4, 3, 0.02, 3.7, missing)
    rig = np.array([[1.0, 0, 0, 0, 0.1, 0.2, 0.3], [0.8, 0.6, 0, 0, -1.0, 0.5, 2.0]])
    ctr = np.array([[1.0, 0, 0, 0, 0.0, 0.0, 0.0], [0.6, 0, 0.8, 0, 0.3, -0.1, 0.05]])
    f_ref, p_ref, rig_ref, ctr_ref = ref.f1_scale_to_metric(0.02, pos, ids, pts, rig, ctr)
    f_orc = orc.scale_to_metric_factor([(np.float32(0.02), pos)], pts, ids)
    ds = cio.DatasetData(image_sizes=[(10, 10), (10, 10)], known_geometries=[cio.KnownGeometry(0.02, pos)])
    f_host, st2 = cal.scale_to_metric(ds, State(rig, ctr, pts, [np.zeros((4, 3)), np.zeros((4, 3))]), ids)
    assert abs(f_orc - f_ref) <= 1e-14 * f_ref and abs(f_host - f_ref) <= 1e-14 * f_ref and abs(f_ref - 1 / 3.7) < 1e-2
    np.testing.assert_allclose(st2.points, p_ref, rtol=1e-15)
    np.testing.assert_allclose(st2.rig_tr_global, rig_ref, rtol=1e-15)      # rotations untouched, translations scaled (ba_state.cc ScaleState)
    np.testing.assert_allclose(st2.camera_tr_rig, ctr_ref, rtol=1e-15)


# ---- F4: ComputeAllReprojectionErrors, histogram, median -------------------------------------------------------------------
def _rig_problem(seed=3):
    return syn.reference_test_problem(2, orc.project, seed=seed, num_points=40, num_poses=6)[:2]


def test_compute_all_reprojection_errors_is_the_references():
    pb, st = _rig_problem()
    for used in (None, np.array([1, 0, 1, 1, 0, 1], dtype=bool)):
        for c in range(pb.n_cameras):
            r = ref.f4_compute_all_reprojection_errors(c, pb, st, used)
            if used is None:
                o = orc.all_reprojection_errors(c, pb, st)
                assert o["count"] == r["count"]
                np.testing.assert_allclose(o["errors"], r["errors"], rtol=0, atol=_TRAJECTORY_ATOL)
                np.testing.assert_array_equal(o["features"], r["features"])
            sub = pb if used is None else _subset(pb, used[pb.obs_image])       # the product takes image_used as a smaller problem
            h = rp.compute_all_reprojection_errors(c, sub, st, project_fn=_project)
            assert h["count"] == r["count"] and r["count"] > 50
            np.testing.assert_allclose(h["errors"], r["errors"], rtol=0, atol=_TRAJECTORY_ATOL)        # pixels of order 1e2-1e3: 1e-13 relative
            np.testing.assert_array_equal(h["features"], r["features"])                     # same observations, same order
            assert abs(h["sum"] - r["sum"]) <= 1e-12 * r["sum"] and abs(h["max"] - r["max"]) <= 1e-12 * r["max"]


def test_reprojection_error_histogram_and_median_are_the_references():
    rng = np.random.default_rng(0)
    for n, res, extent in ((1000, 50, 2.0), (37, 7, 0.5), (2, 4, 1.0), (513, 64, 0.2)):
        e = rng.normal(0, extent * 0.6, (n, 2))
        e[: n // 10] *= 4.0                                                 # a tail outside the extent, both signs
        e[0] = [-extent, extent]; e[1] = [extent * (1 - 1e-16), -extent * (1 + 1e-15)]    # bin edges
        h_ref = ref.f4_reprojection_error_histogram(res, extent, e)
        np.testing.assert_array_equal(orc.reprojection_error_histogram(res, extent, e), h_ref)
        np.testing.assert_array_equal(rp.reprojection_error_histogram(res, extent, e), h_ref)
        assert h_ref.sum() <= n - 2 and (h_ref.sum() > 0 or n == 2)             # the two edge rows fall outside (upper edge is exclusive)
        mags = np.sqrt(e[:, 0] ** 2 + e[:, 1] ** 2)
        summary = rp.reprojection_error_summary(dict(errors=e, count=n, sum=float(mags.sum()), max=float(mags.max())))
        med_ref = ref.f4_reprojection_error_median(e)
        assert abs(summary["reprojection_error_median"] - med_ref) <= 4e-16 * med_ref       # upper median (index n / 2), 17 printed digits


# ---- F1: DeleteOutlierFeatures ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("factor", [0.5, 1.5, 3.0])
def test_delete_outlier_features_is_the_references(factor):
    pb, st = _rig_problem()
    for used in (None, np.array([1, 1, 0, 1, 1, 1], dtype=bool)):
        for c in range(pb.n_cameras):
            k_ref, u_ref = ref.f1_delete_outlier_features(c, pb, st, factor, used)
            k_orc, u_orc, thr = orc.delete_outlier_features(c, pb, st, factor, used)
            k_host, u_host, thr_host = rp.delete_outlier_features(c, pb, st, factor, used, project_fn=_project)
            np.testing.assert_array_equal(k_orc, k_ref)
            np.testing.assert_array_equal(k_host, k_ref)
            np.testing.assert_array_equal(u_orc, u_ref)
            np.testing.assert_array_equal(u_host, u_ref)
            assert (~k_ref).sum() >= 1 and abs(thr - thr_host) <= 1e-12 * thr
            assert k_ref[pb.obs_camera != c].all()                          # other cameras' features untouched
            if used is not None:
                assert k_ref[~used[pb.obs_image]].all() and not u_ref[2]    # unused imagesets neither measured nor removed


def test_delete_outlier_features_edge_cases_are_the_references():
    """Fewer than eight measurable features -> nothing happens (calibration.cc:97-100); features that do not project are removed
    (:140-144); an imageset left with fewer than three features of the camera is marked unused (:165-167)."""
    pb, st = _rig_problem(seed=8)
    # (a) 7 observations of camera 0 only
    sel = np.flatnonzero(pb.obs_camera == 0)[:7]
    few = _subset(pb, np.isin(np.arange(pb.n_obs), sel))
    k_ref, u_ref = ref.f1_delete_outlier_features(0, few, st, 1.5)
    k_host, u_host, thr = rp.delete_outlier_features(0, few, st, 1.5, project_fn=_project)
    assert thr is None and k_ref.all() and u_ref.all()
    np.testing.assert_array_equal(k_host, k_ref)
    np.testing.assert_array_equal(u_host, u_ref)
    # (b) move three points behind the camera: their features fail to project and go; tight factor empties some imagesets
    st2 = st.copy()
    st2.points[[0, 1, 2]] += np.array([0.0, 0.0, -50.0])
    for factor in (0.0, 1.5):
        k_ref, u_ref = ref.f1_delete_outlier_features(1, pb, st2, factor)
        k_orc, u_orc, _ = orc.delete_outlier_features(1, pb, st2, factor)
        k_host, u_host, _ = rp.delete_outlier_features(1, pb, st2, factor, project_fn=_project)
        np.testing.assert_array_equal(k_orc, k_ref)
        np.testing.assert_array_equal(k_host, k_ref)
        np.testing.assert_array_equal(u_orc, u_ref)
        np.testing.assert_array_equal(u_host, u_ref)
        gone = np.isin(pb.obs_point, [0, 1, 2]) & (pb.obs_camera == 1)
        assert gone.any() and not k_ref[gone].any()


# ---- F1: RunBundleAdjustment -----------------------------------------------------------------------------------------------
# Two runs of the loop whose states agree to 6e-16 after the first call are 3e-10 ... 2e-9 apart after the second and stay there
# (measured, 1-6 calls): the iterative projection stops at a finite tolerance, so an ulp in the state can change one observation's
# pixel by 1e-7 px, and the weakly damped gauge directions of these small problems carry that into the state.  The bound keeps ~20x.
_TRAJECTORY_ATOL = 5e-8


def _restated_run_bundle_adjustment(pb, st0, max_iteration_count, threshold, localize_only=False):
    """The loop as camera_calibration_amd.calibration.run_bundle_adjustment runs it, with the oracle in place of the HIP engine."""
    pb = dataclasses.replace(pb, localize_only=localize_only)
    op = orc.OracleProblem(pb)
    st = st0.copy()
    lam, last, costs = -1.0, float("inf"), []
    for _ in range(max_iteration_count):
        r = op.optimize_jointly(st, 1, lam)
        lam = r["final_lambda"]
        costs.append(r["cost"])
        if not localize_only:
            for c, cam in enumerate(pb.cameras):
                R, g = cal.choose_nice_camera_orientation(cam, st.grids[c], unproject_fn=_unproject)
                st.grids[c][...] = np.asarray(g).reshape(st.grids[c].shape)
                st.camera_tr_rig[c] = se3_mul(cal.rotation_to_pose(R), st.camera_tr_rig[c])
        if r["cost"] >= last - threshold:
            break
        last = r["cost"]
    return st, costs


@pytest.mark.parametrize("n_cameras,seed,threshold", [(1, 0, 1e-4), (2, 1, 1e-4), (1, 2, 10.0)])
def test_run_bundle_adjustment_loop_is_the_references(n_cameras, seed, threshold):
    """The reference's loop text (stopping rule :298, lambda carried from call to call, numerical_diff_delta 1e-4, the orientation
    beautification after every iteration applied to camera_tr_rig from the left) around the same OptimizeJointly as the restated loop:
    same number of calls, same final state."""
    pb, st0, _ = syn.reference_test_problem(n_cameras, orc.project, seed=seed, num_points=30, num_poses=5)
    st_ref, calls, delta = ref.f1_run_bundle_adjustment(pb, st0, 40, threshold)
    st_host, costs = _restated_run_bundle_adjustment(pb, st0, 40, threshold)
    assert delta == 1e-4 == pb.fd_delta
    assert calls == len(costs) and calls >= 3
    if threshold >= 1:                                                      # a coarse threshold stops the loop before convergence
        assert calls < ref.f1_run_bundle_adjustment(pb, st0, 40, 1e-4)[1]
    np.testing.assert_allclose(st_host.points, st_ref.points, rtol=0, atol=_TRAJECTORY_ATOL)
    np.testing.assert_allclose(st_host.rig_tr_global, st_ref.rig_tr_global, rtol=0, atol=_TRAJECTORY_ATOL)
    np.testing.assert_allclose(st_host.camera_tr_rig, st_ref.camera_tr_rig, rtol=0, atol=_TRAJECTORY_ATOL)
    for a, b in zip(st_host.grids, st_ref.grids):
        np.testing.assert_allclose(a, b, rtol=0, atol=_TRAJECTORY_ATOL)
    # the beautified orientation holds at exit: the image centre looks along +z
    cam = pb.cameras[0]
    centre = orc.unproject(cam, st_ref.grids[0], np.array([[0.5 * cam.width, 0.5 * cam.height]]))[0][0, :3]
    np.testing.assert_allclose(centre, [0, 0, 1], atol=1e-12)


def test_run_bundle_adjustment_max_iteration_count_and_localize_only_are_the_references():
    pb, st0, _ = syn.reference_test_problem(1, orc.project, seed=4, num_points=30, num_poses=5)
    st_ref, calls, _ = ref.f1_run_bundle_adjustment(pb, st0, 2, 0.0)
    assert calls == 2                                                       # max_iteration_count bounds the loop
    st_host, costs = _restated_run_bundle_adjustment(pb, st0, 2, 0.0)
    np.testing.assert_allclose(st_host.points, st_ref.points, rtol=0, atol=_TRAJECTORY_ATOL)
    # localize_only: no beautification (:247); the intrinsics are not in the state (the points still are)
    st_ref, calls, _ = ref.f1_run_bundle_adjustment(pb, st0, 5, 1e-4, localize_only=True)
    st_host, costs = _restated_run_bundle_adjustment(pb, st0, 5, 1e-4, localize_only=True)
    assert calls == len(costs)
    np.testing.assert_array_equal(st_ref.grids[0], st0.grids[0])
    assert np.abs(st_ref.points - st0.points).max() > 1e-3
    np.testing.assert_allclose(st_host.points, st_ref.points, rtol=0, atol=_TRAJECTORY_ATOL)
    np.testing.assert_array_equal(st_ref.camera_tr_rig, st0.camera_tr_rig)
    np.testing.assert_allclose(st_host.rig_tr_global, st_ref.rig_tr_global, rtol=0, atol=_TRAJECTORY_ATOL)
