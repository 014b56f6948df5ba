"""GPU test of the C++ host adapter (camera_calibration_amd/host): vis::OptimizeJointly with the
reference's signature over mirrored Dataset / BAState / CameraModel objects, driven through an
extern "C" shim.  One imageset is marked unused (BAState::image_used) to exercise the sequential
re-indexing of joint_optimization.cc:80-90."""
import ctypes as C
import os

import numpy as np
import pytest

from camera_calibration_amd import engine as eng
from camera_calibration_amd import synthetic as syn
from camera_calibration_amd.problem import Problem, State
from oracle import oracle as orc
from parity_record import check, check_equal

pytestmark = pytest.mark.gpu


def oracle_project(cam, grid, pts):
    return orc.project(cam, grid, pts)


def _host_lib():
    eng.load()
    path = os.path.join(os.path.dirname(eng.LIB_PATH), "libcalib_ba_host_test.so")
    L = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    L.cba_host_optimize_jointly.argtypes = [
        C.c_int, C.POINTER(eng.CbaCamera), C.POINTER(dp), C.POINTER(dp), C.c_int, C.POINTER(C.c_uint8), dp, dp, C.c_int, dp,
        C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
        C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, dp, dp, C.POINTER(C.c_int), dp]
    return L


@pytest.mark.parametrize("num_cameras", [1, 2])
def test_cpp_optimize_jointly_matches_engine_and_oracle(num_cameras):
    pb, st0, _ = syn.reference_test_problem(num_cameras, oracle_project, seed=13, num_points=50, num_poses=16)
    unused = 5
    image_used = np.ones(pb.n_images, dtype=np.uint8)
    image_used[unused] = 0
    # reference semantics: the unused imageset keeps its features in the Dataset but is skipped
    keep = pb.obs_image != unused
    seq = np.cumsum(image_used) - 1
    sub = Problem(pb.cameras, pb.n_images - 1, pb.n_points, pb.obs_xy[keep], pb.obs_point[keep],
                  seq[pb.obs_image[keep]].astype(np.int32), pb.obs_camera[keep], pb.fd_delta)
    sub_st = State(st0.rig_tr_global[image_used.astype(bool)], st0.camera_tr_rig, st0.points, st0.grids)
    iters = 3
    # oracle, with the same warm-start history as the adapter: OptimizeJointly(debug_verify_cost = true) runs two cost passes
    # first (VerifyCost, joint_optimization.cc:866-877), and every pass rewrites PointFeature::last_projection
    op = orc.OracleProblem(sub)
    st_ref = sub_st.copy()
    op.cost_pass(st_ref); op.cost_pass(st_ref)
    lam = -1.0
    for _ in range(iters):
        r = op.optimize_jointly(st_ref, 1, lam); lam = r["final_lambda"]
    # C++ adapter
    L = _host_lib()
    cams = (eng.CbaCamera * pb.n_cameras)(*[eng._cam_struct(c) for c in pb.cameras])
    dp = C.POINTER(C.c_double)
    g_in = [np.ascontiguousarray(g) for g in st0.grids]
    g_out = [np.zeros_like(g) for g in g_in]
    gi = (dp * pb.n_cameras)(*[g.ctypes.data_as(dp) for g in g_in])
    go = (dp * pb.n_cameras)(*[g.ctypes.data_as(dp) for g in g_out])
    rig = st0.rig_tr_global.copy(); camrig = st0.camera_tr_rig.copy(); pts = st0.points.copy()
    cost, flam, performed = C.c_double(0), C.c_double(0), C.c_int(0)
    lastp = np.zeros((pb.n_obs, 2))
    rc = L.cba_host_optimize_jointly(
        pb.n_cameras, cams, gi, go, pb.n_images, image_used.ctypes.data_as(C.POINTER(C.c_uint8)), rig.ctypes.data_as(dp),
        camrig.ctypes.data_as(dp), pb.n_points, pts.ctypes.data_as(dp), pb.n_obs, pb.obs_xy.ctypes.data_as(C.POINTER(C.c_float)),
        pb.obs_point.ctypes.data_as(C.POINTER(C.c_int32)), pb.obs_image.ctypes.data_as(C.POINTER(C.c_int32)),
        pb.obs_camera.ctypes.data_as(C.POINTER(C.c_int32)), iters, -1.0, pb.fd_delta, 0, 0,
        C.byref(cost), C.byref(flam), C.byref(performed), lastp.ctypes.data_as(dp))
    assert rc == 0 and performed.value == 1
    case = f"C++ OptimizeJointly adapter, {num_cameras} camera(s)"
    check(case, "final cost rel", abs(cost.value - r["cost"]) / abs(r["cost"]), 1e-6)
    check(case, "final lambda rel", abs(flam.value - lam) / lam, 1e-10)
    # unused imageset untouched, used ones updated like the oracle's
    np.testing.assert_array_equal(rig[unused], st0.rig_tr_global[unused])
    check(case, "poses abs", np.abs(rig[image_used.astype(bool)] - st_ref.rig_tr_global).max(), 5e-8)   # three iterations deep: 5e-9 observed, atomics order
    check(case, "points abs", np.abs(pts - st_ref.points).max(), 1e-8)
    check(case, "camera_tr_rig abs", np.abs(camrig - st_ref.camera_tr_rig).max(), 1e-8)
    for a, b in zip(g_out, st_ref.grids):
        check(case, "grids abs", np.abs(a - b).max(), 1e-8)
    # warm-start cache written back for used imagesets only
    assert np.all(lastp[~keep] == 0)
    check(case, "last_projection abs [px]", np.abs(lastp[keep] - op.last_projection).max(), 1e-8)


def test_cpp_run_bundle_adjustment_session_matches_per_call_loop():
    """vis::RunBundleAdjustment keeps ONE device-resident problem alive across the outer iterations (JointOptimizationSession);
    the reference's loop re-enters OptimizeJointly(max_iteration_count = 1) every iteration.  Same iterates, and the time the
    session saves per iteration is recorded (cba_create + observation marshalling / upload + cba_destroy)."""
    gpu_project = lambda cam, grid, pts: eng.project(cam, grid, pts)
    pb, st0, _ = syn.baseline_config(2, gpu_project, n_imagesets=120)
    L = _host_lib()
    dp = C.POINTER(C.c_double)
    L.cba_host_run_bundle_adjustment.argtypes = [
        C.c_int, C.POINTER(eng.CbaCamera), C.POINTER(dp), C.POINTER(dp), C.c_int, C.POINTER(C.c_uint8), dp, dp, C.c_int, dp,
        C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
        C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int), dp, dp]
    cams = (eng.CbaCamera * pb.n_cameras)(*[eng._cam_struct(c) for c in pb.cameras])
    image_used = np.ones(pb.n_images, dtype=np.uint8)
    out = {}
    for mode in (0, 1):
        g_in = [np.ascontiguousarray(g) for g in st0.grids]
        g_out = [np.zeros_like(g) for g in g_in]
        gi = (dp * pb.n_cameras)(*[g.ctypes.data_as(dp) for g in g_in])
        go = (dp * pb.n_cameras)(*[g.ctypes.data_as(dp) for g in g_out])
        rig = st0.rig_tr_global.copy(); camrig = st0.camera_tr_rig.copy(); pts = st0.points.copy()
        iters, secs = C.c_int(0), C.c_double(0)
        lastp = np.zeros((pb.n_obs, 2))
        rc = L.cba_host_run_bundle_adjustment(
            pb.n_cameras, cams, gi, go, pb.n_images, image_used.ctypes.data_as(C.POINTER(C.c_uint8)), rig.ctypes.data_as(dp),
            camrig.ctypes.data_as(dp), pb.n_points, pts.ctypes.data_as(dp), pb.n_obs, pb.obs_xy.ctypes.data_as(C.POINTER(C.c_float)),
            pb.obs_point.ctypes.data_as(C.POINTER(C.c_int32)), pb.obs_image.ctypes.data_as(C.POINTER(C.c_int32)),
            pb.obs_camera.ctypes.data_as(C.POINTER(C.c_int32)), 4, 1e-4, mode, C.byref(iters), C.byref(secs), lastp.ctypes.data_as(dp))
        assert rc == 0
        out[mode] = dict(rig=rig, camrig=camrig, pts=pts, grid=g_out[0], lastp=lastp, seconds=secs.value, iterations=iters.value)
    case = "C++ RunBundleAdjustment: session vs per-call loop (cfg 2 grid, 120 imagesets, 4 iterations)"
    # two runs of the SAME code: they differ by the order of the floating-point atomics of the accumulation and, from there, by the
    # occasional projection that stops one LM iterate apart (~1e-7 px in one residual); four LM iterations amplify that.  The spread
    # is heavy-tailed: poses 5e-12, 3e-11 and 1.2e-9 over three recorded runs of rounds 3-4 -- hence tolerances ~20x the largest
    note = "run-to-run spread of the default (atomic) accumulation after 4 LM iterations, heavy-tailed: 5e-12 ... 1.2e-9 seen for the poses"
    check(case, "poses abs", np.abs(out[0]["rig"] - out[1]["rig"]).max(), 2e-8, note=note)
    check(case, "points abs", np.abs(out[0]["pts"] - out[1]["pts"]).max(), 2e-8, note=note)
    check(case, "grid abs", np.abs(out[0]["grid"] - out[1]["grid"]).max(), 2e-8, note=note)
    check(case, "last_projection abs [px]", np.abs(out[0]["lastp"] - out[1]["lastp"]).max(), 1e-6, note=note)
    # measurement, not a bound (recorded in profiles/r02_parity_deviations.json): seconds of the whole loop
    check(case, "seconds, session (mode 0) (bound: one minute)", out[0]["seconds"], 60.0)
    check(case, "seconds, per-call OptimizeJointly (mode 1) (bound: one minute)", out[1]["seconds"], 60.0)
