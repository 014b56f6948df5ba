"""calibration report statistics (APP/calibration_report.cc:101-168, 676-693).

CPU: the host logic (ordering, skipping failed projections, histogram binning, median rule) against the
oracle's scalar restatement, with the oracle's projection injected.  GPU: the same through the HIP
projection kernel (cba_project): validity bit-exact, errors <= 1e-9 px, histogram counts identical,
median / average / maximum rel 1e-9.
"""
import numpy as np
import pytest

from camera_calibration_amd import report, synthetic as syn
from oracle import oracle as orc


def _problem(num_cameras, seed):
    pb, st, gt = syn.reference_test_problem(num_cameras, orc.project, seed=seed, num_points=30, num_poses=6)
    # push some points out of view so that failed projections are skipped in both implementations
    st.points[:3] += np.array([5.0, -4.0, 0.5])
    return pb, st


def _compare(res, ref):
    assert res["count"] == ref["count"]
    np.testing.assert_allclose(res["errors"], ref["errors"], atol=1e-9)
    np.testing.assert_array_equal(res["features"], ref["features"])
    assert abs(res["sum"] - ref["sum"]) <= 1e-9 * max(1.0, ref["sum"])
    assert abs(res["max"] - ref["max"]) <= 1e-9 * max(1.0, ref["max"])
    h = report.reprojection_error_histogram(50, 2.0, res["errors"])
    h_ref = orc.reprojection_error_histogram(50, 2.0, ref["errors"])
    np.testing.assert_array_equal(h, h_ref)
    s = report.reprojection_error_summary(res)
    mags = np.sort(np.linalg.norm(ref["errors"], axis=1))
    assert abs(s["reprojection_error_median"] - mags[mags.size // 2]) <= 1e-9
    assert abs(s["reprojection_error_average"] - ref["sum"] / ref["count"]) <= 1e-9


@pytest.mark.parametrize("num_cameras", [1, 2])
def test_report_host_logic_matches_oracle(num_cameras):
    pb, st = _problem(num_cameras, seed=21)
    for c in range(num_cameras):
        ref = orc.all_reprojection_errors(c, pb, st)
        res = report.compute_all_reprojection_errors(c, pb, st, project_fn=lambda cam, g, p: orc.project(cam, g, p))
        assert ref["count"] < int((pb.obs_camera == c).sum())   # some projections failed and were skipped
        _compare(res, ref)


def test_histogram_binning_rule():
    # negative fractional bins round towards -inf (the "- (hx_f < 0)" fix-up), the upper edge is excluded
    e = np.array([[-2.0, 0.0], [-2.01, 0.0], [1.999, -1.999], [2.0, 0.0], [0.0, 0.0]])
    h = report.reprojection_error_histogram(4, 2.0, e)
    assert h.sum() == 3 and h[2, 0] == 1 and h[0, 3] == 1 and h[2, 2] == 1
    np.testing.assert_array_equal(h, orc.reprojection_error_histogram(4, 2.0, e))


@pytest.mark.gpu
@pytest.mark.parametrize("num_cameras", [1, 2])
def test_report_statistics_on_gpu(num_cameras):
    pb, st = _problem(num_cameras, seed=22)
    for c in range(num_cameras):
        ref = orc.all_reprojection_errors(c, pb, st)
        res = report.compute_all_reprojection_errors(c, pb, st)
        _compare(res, ref)


@pytest.mark.gpu
def test_cpp_report_mirror_on_gpu():
    """vis::ComputeAllReprojectionErrors / ComputeReprojectionErrorHistogram of the C++ host mirror
    (camera_calibration_amd/host/calibration_report.h) driven through the extern "C" shim."""
    import ctypes as C
    import os
    from camera_calibration_amd import engine as eng
    pb, st = _problem(2, seed=23)
    eng.load()
    L = C.CDLL(os.path.join(os.path.dirname(eng.LIB_PATH), "libcalib_ba_host_test.so"))
    dp = C.POINTER(C.c_double)
    cams = (eng.CbaCamera * pb.n_cameras)(*[eng._cam_struct(c) for c in pb.cameras])
    grids = [np.ascontiguousarray(g, dtype=np.float64) for g in st.grids]
    gp = (dp * pb.n_cameras)(*[g.ctypes.data_as(dp) for g in grids])
    res = 50
    for cam in range(pb.n_cameras):
        count = C.c_int64(0); s = C.c_double(0); mx = C.c_double(0)
        errs = np.zeros((pb.n_obs, 2)); feats = np.zeros((pb.n_obs, 2), dtype=np.float32); hist = np.zeros((res, res))
        rig = np.ascontiguousarray(st.rig_tr_global); ctr = np.ascontiguousarray(st.camera_tr_rig); pts = np.ascontiguousarray(st.points)
        rc = L.cba_host_reprojection_report(
            C.c_int(pb.n_cameras), cams, gp, C.c_int(cam), C.c_int(pb.n_images), rig.ctypes.data_as(dp), ctr.ctypes.data_as(dp),
            C.c_int(pb.n_points), pts.ctypes.data_as(dp), C.c_int64(pb.n_obs), pb.obs_xy.ctypes.data_as(C.POINTER(C.c_float)),
            pb.obs_point.ctypes.data_as(C.POINTER(C.c_int32)), pb.obs_image.ctypes.data_as(C.POINTER(C.c_int32)),
            pb.obs_camera.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(count), C.byref(s), C.byref(mx),
            errs.ctypes.data_as(dp), feats.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(res), C.c_double(2.0), hist.ctypes.data_as(dp),
            C.c_float(1.5), None, None)
        assert rc == 0
        ref = orc.all_reprojection_errors(cam, pb, st)
        n = count.value
        assert n == ref["count"]
        np.testing.assert_allclose(errs[:n], ref["errors"], atol=1e-9)
        np.testing.assert_array_equal(feats[:n], ref["features"])
        assert abs(s.value - ref["sum"]) <= 1e-9 * max(1.0, ref["sum"]) and abs(mx.value - ref["max"]) <= 1e-9
        np.testing.assert_array_equal(hist, orc.reprojection_error_histogram(res, 2.0, ref["errors"]))


def _outlier_problem(seed):
    pb, st = _problem(1, seed)
    rng = np.random.default_rng(seed)
    xy = pb.obs_xy.copy()
    bad = rng.choice(pb.n_obs, size=8, replace=False)
    xy[bad] += rng.normal(0, 25.0, size=(8, 2)).astype(np.float32)      # gross outliers
    pb.obs_xy = xy
    # imageset 2 keeps only two observations -> it must become unused
    drop = np.flatnonzero(pb.obs_image == 2)[2:]
    m = np.ones(pb.n_obs, dtype=bool); m[drop] = False
    from camera_calibration_amd.problem import Problem
    return Problem(pb.cameras, pb.n_images, pb.n_points, pb.obs_xy[m], pb.obs_point[m], pb.obs_image[m], pb.obs_camera[m], pb.fd_delta), st


def test_delete_outlier_features_host_logic_matches_oracle():
    pb, st = _outlier_problem(31)
    keep_ref, used_ref, thr_ref = orc.delete_outlier_features(0, pb, st, 1.5)
    keep, used, thr = report.delete_outlier_features(0, pb, st, 1.5, project_fn=lambda cam, g, p: orc.project(cam, g, p))
    assert (~keep_ref).sum() >= 8 and not used_ref[2]
    np.testing.assert_array_equal(keep, keep_ref)
    np.testing.assert_array_equal(used, used_ref)
    assert abs(thr - thr_ref) <= 1e-12 * thr_ref


@pytest.mark.gpu
def test_delete_outlier_features_on_gpu():
    pb, st = _outlier_problem(32)
    keep_ref, used_ref, thr_ref = orc.delete_outlier_features(0, pb, st, 1.5)
    keep, used, thr = report.delete_outlier_features(0, pb, st, 1.5)
    np.testing.assert_array_equal(keep, keep_ref)        # removal decisions bit-exact
    np.testing.assert_array_equal(used, used_ref)
    assert abs(thr - thr_ref) <= 1e-9


@pytest.mark.gpu
def test_cpp_delete_outlier_features_on_gpu():
    import ctypes as C
    import os
    from camera_calibration_amd import engine as eng
    pb, st = _outlier_problem(33)
    eng.load()
    L = C.CDLL(os.path.join(os.path.dirname(eng.LIB_PATH), "libcalib_ba_host_test.so"))
    dp = C.POINTER(C.c_double)
    cams = (eng.CbaCamera * pb.n_cameras)(*[eng._cam_struct(c) for c in pb.cameras])
    grids = [np.ascontiguousarray(g, dtype=np.float64) for g in st.grids]
    gp = (dp * pb.n_cameras)(*[g.ctypes.data_as(dp) for g in grids])
    count = C.c_int64(0); s = C.c_double(0); mx = C.c_double(0)
    errs = np.zeros((pb.n_obs, 2)); feats = np.zeros((pb.n_obs, 2), dtype=np.float32); hist = np.zeros((10, 10))
    keep = np.zeros(pb.n_obs, dtype=np.uint8); used = np.zeros(pb.n_images, dtype=np.uint8)
    rig = np.ascontiguousarray(st.rig_tr_global); ctr = np.ascontiguousarray(st.camera_tr_rig); pts = np.ascontiguousarray(st.points)
    rc = L.cba_host_reprojection_report(
        C.c_int(pb.n_cameras), cams, gp, C.c_int(0), C.c_int(pb.n_images), rig.ctypes.data_as(dp), ctr.ctypes.data_as(dp),
        C.c_int(pb.n_points), pts.ctypes.data_as(dp), C.c_int64(pb.n_obs), pb.obs_xy.ctypes.data_as(C.POINTER(C.c_float)),
        pb.obs_point.ctypes.data_as(C.POINTER(C.c_int32)), pb.obs_image.ctypes.data_as(C.POINTER(C.c_int32)),
        pb.obs_camera.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(count), C.byref(s), C.byref(mx),
        errs.ctypes.data_as(dp), feats.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(10), C.c_double(2.0), hist.ctypes.data_as(dp),
        C.c_float(1.5), keep.ctypes.data_as(C.POINTER(C.c_uint8)), used.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0
    keep_ref, used_ref, _ = orc.delete_outlier_features(0, pb, st, 1.5)
    np.testing.assert_array_equal(keep.astype(bool), keep_ref)
    np.testing.assert_array_equal(used.astype(bool), used_ref)
