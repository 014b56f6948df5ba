"""ChooseNiceCameraOrientation, ScaleToMetric and the RunBundleAdjustment loop
(APP/models/central_generic.cc:570-621, APP/calibration.cc:187-370)."""
import numpy as np
import pytest

from camera_calibration_amd import calibration as cal
from camera_calibration_amd import calibration_io as cio
from camera_calibration_amd import synthetic as syn
from camera_calibration_amd.problem import State
from oracle import oracle as orc


def _rotated_model(seed):
    pb, st, _ = syn.reference_test_problem(1, orc.project, seed=seed, num_points=20, num_poses=4)
    cam, g = pb.cameras[0], st.grids[0]
    # tilt the model so that the "nice" orientation is a non-trivial rotation
    a, b = 0.2, -0.15
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Rz = np.array([[np.cos(b), -np.sin(b), 0], [np.sin(b), np.cos(b), 0], [0, 0, 1]])
    return pb, st, cam, g @ (Rz @ Rx).T


def test_choose_nice_camera_orientation_host_logic_matches_oracle():
    pb, st, cam, g = _rotated_model(41)
    R_ref, g_ref = orc.choose_nice_camera_orientation(cam, g)
    R, g2 = cal.choose_nice_camera_orientation(cam, g, unproject_fn=lambda c, gr, px: orc.unproject(c, gr, px))
    np.testing.assert_allclose(R, R_ref, atol=1e-12)
    np.testing.assert_allclose(g2, g_ref, atol=1e-12)
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
    # the property the function establishes: centre looks along +z, the right band along +x (no y component)
    centre = orc.unproject(cam, g2, np.array([[0.5 * cam.width, 0.5 * cam.height]]))[0][0, :3]
    np.testing.assert_allclose(centre, [0, 0, 1], atol=1e-9)
    pose = cal.rotation_to_pose(R)
    from camera_calibration_amd.se3 import quat_to_matrix
    np.testing.assert_allclose(quat_to_matrix(pose[:4]), R, atol=1e-12)


def test_scale_to_metric_matches_oracle():
    rng = np.random.default_rng(5)
    # a 4x3 lattice with pitch 0.02 m, stored at a wrong scale with noise
    ids, pos, pts = {}, {}, []
    for y in range(3):
        for x in range(4):
            fid = 100 + len(pts); pos[fid] = (x, y); ids[fid] = len(pts)
            pts.append(np.array([x, y, 0.0]) * 0.02 * 3.7 + rng.normal(0, 1e-4, 3))
    pts = np.array(pts)
    ds = cio.DatasetData(image_sizes=[(10, 10)], known_geometries=[cio.KnownGeometry(0.02, pos)])
    st = State(np.array([[1.0, 0, 0, 0, 0.1, 0.2, 0.3]]), np.array([[1.0, 0, 0, 0, 0.0, 0.0, 0.0]]), pts, [np.zeros((4, 3))])
    f, st2 = cal.scale_to_metric(ds, st, ids)
    f_ref = orc.scale_to_metric_factor([(np.float32(0.02), pos)], pts, ids)
    assert abs(f - f_ref) <= 1e-12 * f_ref and abs(f - 1 / 3.7) < 1e-2
    np.testing.assert_allclose(st2.points, pts * f)
    np.testing.assert_allclose(st2.rig_tr_global[0, 4:], np.array([0.1, 0.2, 0.3]) * f)
    np.testing.assert_allclose(st2.rig_tr_global[0, :4], st.rig_tr_global[0, :4])


@pytest.mark.gpu
def test_choose_nice_camera_orientation_on_gpu():
    pb, st, cam, g = _rotated_model(42)
    R_ref, g_ref = orc.choose_nice_camera_orientation(cam, g)
    R, g2 = cal.choose_nice_camera_orientation(cam, g)
    np.testing.assert_allclose(R, R_ref, atol=1e-10)
    np.testing.assert_allclose(g2, g_ref, atol=1e-10)


@pytest.mark.gpu
def test_run_bundle_adjustment_loop_converges_like_the_reference_gtest():
    """RunBundleAdjustment incl. the per-iteration orientation beautification: the re-parametrisation must not
    disturb convergence (cost <= C * 1e-6 as in TestOptimizeJointly, APP/test/util.h:275-571)."""
    pb, st0, gt = syn.reference_test_problem(1, orc.project, seed=0)
    st, costs = cal.run_bundle_adjustment(pb, st0, 40, 0.0)
    assert costs[-1] <= 1e-6 and all(b <= a * (1 + 1e-9) + 1e-9 for a, b in zip(costs, costs[1:]))
    centre = orc.unproject(pb.cameras[0], st.grids[0], np.array([[0.5 * pb.cameras[0].width, 0.5 * pb.cameras[0].height]]))[0][0, :3]
    np.testing.assert_allclose(centre, [0, 0, 1], atol=1e-6)


def _host_lib():
    import ctypes as C
    import os
    from camera_calibration_amd import engine as eng
    lib_dir = os.path.dirname(os.path.abspath(cal.__file__))
    C.CDLL(os.path.join(lib_dir, "libcalib_ba_hip.so"), mode=C.RTLD_GLOBAL)
    return C.CDLL(os.path.join(lib_dir, "libcalib_ba_host_test.so"))


def test_cpp_scale_to_metric_matches_oracle():
    import ctypes as C
    try:
        L = _host_lib()
    except OSError as e:
        pytest.skip(f"cannot load host library here: {e}")
    rng = np.random.default_rng(6)
    fid, pos, pts = [], [], []
    for y in range(3):
        for x in range(5):
            fid.append(200 + len(pts)); pos.append((x, y)); pts.append(np.array([x, y, 0.0]) * 0.015 * 2.2 + rng.normal(0, 1e-4, 3))
    pts = np.ascontiguousarray(np.array(pts)); pts_in = pts.copy()
    fid_a = np.array(fid, dtype=np.int32); pos_a = np.ascontiguousarray(np.array(pos, dtype=np.int32))
    pose = np.array([1.0, 0, 0, 0, 0.3, -0.2, 0.9])
    dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int32)
    rc = L.cba_host_scale_to_metric(C.c_float(0.015), C.c_int(len(fid)), fid_a.ctypes.data_as(ip), pos_a.ctypes.data_as(ip),
                                    pts.ctypes.data_as(dp), pose.ctypes.data_as(dp))
    assert rc == 0
    f_ref = orc.scale_to_metric_factor([(np.float32(0.015), dict(zip(fid, pos)))], pts_in, {f: i for i, f in enumerate(fid)})
    np.testing.assert_allclose(pts, pts_in * f_ref, rtol=1e-13)
    np.testing.assert_allclose(pose[4:], np.array([0.3, -0.2, 0.9]) * f_ref, rtol=1e-13)


@pytest.mark.gpu
def test_cpp_choose_nice_camera_orientation_on_gpu():
    import ctypes as C
    from camera_calibration_amd import engine as eng
    pb, st, cam, g = _rotated_model(43)
    L = _host_lib()
    cs = eng._cam_struct(cam)
    grid = np.ascontiguousarray(g, dtype=np.float64).copy()
    R = np.zeros(9)
    dp = C.POINTER(C.c_double)
    assert L.cba_host_nice_orientation(C.byref(cs), grid.ctypes.data_as(dp), R.ctypes.data_as(dp)) == 0
    R_ref, g_ref = orc.choose_nice_camera_orientation(cam, g)
    np.testing.assert_allclose(R.reshape(3, 3), R_ref, atol=1e-10)
    np.testing.assert_allclose(grid, g_ref, atol=1e-10)
