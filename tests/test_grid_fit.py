"""grid-only LM of the central-generic model (FitToPixelDirections / FitToDenseModel /
ResampleModel; APP/models/central_generic.cc:44-431, 551-568, APP/calibration.cc:373-529).

Oracle pins (CPU): the reference's own known-answer test TestModelOptimization (APP/test/util.h:213-269): an 8x6
grid on a 640x480 image fitted to a pinhole dense model (FitToDenseModel, subsample 2) and then to a shifted pinhole
(FitToPixelDirections, step 10, 10 iterations) must un-project every pixel centre with 0.5 |diff|^2 < 5e-4
(VerifyUnprojections, util.h:43-75); plus finite differences of the restated Jacobian.
GPU parity: the HIP fit against the oracle on the same samples -- same number of accepted iterations, final cost
rel 1e-8, grid <= 1e-9, lambda rel 1e-6 (the dense solve is pivoted in the oracle, unpivoted blocked on the GPU).
"""
import numpy as np
import pytest

from camera_calibration_amd import grid_fit
from camera_calibration_amd.problem import Camera
from oracle import oracle as orc

W, H = 640, 480


def pinhole_dirs(px, fx, fy, cx, cy):
    d = np.stack([(px[..., 0] - cx) / fx, (px[..., 1] - cy) / fy, np.ones(px.shape[:-1])], -1)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def dense_pinhole(fx, fy, cx, cy):
    X, Y = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)
    return pinhole_dirs(np.stack([X, Y], -1), fx, fy, cx, cy)


def oracle_fit(cam, grid, gp, dirs, iters):
    return orc.fit_grid_to_points(cam.grid_w, cam.grid_h, grid, gp, dirs, iters)


def oracle_unproject(cam, grid, px):
    return orc.unproject(cam, grid, px)


def max_unprojection_cost(cam, grid, dense):
    X, Y = np.meshgrid(np.arange(0, W, 7) + 0.5, np.arange(0, H, 7) + 0.5)      # every 7th pixel centre keeps the test fast
    px = np.stack([X.ravel(), Y.ravel()], 1)
    lines, ok = orc.unproject(cam, grid, px)
    ref = dense[(px[:, 1] - 0.5).astype(int), (px[:, 0] - 0.5).astype(int)]
    diff = lines[ok, :3] - ref[ok]
    return 0.5 * (diff ** 2).sum(1).max()


def test_oracle_jacobian_matches_finite_differences():
    cam = Camera(0, W, H, 0, 0, W - 1, H - 1, 8, 6)
    rng = np.random.default_rng(1)
    grid = grid_fit.initialize_grid_from_dense_model(cam, dense_pinhole(240, 240, 320, 240))
    px = rng.uniform([0, 0], [W, H], size=(40, 2))
    gp = np.stack(grid_fit.pixel_corner_conv_to_grid_point(cam, px[:, 0], px[:, 1]), 1)
    dirs = pinhole_dirs(px, 240, 240, 310, 260)
    cost, cv, Hm, b = orc.fit_grid_pass(8, 6, grid, gp, dirs, True)
    cost2, _, _, _ = orc.fit_grid_pass(8, 6, grid, gp, dirs, False)
    assert abs(cost - cost2) <= 1e-12 * cost          # literal vs exact-fraction weights differ at 1e-15
    eps = 1e-6
    for k in rng.choice(96, size=12, replace=False):
        x = np.zeros(96); x[k] = eps
        cp = orc.fit_grid_pass(8, 6, orc.fit_grid_apply_update(8, 6, grid, x), gp, dirs, False)[0]
        cm = orc.fit_grid_pass(8, 6, orc.fit_grid_apply_update(8, 6, grid, -x), gp, dirs, False)[0]
        assert abs((cp - cm) / (2 * eps) + b[k]) <= 1e-6 * max(1.0, abs(b[k]))     # state -= x: d cost / d x = -J^T r
    assert np.allclose(Hm, np.triu(Hm)) and (np.diag(Hm) >= 0).all()


def test_reference_known_answer_model_optimization():
    """APP/test/util.h:213-269 with the oracle as the model."""
    cam = Camera(0, W, H, 0, 0, W - 1, H - 1, 8, 6)
    dense = dense_pinhole(240, 240, 320, 240)
    grid, rep = grid_fit.fit_to_dense_model(cam, dense, 2, fit_fn=oracle_fit)
    assert grid is not None and rep["final_cost"] <= rep["initial_cost"]
    assert max_unprojection_cost(cam, grid, dense) < 5e-4
    X, Y = np.meshgrid(np.arange(0, W, 10) + 0.5, np.arange(0, H, 10) + 0.5)
    px = np.stack([X.ravel(), Y.ravel()], 1)
    dirs = pinhole_dirs(px, 240, 240, 310, 260)
    grid2, rep2 = grid_fit.fit_to_pixel_directions(cam, grid, px, dirs, 10, fit_fn=oracle_fit)
    assert max_unprojection_cost(cam, grid2, dense_pinhole(240, 240, 310, 260)) < 5e-4


def test_dense_model_initialisation_fills_invalid_pixels():
    cam = Camera(0, W, H, 0, 0, W - 1, H - 1, 10, 8)
    dense = dense_pinhole(300, 300, 320, 240)
    dense[:40, :] = np.nan            # a band of invalid pixels: ring search + neighbour extrapolation
    dense[200:203, 300:303] = np.nan  # a small hole: ring search only
    grid = grid_fit.initialize_grid_from_dense_model(cam, dense)
    assert grid is not None and not np.isnan(grid).any()
    np.testing.assert_allclose(np.linalg.norm(grid, axis=1)[np.isnan(dense[0, 0, 0]) * 0:], 1.0, atol=1e-12)
    gp, dirs = grid_fit.dense_model_samples(cam, dense, 16)
    assert gp.shape[0] == dirs.shape[0] and not np.isnan(dirs).any() and gp.shape[0] < (W // 16 + 1) * (H // 16 + 1)


def test_resample_model_host_logic_with_oracle():
    cam = Camera(0, W, H, 20, 10, W - 31, H - 21, 8, 6)
    grid0 = grid_fit.initialize_grid_from_dense_model(cam, dense_pinhole(260, 250, 330, 235))
    new_cam, new_grid, rep = grid_fit.resample_model(cam, grid0, 14, 11, fit_fn=oracle_fit, unproject_fn=oracle_unproject)
    assert (new_cam.grid_w, new_cam.grid_h) == (14, 11) and new_grid.shape == (14 * 11, 3)
    # the finer model reproduces the coarse model's un-projections inside the calibrated area
    px = np.random.default_rng(3).uniform([25, 15], [W - 35, H - 25], size=(300, 2))
    a, oka = orc.unproject(cam, grid0, px); b, okb = orc.unproject(new_cam, new_grid, px)
    assert oka.all() and okb.all()
    assert np.abs(a[:, :3] - b[:, :3]).max() < 2e-3


def _gpu_case(seed, gw, gh, n):
    cam = Camera(0, W, H, 0, 0, W - 1, H - 1, gw, gh)
    rng = np.random.default_rng(seed)
    grid = grid_fit.initialize_grid_from_dense_model(cam, dense_pinhole(240, 240, 320, 240))
    grid = grid + rng.normal(0, 2e-3, grid.shape); grid /= np.linalg.norm(grid, axis=1, keepdims=True)
    px = rng.uniform([0, 0], [W, H], size=(n, 2))
    gp = np.stack(grid_fit.pixel_corner_conv_to_grid_point(cam, px[:, 0], px[:, 1]), 1)
    dirs = pinhole_dirs(px, 250, 245, 310, 255)
    return cam, grid, gp, dirs


@pytest.mark.gpu
@pytest.mark.parametrize("gw,gh,n,iters", [(8, 6, 600, 4), (20, 15, 5000, 3), (34, 26, 20000, 2)])
def test_gpu_fit_matches_oracle(gw, gh, n, iters):
    from camera_calibration_amd import engine as eng
    cam, grid, gp, dirs = _gpu_case(7, gw, gh, n)
    g_ref, r_ref = oracle_fit(cam, grid, gp, dirs, iters)
    g_gpu, r_gpu = eng.fit_grid_to_directions(cam, grid, gp, dirs, iters)
    assert r_gpu["iterations"] == r_ref["iterations"]
    assert abs(r_gpu["initial_cost"] - r_ref["initial_cost"]) <= 1e-10 * r_ref["initial_cost"]
    assert abs(r_gpu["final_cost"] - r_ref["final_cost"]) <= 1e-8 * max(r_ref["final_cost"], 1e-12)
    assert abs(r_gpu["final_lambda"] - r_ref["final_lambda"]) <= 1e-6 * r_ref["final_lambda"]
    np.testing.assert_allclose(g_gpu, g_ref, atol=1e-9)


@pytest.mark.gpu
def test_gpu_fit_rejects_samples_outside_the_grid():
    from camera_calibration_amd import engine as eng
    cam, grid, gp, dirs = _gpu_case(8, 8, 6, 50)
    gp[3] = [-5.0, 2.0]
    with pytest.raises(eng.EngineError):
        eng.fit_grid_to_directions(cam, grid, gp, dirs, 2)


@pytest.mark.gpu
def test_gpu_reference_known_answer_model_optimization():
    """The reference's TestModelOptimization through the HIP fit (host mirror grid_fit with its default backend)."""
    cam = Camera(0, W, H, 0, 0, W - 1, H - 1, 8, 6)
    dense = dense_pinhole(240, 240, 320, 240)
    grid, rep = grid_fit.fit_to_dense_model(cam, dense, 2)
    assert max_unprojection_cost(cam, grid, dense) < 5e-4
    X, Y = np.meshgrid(np.arange(0, W, 10) + 0.5, np.arange(0, H, 10) + 0.5)
    px = np.stack([X.ravel(), Y.ravel()], 1)
    grid2, rep2 = grid_fit.fit_to_pixel_directions(cam, grid, px, pinhole_dirs(px, 240, 240, 310, 260), 10)
    assert max_unprojection_cost(cam, grid2, dense_pinhole(240, 240, 310, 260)) < 5e-4


@pytest.mark.gpu
def test_cpp_mirror_fit_and_resample_match_python_mirror():
    """CentralGenericModel::FitToDenseModel and ResampleModel of the C++ host mirror against the Python mirror
    (same engine underneath; the host logic -- initialisation, sampling, sub-sample step -- must agree)."""
    import ctypes as C
    import os
    from camera_calibration_amd import engine as eng
    cam = Camera(0, W, H, 10, 5, W - 21, H - 11, 9, 7)
    dense = dense_pinhole(255, 250, 325, 238)
    dense[:30, :] = np.nan
    eng.load()
    L = C.CDLL(os.path.join(os.path.dirname(eng.LIB_PATH), "libcalib_ba_host_test.so"))
    dp = C.POINTER(C.c_double)
    cs = eng._cam_struct(cam)
    g_out = np.zeros((cam.grid_points, 3)); r_out = np.zeros((13 * 10, 3))
    dflat = np.ascontiguousarray(dense.reshape(-1, 3))
    rc = L.cba_host_fit_and_resample(C.byref(cs), C.c_int(W), C.c_int(H), dflat.ctypes.data_as(dp), C.c_int(3), C.c_int(4),
                                     g_out.ctypes.data_as(dp), C.c_int(13), C.c_int(10), r_out.ctypes.data_as(dp))
    assert rc == 0
    g_py, rep = grid_fit.fit_to_dense_model(cam, dense, 3, 4)
    np.testing.assert_allclose(g_out, g_py, atol=1e-12)
    new_cam, r_py, _ = grid_fit.resample_model(cam, g_py, 13, 10)
    np.testing.assert_allclose(r_out, r_py, atol=1e-10)


def test_noncentral_resampling_and_initialisation():
    from camera_calibration_amd.problem import NONCENTRAL_GENERIC
    cam = Camera(0, W, H, 0, 0, W - 1, H - 1, 8, 6)
    grid = grid_fit.initialize_grid_from_dense_model(cam, dense_pinhole(240, 240, 320, 240))
    nc, grids = grid_fit.initialize_noncentral_from_central(cam, grid)
    assert nc.model_type == NONCENTRAL_GENERIC and grids.shape == (2, 48, 3) and not grids[1].any()
    np.testing.assert_array_equal(grids[0], grid)
    rng = np.random.default_rng(2)
    grids[1] = rng.normal(0, 1e-3, grids[1].shape)
    new_cam, g2 = grid_fit.resample_noncentral_model(nc, grids, 12, 9)
    assert (new_cam.grid_w, new_cam.grid_h) == (12, 9) and g2.shape == (2, 108, 3)
    # scalar restatement of one interior and one clamped grid point (calibration.cc:392-412, LV/image.h:152-176)
    f = np.float32
    for (x, y) in [(5, 4), (0, 0), (11, 8)]:
        px = float(f(0) + ((f(x) - f(1)) / (f(12) - f(3))) * f(W)); py = float(f(0) + ((f(y) - f(1)) / (f(9) - f(3))) * f(H))
        ogx = 1.0 + float(f(8) - f(3)) * px / W; ogy = 1.0 + float(f(6) - f(3)) * py / H
        ogx = min(max(ogx, 0.0), 8 - 1.001); ogy = min(max(ogy, 0.0), 6 - 1.001)
        ix, iy = int(ogx), int(ogy)
        fx, fy = f(ogx - ix), f(ogy - iy)
        G = grids[1].reshape(6, 8, 3)
        ref = (float((f(1) - fx) * (f(1) - fy)) * G[iy, ix] + float(fx * (f(1) - fy)) * G[iy, ix + 1] +
               float((f(1) - fx) * fy) * G[iy + 1, ix] + float(fx * fy) * G[iy + 1, ix + 1])
        np.testing.assert_allclose(g2[1].reshape(9, 12, 3)[y, x], ref, rtol=1e-15, atol=1e-18)


def test_cpp_mirror_noncentral_helpers_match_python_mirror():
    """NoncentralGenericModel::InitializeFromCentralGenericModel / Scale (noncentral_generic.cc:136-154) and the non-central
    branch of ResampleModel (calibration.cc:386-425) of the C++ host mirror against the Python mirror -- host code only."""
    import ctypes as C
    from camera_calibration_amd import build, engine as eng
    L = C.CDLL(build.build_host_test())
    cam = Camera(0, W, H, 0, 0, W - 1, H - 1, 8, 6)
    grid = grid_fit.initialize_grid_from_dense_model(cam, dense_pinhole(240, 240, 320, 240))
    nc, grids = grid_fit.initialize_noncentral_from_central(cam, grid)
    rng = np.random.default_rng(2)
    pts = rng.normal(0, 1e-3, grids[1].shape)
    grids[1] = 2.5 * pts
    new_cam, g2 = grid_fit.resample_noncentral_model(nc, grids, 12, 9)
    cs = eng._cam_struct(cam)
    dp = C.POINTER(C.c_double)
    init_out = np.zeros(6 * 48); res_out = np.zeros(6 * 108)
    gflat = np.ascontiguousarray(grid, dtype=np.float64).ravel(); pflat = np.ascontiguousarray(pts).ravel()
    L.cba_host_noncentral_init_and_resample.argtypes = [C.c_void_p, dp, dp, C.c_double, dp, C.c_int, C.c_int, dp]
    rc = L.cba_host_noncentral_init_and_resample(C.byref(cs), gflat.ctypes.data_as(dp), pflat.ctypes.data_as(dp), 2.5,
                                                 init_out.ctypes.data_as(dp), 12, 9, res_out.ctypes.data_as(dp))
    assert rc == 0
    np.testing.assert_array_equal(init_out.reshape(2, 48, 3)[0], grids[0])
    np.testing.assert_allclose(init_out.reshape(2, 48, 3)[1], grids[1], rtol=1e-16, atol=0)
    np.testing.assert_allclose(res_out.reshape(2, 108, 3), g2, rtol=1e-15, atol=1e-18)

