"""SURVEY 8f row F3 pinned to the REFERENCE'S OWN CODE: CentralGenericModel::FitToPixelDirections and FitToDenseModel
(APP/models/central_generic.cc:267-431 and the LMOptimizer run behind them), compiled whole into oracle/_ref/libcalibref_ba.so
(oracle/ref_ba_glue.cc), against the oracle's grid-only LM (orc_fit_grid_*) and the product's host logic
(camera_calibration_amd.grid_fit: initialisation from the dense model, neighbour fill, sample selection) with the oracle as the fit.
Rounds 1-4 pinned this row through the known answers of the reference's TestModelOptimization only."""
import numpy as np
import pytest

from camera_calibration_amd import grid_fit
from camera_calibration_amd.problem import Camera
from oracle import oracle as orc
from oracle import ref

pytestmark = pytest.mark.skipif(not ref.ba_available(), reason="oracle/_ref/libcalibref_ba.so not built (needs /root/reference)")

W, H = 320, 240


def _dirs(px, fx, fy, cx, cy, k1=0.0):
    x = (px[..., 0] - cx) / fx; y = (px[..., 1] - cy) / fy
    r2 = x * x + y * y
    d = np.stack([x * (1 + k1 * r2), y * (1 + k1 * r2), np.ones_like(x)], -1)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def _dense(fx, fy, cx, cy, k1=0.0):
    X, Y = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)
    return _dirs(np.stack([X, Y], -1), fx, fy, cx, cy, k1)


def _oracle_fit(cam, grid, gp, dirs, iters):
    return orc.fit_grid_to_points(cam.grid_w, cam.grid_h, grid, gp, dirs, iters)


@pytest.mark.parametrize("gw,gh,n,iters", [(8, 6, 400, 1), (8, 6, 400, 4), (12, 9, 1500, 3)])
def test_fit_to_pixel_directions_is_the_references(gw, gh, n, iters):
    cam = Camera(0, W, H, 3, 2, W - 5, H - 4, gw, gh)
    rng = np.random.default_rng(gw + iters)
    grid0 = grid_fit.initialize_grid_from_dense_model(cam, _dense(150, 150, 160, 120))
    px = rng.uniform([cam.calib_min_x, cam.calib_min_y], [cam.calib_max_x + 1, cam.calib_max_y + 1], size=(n, 2))
    dirs = _dirs(px, 150, 152, 157, 124, k1=-0.08)                     # a different camera than the grid was initialised from
    g_ref = ref.f3_fit_to_pixel_directions(cam, grid0, px, dirs, iters)
    out = grid_fit.fit_to_pixel_directions(cam, grid0, px, dirs, iters, fit_fn=_oracle_fit)
    g_orc = np.asarray(out[0] if isinstance(out, tuple) else out).reshape(-1, 3)
    assert np.abs(g_ref - grid0.reshape(-1, 3)).max() > 1e-3             # the fit moved the grid
    np.testing.assert_allclose(g_orc, g_ref, rtol=0, atol=1e-11)            # observed 2e-14 ... 3e-14
    np.testing.assert_allclose(np.linalg.norm(g_ref, axis=1), 1.0, atol=1e-12)


def test_fit_to_dense_model_is_the_references():
    """Initialisation (closest valid pixel, ring search, neighbour fill in place) + sample selection + LM, with holes in the dense model:
    (a) the initial grid (0 LM iterations) equal to the last bit, (b) the fitted grid."""
    cam = Camera(0, W, H, 0, 0, W - 1, H - 1, 10, 8)
    dense = _dense(140, 141, 161, 119, k1=-0.1)
    dense[:14, :, :] = np.nan                                             # a band the ring search (radius < 5) cannot bridge: neighbour fill
    dense[100:104, 200:203, :] = np.nan                                   # a hole the ring search bridges
    dense[:, -9:, :] = np.nan
    g0_ref = ref.f3_fit_to_dense_model(cam, dense, 4, 0)
    g0 = grid_fit.initialize_grid_from_dense_model(cam, dense)
    assert g0_ref is not None and g0 is not None
    np.testing.assert_allclose(g0.reshape(-1, 3), g0_ref, rtol=0, atol=2e-16)
    g_ref = ref.f3_fit_to_dense_model(cam, dense, 4, 3)
    out = grid_fit.fit_to_dense_model(cam, dense, 4, 3, fit_fn=_oracle_fit)
    g_orc = np.asarray(out[0]).reshape(-1, 3)
    assert np.abs(g_ref - g0_ref).max() > 1e-4
    np.testing.assert_allclose(g_orc, g_ref, rtol=0, atol=1e-11)


def test_fit_to_dense_model_failure_is_the_references():
    """A dense model that is valid only in one corner: grid points stay undefined after the fill -> the reference returns false."""
    cam = Camera(0, W, H, 0, 0, W - 1, H - 1, 10, 8)
    dense = np.full((H, W, 3), np.nan)
    dense[:20, :20] = _dense(140, 140, 160, 120)[:20, :20]
    assert ref.f3_fit_to_dense_model(cam, dense, 4, 1) is None
    assert grid_fit.fit_to_dense_model(cam, dense, 4, 1, fit_fn=_oracle_fit)[0] is None


# ---- ResampleModel (APP/calibration.cc:373-528), piped into the same library ----
def _oracle_unproject(cam, grid, px):
    return orc.unproject(cam, grid, px)


def test_resample_model_central_to_central_is_the_references():
    """Dense direction image from the old model (pixel centres; NaN outside its calibrated area), the subsample step
    max(1, min(round(int / 300), round(int / 300))), FitToDenseModel(.., 3) on the target grid."""
    cam = Camera(0, W, H, 12, 8, W - 15, H - 11, 8, 6)
    grid0 = grid_fit.initialize_grid_from_dense_model(cam, _dense(150, 149, 158, 121, k1=-0.05))
    g_ref = ref.f3_resample_model(cam, grid0, 0, 13, 10)
    new_cam, g_host, _ = grid_fit.resample_model(cam, grid0, 13, 10, fit_fn=_oracle_fit, unproject_fn=_oracle_unproject)
    assert g_ref is not None and (new_cam.grid_w, new_cam.grid_h) == (13, 10)
    np.testing.assert_allclose(np.asarray(g_host).reshape(-1, 3), g_ref, rtol=0, atol=1e-11)


def test_resample_model_central_to_noncentral_is_the_references():
    """Central source, non-central target: the fitted central grid becomes the direction grid, the point grid is zero
    (InitializeFromCentralGenericModel, noncentral_generic.cc:136-146)."""
    cam = Camera(0, W, H, 0, 0, W - 1, H - 1, 8, 6)
    grid0 = grid_fit.initialize_grid_from_dense_model(cam, _dense(150, 150, 160, 120))
    g_ref = ref.f3_resample_model(cam, grid0, 1, 10, 8)
    new_cam, g_c, _ = grid_fit.resample_model(cam, grid0, 10, 8, fit_fn=_oracle_fit, unproject_fn=_oracle_unproject)
    nc, g_host = grid_fit.initialize_noncentral_from_central(new_cam, g_c)
    assert g_ref is not None and nc.model_type == 1
    np.testing.assert_allclose(g_host[0], g_ref[0], rtol=0, atol=1e-11)
    np.testing.assert_array_equal(g_ref[1], 0.0)
    np.testing.assert_array_equal(g_host[1], 0.0)


def test_resample_model_noncentral_to_noncentral_is_the_references():
    """Bilinear resampling of both grids (float fractions, clamped to the old grid, directions not re-normalised; calibration.cc:386-425)."""
    from camera_calibration_amd.problem import NONCENTRAL_GENERIC
    cam = Camera(NONCENTRAL_GENERIC, W, H, 5, 3, W - 7, H - 4, 8, 6)
    rng = np.random.default_rng(4)
    d = grid_fit.initialize_grid_from_dense_model(Camera(0, W, H, 5, 3, W - 7, H - 4, 8, 6), _dense(150, 150, 160, 120))
    grids = np.stack([d, rng.normal(0, 1e-3, d.shape)])
    for (gw, gh) in ((12, 9), (5, 4), (8, 6)):
        g_ref = ref.f3_resample_model(cam, grids, 1, gw, gh)
        new_cam, g_host = grid_fit.resample_noncentral_model(cam, grids, gw, gh)
        assert g_ref is not None and (new_cam.grid_w, new_cam.grid_h) == (gw, gh)
        # Equal to 1e-15 when the reference code is compiled with -ffp-contract=off (checked).  With g++'s default (and the reference's own
        # -O2 -march=native build) the FLOAT expression of the static GridPointToPixelCornerConv, min + ((x - 1.f) / (gw - 3.f)) * span
        # (central_grid.h:132-140), becomes one fused multiply-add: one float ulp in the pixel of some grid columns, 3e-8 in their
        # interpolated values (observed: column 0 only) -- an initial state for the optimisation that follows, as the reference notes.
        np.testing.assert_allclose(g_host, g_ref, rtol=0, atol=1e-7)
        assert (np.abs(g_host - g_ref).reshape(2, gh, gw, 3).max(axis=(0, 3)) > 1e-14).sum() <= gh            # at most one column
    # a non-central source cannot be resampled into a central target: the reference returns false
    assert ref.f3_resample_model(cam, grids, 0, 8, 6) is None

