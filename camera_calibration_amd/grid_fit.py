"""Grid-only fitting of the central-generic model.

Host mirror of (APP = applications/camera_calibration/src/camera_calibration):

* ``fit_to_pixel_directions``  -- CentralGenericModel::FitToPixelDirections, APP/models/central_generic.cc:424-431:
  pixels -> grid points, then the LM fit (``FitToPixelDirectionsImpl`` :551-568) on the GPU through the C-ABI
  (``cba_fit_grid_to_directions``);
* ``fit_to_dense_model``       -- CentralGenericModel::FitToDenseModel, :267-422: grid initialisation from the dense
  direction image (closest valid pixel, ring search of radius < 5, neighbour extrapolation for the rest),
  sub-sampled (grid point, direction) samples over the calibrated area, then the same LM fit;
* ``resample_model``           -- ResampleModel for central-generic source and target, APP/calibration.cc:373-529:
  dense model by un-projecting every pixel centre (``cba_unproject``), then ``fit_to_dense_model`` with at most
  300 x 300 samples and 3 iterations.

* ``initialize_noncentral_from_central`` -- NoncentralGenericModel::InitializeFromCentralGenericModel,
  APP/models/noncentral_generic.cc:136-146; ``resample_noncentral_model`` -- the non-central -> non-central branch of
  ResampleModel (calibration.cc:386-425): bilinear re-gridding of both grids (libvis InterpolateBilinear, float
  weights), no optimisation.

`fit_fn(cam, grid, grid_points, directions, max_iteration_count) -> (grid, report)` is injectable so that the tests
can run the host logic against the oracle; the default is the HIP engine (no CPU fallback).
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from . import engine as _engine
from .problem import CENTRAL_GENERIC, Camera


def _default_fit(cam, grid, grid_points, directions, max_iteration_count):
    return _engine.fit_grid_to_directions(cam, grid, grid_points, directions, max_iteration_count)


def pixel_corner_conv_to_grid_point(cam: Camera, x, y):
    """central_grid.h:150-154; the literals 1.f / 3.f are floats but exactly representable."""
    x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    gx = 1.0 + float(np.float32(cam.grid_w) - np.float32(3.0)) * (x - cam.calib_min_x) / (cam.calib_max_x + 1 - cam.calib_min_x)
    gy = 1.0 + float(np.float32(cam.grid_h) - np.float32(3.0)) * (y - cam.calib_min_y) / (cam.calib_max_y + 1 - cam.calib_min_y)
    return gx, gy


def grid_point_to_pixel_corner_conv(cam: Camera, gx: int, gy: int):
    """central_grid.h:127-131: the int overload evaluates the whole expression in float."""
    f = np.float32
    px = f(cam.calib_min_x) + ((f(gx) - f(1.0)) / (f(cam.grid_w) - f(3.0))) * f(cam.calib_max_x + 1 - cam.calib_min_x)
    py = f(cam.calib_min_y) + ((f(gy) - f(1.0)) / (f(cam.grid_h) - f(3.0))) * f(cam.calib_max_y + 1 - cam.calib_min_y)
    return float(px), float(py)


def fit_to_pixel_directions(cam: Camera, grid: np.ndarray, pixels: np.ndarray, directions: np.ndarray,
                            max_iteration_count: int, fit_fn: Optional[Callable] = None):
    assert cam.model_type == CENTRAL_GENERIC
    px = np.asarray(pixels, dtype=np.float64).reshape(-1, 2)
    gx, gy = pixel_corner_conv_to_grid_point(cam, px[:, 0], px[:, 1])
    return (fit_fn or _default_fit)(cam, grid, np.stack([gx, gy], 1), directions, max_iteration_count)


def initialize_grid_from_dense_model(cam: Camera, dense_model: np.ndarray):
    """First half of FitToDenseModel (:267-392).  dense_model: (H, W, 3) with NaN for invalid pixels.
    Returns the (G,3) grid or None if some grid points could not be initialised."""
    dh, dw = dense_model.shape[:2]
    scale_x = dw / float(cam.width); scale_y = dh / float(cam.height)
    gw, gh = cam.grid_w, cam.grid_h
    grid = np.full((gh, gw, 3), np.nan)
    valid = ~np.isnan(dense_model[:, :, 0])
    for gy in range(gh):
        for gx in range(gw):
            px, py = grid_point_to_pixel_corner_conv(cam, gx, gy)
            cx, cy = int(scale_x * px), int(scale_y * py)          # .cast<int>() truncates
            if cx < 0 or cy < 0 or cx >= dw or cy >= dh:
                continue
            if valid[cy, cx]:
                grid[gy, gx] = dense_model[cy, cx]
                continue
            found = False
            for radius in range(1, 5):
                x0, x1, y0, y1 = cx - radius, cx + radius, cy - radius, cy + radius
                for x in range(max(0, x0), min(dw - 1, x1) + 1):      # top and bottom
                    if y0 >= 0 and valid[y0, x]:
                        grid[gy, gx] = dense_model[y0, x]; found = True; break
                    if y1 < dh and valid[y1, x]:
                        grid[gy, gx] = dense_model[y1, x]; found = True; break
                if found:
                    break
                for y in range(max(0, y0), min(dh - 1, y1) + 1):      # left and right
                    if x0 >= 0 and valid[y, x0]:
                        grid[gy, gx] = dense_model[y, x0]; found = True; break
                    if x1 < dw and valid[y, x1]:
                        grid[gy, gx] = dense_model[y, x1]; found = True; break
                if found:
                    break
    # fill the rest by linear steps from neighbours, in place, in row-major order (:340-386)
    have_nan = bool(np.isnan(grid[:, :, 0]).any())
    it = 0
    while have_nan and it < dw + dh:
        have_nan = False
        for gy in range(gh):
            for gx in range(gw):
                if not np.isnan(grid[gy, gx]).any():
                    continue
                s = np.zeros(3); count = 0
                for dx, dy in ((0, 1), (0, -1), (1, 0), (-1, 0)):
                    nx2, ny2 = gx + 2 * dx, gy + 2 * dy
                    if nx2 < 0 or ny2 < 0 or nx2 >= gw or ny2 >= gh:
                        continue
                    v1, v2 = grid[gy + dy, gx + dx], grid[ny2, nx2]
                    if np.isnan(v1).any() or np.isnan(v2).any():
                        continue
                    s += v1 + (v1 - v2); count += 1
                if count > 0:
                    grid[gy, gx] = s / np.linalg.norm(s)
                else:
                    have_nan = True
        it += 1
    if have_nan:
        return None
    return grid.reshape(-1, 3)


def dense_model_samples(cam: Camera, dense_model: np.ndarray, subsample_step: int):
    """Second half of FitToDenseModel (:394-417): (grid point, direction) samples over the calibrated area."""
    dh, dw = dense_model.shape[:2]
    scale_x = dw / float(cam.width); scale_y = dh / float(cam.height)
    m2c_x = float(cam.width) / dw; m2c_y = float(cam.height) / dh
    ys = np.arange(cam.calib_min_y, cam.calib_max_y + 1, subsample_step)
    xs = np.arange(cam.calib_min_x, cam.calib_max_x + 1, subsample_step)
    X, Y = np.meshgrid(xs, ys)                       # y outer, x inner, as the reference loops
    mx = (scale_x * X.ravel()).astype(np.int64); my = (scale_y * Y.ravel()).astype(np.int64)
    meas = dense_model[my, mx]
    keep = ~np.isnan(meas).any(axis=1)
    half = float(np.float32(0.5))
    gx, gy = pixel_corner_conv_to_grid_point(cam, m2c_x * (mx[keep] + half), m2c_y * (my[keep] + half))
    return np.stack([gx, gy], 1), meas[keep]


def fit_to_dense_model(cam: Camera, dense_model: np.ndarray, subsample_step: int, max_iteration_count: int = 10,
                       fit_fn: Optional[Callable] = None):
    """Returns (grid, report) or (None, None) when the initialisation fails (the reference returns false)."""
    assert cam.model_type == CENTRAL_GENERIC
    grid = initialize_grid_from_dense_model(cam, dense_model)
    if grid is None:
        return None, None
    gp, dirs = dense_model_samples(cam, dense_model, subsample_step)
    return (fit_fn or _default_fit)(cam, grid, gp, dirs, max_iteration_count)


def resample_model(cam: Camera, grid: np.ndarray, target_resolution_x: int, target_resolution_y: int,
                   fit_fn: Optional[Callable] = None, unproject_fn: Optional[Callable] = None):
    """ResampleModel, central-generic -> central-generic (calibration.cc:373-529).  Returns (new Camera, grid, report)."""
    unproject_fn = unproject_fn or (lambda c, g, px: _engine.unproject(c, g, px))
    X, Y = np.meshgrid(np.arange(cam.width) + 0.5, np.arange(cam.height) + 0.5)
    lines, ok = unproject_fn(cam, grid, np.stack([X.ravel(), Y.ravel()], 1))
    dense = np.where(np.asarray(ok, dtype=bool)[:, None], np.asarray(lines)[:, :3], np.nan).reshape(cam.height, cam.width, 3)
    new_cam = Camera(CENTRAL_GENERIC, cam.width, cam.height, cam.calib_min_x, cam.calib_min_y, cam.calib_max_x, cam.calib_max_y,
                     target_resolution_x, target_resolution_y)
    aw = cam.calib_max_x - cam.calib_min_x + 1; ah = cam.calib_max_y - cam.calib_min_y + 1
    # std::round(int / int): the integer division happens first (calibration.cc:452-453)
    step = max(1, min(int(round(aw // 300)), int(round(ah // 300))))
    new_grid, rep = fit_to_dense_model(new_cam, dense, step, 3, fit_fn)
    return new_cam, new_grid, rep


def initialize_noncentral_from_central(cam: Camera, grid: np.ndarray):
    """Returns (non-central Camera, grids (2,G,3) = direction grid, zero point grid)."""
    from .problem import NONCENTRAL_GENERIC
    g = np.asarray(grid, dtype=np.float64).reshape(-1, 3)
    nc = Camera(NONCENTRAL_GENERIC, cam.width, cam.height, cam.calib_min_x, cam.calib_min_y, cam.calib_max_x, cam.calib_max_y,
                cam.grid_w, cam.grid_h)
    return nc, np.stack([g, np.zeros_like(g)])


def _interpolate_bilinear(img: np.ndarray, x: float, y: float) -> np.ndarray:
    """libvis Image::InterpolateBilinear for Vec3d pixels (LV/image.h:152-176): the fractions are floats."""
    ix, iy = int(x), int(y)
    fx = np.float32(x - ix); fy = np.float32(y - iy)
    fxi = np.float32(1.0) - fx; fyi = np.float32(1.0) - fy
    return (float(fxi * fyi) * img[iy, ix] + float(fx * fyi) * img[iy, ix + 1] +
            float(fxi * fy) * img[iy + 1, ix] + float(fx * fy) * img[iy + 1, ix + 1])


def resample_noncentral_model(cam: Camera, grids: np.ndarray, target_resolution_x: int, target_resolution_y: int):
    """ResampleModel, non-central generic -> non-central generic (calibration.cc:386-425).  Returns (Camera, grids)."""
    from .problem import NONCENTRAL_GENERIC
    assert cam.model_type == NONCENTRAL_GENERIC
    g = np.asarray(grids, dtype=np.float64).reshape(2, cam.grid_h, cam.grid_w, 3)
    new_cam = Camera(NONCENTRAL_GENERIC, cam.width, cam.height, cam.calib_min_x, cam.calib_min_y, cam.calib_max_x, cam.calib_max_y,
                     target_resolution_x, target_resolution_y)
    out = np.zeros((2, target_resolution_y, target_resolution_x, 3))
    f = np.float32
    for y in range(target_resolution_y):
        for x in range(target_resolution_x):
            # static GridPointToPixelCornerConv (central_grid.h:132-140): evaluated in float
            px = float(f(cam.calib_min_x) + ((f(x) - f(1.0)) / (f(target_resolution_x) - f(3.0))) * f(cam.calib_max_x + 1 - cam.calib_min_x))
            py = float(f(cam.calib_min_y) + ((f(y) - f(1.0)) / (f(target_resolution_y) - f(3.0))) * f(cam.calib_max_y + 1 - cam.calib_min_y))
            ogx, ogy = pixel_corner_conv_to_grid_point(cam, px, py)      # same formula in the non-central model
            ogx = min(max(float(ogx), 0.0), cam.grid_w - 1.001); ogy = min(max(float(ogy), 0.0), cam.grid_h - 1.001)
            out[1, y, x] = _interpolate_bilinear(g[1], ogx, ogy)          # point grid
            out[0, y, x] = _interpolate_bilinear(g[0], ogx, ogy)          # direction grid (not re-normalised, as in the reference)
    return new_cam, out.reshape(2, -1, 3)
