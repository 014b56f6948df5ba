"""Calibration report statistics on the HIP projection kernel.

Mirrors the reference's report code for the numbers a user reads off `report_cameraN_info.txt`
(APP = applications/camera_calibration/src/camera_calibration):

* ``compute_all_reprojection_errors``  -- ``ComputeAllReprojectionErrors``, APP/calibration_report.cc:101-148:
  every feature of one camera over the used imagesets is projected with ``CameraModel::Project`` (from the
  centre of the calibrated area -- no warm start, unlike the optimisation passes); error = pixel - xy;
  failed projections are skipped.  Returns count, sum and maximum of the error magnitudes, the errors
  and the features, in the reference's order.
* ``reprojection_error_histogram``     -- ``ComputeReprojectionErrorHistogram``, :151-168 (incl. its float
  literals and the truncation-with-fix-up rounding of the bin index).
* ``reprojection_error_summary``       -- the ``reprojection_error_*`` lines written at :676-693
  (average = sum / count, maximum, median = sorted magnitudes[size / 2]).

* ``delete_outlier_features``          -- ``DeleteOutlierFeatures``, APP/calibration.cc:62-184:
  quartile rule on the sorted error magnitudes, features that fail to project or exceed
  ``q3 + factor * (q3 - q1)`` are removed, imagesets left with fewer than 3 features of the camera become
  unused.

The projections run on the GPU through the C-ABI (``cba_project``); the few reductions are host code,
as in the reference.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np

from . import engine as _engine
from .problem import Problem, State
from .se3 import quat_to_matrix, se3_mul


def compute_all_reprojection_errors(camera_index: int, problem: Problem, state: State,
                                    project_fn: Optional[Callable] = None, device: int = 0) -> Dict[str, object]:
    project_fn = project_fn or (lambda cam, grid, pts: _engine.project(cam, grid, pts, device=device))
    sel = problem.obs_camera == camera_index
    img = problem.obs_image[sel]
    xy = problem.obs_xy[sel]
    # image_tr_global(camera, imageset) = camera_tr_rig[camera] * rig_tr_global[imageset]  (ba_state.h)
    itg = se3_mul(state.camera_tr_rig[camera_index][None, :], state.rig_tr_global)
    R = quat_to_matrix(itg[:, :4])
    pts = state.points[problem.obs_point[sel]]
    local = np.einsum("nij,nj->ni", R[img], pts) + itg[img, 4:]
    pixels, ok = project_fn(problem.cameras[camera_index], state.grids[camera_index], local)
    ok = np.asarray(ok, dtype=bool)
    errors = (np.asarray(pixels)[ok] - xy[ok].astype(np.float64))
    mags = np.sqrt(errors[:, 0] ** 2 + errors[:, 1] ** 2)
    return dict(count=int(ok.sum()), sum=float(mags.sum()), max=float(mags.max()) if mags.size else 0.0,
                errors=errors, features=xy[ok], ok=ok)


def reprojection_error_histogram(resolution: int, extent_in_px: float, errors: np.ndarray) -> np.ndarray:
    """hist[hy, hx] (Image<double>(x, y) is row-major in y)."""
    hist = np.zeros((resolution, resolution))
    e = np.asarray(errors, dtype=np.float64).reshape(-1, 2)
    half = np.float32(0.5) * np.float32(1.0)          # the literals are floats; 0.5f and 1.f are exact
    hx_f = resolution * float(half) * ((e[:, 0] / extent_in_px) + 1.0)
    hy_f = resolution * float(half) * ((e[:, 1] / extent_in_px) + 1.0)
    hx = np.trunc(hx_f).astype(np.int64) - (hx_f < 0)
    hy = np.trunc(hy_f).astype(np.int64) - (hy_f < 0)
    inside = (hx >= 0) & (hy >= 0) & (hx < resolution) & (hy < resolution)
    np.add.at(hist, (hy[inside], hx[inside]), 1.0)
    return hist


def reprojection_error_summary(res: Dict[str, object]) -> Dict[str, float]:
    errors = np.asarray(res["errors"]).reshape(-1, 2)
    mags = np.sort(np.sqrt(errors[:, 0] ** 2 + errors[:, 1] ** 2))
    n = int(res["count"])
    return dict(reprojection_error_count=n,
                reprojection_error_average=float(res["sum"]) / n if n else float("nan"),
                reprojection_error_maximum=float(res["max"]),
                reprojection_error_median=float(mags[mags.size // 2]) if mags.size else float("nan"))


def delete_outlier_features(camera_index: int, problem: Problem, state: State, outlier_removal_factor: float,
                            image_used: Optional[np.ndarray] = None, project_fn: Optional[Callable] = None,
                            device: int = 0):
    """Returns (keep mask over the problem's observations, new image_used, outlier_threshold or None).

    `image_used` (bool per imageset, default all used) is the reference's BAState::image_used restricted to the
    problem's imagesets; observations of unused imagesets are neither measured nor removed."""
    project_fn = project_fn or (lambda cam, grid, pts: _engine.project(cam, grid, pts, device=device))
    used = np.ones(problem.n_images, dtype=bool) if image_used is None else np.asarray(image_used, dtype=bool).copy()
    keep = np.ones(problem.n_obs, dtype=bool)
    sel = np.flatnonzero((problem.obs_camera == camera_index) & used[problem.obs_image])
    itg = se3_mul(state.camera_tr_rig[camera_index][None, :], state.rig_tr_global)
    R = quat_to_matrix(itg[:, :4])
    img = problem.obs_image[sel]
    local = np.einsum("nij,nj->ni", R[img], state.points[problem.obs_point[sel]]) + itg[img, 4:]
    pixels, ok = project_fn(problem.cameras[camera_index], state.grids[camera_index], local)
    ok = np.asarray(ok, dtype=bool)
    e = np.asarray(pixels) - problem.obs_xy[sel].astype(np.float64)
    mags = np.sqrt(e[:, 0] ** 2 + e[:, 1] ** 2)
    valid = np.sort(mags[ok])
    if valid.size < 8:                      # "arbitrary threshold", calibration.cc:97
        return keep, used, None
    n = valid.size
    q1 = valid[int(np.float32(0.25) * np.float32(n) + np.float32(0.5))]     # float index arithmetic as in :104-105
    q3 = valid[int(np.float32(0.75) * np.float32(n) + np.float32(0.5))]
    threshold = q3 + float(np.float32(outlier_removal_factor)) * (q3 - q1)
    remove = (~ok) | (mags > threshold)
    keep[sel[remove]] = False
    # imagesets with fewer than 3 remaining features of this camera become unused (:165-167)
    remaining = np.bincount(problem.obs_image[sel[~remove]], minlength=problem.n_images)
    touched = used.copy()
    used[touched & (remaining < 3)] = False
    return keep, used, float(threshold)
