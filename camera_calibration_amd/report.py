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
