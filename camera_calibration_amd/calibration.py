"""Outer-loop helpers of RunBundleAdjustment / Calibrate.

Mirrors (APP = applications/camera_calibration/src/camera_calibration):

* ``choose_nice_camera_orientation`` -- CentralGenericModel::ChooseNiceCameraOrientation, APP/models/central_generic.cc:570-621:
  rotate the model so that the image centre looks along +z and the mean direction of a 21-row band right of the
  centre has maximal +x; the un-projections run on the GPU (``cba_unproject``); returns the rotation and the rotated
  grid (``Rotate``, central_grid.h:70-76); the caller left-multiplies ``camera_tr_rig`` (calibration.cc:248-254).
* ``scale_to_metric``                -- ScaleToMetric, APP/calibration.cc:307-370 + BAState::ScaleState, ba_state.cc:60-76.
* ``run_bundle_adjustment``          -- RunBundleAdjustment, APP/calibration.cc:187-304: OptimizeJointly(1) per iteration with the
  state device-resident, camera orientations beautified after every iteration, stop when
  ``cost >= last_cost - cost_reduction_threshold``.
* ``compute_grid_resolution`` / ``calc_grid_resolution_for_level`` -- the grid-resolution rules of the pyramid, APP/calibration.cc:531-568.
* ``calibrate_refinement_stage``     -- the refinement stage of Calibrate(), APP/calibration.cc:1030-1142: the pyramid levels (two
  RunBundleAdjustment runs and a ResampleModel per level), the optional outlier stage, the main bundle adjustment and ScaleToMetric.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np

from . import engine as _engine
from .calibration_io import DatasetData
from .problem import CENTRAL_GENERIC, Camera, Problem, State
from .se3 import se3_mul


def _quat_from_two_vectors(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Eigen Quaternion::FromTwoVectors (Geometry/Quaternion.h): unit inputs, the general (non-antiparallel) branch.
    Returns the 3x3 rotation matrix."""
    v0 = a / np.linalg.norm(a); v1 = b / np.linalg.norm(b)
    c = float(v1 @ v0)
    if c < -1 + 1e-12:                       # antiparallel: SVD branch of Eigen; any perpendicular axis
        axis = np.cross(v0, [1.0, 0, 0]) if abs(v0[0]) < 0.9 else np.cross(v0, [0, 1.0, 0])
        axis /= np.linalg.norm(axis)
        w, xyz = 0.0, axis
    else:
        axis = np.cross(v0, v1)
        s = np.sqrt((1 + c) * 2)
        w, xyz = s * 0.5, axis / s
    x, y, z = xyz
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def choose_nice_camera_orientation(cam: Camera, grid: np.ndarray, unproject_fn: Optional[Callable] = None):
    """Returns (rotation 3x3, rotated grid (G,3))."""
    assert cam.model_type == CENTRAL_GENERIC
    unproject_fn = unproject_fn or (lambda c, g, px: _engine.unproject(c, g, px))
    w, h = cam.width, cam.height
    half = float(np.float32(0.5))
    right_min_x = min(w - 1, w // 2 + 11); right_max_x = w - 1
    right_min_y = max(0, h // 2 - 10); right_max_y = min(h - 1, h // 2 + 10)
    X, Y = np.meshgrid(np.arange(right_min_x, right_max_x + 1) + half, np.arange(right_min_y, right_max_y + 1) + half)
    px = np.concatenate([[[half * w, half * h]], np.stack([X.ravel(), Y.ravel()], 1)])
    lines, ok = unproject_fn(cam, grid, px)
    ok = np.asarray(ok, dtype=bool)
    forward = lines[0, :3] if ok[0] else np.array([0.0, 0.0, 1.0])
    forward_rotation = _quat_from_two_vectors(forward, np.array([0.0, 0.0, 1.0]))
    sel = ok[1:]
    if sel.any():
        # row-major accumulation order of the reference (y outer, x inner); plain sums are order-insensitive to 1e-16
        right = lines[1:][sel, :3].sum(0) / int(sel.sum())
        fr = forward_rotation @ right
        angle = np.arctan2(-fr[1], fr[0])
        c, s = np.cos(angle), np.sin(angle)
        right_rotation = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    else:
        right_rotation = np.eye(3)
    rotation = right_rotation @ forward_rotation
    return rotation, np.asarray(grid).reshape(-1, 3) @ rotation.T


def rotation_to_pose(rotation: np.ndarray) -> np.ndarray:
    """SE3d(rotation, 0) as qw qx qy qz tx ty tz (Eigen's matrix -> quaternion conversion, Shepperd's branches)."""
    m = rotation
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(m))); j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0) * 2
        q = [0.0, 0.0, 0.0, 0.0]
        q[0] = (m[k, j] - m[j, k]) / s
        q[1 + i] = 0.25 * s; q[1 + j] = (m[j, i] + m[i, j]) / s; q[1 + k] = (m[k, i] + m[i, k]) / s
    q = np.array(q) / np.linalg.norm(q)
    return np.concatenate([q, np.zeros(3)])


def scale_to_metric(dataset: DatasetData, state: State, feature_id_to_points_index: Dict[int, int]):
    """Returns (scaling factor, scaled State).  Central-generic intrinsics are scale-free (CameraModel::Scale is a
    no-op for them); the non-central point grid would be multiplied by the factor (noncentral_generic.cc:148-154)."""
    log_sum, count = 0.0, 0
    for g in dataset.known_geometries:
        pos_to_index = {}
        for fid, pos in g.feature_id_to_position.items():
            idx = feature_id_to_points_index.get(fid)
            if idx is not None:
                pos_to_index[tuple(pos)] = idx
        if not pos_to_index:
            continue
        for fid, pos in g.feature_id_to_position.items():
            index = pos_to_index.get(tuple(pos))
            if index is None:
                continue
            for dx, dy in ((1, 0), (0, 1)):
                nb = pos_to_index.get((pos[0] + dx, pos[1] + dy))
                if nb is None:
                    continue
                actual = np.linalg.norm(state.points[index] - state.points[nb])
                log_sum += np.log(float(np.float32(g.cell_length_in_meters)) / actual)   # the field is a float (dataset.h:54)
                count += 1
    factor = float(np.exp(log_sum / count))
    rig = state.rig_tr_global.copy(); rig[:, 4:] *= factor
    ctr = state.camera_tr_rig.copy(); ctr[:, 4:] *= factor
    grids = []
    for g in state.grids:
        g = np.asarray(g)
        if g.ndim == 3:                      # non-central: (direction grid, point grid)
            g = g.copy(); g[1] *= factor
        grids.append(g)
    return factor, State(rig, ctr, state.points * factor, grids)


def run_bundle_adjustment(problem: Problem, state: State, max_iteration_count: int, cost_reduction_threshold: float,
                          localize_only: bool = False, device: int = 0):
    """RunBundleAdjustment (calibration.cc:187-304) on the HIP engine.  Returns (State, list of costs)."""
    e = _engine.Engine(problem, device=device)
    e.set_state(state)
    lam = -1.0
    last_cost = float("inf")
    costs = []
    st = state
    try:
        for _ in range(max_iteration_count):
            rep = e.step(lam)
            lam = rep.final_lambda
            cost = rep.final_cost
            costs.append(cost)
            if not localize_only:
                # beautify all camera orientations (calibration.cc:248-254); the grids live on the device,
                # so this round-trips 0.3 MB per iteration
                st = e.get_state(st)
                ctr = st.camera_tr_rig.copy(); grids = list(st.grids)
                for c, cam in enumerate(problem.cameras):
                    if cam.model_type != CENTRAL_GENERIC:
                        continue
                    R, grids[c] = choose_nice_camera_orientation(cam, grids[c])
                    ctr[c] = se3_mul(rotation_to_pose(R), ctr[c])
                st = State(st.rig_tr_global, ctr, st.points, grids)
                e.set_state(st)
            if cost >= last_cost - cost_reduction_threshold:
                break
            last_cost = cost
        if localize_only:
            st = e.get_state(st)
    finally:
        e.close()
    return st, costs


# ---------------------------------------------------------------------------------------------------
# the pyramid's grid-resolution rules and the refinement stage of Calibrate()
# ---------------------------------------------------------------------------------------------------
def compute_grid_resolution(cam: Camera, approx_pixels_per_cell: int):
    """ComputeGridResolution, APP/calibration.cc:531-559: area / pixels-per-cell in INTEGER division, + 0.5f + 2 * exterior cells, truncated;
    both generic models have one exterior cell per side (central_generic.h:123, noncentral_generic.h:146)."""
    aw = cam.calib_max_x - cam.calib_min_x + 1; ah = cam.calib_max_y - cam.calib_min_y + 1
    f = np.float32
    return (int(f(aw // approx_pixels_per_cell) + f(0.5) + f(2)), int(f(ah // approx_pixels_per_cell) + f(0.5) + f(2)))


def calc_grid_resolution_for_level(pyramid_level: int, full_resolution_x: int, full_resolution_y: int):
    """CalcGridResolutionForLevel, APP/calibration.cc:565-568: full * 1.333^-level + 0.5, truncated."""
    s = 1.333 ** (-pyramid_level)
    return int(full_resolution_x * s + 0.5), int(full_resolution_y * s + 0.5)


def _default_resample(cam: Camera, grid: np.ndarray, target_x: int, target_y: int):
    """ResampleModel between generic models of the SAME type (what the pyramid does; APP/calibration.cc:373-528)."""
    from . import grid_fit
    if cam.model_type == CENTRAL_GENERIC:
        new_cam, new_grid, _ = grid_fit.resample_model(cam, grid, target_x, target_y)
        if new_grid is None:
            raise RuntimeError("ResampleModel failed: the dense model could not initialise every grid point")
        return new_cam, np.asarray(new_grid).reshape(-1, 3)
    return grid_fit.resample_noncentral_model(cam, grid, target_x, target_y)


def _restrict(problem: Problem, state: State, keep: np.ndarray, image_used: np.ndarray):
    """The problem OptimizeJointly sees: surviving features of the used imagesets, imagesets renumbered (joint_optimization.cc:60-110)."""
    seq = -np.ones(problem.n_images, dtype=np.int64)
    seq[image_used] = np.arange(int(image_used.sum()))
    m = keep & image_used[problem.obs_image]
    sub = Problem(problem.cameras, int(image_used.sum()), problem.n_points, problem.obs_xy[m], problem.obs_point[m],
                  seq[problem.obs_image[m]].astype(np.int32), problem.obs_camera[m], problem.fd_delta, problem.localize_only,
                  problem.eliminate_points)
    return sub, State(state.rig_tr_global[image_used], state.camera_tr_rig, state.points, state.grids)


def calibrate_refinement_stage(problem: Problem, state: State, dataset: DatasetData, feature_id_to_points_index: Dict[int, int],
                               num_pyramid_levels: int, approx_pixels_per_cell: int, outlier_removal_factor: float = 0.0,
                               localize_only: bool = False, run_ba_fn: Optional[Callable] = None, resample_fn: Optional[Callable] = None,
                               delete_outliers_fn: Optional[Callable] = None):
    """The refinement stage of Calibrate() (APP/calibration.cc:1030-1142) on a packed problem whose cameras are at the COARSEST pyramid level.

    Pyramid levels L-1 ... 1 (skipped when localising only): RunBundleAdjustment(10 iterations, threshold 1e-4), RunBundleAdjustment(50, 1), then
    every camera resampled to the next level's resolution; outlier stage when ``outlier_removal_factor`` > 0: RunBundleAdjustment(100 if L == 1
    else 10, 1e-4) and DeleteOutlierFeatures camera by camera (an imageset that drops below three features of a camera is unused from then on);
    main RunBundleAdjustment(100, 1e-4); ScaleToMetric unless localising only.

    The functions that touch the GPU are injectable (tests run the stage on the CPU oracle): ``run_ba_fn(problem, state, max_iterations,
    threshold, localize_only) -> (State, costs)``, ``resample_fn(cam, grid, target_x, target_y) -> (Camera, grid)``,
    ``delete_outliers_fn(camera_index, problem, state, factor, image_used) -> (keep, image_used, threshold)``.
    Returns dict(problem (full-resolution cameras, all observations), state, keep, image_used, scale, ba_runs)."""
    from . import report as _report
    run_ba_fn = run_ba_fn or (lambda pb, st, it, thr, loc: run_bundle_adjustment(pb, st, it, thr, localize_only=loc))
    resample_fn = resample_fn or _default_resample
    delete_outliers_fn = delete_outliers_fn or (lambda c, pb, st, f, used: _report.delete_outlier_features(c, pb, st, f, used))
    cameras = list(problem.cameras)
    full = [compute_grid_resolution(cam, approx_pixels_per_cell) for cam in cameras]
    keep = np.ones(problem.n_obs, dtype=bool)
    used = np.ones(problem.n_images, dtype=bool)
    st = state.copy()
    ba_runs = []

    def with_cameras(cams):
        return Problem(cams, problem.n_images, problem.n_points, problem.obs_xy, problem.obs_point, problem.obs_image, problem.obs_camera,
                       problem.fd_delta, localize_only, problem.eliminate_points)

    def run(pb_full, st_full, iterations, threshold):
        sub, st_sub = _restrict(pb_full, st_full, keep, used)
        out, costs = run_ba_fn(sub, st_sub, iterations, threshold, localize_only)
        ba_runs.append(dict(max_iteration_count=iterations, threshold=threshold, iterations=len(costs), final_cost=costs[-1] if costs else None))
        rig = st_full.rig_tr_global.copy(); rig[used] = out.rig_tr_global
        return State(rig, out.camera_tr_rig, out.points, out.grids)

    pb = with_cameras(cameras)
    if not localize_only:
        for level in range(num_pyramid_levels - 1, 0, -1):
            for c, cam in enumerate(cameras):          # the reference CHECKs this (:1063-1066)
                want = calc_grid_resolution_for_level(level, *full[c])
                if (cam.grid_w, cam.grid_h) != want:
                    raise ValueError(f"camera {c}: grid {cam.grid_w} x {cam.grid_h} on pyramid level {level}, expected {want[0]} x {want[1]}")
            st = run(pb, st, 10, 0.0001)
            st = run(pb, st, 50, 1.0)
            grids = []
            for c, cam in enumerate(cameras):
                tx, ty = calc_grid_resolution_for_level(level - 1, *full[c])
                cameras[c], g = resample_fn(cam, st.grids[c], tx, ty)
                grids.append(np.ascontiguousarray(g, dtype=np.float64))
            st = State(st.rig_tr_global, st.camera_tr_rig, st.points, grids)
            pb = with_cameras(cameras)
    if outlier_removal_factor > 0:
        st = run(pb, st, 100 if num_pyramid_levels == 1 else 10, 0.0001)
        for c in range(len(cameras)):
            k, used_after, _ = delete_outliers_fn(c, _subset_observations(pb, keep), st, outlier_removal_factor, used)
            keep[np.flatnonzero(keep)[~np.asarray(k, dtype=bool)]] = False
            used = np.asarray(used_after, dtype=bool).copy()
    st = run(pb, st, 100, 0.0001)
    scale = None
    if not localize_only:
        scale, st = scale_to_metric(dataset, st, feature_id_to_points_index)
    return dict(problem=pb, state=st, keep=keep, image_used=used, scale=scale, ba_runs=ba_runs)


def _subset_observations(problem: Problem, keep: np.ndarray) -> Problem:
    return Problem(problem.cameras, problem.n_images, problem.n_points, problem.obs_xy[keep], problem.obs_point[keep], problem.obs_image[keep],
                   problem.obs_camera[keep], problem.fd_delta, problem.localize_only, problem.eliminate_points)

