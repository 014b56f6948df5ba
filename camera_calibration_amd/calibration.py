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
