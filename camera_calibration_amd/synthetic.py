"""Deterministic synthetic bundle-adjustment problems.

Two generators:

* ``reference_test_problem`` restates the reference's own BA fixture ``TestOptimizeJointly``
  (APP/test/util.h:275-571: 600x400 px, 5x5 grid, 150 points, 100 poses, perturbations
  0.05 / 0.04 / 0.04 / 0.02).  The reference seeds libc ``rand()`` which is not reproducible across
  libcs, so the draws come from numpy's PCG64 instead and the 5x5 grid is set to the pinhole rays at
  the grid-point pixels rather than fitted with FitToDenseModel (out of scope).
* ``baseline_config`` builds the five BASELINE.json configurations.

Observation pixels are produced by the *iterative* projection of the ground-truth model, supplied by
the caller as ``project_fn(camera, grid, local_points) -> (pixels, ok)`` -- the HIP engine's
``cba_project`` in bench.py / GPU tests, the oracle in CPU tests.  Nothing here touches the oracle.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np

from .problem import CENTRAL_GENERIC, NONCENTRAL_GENERIC, Camera, Problem, State
from .se3 import se3_exp, se3_identity, se3_mul, transform_points

ProjectFn = Callable[[Camera, np.ndarray, np.ndarray], Tuple[np.ndarray, np.ndarray]]


def grid_point_to_pixel(cam: Camera, gx: np.ndarray, gy: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """GridPointToPixelCornerConv (APP/models/central_grid.h:127-131), evaluated in fp64."""
    px = cam.calib_min_x + ((gx - 1.0) / (cam.grid_w - 3.0)) * (cam.calib_max_x + 1 - cam.calib_min_x)
    py = cam.calib_min_y + ((gy - 1.0) / (cam.grid_h - 3.0)) * (cam.calib_max_y + 1 - cam.calib_min_y)
    return px, py


def pinhole_direction_grid(cam: Camera, fx: float, fy: float, cx: float, cy: float, k1: float = 0.0) -> np.ndarray:
    """(G,3) unit directions: control point (gx,gy) = pinhole ray (with radial term k1) of its pixel."""
    gy, gx = np.meshgrid(np.arange(cam.grid_h, dtype=np.float64), np.arange(cam.grid_w, dtype=np.float64), indexing="ij")
    px, py = grid_point_to_pixel(cam, gx, gy)
    xn = (px - cx) / fx
    yn = (py - cy) / fy
    s = 1.0 + k1 * (xn * xn + yn * yn)
    d = np.stack([xn * s, yn * s, np.ones_like(xn)], axis=-1).reshape(-1, 3)
    return d / np.linalg.norm(d, axis=1, keepdims=True)


def _make_observations(cams: List[Camera], gt_grids, camera_tr_rig, rig_tr_global, points, project_fn: ProjectFn,
                       noise_px: float, rng: np.random.Generator):
    xy, pt, im, cm = [], [], [], []
    for i in range(rig_tr_global.shape[0]):
        for c, cam in enumerate(cams):
            pose = se3_mul(camera_tr_rig[c], rig_tr_global[i])
            local = transform_points(pose, points)
            px, ok = project_fn(cam, gt_grids[c], local)
            px = px.astype(np.float32)
            if noise_px > 0:
                px = (px.astype(np.float64) + rng.normal(0.0, noise_px, size=px.shape)).astype(np.float32)
            inside = (px[:, 0] >= cam.calib_min_x) & (px[:, 1] >= cam.calib_min_y) & \
                     (px[:, 0] < cam.calib_max_x + 1) & (px[:, 1] < cam.calib_max_y + 1)
            keep = np.nonzero(ok & inside)[0]
            xy.append(px[keep]); pt.append(keep.astype(np.int32))
            im.append(np.full(keep.size, i, np.int32)); cm.append(np.full(keep.size, c, np.int32))
    return (np.concatenate(xy).astype(np.float32), np.concatenate(pt), np.concatenate(im), np.concatenate(cm))


def reference_test_problem(num_cameras: int, project_fn: ProjectFn, seed: int = 0, num_points: int = 150,
                           num_poses: int = 100, model_type: int = CENTRAL_GENERIC):
    """TestOptimizeJointly fixture (APP/test/util.h:275-571). Returns (problem, perturbed_state, gt_state)."""
    rng = np.random.default_rng(seed)
    U = lambda *shape: rng.uniform(-1.0, 1.0, size=shape)  # Eigen ::Random() is uniform in [-1,1]
    W, H = 600, 400
    cams, grids = [], []
    camera_tr_rig = []
    for c in range(num_cameras):
        cam = Camera(model_type, W, H, 0, 0, W - 1, H - 1, 5, 5)
        cams.append(cam)
        d = pinhole_direction_grid(cam, H / 2.0 + 2.0 * c, H / 2.0, W / 2.0, H / 2.0)
        if model_type == NONCENTRAL_GENERIC:
            d = np.stack([d, np.zeros_like(d)])
        grids.append(d)
        camera_tr_rig.append(se3_identity() if c == 0 else se3_exp(0.05 * U(6)))
    camera_tr_rig = np.array(camera_tr_rig)
    gt_points = U(num_points, 3) * np.array([6.5, 3.5, 1.0])
    base = np.array([1.0, 0, 0, 0, 0, 0, 5.0])
    gt_poses = []
    for _ in range(num_poses):
        b = base.copy(); b[4:] += U(3)
        gt_poses.append(se3_mul(se3_exp(0.05 * U(6)), b))
    gt_poses = np.array(gt_poses)
    xy, pt, im, cm = _make_observations(cams, grids, camera_tr_rig, gt_poses, gt_points, project_fn, 0.0, rng)
    problem = Problem(cams, num_poses, num_points, xy, pt, im, cm, fd_delta=1e-4)
    gt = State(gt_poses, camera_tr_rig, gt_points, grids)
    st = gt.copy()
    st.points += 0.05 * U(num_points, 3)
    for i in range(num_poses):
        st.rig_tr_global[i] = se3_mul(st.rig_tr_global[i], se3_exp(0.04 * U(6)))
    for c in range(num_cameras):  # the gtest perturbs camera_tr_rig[0] even for one camera (util.h:386-390)
        st.camera_tr_rig[c] = se3_mul(st.camera_tr_rig[c], se3_exp(0.04 * U(6)))
    for c in range(num_cameras):
        g = st.grids[c]
        dgrid = g if model_type == CENTRAL_GENERIC else g[0]
        dgrid += 0.02 * U(*dgrid.shape)
        dgrid /= np.linalg.norm(dgrid, axis=-1, keepdims=True)
    return problem, st, gt


# ------------------------------------------------------------------------------------------------
# BASELINE.json configurations
# ------------------------------------------------------------------------------------------------
BASELINE_CONFIGS = {
    # cfg: (cameras, model, W, H, grid_w, grid_h, lattice_x, lattice_y, n_imagesets, fd_delta)
    1: (1, CENTRAL_GENERIC, 640, 480, 16, 12, 16, 23, 30, 1e-4),
    2: (1, CENTRAL_GENERIC, 2048, 1456, 84, 60, 24, 35, 500, 1e-4),
    3: (2, CENTRAL_GENERIC, 2048, 1456, 84, 60, 24, 35, 1000, 1e-4),
    4: (1, NONCENTRAL_GENERIC, 1280, 960, 52, 40, 24, 35, 800, 1e-3),
    5: (4, CENTRAL_GENERIC, 2048, 1456, 84, 60, 24, 35, 4000, 1e-4),
}


def pattern_points(lattice_x: int, lattice_y: int, pitch: float, rng: np.random.Generator) -> np.ndarray:
    """Planar corner lattice with the (5x5) AprilTag hole removed, z jitter sigma = 0.3 mm.
    P = lattice_x*lattice_y - 25 (APP/feature_detection/feature_detector_tagged_pattern.h:71-87)."""
    ys, xs = np.meshgrid(np.arange(lattice_y), np.arange(lattice_x), indexing="ij")
    cx, cy = (lattice_x - 1) / 2.0, (lattice_y - 1) / 2.0
    hole_x0, hole_y0 = lattice_x // 2 - 2, lattice_y // 2 - 2
    hole = (xs >= hole_x0) & (xs < hole_x0 + 5) & (ys >= hole_y0) & (ys < hole_y0 + 5)
    keep = ~hole
    pts = np.stack([(xs[keep] - cx) * pitch, (ys[keep] - cy) * pitch,
                    rng.normal(0.0, 0.0003, size=int(keep.sum()))], axis=-1)
    return pts


def pattern_positions(lattice_x: int, lattice_y: int) -> np.ndarray:
    """Integer lattice position of every point of ``pattern_points`` (same order): the known geometry of the synthetic pattern
    (KnownGeometry::feature_id_to_position, APP/dataset.h:49-56, with feature id = point index)."""
    ys, xs = np.meshgrid(np.arange(lattice_y), np.arange(lattice_x), indexing="ij")
    hole_x0, hole_y0 = lattice_x // 2 - 2, lattice_y // 2 - 2
    hole = (xs >= hole_x0) & (xs < hole_x0 + 5) & (ys >= hole_y0) & (ys < hole_y0 + 5)
    return np.stack([xs[~hole], ys[~hole]], axis=-1).astype(np.int32)


def _rig_layout(n_cams: int) -> np.ndarray:
    if n_cams == 1:
        return se3_identity(1)
    out = [se3_identity()]
    if n_cams == 2:  # stereo: baseline 0.12 m, 5 degree toe-in
        out.append(se3_exp(np.array([-0.12, 0, 0, 0, np.deg2rad(5.0), 0])))
    else:            # ring rig, r = 0.1 m, slight toe-in towards the axis
        for k in range(1, n_cams):
            a = 2 * np.pi * k / n_cams
            out.append(se3_exp(np.array([-0.1 * (np.cos(a) - 1.0), -0.1 * np.sin(a), 0,
                                         np.deg2rad(3.0) * np.sin(a), -np.deg2rad(3.0) * (np.cos(a) - 1.0), 0])))
    return np.array(out)


def baseline_config(cfg: int, project_fn: ProjectFn, n_imagesets: int | None = None, noise_px: float = 0.03,
                    seed: int | None = None, grid_perturbation: float = 0.1, pose_perturbation: float = 0.01,
                    point_perturbation: float = 0.002, image_offset: int = 0, grid_wh: Tuple[int, int] | None = None,
                    lattice_xy: Tuple[int, int] | None = None):
    """Build BASELINE.json config ``cfg`` (optionally with fewer imagesets). Returns (problem, state, gt).

    Perturbations: points +-``point_perturbation`` m, poses exp(``pose_perturbation``*U^6), grid directions
    += ``grid_perturbation`` * (angular cell spacing) * U^3 then renormalised -- scaled to the grid
    resolution so that the fine 84x60 grids stay monotone (the gtest's 0.02 rad is for a 5x5 grid).
    ``image_offset`` shifts the per-imageset random stream so that shards of one big problem can be
    generated independently (rank r of an N-GPU run passes image_offset = r * n_imagesets).
    """
    n_cams, model, W, H, gw, gh, lx, ly, n_default, fd = BASELINE_CONFIGS[cfg]
    if grid_wh is not None:      # coarser grid for memory-bound test hosts (the configuration's own grid is the default)
        gw, gh = grid_wh
    if lattice_xy is not None:   # a smaller pattern (fewer points) for tests whose checker is slow
        lx, ly = lattice_xy
    N = n_default if n_imagesets is None else n_imagesets
    seed = 1000 + cfg if seed is None else seed
    rng = np.random.default_rng(seed)
    f = 0.8 * H
    cams, gt_grids = [], []
    for c in range(n_cams):
        cam = Camera(model, W, H, 0, 0, W - 1, H - 1, gw, gh)
        cams.append(cam)
        d = pinhole_direction_grid(cam, f, f, W / 2.0, H / 2.0, k1=-0.12)
        if model == NONCENTRAL_GENERIC:
            gy, gx = np.meshgrid(np.arange(gh, dtype=np.float64), np.arange(gw, dtype=np.float64), indexing="ij")
            o = 0.002 * np.stack([np.sin(0.3 * gx + 0.1 * gy), np.cos(0.2 * gy - 0.15 * gx),
                                  0.5 * np.sin(0.11 * gx - 0.07 * gy)], axis=-1).reshape(-1, 3)
            d = np.stack([d, o])
        gt_grids.append(d)
    camera_tr_rig = _rig_layout(n_cams)
    pitch = 0.01188
    points = pattern_points(lx, ly, pitch, rng)
    P = points.shape[0]

    # imageset poses: pattern centre at 0.35..0.9 m along the ray of a uniformly drawn pixel, tilt <= 40 deg
    poses = np.empty((N, 7))
    for i in range(N):
        r = np.random.default_rng([seed, 7919, image_offset + i])
        z = r.uniform(0.35, 0.9)
        u = r.uniform(0.15 * W, 0.85 * W); v = r.uniform(0.15 * H, 0.85 * H)
        t = np.array([(u - W / 2.0) / f * z, (v - H / 2.0) / f * z, z])
        tilt = np.deg2rad(40.0) * np.sqrt(r.uniform())
        ax = r.uniform(0, 2 * np.pi)
        roll = r.uniform(-0.5, 0.5)
        rot = se3_mul(se3_exp(np.array([0, 0, 0, tilt * np.cos(ax), tilt * np.sin(ax), 0])),
                      se3_exp(np.array([0, 0, 0, 0, 0, roll])))
        rot[4:] = t
        poses[i] = rot
    xy, pt, im, cm = _make_observations(cams, gt_grids, camera_tr_rig, poses, points, project_fn, noise_px,
                                        np.random.default_rng([seed, 104729, image_offset]))
    problem = Problem(cams, N, P, xy, pt, im, cm, fd_delta=fd)
    gt = State(poses, camera_tr_rig, points, gt_grids)

    st = gt.copy()
    prng = np.random.default_rng([seed, 15485863])
    U = lambda *shape: prng.uniform(-1.0, 1.0, size=shape)
    st.points += point_perturbation * U(P, 3)
    if n_cams > 1:
        for c in range(n_cams):
            st.camera_tr_rig[c] = se3_mul(st.camera_tr_rig[c], se3_exp(pose_perturbation * U(6)))
    for c in range(n_cams):
        cell = (W / (gw - 3.0)) / f  # angular size of one grid cell
        g = st.grids[c]
        dgrid = g if model == CENTRAL_GENERIC else g[0]
        dgrid += grid_perturbation * cell * U(*dgrid.shape)
        dgrid /= np.linalg.norm(dgrid, axis=-1, keepdims=True)
        if model == NONCENTRAL_GENERIC:
            g[1] += 0.0002 * U(*g[1].shape)
    for i in range(N):
        r = np.random.default_rng([seed, 32452843, image_offset + i])
        st.rig_tr_global[i] = se3_mul(st.rig_tr_global[i], se3_exp(pose_perturbation * r.uniform(-1.0, 1.0, size=6)))
    return problem, st, gt
