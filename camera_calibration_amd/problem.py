"""Plain-numpy containers for one bundle-adjustment problem.

These mirror what the reference's ``Dataset`` / ``BAState`` hold for the JointOptimization path
(APP/dataset.h:57-212, APP/bundle_adjustment/ba_state.h:46-97; APP = applications/camera_calibration/
src/camera_calibration in the reference tree), flattened into the packed arrays that the C-ABI
(include/cba.h) takes:

* observations sorted image-major, then camera, then feature order -- the reference's loop order
  (APP/bundle_adjustment/joint_optimization.cc:273-291), which also defines the residual order that
  ``CostIsSmallerThan`` pairs up (libvis/src/libvis/lm_optimizer.h:993-1011);
* poses as 7 doubles ``qw qx qy qz tx ty tz`` (Sophus SE3d = unit quaternion + translation);
* grids row-major ``index = gx + gy*grid_w`` as in ``Image<Vec3d>`` (APP/models/central_grid.h:213);
  the non-central model stores the direction grid followed by the point grid.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np

CENTRAL_GENERIC = 0
NONCENTRAL_GENERIC = 1


@dataclass
class Camera:
    """Calibrated rectangle + grid resolution of one generic camera (APP/models/camera_model.h:42-204)."""
    model_type: int
    width: int
    height: int
    calib_min_x: int
    calib_min_y: int
    calib_max_x: int
    calib_max_y: int
    grid_w: int
    grid_h: int

    @property
    def grid_points(self) -> int:
        return self.grid_w * self.grid_h

    @property
    def params_per_grid_point(self) -> int:
        return 2 if self.model_type == CENTRAL_GENERIC else 5

    @property
    def intrinsics_param_count(self) -> int:
        """update_parameter_count(): central_grid.h:120-122 / noncentral_generic.h:141-143."""
        return self.params_per_grid_point * self.grid_points

    @property
    def grid_doubles(self) -> int:
        return (3 if self.model_type == CENTRAL_GENERIC else 6) * self.grid_points


@dataclass
class Problem:
    cameras: List[Camera]
    n_images: int
    n_points: int
    obs_xy: np.ndarray       # (n,2) float32  PointFeature::xy
    obs_point: np.ndarray    # (n,) int32     PointFeature::index
    obs_image: np.ndarray    # (n,) int32     sequential (used) image index
    obs_camera: np.ndarray   # (n,) int32
    fd_delta: float = 1e-4   # numerical_diff_delta (APP/calibration.cc:201)
    localize_only: bool = False
    eliminate_points: bool = False

    def __post_init__(self):
        self.obs_xy = np.ascontiguousarray(self.obs_xy, dtype=np.float32).reshape(-1, 2)
        self.obs_point = np.ascontiguousarray(self.obs_point, dtype=np.int32)
        self.obs_image = np.ascontiguousarray(self.obs_image, dtype=np.int32)
        self.obs_camera = np.ascontiguousarray(self.obs_camera, dtype=np.int32)
        n = self.n_obs
        assert self.obs_point.shape == (n,) and self.obs_image.shape == (n,) and self.obs_camera.shape == (n,)
        if n:
            key = self.obs_image.astype(np.int64) * len(self.cameras) + self.obs_camera
            assert np.all(np.diff(key) >= 0), "observations must be sorted image-major, then camera"
            assert self.obs_point.min() >= 0 and self.obs_point.max() < self.n_points
            assert self.obs_image.min() >= 0 and self.obs_image.max() < self.n_images

    @property
    def n_obs(self) -> int:
        return int(self.obs_xy.shape[0])

    @property
    def n_cameras(self) -> int:
        return len(self.cameras)

    # --- variable ordering: JointOptimizationState, joint_optimization.cc:49-59, 142-170 ---
    @property
    def rig_in_state(self) -> bool:
        return self.n_cameras > 1

    @property
    def rig_dof(self) -> int:
        return 6 * self.n_cameras if self.rig_in_state else 0

    @property
    def intrinsics_dof(self) -> int:
        return 0 if self.localize_only else sum(c.intrinsics_param_count for c in self.cameras)

    @property
    def block_size(self) -> int:
        return 3 if self.eliminate_points else 6

    @property
    def n_blocks(self) -> int:
        return self.n_points if self.eliminate_points else self.n_images

    @property
    def block_dof(self) -> int:
        return self.block_size * self.n_blocks

    @property
    def total_dof(self) -> int:
        return 6 * self.n_images + self.rig_dof + 3 * self.n_points + self.intrinsics_dof

    @property
    def dense_dof(self) -> int:
        return self.total_dof - self.block_dof

    def image_slice(self, begin: int, end: int) -> "Problem":
        """Sub-problem with images [begin, end) re-indexed from 0 (image sharding)."""
        sel = (self.obs_image >= begin) & (self.obs_image < end)
        return Problem(self.cameras, end - begin, self.n_points, self.obs_xy[sel], self.obs_point[sel],
                       self.obs_image[sel] - begin, self.obs_camera[sel], self.fd_delta,
                       self.localize_only, self.eliminate_points)


@dataclass
class State:
    rig_tr_global: np.ndarray            # (N,7)
    camera_tr_rig: np.ndarray            # (C,7)
    points: np.ndarray                   # (P,3)
    grids: List[np.ndarray] = field(default_factory=list)  # per camera (G,3) or (2,G,3)

    def __post_init__(self):
        self.rig_tr_global = np.ascontiguousarray(self.rig_tr_global, dtype=np.float64).reshape(-1, 7)
        self.camera_tr_rig = np.ascontiguousarray(self.camera_tr_rig, dtype=np.float64).reshape(-1, 7)
        self.points = np.ascontiguousarray(self.points, dtype=np.float64).reshape(-1, 3)
        self.grids = [np.ascontiguousarray(g, dtype=np.float64) for g in self.grids]

    def copy(self) -> "State":
        return State(self.rig_tr_global.copy(), self.camera_tr_rig.copy(), self.points.copy(),
                     [g.copy() for g in self.grids])

    def image_slice(self, begin: int, end: int) -> "State":
        return State(self.rig_tr_global[begin:end].copy(), self.camera_tr_rig.copy(), self.points.copy(),
                     [g.copy() for g in self.grids])
