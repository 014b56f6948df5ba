"""On-disk formats either side of the bundle-adjustment path.

Mirrors APP/io/calibration_io.cc (APP = applications/camera_calibration/src/camera_calibration):

* ``save_dataset`` / ``load_dataset``  -- `dataset.bin`, :51-247.  Layout: magic ``calib_data`` (10 bytes), u32
  version 0, u32 camera count, per camera u32 width, u32 height; u32 imageset count, per imageset u32 filename
  length + bytes, per camera u32 feature count + features ``{f32 x, f32 y, i32 id}``; u32 known-geometry count,
  per geometry f32 cell_length_in_meters, u32 map size, entries ``{i32 id, i32 x, i32 y}``.  Integers are written
  through htonl (big-endian), floats raw (little-endian on the hosts the reference runs on), APP/io/io_util.h:37-67.
* ``save_ba_state`` / ``load_ba_state`` -- the BAState directory, :432-524: ``rig_tr_global.yaml``,
  ``camera_tr_rig.yaml`` (SavePoses/LoadPoses :785-888), ``intrinsicsN.yaml`` (SaveCameraModel/LoadCameraModel
  :527-783, generic models; 14 significant digits; direction grids re-normalised on load), ``points.yaml``
  (:890-985).  The reference's convenience ``.obj`` side files are not written.

Plain Python + numpy; the YAML subset the reference emits is written by hand (byte-compatible line format) and
read back with PyYAML.  `dataset_to_problem` packs a loaded dataset + state into the engine's arrays.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .problem import CENTRAL_GENERIC, NONCENTRAL_GENERIC, Camera, Problem, State

MAGIC = b"calib_data"


@dataclass
class KnownGeometry:
    cell_length_in_meters: float = 0.0
    feature_id_to_position: Dict[int, Tuple[int, int]] = field(default_factory=dict)


@dataclass
class ImagesetData:
    filename: str = ""
    features: List[np.ndarray] = field(default_factory=list)   # per camera: structured (x f32, y f32, id i32)


FEATURE_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("id", ">i4")])


@dataclass
class DatasetData:
    image_sizes: List[Tuple[int, int]] = field(default_factory=list)
    imagesets: List[ImagesetData] = field(default_factory=list)
    known_geometries: List[KnownGeometry] = field(default_factory=list)

    @property
    def num_cameras(self) -> int:
        return len(self.image_sizes)


def save_dataset(path: str, ds: DatasetData) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack(">I", 0))
        f.write(struct.pack(">I", ds.num_cameras))
        for w, h in ds.image_sizes:
            f.write(struct.pack(">II", w, h))
        f.write(struct.pack(">I", len(ds.imagesets)))
        for s in ds.imagesets:
            name = s.filename.encode()
            f.write(struct.pack(">I", len(name))); f.write(name)
            for c in range(ds.num_cameras):
                feats = np.ascontiguousarray(s.features[c], dtype=FEATURE_DTYPE)
                f.write(struct.pack(">I", feats.shape[0]))
                f.write(feats.tobytes())
        f.write(struct.pack(">I", len(ds.known_geometries)))
        for g in ds.known_geometries:
            f.write(struct.pack("<f", g.cell_length_in_meters))
            f.write(struct.pack(">I", len(g.feature_id_to_position)))
            for fid, (x, y) in g.feature_id_to_position.items():
                f.write(struct.pack(">iii", fid, x, y))


def load_dataset(path: str) -> DatasetData:
    with open(path, "rb") as f:
        data = f.read()
    if data[:10] != MAGIC:
        raise ValueError(f"{path}: invalid file header")
    off = 10

    def u32():
        nonlocal off
        v = struct.unpack_from(">I", data, off)[0]; off += 4
        return v

    if u32() != 0:
        raise ValueError(f"{path}: unsupported file format version")
    ds = DatasetData()
    for _ in range(u32()):
        w = u32(); h = u32()
        ds.image_sizes.append((w, h))
    for _ in range(u32()):
        n = u32()
        if off + n > len(data):
            raise ValueError(f"{path}: unexpected end of file")
        s = ImagesetData(filename=data[off:off + n].decode()); off += n
        for _c in range(ds.num_cameras):
            k = u32()
            s.features.append(np.frombuffer(data, dtype=FEATURE_DTYPE, count=k, offset=off).copy()); off += 12 * k
        ds.imagesets.append(s)
    for _ in range(u32()):
        g = KnownGeometry(cell_length_in_meters=struct.unpack_from("<f", data, off)[0]); off += 4
        for _i in range(u32()):
            fid, x, y = struct.unpack_from(">iii", data, off); off += 12
            g.feature_id_to_position[fid] = (x, y)
        ds.known_geometries.append(g)
    return ds


# ---------------------------------------------------------------------------------------------------
# BAState directory
# ---------------------------------------------------------------------------------------------------
def _g14(v: float) -> str:
    """operator<< of a double under std::setprecision(14): %.14g"""
    return "%.14g" % float(v)


def save_poses(path: str, image_used, poses: np.ndarray) -> None:
    """SavePoses, :785-839 (poses as qw qx qy qz tx ty tz rows)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        f.write("# Each pose gives the B_tr_A transformation (i.e., A to B with right-multiplication), where the spaces A and B "
                "are defined by the filename. Quaternions are written as used by the Eigen library.\n")
        f.write(f"pose_count: {len(image_used)}\n")
        f.write("poses:\n")
        for i, used in enumerate(image_used):
            if not used:
                continue
            qw, qx, qy, qz, tx, ty, tz = poses[i]
            f.write(f"  - index: {i}\n")
            for k, v in (("tx", tx), ("ty", ty), ("tz", tz), ("qx", qx), ("qy", qy), ("qz", qz), ("qw", qw)):
                f.write(f"    {k}: {_g14(v)}\n")
    # "For convenience, we always also save an .obj file that can be used to visualize the pose positions.  All vertices are colored
    # red." (:817-836): the camera centre  -R^T t  of every used pose
    from .se3 import quat_to_matrix
    with open(path + ".obj", "w") as f:
        for i, used in enumerate(image_used):
            if not used:
                continue
            R = quat_to_matrix(np.asarray(poses[i][:4], dtype=np.float64))
            c = R.T @ (-np.asarray(poses[i][4:], dtype=np.float64))
            f.write(f"v {_g14(c[0])} {_g14(c[1])} {_g14(c[2])} 1 0 0\n")


def load_poses(path: str):
    """LoadPoses, :841-888.  Returns (image_used bool array, poses (n,7)); unused entries are identity.
    setQuaternion normalises (Sophus so3.hpp), restated here."""
    import yaml
    with open(path) as f:
        node = yaml.safe_load(f)
    n = int(node["pose_count"])
    used = np.zeros(n, dtype=bool)
    poses = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0]), (n, 1))
    for p in node["poses"] or []:
        i = int(p["index"])
        if i >= n:
            raise ValueError(f"{path}: pose index {i} >= pose_count {n}")
        used[i] = True
        q = np.array([p["qw"], p["qx"], p["qy"], p["qz"]], dtype=np.float64)
        poses[i, :4] = q / np.linalg.norm(q)
        poses[i, 4:] = [p["tx"], p["ty"], p["tz"]]
    return used, poses


def _grid_text(grid: np.ndarray) -> str:
    return "[" + ", ".join(_g14(v) for v in np.asarray(grid).reshape(-1)) + "]\n"


def save_camera_model(path: str, cam: Camera, grid: np.ndarray) -> None:
    """SaveCameraModel, :527-651 (generic models).  grid: (G,3) central, (2,G,3) non-central = direction, point."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    central = cam.model_type == CENTRAL_GENERIC
    with open(path, "w") as f:
        f.write(f"type : {'CentralGenericModel' if central else 'NoncentralGenericModel'}\n")
        f.write(f"width : {cam.width}\nheight : {cam.height}\n")
        f.write(f"calibration_min_x : {cam.calib_min_x}\ncalibration_min_y : {cam.calib_min_y}\n")
        f.write(f"calibration_max_x : {cam.calib_max_x}\ncalibration_max_y : {cam.calib_max_y}\n")
        f.write(f"grid_width : {cam.grid_w}\ngrid_height : {cam.grid_h}\n")
        if central:
            f.write("# The grid is stored in row-major order, top to bottom. Each row is stored left to right. "
                    "Each grid point is stored as x, y, z.\n")
            f.write("grid : " + _grid_text(grid))
        else:
            g = np.asarray(grid).reshape(2, -1, 3)
            f.write("# The grids are stored in row-major order, top to bottom. Each row is stored left to right. "
                    "Each grid point is stored as x, y, z.\n")
            f.write("point_grid : " + _grid_text(g[1]))
            f.write("direction_grid : " + _grid_text(g[0]))


def load_camera_model(path: str):
    """LoadCameraModel, :653-783 (generic models; direction grids are re-normalised, :667-670)."""
    import yaml
    with open(path) as f:
        node = yaml.safe_load(f)
    w, h = int(node["width"]), int(node["height"])
    if w < 1 or h < 1:
        raise ValueError(f"{path}: invalid image dimensions")
    t = node["type"]
    gw, gh = int(node["grid_width"]), int(node["grid_height"])
    if gw < 4 or gh < 4:
        raise ValueError(f"{path}: invalid grid dimensions")
    args = (w, h, int(node["calibration_min_x"]), int(node["calibration_min_y"]), int(node["calibration_max_x"]),
            int(node["calibration_max_y"]), gw, gh)

    def grid_of(key, normalized):
        a = np.asarray(node[key], dtype=np.float64)
        if a.size != 3 * gw * gh:
            raise ValueError(f"{path}: expected {3 * gw * gh} entries in '{key}', got {a.size}")
        a = a.reshape(-1, 3)
        return a / np.linalg.norm(a, axis=1, keepdims=True) if normalized else a

    if t in ("CentralGenericModel", "CentralGenericBSplineModel"):
        return Camera(CENTRAL_GENERIC, *args), grid_of("grid", True)
    if t in ("NoncentralGenericModel", "NoncentralGenericBSplineModel"):
        return Camera(NONCENTRAL_GENERIC, *args), np.stack([grid_of("direction_grid", True), grid_of("point_grid", False)])
    raise ValueError(f"{path}: cannot load camera model type {t}")


def save_points(path: str, points: np.ndarray, feature_id_to_points_index: Dict[int, int]) -> None:
    """SavePointsAndIndexMapping, :890-937."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        f.write("# Each point is stored as x, y, z.\n")
        f.write("points : [" + ", ".join(_g14(v) for v in np.asarray(points).reshape(-1)) + "]\n")
        f.write("feature_id_to_point_index:\n")
        for fid, idx in feature_id_to_points_index.items():
            f.write(f"  - feature_id: {fid}\n    point_index: {idx}\n")
    # the pattern points as blue vertices (:923-935)
    with open(path + ".obj", "w") as f:
        for p in np.asarray(points, dtype=np.float64).reshape(-1, 3):
            f.write(f"v {_g14(p[0])} {_g14(p[1])} {_g14(p[2])} 0 0 1\n")


def load_points(path: str):
    """LoadPointsAndIndexMapping, :939-985."""
    import yaml
    with open(path) as f:
        node = yaml.safe_load(f)
    pts = np.asarray(node["points"], dtype=np.float64)
    if pts.size % 3 != 0:
        raise ValueError(f"{path}: points node size is not an integer multiple of 3")
    mapping = {int(e["feature_id"]): int(e["point_index"]) for e in (node["feature_id_to_point_index"] or [])}
    return pts.reshape(-1, 3), mapping


def save_ba_state(base_path: str, image_used, cameras: List[Camera], state: State, feature_id_to_points_index: Dict[int, int]) -> None:
    """SaveBAState, :432-466."""
    save_poses(os.path.join(base_path, "rig_tr_global.yaml"), image_used, state.rig_tr_global)
    save_poses(os.path.join(base_path, "camera_tr_rig.yaml"), [True] * len(cameras), state.camera_tr_rig)
    for c, cam in enumerate(cameras):
        save_camera_model(os.path.join(base_path, f"intrinsics{c}.yaml"), cam, state.grids[c])
    save_points(os.path.join(base_path, "points.yaml"), state.points, feature_id_to_points_index)


def load_ba_state(base_path: str):
    """LoadBAState, :468-524.  Returns (image_used, cameras, State, feature_id_to_points_index)."""
    used, rig = load_poses(os.path.join(base_path, "rig_tr_global.yaml"))
    _, ctr = load_poses(os.path.join(base_path, "camera_tr_rig.yaml"))
    cameras, grids = [], []
    c = 0
    while os.path.exists(os.path.join(base_path, f"intrinsics{c}.yaml")):
        cam, g = load_camera_model(os.path.join(base_path, f"intrinsics{c}.yaml"))
        cameras.append(cam); grids.append(g); c += 1
    if not cameras:
        raise FileNotFoundError(os.path.join(base_path, "intrinsics0.yaml"))
    pts, mapping = load_points(os.path.join(base_path, "points.yaml"))
    return used, cameras, State(rig, ctr, pts, grids), mapping


def dataset_to_problem(ds: DatasetData, image_used, cameras: List[Camera], state: State, mapping: Dict[int, int],
                       fd_delta: float = 1e-4):
    """Packs (Dataset, BAState) into the engine's arrays the way the host adapter does: used imagesets get
    sequential indices (joint_optimization.cc:80-90); features whose id has no point are dropped, as
    ComputeFeatureIdToPointsIndex leaves them at index -1 and the reference's initialisation never keeps such
    features (ba_state.cc:78-91)."""
    used = np.asarray(image_used, dtype=bool)
    seq = np.cumsum(used) - 1
    xy, pt, img, cam = [], [], [], []
    for i, s in enumerate(ds.imagesets):
        if not used[i]:
            continue
        for c in range(ds.num_cameras):
            for f in s.features[c]:
                idx = mapping.get(int(f["id"]), -1)
                if idx < 0:
                    continue
                xy.append((f["x"], f["y"])); pt.append(idx); img.append(seq[i]); cam.append(c)
    pb = Problem(cameras, int(used.sum()), state.points.shape[0], np.array(xy, dtype=np.float32).reshape(-1, 2),
                 np.array(pt, dtype=np.int32), np.array(img, dtype=np.int32), np.array(cam, dtype=np.int32), fd_delta)
    st = State(state.rig_tr_global[used], state.camera_tr_rig, state.points, state.grids)
    return pb, st
