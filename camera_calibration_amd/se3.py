"""Minimal numpy SE(3) helpers (poses stored as qw qx qy qz tx ty tz) for host-side set-up code.

Semantics follow Sophus as vendored by the reference (libvis/third_party/sophus/sophus/se3.hpp:183-207,
so3.hpp:215-232): group product = quaternion product + rotated translation; quaternions are unit.
Used by the synthetic generators and the host adapter only -- the hot path runs in HIP.
"""
from __future__ import annotations

import numpy as np


def quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx], axis=-1)


def quat_to_matrix(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - w * z); R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y); R[..., 2, 1] = 2 * (y * z + w * x); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def se3_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a * b, broadcasting over leading dims."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    q = quat_mul(a[..., :4], b[..., :4])
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    t = a[..., 4:] + np.einsum("...ij,...j->...i", quat_to_matrix(a[..., :4]), b[..., 4:])
    return np.concatenate([q, t], axis=-1)


def se3_exp(tangent: np.ndarray) -> np.ndarray:
    """Sophus SE3::exp with tangent = [upsilon(3), omega(3)]."""
    tangent = np.asarray(tangent, dtype=np.float64)
    u, w = tangent[..., :3], tangent[..., 3:]
    th2 = np.sum(w * w, axis=-1, keepdims=True)
    th = np.sqrt(th2)
    small = th < 1e-10
    ths = np.where(small, 1.0, th)
    imag = np.where(small, 0.5 - th2 / 48.0, np.sin(0.5 * ths) / ths)
    real = np.where(small, 1.0 - th2 / 8.0, np.cos(0.5 * ths))
    q = np.concatenate([real, imag * w], axis=-1)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    A = np.where(small, 0.5, (1 - np.cos(ths)) / np.where(small, 1.0, th2))
    B = np.where(small, 1.0 / 6.0, (ths - np.sin(ths)) / np.where(small, 1.0, th2 * ths))
    wu = np.cross(w, u)
    wwu = np.cross(w, wu)
    t = u + A * wu + B * wwu
    return np.concatenate([q, t], axis=-1)


def se3_identity(n: int | None = None) -> np.ndarray:
    e = np.array([1.0, 0, 0, 0, 0, 0, 0])
    return e if n is None else np.tile(e, (n, 1))


def transform_points(pose: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """R(q) p + t for one pose (7,) and points (...,3)."""
    R = quat_to_matrix(np.asarray(pose[:4], dtype=np.float64))
    return np.asarray(pts) @ R.T + pose[4:]
