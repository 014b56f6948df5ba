"""ctypes binding of oracle/_ref/libcalibref.so -- TEST INFRASTRUCTURE ONLY.

libcalibref.so is the REFERENCE's own code (generic_models/src/*.h, the generated Jacobian files, the local
parametrisations, b_spline.h, HuberLoss) compiled from /root/reference by oracle/Makefile against the Eigen /
libvis stand-ins in oracle/ref_shim.  It exists to check the restated oracle (oracle/cba_oracle.c) against code the
reference authors wrote; nothing in the product path may load it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libcalibref.so")
REFERENCE_ROOT = "/root/reference"


def reference_present() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "applications", "camera_calibration", "generic_models", "src"))


def build(force: bool = False) -> Optional[str]:
    """Builds oracle/_ref when the reference tree is present (this container); elsewhere the prebuilt file is used."""
    if reference_present():
        args = ["make", "-C", _HERE, "-s", "ref"] + (["-B"] if force else [])
        subprocess.check_call(args)
    return LIB_PATH if os.path.exists(LIB_PATH) else None


def available() -> bool:
    return build() is not None


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = build()
        if path is None:
            raise RuntimeError("oracle/_ref/libcalibref.so is missing and /root/reference is not present")
        L = C.CDLL(path)
        dp, fp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
        for name in ("ref_central_create", "ref_noncentral_create"):
            getattr(L, name).argtypes = [C.c_int] * 8 + [dp]
            getattr(L, name).restype = vp
        L.ref_central_read.argtypes = [C.c_char_p, ip]
        L.ref_central_read.restype = vp
        L.ref_central_get_grid.argtypes = [vp, dp]
        L.ref_central_destroy.argtypes = [vp]
        L.ref_noncentral_destroy.argtypes = [vp]
        for name in ("ref_central_project", "ref_central_project_init", "ref_central_unproject", "ref_noncentral_project",
                     "ref_noncentral_project_init", "ref_noncentral_unproject"):
            getattr(L, name).argtypes = [vp, dp, dp]
            getattr(L, name).restype = C.c_int
        for name in ("ref_central_unproject_jacobian", "ref_noncentral_unproject_jacobian"):
            getattr(L, name).argtypes = [vp, dp, dp, dp]
            getattr(L, name).restype = C.c_int
        L.ref_central_project_jacobian.argtypes = [vp, dp, dp, dp, C.c_double]
        L.ref_central_project_jacobian.restype = C.c_int
        L.ref_compute_jacobian.argtypes = [dp, dp, dp]
        L.ref_compute_rig_jacobian.argtypes = [dp, dp, dp, dp, dp]
        L.ref_central_unproject_patch.argtypes = [C.c_double, C.c_double, dp, dp, dp]
        L.ref_noncentral_unproject_patch.argtypes = [C.c_double, C.c_double, dp, dp, dp]
        L.ref_tangents.argtypes = [dp, dp, dp]
        L.ref_tangents_jacobian.argtypes = [dp, dp]
        L.ref_apply_direction_update.argtypes = [dp, C.c_double, C.c_double, dp]
        L.ref_apply_line_update.argtypes = [dp, dp, dp]
        L.ref_local_update_jacobian_wrt_direction.argtypes = [dp, dp]
        L.ref_convert_direction_to_local_update.argtypes = [dp, dp, dp]
        L.ref_apply_quaternion_update.argtypes = [dp, dp, dp]
        L.ref_quaternion_jacobian.argtypes = [dp, dp]
        L.ref_bspline_surface.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, dp]
        L.ref_bspline_surface_f32.argtypes = [fp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, fp]
        for name in ("ref_huber_cost_sq", "ref_huber_weight_sq", "ref_huber_cost", "ref_huber_weight"):
            getattr(L, name).argtypes = [C.c_double, C.c_double]
            getattr(L, name).restype = C.c_double
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_double))


class RefCamera:
    """One camera of the reference's generic_models classes (CentralGenericCamera / NoncentralGenericCamera<double>)."""

    def __init__(self, cam, grid: np.ndarray):
        """cam: any object with model_type, width, height, calib_min_x/.., grid_w, grid_h (problem.Camera / OrcCamera)."""
        L = lib()
        self.noncentral = int(cam.model_type) == 1
        self._grid = np.ascontiguousarray(grid, dtype=np.float64).ravel()
        make = L.ref_noncentral_create if self.noncentral else L.ref_central_create
        self._h = make(cam.width, cam.height, cam.calib_min_x, cam.calib_min_y, cam.calib_max_x, cam.calib_max_y,
                       cam.grid_w, cam.grid_h, _dp(self._grid))
        self._pfx = "ref_noncentral_" if self.noncentral else "ref_central_"

    @classmethod
    def read_yaml(cls, path: str):
        """CentralGenericCamera::Read; returns (camera, params8, grid[gh, gw, 3])."""
        L = lib()
        params = (C.c_int * 8)()
        h = L.ref_central_read(path.encode(), params)
        if not h:
            raise RuntimeError("reference reader rejected " + path)
        self = cls.__new__(cls)
        self.noncentral = False
        self._h = h
        self._pfx = "ref_central_"
        p = [int(v) for v in params]
        grid = np.zeros(p[6] * p[7] * 3)
        L.ref_central_get_grid(h, _dp(grid))
        self._grid = grid
        return self, p, grid.reshape(p[7], p[6], 3)

    def close(self):
        if self._h:
            (lib().ref_noncentral_destroy if self.noncentral else lib().ref_central_destroy)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def project(self, p, init=None):
        p = np.ascontiguousarray(p, dtype=np.float64)
        px = np.zeros(2) if init is None else np.array(init, dtype=np.float64)
        f = getattr(lib(), self._pfx + ("project" if init is None else "project_init"))
        ok = f(self._h, _dp(p), _dp(px))
        return bool(ok), px

    def unproject(self, px, jacobian=False):
        px = np.ascontiguousarray(px, dtype=np.float64)
        n = 6 if self.noncentral else 3
        out = np.zeros(n)
        if jacobian:
            jac = np.zeros(2 * n)
            ok = getattr(lib(), self._pfx + "unproject_jacobian")(self._h, _dp(px), _dp(out), _dp(jac))
            return bool(ok), out, jac.reshape(n, 2)
        ok = getattr(lib(), self._pfx + "unproject")(self._h, _dp(px), _dp(out))
        return bool(ok), out
