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
        _build_patched(force)
    return LIB_PATH if os.path.exists(LIB_PATH) else None


PATCHED_LIB_PATH = os.path.join(_HERE, "_ref", "patched", "libcalibref_ba.so")
PATCHED_DOUBLE_LIB_PATH = os.path.join(_HERE, "_ref", "patched_double", "libcalibref_ba.so")


def _build_patched(force: bool = False) -> None:
    """oracle/_ref/patched/libcalibref_ba.so: the reference with integration/reference.patch applied (SchurMode::HIP + the adapter), compiled
    by the same recipe and LINKED with camera_calibration_amd/libcalib_ba_hip.so (`make patched`).  Rebuilt when the patch, the glue or the
    HIP library's header changed; skipped where patch(1) or the HIP library is missing."""
    import shutil
    root = os.path.dirname(_HERE)
    hip = os.path.join(root, "camera_calibration_amd", "libcalib_ba_hip.so")
    if shutil.which("patch") is None or not os.path.exists(hip):
        return
    deps = [os.path.join(root, "integration", "reference.patch"), os.path.join(_HERE, "ref_ba_glue.cc"), os.path.join(_HERE, "ref_f14_glue.cc"),
            os.path.join(_HERE, "Makefile"), os.path.join(root, "include", "cba.h"), os.path.join(_HERE, "_ref", "libcalibref_ba.so")]
    def fresh(path, extra=()):
        return os.path.exists(path) and all(os.path.getmtime(path) >= os.path.getmtime(d) for d in list(deps) + list(extra) if os.path.exists(d))
    if force or not fresh(PATCHED_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s", "patched"])
    # the same patched build with the CPU test double of the C-ABI compiled in (oracle/cabi_test_double.c): executes the adapter without a GPU
    if force or not fresh(PATCHED_DOUBLE_LIB_PATH, [os.path.join(_HERE, "cabi_test_double.c")]):
        subprocess.check_call(["make", "-C", _HERE, "-s", "patched_double"])


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
        # round 3: LV accumulators + APP/models/central_grid.h (oracle/ref_lm.cc)
        L.ref_accum_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.ref_accum_create.restype = vp
        L.ref_accum_destroy.argtypes = [vp]
        L.ref_accum_add.argtypes = [vp, C.c_int, C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int, dp, ip, dp]
        L.ref_accum_add_residual.argtypes = [vp, dp]
        L.ref_accum_add_invalid.argtypes = [vp]
        L.ref_accum_get.argtypes = [vp, dp, dp, dp, dp, dp, dp, dp, ip]
        L.ref_central_grid_projection_jacobian.argtypes = [ip, dp, dp, dp, dp, C.c_double, ip, dp]
        L.ref_central_grid_projection_jacobian.restype = C.c_int
        L.ref_central_grid_subtract_delta.argtypes = [ip, dp, dp]
        L.ref_central_grid_subtract_delta.restype = C.c_int
        L.ref_noncentral_grid_projection_jacobian.argtypes = [ip, dp, dp, dp, dp, C.c_double, ip, dp]
        L.ref_noncentral_grid_projection_jacobian.restype = C.c_int
        L.ref_noncentral_grid_subtract_delta.argtypes = [ip, dp, dp]
        L.ref_noncentral_grid_subtract_delta.restype = C.c_int
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


def _ip(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_int))


def accumulate_records(pb, recs):
    """Feeds the per-observation records of an oracle Jacobian pass (oracle.OrcObsRecord: residual, Jacobian blocks, grid
    indices) through the REFERENCE's UpdateEquationAccumulator with the call sequence of AccumulateModelJacobian
    (APP/bundle_adjustment/joint_optimization.cc:479-590; oracle/ref_lm.cc).  Returns a dict with the reference's
    block_diag_H [nb, bs, bs], off_diag_H, dense_H, block_diag_b, dense_b, cost, cost_vector."""
    L = lib()
    bs, nb, dd = pb.block_size, pb.n_blocks, pb.dense_dof
    h = L.ref_accum_create(bs * nb, bs, dd)
    first_rig = 0
    first_cam = 6 * pb.n_images
    first_pts = first_cam + pb.rig_dof
    intr = [first_pts + 3 * pb.n_points]
    for c in pb.cameras[:-1]:
        intr.append(intr[-1] + c.intrinsics_param_count)
    if pb.eliminate_points:        # JointOptimizationState ordering with points first (joint_optimization.cc:142-170)
        first_pts = 0
        first_rig = 3 * pb.n_points
        first_cam = first_rig + 6 * pb.n_images
    mode = (1 if pb.rig_in_state else 0) | (2 if pb.eliminate_points else 0) | (4 if pb.localize_only else 0)
    try:
        for o in range(pb.n_obs):
            r = recs[o]
            res = np.array([r.residual[0], r.residual[1]])
            if not r.valid:
                L.ref_accum_add_invalid(h)
                continue
            if not r.has_jacobian:
                L.ref_accum_add_residual(h, _dp(res))
                continue
            cam = pb.cameras[int(pb.obs_camera[o])]
            kg = 0 if pb.localize_only else (32 if int(cam.model_type) == 0 else 80)
            gi = np.zeros(max(kg, 1), dtype=np.int32)
            gj = np.zeros(2 * max(kg, 1))
            if kg:
                gi[:] = np.array(r.grid_indices[:kg], dtype=np.int32) + intr[int(pb.obs_camera[o])]
                gj[:] = np.array(r.grid_jac[:2 * kg])
            L.ref_accum_add(h, mode, kg, _dp(res), first_rig + 6 * int(pb.obs_image[o]), _dp(np.array(r.pose_jac[:])),
                            first_cam + 6 * int(pb.obs_camera[o]), _dp(np.array(r.rig_jac[:])),
                            first_pts + 3 * int(pb.obs_point[o]), _dp(np.array(r.point_jac[:])), _ip(gi), _dp(gj))
        out = dict(block_diag_H=np.zeros((nb, bs, bs)), off_diag_H=np.zeros((nb * bs, dd)), dense_H=np.zeros((dd, dd)),
                   dense_b=np.zeros(dd), block_diag_b=np.zeros(nb * bs))
        cost = np.zeros(1)
        vec = np.zeros(pb.n_obs)
        n = (C.c_int * 1)()
        L.ref_accum_get(h, _dp(out["block_diag_H"]), _dp(out["off_diag_H"]), _dp(out["dense_H"]), _dp(out["dense_b"]),
                        _dp(out["block_diag_b"]), _dp(cost), _dp(vec), n)
        assert n[0] == pb.n_obs
        out["cost"] = float(cost[0])
        out["cost_vector"] = vec
        return out
    finally:
        L.ref_accum_destroy(h)


def _cam_params8(cam) -> np.ndarray:
    return np.array([cam.width, cam.height, cam.calib_min_x, cam.calib_min_y, cam.calib_max_x, cam.calib_max_y, cam.grid_w, cam.grid_h],
                    dtype=np.int32)


def central_grid_projection_jacobian(cam, grid, local_point, pixel, delta: float):
    """CentralGridModel::ProjectionJacobianWrtIntrinsics (APP/models/central_grid.h:187-245): (ok, indices[32], J[2, 32])."""
    L = lib()
    g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1, 3)
    tang = np.zeros((g.shape[0], 6))
    for i in range(g.shape[0]):
        L.ref_tangents(_dp(g[i].copy()), _dp(tang[i, :3]), _dp(tang[i, 3:]))
    idx = np.zeros(32, dtype=np.int32)
    J = np.zeros(64)
    ok = L.ref_central_grid_projection_jacobian(_ip(_cam_params8(cam)), _dp(g.ravel()), _dp(tang.ravel()),
                                                _dp(np.ascontiguousarray(local_point, dtype=np.float64)),
                                                _dp(np.ascontiguousarray(pixel, dtype=np.float64)), float(delta), _ip(idx), _dp(J))
    return int(ok), idx, J.reshape(2, 32)


def central_grid_subtract_delta(cam, grid, delta):
    """CentralGridModel::SubtractDelta (central_grid.h:168-185) on a copy of the grid."""
    g = np.ascontiguousarray(grid, dtype=np.float64).ravel().copy()
    ok = lib().ref_central_grid_subtract_delta(_ip(_cam_params8(cam)), _dp(g), _dp(np.ascontiguousarray(delta, dtype=np.float64)))
    if not ok:
        raise ValueError("grid size not compiled into oracle/ref_lm.cc")
    return g.reshape(-1, 3)


def noncentral_grid_projection_jacobian(cam, grids, local_point, pixel, delta: float):
    """NoncentralGenericModel::ProjectionJacobianWrtIntrinsics (APP/models/noncentral_generic.h:224-283): (ok, indices[80], J[2, 80]).
    grids: [2, G, 3] = direction grid, point grid."""
    L = lib()
    g = np.ascontiguousarray(grids, dtype=np.float64).reshape(2, -1, 3)
    tang = np.zeros((g.shape[1], 6))
    for i in range(g.shape[1]):
        L.ref_tangents(_dp(g[0, i].copy()), _dp(tang[i, :3]), _dp(tang[i, 3:]))
    idx = np.zeros(80, dtype=np.int32)
    J = np.zeros(160)
    ok = L.ref_noncentral_grid_projection_jacobian(_ip(_cam_params8(cam)), _dp(g.ravel()), _dp(tang.ravel()),
                                                   _dp(np.ascontiguousarray(local_point, dtype=np.float64)),
                                                   _dp(np.ascontiguousarray(pixel, dtype=np.float64)), float(delta), _ip(idx), _dp(J))
    return int(ok), idx, J.reshape(2, 80)


def noncentral_grid_subtract_delta(cam, grids, delta):
    """NoncentralGenericModel::SubtractDelta (noncentral_generic.h:195-222) on a copy of the grids [2, G, 3]."""
    g = np.ascontiguousarray(grids, dtype=np.float64).ravel().copy()
    ok = lib().ref_noncentral_grid_subtract_delta(_ip(_cam_params8(cam)), _dp(g), _dp(np.ascontiguousarray(delta, dtype=np.float64)))
    if not ok:
        raise ValueError("grid size not compiled into oracle/ref_lm.cc")
    return g.reshape(2, -1, 3)


# ---- part 4: LV/lm_optimizer.h itself (oracle/ref_lmopt.cc -> _ref/libcalibref_lm.so) -----------------------------------
LM_LIB_PATH = os.path.join(_HERE, "_ref", "libcalibref_lm.so")
_lm_lib: Optional[C.CDLL] = None


def lm_available() -> bool:
    build()
    return os.path.exists(LM_LIB_PATH)


def lm_lib() -> C.CDLL:
    """The reference's LMOptimizer<double> (OptimizeImpl, CostIsSmallerThan, SolveWithSchurComplementDenseOffDiag) compiled from
    /root/reference/libvis/src/libvis/lm_optimizer.h; links liboracle.so for the per-observation numbers and Eigen's LDLT."""
    global _lm_lib
    if _lm_lib is None:
        build()
        if not os.path.exists(LM_LIB_PATH):
            raise RuntimeError("oracle/_ref/libcalibref_lm.so is missing and /root/reference is not present")
        from oracle import oracle as orc
        orc.lib()                               # liboracle.so first (the library resolves it through its rpath as well)
        L = C.CDLL(LM_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.ref_lmopt_schur_solve.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp]
        L.ref_lmopt_cost_is_smaller_than.argtypes = [dp, dp, C.c_int]
        L.ref_lmopt_cost_is_smaller_than.restype = C.c_int
        L.ref_lmopt_optimize_jointly.argtypes = [C.POINTER(orc.OrcProblem), C.POINTER(orc.OrcState), C.c_int, C.c_double, dp,
                                                 C.POINTER(C.c_int32), dp]
        L.ref_lmopt_optimize_jointly.restype = C.c_double
        _lm_lib = L
    return _lm_lib


def lmopt_schur_solve(system) -> np.ndarray:
    """LMOptimizer::SolveWithSchurComplementDenseOffDiag (LV/lm_optimizer.h:1247-1369) on an oracle.System (upper triangles; no
    lambda is added here, as in orc_schur_solve)."""
    x = np.zeros(system.n_blocks * system.block_size + system.dense_dof)
    lm_lib().ref_lmopt_schur_solve(system.block_size, system.n_blocks, system.dense_dof, _dp(system.block_diag_H),
                                   _dp(system.off_diag_H), _dp(system.dense_H), _dp(system.block_diag_b), _dp(system.dense_b), _dp(x))
    return x


def lmopt_cost_is_smaller_than(left: np.ndarray, right: np.ndarray) -> bool:
    """LMOptimizer::CostIsSmallerThan (LV/lm_optimizer.h:993-1011)."""
    left = np.ascontiguousarray(left, dtype=np.float64)
    right = np.ascontiguousarray(right, dtype=np.float64)
    assert left.shape == right.shape
    return bool(lm_lib().ref_lmopt_cost_is_smaller_than(_dp(left), _dp(right), int(left.size)))


def lmopt_optimize_jointly(oracle_problem, st, max_iteration_count: int = 1, init_lambda: float = -1.0):
    """OptimizeJointly's optimizer calls (APP/bundle_adjustment/joint_optimization.cc:797-812, :916-940) with the REFERENCE's
    LMOptimizer<double>::Optimize; in place on st (and on the problem's warm-start cache).  Returns dict(cost, final_lambda,
    performed, trace) with one trace row per outer iteration: initial_cost, final_cost, lambda, iterations_performed,
    cost-only passes (= LM attempts that reached the cost test), Jacobian passes."""
    lam = C.c_double(0)
    performed = C.c_int32(0)
    trace = np.zeros((max_iteration_count, 6))
    cost = lm_lib().ref_lmopt_optimize_jointly(C.byref(oracle_problem.c), C.byref(oracle_problem._state(st)), max_iteration_count,
                                               init_lambda, C.byref(lam), C.byref(performed), _dp(trace))
    return dict(cost=cost, final_lambda=lam.value, performed=bool(performed.value), trace=trace)


# ---------------------------------------------------------------------------------------------------
# the reference's outer-loop and report FUNCTIONS (oracle/_ref/libcalibref_f14.so; see ref_f14_glue.cc and the
# rule in oracle/Makefile that pipes their line ranges from /root/reference into the compiler)
# ---------------------------------------------------------------------------------------------------
F14_LIB_PATH = os.path.join(_HERE, "_ref", "libcalibref_f14.so")
_f14_lib: Optional[C.CDLL] = None


def f14_available() -> bool:
    build()
    return os.path.exists(F14_LIB_PATH)


def f14_lib() -> C.CDLL:
    global _f14_lib
    if _f14_lib is None:
        build()
        if not os.path.exists(F14_LIB_PATH):
            raise RuntimeError("oracle/_ref/libcalibref_f14.so is missing and /root/reference is not present")
        from oracle import oracle as orc
        orc.lib()
        L = C.CDLL(F14_LIB_PATH)
        dp, fp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint8)
        dpp = C.POINTER(dp)
        packed = [C.c_int, C.c_int, C.c_int, ip, C.c_int64, fp, ip, ip, ip]       # n_cameras n_images n_points cam8 n_obs xy point image camera
        L.ref_f1_choose_nice_camera_orientation.argtypes = [ip, dp, dp]
        L.ref_f1_scale_to_metric.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_float, C.c_int, ip, ip, ip, ip, C.c_int]
        L.ref_f1_scale_to_metric.restype = C.c_double
        L.ref_f1_delete_outlier_features.argtypes = [C.c_int] + packed + [dp, dp, dp, dpp, C.c_float, bp, bp]
        L.ref_f4_compute_all_reprojection_errors.argtypes = [C.c_int] + packed + [dp, dp, dp, dpp, bp, dp, fp, dp]
        L.ref_f4_compute_all_reprojection_errors.restype = C.c_int64
        L.ref_f4_reprojection_error_histogram.argtypes = [C.c_int, C.c_double, C.c_int64, dp, dp]
        L.ref_f4_reprojection_error_median.argtypes = [C.c_int64, dp]
        L.ref_f4_reprojection_error_median.restype = C.c_double
        L.ref_f1_run_bundle_adjustment.argtypes = [C.c_int, C.c_double, C.c_int] + packed + [dp, dp, dp, dpp, dp]
        _f14_lib = L
    return _f14_lib


class _Packed:
    """ctypes views of a camera_calibration_amd.problem.Problem / State pair (central cameras only; keeps the arrays alive)."""

    def __init__(self, pb, st):
        assert all(c.model_type == 0 for c in pb.cameras), "the reference-side model of ref_f14_model.h is the central-generic one"
        self.cam8 = np.concatenate([_cam_params8(c) for c in pb.cameras]).astype(np.int32)
        self.xy = np.ascontiguousarray(pb.obs_xy, dtype=np.float32)
        self.point = np.ascontiguousarray(pb.obs_point, dtype=np.int32)
        self.image = np.ascontiguousarray(pb.obs_image, dtype=np.int32)
        self.camera = np.ascontiguousarray(pb.obs_camera, dtype=np.int32)
        self.rig = np.ascontiguousarray(st.rig_tr_global, dtype=np.float64).copy()
        self.ctr = np.ascontiguousarray(st.camera_tr_rig, dtype=np.float64).copy()
        self.points = np.ascontiguousarray(st.points, dtype=np.float64).copy()
        self.grids = [np.ascontiguousarray(g, dtype=np.float64).copy() for g in st.grids]
        self.grid_ptrs = (C.POINTER(C.c_double) * len(self.grids))(*[_dp(g) for g in self.grids])
        self.head = [pb.n_cameras, pb.n_images, pb.n_points, _ip(self.cam8), int(pb.n_obs),
                     self.xy.ctypes.data_as(C.POINTER(C.c_float)), _ip(self.point), _ip(self.image), _ip(self.camera)]
        self.state = [_dp(self.rig), _dp(self.ctr), _dp(self.points), self.grid_ptrs]


def _bp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def f1_choose_nice_camera_orientation(cam, grid):
    """CentralGenericModel::ChooseNiceCameraOrientation (APP/models/central_generic.cc:570-621), the reference's text compiled on the
    reference's CentralGridModel.  Returns (rotation 3x3, rotated grid (G, 3))."""
    g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1, 3).copy()
    R = np.zeros(9)
    f14_lib().ref_f1_choose_nice_camera_orientation(_ip(_cam_params8(cam)), _dp(g), _dp(R))
    return R.reshape(3, 3), g


def f1_scale_to_metric(cell_length: float, feature_id_to_position, feature_id_to_points_index, points, rig_tr_global, camera_tr_rig):
    """ScaleToMetric (APP/calibration.cc:307-370) + BAState::ScaleState (ba_state.cc).  Returns (factor, points, rig_tr_global,
    camera_tr_rig) after scaling."""
    pts = np.ascontiguousarray(points, dtype=np.float64).copy()
    rig = np.ascontiguousarray(rig_tr_global, dtype=np.float64).copy()
    ctr = np.ascontiguousarray(camera_tr_rig, dtype=np.float64).copy()
    ids = np.array(list(feature_id_to_position.keys()), dtype=np.int32)
    pos = np.array([feature_id_to_position[int(i)] for i in ids], dtype=np.int32).reshape(-1, 2)
    mids = np.array(list(feature_id_to_points_index.keys()), dtype=np.int32)
    midx = np.array([feature_id_to_points_index[int(i)] for i in mids], dtype=np.int32)
    f = f14_lib().ref_f1_scale_to_metric(len(pts), _dp(pts), len(rig), _dp(rig), len(ctr), _dp(ctr), float(cell_length), len(ids), _ip(ids),
                                         _ip(pos), _ip(mids), _ip(midx), len(mids))
    return f, pts, rig, ctr


def f1_delete_outlier_features(camera_index: int, pb, st, outlier_removal_factor: float, image_used=None):
    """DeleteOutlierFeatures (APP/calibration.cc:62-184).  Returns (keep mask over the problem's observations, image_used)."""
    pk = _Packed(pb, st)
    used = (np.ones(pb.n_images, dtype=np.uint8) if image_used is None else np.asarray(image_used).astype(np.uint8).copy())
    keep = np.zeros(pb.n_obs, dtype=np.uint8)
    f14_lib().ref_f1_delete_outlier_features(camera_index, *pk.head, *pk.state, float(outlier_removal_factor), _bp(used), _bp(keep))
    return keep.astype(bool), used.astype(bool)


def f4_compute_all_reprojection_errors(camera_index: int, pb, st, image_used=None):
    """ComputeAllReprojectionErrors (APP/calibration_report.cc:101-148)."""
    pk = _Packed(pb, st)
    used = (np.ones(pb.n_images, dtype=np.uint8) if image_used is None else np.asarray(image_used).astype(np.uint8).copy())
    errors = np.zeros((pb.n_obs, 2)); feats = np.zeros((pb.n_obs, 2), dtype=np.float32); sm = np.zeros(2)
    n = f14_lib().ref_f4_compute_all_reprojection_errors(camera_index, *pk.head, *pk.state, _bp(used), _dp(errors),
                                                         feats.ctypes.data_as(C.POINTER(C.c_float)), _dp(sm))
    return dict(count=int(n), sum=float(sm[0]), max=float(sm[1]), errors=errors[:n].copy(), features=feats[:n].copy())


def f4_reprojection_error_histogram(resolution: int, extent_in_px: float, errors):
    """ComputeReprojectionErrorHistogram (APP/calibration_report.cc:151-168); hist[hy, hx]."""
    e = np.ascontiguousarray(errors, dtype=np.float64).reshape(-1, 2)
    hist = np.zeros((resolution, resolution))
    f14_lib().ref_f4_reprojection_error_histogram(resolution, float(extent_in_px), len(e), _dp(e), _dp(hist))
    return hist


def f4_reprojection_error_median(errors) -> float:
    """The reprojection_error_median line of WriteReportInfoFile (APP/calibration_report.cc:686-692), printed with 17 digits."""
    e = np.ascontiguousarray(errors, dtype=np.float64).reshape(-1, 2)
    return float(f14_lib().ref_f4_reprojection_error_median(len(e), _dp(e)))


def f1_run_bundle_adjustment(pb, st, max_iteration_count: int, cost_reduction_threshold: float, localize_only: bool = False):
    """RunBundleAdjustment (APP/calibration.cc:187-304, CPU branch) around the oracle's OptimizeJointly.  Returns (State, number of
    OptimizeJointly calls made, numerical_diff_delta the loop passed)."""
    pk = _Packed(pb, st)
    trace = np.zeros(4)
    f14_lib().ref_f1_run_bundle_adjustment(max_iteration_count, float(cost_reduction_threshold), int(localize_only), *pk.head, *pk.state,
                                           _dp(trace))
    out = st.copy()
    out.rig_tr_global[...] = pk.rig; out.camera_tr_rig[...] = pk.ctr; out.points[...] = pk.points
    for g, h in zip(out.grids, pk.grids):
        g[...] = h.reshape(g.shape)
    return out, int(trace[0]), float(trace[1])


# ---------------------------------------------------------------------------------------------------
# The reference's ENTIRE CPU bundle-adjustment path (oracle/_ref/libcalibref_ba.so; see ref_ba_glue.cc): joint_optimization.cc, both
# generic models and lm_optimizer.h compiled whole, plus the F1 / F4 functions on the reference's real models
# ---------------------------------------------------------------------------------------------------
BA_LIB_PATH = os.path.join(_HERE, "_ref", "libcalibref_ba.so")
_ba_lib: Optional[C.CDLL] = None


def ba_available() -> bool:
    build()
    return os.path.exists(BA_LIB_PATH)


def ba_lib() -> C.CDLL:
    global _ba_lib
    if _ba_lib is None:
        build()
        if not os.path.exists(BA_LIB_PATH):
            raise RuntimeError("oracle/_ref/libcalibref_ba.so is missing and /root/reference is not present")
        from oracle import oracle as orc
        orc.lib()
        L = C.CDLL(BA_LIB_PATH)
        dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)
        dpp = C.POINTER(dp)
        L.ref_ba_optimize_jointly.argtypes = [C.c_int, C.c_int, C.c_int, ip, C.c_int64, fp, ip, ip, ip, dp, dp, dp, dpp, dp,
                                              C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, dp, ip]
        L.ref_ba_optimize_jointly.restype = C.c_double
        L.ref_ba_system.argtypes = [C.c_int, C.c_int, C.c_int, ip, C.c_int64, fp, ip, ip, ip, dp, dp, dp, dpp, dp, C.c_double, C.c_int, C.c_int,
                                    dp, dp, dp, dp, dp, dp, C.POINTER(C.c_int64)]
        L.ref_ba_system.restype = C.c_double
        packed = [C.c_int, C.c_int, C.c_int, ip, C.c_int64, fp, ip, ip, ip]
        L.ref_f1_run_bundle_adjustment.argtypes = [C.c_int, C.c_double, C.c_int] + packed + [dp, dp, dp, dpp, dp]
        L.ref_f1_choose_nice_camera_orientation.argtypes = [ip, dp, dp]
        L.ref_f3_fit_to_pixel_directions.argtypes = [ip, dp, C.c_int64, dp, dp, C.c_int]
        L.ref_f3_fit_to_dense_model.argtypes = [ip, C.c_int, C.c_int, dp, C.c_int, C.c_int, dp]
        L.ref_f3_fit_to_dense_model.restype = C.c_int
        L.ref_f1_calibrate_refinement_stage.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, ip, C.c_int, C.c_int, C.c_int, ip, C.c_int64,
                                                        fp, ip, ip, ip, dp, dp, dp, dpp, ip, dpp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), dp]
        L.ref_f1_calibrate_refinement_stage.restype = C.c_int
        L.ref_f3_resample_model.argtypes = [ip, dp, C.c_int, C.c_int, C.c_int, dp]
        L.ref_f3_resample_model.restype = C.c_int
        L.ref_f2_dataset_load_and_save.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int64)]
        L.ref_f2_save_camera_model.argtypes = [ip, dp, C.c_char_p]
        L.ref_f2_save_poses.argtypes = [C.c_int, C.POINTER(C.c_uint8), dp, C.c_char_p]
        L.ref_f2_save_points.argtypes = [C.c_int, dp, C.c_int, ip, ip, C.c_char_p]
        _ba_lib = L
    return _ba_lib


def patched_available() -> bool:
    build()
    return os.path.exists(PATCHED_LIB_PATH)


_patched_lib: Optional[C.CDLL] = None


def _bind_mode(L):
    dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.ref_ba_optimize_jointly_mode.argtypes = [C.c_int, C.c_int, C.c_int, ip, C.c_int64, fp, ip, ip, ip, dp, dp, dp, C.POINTER(dp), dp,
                                               C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, dp, ip, C.c_int]
    L.ref_ba_optimize_jointly_mode.restype = C.c_double
    return L


_patched_double_lib: Optional[C.CDLL] = None


def patched_double_available() -> bool:
    build()
    return os.path.exists(PATCHED_DOUBLE_LIB_PATH)


def patched_double_lib() -> C.CDLL:
    """The patched reference with the CPU test double of the C-ABI compiled in (hidden symbols; never the product's library): SchurMode::HIP runs the adapter of
    integration/reference.patch on the CPU."""
    global _patched_double_lib
    if _patched_double_lib is None:
        build()
        _patched_double_lib = _bind_mode(C.CDLL(PATCHED_DOUBLE_LIB_PATH))
    return _patched_double_lib


def patched_lib() -> C.CDLL:
    """The PATCHED reference (integration/reference.patch) as a library: the reference's own OptimizeJointly, generic models and LMOptimizer plus
    the adapter the patch adds, linked with camera_calibration_amd/libcalib_ba_hip.so.  schur_mode 0 = Dense (CPU), 5 = SchurMode::HIP."""
    global _patched_lib
    if _patched_lib is None:
        build()
        if not os.path.exists(PATCHED_LIB_PATH):
            raise RuntimeError("oracle/_ref/patched/libcalibref_ba.so is missing")
        L = C.CDLL(PATCHED_LIB_PATH)
        dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.ref_ba_optimize_jointly_mode.argtypes = [C.c_int, C.c_int, C.c_int, ip, C.c_int64, fp, ip, ip, ip, dp, dp, dp, C.POINTER(dp), dp,
                                                   C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, dp, ip, C.c_int]
        L.ref_ba_optimize_jointly_mode.restype = C.c_double
        _patched_lib = L
    return _patched_lib


SCHUR_MODE_DENSE, SCHUR_MODE_HIP = 0, 5          # APP/bundle_adjustment/joint_optimization.h:45-51 + the patch's SchurMode::HIP


def patched_optimize_jointly(pb, st, last_projection=None, max_iteration_count: int = 1, init_lambda: float = -1.0, schur_mode: int = SCHUR_MODE_DENSE,
                             lib=None):
    """vis::OptimizeJointly of the PATCHED reference tree on a Problem / State; in place on `st` and `last_projection`.  `lib`: patched_lib()
    (default: linked with the HIP library) or patched_double_lib() (the CPU test double of the C-ABI compiled in)."""
    cam9 = np.concatenate([np.concatenate([[c.model_type], _cam_params8(c)]) for c in pb.cameras]).astype(np.int32)
    xy = np.ascontiguousarray(pb.obs_xy, dtype=np.float32)
    lp = np.zeros((pb.n_obs, 2)) if last_projection is None else last_projection
    grids = (C.POINTER(C.c_double) * len(st.grids))(*[_dp(g) for g in st.grids])
    lam = C.c_double(0); performed = C.c_int(0)
    cost = (lib or patched_lib()).ref_ba_optimize_jointly_mode(pb.n_cameras, pb.n_images, pb.n_points, _ip(cam9), int(pb.n_obs),
                                                      xy.ctypes.data_as(C.POINTER(C.c_float)), _ip(pb.obs_point), _ip(pb.obs_image), _ip(pb.obs_camera),
                                                      _dp(st.rig_tr_global), _dp(st.camera_tr_rig), _dp(st.points), grids, _dp(lp),
                                                      max_iteration_count, float(init_lambda), float(pb.fd_delta), int(pb.localize_only),
                                                      int(pb.eliminate_points), C.byref(lam), C.byref(performed), int(schur_mode))
    return dict(cost=float(cost), final_lambda=lam.value, performed=bool(performed.value), last_projection=lp)


def ba_optimize_jointly(pb, st, last_projection=None, max_iteration_count: int = 1, init_lambda: float = -1.0):
    """The reference's own vis::OptimizeJointly (APP/bundle_adjustment/joint_optimization.cc:757-953, SchurMode::Dense, CPU) on a
    camera_calibration_amd.problem.Problem / State.  In place on `st` and on `last_projection` ((n_obs, 2) warm-start cache, zeros when
    None).  Returns dict(cost, final_lambda, performed, last_projection)."""
    cam9 = np.concatenate([np.concatenate([[c.model_type], _cam_params8(c)]) for c in pb.cameras]).astype(np.int32)
    xy = np.ascontiguousarray(pb.obs_xy, dtype=np.float32)
    lp = np.zeros((pb.n_obs, 2)) if last_projection is None else last_projection
    assert lp.dtype == np.float64 and lp.flags.c_contiguous and lp.shape == (pb.n_obs, 2)
    grids = (C.POINTER(C.c_double) * len(st.grids))(*[_dp(g) for g in st.grids])
    lam = C.c_double(0); performed = C.c_int(0)
    cost = ba_lib().ref_ba_optimize_jointly(pb.n_cameras, pb.n_images, pb.n_points, _ip(cam9), int(pb.n_obs),
                                            xy.ctypes.data_as(C.POINTER(C.c_float)), _ip(pb.obs_point), _ip(pb.obs_image), _ip(pb.obs_camera),
                                            _dp(st.rig_tr_global), _dp(st.camera_tr_rig), _dp(st.points), grids, _dp(lp),
                                            max_iteration_count, float(init_lambda), float(pb.fd_delta), int(pb.localize_only),
                                            int(pb.eliminate_points), C.byref(lam), C.byref(performed))
    return dict(cost=float(cost), final_lambda=lam.value, performed=bool(performed.value), last_projection=lp)


def ba_system(pb, st, last_projection=None):
    """One JointOptimizationCostFunction::Compute<true> of the reference (APP/bundle_adjustment/joint_optimization.cc:240-593) into the
    reference's UpdateEquationAccumulator: returns dict(cost, system (an oracle.System, upper triangles), cost_vector, last_projection)."""
    from oracle import oracle as orc
    cam9 = np.concatenate([np.concatenate([[c.model_type], _cam_params8(c)]) for c in pb.cameras]).astype(np.int32)
    xy = np.ascontiguousarray(pb.obs_xy, dtype=np.float32)
    lp = np.zeros((pb.n_obs, 2)) if last_projection is None else last_projection
    grids = (C.POINTER(C.c_double) * len(st.grids))(*[_dp(g) for g in st.grids])
    system = orc.System(pb.block_size, pb.n_blocks, pb.dense_dof)
    cost_vector = np.zeros(pb.n_obs); n_costs = C.c_int64(0)
    cost = ba_lib().ref_ba_system(pb.n_cameras, pb.n_images, pb.n_points, _ip(cam9), int(pb.n_obs), xy.ctypes.data_as(C.POINTER(C.c_float)),
                                  _ip(pb.obs_point), _ip(pb.obs_image), _ip(pb.obs_camera), _dp(st.rig_tr_global), _dp(st.camera_tr_rig),
                                  _dp(st.points), grids, _dp(lp), float(pb.fd_delta), int(pb.localize_only), int(pb.eliminate_points),
                                  _dp(system.block_diag_H), _dp(system.off_diag_H), _dp(system.dense_H), _dp(system.block_diag_b),
                                  _dp(system.dense_b), _dp(cost_vector), C.byref(n_costs))
    return dict(cost=float(cost), system=system, cost_vector=cost_vector, n_costs=int(n_costs.value), last_projection=lp)


def ba_run_bundle_adjustment(pb, st, max_iteration_count: int, cost_reduction_threshold: float, localize_only: bool = False):
    """RunBundleAdjustment (APP/calibration.cc:187-304) around the reference's own OptimizeJointly, on the reference's own
    CentralGenericModel: the reference's whole CPU calibration loop.  Returns (State, OptimizeJointly calls, numerical_diff_delta)."""
    pk = _Packed(pb, st)
    trace = np.zeros(4)
    ba_lib().ref_f1_run_bundle_adjustment(max_iteration_count, float(cost_reduction_threshold), int(localize_only), *pk.head, *pk.state,
                                          _dp(trace))
    out = st.copy()
    out.rig_tr_global[...] = pk.rig; out.camera_tr_rig[...] = pk.ctr; out.points[...] = pk.points
    for g, h in zip(out.grids, pk.grids):
        g[...] = h.reshape(g.shape)
    return out, int(trace[0]), float(trace[1])


def f3_fit_to_pixel_directions(cam, grid, pixels, directions, max_iteration_count: int):
    """CentralGenericModel::FitToPixelDirections (APP/models/central_generic.cc:419-431 and the LM behind it), the reference's own code."""
    g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1, 3).copy()
    px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
    ba_lib().ref_f3_fit_to_pixel_directions(_ip(_cam_params8(cam)), _dp(g), len(px), _dp(px), _dp(d), int(max_iteration_count))
    return g


def f3_fit_to_dense_model(cam, dense_model, subsample_step: int, max_iteration_count: int = 10):
    """CentralGenericModel::FitToDenseModel (APP/models/central_generic.cc:267-417), the reference's own code.  dense_model: (H, W, 3)
    with NaN for invalid pixels.  Returns the (G, 3) grid or None when the reference returns false."""
    dm = np.ascontiguousarray(dense_model, dtype=np.float64)
    g = np.zeros((cam.grid_w * cam.grid_h, 3))
    ok = ba_lib().ref_f3_fit_to_dense_model(_ip(_cam_params8(cam)), dm.shape[1], dm.shape[0], _dp(dm), int(subsample_step),
                                            int(max_iteration_count), _dp(g))
    return g if ok else None




def f2_dataset_load_and_save(path_in: str, path_out: str):
    """The reference's own LoadDataset then SaveDataset (APP/io/calibration_io.cc:51-246).  Returns None if either fails, else
    dict(cameras, imagesets, features, known_geometries) of what was loaded."""
    counts = (C.c_int64 * 4)()
    ok = ba_lib().ref_f2_dataset_load_and_save(path_in.encode(), path_out.encode(), counts)
    return dict(cameras=counts[0], imagesets=counts[1], features=counts[2], known_geometries=counts[3]) if ok else None


def f2_save_camera_model(cam, grid, path: str) -> bool:
    """The reference's own SaveCameraModel (:526-647).  grid: (G, 3) central; (2, G, 3) = direction, point for the non-central model."""
    cam9 = np.concatenate([[cam.model_type], _cam_params8(cam)]).astype(np.int32)
    g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1)
    return bool(ba_lib().ref_f2_save_camera_model(_ip(cam9), _dp(g), path.encode()))


def f2_save_poses(image_used, poses, path: str) -> bool:
    """The reference's own SavePoses (:785-839), incl. the .obj file next to it."""
    used = np.asarray(image_used).astype(np.uint8)
    p = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 7)
    return bool(ba_lib().ref_f2_save_poses(len(used), used.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(p), path.encode()))


def f2_save_points(points, feature_id_to_points_index, path: str) -> bool:
    """The reference's own SavePointsAndIndexMapping (:890-937), incl. the .obj file next to it."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    ids = np.array(list(feature_id_to_points_index.keys()), dtype=np.int32)
    idx = np.array([feature_id_to_points_index[int(i)] for i in ids], dtype=np.int32)
    return bool(ba_lib().ref_f2_save_points(len(pts), _dp(pts), len(ids), _ip(ids), _ip(idx), path.encode()))




def f3_resample_model(cam, grid, target_type: int, target_gw: int, target_gh: int):
    """ResampleModel (APP/calibration.cc:373-528) between generic models, the reference's own code.  grid: (G, 3) central or (2, G, 3) =
    direction, point.  Returns the new grid in the same convention ((2, G', 3) for a non-central target) or None."""
    cam9 = np.concatenate([[cam.model_type], _cam_params8(cam)]).astype(np.int32)
    g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1)
    out = np.zeros((2 if target_type else 1) * target_gw * target_gh * 3)
    ok = ba_lib().ref_f3_resample_model(_ip(cam9), _dp(g), int(target_type), int(target_gw), int(target_gh), _dp(out))
    if not ok:
        return None
    return out.reshape(2, -1, 3) if target_type else out.reshape(-1, 3)




def ba_calibrate_refinement_stage(pb, st, cell_length: float, positions_xy, num_pyramid_levels: int, approx_pixels_per_cell: int,
                                  outlier_removal_factor: float = 0.0, localize_only: bool = False, full_resolutions=None):
    """The refinement stage of Calibrate() (APP/calibration.cc:1030-1142), reference code all the way down (RunBundleAdjustment, OptimizeJointly,
    ResampleModel, DeleteOutlierFeatures, ScaleToMetric): pb / st at the coarsest pyramid level in, full resolution out.  ``full_resolutions``:
    [(gw, gh)] per camera of the result (to size the output grids).  Returns None where the reference returns false, else
    dict(cameras8 (grid sizes of the result), state, keep, image_used, optimize_calls)."""
    from camera_calibration_amd.problem import State
    cam9 = np.concatenate([np.concatenate([[c.model_type], _cam_params8(c)]) for c in pb.cameras]).astype(np.int32)
    xy = np.ascontiguousarray(pb.obs_xy, dtype=np.float32)
    pos = np.ascontiguousarray(positions_xy, dtype=np.int32).reshape(-1, 2)
    assert len(pos) == pb.n_points
    rig = np.ascontiguousarray(st.rig_tr_global, dtype=np.float64).copy(); ctr = np.ascontiguousarray(st.camera_tr_rig, dtype=np.float64).copy()
    pts = np.ascontiguousarray(st.points, dtype=np.float64).copy()
    gin = [np.ascontiguousarray(g, dtype=np.float64).copy() for g in st.grids]
    gout = [np.zeros((2 if c.model_type else 1) * fr[0] * fr[1] * 3) for c, fr in zip(pb.cameras, full_resolutions)]
    pin = (C.POINTER(C.c_double) * len(gin))(*[_dp(g) for g in gin]); pout = (C.POINTER(C.c_double) * len(gout))(*[_dp(g) for g in gout])
    cam9_out = np.zeros_like(cam9)
    used = np.zeros(pb.n_images, dtype=np.uint8); keep = np.zeros(pb.n_obs, dtype=np.uint8); trace = np.zeros(4)
    ok = ba_lib().ref_f1_calibrate_refinement_stage(num_pyramid_levels, approx_pixels_per_cell, float(outlier_removal_factor), int(localize_only),
                                                    float(cell_length), _ip(pos), pb.n_cameras, pb.n_images, pb.n_points, _ip(cam9), int(pb.n_obs),
                                                    xy.ctypes.data_as(C.POINTER(C.c_float)), _ip(pb.obs_point), _ip(pb.obs_image), _ip(pb.obs_camera),
                                                    _dp(rig), _dp(ctr), _dp(pts), pin, _ip(cam9_out), pout, used.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                    keep.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(trace))
    if not ok:
        return None
    grids = [g.reshape(2, -1, 3) if c.model_type else g.reshape(-1, 3) for c, g in zip(pb.cameras, gout)]
    return dict(grid_sizes=[(int(cam9_out[9 * c + 7]), int(cam9_out[9 * c + 8])) for c in range(pb.n_cameras)], state=State(rig, ctr, pts, grids),
                keep=keep.astype(bool), image_used=used.astype(bool), optimize_calls=int(trace[0]))

