"""ctypes binding of the CPU ORACLE (oracle/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (camera_calibration_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cba_oracle.c")
    hdr = os.path.join(_HERE, "cba_oracle.h")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


class OrcCamera(C.Structure):
    _fields_ = [("model_type", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("calib_min_x", C.c_int32), ("calib_min_y", C.c_int32),
                ("calib_max_x", C.c_int32), ("calib_max_y", C.c_int32),
                ("grid_w", C.c_int32), ("grid_h", C.c_int32)]


class OrcProblem(C.Structure):
    _fields_ = [("n_cameras", C.c_int32), ("n_images", C.c_int32), ("n_points", C.c_int32),
                ("n_obs", C.c_int64), ("cams", C.POINTER(OrcCamera)),
                ("obs_xy", C.POINTER(C.c_float)), ("obs_point", C.POINTER(C.c_int32)),
                ("obs_image", C.POINTER(C.c_int32)), ("obs_camera", C.POINTER(C.c_int32)),
                ("last_projection", C.POINTER(C.c_double)), ("fd_delta", C.c_double),
                ("localize_only", C.c_int32), ("eliminate_points", C.c_int32)]


class OrcState(C.Structure):
    _fields_ = [("rig_tr_global", C.POINTER(C.c_double)), ("camera_tr_rig", C.POINTER(C.c_double)),
                ("points", C.POINTER(C.c_double)), ("grids", C.POINTER(C.POINTER(C.c_double)))]


class OrcSystem(C.Structure):
    _fields_ = [("block_size", C.c_int32), ("n_blocks", C.c_int32), ("dense_dof", C.c_int32),
                ("block_diag_H", C.POINTER(C.c_double)), ("off_diag_H", C.POINTER(C.c_double)),
                ("dense_H", C.POINTER(C.c_double)), ("block_diag_b", C.POINTER(C.c_double)),
                ("dense_b", C.POINTER(C.c_double))]


class OrcObsRecord(C.Structure):
    _fields_ = [("valid", C.c_int32), ("has_jacobian", C.c_int32),
                ("pixel", C.c_double * 2), ("residual", C.c_double * 2),
                ("cost", C.c_double), ("weight", C.c_double),
                ("pose_jac", C.c_double * 12), ("rig_jac", C.c_double * 12), ("point_jac", C.c_double * 6),
                ("grid_indices", C.c_int32 * 80), ("grid_jac", C.c_double * 160)]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.orc_bspline_surface.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, dp]
        L.orc_bspline_surface_slow.argtypes = L.orc_bspline_surface.argtypes
        L.orc_unproject.argtypes = [C.POINTER(OrcCamera), dp, C.c_double, C.c_double, dp]
        L.orc_unproject.restype = C.c_int
        L.orc_unproject_with_jacobian.argtypes = [C.POINTER(OrcCamera), dp, C.c_double, C.c_double, dp, dp]
        L.orc_unproject_with_jacobian.restype = C.c_int
        L.orc_project_with_initial_estimate.argtypes = [C.POINTER(OrcCamera), dp, dp, dp]
        L.orc_project_with_initial_estimate.restype = C.c_int
        L.orc_project.argtypes = [C.POINTER(OrcCamera), dp, dp, dp]
        L.orc_project.restype = C.c_int
        L.orc_grid_point_to_pixel.argtypes = [C.POINTER(OrcCamera), C.c_double, C.c_double, dp]
        L.orc_pixel_to_grid_point.argtypes = [C.POINTER(OrcCamera), C.c_double, C.c_double, dp]
        L.orc_tangents.argtypes = [dp, dp, dp]
        L.orc_apply_quaternion_update.argtypes = [dp, dp, dp]
        L.orc_se3_mul.argtypes = [dp, dp, dp]
        L.orc_se3_exp.argtypes = [dp, dp]
        L.orc_huber_cost_sq.argtypes = [C.c_double, C.c_double]
        L.orc_huber_cost_sq.restype = C.c_double
        L.orc_huber_weight_sq.argtypes = [C.c_double, C.c_double]
        L.orc_huber_weight_sq.restype = C.c_double
        L.orc_compute_jacobian.argtypes = [dp, dp, dp]
        L.orc_compute_rig_jacobian.argtypes = [dp, dp, dp, dp, dp]
        L.orc_dense_dof.argtypes = [C.POINTER(OrcProblem)]
        L.orc_dense_dof.restype = C.c_int32
        L.orc_total_dof.argtypes = [C.POINTER(OrcProblem)]
        L.orc_total_dof.restype = C.c_int32
        L.orc_cost_pass.argtypes = [C.POINTER(OrcProblem), C.POINTER(OrcState), dp]
        L.orc_cost_pass.restype = C.c_double
        L.orc_jacobian_pass.argtypes = [C.POINTER(OrcProblem), C.POINTER(OrcState), C.POINTER(OrcSystem), dp,
                                        C.POINTER(OrcObsRecord), C.c_int32, C.c_int32]
        L.orc_jacobian_pass.restype = C.c_double
        L.orc_schur_solve.argtypes = [C.POINTER(OrcSystem), dp]
        L.orc_ldlt_solve_upper.argtypes = [dp, C.c_int, dp, dp]
        L.orc_ldlt_solve_upper_unblocked.argtypes = [dp, C.c_int, dp, dp]
        L.orc_apply_update.argtypes = [C.POINTER(OrcProblem), C.POINTER(OrcState), dp, C.POINTER(OrcState)]
        L.orc_optimize_jointly.argtypes = [C.POINTER(OrcProblem), C.POINTER(OrcState), C.c_int, C.c_double,
                                           dp, ip, dp, ip]
        L.orc_optimize_jointly.restype = C.c_double
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_get_num_threads.restype = C.c_int
        L.orc_hardware_threads.restype = C.c_int
        _lib = L
    return _lib


def set_num_threads(n: int) -> int:
    """Host threads the oracle may use (n <= 0: all).  Results are bit-identical for every thread count."""
    lib().orc_set_num_threads(int(n))
    return int(lib().orc_get_num_threads())


def hardware_threads() -> int:
    return int(lib().orc_hardware_threads())


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def camera_struct(cam) -> OrcCamera:
    return OrcCamera(cam.model_type, cam.width, cam.height, cam.calib_min_x, cam.calib_min_y,
                     cam.calib_max_x, cam.calib_max_y, cam.grid_w, cam.grid_h)


class System:
    """Dense normal equations in the reference's layout (libvis lm_optimizer.h:660-685)."""

    def __init__(self, block_size: int, n_blocks: int, dense_dof: int):
        self.block_size, self.n_blocks, self.dense_dof = block_size, n_blocks, dense_dof
        self.block_diag_H = np.zeros((n_blocks, block_size, block_size))
        self.off_diag_H = np.zeros((n_blocks * block_size, dense_dof))
        self.dense_H = np.zeros((dense_dof, dense_dof))
        self.block_diag_b = np.zeros(n_blocks * block_size)
        self.dense_b = np.zeros(dense_dof)

    def struct(self) -> OrcSystem:
        return OrcSystem(self.block_size, self.n_blocks, self.dense_dof, _dp(self.block_diag_H),
                         _dp(self.off_diag_H), _dp(self.dense_H), _dp(self.block_diag_b), _dp(self.dense_b))

    def add_lambda(self, lam: float) -> None:
        i = np.arange(self.block_size)
        self.block_diag_H[:, i, i] += lam
        j = np.arange(self.dense_dof)
        self.dense_H[j, j] += lam


class OracleProblem:
    """Holds the ctypes views of a camera_calibration_amd.problem.Problem (keeps arrays alive)."""

    def __init__(self, problem, last_projection: Optional[np.ndarray] = None):
        self.p = problem
        self.cams = (OrcCamera * problem.n_cameras)(*[camera_struct(c) for c in problem.cameras])
        self.last_projection = (np.zeros((problem.n_obs, 2)) if last_projection is None
                                else np.ascontiguousarray(last_projection, dtype=np.float64))
        self.c = OrcProblem(problem.n_cameras, problem.n_images, problem.n_points, problem.n_obs, self.cams,
                            problem.obs_xy.ctypes.data_as(C.POINTER(C.c_float)), _ip(problem.obs_point),
                            _ip(problem.obs_image), _ip(problem.obs_camera), _dp(self.last_projection),
                            problem.fd_delta, int(problem.localize_only), int(problem.eliminate_points))

    # -- helpers -------------------------------------------------------------------------------
    def _state(self, st):
        grids = (C.POINTER(C.c_double) * len(st.grids))(*[_dp(g) for g in st.grids])
        s = OrcState(_dp(st.rig_tr_global), _dp(st.camera_tr_rig), _dp(st.points), grids)
        s._keep = (grids, st)
        return s

    def new_system(self) -> System:
        return System(self.p.block_size, self.p.n_blocks, self.p.dense_dof)

    # -- passes --------------------------------------------------------------------------------
    def cost_pass(self, st):
        cost_vec = np.zeros(self.p.n_obs)
        cost = lib().orc_cost_pass(C.byref(self.c), C.byref(self._state(st)), _dp(cost_vec))
        return cost, cost_vec

    def jacobian_pass(self, st, system: Optional[System] = None, want_records: bool = False,
                      img_begin: int = 0, img_end: Optional[int] = None):
        cost_vec = np.zeros(self.p.n_obs)
        recs = (OrcObsRecord * self.p.n_obs)() if want_records else None
        sysc = system.struct() if system is not None else None
        cost = lib().orc_jacobian_pass(C.byref(self.c), C.byref(self._state(st)),
                                       C.byref(sysc) if sysc is not None else None, _dp(cost_vec), recs,
                                       img_begin, self.p.n_images if img_end is None else img_end)
        return cost, cost_vec, recs

    def apply_update(self, st, x: np.ndarray):
        out = st.copy()
        x = np.ascontiguousarray(x, dtype=np.float64)
        lib().orc_apply_update(C.byref(self.c), C.byref(self._state(st)), _dp(x), C.byref(self._state(out)))
        return out

    def optimize_jointly(self, st, max_iteration_count: int = 1, init_lambda: float = -1.0):
        """In-place on st. Returns dict(cost, lambda, performed, timings, lm_attempts)."""
        lam = C.c_double(0)
        performed = C.c_int32(0)
        attempts = C.c_int32(0)
        timings = np.zeros(3)
        cost = lib().orc_optimize_jointly(C.byref(self.c), C.byref(self._state(st)), max_iteration_count,
                                          init_lambda, C.byref(lam), C.byref(performed), _dp(timings),
                                          C.byref(attempts))
        return dict(cost=cost, final_lambda=lam.value, performed=bool(performed.value),
                    t_jac=timings[0], t_solve=timings[1], t_cost=timings[2], lm_attempts=attempts.value)


def schur_solve(system: System) -> np.ndarray:
    x = np.zeros(system.n_blocks * system.block_size + system.dense_dof)
    lib().orc_schur_solve(C.byref(system.struct()), _dp(x))
    return x


def ldlt_solve_upper(A: np.ndarray, b: np.ndarray, unblocked: bool = False) -> np.ndarray:
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b)
    f = lib().orc_ldlt_solve_upper_unblocked if unblocked else lib().orc_ldlt_solve_upper
    f(_dp(A), A.shape[0], _dp(b), _dp(x))
    return x


# ---- model-level convenience wrappers (operate on one camera + its grid array) ---------------
def project(cam, grid: np.ndarray, local_points: np.ndarray, init: Optional[np.ndarray] = None):
    """CameraModel::Project (from the centre) or ProjectWithInitialEstimate for a batch of points."""
    cs = camera_struct(cam)
    g = np.ascontiguousarray(grid, dtype=np.float64)
    pts = np.ascontiguousarray(local_points, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((pts.shape[0], 2))
    ok = np.zeros(pts.shape[0], dtype=bool)
    L = lib()
    px = np.zeros(2)
    for i in range(pts.shape[0]):
        if init is None:
            ok[i] = bool(L.orc_project(C.byref(cs), _dp(g), _dp(pts[i].copy()), _dp(px)))
        else:
            px[:] = init[i]
            ok[i] = bool(L.orc_project_with_initial_estimate(C.byref(cs), _dp(g), _dp(pts[i].copy()), _dp(px)))
        out[i] = px
    return out, ok


def unproject(cam, grid: np.ndarray, pixels: np.ndarray, with_jacobian: bool = False):
    cs = camera_struct(cam)
    g = np.ascontiguousarray(grid, dtype=np.float64)
    px = np.asarray(pixels, dtype=np.float64).reshape(-1, 2)
    lines = np.zeros((px.shape[0], 6))
    jac = np.zeros((px.shape[0], 6, 2))
    ok = np.zeros(px.shape[0], dtype=bool)
    L = lib()
    for i in range(px.shape[0]):
        if with_jacobian:
            ok[i] = bool(L.orc_unproject_with_jacobian(C.byref(cs), _dp(g), px[i, 0], px[i, 1], _dp(lines[i]), _dp(jac[i])))
        else:
            ok[i] = bool(L.orc_unproject(C.byref(cs), _dp(g), px[i, 0], px[i, 1], _dp(lines[i])))
    return (lines, jac, ok) if with_jacobian else (lines, ok)


def se3_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    o = np.zeros(7)
    lib().orc_se3_mul(_dp(a), _dp(b), _dp(o))
    return o


def se3_exp(t):
    t = np.ascontiguousarray(t, dtype=np.float64)
    o = np.zeros(7)
    lib().orc_se3_exp(_dp(t), _dp(o))
    return o


# ---------------------------------------------------------------------------------------------------
# calibration report statistics -- restated per observation, scalar loops
#
# Pinned since round 5 to the reference's own code: ComputeAllReprojectionErrors / the histogram / the median rule /
# DeleteOutlierFeatures / ChooseNiceCameraOrientation / ScaleToMetric / RunBundleAdjustment are compiled FROM /root/reference (their
# files need Qt as a whole, so oracle/Makefile pipes the functions' line ranges into the compiler behind oracle/ref_f14_prelude.h;
# Quaterniond::FromTwoVectors / AngleAxisd come from the stand-in oracle/ref_shim_lm/Eigen/Geometry, restated from Eigen 3.3.7's
# published algorithm) and compared with these restatements in tests/test_oracle_vs_ref_outer_loop.py: decisions identical,
# values to 1e-14 (rotations) ... 1e-10 px (reprojection errors).
# ---------------------------------------------------------------------------------------------------
def all_reprojection_errors(camera_index: int, pb, st):
    """ComputeAllReprojectionErrors, APP/calibration_report.cc:101-148: per feature of one camera
    local = R(image_tr_global) * point + t, Project() from the centre, error = pixel - xy; failures skipped."""
    cam = pb.cameras[camera_index]
    grid = st.grids[camera_index]
    count, esum, emax = 0, 0.0, 0.0
    errors, feats = [], []
    for o in range(pb.n_obs):
        if pb.obs_camera[o] != camera_index:
            continue
        itg = se3_mul(st.camera_tr_rig[camera_index], st.rig_tr_global[pb.obs_image[o]])
        q = itg[:4]
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        local = R @ st.points[pb.obs_point[o]] + itg[4:]
        px, ok = project(cam, grid, local[None, :])
        if not ok[0]:
            continue
        e = px[0] - pb.obs_xy[o].astype(np.float64)
        m = float(np.sqrt(e[0] * e[0] + e[1] * e[1]))
        count += 1; esum += m; emax = max(emax, m)
        errors.append(e); feats.append(pb.obs_xy[o])
    return dict(count=count, sum=esum, max=emax, errors=np.array(errors).reshape(-1, 2), features=np.array(feats).reshape(-1, 2))


def reprojection_error_histogram(resolution: int, extent_in_px: float, errors):
    """ComputeReprojectionErrorHistogram, APP/calibration_report.cc:151-168 (scalar loop)."""
    hist = np.zeros((resolution, resolution))
    for ex, ey in np.asarray(errors).reshape(-1, 2):
        hx_f = resolution * 0.5 * ((ex / extent_in_px) + 1.0)
        hx = int(hx_f) - (1 if hx_f < 0 else 0)
        hy_f = resolution * 0.5 * ((ey / extent_in_px) + 1.0)
        hy = int(hy_f) - (1 if hy_f < 0 else 0)
        if 0 <= hx < resolution and 0 <= hy < resolution:
            hist[hy, hx] += 1.0
    return hist


def delete_outlier_features(camera_index: int, pb, st, outlier_removal_factor: float, image_used=None):
    """DeleteOutlierFeatures, APP/calibration.cc:62-184 restated with scalar loops.
    Returns (keep mask, image_used, threshold or None)."""
    used = np.ones(pb.n_images, dtype=bool) if image_used is None else np.array(image_used, dtype=bool)
    cam = pb.cameras[camera_index]
    grid = st.grids[camera_index]

    def project_obs(o):
        itg = se3_mul(st.camera_tr_rig[camera_index], st.rig_tr_global[pb.obs_image[o]])
        w, x, y, z = itg[:4]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        local = R @ st.points[pb.obs_point[o]] + itg[4:]
        px, ok = project(cam, grid, local[None, :])
        if not ok[0]:
            return None
        e = px[0] - pb.obs_xy[o].astype(np.float64)
        return float(np.sqrt(e[0] * e[0] + e[1] * e[1]))

    obs = [o for o in range(pb.n_obs) if pb.obs_camera[o] == camera_index and used[pb.obs_image[o]]]
    mags = {o: project_obs(o) for o in obs}
    errs = sorted(m for m in mags.values() if m is not None)
    keep = np.ones(pb.n_obs, dtype=bool)
    if len(errs) < 8:
        return keep, used, None
    n = len(errs)
    q1 = errs[int(np.float32(0.25) * np.float32(n) + np.float32(0.5))]
    q3 = errs[int(np.float32(0.75) * np.float32(n) + np.float32(0.5))]
    thr = q3 + float(np.float32(outlier_removal_factor)) * (q3 - q1)
    for i in range(pb.n_images):
        if not used[i]:
            continue
        feats = [o for o in obs if pb.obs_image[o] == i]
        left = 0
        for o in feats:
            if mags[o] is None or mags[o] > thr:
                keep[o] = False
            else:
                left += 1
        if left < 3:
            used[i] = False
    return keep, used, thr


# ---------------------------------------------------------------------------------------------------
# grid-only LM (CentralGenericModel::FitToPixelDirections / FitToDenseModel)
# ---------------------------------------------------------------------------------------------------
def _fit_sigs():
    L = lib()
    if getattr(L, "_fit_sigs_done", False):
        return L
    dp = C.POINTER(C.c_double)
    L.orc_fit_grid_pass.restype = C.c_double
    L.orc_fit_grid_pass.argtypes = [C.c_int32, C.c_int32, dp, C.c_int64, dp, dp, dp, dp, dp]
    L.orc_fit_grid_apply_update.restype = None
    L.orc_fit_grid_apply_update.argtypes = [C.c_int32, C.c_int32, dp, dp, dp]
    L.orc_fit_grid_to_points.restype = None
    L.orc_fit_grid_to_points.argtypes = [C.c_int32, C.c_int32, dp, C.c_int64, dp, dp, C.c_int32, dp]
    L.orc_grid_point_to_pixel.restype = None
    L.orc_pixel_to_grid_point.restype = None
    L._fit_sigs_done = True
    return L


def fit_grid_pass(gw: int, gh: int, grid, grid_points, directions, with_jacobian: bool):
    """One Compute<compute_jacobians> of CentralGenericBSplineDirectionCostFunction (central_generic.cc:153-225).
    Returns (cost, cost_vector[3n], H (dof x dof upper) or None, b or None)."""
    L = _fit_sigs()
    g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1, 3)
    gp = np.ascontiguousarray(grid_points, dtype=np.float64).reshape(-1, 2)
    d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
    n = gp.shape[0]
    cv = np.zeros(3 * n)
    dof = 2 * gw * gh
    H = np.zeros((dof, dof)) if with_jacobian else None
    b = np.zeros(dof) if with_jacobian else None
    cost = L.orc_fit_grid_pass(gw, gh, _dp(g), n, _dp(gp), _dp(d), _dp(H) if with_jacobian else None,
                               _dp(b) if with_jacobian else None, _dp(cv))
    return cost, cv, H, b


def fit_grid_apply_update(gw: int, gh: int, grid, x):
    L = _fit_sigs()
    g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1, 3)
    out = np.zeros_like(g)
    xx = np.ascontiguousarray(x, dtype=np.float64)
    L.orc_fit_grid_apply_update(gw, gh, _dp(g), _dp(xx), _dp(out))
    return out


def fit_grid_to_points(gw: int, gh: int, grid, grid_points, directions, max_iteration_count: int):
    """FitToPixelDirectionsImpl (central_generic.cc:551-568).  Returns (new grid, report dict)."""
    L = _fit_sigs()
    g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1, 3).copy()
    gp = np.ascontiguousarray(grid_points, dtype=np.float64).reshape(-1, 2)
    d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
    rep = np.zeros(4)
    L.orc_fit_grid_to_points(gw, gh, _dp(g), gp.shape[0], _dp(gp), _dp(d), max_iteration_count, _dp(rep))
    return g, dict(initial_cost=rep[0], final_cost=rep[1], iterations=int(rep[2]), final_lambda=rep[3])


def projection_jacobian_wrt_intrinsics(cam, grid: np.ndarray, local_point, pixel, delta: float):
    """ProjectionJacobianWrtIntrinsics (M5 / N3) of one point: (ok, indices[K], J[2, K])."""
    L = lib()
    L.orc_debug_projection_jacobian_wrt_intrinsics.argtypes = [C.POINTER(OrcCamera), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                               C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_int32),
                                                               C.POINTER(C.c_double)]
    L.orc_debug_projection_jacobian_wrt_intrinsics.restype = C.c_int
    K = 32 if int(cam.model_type) == 0 else 80
    g = np.ascontiguousarray(grid, dtype=np.float64).ravel()
    lp = np.ascontiguousarray(local_point, dtype=np.float64)
    px = np.ascontiguousarray(pixel, dtype=np.float64)
    idx = np.zeros(K, dtype=np.int32)
    J = np.zeros(2 * K)
    cs = camera_struct(cam)
    ok = L.orc_debug_projection_jacobian_wrt_intrinsics(C.byref(cs), _dp(g), _dp(lp), _dp(px), float(delta),
                                                        idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(J))
    return int(ok), idx, J.reshape(2, K)


def pixel_to_grid_point(cam, pixels):
    L = _fit_sigs()
    cs = camera_struct(cam)
    px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    out = np.zeros_like(px)
    for i in range(px.shape[0]):
        L.orc_pixel_to_grid_point(C.byref(cs), C.c_double(px[i, 0]), C.c_double(px[i, 1]), _dp(out[i]))
    return out


def grid_point_to_pixel(cam, gx: int, gy: int):
    L = _fit_sigs()
    cs = camera_struct(cam)
    out = np.zeros(2)
    L.orc_grid_point_to_pixel(C.byref(cs), C.c_double(float(gx)), C.c_double(float(gy)), _dp(out))
    return out


# ---------------------------------------------------------------------------------------------------
# ChooseNiceCameraOrientation / ScaleToMetric, scalar restatements
# ---------------------------------------------------------------------------------------------------
def choose_nice_camera_orientation(cam, grid):
    """CentralGenericModel::ChooseNiceCameraOrientation, APP/models/central_generic.cc:570-621 (per-pixel loops)."""
    g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1, 3)
    w, h = cam.width, cam.height
    line, ok = unproject(cam, g, np.array([[0.5 * w, 0.5 * h]]))
    forward = line[0, :3] if ok[0] else np.array([0.0, 0.0, 1.0])
    # Quaterniond::FromTwoVectors(forward, (0,0,1)).toRotationMatrix()
    v0 = forward / np.linalg.norm(forward); v1 = np.array([0.0, 0.0, 1.0])
    c = float(v1 @ v0)
    axis = np.cross(v0, v1); s_ = np.sqrt((1 + c) * 2)
    qw = s_ * 0.5; qx, qy, qz = axis / s_
    fr = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                   [2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)],
                   [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)]])
    right_sum = np.zeros(3); right_count = 0
    for y in range(max(0, h // 2 - 10), min(h - 1, h // 2 + 10) + 1):
        for x in range(min(w - 1, w // 2 + 11), w):
            l, o = unproject(cam, g, np.array([[x + 0.5, y + 0.5]]))
            if not o[0]:
                continue
            right_sum += l[0, :3]; right_count += 1
    if right_count > 0:
        r = fr @ (right_sum / right_count)
        a = np.arctan2(-r[1], r[0])
        rr = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    else:
        rr = np.eye(3)
    rot = rr @ fr
    return rot, np.array([rot @ v for v in g])


def scale_to_metric_factor(known_geometries, points, feature_id_to_points_index):
    """ScaleToMetric, APP/calibration.cc:307-370: exp(mean log(ideal / actual neighbour distance))."""
    log_sum, count = 0.0, 0
    for cell, id_to_pos in known_geometries:
        pos_to_idx = {tuple(p): feature_id_to_points_index[f] for f, p in id_to_pos.items() if f in feature_id_to_points_index}
        for f, p in id_to_pos.items():
            if tuple(p) not in pos_to_idx:
                continue
            i = pos_to_idx[tuple(p)]
            for n in ((1, 0), (0, 1)):
                q = (p[0] + n[0], p[1] + n[1])
                if q not in pos_to_idx:
                    continue
                log_sum += np.log(cell / np.linalg.norm(points[i] - points[pos_to_idx[q]])); count += 1
    return float(np.exp(log_sum / count))
