"""ctypes binding of libcalib_ba_hip.so (include/cba.h) and the Python host mirror of the reference's
``OptimizeJointly`` entry point.

The library is hand-written HIP for gfx950; there is no CPU fallback -- importing works anywhere, but
every compute call raises :class:`EngineError` when the shared object or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from .problem import CENTRAL_GENERIC, Camera, Problem, State

_HERE = os.path.dirname(os.path.abspath(__file__))
# CBA_HIP_LIB: another build of the same library (kernel A/B measurements, tools/gpu_ab.sh); default = the in-tree build
LIB_PATH = os.environ.get("CBA_HIP_LIB") or os.path.join(_HERE, "libcalib_ba_hip.so")


class EngineError(RuntimeError):
    pass


class CbaCamera(C.Structure):
    _fields_ = [("model_type", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("calib_min_x", C.c_int32), ("calib_min_y", C.c_int32),
                ("calib_max_x", C.c_int32), ("calib_max_y", C.c_int32),
                ("grid_w", C.c_int32), ("grid_h", C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p)
COLLECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)   # cba_collective_fn
COLL_ALLREDUCE_SUM, COLL_REDUCE_SCATTER_SUM, COLL_ALLGATHER = 0, 1, 2


class CbaSolverOptions(C.Structure):
    """cba_solver_options: scheduling options of the reduced solve (all zero = defaults)."""
    _fields_ = [("factor_tail_rows", C.c_int32), ("back_substitution", C.c_int32), ("elimination", C.c_int32), ("grid_strips", C.c_int32),
                ("grid_single_tile_tasks", C.c_int32)]


ELIMINATION_AUTO, ELIMINATION_POSE_FIRST, ELIMINATION_GRID_FIRST = 0, 1, 2


DEFAULT_FACTOR_TAIL_ROWS = 8192      # what factor_tail_rows = 0 selects on one GPU (include/cba.h; 6144 / 4096 in the distributed solve with 2-3 / >= 4 ranks)


class CbaConfig(C.Structure):
    _fields_ = [("n_cameras", C.c_int32), ("cameras", C.POINTER(CbaCamera)),
                ("n_images", C.c_int32), ("n_points", C.c_int32),
                ("numerical_diff_delta", C.c_double),
                ("localize_only", C.c_int32), ("eliminate_points", C.c_int32), ("device", C.c_int32),
                ("allreduce", ALLREDUCE_FN), ("allreduce_user", C.c_void_p),
                ("n_images_global", C.c_int32),
                ("reduce_buffer", C.c_void_p), ("reduce_buffer_doubles", C.c_int64),
                ("deterministic", C.c_int32), ("distributed_solve", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
                ("collective", COLLECTIVE_FN), ("collective_user", C.c_void_p), ("solver", CbaSolverOptions)]


class CbaFitReport(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("lambda_", C.c_double),
                ("iterations_performed", C.c_int32), ("lm_attempts", C.c_int32), ("t_pass", C.c_double), ("t_solve", C.c_double)]


class CbaReport(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("lambda_", C.c_double),
                ("accepted", C.c_int32), ("lm_attempts", C.c_int32),
                ("n_residuals_valid", C.c_int64), ("n_jacobians_dropped", C.c_int64),
                ("t_jac", C.c_double), ("t_solve", C.c_double), ("t_cost", C.c_double),
                ("t_accumulate", C.c_double), ("t_gemm", C.c_double), ("t_factor", C.c_double)]


# every symbol include/cba.h declares (tests check that the library exports all of them)
EXPORTED_SYMBOLS = [
    "cba_last_error", "cba_version", "cba_create", "cba_destroy", "cba_set_observations", "cba_set_state",
    "cba_get_state", "cba_get_last_projection", "cba_step", "cba_cost", "cba_project", "cba_unproject",
    "cba_schur_solve", "cba_debug_dump", "cba_debug_accumulate", "cba_set_straggler_threshold", "cba_debug_solve", "cba_debug_apply_update",
    "cba_total_dof", "cba_dense_dof", "cba_jacobian_record_doubles", "cba_reduce_buffer_doubles",
    "cba_kernel_stats", "cba_fit_grid_to_directions", "cba_prepare_device",
    "cba_model_create", "cba_model_destroy", "cba_model_set_grid", "cba_model_project", "cba_model_unproject",
    "cba_fd_redo_overflow", "cba_debug_fd_redo_counts", "cba_schur_solve_opt", "cba_set_fd_schedule",
    "cba_gridfirst_plan_query", "cba_elimination_order",
]

DUMP_COST_VECTOR, DUMP_PIXELS, DUMP_FLAGS, DUMP_JACOBIANS = 1, 2, 3, 4
DUMP_BLOCK_DIAG_H, DUMP_BLOCK_DIAG_B, DUMP_OFF_DIAG_H, DUMP_DENSE_H, DUMP_DENSE_B, DUMP_X = 5, 6, 7, 8, 9, 10
DUMP_TEST_COST_VECTOR = 11

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Loads libcalib_ba_hip.so; fails loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own copy of the HIP runtime (same SONAME libamdhip64.so.7).  Importing it first
    # makes the dynamic loader resolve our NEEDED entry to that copy, so that device pointers and
    # streams are shared by one runtime when torch.distributed does the all-reduce.
    # four concurrent engine streams + RCCL's own: ask the runtime for more than its default of four
    # hardware queues (only effective if HIP has not been initialised yet)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise EngineError(f"{LIB_PATH} is missing -- run `python -c 'import __graft_entry__ as g; g.build()'`; "
                          "the engine has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    L.cba_last_error.restype = C.c_char_p
    L.cba_version.restype = C.c_char_p
    L.cba_create.argtypes = [C.POINTER(CbaConfig), C.POINTER(vp)]
    L.cba_destroy.argtypes = [vp]
    L.cba_destroy.restype = None
    L.cba_set_observations.argtypes = [vp, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp]
    L.cba_set_state.argtypes = [vp, dp, dp, dp, C.POINTER(dp)]
    L.cba_get_state.argtypes = [vp, dp, dp, dp, C.POINTER(dp)]
    L.cba_get_last_projection.argtypes = [vp, dp]
    L.cba_step.argtypes = [vp, C.c_double, C.c_int32, C.c_double, C.POINTER(CbaReport)]
    L.cba_cost.argtypes = [vp, dp, C.POINTER(C.c_int64), dp]
    L.cba_project.argtypes = [C.POINTER(CbaCamera), dp, C.c_int64, dp, dp, dp, C.POINTER(C.c_uint8), C.c_int32]
    L.cba_unproject.argtypes = [C.POINTER(CbaCamera), dp, C.c_int64, dp, dp, dp, C.POINTER(C.c_uint8), C.c_int32]
    L.cba_schur_solve.argtypes = [C.c_int32, C.c_int32, C.c_int32, dp, dp, dp, dp, dp, dp, C.c_int32]
    L.cba_fit_grid_to_directions.argtypes = [C.POINTER(CbaCamera), dp, C.c_int64, dp, dp, C.c_int32, C.POINTER(CbaFitReport), C.c_int32]
    L.cba_debug_dump.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    L.cba_debug_accumulate.argtypes = [vp, dp]
    L.cba_set_straggler_threshold.argtypes = [vp, C.c_int32]
    L.cba_set_fd_schedule.argtypes = [vp, C.c_int32]
    L.cba_debug_solve.argtypes = [vp, C.c_double]
    L.cba_debug_apply_update.argtypes = [vp, dp]
    L.cba_total_dof.argtypes = [vp]
    L.cba_dense_dof.argtypes = [vp]
    L.cba_jacobian_record_doubles.argtypes = [vp]
    L.cba_reduce_buffer_doubles.argtypes = [C.POINTER(CbaConfig)]
    L.cba_reduce_buffer_doubles.restype = C.c_int64
    L.cba_kernel_stats.argtypes = [vp, C.c_int32, dp, dp, dp, C.POINTER(C.c_int32)]
    L.cba_prepare_device.argtypes = [C.c_int32]
    L.cba_model_create.argtypes = [C.POINTER(CbaCamera), dp, C.c_int32, C.POINTER(vp)]
    L.cba_model_destroy.argtypes = [vp]
    L.cba_model_destroy.restype = None
    L.cba_model_set_grid.argtypes = [vp, dp]
    L.cba_model_project.argtypes = [vp, C.c_int64, dp, dp, dp, C.POINTER(C.c_uint8)]
    L.cba_model_unproject.argtypes = [vp, C.c_int64, dp, dp, dp, C.POINTER(C.c_uint8)]
    L.cba_gridfirst_plan_query.argtypes = [C.POINTER(CbaCamera), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int64]
    L.cba_gridfirst_plan_query.restype = C.c_int64
    _lib = L
    return L


def gridfirst_plan(cameras: Sequence[Camera], n_images: int, n_points: int, strips: int = 0, single_tile_tasks: bool = False) -> dict:
    """cba_gridfirst_plan_query: the static plan of the grid-first elimination order (host only, no device).

    Keys: the header fields (G, Gf, n_rp, n_border, n_fact, n_pad, nbg, nbf, ntc, n_tasks0, mask_words, half_bandwidth, strips0),
    f_of_grid (G,), chains (n, 4), tasks (n, 4: kind | intervals << 8, r, c, first interval), ivals (n, 2), rowmask (nbf, words),
    flops (3,), gperm (list per camera)."""
    L = load()
    cams = (CbaCamera * len(cameras))(*[_cam_struct(c) for c in cameras])

    def q(what, dtype):
        n = L.cba_gridfirst_plan_query(cams, len(cameras), n_images, n_points, strips, int(single_tile_tasks), what, None, 0)
        if n < 0:
            _check(int(n), "cba_gridfirst_plan_query")
        out = np.zeros(int(n) // np.dtype(dtype).itemsize, dtype=dtype)
        if n:
            L.cba_gridfirst_plan_query(cams, len(cameras), n_images, n_points, strips, int(single_tile_tasks), what, out.ctypes.data_as(C.c_void_p), int(n))
        return out

    h = q(0, np.int32)
    names = ["G", "Gf", "n_rp", "n_border", "n_fact", "n_pad", "nbg", "nbf", "ntc", "n_chains", "n_tasks", "n_tasks0", "n_ivals",
             "mask_words", "half_bandwidth", "strips0"]
    plan = {k: int(v) for k, v in zip(names, h)}
    plan["f_of_grid"] = q(1, np.int32)
    plan["chains"] = q(2, np.int32).reshape(-1, 4)
    plan["tasks"] = q(3, np.int32).reshape(-1, 4)
    plan["ivals"] = q(4, np.int32).reshape(-1, 2)
    plan["rowmask"] = q(5, np.uint64).reshape(plan["nbf"], plan["mask_words"])
    plan["flops"] = q(6, np.float64)
    plan["gperm"] = [q(16 + c, np.int32) for c in range(len(cameras))]
    return plan


def prepare(device: int = 0) -> None:
    """Creates the engine's HIP streams for `device` now (cba_prepare_device).  Call it before the process launches
    its first GPU kernel: streams created that early run the dominant GEMM ~15 % faster than streams created later
    (measured on MI355X / ROCm 7.2 in round 2).  Optional -- cba_create does it on demand."""
    _check(load().cba_prepare_device(int(device)), "cba_prepare_device")


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().cba_last_error()
        raise EngineError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _cam_struct(cam: Camera) -> CbaCamera:
    return CbaCamera(cam.model_type, cam.width, cam.height, cam.calib_min_x, cam.calib_min_y,
                     cam.calib_max_x, cam.calib_max_y, cam.grid_w, cam.grid_h)


@dataclass
class StepReport:
    """OptimizationReport (libvis lm_optimizer.h:55-77) + the values OptimizeJointly returns by pointer."""
    initial_cost: float
    final_cost: float
    final_lambda: float
    accepted: bool
    lm_attempts: int
    n_residuals_valid: int
    n_jacobians_dropped: int
    t_jac: float
    t_solve: float
    t_cost: float
    t_accumulate: float
    t_gemm: float
    t_factor: float


class Engine:
    """Device-resident BA problem (cba_problem)."""

    def __init__(self, problem: Problem, device: int = 0, allreduce: Optional[Callable[[int, int], int]] = None,
                 n_images_global: int = 0, reduce_buffer_ptr: int = 0, reduce_buffer_doubles: int = 0,
                 last_projection: Optional[np.ndarray] = None, deterministic: bool = False,
                 allreduce_native: Optional[tuple] = None, distributed_solve: bool = False, rank: int = 0, world_size: int = 1,
                 collective: Optional[Callable[[int, int, int, int], int]] = None, collective_native: Optional[tuple] = None,
                 factor_tail_rows: int = 0, back_substitution_panels: bool = False, elimination: int = 0, grid_strips: int = 0,
                 grid_single_tile_tasks: bool = False):
        self.L = load()
        self.problem = problem
        self._cams = (CbaCamera * problem.n_cameras)(*[_cam_struct(c) for c in problem.cameras])
        self._cb = None
        if allreduce is not None:
            def _cb(ptr, count, user, _f=allreduce):
                try:
                    return int(_f(int(ptr), int(count)) or 0)
                except Exception:  # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = ALLREDUCE_FN(_cb)
        cb, user = (self._cb if self._cb is not None else ALLREDUCE_FN(0)), None
        if allreduce_native is not None:     # (C function pointer, user pointer), e.g. distributed.NativeRccl.fn / .user
            cb, user = C.cast(allreduce_native[0], ALLREDUCE_FN), allreduce_native[1]
        # collective(op, send_ptr, recv_ptr, count) -> 0: reduce-scatter / all-gather of the distributed solve (cba_collective_fn)
        self._ccb = None
        if collective is not None:
            def _ccb(op, send, recv, count, user, _f=collective):
                try:
                    return int(_f(int(op), int(send or 0), int(recv or 0), int(count)) or 0)
                except Exception:
                    import traceback
                    traceback.print_exc()
                    return 1
            self._ccb = COLLECTIVE_FN(_ccb)
        ccb, cuser = (self._ccb if self._ccb is not None else COLLECTIVE_FN(0)), None
        if collective_native is not None:
            ccb, cuser = C.cast(collective_native[0], COLLECTIVE_FN), collective_native[1]
        cfg = CbaConfig(problem.n_cameras, self._cams, problem.n_images, problem.n_points, problem.fd_delta,
                        int(problem.localize_only), int(problem.eliminate_points), device,
                        cb, user, n_images_global,
                        reduce_buffer_ptr or None, reduce_buffer_doubles, int(deterministic), int(distributed_solve), int(rank), int(world_size),
                        ccb, cuser, CbaSolverOptions(int(factor_tail_rows), int(bool(back_substitution_panels)), int(elimination), int(grid_strips), int(bool(grid_single_tile_tasks))))
        self._cfg = cfg
        self._h = C.c_void_p()
        _check(self.L.cba_create(C.byref(cfg), C.byref(self._h)), "cba_create")
        _check(self.L.cba_set_observations(
            self._h, problem.n_obs, problem.obs_xy.ctypes.data_as(C.POINTER(C.c_float)),
            problem.obs_point.ctypes.data_as(C.POINTER(C.c_int32)),
            problem.obs_image.ctypes.data_as(C.POINTER(C.c_int32)),
            problem.obs_camera.ctypes.data_as(C.POINTER(C.c_int32)),
            None if last_projection is None else _dp(np.ascontiguousarray(last_projection, dtype=np.float64).reshape(-1, 2))),
            "cba_set_observations")

    @staticmethod
    def reduce_buffer_doubles(problem: Problem, distributed_solve: bool = False, world_size: int = 1) -> int:
        cams = (CbaCamera * problem.n_cameras)(*[_cam_struct(c) for c in problem.cameras])
        cfg = CbaConfig(problem.n_cameras, cams, problem.n_images, problem.n_points, problem.fd_delta,
                        int(problem.localize_only), int(problem.eliminate_points), 0, ALLREDUCE_FN(0), None, 0, None, 0, 0,
                        int(distributed_solve), 0, int(world_size), COLLECTIVE_FN(0), None, CbaSolverOptions(0, 0, 0, 0, 0))
        return int(load().cba_reduce_buffer_doubles(C.byref(cfg)))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self.L.cba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state ---------------------------------------------------------------------------------
    def set_state(self, st: State) -> None:
        grids = (C.POINTER(C.c_double) * len(st.grids))(*[_dp(g) for g in st.grids])
        _check(self.L.cba_set_state(self._h, _dp(st.rig_tr_global), _dp(st.camera_tr_rig), _dp(st.points), grids),
               "cba_set_state")

    def get_state(self, like: State) -> State:
        out = like.copy()
        grids = (C.POINTER(C.c_double) * len(out.grids))(*[_dp(g) for g in out.grids])
        _check(self.L.cba_get_state(self._h, _dp(out.rig_tr_global), _dp(out.camera_tr_rig), _dp(out.points), grids),
               "cba_get_state")
        return out

    def get_last_projection(self) -> np.ndarray:
        out = np.zeros((self.problem.n_obs, 2))
        _check(self.L.cba_get_last_projection(self._h, _dp(out)), "cba_get_last_projection")
        return out

    # -- optimisation ----------------------------------------------------------------------------
    def step(self, init_lambda: float = -1.0, max_lm_attempts: int = 50, init_lambda_factor: float = 0.00001) -> StepReport:
        r = CbaReport()
        _check(self.L.cba_step(self._h, init_lambda, max_lm_attempts, init_lambda_factor, C.byref(r)), "cba_step")
        return StepReport(r.initial_cost, r.final_cost, r.lambda_, bool(r.accepted), r.lm_attempts,
                          r.n_residuals_valid, r.n_jacobians_dropped, r.t_jac, r.t_solve, r.t_cost,
                          r.t_accumulate, r.t_gemm, r.t_factor)

    def cost(self, want_vector: bool = False):
        cost = C.c_double(0)
        nv = C.c_int64(0)
        vec = np.zeros(self.problem.n_obs) if want_vector else None
        _check(self.L.cba_cost(self._h, C.byref(cost), C.byref(nv), _dp(vec) if vec is not None else None), "cba_cost")
        return (cost.value, nv.value, vec) if want_vector else (cost.value, nv.value)

    def kernel_stats(self, which: int):
        s, f, b = C.c_double(0), C.c_double(0), C.c_double(0)
        n = C.c_int32(0)
        _check(self.L.cba_kernel_stats(self._h, which, C.byref(s), C.byref(f), C.byref(b), C.byref(n)), "cba_kernel_stats")
        return dict(seconds=s.value, flops=f.value, bytes=b.value, launches=n.value)

    def elimination_order(self) -> dict:
        """cba_elimination_order: which order this problem uses (cba_solver_options.elimination resolved) and its sizes."""
        out = (C.c_int32 * 4)()
        self.L.cba_elimination_order.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        self.L.cba_elimination_order.restype = C.c_int32
        order = int(self.L.cba_elimination_order(self._h, out))
        return dict(order={1: "pose-first", 2: "grid-first"}.get(order, "?"), strips=int(out[0]), dense_rows=int(out[1]), grid_rows=int(out[2]), chains=int(out[3]))

    # -- parity / debug ----------------------------------------------------------------------------
    def set_fd_schedule(self, schedule: int) -> None:
        """cba_set_fd_schedule: -1 = automatic (default), 0 = pooled finite-difference tasks, 1 = one task per lane."""
        _check(self.L.cba_set_fd_schedule(self._h, int(schedule)), "cba_set_fd_schedule")

    def set_straggler_threshold(self, outer_iterations: int) -> None:
        """cba_set_straggler_threshold: outer projection iterations before an observation goes to the straggler kernel."""
        _check(self.L.cba_set_straggler_threshold(self._h, int(outer_iterations)), "cba_set_straggler_threshold")

    def fd_redo_overflow(self) -> int:
        """cba_fd_redo_overflow: finite-difference tasks of the last Jacobian pass that found the follow-up list full (expected 0)."""
        self.L.cba_fd_redo_overflow.restype = C.c_int64
        self.L.cba_fd_redo_overflow.argtypes = [C.c_void_p]
        return int(self.L.cba_fd_redo_overflow(self._h))

    def fd_redo_counts(self):
        """cba_debug_fd_redo_counts: (main list, side-stream list, overflow) of the last Jacobian pass."""
        out = (C.c_int64 * 3)()
        self.L.cba_debug_fd_redo_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        _check(self.L.cba_debug_fd_redo_counts(self._h, out), "cba_debug_fd_redo_counts")
        return int(out[0]), int(out[1]), int(out[2])

    def debug_accumulate(self) -> float:
        cost = C.c_double(0)
        _check(self.L.cba_debug_accumulate(self._h, C.byref(cost)), "cba_debug_accumulate")
        return cost.value

    def debug_solve(self, lam: float) -> np.ndarray:
        _check(self.L.cba_debug_solve(self._h, lam), "cba_debug_solve")
        return self.dump(DUMP_X)

    def debug_apply_update(self, x: np.ndarray) -> None:
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.problem.total_dof
        _check(self.L.cba_debug_apply_update(self._h, _dp(x)), "cba_debug_apply_update")

    def dump(self, what: int) -> np.ndarray:
        p = self.problem
        n, bs, nb, dd = p.n_obs, p.block_size, p.n_blocks, p.dense_dof
        rec = self.L.cba_jacobian_record_doubles(self._h)
        shapes = {
            DUMP_COST_VECTOR: ((n,), np.float64), DUMP_TEST_COST_VECTOR: ((n,), np.float64),
            DUMP_PIXELS: ((n, 2), np.float64), DUMP_FLAGS: ((n,), np.uint8),
            DUMP_JACOBIANS: ((n, rec), np.float64),
            DUMP_BLOCK_DIAG_H: ((nb, bs, bs), np.float64), DUMP_BLOCK_DIAG_B: ((nb * bs,), np.float64),
            DUMP_OFF_DIAG_H: ((nb * bs, dd), np.float64), DUMP_DENSE_H: ((dd, dd), np.float64),
            DUMP_DENSE_B: ((dd,), np.float64), DUMP_X: ((p.total_dof,), np.float64),
        }
        shape, dt = shapes[what]
        out = np.zeros(shape, dtype=dt)
        _check(self.L.cba_debug_dump(self._h, what, out.ctypes.data_as(C.c_void_p), out.nbytes), "cba_debug_dump")
        return out


class DeviceModel:
    """Device-resident camera model (cba_model): the grid is uploaded once, scratch buffers are kept between calls."""

    def __init__(self, cam: Camera, grid: np.ndarray, device: int = 0):
        self.L = load()
        self.cam = cam
        g = np.ascontiguousarray(grid, dtype=np.float64)
        cs = _cam_struct(cam)
        self._h = C.c_void_p()
        _check(self.L.cba_model_create(C.byref(cs), _dp(g), device, C.byref(self._h)), "cba_model_create")

    def close(self):
        if getattr(self, "_h", None):
            self.L.cba_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_grid(self, grid: np.ndarray) -> None:
        g = np.ascontiguousarray(grid, dtype=np.float64)
        _check(self.L.cba_model_set_grid(self._h, _dp(g)), "cba_model_set_grid")

    def project(self, local_points: np.ndarray, init: Optional[np.ndarray] = None):
        pts = np.ascontiguousarray(local_points, dtype=np.float64).reshape(-1, 3)
        n = pts.shape[0]
        px = np.zeros((n, 2)); ok = np.zeros(n, dtype=np.uint8)
        ini = None if init is None else np.ascontiguousarray(init, dtype=np.float64).reshape(-1, 2)
        _check(self.L.cba_model_project(self._h, n, _dp(pts), _dp(ini) if ini is not None else None, _dp(px),
                                        ok.ctypes.data_as(C.POINTER(C.c_uint8))), "cba_model_project")
        return px, ok.astype(bool)

    def unproject(self, pixels: np.ndarray, with_jacobian: bool = False):
        px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
        n = px.shape[0]
        lines = np.zeros((n, 6)); ok = np.zeros(n, dtype=np.uint8)
        jac = np.zeros((n, 6, 2)) if with_jacobian else None
        _check(self.L.cba_model_unproject(self._h, n, _dp(px), _dp(lines), _dp(jac) if jac is not None else None,
                                          ok.ctypes.data_as(C.POINTER(C.c_uint8))), "cba_model_unproject")
        return (lines, jac, ok.astype(bool)) if with_jacobian else (lines, ok.astype(bool))


# ---- stateless model-level API (CameraModel::Project / Unproject) -----------------------------------
def project(cam: Camera, grid: np.ndarray, local_points: np.ndarray, init: Optional[np.ndarray] = None, device: int = 0):
    L = load()
    g = np.ascontiguousarray(grid, dtype=np.float64)
    pts = np.ascontiguousarray(local_points, dtype=np.float64).reshape(-1, 3)
    n = pts.shape[0]
    px = np.zeros((n, 2))
    ok = np.zeros(n, dtype=np.uint8)
    ini = None if init is None else np.ascontiguousarray(init, dtype=np.float64).reshape(-1, 2)
    cs = _cam_struct(cam)
    _check(L.cba_project(C.byref(cs), _dp(g), n, _dp(pts), _dp(ini) if ini is not None else None, _dp(px),
                         ok.ctypes.data_as(C.POINTER(C.c_uint8)), device), "cba_project")
    return px, ok.astype(bool)


def unproject(cam: Camera, grid: np.ndarray, pixels: np.ndarray, with_jacobian: bool = False, device: int = 0):
    L = load()
    g = np.ascontiguousarray(grid, dtype=np.float64)
    px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    n = px.shape[0]
    lines = np.zeros((n, 6))
    jac = np.zeros((n, 6, 2)) if with_jacobian else None
    ok = np.zeros(n, dtype=np.uint8)
    cs = _cam_struct(cam)
    _check(L.cba_unproject(C.byref(cs), _dp(g), n, _dp(px), _dp(lines), _dp(jac) if jac is not None else None,
                           ok.ctypes.data_as(C.POINTER(C.c_uint8)), device), "cba_unproject")
    return (lines, jac, ok.astype(bool)) if with_jacobian else (lines, ok.astype(bool))


def schur_solve(block_diag_H: np.ndarray, off_diag_H: np.ndarray, dense_H: np.ndarray, block_diag_b: np.ndarray,
                dense_b: np.ndarray, device: int = 0, factor_tail_rows: int = 0, back_substitution_panels: bool = False) -> np.ndarray:
    """LMOptimizer::SolveWithSchurComplementDenseOffDiag on host arrays (reference layout); the two keyword options are
    cba_solver_options (scheduling only)."""
    L = load()
    bD = np.ascontiguousarray(block_diag_H, dtype=np.float64)
    nb, bs = bD.shape[0], bD.shape[1]
    oH = np.ascontiguousarray(off_diag_H, dtype=np.float64)
    dH = np.ascontiguousarray(dense_H, dtype=np.float64)
    bb = np.ascontiguousarray(block_diag_b, dtype=np.float64)
    db = np.ascontiguousarray(dense_b, dtype=np.float64)
    dd = dH.shape[0]
    x = np.zeros(nb * bs + dd)
    if factor_tail_rows or back_substitution_panels:
        opt = CbaSolverOptions(int(factor_tail_rows), int(bool(back_substitution_panels)), 0, 0, 0)
        L.cba_schur_solve_opt.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.POINTER(C.c_double)] * 6 + [C.POINTER(CbaSolverOptions), C.c_int32]
        _check(L.cba_schur_solve_opt(bs, nb, dd, _dp(bD), _dp(oH), _dp(dH), _dp(bb), _dp(db), _dp(x), C.byref(opt), device), "cba_schur_solve_opt")
    else:
        _check(L.cba_schur_solve(bs, nb, dd, _dp(bD), _dp(oH), _dp(dH), _dp(bb), _dp(db), _dp(x), device), "cba_schur_solve")
    return x


# ---- host mirror of the reference entry point ---------------------------------------------------------
def fit_grid_to_directions(cam: Camera, grid: np.ndarray, grid_points: np.ndarray, directions: np.ndarray,
                           max_iteration_count: int, device: int = 0):
    """cba_fit_grid_to_directions (CentralGenericModel::FitToPixelDirectionsImpl).  Returns (new grid (G,3), report dict)."""
    L = load()
    g = np.ascontiguousarray(grid, dtype=np.float64).reshape(-1, 3).copy()
    gp = np.ascontiguousarray(grid_points, dtype=np.float64).reshape(-1, 2)
    d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
    assert gp.shape[0] == d.shape[0] and g.shape[0] == cam.grid_points
    cs = _cam_struct(cam)
    rep = CbaFitReport()
    _check(L.cba_fit_grid_to_directions(C.byref(cs), _dp(g), gp.shape[0], _dp(gp) if gp.size else None,
                                        _dp(d) if d.size else None, max_iteration_count, C.byref(rep), device),
           "cba_fit_grid_to_directions")
    return g, dict(initial_cost=rep.initial_cost, final_cost=rep.final_cost, final_lambda=rep.lambda_,
                   iterations=rep.iterations_performed, lm_attempts=rep.lm_attempts, t_pass=rep.t_pass, t_solve=rep.t_solve)


def optimize_jointly(problem: Problem, state: State, max_iteration_count: int, init_lambda: float = -1.0,
                     device: int = 0, engine: Optional[Engine] = None, print_progress: bool = False):
    """Python mirror of ``vis::OptimizeJointly`` (APP/bundle_adjustment/joint_optimization.h:53-70).

    ``problem`` carries numerical_diff_delta / localize_only / eliminate_points; regularization_weight
    must be 0 as in the reference (the branch is disabled there, joint_optimization.cc:299-305).
    Returns (final_cost, final_lambda, performed_an_iteration, new_state, reports).
    """
    own = engine is None
    eng = engine or Engine(problem, device)
    try:
        eng.set_state(state)
        final_cost = -1.0
        final_lambda = init_lambda
        performed = False
        reports: List[StepReport] = []
        for it in range(max_iteration_count):   # joint_optimization.cc:906-940
            rep = eng.step(init_lambda)
            reports.append(rep)
            final_cost = rep.final_cost
            init_lambda = final_lambda = rep.final_lambda
            if print_progress:
                print(f"LMOptimizer: [{it}] cost {rep.initial_cost:.9g} -> {rep.final_cost:.9g}, lambda {rep.final_lambda:.3g}, "
                      f"attempts {rep.lm_attempts}")
            if not rep.accepted:
                break
            performed = True
        return final_cost, final_lambda, performed, eng.get_state(state), reports
    finally:
        if own:
            eng.close()


def run_bundle_adjustment(engine: Engine, state: State, max_iteration_count: int = 100,
                          cost_reduction_threshold: float = 1e-4, print_progress: bool = False):
    """Outer convergence loop of the reference's ``RunBundleAdjustment`` (APP/calibration.cc:187-304):
    repeated ``OptimizeJointly(max_iteration_count=1)`` carrying lambda, stop when no update was
    performed or ``cost >= last_cost - threshold`` (:298).  The state stays device-resident between
    iterations (the reference rebuilds everything per call).  Gauge beautification
    (ChooseNiceCameraOrientation, :248-254) and state saving are outside the hot path and not done here.
    Returns (final_cost, iterations, reports)."""
    engine.set_state(state)
    lam = -1.0
    last_cost = float("inf")
    reports: List[StepReport] = []
    for it in range(max_iteration_count):
        rep = engine.step(lam)
        reports.append(rep)
        lam = rep.final_lambda
        if print_progress:
            print(f"[{it}] cost {rep.final_cost:.9g} lambda {lam:.3g} attempts {rep.lm_attempts}")
        if not rep.accepted:
            break
        if rep.final_cost >= last_cost - cost_reduction_threshold:
            break
        last_cost = rep.final_cost
    return reports[-1].final_cost, len(reports), reports
