"""The persistent tail launch of the reduced-system factorisation (k_ldlt_tail, cba_set_factor_tail_rows) against the blocked
multi-stream schedule and against LAPACK: same solution, whatever share of the matrix the tail takes -- nothing (0), the last
panels (1024), everything (the system is smaller than the default 6144), and sizes that put the junction on every kind of
panel boundary."""
import numpy as np
import pytest

from camera_calibration_amd import engine as eng
from oracle import oracle as orc
from parity_record import check, check_equal

pytestmark = pytest.mark.gpu


def _system(n_blocks, dense_dof, seed):
    rng = np.random.default_rng(seed)
    s = orc.System(6, n_blocks, dense_dof)
    A = rng.normal(size=(dense_dof, min(dense_dof, 768)))
    s.dense_H[:] = np.triu(A @ A.T + dense_dof * np.eye(dense_dof))
    s.off_diag_H[:] = rng.normal(size=(6 * n_blocks, dense_dof)) * 0.1
    for b in range(n_blocks):
        M = rng.normal(size=(6, 6))
        s.block_diag_H[b] = np.triu(M @ M.T + 6 * np.eye(6))
    s.block_diag_b[:] = rng.normal(size=6 * n_blocks)
    s.dense_b[:] = rng.normal(size=dense_dof)
    return s


@pytest.mark.parametrize("dense_dof", [63, 64, 65, 700, 1089, 2240, 3500])
def test_tail_launch_matches_blocked_schedule_and_lapack(dense_dof):
    case = f"factorisation tail, D = {dense_dof}"
    s = _system(12, dense_dof, seed=dense_dof)
    default_rows = eng.factor_tail_rows()
    try:
        xs = {}
        for rows in (0, 1024, default_rows):
            eng.set_factor_tail_rows(rows)
            xs[rows] = eng.schur_solve(s.block_diag_H, s.off_diag_H, s.dense_H, s.block_diag_b, s.dense_b)
            check_equal(case, f"tail rows {rows}: finite", int(np.count_nonzero(~np.isfinite(xs[rows]))))
    finally:
        eng.set_factor_tail_rows(default_rows)
    x_ref = orc.schur_solve(s)                     # Eigen's pivoted LDLT restated (oracle)
    scale = np.abs(x_ref).max()
    for rows, x in xs.items():
        check(case, f"tail rows {rows}: x vs oracle / |x|max", np.abs(x - x_ref).max() / scale, 5e-11)
    check(case, "tail (default) vs blocked schedule / |x|max", np.abs(xs[default_rows] - xs[0]).max() / scale, 5e-12)


@pytest.mark.parametrize("dense_dof", [65, 1089, 3500, 7000])
def test_back_substitution_dataflow_launch_matches_the_panel_version(dense_dof):
    """k_back_dataflow (one launch, {value, tag} pairs, agent-scope polling) against the panel kernels on the same factor."""
    case = f"back substitution, D = {dense_dof}"
    s = _system(12, dense_dof, seed=1000 + dense_dof)
    try:
        eng.set_back_substitution(False)
        x_panels = eng.schur_solve(s.block_diag_H, s.off_diag_H, s.dense_H, s.block_diag_b, s.dense_b)
        eng.set_back_substitution(True)
        x_flow = eng.schur_solve(s.block_diag_H, s.off_diag_H, s.dense_H, s.block_diag_b, s.dense_b)
    finally:
        eng.set_back_substitution(True)
    check_equal(case, "finite", int(np.count_nonzero(~np.isfinite(x_flow)) + np.count_nonzero(~np.isfinite(x_panels))))
    check(case, "dataflow vs panels / |x|max", np.abs(x_flow - x_panels).max() / np.abs(x_panels).max(), 5e-13)


def test_follow_up_list_of_the_finite_difference_kernel_does_not_overflow():
    from camera_calibration_amd import synthetic as syn
    pb, st, _ = syn.baseline_config(4, lambda cam, grid, pts: eng.project(cam, grid, pts), n_imagesets=30)
    e = eng.Engine(pb, last_projection=pb.obs_xy.astype(np.float64))
    e.set_state(st)
    e.debug_accumulate()
    check_equal("FD follow-up list", "tasks that found the list full", e.fd_redo_overflow())
    e.close()
