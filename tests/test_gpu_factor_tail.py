"""The dataflow launches of the reduced-system factorisation (k_ldlt_tail) against the oracle's pivoted LDL^T: same solution
whatever share of the matrix the final launch takes -- the last 1024 rows (super-panels in front of it at the larger sizes),
everything (the system is smaller than the default 8192) -- at sizes that put the junctions on every kind of block boundary, and
the two back substitutions against each other.  Scheduling options travel per call (cba_solver_options)."""
import numpy as np
import pytest

from camera_calibration_amd import engine as eng
from oracle import oracle as orc
from parity_record import check, check_equal

pytestmark = pytest.mark.gpu

# Tolerances of the solver-only rows, relative to |x|max.  The test systems are well conditioned by construction (dense part A A^T +
# n I with 768 random columns: condition number < 10, pose blocks M M^T + 6 I), so x carries a forward error of a few units of
# c * eps with c growing like sqrt(n) for these random sums.  Observed over rounds 3-5 (profiles/r0*_parity_deviations.json):
# engine vs the oracle's pivoted LDL^T 1.6e-16 ... 3.8e-16; two schedules / two back substitutions of the engine against each other
# 0 ... 3.1e-17 (most entries agree bit for bit, the largest |x| entries dominate the norm).  The bounds keep >= 10x headroom over
# the largest value seen AND stay above what a legitimate reordering of the sums can produce (a few ulp of |x|max): 64 eps against
# the oracle, 16 eps between schedules -- a real defect (a stale tile, a missed update) shows up at 1e-9 or worse.
EPS = float(np.finfo(np.float64).eps)
TOL_VS_ORACLE = 64 * EPS        # 1.4e-14
TOL_SCHEDULES = 16 * EPS        # 3.6e-15


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
def test_dataflow_factorisation_matches_the_oracle_for_every_tail_size(dense_dof):
    case = f"factorisation tail, D = {dense_dof}"
    s = _system(12, dense_dof, seed=dense_dof)
    xs = {}
    for rows in (512, 1024, 0):                    # 0 = default (8192): one launch at these sizes
        xs[rows] = eng.schur_solve(s.block_diag_H, s.off_diag_H, s.dense_H, s.block_diag_b, s.dense_b, factor_tail_rows=rows)
        check_equal(case, f"tail rows {rows}: finite", int(np.count_nonzero(~np.isfinite(xs[rows]))))
    x_ref = orc.schur_solve(s)                     # Eigen's pivoted LDLT restated (oracle)
    scale = np.abs(x_ref).max()
    for rows, x in xs.items():
        check(case, f"tail rows {rows}: x vs oracle / |x|max", np.abs(x - x_ref).max() / scale, TOL_VS_ORACLE)
    check(case, "super-panels + tail 512 vs one launch / |x|max", np.abs(xs[512] - xs[0]).max() / scale, TOL_SCHEDULES)


@pytest.mark.parametrize("dense_dof", [65, 1089, 3500, 7000])
def test_back_substitution_dataflow_launch_matches_the_panel_version(dense_dof):
    """k_back_dataflow (one launch, {value, tag} pairs, agent-scope polling) against the panel kernels on the same factor."""
    case = f"back substitution, D = {dense_dof}"
    s = _system(12, dense_dof, seed=1000 + dense_dof)
    x_panels = eng.schur_solve(s.block_diag_H, s.off_diag_H, s.dense_H, s.block_diag_b, s.dense_b, back_substitution_panels=True)
    x_flow = eng.schur_solve(s.block_diag_H, s.off_diag_H, s.dense_H, s.block_diag_b, s.dense_b)
    check_equal(case, "finite", int(np.count_nonzero(~np.isfinite(x_flow)) + np.count_nonzero(~np.isfinite(x_panels))))
    check(case, "dataflow vs panels / |x|max", np.abs(x_flow - x_panels).max() / np.abs(x_panels).max(), TOL_SCHEDULES)


@pytest.mark.parametrize("dense_dof,poison,tail_rows", [(700, "nan", 0), (3500, "nan", 1024), (3500, "zero", 1024), (3500, "zero", 0)])
def test_poisoned_diagonal_is_reported_not_hung(dense_dof, poison, tail_rows):
    """Error path of the dataflow launches: a NaN on the diagonal, or a row / column of exact zeros (a zero pivot), in the middle of
    the reduced system.  The chain flags the pivot (status 2), every flag of the launch is still published -- k_ldlt_tail and
    k_back_dataflow run to their end on NaNs instead of waiting for tiles that never come -- and the call returns
    CBA_ERR_NUMERIC in well under the 3-s spin limit; the LM loop treats that like the reference's NaN update (lambda x 2,
    LV/lm_optimizer.h:905-958)."""
    import time
    case = f"error path, D = {dense_dof}, {poison}, tail rows {tail_rows}"
    s = _system(12, dense_dof, seed=77 + dense_dof)
    k = dense_dof // 2 + 3
    if poison == "nan":
        s.dense_H[k, k] = np.nan
    else:
        s.dense_H[k, :] = 0.0
        s.dense_H[:, k] = 0.0
        s.off_diag_H[:, k] = 0.0
    t0 = time.perf_counter()
    with pytest.raises(eng.EngineError) as ei:
        eng.schur_solve(s.block_diag_H, s.off_diag_H, s.dense_H, s.block_diag_b, s.dense_b, factor_tail_rows=tail_rows)
    dt = time.perf_counter() - t0
    check_equal(case, "error code is CBA_ERR_NUMERIC (-4)", int("code -4" not in str(ei.value)))
    check(case, "seconds until the error is returned (bound: under the 3-s spin limit of one wait)", dt, 2.5)
    # the engine is usable afterwards: the same system without the poison solves
    s2 = _system(12, dense_dof, seed=77 + dense_dof)
    x = eng.schur_solve(s2.block_diag_H, s2.off_diag_H, s2.dense_H, s2.block_diag_b, s2.dense_b, factor_tail_rows=tail_rows)
    check_equal(case, "next solve finite", int(np.count_nonzero(~np.isfinite(x))))


@pytest.mark.parametrize("poison", ["zero", "nan"])
def test_singular_pose_block_is_reported(poison):
    """A 6 x 6 block D_i that cannot be inverted (all zeros: the first pivot is exactly zero; or a NaN on its diagonal): the
    register-resident block inverse flags it, the call returns CBA_ERR_NUMERIC (the LM loop doubles lambda, lm_optimizer.h:905-913)
    and the engine is usable afterwards."""
    case = f"error path, singular pose block ({poison})"
    s = _system(12, 700, seed=99)
    if poison == "zero":
        s.block_diag_H[5][:] = 0.0
    else:
        s.block_diag_H[5][2, 2] = np.nan
    with pytest.raises(eng.EngineError) as ei:
        eng.schur_solve(s.block_diag_H, s.off_diag_H, s.dense_H, s.block_diag_b, s.dense_b)
    check_equal(case, "error code is CBA_ERR_NUMERIC (-4)", int("code -4" not in str(ei.value)))
    s2 = _system(12, 700, seed=99)
    x = eng.schur_solve(s2.block_diag_H, s2.off_diag_H, s2.dense_H, s2.block_diag_b, s2.dense_b)
    check_equal(case, "next solve finite", int(np.count_nonzero(~np.isfinite(x))))


@pytest.mark.parametrize("dense_dof", [1089, 3500])
def test_solve_is_bit_identical_run_to_run(dense_dof):
    """The dataflow launches hand tiles over through device-scope flags (agent-scope stores / loads, LDS-DMA reads): a stale or early
    read of another workgroup's tile would make the same system give different bits from run to run.  (tools/gpu_det_check.py does
    the same with 30-40 repeats up to D = 22 617: profiles/r04_solve_determinism.txt.)"""
    case = f"run-to-run determinism of the reduced solve, D = {dense_dof}"
    s = _system(12, dense_dof, seed=4242 + dense_dof)
    xs = [eng.schur_solve(s.block_diag_H, s.off_diag_H, s.dense_H, s.block_diag_b, s.dense_b, factor_tail_rows=rows)
          for rows in (0, 0, 0, 512, 512, 512)]
    check_equal(case, "default schedule: entries of x that differ between three runs",
                int(np.count_nonzero(xs[0] != xs[1]) + np.count_nonzero(xs[0] != xs[2])))
    check_equal(case, "super-panels + 512-row final launch: entries of x that differ between three runs",
                int(np.count_nonzero(xs[3] != xs[4]) + np.count_nonzero(xs[3] != xs[5])))


def test_follow_up_list_of_the_finite_difference_kernel_does_not_overflow():
    from camera_calibration_amd import synthetic as syn
    pb, st, _ = syn.baseline_config(4, lambda cam, grid, pts: eng.project(cam, grid, pts), n_imagesets=30)
    e = eng.Engine(pb, last_projection=pb.obs_xy.astype(np.float64))
    e.set_state(st)
    e.debug_accumulate()
    check_equal("FD follow-up list", "tasks that found the list full", e.fd_redo_overflow())
    e.close()
