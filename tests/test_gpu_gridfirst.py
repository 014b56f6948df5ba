"""GPU parity of the GRID-FIRST elimination order (cba_solver_options.elimination = 2).

The reference eliminates the 6 x 6 pose blocks and factors the dense rest (LV/lm_optimizer.h:1247-1369,
APP/bundle_adjustment/joint_optimization.cc:794-804); the engine's second order eliminates the (banded) grid part first and factors
the border [rig | points | poses].  (H + lambda I) x = b has one solution, so the two orders must give the same
x up to rounding, the same LM decisions and the same converged calibration.  Checked here through the C-ABI:

  * x of one solve against LAPACK on the engine's own dumped system (third solver), against the pose-first order on the same
    system (deterministic accumulation: both engines see the same normal equations bit for bit), and against the oracle's
    orc_schur_solve on the oracle's system -- one / two / four strips, a rig, the non-central model, the cfg-2 grid;
  * LM trajectories: accept decisions and attempt counts identical to the oracle's OptimizeJointly, costs / lambda close;
  * the order's own failure paths: a zero pivot in the grid part (lambda = 0 with control points nothing observes) is
    CBA_ERR_NUMERIC like a zero pivot of the reduced system, combinations the order does not cover are CBA_ERR_UNSUPPORTED.
Tolerances: x 5e-9 of |x|max against the third solver (observed 3e-11 ... 6e-10), the `x` rows of test_gpu_parity.py otherwise.
"""
import numpy as np
import pytest

from camera_calibration_amd import engine as eng
from camera_calibration_amd import synthetic as syn
from oracle import oracle as orc
from parity_record import check, check_equal

pytestmark = pytest.mark.gpu


def oracle_project(cam, grid, pts):
    return orc.project(cam, grid, pts)


def _dense_system(e, pb):
    bs, nb, dd = pb.block_size, pb.n_blocks, pb.dense_dof
    n = nb * bs + dd
    H = np.zeros((n, n))
    bD = e.dump(eng.DUMP_BLOCK_DIAG_H)
    for i in range(nb):
        u = np.triu(bD[i])
        H[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs] = u + np.triu(u, 1).T
    off = e.dump(eng.DUMP_OFF_DIAG_H)
    H[:nb * bs, nb * bs:] = off
    H[nb * bs:, :nb * bs] = off.T
    D = np.triu(e.dump(eng.DUMP_DENSE_H))
    H[nb * bs:, nb * bs:] = D + np.triu(D, 1).T
    return H, np.concatenate([e.dump(eng.DUMP_BLOCK_DIAG_B), e.dump(eng.DUMP_DENSE_B)])


CASES = [
    # name, baseline config, imagesets, grid, lattice, strips
    ("central 24x18", 2, 12, (24, 18), (10, 13), (1, 2)),
    ("central 40x30, three / four strips", 2, 10, (40, 30), (10, 13), (3, 4)),
    ("tall grid 18x30 (strips cut y)", 2, 8, (18, 30), (10, 13), (2,)),
    ("rig 2 x 24x18", 3, 8, (24, 18), (10, 13), (1, 2)),
    ("non-central 20x16", 4, 8, (20, 16), (10, 13), (1, 2)),
]


@pytest.mark.parametrize("name,cfg,n_img,grid,lattice,strips", CASES)
def test_one_solve_matches_lapack_the_pose_first_order_and_the_oracle(name, cfg, n_img, grid, lattice, strips):
    pb, st, _ = syn.baseline_config(cfg, oracle_project, n_imagesets=n_img, grid_wh=grid, lattice_xy=lattice)
    case = "grid-first order, " + name
    e1 = eng.Engine(pb, deterministic=True, elimination=eng.ELIMINATION_POSE_FIRST)
    e1.set_state(st)
    e1.debug_accumulate()
    H, b = _dense_system(e1, pb)
    lam = 1e-5 * np.trace(H) / pb.total_dof
    x_lapack = np.linalg.solve(H + lam * np.eye(H.shape[0]), b)
    x_pose = e1.debug_solve(lam)
    e1.close()
    check(case, "x pose-first vs LAPACK, engine's system / |x|max", np.abs(x_pose - x_lapack).max() / np.abs(x_lapack).max(), 5e-9)
    # the oracle's own system and solver (orc_schur_solve, LV/lm_optimizer.h:1247-1369 restated)
    op = orc.OracleProblem(pb)
    sysm = op.new_system()
    op.jacobian_pass(st, sysm)
    sysm.add_lambda(lam)
    x_oracle = orc.schur_solve(sysm)
    for S in strips:
        e2 = eng.Engine(pb, deterministic=True, elimination=eng.ELIMINATION_GRID_FIRST, grid_strips=S)
        e2.set_state(st)
        e2.debug_accumulate()
        H2, b2 = _dense_system(e2, pb)
        check_equal(case, f"normal equations of the two engines differ (deterministic accumulation), {S} strip(s)",
                    int(np.count_nonzero(H2 != H)) + int(np.count_nonzero(b2 != b)))
        x = e2.debug_solve(lam)
        e2.close()
        check(case, f"x grid-first ({S} strip(s)) vs LAPACK, engine's system / |x|max", np.abs(x - x_lapack).max() / np.abs(x_lapack).max(), 5e-9)
        check(case, f"x grid-first ({S} strip(s)) vs pose-first / |x|max", np.abs(x - x_pose).max() / np.abs(x_pose).max(), 5e-9)
        check(case, f"x grid-first ({S} strip(s)) vs the oracle's solver on the oracle's system / |x|max",
              np.abs(x - x_oracle).max() / np.abs(x_oracle).max(), 1e-6)


@pytest.mark.parametrize("seed", range(10))
def test_random_geometries_give_the_x_of_the_pose_first_order(seed):
    """Sweep over what the static plan depends on -- model, number of cameras, grid size and aspect, imagesets, requested strips, task
    granularity -- and over lambda (1e-3 ... 1e-7 of the mean diagonal: condition numbers 6e4 ... 4e9): both orders on bit-identical
    normal equations (deterministic accumulation), x against a reference refined in extended precision (three steps of iterative
    refinement with long-double residuals on top of LAPACK).  The error any backward-stable solver may show is ~ cond(A) eps, so that
    is the scale of the bound: 0.5 cond eps (observed on MI355X: grid-first <= 0.05, pose-first <= 0.05, LAPACK itself <= 0.02 cond eps;
    at the ill-conditioned end that is 3.7e-8 / 3.4e-8 / 1.6e-8 of |x|max), and the new order may not be worse than the old one by more
    than 4 x.  (The CPU counterpart over the plan alone: tests/test_gridfirst_plan.py::test_random_geometries_...)"""
    rng = np.random.default_rng(1000 + seed)
    cfg = int(rng.choice([2, 3, 4]))
    big = 26 if cfg == 4 else 44
    grid = (int(rng.integers(10, big)), int(rng.integers(10, big * 3 // 4)))
    n_img = int(rng.integers(3, 14))
    strips = int(rng.integers(0, 5))
    single = bool(rng.integers(2))
    lam_rel = float(rng.choice([1e-3, 1e-5, 1e-7]))
    pb, st, _ = syn.baseline_config(cfg, oracle_project, n_imagesets=n_img, grid_wh=grid, lattice_xy=(10, 13))
    case = f"grid-first order, random geometry {seed}: cfg {cfg}, grid {grid[0]}x{grid[1]}, {n_img} imagesets, strips {strips}, single tiles {single}, lambda {lam_rel:g}"
    e1 = eng.Engine(pb, deterministic=True, elimination=eng.ELIMINATION_POSE_FIRST)
    e1.set_state(st)
    e1.debug_accumulate()
    H, b = _dense_system(e1, pb)
    lam = lam_rel * np.trace(H) / pb.total_dof
    A = H + lam * np.eye(H.shape[0])
    x_ref = np.linalg.solve(A, b)
    Al, bl = A.astype(np.longdouble), b.astype(np.longdouble)
    for _ in range(3):
        res = (bl - Al @ x_ref.astype(np.longdouble)).astype(np.float64)
        x_ref = (x_ref.astype(np.longdouble) + np.linalg.solve(A, res).astype(np.longdouble)).astype(np.float64)
    w = np.linalg.eigvalsh(A)
    scale = (w[-1] / w[0]) * np.finfo(np.float64).eps * np.abs(x_ref).max()
    x_pose = e1.debug_solve(lam)
    e1.close()
    e2 = eng.Engine(pb, deterministic=True, elimination=eng.ELIMINATION_GRID_FIRST, grid_strips=strips, grid_single_tile_tasks=single)
    check_equal(case, "the engine did not take the requested order", int(e2.elimination_order()["order"] != "grid-first"))
    e2.set_state(st)
    e2.debug_accumulate()
    x = e2.debug_solve(lam)
    x_again = e2.debug_solve(lam)                     # a second attempt on the same accumulated system: every tile is formed again
    e2.close()
    err, err_pose = np.abs(x - x_ref).max(), np.abs(x_pose - x_ref).max()
    print(case, f"cond {w[-1] / w[0]:.1e}: grid-first {err / np.abs(x_ref).max():.1e}, pose-first {err_pose / np.abs(x_ref).max():.1e} of |x|max")
    sweep = "grid-first order, random geometries"
    check(sweep, "x grid-first vs the refined solution / (cond eps |x|max), max over the sweep", err / scale, 0.5)
    check(sweep, "x pose-first vs the refined solution / (cond eps |x|max), max over the sweep", err_pose / scale, 0.5)
    check(sweep, "error of the grid-first order / error of the pose-first order, max over the sweep", err / (err_pose + 1e-12 * np.abs(x_ref).max()), 4.0)
    check_equal(case, "second solve of the same system differs (entries)", int(np.count_nonzero(x != x_again)))


@pytest.mark.parametrize("name,cfg,keep", [("left third of the image only", 2, "left"), ("one imageset keeps a single observation, one camera of the rig sees a corner", 3, "thin")])
def test_sparse_coverage_most_of_the_grid_unobserved(name, cfg, keep):
    """Observations that cover a corner of the image only: most control points are observed by nothing (their rows of H hold lambda
    alone), most grid x border tiles are inactive in the per-pass masks, whole strips of the grid do nothing -- the forming kernel, the
    block-sparse launch, the border update and the back substitution must all skip the same tiles.  x against LAPACK and against the
    pose-first order on the bit-identical system; the reference has no such case (its tests observe the whole image)."""
    from camera_calibration_amd.problem import Problem
    pb0, st, _ = syn.baseline_config(cfg, oracle_project, n_imagesets=10, grid_wh=(30, 22), lattice_xy=(10, 13))
    w = pb0.cameras[0].width
    xy = pb0.obs_xy
    if keep == "left":
        sel = xy[:, 0] < w / 3
    else:
        sel = (xy[:, 0] < w / 4) | (pb0.obs_camera == 0)
        first_of_img3 = np.nonzero(pb0.obs_image == 3)[0][:1]
        sel &= pb0.obs_image != 3
        sel[first_of_img3] = True
    pb = Problem(pb0.cameras, pb0.n_images, pb0.n_points, xy[sel], pb0.obs_point[sel], pb0.obs_image[sel], pb0.obs_camera[sel], fd_delta=pb0.fd_delta)
    assert 0 < pb.n_obs < 0.95 * pb0.n_obs
    case = "grid-first order, sparse coverage: " + name
    e1 = eng.Engine(pb, deterministic=True, elimination=eng.ELIMINATION_POSE_FIRST)
    e1.set_state(st)
    e1.debug_accumulate()
    H, b = _dense_system(e1, pb)
    # unknowns nothing observes: a zero ROW of H and of b (a zero diagonal alone is not enough: the fixed-point accumulation of the
    # deterministic mode rounds the tiny diagonal of a control point at the rim of a patch to zero while its couplings survive)
    empty = (np.abs(H).sum(axis=1) == 0.0) & (b == 0.0)
    unobserved = int(np.count_nonzero(empty))
    assert unobserved > 0.15 * pb.dense_dof, unobserved                      # the case is meant to leave a large part of the grid empty
    lam = 1e-4 * np.trace(H) / max(1, np.count_nonzero(np.diag(H)))
    x_lapack = np.linalg.solve(H + lam * np.eye(H.shape[0]), b)
    x_pose = e1.debug_solve(lam)
    e1.close()
    for S in (1, 2, 3):
        e2 = eng.Engine(pb, deterministic=True, elimination=eng.ELIMINATION_GRID_FIRST, grid_strips=S)
        e2.set_state(st)
        e2.debug_accumulate()
        x = e2.debug_solve(lam)
        x2 = e2.debug_solve(2.0 * lam)                                      # another attempt: every tile any launch wrote is formed again
        e2.close()
        check(case, f"x grid-first ({S} strip(s)) vs LAPACK / |x|max", np.abs(x - x_lapack).max() / np.abs(x_lapack).max(), 5e-9)
        check(case, f"x grid-first ({S} strip(s)) vs pose-first / |x|max", np.abs(x - x_pose).max() / np.abs(x_pose).max(), 5e-9)
        x2_ref = np.linalg.solve(H + 2.0 * lam * np.eye(H.shape[0]), b)
        check(case, f"x of the next attempt (2 lambda, {S} strip(s)) vs LAPACK / |x|max", np.abs(x2 - x2_ref).max() / np.abs(x2_ref).max(), 5e-9)
        check_equal(case, f"unobserved control points must get a zero update ({S} strip(s)), entries", int(np.count_nonzero(x[empty] != 0.0)))


@pytest.mark.parametrize("name,cfg,n_img,grid", [("central 24x18", 2, 12, (24, 18)), ("rig 2 x 20x16", 3, 6, (20, 16)), ("non-central 12x10", 4, 8, (12, 10))])
def test_lm_trajectory_matches_the_oracle(name, cfg, n_img, grid):
    """Five calls of OptimizeJointly(max_iteration_count = 1) (APP/calibration.cc:227-237): the engine in the grid-first order against
    the oracle -- decisions identical, costs and lambda close, and the pose-first engine takes the same decisions."""
    pb, st0, _ = syn.baseline_config(cfg, oracle_project, n_imagesets=n_img, grid_wh=grid, lattice_xy=(10, 13))
    op = orc.OracleProblem(pb)
    st_ref = st0.copy()
    e = eng.Engine(pb, elimination=eng.ELIMINATION_GRID_FIRST)
    ep = eng.Engine(pb, elimination=eng.ELIMINATION_POSE_FIRST)
    e.set_state(st0); ep.set_state(st0)
    lam_ref = lam = lam_p = -1.0
    case = "grid-first order, LM trajectory, " + name
    for it in range(5):
        r = op.optimize_jointly(st_ref, 1, lam_ref); lam_ref = r["final_lambda"]
        rep = e.step(lam); lam = rep.final_lambda
        rp = ep.step(lam_p); lam_p = rp.final_lambda
        check_equal(case, f"accept decision or attempt count differs from the oracle, call {it}",
                    int(rep.accepted != int(r["performed"])) + int(rep.lm_attempts != r["lm_attempts"]))
        check_equal(case, f"accept decision or attempt count differs from the pose-first engine, call {it}",
                    int(rep.accepted != rp.accepted) + int(rep.lm_attempts != rp.lm_attempts))
        check(case, "cost rel vs the oracle (max over the calls)", abs(rep.final_cost - r["cost"]) / max(1.0, abs(r["cost"])), 1e-6,
              note="five-call trajectory, default fp64-atomic accumulation")
        check(case, "lambda rel vs the oracle (max over the calls)", abs(lam - lam_ref) / abs(lam_ref), 1e-8 if cfg != 4 else 1e-6)
    st = e.get_state(st0)
    check(case, "points after five calls, abs", np.abs(st.points - st_ref.points).max(), 1e-6)
    e.close(); ep.close()


def test_cfg2_grid_with_60_imagesets_and_the_automatic_choice():
    """The headline grid (84 x 60, 10 080 grid unknowns, half-bandwidth 367) with 60 imagesets: the automatic choice is the grid-first
    order with two strips; x against the pose-first order and LAPACK on the same system."""
    pb, st, _ = syn.baseline_config(2, oracle_project, n_imagesets=60)
    case = "grid-first order, cfg-2 grid, 60 imagesets"
    pl = eng.gridfirst_plan(pb.cameras, pb.n_images, pb.n_points)
    assert pl["strips0"] == 2 and pl["half_bandwidth"] == 367 and pl["n_border"] == 6 * 60 + 3 * 815
    e1 = eng.Engine(pb, deterministic=True, elimination=eng.ELIMINATION_POSE_FIRST)
    e1.set_state(st); e1.debug_accumulate()
    H, b = _dense_system(e1, pb)
    lam = 1e-5 * np.trace(H) / pb.total_dof
    x_pose = e1.debug_solve(lam)
    e1.close()
    x_lapack = np.linalg.solve(H + lam * np.eye(H.shape[0]), b)
    e2 = eng.Engine(pb, deterministic=True)                      # automatic: grid-first here
    e2.set_state(st); e2.debug_accumulate()
    x = e2.debug_solve(lam)
    check(case, "x automatic order vs LAPACK / |x|max", np.abs(x - x_lapack).max() / np.abs(x_lapack).max(), 5e-9)
    check(case, "x automatic order vs pose-first / |x|max", np.abs(x - x_pose).max() / np.abs(x_pose).max(), 5e-9)
    # the same solve again and for another lambda: F is re-formed per solve, nothing is left over from the first factorisation
    x_again = e2.debug_solve(lam)
    check_equal(case, "second solve of the same system differs from the first (deterministic accumulation)", int(np.count_nonzero(x_again != x)))
    x4 = e2.debug_solve(4.0 * lam)
    x4_ref = np.linalg.solve(H + 4.0 * lam * np.eye(H.shape[0]), b)
    check(case, "x for 4 lambda vs LAPACK / |x|max", np.abs(x4 - x4_ref).max() / np.abs(x4_ref).max(), 5e-9)
    e2.close()


def test_zero_pivot_in_the_grid_part_is_a_numeric_error_and_the_lm_loop_recovers():
    """lambda = 0 with control points that no observation touches: their rows of the grid block are zero, the pivot chain of the
    grid meets a zero pivot -> CBA_ERR_NUMERIC (the LM loop doubles lambda, LV/lm_optimizer.h:905-913), not a hang, not a NaN state."""
    pb, st, _ = syn.baseline_config(2, oracle_project, n_imagesets=3, grid_wh=(24, 18), lattice_xy=(6, 7))
    e = eng.Engine(pb, elimination=eng.ELIMINATION_GRID_FIRST, grid_strips=2)
    e.set_state(st)
    e.debug_accumulate()
    H, _ = _dense_system(e, pb)
    assert (np.diag(H) == 0).any(), "the case needs unobserved control points"
    with pytest.raises(eng.EngineError) as ei:
        e.debug_solve(0.0)
    assert "-4" in str(ei.value)
    x = e.debug_solve(1e-3 * np.trace(H) / pb.total_dof)           # the problem object is still usable
    assert np.isfinite(x).all()
    rep = e.step(-1.0)
    assert np.isfinite(rep.final_cost)
    e.close()


def test_unsupported_combinations_are_refused():
    pb, st, _ = syn.baseline_config(2, oracle_project, n_imagesets=3, grid_wh=(24, 18), lattice_xy=(6, 7))
    pb.eliminate_points = True
    with pytest.raises(eng.EngineError) as ei:
        eng.Engine(pb, elimination=eng.ELIMINATION_GRID_FIRST)
    assert "-5" in str(ei.value)
    pb.eliminate_points = False
    pb.localize_only = True
    with pytest.raises(eng.EngineError):
        eng.Engine(pb, elimination=eng.ELIMINATION_GRID_FIRST)
    pb.localize_only = False
    e = eng.Engine(pb, elimination=eng.ELIMINATION_GRID_FIRST)      # fine
    e.close()
