"""GPU parity tests: the HIP engine (through the C-ABI, include/cba.h) against the CPU oracle on the
same seeded inputs.  Tolerances: validity masks / indices bit-exact, pixels
<= 1e-9 px, cost rel <= 1e-12 per pass (we allow 1e-10 for atomically accumulated sums), update vector
x rel <= 1e-8.
"""
import numpy as np
import pytest

from camera_calibration_amd import engine as eng
from camera_calibration_amd import synthetic as syn
from camera_calibration_amd.problem import CENTRAL_GENERIC, NONCENTRAL_GENERIC, Camera, Problem, State
from camera_calibration_amd.se3 import se3_exp, se3_identity, se3_mul
from oracle import oracle as orc
from parity_record import check, check_equal

pytestmark = pytest.mark.gpu


def oracle_project(cam, grid, pts):
    return orc.project(cam, grid, pts)


def _xy1_grid(w, h):
    gy, gx = np.meshgrid(np.arange(float(h)), np.arange(float(w)), indexing="ij")
    g = np.stack([gx, gy, np.ones_like(gx)], -1).reshape(-1, 3)
    return g / np.linalg.norm(g, axis=1, keepdims=True)


# ---------------------------------------------------------------------------------------------------
# model level
# ---------------------------------------------------------------------------------------------------
def test_schur_complement2_golden_vector_on_gpu():
    # libvis/src/libvis/test/lm_optimizer.cc:470-557 through cba_schur_solve
    nan = float("nan")
    bD = np.array([[[1, 5], [nan, 6]], [[9, 5], [nan, 4]]], dtype=float)
    oH = np.array([[3, 4], [7, 8], [7, 6], [3, 2]], dtype=float)
    dH = np.array([[1, 4], [nan, 7]], dtype=float)
    x = eng.schur_solve(bD, oH, dH, np.array([1.0, 2, 3, 4]), np.array([5.0, 6]))
    np.testing.assert_allclose(x, [73.667, 171.667, 189.667, -294.333, 465.667, -582.0], atol=0.3)
    H = np.array([[1, 5, 0, 0, 3, 4], [5, 6, 0, 0, 7, 8], [0, 0, 9, 5, 7, 6], [0, 0, 5, 4, 3, 2],
                  [3, 7, 7, 3, 1, 4], [4, 8, 6, 2, 4, 7]], dtype=float)
    np.testing.assert_allclose(x, np.linalg.solve(H, np.arange(1.0, 7.0)), rtol=1e-9)


@pytest.mark.parametrize("bs,nb,dd", [(6, 7, 300), (3, 11, 200), (6, 40, 1100)])
def test_schur_solve_matches_oracle_random_spd(bs, nb, dd):
    rng = np.random.default_rng(bs * 100 + nb)
    n = bs * nb + dd
    J = rng.normal(size=(2 * n, n))
    H = J.T @ J / n + 1e-3 * np.eye(n)
    # zero the off-block parts of the block-diagonal region to give it the Schur structure
    for b in range(nb):
        for b2 in range(nb):
            if b != b2:
                H[b * bs:(b + 1) * bs, b2 * bs:(b2 + 1) * bs] = 0
    b = rng.normal(size=n)
    s = orc.System(bs, nb, dd)
    for k in range(nb):
        s.block_diag_H[k] = np.triu(H[k * bs:(k + 1) * bs, k * bs:(k + 1) * bs])
    s.off_diag_H[:] = H[:bs * nb, bs * nb:]
    s.dense_H[:] = np.triu(H[bs * nb:, bs * nb:])
    s.block_diag_b[:] = b[:bs * nb]
    s.dense_b[:] = b[bs * nb:]
    x_ref = orc.schur_solve(s)
    x = eng.schur_solve(s.block_diag_H, s.off_diag_H, s.dense_H, s.block_diag_b, s.dense_b)
    np.testing.assert_allclose(x, x_ref, rtol=1e-8, atol=1e-10 * np.abs(x_ref).max())
    np.testing.assert_allclose(H @ x, b, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("model", [CENTRAL_GENERIC, NONCENTRAL_GENERIC])
def test_unproject_and_project_match_oracle(model):
    cam = Camera(model, 640, 480, 10, 20, 640 - 5, 480 - 8, 8, 6)
    g = _xy1_grid(8, 6)
    rng = np.random.default_rng(0)
    if model == NONCENTRAL_GENERIC:
        g = np.stack([g, 0.01 * rng.uniform(-1, 1, size=g.shape)])
    px = np.array([10.0, 20.0]) + rng.uniform(0, 1, (400, 2)) * np.array([640 - 14, 480 - 27])
    l_ref, ok_ref = orc.unproject(cam, g, px)
    l_gpu, ok_gpu = eng.unproject(cam, g, px)
    assert np.array_equal(ok_ref, ok_gpu)
    np.testing.assert_allclose(l_gpu, l_ref, atol=1e-13)
    lj_ref, J_ref, okj_ref = orc.unproject(cam, g, px, with_jacobian=True)
    lj_gpu, J_gpu, okj_gpu = eng.unproject(cam, g, px, with_jacobian=True)
    assert np.array_equal(okj_ref, okj_gpu)
    np.testing.assert_allclose(lj_gpu, lj_ref, atol=1e-13 if model == CENTRAL_GENERIC else 1e-12)
    np.testing.assert_allclose(J_gpu, J_ref, atol=1e-13 + 1e-11 * np.abs(J_ref).max())
    # projection of points along the unprojected lines
    depth = rng.uniform(1.0, 5.0, size=(400, 1))
    pts = l_ref[:, 3:] + depth * l_ref[:, :3]
    p_ref, okp_ref = orc.project(cam, g, pts)
    p_gpu, okp_gpu = eng.project(cam, g, pts)
    assert np.array_equal(okp_ref, okp_gpu)
    assert okp_ref.all()
    np.testing.assert_allclose(p_gpu, p_ref, atol=1e-9)
    np.testing.assert_allclose(p_gpu, px, atol=1e-4)
    # out-of-rectangle pixels and unreachable points are flagged identically
    bad_px = np.array([[5.0, 100.0], [636.0, 100.0], [100.0, 473.0], [100.0, 100.0]])
    assert np.array_equal(orc.unproject(cam, g, bad_px)[1], eng.unproject(cam, g, bad_px)[1])
    far = np.array([[-50.0, 0.3, 1.0], [0.2, -40.0, 1.0], [3.0, 2.0, 1.0]])
    assert np.array_equal(orc.project(cam, g, far)[1], eng.project(cam, g, far)[1])


def test_empty_inputs():
    cam = Camera(CENTRAL_GENERIC, 64, 48, 0, 0, 63, 47, 5, 5)
    g = _xy1_grid(5, 5)
    px, ok = eng.project(cam, g, np.zeros((0, 3)))
    assert px.shape == (0, 2) and ok.shape == (0,)
    l, ok = eng.unproject(cam, g, np.zeros((0, 2)))
    assert l.shape == (0, 6)


# ---------------------------------------------------------------------------------------------------
# problem level
# ---------------------------------------------------------------------------------------------------
def _records_to_arrays(recs, n, Kg):
    valid = np.array([r.valid for r in recs], dtype=np.uint8)
    hasj = np.array([r.has_jacobian for r in recs], dtype=np.uint8)
    pix = np.array([[r.pixel[0], r.pixel[1]] for r in recs])
    J = np.zeros((n, 33 + 2 * Kg))
    for i, r in enumerate(recs):
        if not r.has_jacobian:
            continue
        J[i, 0:2] = r.residual[:]
        J[i, 2] = r.weight
        J[i, 3:15] = r.pose_jac[:]
        J[i, 15:27] = r.rig_jac[:]
        J[i, 27:33] = r.point_jac[:]
        J[i, 33:33 + 2 * Kg] = r.grid_jac[:2 * Kg]
    return valid, hasj, pix, J


def _compare_pass(pb, st, case):
    """Jacobian pass + accumulation + solve + update on identical state: engine vs oracle.  Tolerances ~10x the maxima
    observed on MI355X (profiles/r02_parity_deviations.json)."""
    op = orc.OracleProblem(pb)
    sysm = op.new_system()
    cost_ref, vec_ref, recs = op.jacobian_pass(st, sysm, want_records=True)
    e = eng.Engine(pb)
    e.set_state(st)
    cost = e.debug_accumulate()
    vec = e.dump(eng.DUMP_COST_VECTOR)
    flags = e.dump(eng.DUMP_FLAGS)
    Kg = 0 if pb.localize_only else max(c.params_per_grid_point for c in pb.cameras) * 16
    valid, hasj, pix_ref, J_ref = _records_to_arrays(recs, pb.n_obs, Kg)
    # masks bit-exact
    check_equal(case, "valid mask", int(np.count_nonzero((flags & 1) != valid)))
    check_equal(case, "has-jacobian mask", int(np.count_nonzero(((flags >> 1) & 1) != hasj)))
    check_equal(case, "cost-vector sign mask", int(np.count_nonzero((vec >= 0) != (vec_ref >= 0))))
    m = valid.astype(bool)
    pix = e.dump(eng.DUMP_PIXELS)
    check(case, "pixels abs [px]", np.abs(pix[m] - pix_ref[m]).max(), 2e-10)
    check(case, "cost vector rel", (np.abs(vec[m] - vec_ref[m]) / np.maximum(1e-3, vec_ref[m])).max(), 5e-10)
    check(case, "total cost rel", abs(cost - cost_ref) / max(1.0, abs(cost_ref)), 2e-14)
    # warm-start cache written back
    check(case, "last_projection abs [px]", np.abs(e.get_last_projection()[m] - op.last_projection[m]).max(), 2e-10)
    # per-observation Jacobians
    J = e.dump(eng.DUMP_JACOBIANS)
    hj = hasj.astype(bool)
    scale = np.abs(J_ref[hj]).max()
    check(case, "J records / max", np.abs(J[hj][:, :33 + 2 * Kg] - J_ref[hj]).max() / scale, 5e-9)
    # normal equations
    bD = e.dump(eng.DUMP_BLOCK_DIAG_H)
    for name, a, b in [("block_diag_H", bD, sysm.block_diag_H),
                       ("block_diag_b", e.dump(eng.DUMP_BLOCK_DIAG_B), sysm.block_diag_b),
                       ("off_diag_H", e.dump(eng.DUMP_OFF_DIAG_H), sysm.off_diag_H),
                       ("dense_H", np.triu(e.dump(eng.DUMP_DENSE_H)), np.triu(sysm.dense_H)),
                       ("dense_b", e.dump(eng.DUMP_DENSE_B), sysm.dense_b)]:
        if name == "block_diag_H":
            a = np.array([np.triu(x) for x in a]); b = np.array([np.triu(x) for x in b])
        check(case, name + " / max", np.abs(a - b).max() / np.abs(b).max(), 5e-9)
    # solve the *oracle's* system on the GPU solver and the engine's own system end-to-end
    lam = 1e-5 * (np.trace(sysm.dense_H) + sum(np.trace(b) for b in sysm.block_diag_H)) / pb.total_dof
    s2 = orc.System(sysm.block_size, sysm.n_blocks, sysm.dense_dof)
    for fld in ("block_diag_H", "off_diag_H", "dense_H", "block_diag_b", "dense_b"):
        getattr(s2, fld)[...] = getattr(sysm, fld)
    s2.add_lambda(lam)
    x_ref = orc.schur_solve(s2)
    x_gpu_solver = eng.schur_solve(s2.block_diag_H, s2.off_diag_H, s2.dense_H, s2.block_diag_b, s2.dense_b)
    check(case, "x engine solver vs oracle solver, oracle system / |x|max", np.abs(x_gpu_solver - x_ref).max() / np.abs(x_ref).max(), 5e-9)
    x = e.debug_solve(lam)
    check(case, "x engine (own system) vs oracle / |x|max", np.abs(x - x_ref).max() / np.abs(x_ref).max(), 1e-6)
    # state update on the same x
    st_ref = op.apply_update(st, x_ref)
    e.debug_apply_update(x_ref)
    st_gpu = e.get_state(st)
    check(case, "updated points abs", np.abs(st_gpu.points - st_ref.points).max(), 1e-15)
    # quaternion update goes through an fp32 sine/cosine (reference quirk): 1 ulp(fp32) of the update size
    check(case, "updated poses abs (bound: 1 fp32 ulp of the update's sin / cos)", np.abs(st_gpu.rig_tr_global - st_ref.rig_tr_global).max(),
          2e-7 * max(1e-3, np.abs(x_ref[:6 * pb.n_images]).max()) + 1e-15)
    check(case, "updated camera_tr_rig abs (bound: 1 fp32 ulp of the update's sin / cos)", np.abs(st_gpu.camera_tr_rig - st_ref.camera_tr_rig).max(), 1e-8)
    for g_gpu, g_ref in zip(st_gpu.grids, st_ref.grids):
        check(case, "updated grids abs", np.abs(g_gpu - g_ref).max(), 3e-15)
    e.close()


@pytest.mark.parametrize("num_cameras", [1, 2])
def test_first_iteration_parity_reference_fixture(num_cameras):
    pb, st, gt = syn.reference_test_problem(num_cameras, oracle_project, seed=0)
    _compare_pass(pb, st, f"gtest fixture, {num_cameras} camera(s)")


def test_first_iteration_parity_noncentral():
    pb, st, gt = syn.reference_test_problem(1, oracle_project, seed=3, num_points=60, num_poses=25,
                                            model_type=NONCENTRAL_GENERIC)
    pb.fd_delta = 1e-3
    _compare_pass(pb, st, "gtest fixture, non-central")


def test_first_iteration_parity_eliminate_points():
    pb, st, gt = syn.reference_test_problem(1, oracle_project, seed=5, num_points=70, num_poses=30)
    pb.eliminate_points = True
    _compare_pass(pb, st, "gtest fixture, eliminate_points")


@pytest.mark.parametrize("num_cameras", [1, 2])
def test_first_iteration_parity_localize_only(num_cameras):
    """localize_only = true (OptimizeJointly argument; intrinsics held fixed, joint_optimization.cc:451-593):
    no intrinsics block in the state, the dense part is points (+ rig poses)."""
    pb, st, gt = syn.reference_test_problem(num_cameras, oracle_project, seed=6, num_points=60, num_poses=20)
    pb.localize_only = True
    _compare_pass(pb, st, f"gtest fixture, localize_only, {num_cameras} camera(s)")


def test_localize_only_trajectory_matches_oracle():
    pb, st0, gt = syn.reference_test_problem(1, oracle_project, seed=8, num_points=50, num_poses=12)
    pb.localize_only = True
    op = orc.OracleProblem(pb)
    st_ref = st0.copy()
    e = eng.Engine(pb)
    e.set_state(st0)
    lam_ref, lam = -1.0, -1.0
    for it in range(4):
        r = op.optimize_jointly(st_ref, 1, lam_ref); lam_ref = r["final_lambda"]
        rep = e.step(lam); lam = rep.final_lambda
        assert rep.accepted == int(r["performed"])
        assert abs(rep.final_cost - r["cost"]) <= 1e-5 * max(1.0, abs(r["cost"]))
    st = e.get_state(st0)
    for g, g0 in zip(st.grids, st0.grids):
        np.testing.assert_array_equal(g, g0)          # the intrinsics are not touched
    np.testing.assert_allclose(st.points, st_ref.points, atol=1e-5)
    e.close()


def test_cost_pass_matches_oracle_and_invalid_residuals():
    pb, st, gt = syn.reference_test_problem(1, oracle_project, seed=1, num_points=80, num_poses=30)
    # push some points behind / far outside so that projections fail
    st.points[::7] *= 30.0
    op = orc.OracleProblem(pb)
    c_ref, v_ref = op.cost_pass(st)
    e = eng.Engine(pb)
    e.set_state(st)
    c, nv, v = e.cost(want_vector=True)
    assert np.array_equal(v >= 0, v_ref >= 0)
    assert (v_ref < 0).any()
    assert nv == int((v_ref >= 0).sum())
    m = v_ref >= 0
    np.testing.assert_allclose(v[m], v_ref[m], rtol=1e-9, atol=1e-9)
    assert abs(c - c_ref) <= 1e-9 * abs(c_ref)
    e.close()


@pytest.mark.parametrize("init_lambda", [-1.0, 0.5])
def test_zero_cost_exit_leaves_the_state_alone(init_lambda):
    """lm_optimizer.h:755-760: a pass whose cost is exactly zero (here: no observation projects -- every point sits far behind the
    cameras, so there is no valid residual) returns before any LM attempt.  With a given lambda the engine reads the pass's scalars
    behind its first solve (one host wait fewer per attempt) and has to discard that solve: same report, same state, and the next
    step on a healthy state works."""
    pb, st0, gt = syn.reference_test_problem(1, oracle_project, seed=3, num_points=60, num_poses=20)
    bad = st0.copy()
    bad.points[:] = bad.points * 0.0 + np.array([0.0, 0.0, -50.0])      # behind every camera
    op = orc.OracleProblem(pb)
    c_ref, v_ref = op.cost_pass(bad)
    assert c_ref == 0.0 and (v_ref < 0).all()
    e = eng.Engine(pb)
    e.set_state(bad)
    rep = e.step(init_lambda)
    assert rep.initial_cost == 0.0 and rep.final_cost == 0.0
    assert rep.lm_attempts == 0 and not rep.accepted and rep.n_residuals_valid == 0
    after = e.get_state(bad)
    assert np.array_equal(after.points, bad.points) and np.array_equal(after.rig_tr_global, bad.rig_tr_global)
    for a, b in zip(after.grids, bad.grids):
        assert np.array_equal(a, b)
    e.set_state(st0)                               # the engine is fine afterwards
    rep2 = e.step(-1.0)
    assert rep2.lm_attempts >= 1 and rep2.initial_cost > 0 and np.isfinite(rep2.final_cost)
    e.close()


@pytest.mark.parametrize("num_cameras", [1, 2])
def test_optimize_jointly_trajectory_matches_oracle(num_cameras):
    """Restated TestOptimizeJointly (APP/test/util.h:275-571): same iterates as the oracle, cost <= 1e-6*C."""
    pb, st0, gt = syn.reference_test_problem(num_cameras, oracle_project, seed=0)
    op = orc.OracleProblem(pb)
    st_ref = st0.copy()
    e = eng.Engine(pb)
    e.set_state(st0)
    lam_ref = lam = -1.0
    cost = np.inf
    for it in range(20 * num_cameras):
        r = op.optimize_jointly(st_ref, 1, lam_ref)
        rep = e.step(lam)
        lam_ref, lam, cost = r["final_lambda"], rep.final_lambda, rep.final_cost
        if it < 4:  # while the cost is far above the fp32-measurement floor the trajectories coincide
            assert rep.accepted == r["performed"]
            assert rep.lm_attempts == r["lm_attempts"]
            assert abs(rep.final_cost - r["cost"]) <= 1e-5 * abs(r["cost"]) + 1e-9
            assert abs(lam - lam_ref) <= 1e-6 * lam_ref
        if not rep.accepted:
            break
    assert cost <= num_cameras * 1e-6
    st_gpu = e.get_state(st0)
    # converged parameters agree with the oracle's (same gauge: both follow the same iterates)
    np.testing.assert_allclose(st_gpu.points, st_ref.points, atol=1e-5)
    np.testing.assert_allclose(st_gpu.rig_tr_global, st_ref.rig_tr_global, atol=1e-5)
    for a, b in zip(st_gpu.grids, st_ref.grids):
        np.testing.assert_allclose(a, b, atol=1e-5)
    e.close()


def test_python_optimize_jointly_mirror():
    pb, st0, gt = syn.reference_test_problem(1, oracle_project, seed=2, num_points=60, num_poses=30)
    cost, lam, performed, st, reports = eng.optimize_jointly(pb, st0, max_iteration_count=6)
    assert performed and len(reports) >= 1
    assert cost < reports[0].initial_cost * 1e-3
    assert lam > 0


def test_allreduce_callback_path_matches_plain_path():
    """The multi-GPU code path (reduced system built without lambda, all-reduce callback, lambda added
    after the reduction, scalar reductions) with an identity all-reduce must give the single-GPU result."""
    pb, st0, gt = syn.reference_test_problem(2, oracle_project, seed=6, num_points=60, num_poses=20)
    calls = []

    def identity_allreduce(ptr, count):
        calls.append(count)
        return 0

    e1 = eng.Engine(pb)
    e2 = eng.Engine(pb, allreduce=identity_allreduce, n_images_global=pb.n_images)
    e1.set_state(st0); e2.set_state(st0)
    lam1 = lam2 = -1.0
    for _ in range(3):
        r1 = e1.step(lam1); r2 = e2.step(lam2)
        lam1, lam2 = r1.final_lambda, r2.final_lambda
        assert r1.accepted == r2.accepted and r1.lm_attempts == r2.lm_attempts
        assert abs(r1.final_cost - r2.final_cost) <= 1e-6 * abs(r1.final_cost)
        assert abs(lam1 - lam2) <= 1e-9 * lam1
    assert max(calls) == eng.Engine.reduce_buffer_doubles(pb) and min(calls) in (1, 8)
    s1, s2 = e1.get_state(st0), e2.get_state(st0)
    np.testing.assert_allclose(s1.points, s2.points, atol=1e-7)
    e1.close(); e2.close()


def test_full_size_config2_properties():
    """BASELINE configs[1] at full size (500 imagesets, 84x60 grid, D = 12 525): size-independent checks.
    * the update solves the damped normal equations: both block rows of (H + lambda I) x = b have small residual;
    * an accepted step lowers the cost on the residuals valid in both passes; masks stay consistent."""
    proj = lambda cam, grid, pts: eng.project(cam, grid, pts)
    pb, st0, gt = syn.baseline_config(2, proj, n_imagesets=500)
    assert pb.dense_dof == 12525 and pb.n_obs > 300000
    e = eng.Engine(pb)
    e.set_state(st0)
    cost0 = e.debug_accumulate()
    flags = e.dump(eng.DUMP_FLAGS)
    assert (flags == 3).mean() > 0.99
    bD = e.dump(eng.DUMP_BLOCK_DIAG_H); bb = e.dump(eng.DUMP_BLOCK_DIAG_B)
    B = e.dump(eng.DUMP_OFF_DIAG_H); H = e.dump(eng.DUMP_DENSE_H); bd = e.dump(eng.DUMP_DENSE_B)
    lam = 1e-5 * (np.trace(H) + sum(np.trace(b) for b in bD)) / pb.total_dof
    x = e.debug_solve(lam)
    xb, xd = x[:pb.block_dof], x[pb.block_dof:]
    Hs = np.triu(H) + np.triu(H, 1).T
    r_dense = B.T @ xb + Hs @ xd + lam * xd - bd
    Ds = np.array([np.triu(b) + np.triu(b, 1).T for b in bD])
    r_block = np.einsum("nij,nj->ni", Ds, xb.reshape(-1, 6)).ravel() + lam * xb + B @ xd - bb
    assert np.abs(r_dense).max() <= 1e-7 * np.abs(bd).max()
    assert np.abs(r_block).max() <= 1e-7 * np.abs(bb).max()
    del H, Hs, B
    rep = e.step(-1.0)
    assert rep.accepted and rep.final_cost < 0.5 * rep.initial_cost
    assert abs(rep.initial_cost - cost0) <= 1e-9 * cost0
    assert rep.n_residuals_valid == int((flags & 1).sum())
    v_ref = e.dump(eng.DUMP_COST_VECTOR); v_new = e.dump(eng.DUMP_TEST_COST_VECTOR)
    both = (v_ref >= 0) & (v_new >= 0)
    assert v_new[both].sum() < v_ref[both].sum()
    e.close()


def test_baseline_config1_end_to_end_against_oracle():
    """BASELINE configs[0] (30 imagesets, 16x12 grid, 343 pattern points, D = 1 413): the reference's CPU-runnable
    case, end to end through the C-ABI against the oracle -- three OptimizeJointly(1) calls on the same start:
    accept decisions and attempt counts identical, costs to 1e-5 relative (finite-difference Jacobians of iterative
    projections), lambda to 1e-4."""
    pb, st0, gt = syn.baseline_config(1, oracle_project)
    assert pb.dense_dof == 1413 and pb.n_images == 30
    op = orc.OracleProblem(pb)
    st_ref = st0.copy()
    e = eng.Engine(pb)
    e.set_state(st0)
    lam_ref, lam = -1.0, -1.0
    for it in range(3):
        r = op.optimize_jointly(st_ref, 1, lam_ref); lam_ref = r["final_lambda"]
        rep = e.step(lam); lam = rep.final_lambda
        assert rep.accepted == int(r["performed"]) and rep.lm_attempts == r["lm_attempts"]
        assert abs(rep.final_cost - r["cost"]) <= 1e-5 * max(1.0, abs(r["cost"]))
        assert abs(lam - lam_ref) <= 1e-4 * lam_ref
    st = e.get_state(st0)
    np.testing.assert_allclose(st.points, st_ref.points, atol=1e-6)
    np.testing.assert_allclose(st.grids[0], st_ref.grids[0], atol=1e-6)
    e.close()

@pytest.mark.gpu
@pytest.mark.parametrize("dd", [63, 64, 65, 127, 255, 256, 257, 320, 511, 512, 513, 639, 767, 769, 1023, 1025, 1279, 1281,
                                1536, 1793, 2049, 2305, 2561])
def test_schur_solve_sizes_around_the_panel_boundaries(dd):
    """The blocked LDL^T switches code paths at multiples of 64 / 128 / 256 (last partial panel, width of the look-ahead,
    whole 256-panels on the chain stream): every size class against a dense solve."""
    bs, nb = 6, 5
    rng = np.random.default_rng(1000 + dd)
    n = bs * nb + dd
    J = rng.normal(size=(n + 50, n))
    H = J.T @ J / n + 1e-2 * np.eye(n)
    for b in range(nb):
        for b2 in range(nb):
            if b != b2:
                H[b * bs:(b + 1) * bs, b2 * bs:(b2 + 1) * bs] = 0
    rhs = rng.normal(size=n)
    bD = np.array([np.triu(H[k * bs:(k + 1) * bs, k * bs:(k + 1) * bs]) for k in range(nb)])
    x = eng.schur_solve(bD, H[:bs * nb, bs * nb:], np.triu(H[bs * nb:, bs * nb:]), rhs[:bs * nb], rhs[bs * nb:])
    x_ref = np.linalg.solve(H, rhs)
    np.testing.assert_allclose(x, x_ref, rtol=1e-7, atol=1e-9 * np.abs(x_ref).max())

