"""Pins the restated CPU oracle (oracle/cba_oracle.c) against the REFERENCE'S OWN CODE.

Two layers, both CPU-only:

* `tests/golden/ref_vectors.npz` -- outputs of the reference's functions, produced by `oracle/_ref` (the reference
  sources compiled by oracle/Makefile against the Eigen / libvis stand-ins in oracle/ref_shim) on seeded inputs by
  tests/golden/make_ref_fixtures.py.  These tests always run, also where /root/reference does not exist.
* the live `oracle/_ref/libcalibref.so` and the reference's unmodified self-test binary, when they are present (they
  are built by __graft_entry__.build() in the container that has /root/reference and travel to the GPU box).

Rows of SURVEY 8(a) covered: A4 (ComputeJacobian / ComputeRigJacobian), M1-M4 and N1-N3 (B-spline surface, Unproject,
UnprojectWithJacobian, iterative Project of both generic models), P1-P3 (tangents, local updates, quaternion update),
B1 (Huber).  Observed maxima are printed with -s and recorded in profiles/r02_parity_deviations.json by
tools/record_parity.py; every tolerance below is <= 10x the observed maximum (or exact).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from camera_calibration_amd.problem import CENTRAL_GENERIC, NONCENTRAL_GENERIC, Camera
from oracle import oracle as orc
from oracle import ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
V = np.load(os.path.join(GOLDEN, "ref_vectors.npz"))
dp = orc._dp


def cam17():
    p = [int(v) for v in V["c17_params"]]
    return Camera(CENTRAL_GENERIC, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]), np.ascontiguousarray(V["c17_grid"])


def cam_n8():
    return Camera(NONCENTRAL_GENERIC, 640, 480, 0, 0, 639, 479, 8, 8), np.ascontiguousarray(V["n8_grid"])


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


# ------------------------------------------------------------------------------------------------------------------
# fixture layer (always runs)
# ------------------------------------------------------------------------------------------------------------------
def test_compute_jacobian_matches_reference_generated_code():
    L = orc.lib()
    worst = 0.0
    for i in range(V["jac_q"].shape[0]):
        out = np.zeros(30)
        L.orc_compute_jacobian(dp(V["jac_q"][i].copy()), dp(V["jac_p"][i].copy()), dp(out))
        worst = max(worst, rel(out, V["jac_30"][i]))
    print("ComputeJacobian rel", worst)
    assert worst <= 1e-14


def test_compute_rig_jacobian_matches_reference_generated_code():
    L = orc.lib()
    worst = 0.0
    # Column order of the generated code, established by running it: [rig_tr_global q4 t3 | camera_tr_rig q4 t3 | p3] --
    # columns 11..13 are the identity (d local / d camera_tr_rig translation).  That is how the caller reads the block
    # (joint_optimization.cc:405-425); the "... depending on camera_tr_rig quaternion" labels inside the generated
    # header (joint_optimization_jacobians.h:261-283) name the two poses the other way round and are not what the code does.
    for i in range(V["jac_q"].shape[0]):
        out = np.zeros(51)
        L.orc_compute_rig_jacobian(dp(V["jac_q"][i].copy()), dp(V["jac_p"][i].copy()), dp(V["jac_q2"][i].copy()),
                                   dp(V["jac_t"][i].copy()), dp(out))
        want = V["jac_51"][i].reshape(3, 17)
        assert np.array_equal(want[:, 11:14], np.eye(3))
        worst = max(worst, rel(out.reshape(3, 17), want))
    print("ComputeRigJacobian rel", worst)
    assert worst <= 2e-14


def test_central_unproject_matches_reference_on_the_real_17x13_camera():
    cam, grid = cam17()
    lines, jac, ok = orc.unproject(cam, grid, V["c17_px"], with_jacobian=True)
    assert ok.all()
    d_dir = np.abs(lines[:, :3] - V["c17_dirs"]).max()
    d_jac = rel(jac[:, :3, :], V["c17_jac"])
    print("central unproject |d dir|", d_dir, "jac rel", d_jac)
    assert d_dir <= 2e-15
    assert d_jac <= 2e-13
    plain, ok2 = orc.unproject(cam, grid, V["c17_px"])
    assert ok2.all()
    assert np.abs(plain[:, :3] - V["c17_dirs"]).max() <= 2e-14   # exact fractions vs 15-digit literals of the generated code


def test_central_project_matches_reference_on_the_real_17x13_camera():
    cam, grid = cam17()
    px, ok = orc.project(cam, grid, V["c17_pts"])
    assert np.array_equal(ok, V["c17_reproj_ok"].astype(bool))
    d = np.abs(px[ok] - V["c17_reproj"][ok]).max()
    print("central project |d px|", d)
    assert d <= 1e-9
    # the reference's own acceptance criterion (generic_models/src/main.cc:38-84): round trip within 1e-3 px
    assert np.linalg.norm(px[ok] - V["c17_px"][ok], axis=1).max() <= 1e-3
    px2, ok2 = orc.project(cam, grid, V["c17_pts"], init=V["c17_init"])
    assert np.array_equal(ok2, V["c17_reproj_init_ok"].astype(bool))
    assert np.abs(px2[ok2] - V["c17_reproj_init"][ok2]).max() <= 1e-9
    _, okb = orc.project(cam, grid, V["c17_bad_pts"])
    assert np.array_equal(okb, V["c17_bad_ok"].astype(bool))


def test_noncentral_unproject_and_project_match_reference():
    cam, grid = cam_n8()
    lines, jac, ok = orc.unproject(cam, grid, V["n8_px"], with_jacobian=True)
    assert ok.all()
    d_line = np.abs(lines - V["n8_lines"]).max()
    d_jac = rel(jac, V["n8_jac"])
    print("noncentral unproject |d line|", d_line, "jac rel", d_jac)
    assert d_line <= 1e-13     # incl. the fp32 square root AND fp32 division of the generated code (1 / sqrtf)
    assert d_jac <= 1e-12
    px, okp = orc.project(cam, grid, V["n8_pts"])
    assert np.array_equal(okp, V["n8_reproj_ok"].astype(bool))
    d = np.abs(px[okp] - V["n8_reproj"][okp]).max()
    print("noncentral project |d px|", d)
    assert d <= 1e-9
    assert np.linalg.norm(px[okp] - V["n8_px"][okp], axis=1).max() <= 1e-3


def test_generated_unprojection_patches_of_the_application_match():
    """APP/models/central_generic_jacobians.cc:320-448 and noncentral_generic_jacobians.cc:31-205 (the application's generated
    code, not the generic_models copy) on random 4 x 4 patches: a 4 x 4 grid has exactly one patch, and with a 100 x 100 px
    calibrated area frac = 3 + x / 100."""
    frac = V["patch_frac"]
    px = (frac - 3.0) * 100.0
    scale = float(np.float32(1.0) / np.float32(100.0))          # PixelScaleToGridScale, evaluated in fp32 by the reference
    cam = Camera(CENTRAL_GENERIC, 100, 100, 0, 0, 99, 99, 4, 4)
    ncam = Camera(NONCENTRAL_GENERIC, 100, 100, 0, 0, 99, 99, 4, 4)
    worst = dict(cdir=0.0, cjac=0.0, nline=0.0, njac=0.0)
    for i in range(frac.shape[0]):
        grid = np.ascontiguousarray(V["patch_central"][i])
        lines, jac, ok = orc.unproject(cam, grid, px[i:i + 1], with_jacobian=True)
        assert ok[0]
        worst["cdir"] = max(worst["cdir"], np.abs(lines[0, :3] - V["patch_central_dir"][i]).max())
        worst["cjac"] = max(worst["cjac"], rel(jac[0, :3], V["patch_central_jac"][i].reshape(3, 2) * scale))
        lp = V["patch_lines"][i]
        ngrid = np.ascontiguousarray(np.stack([lp[:, :3], lp[:, 3:]]))
        lines, jac, ok = orc.unproject(ncam, ngrid, px[i:i + 1], with_jacobian=True)
        assert ok[0]
        worst["nline"] = max(worst["nline"], np.abs(lines[0] - V["patch_line_out"][i]).max())
        worst["njac"] = max(worst["njac"], rel(jac[0], V["patch_line_jac"][i].reshape(6, 2) * scale))
    print("APP generated patches", worst)
    assert worst["cdir"] <= 1e-14 and worst["cjac"] <= 1e-12
    assert worst["nline"] <= 1e-13 and worst["njac"] <= 1e-12     # origins up to ~5 with cancelling cubic weights


def test_parametrisations_match_reference():
    L = orc.lib()
    k = V["par_dir"].shape[0]
    worst_t = worst_q = 0.0
    for i in range(k):
        t1, t2 = np.zeros(3), np.zeros(3)
        L.orc_tangents(dp(V["par_dir"][i].copy()), dp(t1), dp(t2))
        worst_t = max(worst_t, np.abs(t1 - V["par_t1"][i]).max(), np.abs(t2 - V["par_t2"][i]).max())
        q = np.zeros(4)
        L.orc_apply_quaternion_update(dp(V["jac_q"][i].copy()), dp(V["quat_update"][i].copy()), dp(q))
        want = V["quat_out"][i] / np.linalg.norm(V["quat_out"][i])     # Sophus normalises on construction (so3.hpp:536-541)
        worst_q = max(worst_q, np.abs(q - want).max())
    print("tangents", worst_t, "quaternion update", worst_q)
    assert worst_t <= 1e-15
    assert worst_q <= 1e-15


def test_huber_and_bspline_match_reference():
    L = orc.lib()
    for v, c, w in zip(V["huber_sq"], V["huber_cost_sq"], V["huber_weight_sq"]):
        assert L.orc_huber_cost_sq(v, 1.0) == c
        assert L.orc_huber_weight_sq(v, 1.0) == w
    net = np.ascontiguousarray(V["bsp_net"])
    worst = 0.0
    for x, f, s in zip(V["bsp_x"], V["bsp_fast"], V["bsp_slow"]):
        a, b = np.zeros(2), np.zeros(2)
        L.orc_bspline_surface(dp(net.ravel()), 4, 4, 2, x, 1.5, dp(a))
        L.orc_bspline_surface_slow(dp(net.ravel()), 4, 4, 2, x, 1.5, dp(b))
        worst = max(worst, np.abs(a - f).max(), np.abs(b - s).max())
    print("bspline", worst)
    assert worst <= 4e-13      # values up to 9; the cubic weights cancel from ~64
    # the reference test's own assertion on its fixed net (APP/test/b_spline_test.cc:41-58), on the reference's numbers
    assert np.abs(V["bsp_fast_f32"] - V["bsp_slow_f32"]).max() <= 1e-5


def test_f2_reader_parses_the_reference_yaml():
    """F2: the camera YAML the reference holds as a test vector (generic_models/src/main.cc:86-98) through the product's
    reader, compared with what the reference's own reader (CentralGenericCamera::Read) produced from the same file."""
    from camera_calibration_amd import calibration_io as cio
    cam, grid = cio.load_camera_model(os.path.join(GOLDEN, "ref_generic_models_17x13.yaml"))
    want_cam, want_grid = cam17()
    assert (cam.width, cam.height, cam.calib_min_x, cam.calib_min_y, cam.calib_max_x, cam.calib_max_y, cam.grid_w, cam.grid_h) == \
           (640, 480, 15, 16, 624, 464, 17, 13)
    assert cam == want_cam
    np.testing.assert_allclose(np.asarray(grid).reshape(-1, 3), want_grid, rtol=0, atol=5e-16)   # re-normalisation on load: one rounding
    assert abs(np.asarray(grid).reshape(-1, 3)[-1, 2] - 0.67986719656337) <= 1e-3      # main.cc:137


# ---- round 3: the accumulators and the grid model (rows B2, B3, A5, M5, M6) ----------------------------------------
ACC = np.load(os.path.join(GOLDEN, "ref_accumulated.npz"))
from tests.ref_modes import ACC_FIELDS, ACC_MODES, load_mode  # noqa: E402


def _upper(name, a):
    a = np.asarray(a)
    if name == "block_diag_H":
        return np.array([np.triu(b) for b in a])
    return np.triu(a) if name == "dense_H" else a


@pytest.mark.parametrize("mode", sorted(ACC_MODES))
def test_accumulation_matches_the_reference_accumulator(mode):
    """oracle accumulate() / add_H / add_b and the index order of add_reprojection_residual (oracle/cba_oracle.c:660-690, 880-910)
    against LV/lm_optimizer_update_accumulator.h + lm_optimizer_jtj_accumulator_base.h compiled from the reference
    (oracle/ref_lm.cc), every touched entry of block_diag_H / off_diag_H / dense_H / b."""
    pb, st = load_mode(mode, GOLDEN)
    op = orc.OracleProblem(pb)
    sysm = op.new_system()
    cost, vec, _ = op.jacobian_pass(st, sysm, want_records=True)
    assert abs(cost - float(ACC[f"{mode}__cost"])) <= 1e-14 * abs(cost)
    assert np.array_equal(vec >= 0, ACC[f"{mode}__cost_vector"] >= 0)                        # AddInvalidResidual pairing
    assert np.abs(vec - ACC[f"{mode}__cost_vector"]).max() <= 1e-13
    for f in ACC_FIELDS:
        ref_v = ACC[f"{mode}__{f}"]
        if f in ("dense_H", "block_diag_H"):       # the reference writes upper triangles only (update_accumulator.h:212, 256)
            low = np.tril(ref_v, -1) if f == "dense_H" else np.array([np.tril(b, -1) for b in ref_v])
            assert not np.any(low)
        worst = rel(_upper(f, getattr(sysm, f)), _upper(f, ref_v)) if np.abs(ref_v).max() > 0 else float(np.abs(getattr(sysm, f)).max())
        print(mode, f, worst)
        assert worst <= 5e-15, (mode, f, worst)
        # same sparsity: an entry the reference never touched stays exactly zero in the oracle
        assert np.array_equal(_upper(f, getattr(sysm, f)) != 0, _upper(f, ref_v) != 0), (mode, f)


def test_projection_jacobian_wrt_intrinsics_matches_reference_on_the_17x13_camera():
    cam, grid = cam17()
    worst = 0.0
    for i in range(V["m5_pts"].shape[0]):
        ok, idx, J = orc.projection_jacobian_wrt_intrinsics(cam, grid, V["m5_pts"][i], V["m5_px"][i], float(V["m5_delta"]))
        assert ok == int(V["m5_ok"][i])
        if ok == 1:
            assert np.array_equal(idx, V["m5_idx"][i])
            worst = max(worst, rel(J, V["m5_jac"][i]))
    print("ProjectionJacobianWrtIntrinsics rel", worst)
    assert worst <= 5e-10          # two iterative projections (8e-13 px apart) divided by delta = 1e-4


def test_projection_jacobian_wrt_intrinsics_matches_reference_at_every_observation():
    pb, st = load_mode("central", GOLDEN)
    op = orc.OracleProblem(pb)
    _, _, recs = op.jacobian_pass(st, op.new_system(), want_records=True)
    worst = 0.0
    for o in range(pb.n_obs):
        if not recs[o].valid:
            continue
        assert bool(recs[o].has_jacobian) == (int(ACC["central__m5_ok"][o]) == 1)       # (the three local-point projections succeed here)
        if recs[o].has_jacobian:
            assert np.array_equal(np.array(recs[o].grid_indices[:32]), ACC["central__m5_idx"][o])
            worst = max(worst, rel(np.array(recs[o].grid_jac[:64]).reshape(2, 32), ACC["central__m5_jac"][o]))
    print("per-observation grid Jacobian rel", worst)
    assert worst <= 5e-9           # observed 5.2e-10 (finite differences of two iterative projections)


def test_noncentral_projection_jacobian_wrt_intrinsics_matches_reference_at_every_observation():
    """N3: projection_jacobian_wrt_intrinsics (oracle/cba_oracle.c, non-central branch) against
    NoncentralGenericModel::ProjectionJacobianWrtIntrinsics (APP/models/noncentral_generic.h:224-283)."""
    pb, st = load_mode("noncentral", GOLDEN)
    op = orc.OracleProblem(pb)
    _, _, recs = op.jacobian_pass(st, op.new_system(), want_records=True)
    worst = 0.0
    n = 0
    for o in range(pb.n_obs):
        if not recs[o].valid:
            continue
        assert bool(recs[o].has_jacobian) == (int(ACC["noncentral__m5_ok"][o]) == 1)
        if recs[o].has_jacobian:
            assert np.array_equal(np.array(recs[o].grid_indices[:80]), ACC["noncentral__m5_idx"][o])
            worst = max(worst, rel(np.array(recs[o].grid_jac[:160]).reshape(2, 80), ACC["noncentral__m5_jac"][o]))
            n += 1
    print("non-central per-observation grid Jacobian rel", worst, "over", n)
    assert n > 100 and worst <= 1e-9      # observed 5.7e-11


def test_noncentral_subtract_delta_matches_reference():
    cam, grids = cam_n8()
    from camera_calibration_amd.problem import Problem, State
    pb = Problem([cam], 1, 1, np.zeros((0, 2), np.float32), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
    st = State(np.array([[1.0, 0, 0, 0, 0, 0, 0]]), np.array([[1.0, 0, 0, 0, 0, 0, 0]]), np.zeros((1, 3)), [grids])
    x = np.zeros(pb.total_dof)
    x[-V["n3_delta"].size:] = V["n3_delta"]
    st1 = orc.OracleProblem(pb).apply_update(st, x)
    assert np.abs(np.asarray(st1.grids[0]).reshape(2, -1, 3) - V["n3_grids"]).max() <= 1e-15


def test_subtract_delta_matches_reference():
    cam, grid = cam17()
    got = np.asarray(orc.fit_grid_apply_update(cam.grid_w, cam.grid_h, grid, V["m6_delta"])).reshape(-1, 3)
    assert np.abs(got - V["m6_grid"]).max() <= 1e-15
    # the same update through the hot path's state update (orc_apply_update)
    from camera_calibration_amd.problem import Problem, State
    pb = Problem([cam], 1, 1, np.zeros((0, 2), np.float32), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
    st = State(np.array([[1.0, 0, 0, 0, 0, 0, 0]]), np.array([[1.0, 0, 0, 0, 0, 0, 0]]), np.zeros((1, 3)), [grid])
    x = np.zeros(pb.total_dof)
    x[-V["m6_delta"].size:] = V["m6_delta"]
    st1 = orc.OracleProblem(pb).apply_update(st, x)
    assert np.abs(np.asarray(st1.grids[0]).reshape(-1, 3) - V["m6_grid"]).max() <= 1e-15


# ------------------------------------------------------------------------------------------------------------------
# live layer: oracle/_ref itself
# ------------------------------------------------------------------------------------------------------------------
needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
def test_reference_selftest_binary_passes():
    exe = os.path.join(os.path.dirname(ref.LIB_PATH), "generic_models_selftest")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    assert out.count("success") == 4 and "failure" not in out, out


@needs_ref
def test_fixture_is_what_the_live_reference_computes():
    cam, grid = cam17()
    rc = ref.RefCamera(cam, grid)
    for i in (0, 17, 399):
        ok, d, j = rc.unproject(V["c17_px"][i], jacobian=True)
        assert ok and np.array_equal(d, V["c17_dirs"][i]) and np.array_equal(j, V["c17_jac"][i])
        ok, p = rc.project(V["c17_pts"][i])
        assert ok and np.array_equal(p, V["c17_reproj"][i])


@needs_ref
def test_oracle_vs_live_reference_on_a_fine_grid():
    """A BASELINE-config-2-sized grid (84 x 60): the fixtures above use the reference's small cameras."""
    from camera_calibration_amd import synthetic as syn
    cam = Camera(CENTRAL_GENERIC, 2048, 1456, 0, 0, 2047, 1455, 84, 60)
    grid = syn.pinhole_direction_grid(cam, 0.8 * 1456, 0.8 * 1456, 1024.0, 728.0, k1=-0.12)
    rng = np.random.default_rng(5)
    grid = grid + 0.0005 * rng.normal(size=grid.shape)
    grid /= np.linalg.norm(grid, axis=1, keepdims=True)
    rc = ref.RefCamera(cam, grid)
    px = np.stack([rng.uniform(0, 2048, 300), rng.uniform(0, 1456, 300)], axis=1)
    lines, jac, ok = orc.unproject(cam, grid, px, with_jacobian=True)
    pts = lines[:, :3] * rng.uniform(0.3, 1.0, (300, 1))
    got, gok = orc.project(cam, grid, pts)
    worst_d = worst_j = worst_p = 0.0
    for i in range(300):
        o, d, j = rc.unproject(px[i], jacobian=True)
        assert o == ok[i]
        worst_d = max(worst_d, np.abs(d - lines[i, :3]).max())
        worst_j = max(worst_j, rel(jac[i, :3], j))
        o, p = rc.project(pts[i])
        assert o == gok[i]
        if o:
            worst_p = max(worst_p, np.abs(p - got[i]).max())
    print("84x60: dir", worst_d, "jac rel", worst_j, "project px", worst_p)
    assert worst_d <= 2e-15 and worst_j <= 1e-12 and worst_p <= 1e-9


@needs_ref
def test_accumulated_fixture_is_what_the_live_reference_computes():
    for mode in ("rig", "localize_eliminate"):
        pb, st = load_mode(mode, GOLDEN)
        op = orc.OracleProblem(pb)
        _, _, recs = op.jacobian_pass(st, op.new_system(), want_records=True)
        r = ref.accumulate_records(pb, recs)
        for f in ACC_FIELDS:
            assert np.array_equal(r[f], ACC[f"{mode}__{f}"]), (mode, f)
    cam, grid = cam17()
    ok, idx, J = ref.central_grid_projection_jacobian(cam, grid, V["m5_pts"][3], V["m5_px"][3], float(V["m5_delta"]))
    assert ok == int(V["m5_ok"][3]) and np.array_equal(idx, V["m5_idx"][3]) and np.array_equal(J, V["m5_jac"][3])
    assert np.array_equal(ref.central_grid_subtract_delta(cam, grid, V["m6_delta"]), V["m6_grid"])


# ------------------------------------------------------------------------------------------------------------------
# live layer, round 5: LV/lm_optimizer.h itself (oracle/ref_lmopt.cc -> _ref/libcalibref_lm.so): the LM loop, CostIsSmallerThan and
# SolveWithSchurComplementDenseOffDiag are the REFERENCE's code here; only Eigen's LDLT underneath stays the oracle's restatement
# ------------------------------------------------------------------------------------------------------------------
needs_ref_lm = pytest.mark.skipif(not ref.lm_available(), reason="oracle/_ref/libcalibref_lm.so not built (needs /root/reference)")


def _random_system(rng, bs, nb, dd, indefinite=False):
    s = orc.System(bs, nb, dd)
    A = rng.normal(size=(dd, dd + 3))
    H = A @ A.T + (0.0 if indefinite else dd) * np.eye(dd)
    if indefinite:
        H -= 0.5 * np.trace(H) / dd * np.eye(dd)
    s.dense_H[:] = np.triu(H)
    s.dense_H[np.tril_indices(dd, -1)] = np.nan            # the reference only ever reads the upper triangles (LV/test/lm_optimizer.cc:493-506)
    s.off_diag_H[:] = rng.normal(size=(bs * nb, dd)) * 0.3
    for b in range(nb):
        M = rng.normal(size=(bs, bs))
        s.block_diag_H[b] = np.triu(M @ M.T + bs * np.eye(bs))
        s.block_diag_H[b][np.tril_indices(bs, -1)] = np.nan
    s.block_diag_b[:] = rng.normal(size=bs * nb)
    s.dense_b[:] = rng.normal(size=dd)
    return s


@needs_ref_lm
def test_schur_solve_matches_the_reference_s_own_function():
    """SolveWithSchurComplementDenseOffDiag (LV/lm_optimizer.h:1247-1369) compiled from the reference, called through the friend
    class the reference declares for its own test: first the reference's golden vector (LMOptimizer.SchurComplement2,
    LV/test/lm_optimizer.cc:476-543), then random systems in both block sizes of the path against orc_schur_solve."""
    nan = float("nan")
    s = orc.System(2, 2, 2)
    s.block_diag_H[0] = [[1, 5], [nan, 6]]; s.block_diag_H[1] = [[9, 5], [nan, 4]]
    s.dense_H[:] = [[1, 4], [nan, 7]]
    s.off_diag_H[:] = [[3, 4], [7, 8], [7, 6], [3, 2]]
    s.block_diag_b[:] = [1, 2, 3, 4]; s.dense_b[:] = [5, 6]
    x_ref = ref.lmopt_schur_solve(s)
    assert np.abs(x_ref - np.array([73.667, 171.667, 189.667, -294.333, 465.667, -582.0])).max() <= 1e-3   # the reference's EXPECT_NEAR values
    assert rel(orc.schur_solve(s), x_ref) <= 2e-11     # an indefinite 6 x 6 system with |x| ~ 100 |b|: observed 1.3e-12
    rng = np.random.default_rng(11)
    worst = 0.0
    for bs, nb, dd, indef in ((6, 5, 40, False), (3, 17, 61, False), (6, 12, 150, False), (6, 4, 33, True)):
        s = _random_system(rng, bs, nb, dd, indef)
        worst = max(worst, rel(orc.schur_solve(s), ref.lmopt_schur_solve(s)))
    print("Schur solve, oracle vs reference code rel", worst)
    assert worst <= 1e-11          # different summation orders of B^T D^-1 B; observed 1e-14 ... 1e-13


@needs_ref_lm
def test_cost_is_smaller_than_matches_the_reference_s_own_function():
    """LMOptimizer::CostIsSmallerThan (LV/lm_optimizer.h:993-1011): pairs only residuals valid on BOTH sides; empty intersection
    -> false; equality -> false."""
    rng = np.random.default_rng(3)

    def restated(l, r):                       # what the oracle / the engine do (cba_oracle.c: cost_is_smaller_than; k_reduce_costs_*)
        m = (l >= 0) & (r >= 0)
        return bool(m.any() and l[m].sum() < r[m].sum())
    for _ in range(200):
        n = int(rng.integers(1, 40))
        l = rng.uniform(0, 2, n); r = rng.uniform(0, 2, n)
        l[rng.uniform(size=n) < 0.2] = -1.0
        r[rng.uniform(size=n) < 0.2] = -1.0
        assert ref.lmopt_cost_is_smaller_than(l, r) == restated(l, r)
    assert ref.lmopt_cost_is_smaller_than(np.array([-1.0, 0.5]), np.array([0.5, -1.0])) is False        # nothing valid on both sides
    assert ref.lmopt_cost_is_smaller_than(np.array([0.5, 0.25]), np.array([0.25, 0.5])) is False        # equal sums
    assert ref.lmopt_cost_is_smaller_than(np.array([0.5, 9.0]), np.array([0.75, -1.0])) is True         # the invalid pair does not count


def _lm_pair(pb, st, iterations, stop_rule=False):
    """orc_optimize_jointly against the reference's LMOptimizer::Optimize on the same problem, call by call."""
    opA, stA = orc.OracleProblem(pb), st.copy()
    opB, stB = orc.OracleProblem(pb), st.copy()
    lamA = lamB = -1.0
    last = float("inf")
    worst = dict(cost=0.0, lam=0.0, state=0.0)
    attempts = []
    for _ in range(iterations):
        a = opA.optimize_jointly(stA, 1, lamA); lamA = a["final_lambda"]
        b = ref.lmopt_optimize_jointly(opB, stB, 1, lamB); lamB = b["final_lambda"]
        t = b["trace"][0]
        # decisions: accepted or not, and how many LM attempts it took (every attempt that produced a finite update runs one
        # cost-only pass, lm_optimizer.h:918-926)
        assert a["performed"] == b["performed"] == bool(t[3])
        assert a["lm_attempts"] == int(t[4]), (a["lm_attempts"], t)
        assert int(t[5]) == 1
        attempts.append(a["lm_attempts"])
        worst["cost"] = max(worst["cost"], abs(a["cost"] - b["cost"]) / max(abs(b["cost"]), 1e-300))
        worst["lam"] = max(worst["lam"], abs(lamA - lamB) / lamB)
        worst["state"] = max(worst["state"], np.abs(stA.points - stB.points).max(), np.abs(stA.rig_tr_global - stB.rig_tr_global).max(),
                             max(np.abs(ga - gb).max() for ga, gb in zip(stA.grids, stB.grids)))
        if stop_rule and (not a["performed"] or a["cost"] >= last - 1e-4):
            break
        last = a["cost"]
    return worst, attempts


@needs_ref_lm
@pytest.mark.parametrize("case", ["1cam", "rig", "noncentral", "eliminate_points", "localize_only"])
def test_lm_loop_matches_the_reference_s_own_optimizer(case):
    """OptimizeImpl (LV/lm_optimizer.h:629-991) compiled from the reference and driven as OptimizeJointly drives it
    (joint_optimization.cc:797-812, :916-940) on the gtest-sized bundle-adjustment problems, against the oracle's restatement:
    same accept decisions and LM attempt counts, lambda identical to rounding, iterates equal up to what the gauge directions
    amplify (the two sides sum B^T D^-1 B in different orders)."""
    from camera_calibration_amd import synthetic as syn
    proj = lambda cam, grid, pts: orc.project(cam, grid, pts)
    if case == "noncentral":
        pb, st, _ = syn.noncentral_test_problem(proj) if hasattr(syn, "noncentral_test_problem") else syn.baseline_config(4, proj, n_imagesets=5, grid_wh=(8, 6))
    else:
        pb, st, _ = syn.reference_test_problem(2 if case == "rig" else 1, proj, seed=7, num_points=60, num_poses=20)
    if case == "eliminate_points":
        pb.eliminate_points = True
    if case == "localize_only":
        pb.localize_only = True
    worst, attempts = _lm_pair(pb, st, 5)
    print(case, worst, attempts)
    assert worst["lam"] <= 1e-12 and worst["cost"] <= 1e-5 and worst["state"] <= 1e-7


@needs_ref_lm
def test_lm_loop_with_rejected_updates_matches_the_reference_s_own_optimizer():
    """A noisy problem run to the reference's stopping rule (APP/calibration.cc:298): the late iterations reject updates and double
    lambda (lm_optimizer.h:959-977) -- the reject branch, the multi-attempt lambda trajectory and CostIsSmallerThan inside the
    loop, reference code against the oracle."""
    from camera_calibration_amd import synthetic as syn
    pb, st, _ = syn.baseline_config(1, lambda cam, grid, pts: orc.project(cam, grid, pts), n_imagesets=6, grid_wh=(8, 6))
    worst, attempts = _lm_pair(pb, st, 40, stop_rule=True)
    print(worst, attempts)
    assert max(attempts) >= 2, "the case is meant to contain rejected updates"
    assert worst["lam"] <= 1e-12 and worst["cost"] <= 1e-6 and worst["state"] <= 1e-5
