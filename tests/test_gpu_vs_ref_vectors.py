"""HIP kernels against numbers computed by the REFERENCE'S OWN CODE (tests/golden/ref_vectors.npz, produced by oracle/_ref =
the reference sources compiled by oracle/Makefile; see tests/golden/make_ref_fixtures.py).  No oracle in between:
CameraModel::Unproject / UnprojectWithJacobian / Project of both generic models through cba_unproject / cba_project on
the real calibrated 17 x 13 camera the reference ships as a test vector (generic_models/src/main.cc:86-98) and on the
non-central 8 x 8 camera of its self-test (main.cc:146-160), incl. the reference's own acceptance criterion (round trip
within 1e-3 px, main.cc:38-84)."""
import os

import numpy as np
import pytest

from camera_calibration_amd import calibration_io as cio
from camera_calibration_amd import engine as eng
from camera_calibration_amd.problem import NONCENTRAL_GENERIC, Camera
from parity_record import check, check_equal

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
V = np.load(os.path.join(GOLDEN, "ref_vectors.npz"))


def test_central_17x13_reference_camera_on_gpu():
    case = "GPU vs reference code, central 17x13"
    cam, grid = cio.load_camera_model(os.path.join(GOLDEN, "ref_generic_models_17x13.yaml"))     # F2 reader
    grid = np.ascontiguousarray(np.asarray(grid).reshape(-1, 3))
    lines, jac, ok = eng.unproject(cam, grid, V["c17_px"], with_jacobian=True)
    check_equal(case, "unproject ok flags", int(np.count_nonzero(~ok)))
    check(case, "unproject direction abs", np.abs(lines[:, :3] - V["c17_dirs"]).max(), 1e-14)
    check(case, "unproject jacobian / max", np.abs(jac[:, :3] - V["c17_jac"]).max() / np.abs(V["c17_jac"]).max(), 5e-13)
    px, pok = eng.project(cam, grid, V["c17_pts"])
    check_equal(case, "project ok flags", int(np.count_nonzero(pok != V["c17_reproj_ok"].astype(bool))))
    check(case, "project pixel abs [px]", np.abs(px[pok] - V["c17_reproj"][pok]).max(), 5e-11)
    check(case, "round trip [px] (reference criterion 1e-3)", np.linalg.norm(px[pok] - V["c17_px"][pok], axis=1).max(), 1e-3)
    px2, pok2 = eng.project(cam, grid, V["c17_pts"], init=V["c17_init"])
    check_equal(case, "project-with-initial-estimate ok flags", int(np.count_nonzero(pok2 != V["c17_reproj_init_ok"].astype(bool))))
    check(case, "project-with-initial-estimate pixel abs [px]", np.abs(px2[pok2] - V["c17_reproj_init"][pok2]).max(), 5e-11)
    _, okb = eng.project(cam, grid, V["c17_bad_pts"])
    check_equal(case, "unreachable points flagged", int(np.count_nonzero(okb != V["c17_bad_ok"].astype(bool))))


def test_noncentral_8x8_reference_camera_on_gpu():
    case = "GPU vs reference code, non-central 8x8"
    cam = Camera(NONCENTRAL_GENERIC, 640, 480, 0, 0, 639, 479, 8, 8)
    grid = np.ascontiguousarray(V["n8_grid"])
    lines, jac, ok = eng.unproject(cam, grid, V["n8_px"], with_jacobian=True)
    check_equal(case, "unproject ok flags", int(np.count_nonzero(~ok)))
    check(case, "unproject line abs", np.abs(lines - V["n8_lines"]).max(), 1e-13)
    check(case, "unproject jacobian / max", np.abs(jac - V["n8_jac"]).max() / np.abs(V["n8_jac"]).max(), 5e-13)
    px, pok = eng.project(cam, grid, V["n8_pts"])
    check_equal(case, "project ok flags", int(np.count_nonzero(pok != V["n8_reproj_ok"].astype(bool))))
    check(case, "project pixel abs [px]", np.abs(px[pok] - V["n8_reproj"][pok]).max(), 5e-11)
    check(case, "round trip [px] (reference criterion 1e-3)", np.linalg.norm(px[pok] - V["n8_px"][pok], axis=1).max(), 1e-3)


# ---- round 3: normal equations against the REFERENCE's accumulator, grid Jacobians against the reference's grid model --------
ACC = np.load(os.path.join(GOLDEN, "ref_accumulated.npz"))
from tests.ref_modes import ACC_MODES, load_mode  # noqa: E402


@pytest.mark.parametrize("mode", sorted(ACC_MODES))
def test_normal_equations_against_the_reference_accumulator(mode):
    """Engine: residuals, Jacobians AND accumulation on the GPU.  Expected: the reference's own UpdateEquationAccumulator
    (LV/lm_optimizer_update_accumulator.h compiled by oracle/Makefile) fed with the per-observation Jacobians of the fixture
    (tests/golden/make_ref_fixtures.py).  Every branch of AccumulateModelJacobian (joint_optimization.cc:479-590)."""
    case = f"GPU vs reference accumulator, {mode}"
    pb, st = load_mode(mode, GOLDEN)
    e = eng.Engine(pb)
    e.set_state(st)
    cost = e.debug_accumulate()
    check(case, "cost rel", abs(cost - float(ACC[f"{mode}__cost"])) / float(ACC[f"{mode}__cost"]), 5e-13)
    vec = e.dump(eng.DUMP_COST_VECTOR)
    check_equal(case, "valid mask", int(np.count_nonzero((vec >= 0) != (ACC[f"{mode}__cost_vector"] >= 0))))
    got = {"block_diag_H": np.array([np.triu(b) for b in e.dump(eng.DUMP_BLOCK_DIAG_H)]), "block_diag_b": e.dump(eng.DUMP_BLOCK_DIAG_B),
           "off_diag_H": e.dump(eng.DUMP_OFF_DIAG_H), "dense_H": np.triu(e.dump(eng.DUMP_DENSE_H)), "dense_b": e.dump(eng.DUMP_DENSE_B)}
    for name, a in got.items():
        b = ACC[f"{mode}__{name}"]
        b = np.array([np.triu(x) for x in b]) if name == "block_diag_H" else (np.triu(b) if name == "dense_H" else b)
        if np.abs(b).max() == 0:
            check_equal(case, name + " (all zero)", int(np.count_nonzero(a)))
        else:
            check(case, name + " / max", np.abs(a - b).max() / np.abs(b).max(), 5e-9)
    e.close()


def test_grid_jacobians_against_the_reference_grid_model():
    """k_fd_tasks / k_fd_redo (the 32 finite-difference re-projections per observation) against
    CentralGridModel::ProjectionJacobianWrtIntrinsics (APP/models/central_grid.h:187-245) compiled from the reference."""
    case = "GPU vs reference ProjectionJacobianWrtIntrinsics"
    pb, st = load_mode("central", GOLDEN)
    e = eng.Engine(pb)
    e.set_state(st)
    e.debug_accumulate()
    flags = e.dump(eng.DUMP_FLAGS)
    rec = e.dump(eng.DUMP_JACOBIANS)
    hasj = ((flags >> 1) & 1).astype(bool)
    check_equal(case, "has-Jacobian flags", int(np.count_nonzero(hasj != (ACC["central__m5_ok"] == 1))))
    gj = rec[:, 33:33 + 64].reshape(-1, 2, 32)[hasj]
    ref_j = ACC["central__m5_jac"][hasj]
    check(case, "grid block of the Jacobian records / max", np.abs(gj - ref_j).max() / np.abs(ref_j).max(), 5e-9)
    e.close()


def test_noncentral_grid_jacobians_against_the_reference_model():
    """N3 on the GPU: the 80 finite-difference re-projections per observation (k_fd_tasks<1>) against
    NoncentralGenericModel::ProjectionJacobianWrtIntrinsics (APP/models/noncentral_generic.h:224-283) compiled from the reference."""
    case = "GPU vs reference non-central ProjectionJacobianWrtIntrinsics"
    pb, st = load_mode("noncentral", GOLDEN)
    e = eng.Engine(pb)
    e.set_state(st)
    e.debug_accumulate()
    flags = e.dump(eng.DUMP_FLAGS)
    rec = e.dump(eng.DUMP_JACOBIANS)
    hasj = ((flags >> 1) & 1).astype(bool)
    check_equal(case, "has-Jacobian flags", int(np.count_nonzero(hasj != (ACC["noncentral__m5_ok"] == 1))))
    gj = rec[:, 33:33 + 160].reshape(-1, 2, 80)[hasj]
    ref_j = ACC["noncentral__m5_jac"][hasj]
    check(case, "grid block of the Jacobian records / max", np.abs(gj - ref_j).max() / np.abs(ref_j).max(), 5e-9)
    e.close()
