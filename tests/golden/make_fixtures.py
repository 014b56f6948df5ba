#!/usr/bin/env python
"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

    python tests/golden/make_fixtures.py

The reference (C++/CUDA/Qt, needs Eigen + CUDA + Qt) cannot be built in this image, so these vectors
are produced by the oracle (oracle/cba_oracle.c), which itself is pinned to the reference's own known
answers (tests/test_oracle_golden.py: the SchurComplement2 vector of LV/test/lm_optimizer.cc, the
TestOptimizeJointly convergence criterion, the model round trips).  Each fixture stores the complete
inputs (cameras, observations, state) next to the expected outputs, so the GPU tests that read it do
not import the oracle or the problem generator, and a drift of either shows up as a fixture mismatch
in tests/test_golden_fixtures.py::test_oracle_reproduces_fixture.

Fixture = one residual+Jacobian pass, the accumulated normal equations, one Schur solve at the
automatic initial lambda, the state update and the cost of the updated state (the first LM attempt of
OptimizeJointly, joint_optimization.cc:916-925).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from camera_calibration_amd import synthetic as syn  # noqa: E402
from camera_calibration_amd.problem import NONCENTRAL_GENERIC  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = {
    # name: (num_cameras, kwargs of reference_test_problem, problem overrides)
    "first_iteration_1cam_central": (1, dict(seed=11, num_points=40, num_poses=8), {}),
    "first_iteration_2cam_central": (2, dict(seed=12, num_points=40, num_poses=8), {}),
    "first_iteration_1cam_noncentral": (1, dict(seed=13, num_points=40, num_poses=10, model_type=NONCENTRAL_GENERIC),
                                        dict(fd_delta=1e-3)),
    "first_iteration_eliminate_points": (1, dict(seed=14, num_points=40, num_poses=8), dict(eliminate_points=True)),
}


def pack_problem(pb, st):
    cams = np.array([[c.model_type, c.width, c.height, c.calib_min_x, c.calib_min_y, c.calib_max_x, c.calib_max_y,
                      c.grid_w, c.grid_h] for c in pb.cameras], dtype=np.int32)
    d = dict(cameras=cams, n_images=pb.n_images, n_points=pb.n_points, obs_xy=pb.obs_xy, obs_point=pb.obs_point,
             obs_image=pb.obs_image, obs_camera=pb.obs_camera, fd_delta=pb.fd_delta,
             localize_only=int(pb.localize_only), eliminate_points=int(pb.eliminate_points),
             rig_tr_global=st.rig_tr_global, camera_tr_rig=st.camera_tr_rig, points=st.points)
    for i, g in enumerate(st.grids):
        d[f"grid{i}"] = np.asarray(g)
    return d


def make_case(name):
    ncam, kw, over = CASES[name]
    pb, st, _ = syn.reference_test_problem(ncam, orc.project, **kw)
    for k, v in over.items():
        setattr(pb, k, v)
    op = orc.OracleProblem(pb)
    sysm = op.new_system()
    cost, vec, recs = op.jacobian_pass(st, sysm, want_records=True)
    lam = 1e-5 * (np.trace(sysm.dense_H) + sum(np.trace(b) for b in sysm.block_diag_H)) / pb.total_dof
    s2 = orc.System(sysm.block_size, sysm.n_blocks, sysm.dense_dof)
    for fld in ("block_diag_H", "off_diag_H", "dense_H", "block_diag_b", "dense_b"):
        getattr(s2, fld)[...] = getattr(sysm, fld)
    s2.add_lambda(lam)
    x = orc.schur_solve(s2)
    st1 = op.apply_update(st, x)
    cost1, vec1 = op.cost_pass(st1)[:2]
    out = pack_problem(pb, st)
    out.update(
        exp_cost=cost, exp_cost_vector=vec,
        exp_valid=np.array([r.valid for r in recs], dtype=np.uint8),
        exp_has_jacobian=np.array([r.has_jacobian for r in recs], dtype=np.uint8),
        exp_pixels=np.array([[r.pixel[0], r.pixel[1]] for r in recs]),
        exp_block_diag_H=np.array([np.triu(b) for b in sysm.block_diag_H]), exp_block_diag_b=sysm.block_diag_b,
        exp_off_diag_H=sysm.off_diag_H, exp_dense_H=np.triu(sysm.dense_H), exp_dense_b=sysm.dense_b,
        lam=lam, exp_x=x,
        exp_rig_tr_global=st1.rig_tr_global, exp_camera_tr_rig=st1.camera_tr_rig, exp_points=st1.points,
        exp_cost_after=cost1, exp_cost_vector_after=vec1)
    for i, g in enumerate(st1.grids):
        out[f"exp_grid{i}"] = np.asarray(g)
    return out


def main():
    for name in CASES:
        d = make_case(name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **d)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, cost {float(d['exp_cost']):.6g} -> {float(d['exp_cost_after']):.6g}")


if __name__ == "__main__":
    main()
