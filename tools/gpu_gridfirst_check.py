"""GPU box helper (developer tool): the grid-first elimination order against the pose-first one on the same accumulated system.

For each case: two engines on the same problem and state (deterministic accumulation: both see the same normal equations bit for
bit), cba_debug_accumulate + cba_debug_solve for one lambda, x compared with each other and with LAPACK on the dumped system;
optionally timed LM steps of both orders.  Usage: python tools/gpu_gridfirst_check.py [small|medium|cfg2|cfg3|cfg4] [--steps K]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from camera_calibration_amd import engine as eng  # noqa: E402
from camera_calibration_amd import synthetic as syn  # noqa: E402


def gpu_project(cam, grid, pts):
    return eng.project(cam, grid, pts)


def dense_system(e, pb):
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
    b = np.concatenate([e.dump(eng.DUMP_BLOCK_DIAG_B), e.dump(eng.DUMP_DENSE_B)])
    return H, b


def solve_case(name, pb, st, strips_list, lam_rel=1e-5, lapack=True):
    print(f"== {name}: N={pb.n_images} P={pb.n_points} obs={pb.n_obs} total_dof={pb.total_dof} dense={pb.dense_dof}", flush=True)
    e1 = eng.Engine(pb, deterministic=True, elimination=eng.ELIMINATION_POSE_FIRST)
    e1.set_state(st)
    e1.debug_accumulate()
    H = b = None
    if lapack:
        H, b = dense_system(e1, pb)
        lam = lam_rel * np.trace(H) / pb.total_dof
    else:
        lam = lam_rel * 1.0
    t0 = time.time(); x1 = e1.debug_solve(lam); t1 = time.time() - t0
    x_ref = None
    if lapack:
        x_ref = np.linalg.solve(H + lam * np.eye(H.shape[0]), b)
        print(f"   pose-first vs LAPACK: {np.abs(x1 - x_ref).max() / np.abs(x_ref).max():.2e}   ({t1 * 1e3:.1f} ms incl. host)")
    e1.close()
    for S in strips_list:
        e2 = eng.Engine(pb, deterministic=True, elimination=eng.ELIMINATION_GRID_FIRST, grid_strips=S)
        e2.set_state(st)
        e2.debug_accumulate()
        try:
            t0 = time.time(); x2 = e2.debug_solve(lam); t2 = time.time() - t0
        except eng.EngineError as ex:
            print(f"   grid-first strips={S}: FAILED {ex}")
            e2.close()
            continue
        msg = f"   grid-first strips={S}: vs pose-first {np.abs(x2 - x1).max() / np.abs(x1).max():.2e}"
        if x_ref is not None:
            msg += f", vs LAPACK {np.abs(x2 - x_ref).max() / np.abs(x_ref).max():.2e}"
        print(msg + f"   ({t2 * 1e3:.1f} ms incl. host)", flush=True)
        e2.close()


def step_case(name, pb, st, steps, modes):
    for label, kw in modes:
        e = eng.Engine(pb, **kw)
        e.set_state(st)
        lam = -1.0
        reps = []
        for _ in range(steps):
            r = e.step(lam)
            lam = r.final_lambda
            reps.append(r)
        ks = e.kernel_stats(4)
        upd = f"  in-factor GEMMs of the last step: {ks['launches']} launches, {ks['flops'] / 1e9:.1f} GFLOP, {ks['seconds'] * 1e3:.2f} ms" if ks["launches"] else ""
        tail = reps[1:] if len(reps) > 1 else reps
        ms = lambda f: 1e3 * np.mean([getattr(r, f) for r in tail])
        print(f"   {label:>22}: costs {[f'{r.final_cost:.6g}' for r in reps]} attempts {[r.lm_attempts for r in reps]}"
              f"  t_jac {ms('t_jac'):.2f} t_solve {ms('t_solve'):.2f} (factor {ms('t_factor'):.2f}, gemm {ms('t_gemm'):.2f}) t_cost {ms('t_cost'):.2f} ms{upd}", flush=True)
        e.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=["small", "medium"])
    ap.add_argument("--steps", type=int, default=0)
    a = ap.parse_args()
    eng.prepare(0)
    for case in a.cases:
        if case == "small":
            pb, st, _ = syn.baseline_config(2, gpu_project, n_imagesets=12, grid_wh=(24, 18), lattice_xy=(10, 13))
            solve_case("small 24x18", pb, st, [1, 2])
            pb, st, _ = syn.baseline_config(3, gpu_project, n_imagesets=8, grid_wh=(24, 18), lattice_xy=(10, 13))
            solve_case("small rig 2 x 24x18", pb, st, [1, 2])
            pb, st, _ = syn.baseline_config(4, gpu_project, n_imagesets=8, grid_wh=(20, 16), lattice_xy=(10, 13))
            solve_case("small non-central 20x16", pb, st, [1, 2])
            if a.steps:
                pb, st, _ = syn.baseline_config(2, gpu_project, n_imagesets=12, grid_wh=(24, 18), lattice_xy=(10, 13))
                step_case("small", pb, st, a.steps, [("pose-first", dict(elimination=1)), ("grid-first", dict(elimination=2))])
        elif case == "medium":
            pb, st, _ = syn.baseline_config(2, gpu_project, n_imagesets=60)
            solve_case("cfg-2 grid, 60 imagesets", pb, st, [1, 2, 4])
            if a.steps:
                step_case("medium", pb, st, a.steps, [("pose-first", dict(elimination=1)), ("grid-first S=1", dict(elimination=2, grid_strips=1)),
                                                      ("grid-first S=4", dict(elimination=2, grid_strips=4))])
        elif case in ("cfg2", "cfg3", "cfg4"):
            cfg = int(case[3])
            pb, st, _ = syn.baseline_config(cfg, gpu_project)
            solve_case(case, pb, st, [1, 4] if cfg != 4 else [1, 2, 4], lapack=(cfg == 2))
            if a.steps:
                step_case(case, pb, st, a.steps, [("pose-first", dict(elimination=1)), ("grid-first S=1", dict(elimination=2, grid_strips=1)),
                                                  ("grid-first auto", dict(elimination=2))])


if __name__ == "__main__":
    main()
