#!/usr/bin/env python
"""Converged-calibration parity: HIP engine and CPU oracle side by side UNDER THE REFERENCE'S STOPPING RULE.

Both sides run the outer loop of RunBundleAdjustment (APP/calibration.cc:187-304): repeated OptimizeJointly(max_iteration_count = 1)
carrying lambda, stop when no update was performed or `cost >= last_cost - 1e-4` (APP/calibration.cc:298), at most 100 iterations
(:1123-1125), from the same perturbed start.  Compared (what BASELINE.json's north_star calls "converged intrinsics, poses and
pattern geometry ... to a stated fp64 tolerance"):

  * the number of outer iterations, and per iteration the LM attempt count and the accept decision (LV/lm_optimizer.h:943-977) --
    exact until the first divergence; if the two sides part, the record says at which iteration and at what cost difference;
  * per-iteration cost and lambda (relative);
  * the final cost (relative; target 1e-9);
  * points / poses / grids of the converged state, raw AND after gauge alignment (target 1e-7 relative) -- the reference's own test
    leaves converged parameters unchecked because of the gauge freedom (APP/test/util.h:557-565): a similarity transform of the
    pattern (scale, rotation, translation; compensated by the poses) and a rotation of the camera frame (compensated by the
    direction grid) change no residual, and nothing but the LM damping holds those directions.

  python tools/converged_parity.py --config 2 --imagesets 60 --out profiles/r05_converged_parity.json     (GPU box; all host threads)

`run_pair` is what tests/test_gpu_converged_parity.py calls for BASELINE configs[0].  The oracle is the checker here, never the
thing measured (test infrastructure; the engine side goes through the C-ABI).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from camera_calibration_amd import se3  # noqa: E402


# ---- gauge alignment -------------------------------------------------------------------------------------------------
def _kabsch(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """Rotation Q minimising sum |Q a_i - b_i|^2."""
    U, _, Vt = np.linalg.svd(B.T @ A)
    d = np.sign(np.linalg.det(U @ Vt))
    return U @ np.diag([1.0, 1.0, d]) @ Vt


def _umeyama(P: np.ndarray, Pb: np.ndarray):
    """Similarity (s, R, t) minimising sum |s R p_i + t - pb_i|^2."""
    mu, mub = P.mean(0), Pb.mean(0)
    X, Y = P - mu, Pb - mub
    U, S, Vt = np.linalg.svd(Y.T @ X)
    d = np.sign(np.linalg.det(U @ Vt))
    D = np.diag([1.0, 1.0, d])
    R = U @ D @ Vt
    s = float(np.trace(np.diag(S) @ D) / (X * X).sum())
    return s, R, mub - s * R @ mu


def gauge_aligned_deviation(pb, a, b) -> dict:
    """State `a` mapped onto the gauge of state `b`, then compared.  Central-generic cameras (direction grids).

    Gauge: p' = s R p + t (pattern), x_cam' = s Q x_cam (camera frame rotated by Q; the scale leaves directions alone), hence
    directions d' = Q d and image_tr_global' = (Q R_i R^T, s Q t_i - R_i' t).  (s, R, t) from the points, Q from the directions of
    the first camera's grid."""
    out = {}
    s, R, t = _umeyama(a.points, b.points)
    extent = float(np.abs(b.points - b.points.mean(0)).max())
    out["gauge"] = {"scale_minus_1": s - 1.0, "rotation_minus_identity_max": float(np.abs(R - np.eye(3)).max()),
                    "translation_over_extent": float(np.abs(t).max() / extent)}
    out["points_raw_rel"] = float(np.abs(a.points - b.points).max() / extent)
    out["points_aligned_rel"] = float(np.abs((s * a.points @ R.T + t) - b.points).max() / extent)
    central = [c for c in range(pb.n_cameras) if pb.cameras[c].model_type == 0]
    Q = np.eye(3)
    if central:
        c0 = central[0]
        Q = _kabsch(a.grids[c0].reshape(-1, 3), b.grids[c0].reshape(-1, 3))
        out["gauge"]["camera_rotation_minus_identity_max"] = float(np.abs(Q - np.eye(3)).max())
        out["grids_raw_abs"] = max(float(np.abs(a.grids[c] - b.grids[c]).max()) for c in central)
        out["grids_aligned_abs"] = max(float(np.abs(a.grids[c].reshape(-1, 3) @ Q.T - b.grids[c].reshape(-1, 3)).max()) for c in central)
    noncentral = [c for c in range(pb.n_cameras) if pb.cameras[c].model_type != 0]
    if noncentral:      # direction + point grids of the non-central model: raw only (its gauge has more directions than (s, R, t, Q) covers)
        out["noncentral_grids_raw_abs"] = max(float(np.abs(a.grids[c] - b.grids[c]).max()) for c in noncentral)
    # poses: image_tr_global = camera_tr_rig[0] * rig_tr_global for the single-camera case; compare the composed transforms of
    # camera 0 (what the residuals see)
    Ta = se3.se3_mul(a.camera_tr_rig[0], a.rig_tr_global)
    Tb = se3.se3_mul(b.camera_tr_rig[0], b.rig_tr_global)
    Ra, Rb = se3.quat_to_matrix(Ta[:, :4]), se3.quat_to_matrix(Tb[:, :4])
    ta, tb = Ta[:, 4:], Tb[:, 4:]
    tscale = float(np.abs(tb).max())
    out["pose_rotation_raw_abs"] = float(np.abs(Ra - Rb).max())
    out["pose_translation_raw_rel"] = float(np.abs(ta - tb).max() / tscale)
    Ra2 = np.einsum("ij,njk,lk->nil", Q, Ra, R)
    ta2 = s * ta @ Q.T - np.einsum("nij,j->ni", Ra2, t)
    out["pose_rotation_aligned_abs"] = float(np.abs(Ra2 - Rb).max())
    out["pose_translation_aligned_rel"] = float(np.abs(ta2 - tb).max() / tscale)
    return out


# ---- the two outer loops -----------------------------------------------------------------------------------------------
def _stop(accepted: bool, cost: float, last_cost: float, threshold: float) -> bool:
    return (not accepted) or cost >= last_cost - threshold


def run_engine(eng, pb, st0, max_iterations=100, threshold=1e-4, deterministic=False, device=0, **engine_kwargs):
    e = eng.Engine(pb, device=device, deterministic=deterministic, **engine_kwargs)
    try:
        e.set_state(st0)
        lam, last = -1.0, float("inf")
        its = []
        t0 = time.perf_counter()
        for _ in range(max_iterations):
            r = e.step(lam)
            lam = r.final_lambda
            its.append({"accepted": bool(r.accepted), "lm_attempts": int(r.lm_attempts), "cost": float(r.final_cost),
                        "initial_cost": float(r.initial_cost), "lambda": float(lam)})
            if _stop(bool(r.accepted), r.final_cost, last, threshold):
                break
            last = r.final_cost
        seconds = time.perf_counter() - t0
        return its, e.get_state(st0), seconds
    finally:
        e.close()


def run_oracle(orc, pb, st0, max_iterations=100, threshold=1e-4, threads=0):
    orc.set_num_threads(threads)
    try:
        op = orc.OracleProblem(pb)
        st = st0.copy()
        lam, last = -1.0, float("inf")
        its = []
        t0 = time.perf_counter()
        for _ in range(max_iterations):
            r = op.optimize_jointly(st, 1, lam)
            lam = r["final_lambda"]
            its.append({"accepted": bool(r["performed"]), "lm_attempts": int(r["lm_attempts"]), "cost": float(r["cost"]),
                        "lambda": float(lam)})
            if _stop(bool(r["performed"]), r["cost"], last, threshold):
                break
            last = r["cost"]
        return its, st, time.perf_counter() - t0
    finally:
        orc.set_num_threads(1)


def compare(pb, eng_its, eng_state, orc_its, orc_state) -> dict:
    n = min(len(eng_its), len(orc_its))
    first_div = None
    for i in range(n):
        if eng_its[i]["accepted"] != orc_its[i]["accepted"] or eng_its[i]["lm_attempts"] != orc_its[i]["lm_attempts"]:
            first_div = i
            break
    if first_div is None and len(eng_its) != len(orc_its):
        first_div = n
    same = n if first_div is None else first_div
    rel = lambda x, y: abs(x - y) / max(abs(y), 1e-300)
    out = {
        "outer_iterations": {"engine": len(eng_its), "oracle": len(orc_its)},
        "lm_attempts_per_iteration": {"engine": [i["lm_attempts"] for i in eng_its], "oracle": [i["lm_attempts"] for i in orc_its]},
        "accepted_per_iteration": {"engine": [i["accepted"] for i in eng_its], "oracle": [i["accepted"] for i in orc_its]},
        "decisions_identical": first_div is None,
        "first_divergence": None if first_div is None else {
            "iteration": first_div,
            "cost_rel_before": rel(eng_its[first_div - 1]["cost"], orc_its[first_div - 1]["cost"]) if 0 < first_div <= n else None},
        "cost_rel_per_iteration": [rel(eng_its[i]["cost"], orc_its[i]["cost"]) for i in range(n)],
        "lambda_rel_per_iteration": [rel(eng_its[i]["lambda"], orc_its[i]["lambda"]) for i in range(same)],
        "final_cost": {"engine": eng_its[-1]["cost"], "oracle": orc_its[-1]["cost"], "rel": rel(eng_its[-1]["cost"], orc_its[-1]["cost"])},
        "state": gauge_aligned_deviation(pb, eng_state, orc_state),
    }
    st = out["state"]
    out["achieved_tolerance"] = {
        "final_cost_rel": out["final_cost"]["rel"],
        "state_aligned": max(st["points_aligned_rel"], st.get("grids_aligned_abs", 0.0), st["pose_rotation_aligned_abs"],
                             st["pose_translation_aligned_rel"]),
        "state_raw": max(st["points_raw_rel"], st.get("grids_raw_abs", 0.0), st.get("noncentral_grids_raw_abs", 0.0), st["pose_rotation_raw_abs"],
                         st["pose_translation_raw_rel"]),
        "targets": {"final_cost_rel": 1e-9, "state": 1e-7, "source": "BASELINE.md section 2"},
    }
    return out


def run_pair(eng, orc, pb, st0, max_iterations=100, threshold=1e-4, threads=0, deterministic=False, **engine_kwargs) -> dict:
    e_its, e_st, e_s = run_engine(eng, pb, st0, max_iterations, threshold, deterministic, **engine_kwargs)
    o_its, o_st, o_s = run_oracle(orc, pb, st0, max_iterations, threshold, threads)
    out = compare(pb, e_its, e_st, o_its, o_st)
    out["seconds"] = {"engine_wall_clock_to_convergence": e_s, "oracle": o_s}
    out["engine_mode"] = "deterministic (fixed-point accumulation)" if deterministic else "default (fp64 atomics)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--imagesets", type=int, default=60, help="0 = the config's own count")
    ap.add_argument("--threads", type=int, default=0, help="oracle threads (0 = all)")
    ap.add_argument("--max-iterations", type=int, default=100)
    ap.add_argument("--deterministic", action="store_true")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from camera_calibration_amd import engine as eng, synthetic as syn
    from oracle import oracle as orc
    eng.load()
    eng.prepare(0)
    pb, st0, _ = syn.baseline_config(args.config, lambda cam, grid, pts: eng.project(cam, grid, pts), n_imagesets=args.imagesets or None)
    rec = run_pair(eng, orc, pb, st0, args.max_iterations, 1e-4, args.threads, args.deterministic)
    rec["workload"] = (f"BASELINE configs[{args.config - 1}] grid ({pb.cameras[0].grid_w}x{pb.cameras[0].grid_h}), {pb.n_images} imagesets, "
                       f"{pb.n_obs} observations, D = {pb.dense_dof}")
    rec["stopping_rule"] = "cost >= last_cost - 1e-4 or no update performed, <= 100 iterations (APP/calibration.cc:298, :1123-1125)"
    rec["host"] = {"cpu_count": os.cpu_count(), "oracle_threads": args.threads or os.cpu_count()}
    text = json.dumps(rec, indent=1)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
