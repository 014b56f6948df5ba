#!/usr/bin/env python
"""GPU box: soak of the solve path -- the same calibration run to convergence over and over (the reference's stopping rule,
APP/calibration.cc:298, :1123-1125), one engine per configuration, the state reset before every run.

The dataflow launches of the factorisation synchronise through device-scope flags with bounded waits; a lost wake-up or a task order
that can starve would show as CBA_ERR_TIMEOUT (an exception here) once in many launches, a race as a run whose attempt counts or final
cost differ from the others'.  Counted per configuration: runs, solves, exceptions, the (iterations, attempts) sequences the runs took
(the first run starts with an empty warm-start cache of the projections, every later one with what the previous run left -- the engine
keeps that cache across cba_set_state as the reference keeps `last_projection` in its features), spread of the final cost (default
accumulation: fp64 atomics, so the last digits differ from run to run).

  python tools/gpu_soak.py --configs 2,3,4 --runs 200,40,60 --out gpurun_out/r06_soak.json
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from camera_calibration_amd import engine as eng  # noqa: E402
from camera_calibration_amd import synthetic as syn  # noqa: E402
import converged_parity as cp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4")
    ap.add_argument("--runs", default="200,40,60")
    ap.add_argument("--elimination", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    eng.load(); eng.prepare(0)
    proj = lambda cam, grid, pts: eng.project(cam, grid, pts, device=0)
    rec = {"how": "tools/gpu_soak.py: one engine per configuration, state reset before every run, run to the reference's stopping rule", "configs": {}}
    for cfg, runs in zip([int(v) for v in a.configs.split(",")], [int(v) for v in a.runs.split(",")]):
        pb, st0, _ = syn.baseline_config(cfg, proj)
        e = eng.Engine(pb, device=0, elimination=a.elimination)
        order = e.elimination_order()
        seqs, costs, errors, solves, t_total = {}, [], [], 0, 0.0
        for run in range(runs):
            try:
                e.set_state(st0)
                lam, last, seq = -1.0, float("inf"), []
                t0 = time.perf_counter()
                for _ in range(100):
                    r = e.step(lam)
                    lam = r.final_lambda
                    seq.append((bool(r.accepted), int(r.lm_attempts)))
                    solves += int(r.lm_attempts)
                    if cp._stop(bool(r.accepted), r.final_cost, last, 1e-4):
                        break
                    last = r.final_cost
                t_total += time.perf_counter() - t0
                costs.append(float(r.final_cost))
                key = ("first run (empty warm-start cache): " if run == 0 else "later runs (cache of the previous run): ") + str([n for _, n in seq]) + \
                    ("" if all(acc for acc, _ in seq) else " accepted " + str([int(acc) for acc, _ in seq]))
                seqs[key] = seqs.get(key, 0) + 1
            except Exception as ex:  # noqa: BLE001  (the point of the soak: count them)
                errors.append(f"run {run}: {ex!r}")
        e.close()
        c = np.array(costs)
        rec["configs"][f"cfg{cfg}"] = {
            "workload": f"BASELINE configs[{cfg - 1}]: {pb.n_images} imagesets, {pb.n_obs} observations", "elimination": order,
            "runs": runs, "solves": solves, "exceptions": len(errors), "exception_texts": errors[:5],
            "attempt_sequences_and_how_many_runs_took_them": seqs, "final_cost_min": float(c.min()) if c.size else None,
            "final_cost_rel_spread": float((c.max() - c.min()) / abs(c.mean())) if c.size else None,
            "mean_seconds_to_convergence": t_total / max(1, len(costs))}
        print(json.dumps({f"cfg{cfg}": rec["configs"][f"cfg{cfg}"]}), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            f.write(json.dumps(rec, indent=1) + "\n")


if __name__ == "__main__":
    main()
