#!/usr/bin/env python
"""Converged-calibration parity at FULL size in three separate steps (the engine needs a GPU for seconds, the all-core oracle needs
CPU cores for hours -- they do not have to be the same machine):

  gen     (CPU)  build BASELINE configs[N-1] ONCE with the oracle's projection and save problem + perturbed start as .npz
                 (both sides then read the SAME fp32 observations; a problem generated separately on each side could differ in the
                 last bit of a pixel)
  engine  (GPU)  run the engine under the reference's stopping rule (APP/calibration.cc:298, :1123-1125) from that start; save the
                 per-iteration record and the converged state
  oracle  (CPU)  the same loop on the CPU oracle (all host threads), checkpointed after every OptimizeJointly call
  compare        tools/converged_parity.py: compare() on the two records -> profiles/rNN_converged_parity_cfgN_full.json

  python tools/converged_parity_offline.py gen --config 3 --work tools/_work
  python tools/converged_parity_offline.py engine --config 3 --work tools/_work --out gpurun_out        (GPU box)
  python tools/converged_parity_offline.py oracle --config 3 --work tools/_work --threads 6            (hours; background)
  python tools/converged_parity_offline.py compare --config 3 --work tools/_work --engine-dir gpurun_out --out profiles/r06_converged_parity_cfg3_full.json

The oracle is the checker (test infrastructure); the engine side goes through the C-ABI.
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

from camera_calibration_amd import synthetic as syn  # noqa: E402
from camera_calibration_amd.problem import Camera, Problem, State  # noqa: E402


def save_problem(path, pb, st):
    cams = np.array([[c.model_type, c.width, c.height, c.calib_min_x, c.calib_min_y, c.calib_max_x, c.calib_max_y, c.grid_w, c.grid_h] for c in pb.cameras])
    np.savez_compressed(path, cams=cams, n_images=pb.n_images, n_points=pb.n_points, xy=pb.obs_xy, pt=pb.obs_point, im=pb.obs_image, cm=pb.obs_camera,
                        fd=pb.fd_delta, rig=st.rig_tr_global, camrig=st.camera_tr_rig, points=st.points, **{f"grid{i}": g for i, g in enumerate(st.grids)})


def load_problem(path):
    d = np.load(path)
    cams = [Camera(*[int(v) for v in row]) for row in d["cams"]]
    pb = Problem(cams, int(d["n_images"]), int(d["n_points"]), d["xy"], d["pt"], d["im"], d["cm"], fd_delta=float(d["fd"]))
    st = State(d["rig"], d["camrig"], d["points"], [d[f"grid{i}"] for i in range(len(cams))])
    return pb, st


def save_state(path, st, its, seconds):
    np.savez_compressed(path, rig=st.rig_tr_global, camrig=st.camera_tr_rig, points=st.points, its=json.dumps(its), seconds=seconds,
                        **{f"grid{i}": g for i, g in enumerate(st.grids)})


def load_state(path, n_cams):
    d = np.load(path)
    return State(d["rig"], d["camrig"], d["points"], [d[f"grid{i}"] for i in range(n_cams)]), json.loads(str(d["its"])), float(d["seconds"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["gen", "engine", "oracle", "compare"])
    ap.add_argument("--config", type=int, required=True)
    ap.add_argument("--imagesets", type=int, default=0)
    ap.add_argument("--work", default=os.path.join(ROOT, "tools", "_work"))
    ap.add_argument("--out", default="")
    ap.add_argument("--engine-dir", default=os.path.join(ROOT, "gpurun_out"))
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--elimination", type=int, default=0)
    ap.add_argument("--tag", default="r06")
    a = ap.parse_args()
    os.makedirs(a.work, exist_ok=True)
    stem = f"conv_cfg{a.config}" + (f"_n{a.imagesets}" if a.imagesets else "")
    prob = os.path.join(a.work, stem + "_problem.npz")
    if a.what == "gen":
        from oracle import oracle as orc
        orc.set_num_threads(1)
        t0 = time.time()
        pb, st, _ = syn.baseline_config(a.config, lambda c, g, p: orc.project(c, g, p), n_imagesets=a.imagesets or None)
        save_problem(prob, pb, st)
        print(f"{prob}: {pb.n_images} imagesets, {pb.n_obs} observations, D = {pb.dense_dof}, {time.time() - t0:.0f} s")
    elif a.what == "engine":
        import converged_parity as cp
        from camera_calibration_amd import engine as eng
        eng.load(); eng.prepare(0)
        pb, st0 = load_problem(prob)
        # (run_engine uses the default elimination order; an explicit one goes through the Engine keyword)
        e = eng.Engine(pb, elimination=a.elimination)
        order = e.elimination_order()
        e.close()
        _Engine = eng.Engine
        if a.elimination:
            class _E(_Engine):
                def __init__(self, *args, **kw):
                    kw.setdefault("elimination", a.elimination)
                    super().__init__(*args, **kw)
            eng.Engine = _E
        its, st, seconds = cp.run_engine(eng, pb, st0, 100, 1e-4)
        eng.Engine = _Engine
        out = a.out or a.engine_dir
        os.makedirs(out, exist_ok=True)
        suffix = {0: "", 1: "_posefirst", 2: "_gridfirst"}[a.elimination]
        save_state(os.path.join(out, f"{a.tag}_{stem}_engine{suffix}.npz"), st, its, seconds)
        print(json.dumps({"order": order, "iterations": len(its), "attempts": [i["lm_attempts"] for i in its], "final_cost": its[-1]["cost"], "seconds": seconds}))
    elif a.what == "oracle":
        from oracle import oracle as orc
        import converged_parity as cp
        pb, st0 = load_problem(prob)
        ck = os.path.join(a.work, stem + "_oracle.npz")
        orc.set_num_threads(a.threads)
        op = orc.OracleProblem(pb)
        st = st0.copy()
        lam, last, its, t_acc = -1.0, float("inf"), [], 0.0
        if os.path.exists(ck):                     # resume from the checkpoint of an interrupted run
            st, its, t_acc = load_state(ck, pb.n_cameras)
            lam = its[-1]["lambda"]
            last = its[-2]["cost"] if len(its) > 1 else float("inf")
            if its[-1].get("stopped"):
                print("already complete"); return
            last = its[-1]["cost"]
            # the warm-start cache (last_projection) is not checkpointed: a resumed run restarts its projections from the centre /
            # the previous pixel as the reference does on a fresh OptimizeJointly call sequence -- only complete runs are compared
            print("resuming behind iteration", len(its), "-- NOTE: warm-start cache lost, record marked")
        while len(its) < 100:
            t0 = time.time()
            r = op.optimize_jointly(st, 1, lam)
            t_acc += time.time() - t0
            lam = r["final_lambda"]
            it = {"accepted": bool(r["performed"]), "lm_attempts": int(r["lm_attempts"]), "cost": float(r["cost"]), "lambda": float(lam)}
            stop = cp._stop(it["accepted"], it["cost"], last, 1e-4)
            it["stopped"] = bool(stop)
            its.append(it)
            save_state(ck, st, its, t_acc)
            print(f"iteration {len(its)}: cost {it['cost']:.9g} attempts {it['lm_attempts']} accepted {it['accepted']} ({time.time() - t0:.0f} s)", flush=True)
            if stop:
                break
            last = it["cost"]
    else:
        import converged_parity as cp
        pb, st0 = load_problem(prob)
        suffix = {0: "", 1: "_posefirst", 2: "_gridfirst"}[a.elimination]
        e_st, e_its, e_s = load_state(os.path.join(a.engine_dir, f"{a.tag}_{stem}_engine{suffix}.npz"), pb.n_cameras)
        o_st, o_its, o_s = load_state(os.path.join(a.work, stem + "_oracle.npz"), pb.n_cameras)
        assert o_its[-1].get("stopped"), "the oracle run is not complete"
        rec = cp.compare(pb, e_its, e_st, o_its, o_st)
        rec["seconds"] = {"engine_wall_clock_to_convergence": e_s, "oracle": o_s}
        rec["workload"] = (f"BASELINE configs[{a.config - 1}] at full size: {pb.n_cameras} camera(s) {pb.cameras[0].grid_w}x{pb.cameras[0].grid_h}, "
                           f"{pb.n_images} imagesets, {pb.n_obs} observations, D = {pb.dense_dof}")
        rec["stopping_rule"] = "cost >= last_cost - 1e-4 or no update performed, <= 100 iterations (APP/calibration.cc:298, :1123-1125)"
        rec["how"] = ("problem generated once (oracle projection) and read by both sides; engine on an MI355X box (C-ABI, default settings), oracle on the "
                      "build container's CPU cores; tools/converged_parity_offline.py")
        text = json.dumps(rec, indent=1)
        print(text)
        if a.out:
            with open(a.out, "w") as f:
                f.write(text + "\n")


if __name__ == "__main__":
    main()
