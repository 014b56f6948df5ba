#!/usr/bin/env python
"""Measurement for SURVEY 8f row F3 (not the headline bench): CentralGenericModel::FitToDenseModel at the size
ResampleModel uses on BASELINE config 2 -- 84x60 grid (10 080 unknowns), 2048x1456 image, <= 300x300 samples,
3 LM iterations -- on one MI355X, next to the CPU oracle on a bounded smaller case.  Prints one JSON line."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from camera_calibration_amd import engine as eng, grid_fit  # noqa: E402
from camera_calibration_amd.problem import Camera  # noqa: E402


def dense_model(w, h):
    X, Y = np.meshgrid(np.arange(w) + 0.5, np.arange(h) + 0.5)
    x = (X - w / 2) / (0.8 * h); y = (Y - h / 2) / (0.8 * h)
    r2 = x * x + y * y
    k = 1 - 0.12 * r2                                  # mild radial distortion so that the model is not a pinhole
    d = np.stack([x * k, y * k, np.ones_like(x)], -1)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def main():
    eng.load()
    W, H, gw, gh = 2048, 1456, 84, 60
    cam = Camera(0, W, H, 0, 0, W - 1, H - 1, gw, gh)
    dense = dense_model(W, H)
    step = max(1, min(int(round(W // 300)), int(round(H // 300))))
    t0 = time.perf_counter()
    grid0 = grid_fit.initialize_grid_from_dense_model(cam, dense)
    gp, dirs = grid_fit.dense_model_samples(cam, dense, step)
    t_host = time.perf_counter() - t0
    eng.fit_grid_to_directions(cam, grid0, gp, dirs, 1)            # warm-up (allocations, first launches)
    t0 = time.perf_counter()
    grid, rep = eng.fit_grid_to_directions(cam, grid0, gp, dirs, 3)
    t_fit = time.perf_counter() - t0
    dof = 2 * gw * gh
    flops = rep["lm_attempts"] * dof ** 3 / 3.0
    out = {"workload": f"FitToDenseModel {gw}x{gh} grid ({dof} unknowns), {gp.shape[0]} samples (step {step}), 3 LM iterations",
           "seconds_total": t_fit, "host_init_seconds": t_host, "iterations": rep["iterations"], "lm_attempts": rep["lm_attempts"],
           "t_pass_s": rep["t_pass"], "t_solve_s": rep["t_solve"], "cost": [rep["initial_cost"], rep["final_cost"]],
           "ldlt_tflops": flops / rep["t_solve"] / 1e12 if rep["t_solve"] > 0 else None, "ldlt_peak_tflops": 78.6}
    # CPU oracle on a bounded case (dense pivoted LDLT is O(dof^3)): 30x22 grid, same sampling density per cell
    from oracle import oracle as orc
    cam_s = Camera(0, 640, 480, 0, 0, 639, 479, 30, 22)
    dn = dense_model(640, 480)
    g0 = grid_fit.initialize_grid_from_dense_model(cam_s, dn)
    gps, ds = grid_fit.dense_model_samples(cam_s, dn, 2)
    t0 = time.perf_counter()
    _, r_cpu = orc.fit_grid_to_points(30, 22, g0, gps, ds, 1)
    t_cpu = time.perf_counter() - t0
    dof_s = 2 * 30 * 22
    out["cpu_oracle"] = {"sample": f"30x22 grid ({dof_s} unknowns), {gps.shape[0]} samples, 1 iteration", "seconds": t_cpu,
                         "extrapolated_seconds_per_attempt_at_full_size": t_cpu * (dof / dof_s) ** 3, "cores": 1}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
