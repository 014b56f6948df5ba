#!/usr/bin/env python
"""Developer probe: validity counts and pass timing on the bench problem (needs a GPU)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from camera_calibration_amd import engine as eng, synthetic as syn
eng.load()
proj = lambda cam, grid, pts: eng.project(cam, grid, pts, device=0)
pb, st0, gt = syn.baseline_config(2, proj, n_imagesets=500)
e = eng.Engine(pb)
e.set_state(st0)
for rep in range(3):
    t0 = time.perf_counter(); c, nv = e.cost(); t1 = time.perf_counter()
    print(f"cost pass {rep}: cost {c:.6g} valid {nv} of {pb.n_obs} ({pb.n_obs - nv} invalid) {1e3 * (t1 - t0):.2f} ms")
rep = e.step()
print("step:", rep.initial_cost, rep.final_cost, rep.n_residuals_valid, rep.n_jacobians_dropped, rep.t_jac, rep.t_cost)
for rep_i in range(3):
    t0 = time.perf_counter(); c, nv = e.cost(); t1 = time.perf_counter()
    print(f"cost pass after step {rep_i}: valid {nv} ({pb.n_obs - nv} invalid) {1e3 * (t1 - t0):.2f} ms")
