"""How many B-spline evaluations does the base projection of an observation take (oracle, CPU)?  The GPU runs one lane
per observation, so a pass lasts as long as its slowest lane: this prints the tail of the distribution for a warm-started
cost pass at a BASELINE config.  usage: python tools/projection_iteration_histogram.py [config] [imagesets]"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
from camera_calibration_amd import synthetic as syn
from oracle import oracle as orc

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pb, st, _ = syn.baseline_config(cfg, lambda cam, grid, pts: orc.project(cam, grid, pts), n_imagesets=n_img)
op = orc.OracleProblem(pb)
L = orc.lib()
trace = np.zeros(pb.n_obs, dtype=np.int32)
L.orc_debug_set_eval_trace.argtypes = [C.c_void_p]
orc.set_num_threads(0)
for name in ("cold (from the observed pixel)", "warm"):
    L.orc_debug_set_eval_trace(trace.ctypes.data)
    op.cost_pass(st)
    L.orc_debug_set_eval_trace(None)
    q = np.percentile(trace, [50, 90, 99, 99.9, 100])
    print(f"cfg {cfg}, {pb.n_obs} observations, {name}: evaluations per projection median {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} "
          f"p99.9 {q[3]:.0f} max {q[4]:.0f}; lanes above 20: {int((trace > 20).sum())}, above 100: {int((trace > 100).sum())}")
r = op.optimize_jointly(st, 1, -1.0)
L.orc_debug_set_eval_trace(trace.ctypes.data)
op.cost_pass(st)
L.orc_debug_set_eval_trace(None)
q = np.percentile(trace, [50, 90, 99, 99.9, 100])
print(f"after one LM iteration: median {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} p99.9 {q[3]:.0f} max {q[4]:.0f}; above 20: {int((trace > 20).sum())}, "
      f"above 100: {int((trace > 100).sum())}")
worst = np.argsort(trace)[-5:]
print("worst observations:", worst, trace[worst], "pixels", op.last_projection[worst] if hasattr(op, "last_projection") else "")
