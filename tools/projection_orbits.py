#!/usr/bin/env python
"""Developer analysis (CPU, oracle): what do the projections that FAIL do in their 100 outer iterations?
For every failing base projection of a BASELINE config's perturbed initial state, records the loop state (pixel, lambda) at the top
of each outer iteration of both attempts (warm start, centre) and looks for exact repeats: state_n == state_{n-k} bit for bit means
the iteration is periodic with period k from then on (the map is deterministic), so the attempt can never converge."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from camera_calibration_amd import synthetic as syn
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pb, st, _ = syn.baseline_config(cfg, lambda c, g, p: orc.project(c, g, p), n_imagesets=n)
op = orc.OracleProblem(pb, last_projection=pb.obs_xy.astype(np.float64))
cost, vec = op.cost_pass(st)
bad = np.nonzero(vec < 0)[0]
print(f"cfg {cfg}, {n} imagesets: {pb.n_obs} observations, {len(bad)} failing projections")
L = orc.lib()
L.orc_debug_set_state_trace.argtypes = [C.POINTER(C.c_double)]
L.orc_se3_mul.argtypes = [C.POINTER(C.c_double)] * 3
from camera_calibration_amd import se3
hist = {}
for o in bad[:400]:
    cam = pb.cameras[int(pb.obs_camera[o])]; cs = orc.camera_struct(cam)
    T = se3.se3_mul(st.camera_tr_rig[int(pb.obs_camera[o])], st.rig_tr_global[int(pb.obs_image[o])])
    local = np.ascontiguousarray(se3.transform_points(T, st.points[int(pb.obs_point[o])]))
    g = np.ascontiguousarray(st.grids[int(pb.obs_camera[o])]).reshape(-1)
    for attempt, start in (("warm", pb.obs_xy[o].astype(np.float64)), ("centre", np.array([0.5 * (cam.calib_min_x + cam.calib_max_x + 1), 0.5 * (cam.calib_min_y + cam.calib_max_y + 1)]))):
        buf = np.zeros(300); px = start.copy()
        L.orc_debug_set_state_trace(buf.ctypes.data_as(C.POINTER(C.c_double)))
        ok = L.orc_project_with_initial_estimate(C.byref(cs), orc._dp(g), orc._dp(local), orc._dp(px))
        cnt = L.orc_debug_state_trace_count(); L.orc_debug_set_state_trace(None)
        s = buf[:3 * cnt].reshape(cnt, 3).view(np.int64)
        first = None
        for i in range(1, cnt):
            for k in range(1, min(i, 16) + 1):
                if (s[i] == s[i - k]).all():
                    first = (i, k); break
            if first: break
        key = ("ok" if ok else "fail", cnt, None if first is None else first[1], None if first is None else first[0])
        hist[(attempt,) + key[:3]] = hist.get((attempt,) + key[:3], 0) + 1
for k in sorted(hist, key=lambda x: -hist[x]):
    print(f"{k[0]:>6} {k[1]:>4} iterations recorded {k[2]:>3}  first exact period {k[3]}: {hist[k]}")
