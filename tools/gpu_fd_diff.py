"""Developer check: Jacobian records of the pooled FD schedule vs one task per lane, entry by entry."""
import sys, numpy as np
sys.path.insert(0, ".")
from camera_calibration_amd import engine as eng, synthetic as syn
cfg, n, gw = int(sys.argv[1]), int(sys.argv[2]), (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else None
pb, st, _ = syn.baseline_config(cfg, lambda c, g, p: eng.project(c, g, p), n_imagesets=n, grid_wh=gw)
out = {}
for sched in (1, 0, 1):
    e = eng.Engine(pb, deterministic=True); e.set_fd_schedule(sched); e.set_state(st); e.debug_accumulate()
    J = np.asarray(e.dump(eng.DUMP_JACOBIANS)).reshape(pb.n_obs, -1); fl = e.dump(eng.DUMP_FLAGS)
    cnt = e.L and None
    import ctypes as C
    c3 = (C.c_int64 * 3)(); e.L.cba_debug_fd_redo_counts(e._h, c3)
    print("schedule", sched, "redo counts", list(c3))
    out.setdefault(sched, []).append((J, fl)); e.close()
J1, f1 = out[1][0]; J1b, _ = out[1][1]; J0, f0 = out[0][0]
print("old vs old again:", np.count_nonzero(J1 != J1b), "flags", np.count_nonzero(f1 != f0))
d = np.argwhere(J1 != J0)
print("differing entries", len(d))
for o, k in d[:40]:
    print(o, k, J1[o, k], J0[o, k], abs(J1[o, k] - J0[o, k]) / max(1e-300, abs(J1[o, k])), "flags", f1[o], f0[o])
