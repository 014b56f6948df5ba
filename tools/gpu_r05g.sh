#!/bin/bash
# round 5, call 7: FD schedules A/B per config (same box) + the full suite + record
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python - > $O/r05g_fd_schedules.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
from camera_calibration_amd import engine as eng, synthetic as syn
eng.load(); eng.prepare(0)
print("# finite-difference kernel: pooled schedule (cba_set_fd_schedule 0) against one task per lane (1); t_fd_kernel = device-side span of the main FD launch (cba_kernel_stats 3), median of 5 Jacobian passes")
for cfg, n in ((2, None), (4, None), (3, 500)):
    pb, st, _ = syn.baseline_config(cfg, lambda c, g, p: eng.project(c, g, p), n_imagesets=n)
    for sched in (1, 0, 1, 0):
        e = eng.Engine(pb); e.set_fd_schedule(sched); e.set_state(st)
        ts = []
        for i in range(6):
            e.set_state(st); r = e.step(-1.0, 1); ts.append(e.kernel_stats(3)["seconds"] * 1e3)
        ts = sorted(ts[1:])
        print(f"cfg {cfg} ({pb.n_obs} observations, {pb.n_cameras} camera(s)) schedule {sched}: t_fd_kernel {ts[len(ts)//2]:.3f} ms (min {ts[0]:.3f})")
        e.close()
PY
cat $O/r05g_fd_schedules.txt
rm -f $O/parity_deviations.json
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $O/r05g_gputests.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r05g_gputests.log
