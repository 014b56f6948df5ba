#!/bin/bash
# FD schedules in the bench trajectory, same box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
for c in 2 4 3; do for sch in 1 0 1 0; do
  timeout 600 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-convergence --fd-schedule $sch > $O/r05h_c${c}_s${sch}.log 2>&1
  tail -1 $O/r05h_c${c}_s${sch}.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg $c schedule $sch: step %.3f ms, t_fd_kernel %.3f, t_jac %.3f' % (d['ms_per_step'], d['stage_ms_per_step']['t_fd_kernel'], d['stage_ms_per_step']['t_jac']))"
done; done 2>&1 | tee $O/r05h_fd_schedules_bench.txt
timeout 600 python -m pytest tests/test_gpu_converged_parity.py -q -m gpu -s --timeout 600 2>&1 | tail -8
