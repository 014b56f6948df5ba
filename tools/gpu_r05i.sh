#!/bin/bash
# round 5, call 9: the next pivot on its own chain (d' = p - v^2 / d) vs taken from the updated column: accuracy, block time, factorisation
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
{
for b in bench_diag_vec bench_diag; do echo "==== $b"; timeout 120 tools/bin/$b 200 2>&1 | tail -12; done
for b in bench_tail_vec bench_tail; do echo "==== $b"; TAILLOG=1 TAILS=8192 REPS=4 timeout 300 tools/bin/$b 12672 12544 2>&1 | grep -v "back substitution"; TAILS=512 REPS=2 timeout 300 tools/bin/$b 2304 2240 2>&1 | grep -v "back substitution" | tail -3; TAILS=8192 REPS=2 timeout 300 tools/bin/$b 22784 22656 2>&1 | grep -v "back substitution" | tail -2; done
} 2>&1 | tee $O/r05i_pivot_chain.txt
timeout 900 python -m pytest tests/test_gpu_factor_tail.py tests/test_gpu_parity.py -q -m gpu -x --timeout 600 2>&1 | tail -4
timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-convergence 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg 2: step %.3f ms, value %.3f, t_factor %.3f' % (d['ms_per_step'], d['value'], d['stage_ms_per_step']['t_factor']))"
