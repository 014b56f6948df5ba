#!/bin/bash
# GPU box: one cfg-N bench line (short) + the FETCH_SIZE of the GEMM; tools/gpu_r06_quick.sh [config] [extra bench args]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; CFG=${1:-2}; shift
cd $R
timeout 300 python bench.py --config $CFG --steps 12 --warmup 3 --no-cpu-baseline --no-convergence --no-other-configs "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline_gemm'] if 'roofline_gemm' in d else d['roofline']
print(d['config']['elimination'], 'ms/step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}, 'gemm: GFLOP', round(r['flops_per_launch']/1e9,1), 'ms', round(r['avg_launch_ms'],3), 'TF', round(r['achieved'],1))"
cd /tmp; rm -rf /tmp/pmc_q
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_q -o pmc -- python $R/bench.py --config $CFG --steps 3 --warmup 0 --no-cpu-baseline --no-convergence --no-other-configs "$@" > /dev/null 2>&1
db=$(find /tmp/pmc_q -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc.py $db | grep -i "gemm_atb\|ldlt\|gf_form" | cut -c1-60,80-140
