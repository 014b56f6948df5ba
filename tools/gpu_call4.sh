#!/bin/bash
# GPU call 4 of round 2: new base-projection (straggler kernel) and cell-accumulation kernels: targeted tests first, then the
# whole suite, bench lines, kernel stats.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
: > $O/r02_call4_times.txt
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_stragglers.py tests/test_gpu_deterministic.py tests/test_gpu_two_ranks.py -q --timeout 300 -s > $O/r02_gputests4a.log 2>&1
echo "targeted pytest rc=$? $(( $(date +%s) - T0 )) s" >> $O/r02_call4_times.txt
grep -E "passed|failed|Error|FAILED|assert|failing projections" $O/r02_gputests4a.log | tail -30
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 > $O/r02_gputests4.log 2>&1
echo "pytest rc=$? $(( $(date +%s) - T0 )) s" >> $O/r02_call4_times.txt
grep -E "passed|failed|Error|FAILED|assert" $O/r02_gputests4.log | tail -30
T0=$(date +%s)
timeout 600 python bench.py > $O/r02_bench4_cfg2.log 2>&1; tail -1 $O/r02_bench4_cfg2.log > $O/r02_bench4_cfg2.json
timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > $O/r02_bench4_cfg4.log 2>&1; tail -1 $O/r02_bench4_cfg4.log > $O/r02_bench4_cfg4.json
timeout 400 python bench.py --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-convergence > $O/r02_bench4_cfg3.log 2>&1; tail -1 $O/r02_bench4_cfg3.log > $O/r02_bench4_cfg3.json
echo "bench $(( $(date +%s) - T0 )) s" >> $O/r02_call4_times.txt
T0=$(date +%s)
cd /tmp
for cfg in 2 4; do
  rm -rf /tmp/prof_c$cfg; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c$cfg -o bench -- python $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-convergence > $R/$O/r02_prof4_cfg$cfg.log 2>&1
  db=$(find /tmp/prof_c$cfg -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $R/$O/r02_kernel_stats4_cfg$cfg.txt 2>&1
done
cd $R
echo "prof $(( $(date +%s) - T0 )) s" >> $O/r02_call4_times.txt
cat $O/r02_call4_times.txt
for f in $O/r02_bench4_*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); st=d.get("stage_ms_per_step",{})
    print("$f", d["config"]["workload"][:30], "ms/step %.2f value %.3g" % (d["ms_per_step"], d["value"]), {k: round(v,2) for k,v in st.items()}, d.get("wall_clock_to_convergence",{}).get("seconds"))
except Exception as e: print("$f FAILED", e)
PY
done
grep -E "k_base_project|k_accumulate|k_fd_" $O/r02_kernel_stats4_cfg2.txt $O/r02_kernel_stats4_cfg4.txt | cut -c1-200
