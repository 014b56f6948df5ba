#!/bin/bash
# GPU call 3 of round 2: whole GPU test suite after the fixes, bench lines for cfg 2 / 3 / 4 (+ all-reduce path), kernel stats
# of the new finite-difference kernel at cfg 2 and cfg 4, VALU counters for it, thread scaling of the CPU oracle's solve.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
: > $O/r02_call3_times.txt
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -s > $O/r02_gputests3.log 2>&1
echo "pytest rc=$? $(( $(date +%s) - T0 )) s" >> $O/r02_call3_times.txt
grep -E "passed|failed|Error|FAILED|assert" $O/r02_gputests3.log | tail -30
T0=$(date +%s)
timeout 600 python bench.py > $O/r02_bench3_cfg2.log 2>&1; tail -1 $O/r02_bench3_cfg2.log > $O/r02_bench3_cfg2.json
timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > $O/r02_bench3_cfg4.log 2>&1; tail -1 $O/r02_bench3_cfg4.log > $O/r02_bench3_cfg4.json
timeout 400 python bench.py --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-convergence > $O/r02_bench3_cfg3.log 2>&1; tail -1 $O/r02_bench3_cfg3.log > $O/r02_bench3_cfg3.json
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-convergence --force-allreduce > $O/r02_bench3_cfg2_allreduce.log 2>&1; tail -1 $O/r02_bench3_cfg2_allreduce.log > $O/r02_bench3_cfg2_allreduce.json
echo "bench $(( $(date +%s) - T0 )) s" >> $O/r02_call3_times.txt
T0=$(date +%s)
cd /tmp
for cfg in 2 4; do
  rm -rf /tmp/prof_c$cfg; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c$cfg -o bench -- python $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-convergence > $R/$O/r02_prof3_cfg$cfg.log 2>&1
  db=$(find /tmp/prof_c$cfg -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $R/$O/r02_kernel_stats3_cfg$cfg.txt 2>&1
done
for c in "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf /tmp/pmc_$n
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$n -o pmc -- python $R/bench.py --config 4 --steps 2 --warmup 0 --no-cpu-baseline --no-convergence > $R/$O/r02_pmc3_$n.log 2>&1
  db=$(find /tmp/pmc_$n -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc_generic.py $db k_fd_tasks,k_accumulate,k_base_project > $R/$O/r02_pmc3_$n.txt 2>&1
done
cd $R
echo "prof $(( $(date +%s) - T0 )) s" >> $O/r02_call3_times.txt
T0=$(date +%s)
for t in 1 8 16 32 64 128 256; do
  timeout 200 tools/bin/oracle_thread_scaling 8192 $t >> $O/r02_oracle_thread_scaling.txt 2>&1
done
echo "scaling $(( $(date +%s) - T0 )) s" >> $O/r02_call3_times.txt
cat $O/r02_oracle_thread_scaling.txt
cat $O/r02_call3_times.txt
for f in $O/r02_bench3_*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); st=d.get("stage_ms_per_step",{})
    print("$f", d["config"]["workload"][:30], "ms/step %.2f value %.3g" % (d["ms_per_step"], d["value"]), {k: round(v,2) for k,v in st.items()})
except Exception as e: print("$f FAILED", e)
PY
done
