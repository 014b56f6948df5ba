#!/bin/bash
# GPU call 9 of round 2: the record run of the current build -- whole GPU suite, bench lines, kernel stats, PMC traffic.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
rm -f $O/parity_deviations.json
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 > $O/r02_gputests9.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|Error|FAILED|assert" $O/r02_gputests9.log | tail -20
timeout 600 python bench.py > $O/r02_v5_bench_cfg2.log 2>&1; tail -1 $O/r02_v5_bench_cfg2.log > $O/r02_v5_bench_cfg2.json
timeout 400 python bench.py --config 3 --steps 4 --warmup 1 --no-cpu-baseline > $O/r02_v5_bench_cfg3.log 2>&1; tail -1 $O/r02_v5_bench_cfg3.log > $O/r02_v5_bench_cfg3.json
timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > $O/r02_v5_bench_cfg4.log 2>&1; tail -1 $O/r02_v5_bench_cfg4.log > $O/r02_v5_bench_cfg4.json
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-convergence --force-allreduce > $O/r02_v5_bench_cfg2_allreduce.log 2>&1; tail -1 $O/r02_v5_bench_cfg2_allreduce.log > $O/r02_v5_bench_cfg2_allreduce.json
cd /tmp
for cfg in 2 4; do
  rm -rf /tmp/prof_c$cfg; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c$cfg -o bench -- python $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-convergence > $R/$O/r02_v5_prof_cfg$cfg.log 2>&1
  db=$(find /tmp/prof_c$cfg -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $R/$O/r02_v5_bench_cfg${cfg}_kernel_stats.txt 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence > $R/$O/r02_v5_pmc_$c.log 2>&1
  db=$(find /tmp/pmc_$c -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc.py $db > $R/$O/r02_v5_pmc_$c.txt 2>&1
done
cd $R
for f in $O/r02_v5_bench_*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); st=d.get("stage_ms_per_step",{})
    print("$f", d["config"]["workload"][:30], "ms/step %.2f value %.3g" % (d["ms_per_step"], d["value"]), {k: round(v,2) for k,v in st.items()}, d.get("roofline",{}).get("achieved"), d.get("wall_clock_to_convergence",{}).get("seconds"), d.get("trajectory_avg_mobs"))
except Exception as e: print("$f FAILED", e)
PY
done
head -5 $O/r02_v5_pmc_FETCH_SIZE.txt $O/r02_v5_pmc_WRITE_SIZE.txt
