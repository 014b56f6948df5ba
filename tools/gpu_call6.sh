#!/bin/bash
# GPU call 6 of round 2: dispatch probe with the rotation histogram; bench lines and kernel stats of the current build.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
( cd /tmp; timeout 60 $R/tools/bin/dispatch_probe ) > $O/r02_dispatch_probe.txt 2>&1
cat $O/r02_dispatch_probe.txt
timeout 600 python bench.py > $O/r02_bench6_cfg2.log 2>&1; tail -1 $O/r02_bench6_cfg2.log > $O/r02_bench6_cfg2.json
timeout 400 python bench.py --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-convergence > $O/r02_bench6_cfg3.log 2>&1; tail -1 $O/r02_bench6_cfg3.log > $O/r02_bench6_cfg3.json
timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > $O/r02_bench6_cfg4.log 2>&1; tail -1 $O/r02_bench6_cfg4.log > $O/r02_bench6_cfg4.json
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-convergence --force-allreduce > $O/r02_bench6_cfg2_allreduce.log 2>&1; tail -1 $O/r02_bench6_cfg2_allreduce.log > $O/r02_bench6_cfg2_allreduce.json
cd /tmp
for cfg in 2 4; do
  rm -rf /tmp/prof_c$cfg; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c$cfg -o bench -- python $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-convergence > $R/$O/r02_prof6_cfg$cfg.log 2>&1
  db=$(find /tmp/prof_c$cfg -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $R/$O/r02_kernel_stats6_cfg$cfg.txt 2>&1
done
cd $R
for f in $O/r02_bench6_*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); st=d.get("stage_ms_per_step",{})
    print("$f", d["config"]["workload"][:30], "ms/step %.2f value %.3g" % (d["ms_per_step"], d["value"]), {k: round(v,2) for k,v in st.items()}, d.get("wall_clock_to_convergence",{}).get("seconds"), d.get("trajectory_avg_mobs"))
except Exception as e: print("$f FAILED", e)
PY
done
