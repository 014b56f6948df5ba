#!/bin/bash
# GPU call 5 of round 2: dispatch probe (masked vs unmasked queues), tests after the fd_slow hint / threshold 8, bench lines.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
( cd /tmp; timeout 60 $R/tools/bin/dispatch_probe ) > $O/r02_dispatch_probe.txt 2>&1
cat $O/r02_dispatch_probe.txt
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 > $O/r02_gputests5.log 2>&1
echo "pytest rc=$? $(( $(date +%s) - T0 )) s"
grep -E "passed|failed|Error|FAILED|assert" $O/r02_gputests5.log | tail -30
timeout 600 python bench.py --no-cpu-baseline > $O/r02_bench5_cfg2.log 2>&1; tail -1 $O/r02_bench5_cfg2.log > $O/r02_bench5_cfg2.json
timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > $O/r02_bench5_cfg4.log 2>&1; tail -1 $O/r02_bench5_cfg4.log > $O/r02_bench5_cfg4.json
for f in $O/r02_bench5_*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); st=d.get("stage_ms_per_step",{})
    print("$f", d["config"]["workload"][:30], "ms/step %.2f value %.3g" % (d["ms_per_step"], d["value"]), {k: round(v,2) for k,v in st.items()}, d.get("wall_clock_to_convergence",{}).get("seconds"))
except Exception as e: print("$f FAILED", e)
PY
done
