#!/bin/bash
# round 5, call 4: shared first evaluation in the FD kernel + fused cost pass + fewer event bubbles: whole GPU suite, bench lines, timeline
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rm -f $O/parity_deviations.json
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 -x > $O/r05d_gputests.log 2>&1; echo "pytest rc=$?"; tail -6 $O/r05d_gputests.log
for c in 2 4 3; do
  timeout 600 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline > $O/r05d_bench_cfg$c.log 2>&1; tail -1 $O/r05d_bench_cfg$c.log > $O/r05d_bench_cfg$c.json
done
python - <<'PY'
import json
for c in (2, 4, 3):
    try:
        d=json.load(open(f"gpurun_out/r05d_bench_cfg{c}.json")); print(c, round(d["ms_per_step"],3), round(d["value"],3), {k: round(v,3) for k,v in d["stage_ms_per_step"].items()}, round(d["roofline"]["frac"],3), (d.get("wall_clock_to_convergence") or {}).get("seconds"))
    except Exception as e: print(c, "FAILED", e)
PY
cd /tmp
rm -rf /tmp/prof_c2; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-convergence > $R/$O/r05d_prof_cfg2.log 2>&1
db=$(find /tmp/prof_c2 -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $O/r05d_bench_cfg2_kernel_stats.txt 2>&1
[ -n "$db" ] && python $R/tools/step_timeline.py $db 15 > $O/r05d_step_timeline_cfg2.txt 2>&1
cat $O/r05d_step_timeline_cfg2.txt | tail -45
