#!/bin/bash
# round 5, last call: parity tests through the solve, the cfg-2 bench line, kernel stats and the two PMC traffic passes of the final code
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out; mkdir -p $O; TAG=${1:-r05_v5}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_factor_tail.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|error" | tail -2
timeout 600 python bench.py --steps 20 --warmup 2 > $O/${TAG}_bench_cfg2.log 2>&1; tail -1 $O/${TAG}_bench_cfg2.log > $O/${TAG}_bench_cfg2.json
python - $O/${TAG}_bench_cfg2.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); st = d.get("stage_ms_per_step", {})
print(d["config"]["workload"][:50], "ms/step %.3f value %.3f frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"]), {k: round(v, 3) for k, v in st.items()}, d.get("wall_clock_to_convergence", {}).get("seconds"))
PY
cd /tmp
rm -rf /tmp/prof_c2; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-convergence > $R/$O/${TAG}_prof_cfg2.log 2>&1
db=$(find /tmp/prof_c2 -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $R/$O/${TAG}_bench_cfg2_kernel_stats.txt 2>&1
[ -n "$db" ] && python $R/tools/step_timeline.py $db 15 > $R/$O/${TAG}_step_timeline_cfg2.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence > $R/$O/${TAG}_pmc_$c.log 2>&1
  db=$(find /tmp/pmc_$c -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc.py $db > $R/$O/${TAG}_pmc_$c.txt 2>&1
done
cd $R; python tools/make_pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE.txt $O/${TAG}_pmc_WRITE_SIZE.txt $O/${TAG}_pmc_traffic.json ${TAG} | cut -c1-300
tail -2 $O/${TAG}_step_timeline_cfg2.txt
