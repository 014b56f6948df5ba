#!/bin/bash
# Final check of a build: what the driver runs at round end (GPU tests, smoke, bench) + the record files for profiles/.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
rm -f $O/parity_deviations.json
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 600 > $O/final_gputests.log 2>&1
echo "pytest rc=$?"; tail -3 $O/final_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 2 > $O/final_bench.log 2>&1; tail -1 $O/final_bench.log > $O/final_bench.json
python - <<PY
import json
d=json.load(open("$O/final_bench.json")); print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["stage_ms_per_step"], d["cpu_baseline"]["value"], d["wall_clock_to_convergence"])
PY
