#!/bin/bash
# round 3, call 1: persistent tail launch -- correctness against the blocked schedule + timing; then the solver parity tests and a bench line
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 300 tools/bin/bench_tail > $O/r03_tail1.txt 2>&1; echo "bench_tail rc=$?"
cat $O/r03_tail1.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_fixtures.py -x -q -m gpu --timeout 300 > $O/r03_call1_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r03_call1_pytest.log
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-convergence > $O/r03_call1_bench.log 2>&1; tail -1 $O/r03_call1_bench.log > $O/r03_call1_bench.json
python - <<PY
import json
d=json.load(open("$O/r03_call1_bench.json")); print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["stage_ms_per_step"])
PY
