#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
echo "--- correctness + timing"; TAILS=1024,6144 timeout 300 tools/bin/bench_tail 2>&1 | grep -v "^   chain"
echo "--- timeline"; TAILLOG=1 TAILS=1024,6144 timeout 300 tools/bin/bench_tail 12672 12544 2>&1 | tee $O/r03_tail15.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden_fixtures.py tests/test_grid_fit.py tests/test_gpu_parity_fullsize.py -x -q -m gpu --timeout 600 > $O/r03_call15_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r03_call15_pytest.log
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-convergence > $O/r03_call15_bench.log 2>&1; tail -1 $O/r03_call15_bench.log > $O/r03_call15_bench.json
python - <<PY
import json
d=json.load(open("$O/r03_call15_bench.json")); print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["stage_ms_per_step"])
PY
