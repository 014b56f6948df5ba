#!/bin/bash
# GPU call 7 of round 2: SE-balanced slot assignment of the bulk GEMM on the CU-masked streams: A/B in the linear-algebra
# harness, then the GPU tests that exercise the factorisation, then bench lines.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
( cd /tmp
  echo "== SE-balanced (default)"; timeout 120 $R/tools/bin/bench_linalg 12672 3008 2>&1 | grep -E "schur_gemm|masked|K=512|ldlt_factor|status"
  echo "== CBA_NO_SE_BALANCE=1";   CBA_NO_SE_BALANCE=1 timeout 120 $R/tools/bin/bench_linalg 12672 3008 2>&1 | grep -E "schur_gemm|masked|K=512|ldlt_factor|status"
) > $O/r02_se_balance_ab.txt 2>&1
cat $O/r02_se_balance_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_deterministic.py -q --timeout 600 > $O/r02_gputests7.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|Error|FAILED|assert" $O/r02_gputests7.log | tail -20
timeout 600 python bench.py --no-cpu-baseline > $O/r02_bench7_cfg2.log 2>&1; tail -1 $O/r02_bench7_cfg2.log > $O/r02_bench7_cfg2.json
timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline --no-convergence > $O/r02_bench7_cfg4.log 2>&1; tail -1 $O/r02_bench7_cfg4.log > $O/r02_bench7_cfg4.json
timeout 400 python bench.py --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-convergence > $O/r02_bench7_cfg3.log 2>&1; tail -1 $O/r02_bench7_cfg3.log > $O/r02_bench7_cfg3.json
for f in $O/r02_bench7_*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); st=d.get("stage_ms_per_step",{})
    print("$f", d["config"]["workload"][:30], "ms/step %.2f value %.3g" % (d["ms_per_step"], d["value"]), {k: round(v,2) for k,v in st.items()}, d.get("roofline",{}).get("achieved"), d.get("wall_clock_to_convergence",{}).get("seconds"))
except Exception as e: print("$f FAILED", e)
PY
done
