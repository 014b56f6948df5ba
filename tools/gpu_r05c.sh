#!/bin/bash
# round 5, call 3: ring K loop (4 stages x 16 rows) in situ vs the 2 x 32 loop; fused cost pass (one host wait per attempt): tests + bench
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
{
for b in bench_tail bench_tail_ring; do
echo "==== $b"
echo "== small"; TAILS=512,1024 REPS=2 timeout 300 tools/bin/$b 2304 2240 2>&1 | grep -v "back substitution"
echo "== cfg-2 size"; TAILLOG=1 TAILS=6144,8192 REPS=4 timeout 300 tools/bin/$b 12672 12544 2>&1 | grep -v "back substitution"
echo "== cfg-3 size"; TAILS=8192 REPS=2 timeout 300 tools/bin/$b 22784 22656 2>&1 | grep -v "back substitution"
done
} > $O/r05c_ring.txt 2>&1
grep -v "chain phases" $O/r05c_ring.txt | tail -40
cd /tmp
rm -rf /tmp/pmc_r
TAILS=8192 REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d /tmp/pmc_r -o pmc -- $R/tools/bin/bench_tail_ring 12672 12544 > $O/r05c_pmc_sq_ring.log 2>&1
db=$(find /tmp/pmc_r -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc_per_dispatch.py $db ldlt_tail > $O/r05c_pmc_sq_ring.txt 2>&1
head -5 $O/r05c_pmc_sq_ring.txt
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_factor_tail.py tests/test_gpu_stragglers.py -q -m gpu -x --timeout 600 2>&1 | tail -5
timeout 600 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > $O/r05c_bench_cfg2.log 2>&1; tail -1 $O/r05c_bench_cfg2.log > $O/r05c_bench_cfg2.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05c_bench_cfg2.json")); print(d["ms_per_step"], d["value"], d["stage_ms_per_step"], d["roofline"]["frac"], d.get("wall_clock_to_convergence"))
PY
