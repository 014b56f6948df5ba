#!/bin/bash
# round 5, call 1: converged-calibration parity (cfg 1, cfg-2 grid), per-dispatch PMC of the dataflow launches, look-ahead sweep
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
nproc; rocm-smi --showproductname 2>/dev/null | head -5
# (1) look-ahead sweep + baseline timelines (no python: fast)
for la in 0 4 8 16; do
  echo "== CBA_TAIL_LA=$la"; CBA_TAIL_LA=$la TAILLOG=1 TAILS=8192 REPS=3 timeout 120 tools/bin/bench_tail 12672 12544 2>&1 | grep -v "back substitution" 
done > $O/r05a_tail_la.txt 2>&1
for la in 0 8; do
  echo "== HELPLOG final launch CBA_TAIL_LA=$la"; CBA_TAIL_LA=$la HELPLOG=1 HL_K0=3584 HL_W=0 timeout 120 tools/bin/bench_tail
  echo "== HELPLOG super-panel CBA_TAIL_LA=$la"; CBA_TAIL_LA=$la HELPLOG=1 HL_K0=0 HL_W=2048 timeout 120 tools/bin/bench_tail
done > $O/r05a_helplog.txt 2>&1
cat $O/r05a_tail_la.txt | grep -v "^   chain phases" | tail -30
# (2) per-dispatch PMC of k_ldlt_tail (three separate passes; counters only, kernel-trace for names / durations)
cd /tmp
pass() { # name counters...
  n=$1; shift
  rm -rf /tmp/pmc_$n
  TAILS=8192 REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$n -o pmc -- $R/tools/bin/bench_tail 12672 12544 > $O/r05a_pmc_$n.log 2>&1
  db=$(find /tmp/pmc_$n -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_pmc_per_dispatch.py $db ldlt_tail,gemm_atb > $O/r05a_pmc_$n.txt 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
head -20 $O/r05a_pmc_fetch.txt $O/r05a_pmc_write.txt $O/r05a_pmc_sq.txt $O/r05a_pmc_sq2.txt
cd $R
# (3) converged parity
timeout 600 python tools/converged_parity.py --config 1 --imagesets 0 --out $O/r05a_converged_cfg1.json > $O/r05a_converged_cfg1.log 2>&1; tail -40 $O/r05a_converged_cfg1.log | head -60
timeout 1500 python tools/converged_parity.py --config 2 --imagesets 60 --out $O/r05a_converged_cfg2_60.json > $O/r05a_converged_cfg2_60.log 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/r05a_converged_cfg1.json", "gpurun_out/r05a_converged_cfg2_60.json"):
    try:
        d = json.load(open(f)); print(f, d["outer_iterations"], d["lm_attempts_per_iteration"], d["decisions_identical"], d["first_divergence"], d["achieved_tolerance"], d["seconds"])
    except Exception as e: print(f, "FAILED", e)
PY
