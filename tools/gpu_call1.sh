#!/bin/bash
# GPU call 1 of round 2: full-size parity, regression of the touched model code, PMC counters, ceilings, CU-mask patterns.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out
{ nproc; lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"; free -g; } > $O/r02_host.txt 2>&1
T0=$(date +%s)
timeout 1100 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_vs_ref_vectors.py -q -s --timeout 500 > $O/r02_parity_fullsize.log 2>&1
echo "parity_fullsize rc=$? $(( $(date +%s) - T0 )) s" >> $O/r02_call1_times.txt
T0=$(date +%s)
timeout 500 python -m pytest tests/test_golden_fixtures.py tests/test_gpu_parity.py -q -m gpu --timeout 300 > $O/r02_parity_small.log 2>&1
echo "parity_small rc=$? $(( $(date +%s) - T0 )) s" >> $O/r02_call1_times.txt
# ceilings and CU-mask patterns (standalone linear algebra)
T0=$(date +%s)
( cd /tmp; for n in 8 16 32; do echo "== CBA_PANEL_CUS=$n"; CBA_PANEL_CUS=$n timeout 120 $GRAFT_REPO_ROOT/tools/bin/bench_linalg 12672 3008 2>&1 | grep -E "schur_gemm|masked|K=512|ldlt_factor|panel CUs|status"; done ) > $O/r02_mask_patterns.txt 2>&1
( cd /tmp; timeout 60 $GRAFT_REPO_ROOT/tools/bin/dgemm_ref 12544 3008; timeout 60 $GRAFT_REPO_ROOT/tools/bin/mfma_peak ) > $O/r02_ceilings.txt 2>&1
echo "linalg rc=$? $(( $(date +%s) - T0 )) s" >> $O/r02_call1_times.txt
# PMC passes (own runs, kernel-trace only)
T0=$(date +%s)
cd /tmp
rocprofv3 -L 2>/dev/null | grep -iE "mfma|SQ_BUSY|GRBM_GUI|VALU" | head -60 > $GRAFT_REPO_ROOT/$O/r02_counter_list.txt
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence > $GRAFT_REPO_ROOT/$O/r02_pmc_$i.log 2>&1
  db=$(find /tmp/pmc_$i -name "*.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_pmc_generic.py $db gemm_atb,fd_tasks,accumulate,base_project,near_fused,ldlt_diag,panel_solve > $GRAFT_REPO_ROOT/$O/r02_pmc_$i.txt 2>&1
done
cd $GRAFT_REPO_ROOT
echo "pmc $(( $(date +%s) - T0 )) s" >> $O/r02_call1_times.txt
tail -5 $O/r02_parity_fullsize.log
cat $O/r02_call1_times.txt
