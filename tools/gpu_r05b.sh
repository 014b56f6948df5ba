#!/bin/bash
# round 5, call 2: REG2 (64 x 128) tasks of the dataflow launches in situ: correctness vs single-tile tasks, time, FETCH_SIZE
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
{
echo "== small cases, every junction"; PAIRS=0,1 TAILS=512,1024 REPS=2 timeout 300 tools/bin/bench_tail 2304 2240 2>&1 | grep -v "back substitution"
echo "== cfg-2 size"; PAIRS=0,1 TAILLOG=1 TAILS=8192 REPS=4 timeout 300 tools/bin/bench_tail 12672 12544 2>&1 | grep -v "back substitution"
echo "== cfg-2 size, tail sweep with pairs"; CBA_TAIL_PAIR=1 TAILS=6144,8192,10240 REPS=3 timeout 300 tools/bin/bench_tail 12672 12544 2>&1 | grep -v "back substitution"
echo "== cfg-3 size"; PAIRS=0,1 TAILS=8192 REPS=2 timeout 300 tools/bin/bench_tail 22784 22656 2>&1 | grep -v "back substitution"
} > $O/r05b_pair.txt 2>&1
grep -v "chain phases" $O/r05b_pair.txt | tail -40
cd /tmp
for pm in 0 1; do
  rm -rf /tmp/pmc_f$pm
  CBA_TAIL_PAIR=$pm TAILS=8192 REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f$pm -o pmc -- $R/tools/bin/bench_tail 12672 12544 > $O/r05b_pmc_fetch_pair$pm.log 2>&1
  db=$(find /tmp/pmc_f$pm -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc_per_dispatch.py $db ldlt_tail > $O/r05b_pmc_fetch_pair$pm.txt 2>&1
  rm -rf /tmp/pmc_s$pm
  CBA_TAIL_PAIR=$pm TAILS=8192 REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d /tmp/pmc_s$pm -o pmc -- $R/tools/bin/bench_tail 12672 12544 > $O/r05b_pmc_sq_pair$pm.log 2>&1
  db=$(find /tmp/pmc_s$pm -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc_per_dispatch.py $db ldlt_tail > $O/r05b_pmc_sq_pair$pm.txt 2>&1
done
head -8 $O/r05b_pmc_fetch_pair0.txt $O/r05b_pmc_fetch_pair1.txt $O/r05b_pmc_sq_pair0.txt $O/r05b_pmc_sq_pair1.txt
cd $R
timeout 900 python -m pytest tests/test_gpu_converged_parity.py -q -m gpu -s --timeout 600 2>&1 | tail -12
