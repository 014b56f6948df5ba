#!/bin/bash
# round 5, call 5: lane utilisation of the FD kernel (is it divergence-bound?) + the step timeline
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp
for c in 2 4; do
rm -rf /tmp/pmc_v$c
timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d /tmp/pmc_v$c -o pmc -- python $R/bench.py --config $c --steps 2 --warmup 0 --no-cpu-baseline --no-convergence > $O/r05e_pmc_valu_cfg$c.log 2>&1
db=$(find /tmp/pmc_v$c -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc_generic.py $db fd_tasks,base_project,fd_redo > $O/r05e_pmc_valu_cfg$c.txt 2>&1
cat $O/r05e_pmc_valu_cfg$c.txt
done
rm -rf /tmp/prof_c2; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-convergence > $O/r05e_prof_cfg2.log 2>&1
db=$(find /tmp/prof_c2 -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $O/r05e_bench_cfg2_kernel_stats.txt 2>&1
[ -n "$db" ] && python $R/tools/step_timeline.py $db 15 > $O/r05e_step_timeline_cfg2.txt 2>&1
tail -45 $O/r05e_step_timeline_cfg2.txt
