#!/bin/bash
# Round-5 GPU-box experiments, one case per record under profiles/r05_*.txt (tools/gpu_run.sh is the suite / bench / profile driver).
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/gpu_r05.sh <case>'
# The harness binaries travel with the snapshot (tools/bin is git-ignored, not gpurun-ignored); build them first:
#   F="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form"
#   hipcc $F tools/bench_tail.hip -o tools/bin/bench_tail ; hipcc $F -DCBA_TAIL_RING tools/bench_tail.hip -o tools/bin/bench_tail_ring
#   hipcc $F tools/bench_diag.hip -o tools/bin/bench_diag
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
pmc() { # pmc <name> <binary + args ...> -- <counters ...>: one --pmc pass over the harness, per-dispatch table of the dataflow launches
  local n=$1; shift; local cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
  ( cd /tmp; rm -rf /tmp/pmc_$n; TAILS=8192 REPS=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$n -o pmc -- "${cmd[@]}" > $O/r05_pmc_$n.log 2>&1
    db=$(find /tmp/pmc_$n -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc_per_dispatch.py $db ldlt_tail,gemm_atb > $O/r05_pmc_$n.txt 2>&1 )
  cat $O/r05_pmc_$n.txt
}
case ${1:-help} in
  tail-traffic)      # profiles/r05_tail_traffic.txt: counters of every launch of a factorisation, per dispatch
    pmc fetch $R/tools/bin/bench_tail 12672 12544 -- FETCH_SIZE
    pmc write $R/tools/bin/bench_tail 12672 12544 -- WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
    pmc sq $R/tools/bin/bench_tail 12672 12544 -- SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE ;;
  pair)              # profiles/r05_pair_tasks_in_situ.txt: 64 x 128 REG2 tasks against single tiles, same process
    PAIRS=0,1 TAILS=512,1024 REPS=2 timeout 300 tools/bin/bench_tail 2304 2240 | grep -v "back substitution"
    PAIRS=0,1 TAILLOG=1 TAILS=8192 REPS=4 timeout 300 tools/bin/bench_tail 12672 12544 | grep -v "back substitution"
    for pm in 0 1; do CBA_TAIL_PAIR=$pm pmc fetch_pair$pm $R/tools/bin/bench_tail 12672 12544 -- FETCH_SIZE; done ;;
  ring)              # profiles/r05_ring_kloop_in_situ.txt: 4-stage ring K loop against the 2 x 32 double buffer
    for b in bench_tail bench_tail_ring; do echo "==== $b"; TAILLOG=1 TAILS=6144,8192 REPS=4 timeout 300 tools/bin/$b 12672 12544 | grep -v "back substitution"; done ;;
  converged)         # profiles/r05_converged_parity*.json (the cfg-2 grid takes ~9 minutes of all-core oracle)
    timeout 600 python tools/converged_parity.py --config 1 --imagesets 0 --out $O/r05_converged_parity_cfg1.json | tail -30
    timeout 1500 python tools/converged_parity.py --config 2 --imagesets 60 --out $O/r05_converged_parity.json | tail -30
    timeout 1500 python tools/converged_parity.py --config 2 --imagesets 0 --out $O/r05_converged_parity_cfg2_full.json | tail -30 ;;   # the bench workload itself: ~6 minutes of oracle
  fd-schedules)      # profiles/r05_fd_schedules_bench.txt: pooled vs one task per lane inside the bench trajectories, alternating
    for c in 2 4 3; do for sch in 1 0 1 0; do
      timeout 600 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-convergence --fd-schedule $sch 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg $c schedule $sch: step %.3f ms, t_fd_kernel %.3f, t_jac %.3f' % (d['ms_per_step'], d['stage_ms_per_step']['t_fd_kernel'], d['stage_ms_per_step']['t_jac']))"
    done; done ;;
  fd-lanes)          # profiles/r05_fd_lane_utilisation.txt
    for c in 2 4; do ( cd /tmp; rm -rf /tmp/pmc_v$c
      timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d /tmp/pmc_v$c -o pmc -- python $R/bench.py --config $c --steps 2 --warmup 0 --no-cpu-baseline --no-convergence --fd-schedule ${FD_SCHEDULE:-1} > $O/r05_pmc_valu_cfg$c.log 2>&1
      db=$(find /tmp/pmc_v$c -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc_generic.py $db fd_tasks,fd_pool ); done ;;
  *) sed -n 2,4p $0 ;;
esac
