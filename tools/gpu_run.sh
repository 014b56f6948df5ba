#!/bin/bash
# One parameterised GPU-box script (replaces the per-call scripts of rounds 1-2):  tools/gpu_run.sh <what> [tag]
#   suite   whole GPU test suite + smoke + default bench line                      -> gpurun_out/<tag>_*
#   record  suite + bench lines of configs 2 / 3 / 4 + kernel stats + PMC traffic  -> gpurun_out/<tag>_*
#   profile record without the test suite / smoke
#   timeline kernel trace of five cfg-2 steps -> step timeline + kernel stats
#   tail    tools/bin/bench_tail (factorisation tail vs blocked schedule, chain timeline)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
WHAT=${1:-suite}; TAG=${2:-r04}
summary() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); st = d.get("stage_ms_per_step", {})
    print(sys.argv[1], d["config"]["workload"][:40], "ms/step %.2f value %.3f frac %.3f lib %.1f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("library_tflops") or 0),
          {k: round(v, 2) for k, v in st.items()}, d.get("wall_clock_to_convergence", {}).get("seconds"), d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
case $WHAT in
  timeline)
    cd /tmp
    rm -rf /tmp/prof_c2; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-convergence > $R/$O/${TAG}_prof_cfg2.log 2>&1
    db=$(find /tmp/prof_c2 -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $R/$O/${TAG}_bench_cfg2_kernel_stats.txt 2>&1
    [ -n "$db" ] && python $R/tools/step_timeline.py $db 15 > $R/$O/${TAG}_step_timeline_cfg2.txt 2>&1
    cat $R/$O/${TAG}_step_timeline_cfg2.txt ;;
  chain)
    # round 4: the chain's blocked diagonal factorisation alone (phase by phase), then the dataflow launches with the chain timeline
    timeout 120 tools/bin/bench_diag 200 2>&1 | tee $O/${TAG}_diag.txt
    TAILLOG=1 TAILS=${TAILS:-8192} timeout 300 tools/bin/bench_tail 12672 12544 2>&1 | tee $O/${TAG}_tail.txt ;;
  tail)
    TAILLOG=1 TAILS=${TAILS:-1024,6144,8192} timeout 300 tools/bin/bench_tail 12672 12544 2>&1 | tee $O/${TAG}_tail.txt ;;
  suite|record|profile)
    if [ "$WHAT" != profile ]; then
    rm -f $O/parity_deviations.json
    timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $O/${TAG}_gputests.log 2>&1; echo "pytest rc=$?"; tail -4 $O/${TAG}_gputests.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
    fi
    timeout 900 python bench.py --steps 20 --warmup 2 > $O/${TAG}_bench_cfg2.log 2>&1; tail -1 $O/${TAG}_bench_cfg2.log > $O/${TAG}_bench_cfg2.json; summary $O/${TAG}_bench_cfg2.json
    if [ "$WHAT" != suite ]; then
      timeout 400 python bench.py --config 3 --steps 4 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_cfg3.log 2>&1; tail -1 $O/${TAG}_bench_cfg3.log > $O/${TAG}_bench_cfg3.json; summary $O/${TAG}_bench_cfg3.json
      timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_cfg4.log 2>&1; tail -1 $O/${TAG}_bench_cfg4.log > $O/${TAG}_bench_cfg4.json; summary $O/${TAG}_bench_cfg4.json
      # both reduced solves in one invocation with one rank (the two-leg path of bench.py --gpus N: replicated first, distributed second)
      timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-convergence --force-allreduce --both-legs > $O/${TAG}_bench_cfg2_both_legs.log 2>&1; tail -1 $O/${TAG}_bench_cfg2_both_legs.log > $O/${TAG}_bench_cfg2_both_legs.json; summary $O/${TAG}_bench_cfg2_both_legs.json
      timeout 600 python bench.py --config 5 --imagesets 500 --steps 3 --warmup 1 --no-cpu-baseline --no-convergence > $O/${TAG}_bench_cfg5_share.log 2>&1; tail -1 $O/${TAG}_bench_cfg5_share.log > $O/${TAG}_bench_cfg5_share.json; summary $O/${TAG}_bench_cfg5_share.json
      timeout 600 python bench.py --config 5 --imagesets 500 --steps 3 --warmup 1 --no-cpu-baseline --no-convergence --force-allreduce --distributed-solve 1 > $O/${TAG}_bench_cfg5_share_distributed.log 2>&1; tail -1 $O/${TAG}_bench_cfg5_share_distributed.log > $O/${TAG}_bench_cfg5_share_distributed.json; summary $O/${TAG}_bench_cfg5_share_distributed.json
      cd /tmp
      rm -rf /tmp/prof_c2; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-convergence > $R/$O/${TAG}_prof_cfg2.log 2>&1
      db=$(find /tmp/prof_c2 -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $R/$O/${TAG}_bench_cfg2_kernel_stats.txt 2>&1
      [ -n "$db" ] && python $R/tools/step_timeline.py $db 15 > $R/$O/${TAG}_step_timeline_cfg2.txt 2>&1
      for c in 3 4; do
        rm -rf /tmp/prof_c$c; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c$c -o bench -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-convergence > $R/$O/${TAG}_prof_cfg$c.log 2>&1
        db=$(find /tmp/prof_c$c -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $R/$O/${TAG}_bench_cfg${c}_kernel_stats.txt 2>&1
      done
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_$c
        timeout 400 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence > $R/$O/${TAG}_pmc_$c.log 2>&1
        db=$(find /tmp/pmc_$c -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc.py $db > $R/$O/${TAG}_pmc_$c.txt 2>&1
      done
      # MFMA-busy counters of the GEMM and of the dataflow launches (two more passes: busy cycles, instruction counts)
      rm -rf /tmp/pmc_m1 /tmp/pmc_m2
      timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/pmc_m1 -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence > $R/$O/${TAG}_pmc_mfma1.log 2>&1
      timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY -d /tmp/pmc_m2 -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence > $R/$O/${TAG}_pmc_mfma2.log 2>&1
      : > $R/$O/${TAG}_pmc_mfma.txt
      for d in /tmp/pmc_m1 /tmp/pmc_m2; do db=$(find $d -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc_generic.py $db gemm_atb,ldlt_tail,back_dataflow,fd_tasks >> $R/$O/${TAG}_pmc_mfma.txt 2>&1; done
      cd $R; python tools/make_pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE.txt $O/${TAG}_pmc_WRITE_SIZE.txt $O/${TAG}_pmc_traffic.json ${TAG}
      head -6 $O/${TAG}_pmc_FETCH_SIZE.txt $O/${TAG}_pmc_WRITE_SIZE.txt; cat $O/${TAG}_pmc_mfma.txt; head -30 $O/${TAG}_bench_cfg2_kernel_stats.txt
    fi ;;
esac
