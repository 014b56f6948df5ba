#!/bin/bash
# One parameterised GPU-box script:  tools/gpu_run.sh <what> [tag] [args]          (outputs -> gpurun_out/<tag>_*)
#   suite     whole GPU test suite + smoke + default bench line
#   record    suite + bench lines of configs 2 / 3 / 4 (+ cfg 2 in the pose-first order) + kernel stats, step timelines, PMC traffic
#             (FETCH / WRITE, separate passes, configs 2 and 4), MFMA counters
#   profile   record without the test suite / smoke
#   quick     one short bench line of config [args: N extra-bench-args] + the FETCH_SIZE of the GEMM and the dataflow launches
#   orders    kernel-trace timelines of one step: grid-first with 1 strip, grid-first automatic, pose-first   [args: config]
#   gridfirst tools/bin/bench_gridfirst (the three launches of the grid-first order on a synthetic system, chain timelines)
#   chain     tools/bin/bench_diag + bench_tail (the chain's blocked diagonal factorisation alone; the dense dataflow launch)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; mkdir -p $O
WHAT=${1:-suite}; TAG=${2:-r06}; shift; shift
summary() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); st = d.get("stage_ms_per_step", {})
    print(sys.argv[1].split("/")[-1], "ms/step %.2f value %.3f frac %.3f lib %.1f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("library_tflops") or 0),
          {k: round(v, 2) for k, v in st.items()}, d.get("wall_clock_to_convergence", {}).get("seconds"), d.get("cpu_baseline", {}).get("value"),
          {k: (round(v.get("ms_per_step", 0), 2) if isinstance(v, dict) else v) for k, v in d.get("other_configs", {}).items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
trace() {   # trace <dir> <stats-out> <timeline-out> <bench args...>
  local dir=$1 stats=$2 tl=$3; shift; shift; shift
  (cd /tmp; rm -rf $dir; timeout 400 rocprofv3 --kernel-trace --stats -d $dir -o bench -- python $R/bench.py "$@" > $dir.log 2>&1)
  local db=$(find $dir -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $stats 2>&1
  [ -n "$db" ] && python $R/tools/step_timeline.py $db 15 > $tl 2>&1
}
case $WHAT in
  quick)
    CFG=${1:-2}; shift
    timeout 300 python bench.py --config $CFG --steps 12 --warmup 3 --no-cpu-baseline --no-convergence --no-other-configs "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline_gemm'] if 'roofline_gemm' in d else d['roofline']
print(d['config']['elimination'], 'ms/step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}, 'gemm: GFLOP', round(r['flops_per_launch']/1e9,1), 'ms', round(r['avg_launch_ms'],3), 'TF', round(r['achieved'],1))"
    (cd /tmp; rm -rf /tmp/pmc_q; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_q -o pmc -- python $R/bench.py --config $CFG --steps 3 --warmup 0 --no-cpu-baseline --no-convergence --no-other-configs "$@" > /dev/null 2>&1)
    db=$(find /tmp/pmc_q -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_pmc.py $db | grep -i "gemm_atb\|ldlt\|gf_form" | cut -c1-60,80-140 ;;
  orders)
    CFG=${1:-2}
    for variant in "gf1 --elimination 2 --grid-strips 1" "gfauto --elimination 2" "pose --elimination 1"; do
      set -- $variant; name=$1; shift
      trace /tmp/prof_$name $O/${TAG}_kernel_stats_cfg${CFG}_$name.txt $O/${TAG}_step_timeline_cfg${CFG}_$name.txt --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --no-convergence --no-other-configs "$@"
      echo "== $name"; cat $O/${TAG}_step_timeline_cfg${CFG}_$name.txt
    done ;;
  gridfirst)
    TAILLOG=1 REPS=${REPS:-4} timeout 200 tools/bin/bench_gridfirst "$@" 2>&1 | tee $O/${TAG}_gridfirst.txt ;;
  chain)
    timeout 120 tools/bin/bench_diag 200 2>&1 | tee $O/${TAG}_diag.txt
    TAILLOG=1 TAILS=${TAILS:-8192} timeout 300 tools/bin/bench_tail 12672 12544 2>&1 | tee $O/${TAG}_tail.txt ;;
  suite|record|profile)
    if [ "$WHAT" != profile ]; then
      rm -f $O/parity_deviations.json
      timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $O/${TAG}_gputests.log 2>&1; echo "pytest rc=$?"; tail -4 $O/${TAG}_gputests.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
    fi
    timeout 900 python bench.py --steps 20 --warmup 2 > $O/${TAG}_bench_cfg2.log 2>&1; tail -1 $O/${TAG}_bench_cfg2.log > $O/${TAG}_bench_cfg2.json; summary $O/${TAG}_bench_cfg2.json
    [ "$WHAT" = suite ] && exit 0
    timeout 400 python bench.py --steps 20 --warmup 2 --elimination 1 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_cfg2_posefirst.log 2>&1; tail -1 $O/${TAG}_bench_cfg2_posefirst.log > $O/${TAG}_bench_cfg2_posefirst.json; summary $O/${TAG}_bench_cfg2_posefirst.json
    timeout 400 python bench.py --config 3 --steps 8 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_cfg3.log 2>&1; tail -1 $O/${TAG}_bench_cfg3.log > $O/${TAG}_bench_cfg3.json; summary $O/${TAG}_bench_cfg3.json
    timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_cfg4.log 2>&1; tail -1 $O/${TAG}_bench_cfg4.log > $O/${TAG}_bench_cfg4.json; summary $O/${TAG}_bench_cfg4.json
    for c in 2 3 4; do
      trace /tmp/prof_c$c $O/${TAG}_bench_cfg${c}_kernel_stats.txt $O/${TAG}_step_timeline_cfg$c.txt --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-convergence --no-other-configs
    done
    for c in 2 4; do
      for ctr in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp; rm -rf /tmp/pmc_$ctr; timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$ctr -o pmc -- python $R/bench.py --config $c --steps 2 --warmup 0 --no-cpu-baseline --no-convergence --no-other-configs > $O/${TAG}_pmc_cfg${c}_$ctr.log 2>&1)
        db=$(find /tmp/pmc_$ctr -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_pmc.py $db > $O/${TAG}_cfg${c}_pmc_$ctr.txt 2>&1
      done
      python tools/make_pmc_traffic.py $O/${TAG}_cfg${c}_pmc_FETCH_SIZE.txt $O/${TAG}_cfg${c}_pmc_WRITE_SIZE.txt $O/${TAG}_cfg${c}_pmc_traffic.json ${TAG}_cfg${c} 2 > /dev/null
    done
    # MFMA-busy counters of the GEMM and of the dataflow launches (two more passes: busy cycles, instruction counts)
    (cd /tmp; rm -rf /tmp/pmc_m1 /tmp/pmc_m2
     timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/pmc_m1 -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence --no-other-configs > $O/${TAG}_pmc_mfma1.log 2>&1
     timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY -d /tmp/pmc_m2 -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence --no-other-configs > $O/${TAG}_pmc_mfma2.log 2>&1)
    : > $O/${TAG}_pmc_mfma.txt
    for d in /tmp/pmc_m1 /tmp/pmc_m2; do db=$(find $d -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_pmc_generic.py $db gemm_atb,ldlt_tail,ldlt_sparse,back_dataflow,fd_tasks,gf_form >> $O/${TAG}_pmc_mfma.txt 2>&1; done
    tail -12 $O/${TAG}_step_timeline_cfg2.txt; head -8 $O/${TAG}_cfg2_pmc_FETCH_SIZE.txt $O/${TAG}_cfg2_pmc_WRITE_SIZE.txt; cat $O/${TAG}_pmc_mfma.txt ;;
esac
