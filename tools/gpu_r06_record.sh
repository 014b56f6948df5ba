#!/bin/bash
# GPU box, round 6: the record of the final code.   tools/gpu_r06_record.sh <tag> [suite]
#   bench lines of configs 2 / 3 / 4 (+ cfg-2 pose-first for comparison), kernel stats + step timelines, PMC traffic (FETCH / WRITE,
#   separate passes) for cfg 2 and cfg 4, MFMA counters; with `suite`: the whole GPU test suite + smoke first
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r06}
cd $R
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
if [ "$2" = suite ]; then
  rm -f $O/parity_deviations.json
  timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $O/${TAG}_gputests.log 2>&1; echo "pytest rc=$?"; tail -4 $O/${TAG}_gputests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
fi
timeout 900 python bench.py --steps 20 --warmup 2 > $O/${TAG}_bench_cfg2.log 2>&1; tail -1 $O/${TAG}_bench_cfg2.log > $O/${TAG}_bench_cfg2.json; summary $O/${TAG}_bench_cfg2.json
timeout 400 python bench.py --steps 20 --warmup 2 --elimination 1 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_cfg2_posefirst.log 2>&1; tail -1 $O/${TAG}_bench_cfg2_posefirst.log > $O/${TAG}_bench_cfg2_posefirst.json; summary $O/${TAG}_bench_cfg2_posefirst.json
timeout 400 python bench.py --config 3 --steps 8 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_cfg3.log 2>&1; tail -1 $O/${TAG}_bench_cfg3.log > $O/${TAG}_bench_cfg3.json; summary $O/${TAG}_bench_cfg3.json
timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_cfg4.log 2>&1; tail -1 $O/${TAG}_bench_cfg4.log > $O/${TAG}_bench_cfg4.json; summary $O/${TAG}_bench_cfg4.json
cd /tmp
for c in 2 3 4; do
  rm -rf /tmp/prof_c$c; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c$c -o bench -- python $R/bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-convergence --no-other-configs > $O/${TAG}_prof_cfg$c.log 2>&1
  db=$(find /tmp/prof_c$c -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $O/${TAG}_bench_cfg${c}_kernel_stats.txt 2>&1
  [ -n "$db" ] && python $R/tools/step_timeline.py $db 15 > $O/${TAG}_step_timeline_cfg$c.txt 2>&1
done
for c in 2 4; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$ctr
    timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$ctr -o pmc -- python $R/bench.py --config $c --steps 2 --warmup 0 --no-cpu-baseline --no-convergence --no-other-configs > $O/${TAG}_pmc_cfg${c}_$ctr.log 2>&1
    db=$(find /tmp/pmc_$ctr -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc.py $db > $O/${TAG}_cfg${c}_pmc_$ctr.txt 2>&1
  done
  (cd $R; python tools/make_pmc_traffic.py $O/${TAG}_cfg${c}_pmc_FETCH_SIZE.txt $O/${TAG}_cfg${c}_pmc_WRITE_SIZE.txt $O/${TAG}_cfg${c}_pmc_traffic.json ${TAG}_cfg${c} 2 > /dev/null)
done
rm -rf /tmp/pmc_m1 /tmp/pmc_m2
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/pmc_m1 -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence --no-other-configs > $O/${TAG}_pmc_mfma1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY -d /tmp/pmc_m2 -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence --no-other-configs > $O/${TAG}_pmc_mfma2.log 2>&1
: > $O/${TAG}_pmc_mfma.txt
for d in /tmp/pmc_m1 /tmp/pmc_m2; do db=$(find $d -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc_generic.py $db gemm_atb,ldlt_tail,ldlt_sparse,back_dataflow,fd_tasks,gf_form >> $O/${TAG}_pmc_mfma.txt 2>&1; done
cat $O/${TAG}_step_timeline_cfg2.txt | tail -12; head -8 $O/${TAG}_cfg2_pmc_FETCH_SIZE.txt $O/${TAG}_cfg2_pmc_WRITE_SIZE.txt; cat $O/${TAG}_pmc_mfma.txt
