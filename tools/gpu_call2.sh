#!/bin/bash
# GPU call 2 of round 2: A/B of the finite-difference kernel variants, the whole GPU test suite, bench lines, kernel stats.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
: > $O/r02_call2_times.txt
T0=$(date +%s)
for v in old default fd43; do
  case $v in old) lib=$GRAFT_REPO_ROOT/tools/bin/libcba_fdold.so;; fd43) lib=$GRAFT_REPO_ROOT/tools/bin/libcba_fd43.so;; *) lib="";; esac
  for cfg in 2 4; do
    CBA_HIP_LIB=$lib timeout 300 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-convergence 2>/dev/null | tail -1 > $O/r02_ab_${v}_cfg$cfg.json
    python - <<PY >> $O/r02_ab_fd.txt
import json
try:
    d=json.load(open("$O/r02_ab_${v}_cfg$cfg.json")); st=d["stage_ms_per_step"]
    print("$v cfg$cfg ms/step %.2f  t_jac %.2f  fd %.2f  acc %.2f  cost %.2f  factor %.2f  gemm %.2f" % (d["ms_per_step"], st["t_jac"], st["t_fd_kernel"], st["t_accumulate"], st["t_cost"], st["t_factor"], st["t_schur_gemm"]))
except Exception as e:
    print("$v cfg$cfg FAILED", e)
PY
  done
done
echo "ab $(( $(date +%s) - T0 )) s" >> $O/r02_call2_times.txt
cat $O/r02_ab_fd.txt
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 > $O/r02_gputests.log 2>&1
echo "pytest rc=$? $(( $(date +%s) - T0 )) s" >> $O/r02_call2_times.txt
tail -15 $O/r02_gputests.log
T0=$(date +%s)
timeout 600 python bench.py > $O/r02_bench_cfg2.log 2>&1; tail -1 $O/r02_bench_cfg2.log > $O/r02_bench_cfg2.json
echo "bench rc=$? $(( $(date +%s) - T0 )) s" >> $O/r02_call2_times.txt
T0=$(date +%s)
cd /tmp
rm -rf /tmp/prof_r02; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r02 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-convergence > $GRAFT_REPO_ROOT/$O/r02_prof.log 2>&1
f=$(find /tmp/prof_r02 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$O/r02_kernel_stats.csv
db=$(find /tmp/prof_r02 -name "*.db" | head -1); [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $GRAFT_REPO_ROOT/$O/r02_kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence > $GRAFT_REPO_ROOT/$O/r02_pmc_$c.log 2>&1
  db=$(find /tmp/pmc_$c -name "*.db" | head -1); [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_pmc.py $db > $GRAFT_REPO_ROOT/$O/r02_pmc_$c.txt 2>&1
done
cd $GRAFT_REPO_ROOT
echo "prof $(( $(date +%s) - T0 )) s" >> $O/r02_call2_times.txt
cat $O/r02_call2_times.txt
