export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp
for cfg in 3 4; do
  rm -rf /tmp/prof_c$cfg; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c$cfg -o bench -- python $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-convergence > $O/r02_v4_prof_cfg$cfg.log 2>&1
  db=$(find /tmp/prof_c$cfg -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $O/r02_v4_bench_cfg${cfg}_kernel_stats.txt 2>&1
done
grep -E "k_accumulate|k_assemble|k_fd_|k_base|k_cell|k_strip|k_tangents|k_det|rocclr" $O/r02_v4_bench_cfg3_kernel_stats.txt | cut -c1-160
echo; grep -E "k_accumulate|k_assemble|k_fd_|k_base|k_cell|k_strip|rocclr" $O/r02_v4_bench_cfg4_kernel_stats.txt | cut -c1-160
