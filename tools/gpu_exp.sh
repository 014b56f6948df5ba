cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pmc_m; timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_m -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence > $R/gpurun_out/exp_pmc_mfma.log 2>&1
db=$(find /tmp/pmc_m -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc_generic.py $db gemm_atbILi128,ldlt_tail,back_dataflow,fd_tasks,Cijk | tee $R/gpurun_out/exp_pmc_mfma.txt
rm -rf /tmp/pmc_m2; timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_ANY -d /tmp/pmc_m2 -o pmc -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-convergence > $R/gpurun_out/exp_pmc_mfma2.log 2>&1
db=$(find /tmp/pmc_m2 -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocprof_pmc_generic.py $db gemm_atbILi128,ldlt_tail,Cijk | tee -a $R/gpurun_out/exp_pmc_mfma.txt
