cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_d; timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_d -o bench -- python $R/bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-convergence > $R/gpurun_out/exp_prof.log 2>&1
db=$(find /tmp/prof_d -name "*.db" | head -1)
python $R/tools/step_timeline.py $db 40 | tee $R/gpurun_out/exp_step_timeline_cfg4.txt
