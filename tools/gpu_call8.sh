#!/bin/bash
# GPU call 8 of round 2: A/B of the unmasked stream for everything outside the factorisation (dev build with the switch).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
: > $O/r02_obs_stream_ab.txt
for v in masked obs masked obs; do
  for cfg in 2 4; do
    if [ $v = masked ]; then export CBA_NO_OBS_STREAM=1; else unset CBA_NO_OBS_STREAM; fi
    CBA_HIP_LIB=$R/tools/bin/libcba_dev.so timeout 300 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-convergence 2>/dev/null | tail -1 > $O/r02_ab8_${v}_cfg$cfg.json
    python - <<PY >> $O/r02_obs_stream_ab.txt
import json
try:
    d=json.load(open("$O/r02_ab8_${v}_cfg$cfg.json")); st=d["stage_ms_per_step"]
    print("$v cfg$cfg ms/step %.2f  t_jac %.2f  fd %.2f  acc %.2f  cost %.2f  factor %.2f  schur %.2f solve %.2f" % (d["ms_per_step"], st["t_jac"], st["t_fd_kernel"], st["t_accumulate"], st["t_cost"], st["t_factor"], st["t_schur_gemm"], st["t_solve"]))
except Exception as e:
    print("$v cfg$cfg FAILED", e)
PY
  done
done
unset CBA_NO_OBS_STREAM
cat $O/r02_obs_stream_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stragglers.py tests/test_gpu_deterministic.py tests/test_gpu_two_ranks.py tests/test_gpu_host_adapter.py -q --timeout 600 > $O/r02_gputests8.log 2>&1
echo "pytest (product lib, unmasked stream on) rc=$?"
grep -E "passed|failed|Error|FAILED|assert" $O/r02_gputests8.log | tail -20
