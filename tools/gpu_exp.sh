cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_factor_tail.py tests/test_gpu_parity.py tests/test_gpu_two_ranks.py tests/test_gpu_parity_fullsize.py -x -q -m gpu > gpurun_out/exp_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/exp_tests.log | tail -3
for c in 3 4; do
  timeout 300 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-convergence 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg$c ms/step %.2f' % d['ms_per_step'], 'frac %.3f' % d['roofline_gemm']['frac'], {k: round(v,2) for k,v in d['stage_ms_per_step'].items()})"
done
