cd "$GRAFT_REPO_ROOT"
TAILS=6144 REPS=2 timeout 300 tools/bin/bench_tail 2>&1 | grep -v "^   chain"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_factor_tail.py tests/test_grid_fit.py -x -q -m gpu 2>&1 | tail -3
