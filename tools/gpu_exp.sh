cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_two_ranks.py -x -q -m gpu --durations=8 > gpurun_out/exp_tests.log 2>&1; grep -E "passed|failed|error|Error|s call|assert" gpurun_out/exp_tests.log | tail -14
