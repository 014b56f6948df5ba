cd "$GRAFT_REPO_ROOT"
MMA_ONLY=1 timeout 200 tools/bin/bench_tail 2>&1 | grep mma_only
