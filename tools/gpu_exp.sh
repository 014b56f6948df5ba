cd "$GRAFT_REPO_ROOT"
timeout 120 tools/bin/bench_gemm2; timeout 120 tools/bin/bench_gemm3
