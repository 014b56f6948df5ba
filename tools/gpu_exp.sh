cd "$GRAFT_REPO_ROOT"
echo base; timeout 120 tools/bin/bench_gemm_base | tail -4; echo pipelined; timeout 120 tools/bin/bench_gemm_pipe | tail -4
