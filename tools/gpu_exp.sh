cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/gpu_exp.py 2>&1 | grep -v Warning | tail -40
