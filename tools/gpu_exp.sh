cd "$GRAFT_REPO_ROOT"
for r in 0 1; do echo "== REG2=$r"; REG2=$r TAILS=1024,6144 REPS=3 timeout 300 tools/bin/bench_tail 2>&1 | grep -E "^tail|^===|status"; done
echo "== chain timeline REG2=1"; REG2=1 TAILLOG=1 TAILS=6144 REPS=1 timeout 200 tools/bin/bench_tail 12672 12544 2>&1 | grep -E "per block|chain" | head -6
