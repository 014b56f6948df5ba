cd "$GRAFT_REPO_ROOT"
echo "--- small sizes (default super 2048)"; TAILS=1024,6144 timeout 300 tools/bin/bench_tail 2304 2240 2>&1 | grep -v "^   chain"
for cus in 8 0; do for sw in 0 1024 2048 3072; do
echo "--- CBA_PANEL_CUS=$cus SUPER_W=$sw"
CBA_SUPER_W=$sw CBA_PANEL_CUS=$cus REPS=4 TAILS=4096,6144,8192 timeout 300 tools/bin/bench_tail 12672 12544 2>&1 | grep -v "^   chain\|^==="
done; done
