cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_deviations.json
timeout 900 python -m pytest tests/test_gpu_deterministic.py -x -q -m gpu > gpurun_out/exp_tests.log 2>&1; grep -E "passed|failed|error|assert" gpurun_out/exp_tests.log | tail -5
python - <<'PY'
import json
rows = json.load(open('gpurun_out/parity_deviations.json'))
rows = rows if isinstance(rows, list) else rows.get('rows', [])
for r in rows:
    s = json.dumps(r)
    if "per entry" in s: print(s[:300])
PY
