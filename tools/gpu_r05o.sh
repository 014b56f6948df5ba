#!/bin/bash
# round 5: Schur-product row order (greedy footprint chain, cba_set_observations) -- parity tests that go through the solve + bench lines
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; TAG=${1:-r05o}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_two_ranks.py tests/test_gpu_host_adapter.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|error" | tail -3
for c in ${CFGS:-2 4 3}; do
  steps=20; [ $c = 4 ] && steps=8; [ $c = 3 ] && steps=4
  timeout 600 python bench.py --config $c --steps $steps --warmup 2 --no-cpu-baseline --no-convergence > $O/${TAG}_bench_cfg$c.log 2>&1
  tail -1 $O/${TAG}_bench_cfg$c.log > $O/${TAG}_bench_cfg$c.json
  python - $O/${TAG}_bench_cfg$c.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); st = d.get("stage_ms_per_step", {})
print(d["config"]["workload"][:50], "ms/step %.3f value %.3f" % (d["ms_per_step"], d["value"]), {k: round(v, 3) for k, v in st.items()}, "flops/launch %.4e" % d["roofline"]["flops_per_launch"])
PY
done
