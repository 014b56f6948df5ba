#!/bin/bash
# round 5: quick A/B harness -- accumulation-related tests + bench lines of cfg 2 / 4 / 3 (short)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; TAG=${1:-r05j}
timeout 900 python -m pytest tests/test_gpu_vs_ref_vectors.py tests/test_gpu_deterministic.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "accum or system or jacobian_pass or H_and_b or normal" 2>&1 | tail -3
for c in 2 4 3; do
  steps=10; [ $c = 4 ] && steps=6; [ $c = 3 ] && steps=3
  timeout 600 python bench.py --config $c --steps $steps --warmup 2 --no-cpu-baseline --no-convergence > $O/${TAG}_bench_cfg$c.log 2>&1
  tail -1 $O/${TAG}_bench_cfg$c.log > $O/${TAG}_bench_cfg$c.json
  python - $O/${TAG}_bench_cfg$c.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); st = d.get("stage_ms_per_step", {})
print(d["config"]["workload"][:50], "ms/step %.3f value %.3f" % (d["ms_per_step"], d["value"]), {k: round(v, 3) for k, v in st.items()})
PY
done
