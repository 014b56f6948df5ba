#!/bin/bash
# round 5: FD patch-row padding sweep (built with CBA_BUILD_EXTRA_FLAGS=-DCBA_FD_PATCH_PAD=n before the call): cfg 4 and cfg 3 bench lines
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; TAG=${1:-r05i}
for c in 4 3 2; do
  steps=8; [ $c = 3 ] && steps=4; [ $c = 2 ] && steps=10
  timeout 600 python bench.py --config $c --steps $steps --warmup 2 --no-cpu-baseline --no-convergence --no-library > $O/${TAG}_bench_cfg$c.log 2>&1 || timeout 600 python bench.py --config $c --steps $steps --warmup 2 --no-cpu-baseline --no-convergence > $O/${TAG}_bench_cfg$c.log 2>&1
  tail -1 $O/${TAG}_bench_cfg$c.log > $O/${TAG}_bench_cfg$c.json
  python - $O/${TAG}_bench_cfg$c.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); st = d.get("stage_ms_per_step", {})
print(d["config"]["workload"][:50], "ms/step %.3f value %.3f" % (d["ms_per_step"], d["value"]), {k: round(v, 3) for k, v in st.items()})
PY
done
