#!/bin/bash
# round 5: bench lines only (cfg 2 / 3 / 4), for one-constant A/B builds
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; TAG=${1:-r05r}
for c in ${CFGS:-2 3}; do
  steps=20; [ $c = 4 ] && steps=8; [ $c = 3 ] && steps=4
  timeout 600 python bench.py --config $c --steps $steps --warmup 2 --no-cpu-baseline --no-convergence > $O/${TAG}_bench_cfg$c.log 2>&1
  tail -1 $O/${TAG}_bench_cfg$c.log > $O/${TAG}_bench_cfg$c.json
  python - $O/${TAG}_bench_cfg$c.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); st = d.get("stage_ms_per_step", {})
print(d["config"]["workload"][:50], "ms/step %.3f value %.3f" % (d["ms_per_step"], d["value"]), {k: round(v, 3) for k, v in st.items()})
PY
done
