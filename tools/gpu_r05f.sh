#!/bin/bash
# round 5, call 6: pooled FD schedule: bit-identity vs one task per lane, parity, bench cfg 2 / 4 / 3
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_stragglers.py tests/test_gpu_parity.py tests/test_gpu_vs_ref_vectors.py tests/test_gpu_deterministic.py -q -m gpu -x --timeout 600 2>&1 | tail -8
for c in 2 4 3; do
  timeout 600 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline > $O/r05f_bench_cfg$c.log 2>&1; tail -1 $O/r05f_bench_cfg$c.log > $O/r05f_bench_cfg$c.json
done
python - <<'PY'
import json
for c in (2, 4, 3):
    try:
        d=json.load(open(f"gpurun_out/r05f_bench_cfg{c}.json")); print(c, round(d["ms_per_step"],3), round(d["value"],3), {k: round(v,3) for k,v in d["stage_ms_per_step"].items()}, round(d["roofline"]["frac"],3), (d.get("wall_clock_to_convergence") or {}).get("seconds"))
    except Exception as e: print(c, "FAILED", e)
PY
