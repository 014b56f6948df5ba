cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider > gpurun_out/exp_suite_$i.log 2>&1; echo "run $i rc=$?"; grep -E "passed|failed|error" gpurun_out/exp_suite_$i.log | tail -2
done
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-convergence 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 ms/step %.2f value %.2f' % (d['ms_per_step'], d['value']), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()})"
done
