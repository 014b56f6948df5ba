cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_two_ranks.py -x -q -m gpu > gpurun_out/exp_tests.log 2>&1; grep -E "passed|failed|error|Error" gpurun_out/exp_tests.log | tail -8
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms/step %.2f' % d['ms_per_step'], d['config']['parallelism'], {k: round(v,2) for k,v in d['stage_ms_per_step'].items()})"; }

timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-convergence --force-allreduce --distributed-solve 1 2>gpurun_out/exp_b3.err | tail -1 | show forced-distributed
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_d; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-convergence --force-allreduce --distributed-solve 1 > $R/gpurun_out/exp_prof.log 2>&1
db=$(find /tmp/prof_d -name "*.db" | head -1)
python - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else kcols[1]
rows = cur.execute(f"select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_back_dataflow" in r[0]]
end = idx[-1]; beg = idx[-2] + 1
t0 = None
for name, st, en in rows[beg:end + 1]:
    short = name.split("(")[0][:60]
    if "dinv_times" in short: t0 = st
    if t0 is None: continue
    if any(k in short for k in ("gemm", "ldlt_tail", "dist_copy", "ccl", "pack", "back_dataflow")) or (en - st) > 100000:
        print("%9.3f ms  +%8.3f ms  %s" % ((st - t0) / 1e6, (en - st) / 1e6, short))
PY
