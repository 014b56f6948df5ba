#!/bin/bash
# GPU box: kernel-trace timelines of one cfg-2 step for the grid-first order with 1 and with the automatic number of strips,
# and for the pose-first order.   tools/gpu_r06_timeline.sh [tag] [config]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r06}; CFG=${2:-2}
cd /tmp
for variant in "gf1 --elimination 2 --grid-strips 1" "gfauto --elimination 2" "pose --elimination 1"; do
  set -- $variant; name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o bench -- python $R/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --no-convergence "$@" > $O/${TAG}_prof_cfg${CFG}_$name.log 2>&1
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/step_timeline.py $db 15 > $O/${TAG}_step_timeline_cfg${CFG}_$name.txt 2>&1
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $O/${TAG}_kernel_stats_cfg${CFG}_$name.txt 2>&1
  echo "== $name"; tail -1 $O/${TAG}_prof_cfg${CFG}_$name.log | cut -c1-300; cat $O/${TAG}_step_timeline_cfg${CFG}_$name.txt
done
