#!/bin/bash
# round 5: step timeline of the non-central config 4 (which launches are on the critical path of the Jacobian pass)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $R/$O; TAG=${1:-r05m}; CFG=${2:-4}
cd /tmp
rm -rf /tmp/prof_t; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o bench -- python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-convergence > $R/$O/${TAG}_prof_cfg$CFG.log 2>&1
db=$(find /tmp/prof_t -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/step_timeline.py $db 15 > $R/$O/${TAG}_step_timeline_cfg$CFG.txt 2>&1
cat $R/$O/${TAG}_step_timeline_cfg$CFG.txt | cut -c1-150
