#!/bin/bash
# gpurun -- 'bash tools/trace_step.sh <tag> [extra env assignments...]': rocprofv3 kernel trace of the bench step -> per-kernel stats + timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
O=gpurun_out/trace_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --no-reference-loop > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/kstats.py $DB 26 > $O/kernel_stats.txt
python tools/timeline.py $DB full > $O/timeline_full.txt
python tools/instep_summary.py $DB $O/instep_durations.json > /dev/null
rm -rf $O/kt
tail -3 $O/kt.log
