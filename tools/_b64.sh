cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/kt8; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > $O/kt.log 2>&1
python tools/kstats.py $(find $O/kt -name "*.db" | head -1) 30 > $O/kernel_stats.txt
python tools/timeline.py $(find $O/kt -name "*.db" | head -1) > $O/timeline.txt
rm -rf $O/kt
head -24 $O/kernel_stats.txt; tail -8 $O/timeline.txt
python tools/exp_chain.py 2>&1 | tail -12
