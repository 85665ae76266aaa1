cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --no-reference-loop > $O/kt.log 2>&1
python tools/timeline.py $(find $O/kt -name "*.db" | head -1) full > $O/timeline_full.txt
python tools/kstats.py $(find $O/kt -name "*.db" | head -1) 26 > $O/kernel_stats.txt
python tools/step_stamps.py 2>/dev/null | grep -v amdgpu.ids > $O/step_stamps.txt
rm -rf $O/kt
