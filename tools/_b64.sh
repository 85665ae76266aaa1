cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/b64; rm -rf $O; mkdir -p $O
python tools/chain_gemms.py --batch 64 > $O/chain_gemms_b64.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/kt.log 2>&1
python tools/kstats.py $(find $O/kt -name "*.db" | head -1) 30 > $O/kernel_stats_b64.txt
python tools/timeline.py $(find $O/kt -name "*.db" | head -1) > $O/timeline_b64.txt
rm -rf $O/kt
cat $O/chain_gemms_b64.txt | tail -45; head -32 $O/kernel_stats_b64.txt; tail -12 $O/timeline_b64.txt
