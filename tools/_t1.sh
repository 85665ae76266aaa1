cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
{
timeout 600 python -m pytest tests/test_swinw_gpu.py -x -q -m gpu 2>&1 | tail -15
for v in TULIP_SWINW_SPLIT=1 "TULIP_SWINW_SPLIT=1 TULIP_SWINW_SPLIT_BWD=0"; do echo "== $v"; env $v timeout 300 python tools/cold_probe.py 8 2>/dev/null | grep -E "swinw_block_... C=384|warm" ; done
for i in 1 2 3; do
timeout 600 bash tools/ab_env.sh "TULIP_SWINW_SPLIT=1" "TULIP_SWINW_SPLIT=1 TULIP_SWINW_SPLIT_BWD=0" 1
done
} > gpurun_out/ab/split.txt 2>&1
cat gpurun_out/ab/split.txt
