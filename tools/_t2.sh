cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
{
timeout 900 python -m pytest tests/test_round4_gpu.py -x -q -m gpu -k "split" 2>&1 | tail -15
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
} > gpurun_out/ab/tests.txt 2>&1
cat gpurun_out/ab/tests.txt
