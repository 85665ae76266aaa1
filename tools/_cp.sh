cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
python tools/cold_probe.py 8 2>/dev/null | grep -E "wgrad|reduce_rows|warm" > gpurun_out/ab/cold_wgrad.txt
cat gpurun_out/ab/cold_wgrad.txt
