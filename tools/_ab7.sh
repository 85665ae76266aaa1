cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
L=/root/repo/tulip_amd/libtulip_hip
{
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "unmerg or shuf or epilog or skip or gemm" 2>&1 | tail -5
for v in "TULIP_HIP_LIB=${L}.so" "TULIP_HIP_LIB=${L}_e0.so"; do echo "== $v"; env $v python tools/chain_gemms.py 2>/dev/null | grep -E " 5 | 8 |launches" ; done
for i in 1 2 3; do
bash tools/ab_env.sh "TULIP_HIP_LIB=${L}.so" "TULIP_HIP_LIB=${L}_e0.so" 1
done
} > gpurun_out/ab/shuf.txt 2>&1
cat gpurun_out/ab/shuf.txt
