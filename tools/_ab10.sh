cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
L=/root/repo/tulip_amd/libtulip_hip
{
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_round4_gpu.py -x -q -m gpu -k "swin96 or block96 or recomput or fused_block" 2>&1 | tail -3
for v in "" _base; do echo "== lib$v"; TULIP_HIP_LIB=${L}$v.so python tools/cold_probe.py 8 2>/dev/null | grep -E "swin96" ; TULIP_HIP_LIB=${L}$v.so python tools/cold_probe.py 64 2>/dev/null | grep -E "swin96" ; done
for i in 1 2 3; do
bash tools/ab_env.sh "TULIP_HIP_LIB=${L}.so" "TULIP_HIP_LIB=${L}_base.so" 1
done
} > gpurun_out/ab/opaque.txt 2>&1
cat gpurun_out/ab/opaque.txt
