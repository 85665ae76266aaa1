cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
L=/root/repo/tulip_amd/libtulip_hip
{
for i in 1 2 3; do
bash tools/ab_env.sh "TULIP_HIP_LIB=${L}_a1.so" "TULIP_HIP_LIB=${L}.so" 1
bash tools/ab_env.sh "TULIP_HIP_LIB=${L}_a3.so" "TULIP_HIP_LIB=${L}.so" 1
done
} > gpurun_out/ab/adam_nt.txt 2>&1
cat gpurun_out/ab/adam_nt.txt
