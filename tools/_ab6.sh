cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
L=/root/repo/tulip_amd/libtulip_hip
{
for i in 1 2 3; do
for V in "$@"; do
bash tools/ab_env.sh "TULIP_HIP_LIB=${L}.so" "TULIP_HIP_LIB=${L}_$V.so" 1
done; done
} > gpurun_out/ab/multi.txt 2>&1
cat gpurun_out/ab/multi.txt
