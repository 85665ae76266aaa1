cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
L=/root/repo/tulip_amd/libtulip_hip
V=${1:-p0}
{
for v in ; do :; done
for i in 1 2 3 4; do
bash tools/ab_env.sh "TULIP_HIP_LIB=${L}.so" "TULIP_HIP_LIB=${L}_$V.so" 1
done
} > gpurun_out/ab/prefetch_$V.txt 2>&1
cat gpurun_out/ab/prefetch_$V.txt
