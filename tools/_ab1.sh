set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
NT="TULIP_HIP_LIB=/root/repo/tulip_amd/libtulip_hip_nt.so"
PL="TULIP_HIP_LIB=/root/repo/tulip_amd/libtulip_hip.so"
NTW="TULIP_HIP_LIB=/root/repo/tulip_amd/libtulip_hip_ntw.so"
for i in 1 2 3 4; do bash tools/ab_env.sh "$NTW" "$PL" 1; done > gpurun_out/ab/nt_step.txt 2>&1
cat gpurun_out/ab/nt_step.txt gpurun_out/ab/nt_probe.txt
