cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
{
for i in 1 2; do
bash tools/ab_env.sh "TULIP_CARRY=dec2,dec1,dec0" "TULIP_CARRY=" 1
bash tools/ab_env.sh "TULIP_CARRY=dec2,dec1" "TULIP_CARRY=dec2" 1
bash tools/ab_env.sh "TULIP_CARRY=dec2,dec1,dec0 TULIP_CARRY_AT=1" "TULIP_CARRY=dec2,dec1,dec0 TULIP_CARRY_AT=3" 1
done
} > gpurun_out/ab/carry.txt 2>&1
cat gpurun_out/ab/carry.txt
