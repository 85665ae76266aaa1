cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
L=/root/repo/tulip_amd/libtulip_hip
{
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "embed" 2>&1 | tail -3
for v in "" _pf1 _fc1 _fc2; do echo "== lib$v"; TULIP_HIP_LIB=${L}$v.so python tools/cold_probe.py 8 2>/dev/null | grep -E "patch_embed" ; done
for i in 1 2 3; do
for v in "" _pf1 _fc1 _fc2; do
TULIP_HIP_LIB=${L}$v.so python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline --no-reference-loop --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('lib$v', 'b8', d['ms_per_step'], 'med', d['step_ms_median'], 'min', d['step_ms_min'], d['final_loss'])"
done; done
} > gpurun_out/ab/pe.txt 2>&1
cat gpurun_out/ab/pe.txt
