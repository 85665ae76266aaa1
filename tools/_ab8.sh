cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
{
for i in 1 2; do
for v in "TULIP_EARLY_FLUSH_BLOCKS=0" "TULIP_EARLY_FLUSH_BLOCKS=1" "TULIP_EARLY_FLUSH_BLOCKS=2" "TULIP_WGRAD_CTAS_MAP=100:128;20:128;5:128" "TULIP_WGRAD_CTAS_MAP=100:96;20:112;5:112" "TULIP_MERGE_EMBED_FOLD=1"; do
env "$v" python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline --no-reference-loop --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v', 'b8', d['ms_per_step'], 'med', d['step_ms_median'], 'min', d['step_ms_min'])"
done; done
} > gpurun_out/ab/sweep.txt 2>&1
cat gpurun_out/ab/sweep.txt
