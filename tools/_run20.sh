cd $GRAFT_REPO_ROOT
for r in 1 2 3 4; do for v in "TULIP_PACK_LAYOUT=a" "TULIP_PACK_LAYOUT=f" "TULIP_SPLIT_PACK=0"; do
  env $v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-reference-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v', 'b8', d['ms_per_step'], 'med', d['step_ms_median'], 'min', d['step_ms_min'], '| b64', d['secondary']['ms_per_step'], '| loss', d['final_loss'])"
done; done 2>&1 | tee gpurun_out/r5_ab_pack_layout3.txt
