#!/bin/bash
# dev: A/B of the C = 96 recomputation form (TULIP_SWIN96_RECOMPUTE=1/0) -- step time at batch 8 and 64, same box
mkdir -p gpurun_out
for v in 1 0 1 0; do
  TULIP_SWIN96_RECOMPUTE=$v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-reference-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('recompute=$v', 'b8 ms', d['ms_per_step'], 'median', d['step_ms_median'], 'min', d['step_ms_min'], '| b64 ms', d['secondary']['ms_per_step'], '| eval b8/b64', d['secondary_eval_forward']['batch8']['ms_per_forward'], d['secondary_eval_forward']['batch64']['ms_per_forward'])"
done
