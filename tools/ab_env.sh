#!/bin/bash
# dev: same-box A/B of the step under environment settings; usage: ab_env.sh "ENV=1" "ENV=0" [rounds]
A="$1"; B="$2"; N=${3:-2}
for i in $(seq $N); do for v in "$A" "$B"; do
  env $v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-reference-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v', 'b8', d['ms_per_step'], 'med', d['step_ms_median'], 'min', d['step_ms_min'], '| b64', d['secondary']['ms_per_step'], '| loss', d['final_loss'])"
done; done
