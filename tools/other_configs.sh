#!/bin/bash
# The training step of the other BASELINE configurations on one GPU (same bench.py; value, ms/step, metric).
for cfg in "--model tulip_large --img 32 2048 --target 128 2048 --batch 4" "--model tulip_large --img 32 2048 --target 128 2048 --batch 8" \
           "--model tulip_base --img 16 2048 --target 64 2048 --batch 8" "--model tulip_large --img 16 2048 --target 64 2048 --batch 8"; do
  echo "== $cfg"
  python bench.py $cfg --no-cpu-baseline --no-roofline --no-secondary --steps 50 --warmup 15 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s ', d['ms_per_step'], 'ms/step ', d['metric'])"
done
