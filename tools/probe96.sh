#!/bin/bash
# dev: the fused C = 96 kernels isolated (warm / L2 cold / all cold); usage: probe96.sh "8 64" "ENV=1 ENV=0"
for b in ${1:-8 64}; do for v in ${2:-TULIP_FC1_GRAD=1 TULIP_FC1_GRAD=0}; do
  echo "== batch $b $v"
  env $v python tools/cold_probe.py $b 2>/dev/null | grep -E "swin96"
done; done
