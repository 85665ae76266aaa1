for r in 1 2 3 4; do
  for v in 0 1; do
    TULIP_PAIR96=$v python bench.py --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair96=$v', d['ms_per_step'], d['value'])"
  done
done
