run() { env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | python -c "import sys,json
L=sys.stdin.readlines()
try:
    d=json.loads(L[-1]); print('$*', d['ms_per_step'])
except Exception as e: print('$*', 'FAILED')"; }
for s in 2 3 4 5 6 7 8 12; do
run TULIP_SIDE_STREAMS=$s TULIP_SIDE_MODE=item
done
run TULIP_SIDE_STREAMS=4
run TULIP_SIDE_STREAMS=4
run TULIP_SIDE_STREAMS=8
run TULIP_SIDE_STREAMS=12
run TULIP_SIDE_STREAMS=16
