run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms_median'], d['step_ms_min'])"; }
run64() { echo "== B64 $*"; env "$@" python bench.py --batch 64 --steps 30 --warmup 15 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
run TULIP_WGRAD_TILES=1
run TULIP_WGRAD_TILES=1 TULIP_WGRAD_BIG_MINK=256
run TULIP_WGRAD_TILES=1 TULIP_WGRAD_BIG_MINK=1024
run64 TULIP_WGRAD_TILES=1
python -m pytest tests/test_model_gpu.py tests/test_round2_gpu.py tests/test_rccl_gpu.py -x -q 2>&1 | tail -3
