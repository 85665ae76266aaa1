cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
python tools/exp_segments.py > gpurun_out/refresh/exp_segments.txt 2>/dev/null
python tools/refloop_timeline.py > gpurun_out/refresh/refloop_timeline.txt 2>/dev/null
python tools/deep_phases.py > gpurun_out/refresh/deep_phases.txt 2>/dev/null
python tools/bench_deep.py > gpurun_out/refresh/bench_deep.txt 2>/dev/null
( timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/refresh/gpu_tests.txt
tail -4 gpurun_out/refresh/gpu_tests.txt
python -c "
import json; d=json.load(open('gpurun_out/refresh/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['timing'][:30], d['secondary']['ms_per_step'], d['secondary_reference_loop']['ms_per_step'], d['config'].get('knobs_non_default'))"
