#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh'): regenerates everything profiles/ cites.
# The PMC passes run first and their per-family summary is put where bench.py looks for it, so that the bench line of the
# same run carries `traffic` (bench.py only accepts a summary whose source stamp matches the code it runs).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-reference-loop > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-reference-loop > $O/pmc_w.log 2>&1
python tools/pmc_summary.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_hbm_traffic.txt
cp $O/pmc_traffic.json profiles/pmc_traffic.json
# MFMA utilisation from counters (its own pass; GRBM slots are independent of the SQ ones)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-reference-loop > $O/pmc_m.log 2>&1
python tools/mfma_summary.py $(find $O/pmc_m -name "*counter_collection.csv" | head -1) $O/pmc_mfma.json > $O/pmc_mfma.txt
cp $O/pmc_mfma.json profiles/pmc_mfma.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_f -- python tools/pmc_calib.py > $O/pmc_calibration.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_w -- python tools/pmc_calib.py > /dev/null 2>&1
python tools/pmc_summary.py $(find $O/cal_f -name "*counter_collection.csv" | head -1) $(find $O/cal_w -name "*counter_collection.csv" | head -1) >> $O/pmc_calibration.txt
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --no-reference-loop > $O/kt.log 2>&1
python tools/kstats.py $(find $O/kt -name "*.db" | head -1) 26 > $O/kernel_stats.txt
python tools/timeline.py $(find $O/kt -name "*.db" | head -1) > $O/timeline_summary.txt
python tools/timeline.py $(find $O/kt -name "*.db" | head -1) full > $O/timeline_full.txt
python tools/instep_summary.py $(find $O/kt -name "*.db" | head -1) $O/instep_durations.json > /dev/null
cp $O/instep_durations.json profiles/instep_durations.json
python tools/step_stamps.py 8 $O/step_stamps.json 2>/dev/null | grep -v amdgpu.ids > $O/step_stamps.txt
cp $O/step_stamps.json profiles/step_stamps.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats -d $O/kt64 -- python bench.py --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary --no-reference-loop > $O/kt64.log 2>&1
python tools/kstats.py $(find $O/kt64 -name "*.db" | head -1) 20 > $O/kernel_stats_b64.txt
python tools/bench_eval.py > $O/eval_bench.json 2>/dev/null
python tools/exp_chain.py > $O/exp_chain.txt 2>/dev/null
python tools/wgrad_phases.py > $O/wgrad_phases.txt 2>/dev/null
python tools/bench_wgrad.py > $O/wgrad_isolated.txt 2>/dev/null
python tools/chain_gemms.py > $O/chain_gemms.txt 2>/dev/null
python tools/bench_glue.py 8 2>/dev/null | grep -v amdgpu.ids > $O/bench_glue_b8.txt
python tools/bench_glue.py 64 2>/dev/null | grep -v amdgpu.ids > $O/bench_glue_b64.txt
python tools/exp_segments.py 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl\|socket.cpp\|UserWarning\|capture_end" > $O/exp_segments.txt
python tools/det_glue.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 > $O/det_glue.txt
bash tools/other_configs.sh > $O/other_configs.txt 2>/dev/null
python tools/exp_wgrad_contig.py 2>/dev/null | grep -v amdgpu.ids > $O/exp_wgrad_contig.txt
python tools/cold_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/cold_probe.txt
rm -rf $O/kt $O/kt64 $O/pmc_f $O/pmc_w $O/pmc_m $O/cal_f $O/cal_w
ls -la $O
