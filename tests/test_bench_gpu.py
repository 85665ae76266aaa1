"""GPU: the bench.py contract the driver depends on -- `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with
the metric of BASELINE.json, the whole-job value, and the `roofline` / `cpu_baseline` objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "3",
                        "--no-secondary", "--cpu-steps", "1", "--cpu-warmup", "0", "--cpu-batch", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == base.get("unit", d["unit"]) and "range-images" in d["metric"]
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["config"]["step_structure"]["form"] == "one_graph"
    assert d["config"]["step_structure"]["optimizer"] == "in_weight_gradient_write_out"
    assert d["value"] > 0 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3      # whole-job img/s of batch 8
    assert d["step_ms_min"] <= d["step_ms_median"] <= 1.5 * d["ms_per_step"]
    ro = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro
    assert ro["bound"] in ("hbm", "mfma") and 0 < ro["frac"] < 1 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    assert ro["traffic"] is None or ro["traffic"] > 0
    # the dominant family leads; every family says where its duration comes from (the committed in-step trace when its
    # source stamp matches the kernels that ran, the live isolated measurement otherwise) and carries the live one
    assert len(ro["others"]) >= 5 and all("frac" in e and "timing" in e for e in ro["others"])
    assert ro["timing"].startswith(("in-step", "isolated (live)")) and ro["isolated"]["mean_launch_us"] > 0
    assert "wgrad_group_kernel" in ro["kernel"] or "gemm_kernel" in ro["kernel"]
    assert ro["algorithmic_bytes_per_launch"] > 0 and "mfma_util_counter" in ro
    assert "secondary_reference_loop" in d and d["secondary_reference_loop"]["grad_scale"] == 65536.0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1
