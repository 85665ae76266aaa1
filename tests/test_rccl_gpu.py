"""SURVEY 8(e) / main_lidar_upsampling.py:277 + util/misc.py:279-284 on ONE GPU: the RCCL (`nccl`) code path of the
data-parallel step.  RCCL refuses two ranks on one device, so the multi-rank arithmetic is covered over gloo
(tests/test_ddp_gpu.py) and the RCCL-specific mechanics -- collectives issued between HIP-graph segment replays,
work.wait() as a stream dependency, capture beside the watchdog thread, the optimizer stream -- here with a one-rank
group and the segmentation forced on."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_segments_over_rccl_match_the_plain_step_bit_for_bit(tmp_path):
    out = tmp_path / "ws1.pt"
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_ws1_worker.py"), str(out), "4"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out)
    assert got["backend"] == "nccl"
    plain = got["plain"]
    assert not plain["segmented"] and plain["segments"] == 1
    # gradients exchanged in bf16 (Trainer(grad_dtype="bf16")): same structure, the update differs by the rounding of
    # the exchanged gradients only
    u_ref = plain["flat"] - got["init"]
    for name in ("segments_bf16", "segments_bf16_bucket_adamw", "captured_bf16"):
        g = got[name]
        u = g["flat"] - got["init"]
        rel = ((u - u_ref).norm() / u_ref.norm()).item()
        print(f"{name}: update-vector relative L2 difference to the fp32 exchange {rel:.3e}")
        assert g["segmented"] and 0 < rel <= 3e-2, (name, rel)
        assert (g["losses"][-1, 0] - plain["losses"][-1, 0]).abs().item() <= 2e-2 * plain["losses"][-1, 0].item()
    # the bucket's last side group as a graph of its own behind the segment (the default) / forked and joined inside it
    assert got["segments"]["detached"] >= 2 and got["segments_joined"]["detached"] == 0
    # "segments_sharded": Trainer(exchange="sharded") -- RCCL's reduce_scatter_tensor / all_gather_into_tensor between the segment
    # replays; with one rank both are the identity and the rank owns every shard, so the plain step's bits again
    # round 6: the collectives captured into ONE graph (Trainer.graph_collectives, the N > 1 default on RCCL)
    for name in ("captured", "captured_bucket_adamw", "captured_bf16"):
        assert got[name]["form"] == "one_graph_captured_collectives" and got[name]["segments"] == 1 and got[name]["detached"] == 0, got[name]["form"]
    assert got["segments"]["form"] == "segments"
    # a capture of the collectives that fails half-way leaves capture mode, and the step falls to the segmented rung: same bits
    fb = got["captured_fallback"]
    assert fb["form"] == "segments" and fb["segments"] == fb["buckets"] + 1, (fb["form"], fb["segments"])
    assert torch.equal(fb["losses"], plain["losses"]) and torch.equal(fb["flat"], plain["flat"])
    for name in ("segments", "segments_joined", "segments_bucket_adamw", "segments_sharded", "eager_segments", "captured",
                 "captured_bucket_adamw"):
        g = got[name]
        assert g["segmented"] and g["buckets"] >= 2
        if name in ("segments", "segments_joined"):
            assert g["segments"] == g["buckets"] + 1 and not g["bucket_adamw"]
        if name in ("segments_bucket_adamw", "segments_sharded"):
            assert g["segments"] == g["buckets"] and g["bucket_adamw"]
        # (the graphed trainers run one un-captured warm-up pass, which advances the DropPath counter: eager runs are
        # compared with an eager plain step)
        ref = got["eager_plain"] if name.startswith("eager") else plain
        assert torch.equal(g["losses"], ref["losses"]), name
        assert torch.equal(g["flat"], ref["flat"]), (name, g["diff"])


def test_one_graph_step_with_captured_collectives_at_the_bench_configuration(tmp_path):
    """Round 6: KITTI tulip_base, batch 8 -- fused wide / deep blocks and fused stage boundaries on the chain, every bucket's RCCL
    all-reduce a branch of the ONE captured graph behind the bucket's last side group (Trainer.graph_collectives), AdamW behind
    the last of them (or per bucket on the optimizer stream).  One rank: the all-reduces are the identity, so parameters, both
    moments and the losses equal the plain one-GPU step's -- whose optimizer step is taken in the weight-gradient write-outs --
    bit for bit (both paths run adamw_step4 on the same sums)."""
    out = tmp_path / "ws1_base.pt"
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_ws1_worker.py"), str(out), "3", "base"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out)
    assert got["backend"] == "nccl" and got["plain"]["form"] == "one_graph" and got["plain"]["glue"]
    for name in ("captured", "captured_bucket_adamw"):
        g = got[name]
        assert g["form"] == "one_graph_captured_collectives" and g["buckets"] >= 3 and g["glue"], (name, g["form"], g["buckets"])
        assert torch.equal(g["losses"], got["plain"]["losses"]), name
        assert torch.equal(g["flat"], got["plain"]["flat"]), (name, g["diff"])
        assert torch.equal(g["m"], got["plain"]["m"]) and torch.equal(g["v"], got["plain"]["v"]), name
