"""SURVEY 8(e) on the GPU box: the N>1 training path (graph segments cut at the DDP bucket points, bucketed
all-reduce of the flat gradient buffer between segment replays, side-stream joins, mean folded into AdamW) run as
two ranks sharing the one GPU over gloo, against a single-process run on the concatenated batch.
DDP semantics: mean of the two ranks' gradients = gradient of the mean loss over the 4 images."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from oracle import tulip_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("use_graph,accum,bucket_adamw,diffseed", [(True, 1, False, True), (False, 1, False, False),
                                                                  (True, 2, False, False), (True, 1, True, False)])
def test_two_ranks_match_single_process_on_the_full_batch(tmp_path, use_graph, accum, bucket_adamw, diffseed):
    from tests.test_model_gpu import build
    from tulip_amd.trainer import Trainer
    steps = 3
    out = tmp_path / "r0.pt"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_gpu_worker.py"), str(out),
           "1" if use_graph else "0", str(steps), str(accum), "diffseed" if diffseed else "same"]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", TULIP_BUCKET_ADAMW="1" if bucket_adamw else "0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out)
    assert got["same_start"]                                     # rank 0's parameters were broadcast at construction
    assert got["same_on_all_ranks"]                              # replicas stay bit-identical
    assert got["bucket_adamw"] == bucket_adamw
    if use_graph:
        # one graph segment per bucket, then one AdamW segment behind the last all-reduce (the default), or the
        # optimizer per bucket behind its own all-reduce (TULIP_BUCKET_ADAMW=1: no AdamW segment)
        assert got["segments"] == len(got["buckets"]) + (0 if bucket_adamw else 1)
    cfg = O.tiny_config(drop_path_rate=0.0)
    sd = O.key_seeded_state_dict(cfg, seed=3)
    lo, hi = O.synthetic_batch(cfg, 4, seed=77)
    m = build(cfg, sd, train=True)
    tr = Trainer(m, 4, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01, use_graph=use_graph, accum_iter=accum)
    tr.load_batch(lo.cuda(), hi.cuda())
    losses = [tr.step().clone() for _ in range(steps * accum)]     # accum micro-steps per optimizer step
    assert tr.t == steps
    torch.cuda.synchronize()
    ref = tr.eng.params.flat.cpu()
    # rank 0 saw images 0-1: its loss differs from the full-batch loss, but the parameter UPDATES must agree.
    # Summation orders differ (2+2 images reduced across ranks vs 4 images in one pass), and AdamW turns a
    # gradient element that is pure rounding noise into a +-lr step, so the comparison is on the update vector:
    # relative L2 <= 1 %, no element further apart than the 2*lr*steps a sign flip can cost.
    init = Trainer(build(cfg, sd, train=True), 4, use_graph=False).eng.params.flat.cpu()
    u_ddp, u_ref = got["flat"] - init, ref - init
    assert u_ref.abs().max().item() > 1e-4                        # the parameters actually moved
    rel = ((u_ddp - u_ref).norm() / u_ref.norm()).item()
    print(f"update-vector relative L2 difference: {rel:.4e}")
    assert rel <= 1e-2, rel
    assert (u_ddp - u_ref).abs().max().item() <= 2 * 5e-4 * steps + 1e-6
    assert ((u_ddp - u_ref).abs() <= 2e-5).float().mean().item() >= 0.995
    full = torch.stack(losses).cpu()[:, 0]
    assert abs(got["losses"][-1, 0].item() - full[-1].item()) < 0.2 * full[-1].item()


@pytest.mark.parametrize("use_graph", [True, False])
def test_sharded_exchange_equals_the_per_bucket_allreduce_plan(tmp_path, use_graph):
    """Trainer(exchange="sharded") (round 5, optional plan: reduce-scatter -> AdamW on the owned shard and on the fp32-read
    parameters -> all-gather of the bf16 shadow) against the per-bucket all-reduce plan, two ranks sharing the GPU over gloo, three
    steps: with two ranks both plans add the same two gradients, so the bf16 shadows the kernels read, and after gather_state()
    the fp32 master and both moments, must be identical bit for bit; the replicas' shadows agree with each other before any
    gather; a small fraction of the parameters stays replicated; each rank's master was indeed partial before the gather."""
    res = {}
    for exchange in ("allreduce", "sharded"):
        out = tmp_path / f"{exchange}.pt"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_gpu_worker.py"), str(out),
               "1" if use_graph else "0", "3", "1", "same", exchange]
        env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", TULIP_BUCKET_ADAMW="1")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        res[exchange] = torch.load(out)
    a, s = res["allreduce"], res["sharded"]
    assert s["same_on_all_ranks"] and s["shadow_same"] and s["master_was_partial"] and s["bucket_adamw"]
    assert 0 < s["replicated_fraction"] < 0.10, s["replicated_fraction"]      # (the tiny test model; ~1 % for tulip_base)
    w = s["wire"]
    print(f"replicated fraction {s['replicated_fraction']:.4f}; wire bytes sharded / all-reduce {w['sharded'] / w['allreduce_fp32']:.3f}")
    assert torch.equal(s["shadow"], a["shadow"])
    assert torch.equal(s["flat"], a["flat"]) and torch.equal(s["m"], a["m"]) and torch.equal(s["v"], a["v"])
    assert (a["losses"] - s["losses"]).abs().max().item() == 0.0
