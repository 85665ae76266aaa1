"""CPU, world_size 2, gloo: the bucketed flat-gradient all-reduce used for N>1 GPUs (tulip_amd/ddp.py).
The RCCL path is the same code with backend "nccl"; 8-GPU runs are launched by the driver."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tulip_amd.ddp import GradBucketer, plan_buckets


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, groups, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(total, generator=g)
        mine = flat.clone()
        b = GradBucketer(groups, total, bucket_mb=0.002)
        assert b.world == world and len(b.buckets) >= 3
        sent = []
        # the backward fires the hooks in completion order; buckets go out as soon as they are complete
        for tag, _ in groups:
            before = len(b.pending)
            b.on_group_done(tag, flat)
            if len(b.pending) > before:
                sent.append(tag)
        b.wait_all()
        assert sent == [t for t, _, _ in b.buckets]
        other = torch.randn(total, generator=torch.Generator().manual_seed(100 + (1 - rank)))
        ok = torch.allclose(flat, mine + other, rtol=0, atol=1e-6)
        # mean (DDP semantics) = SUM * hyper[7]=1/world, applied inside the fused AdamW
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks_gloo():
    total = 5000
    groups = [("head", 300), ("dec1", 900), ("dec0", 1500), ("enc2", 3200), ("enc1", 4100), ("enc0", 4700),
              ("embed", 5000)]
    assert plan_buckets(groups, total, 500)[-1][2] == total
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), groups, total, out), nprocs=2, join=True)
    assert out[0] and out[1]


def test_single_process_is_a_no_op():
    b = GradBucketer([("head", 10), ("embed", 20)], 20, bucket_mb=0.00001)
    flat = torch.arange(20.0)
    b.on_group_done("head", flat)
    b.on_group_done("embed", flat)
    b.wait_all()
    assert torch.equal(flat, torch.arange(20.0)) and b.world == 1


# ---------------------------------------------------------------------------------------------- the sharded exchange (round 5)
def _sharded_worker(rank, world, port, buckets, replicated, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tulip_amd.ddp import ShardedExchange, shard_bounds
        gs = [torch.randn(total, generator=torch.Generator().manual_seed(300 + r)) for r in range(world)]
        gsum = gs[0] + gs[1]
        p0 = torch.randn(total, generator=torch.Generator().manual_seed(7))
        ref = p0 - 0.1 * gsum                                        # what the all-reduce plan's (toy) optimizer step gives every rank
        g, p = gs[rank].clone(), p0.clone()
        shadow = p0.bfloat16()
        ex = ShardedExchange(buckets, replicated, torch.device("cpu"))
        stepped = torch.zeros(total, dtype=torch.bool)
        for tag, a, b in buckets:                                    # (completion order)
            blocks = ex.reduce(tag, g)
            s_, lo, hi, me = shard_bounds(a, b, world, rank)
            assert torch.equal(g[lo:hi], gsum[lo:hi]) and torch.equal(g[me:b], gsum[me:b])
            for x, y in replicated:
                x, y = max(x, a), min(y, b)
                if x < y:
                    assert torch.equal(g[x:y], gsum[x:y]), (tag, x, y)
            idx = (blocks.long()[:, None] * 64 + torch.arange(64)[None, :]).reshape(-1)
            idx = idx[idx < total]
            assert int(blocks.min()) * 64 >= a and int(blocks.max()) * 64 < b and len(set(blocks.tolist())) == blocks.numel()
            p[idx] = p[idx] - 0.1 * g[idx]                           # the step, on what this rank owns / keeps replicated
            shadow[idx] = p[idx].bfloat16()
            stepped[idx] = True
            ex.gather_shadow(tag, shadow)
        ok_shadow = torch.equal(shadow, ref.bfloat16())              # every rank reads the same, complete bf16 weights
        rep_ok = all(torch.equal(p[x:y], ref[x:y]) for x, y in replicated)      # fp32-read parameters current everywhere
        stale = not torch.equal(p, ref)                               # ... the rest of the master only on its owner
        ex.gather_state(p)
        wb = ex.wire_bytes_per_step()
        out[rank] = (bool(ok_shadow), bool(rep_ok), bool(stale), bool(torch.equal(p, ref)), float(stepped.float().mean()),
                     wb["sharded"] / wb["allreduce_fp32"])
    finally:
        dist.destroy_process_group()


def test_sharded_exchange_two_ranks_gloo():
    """tulip_amd.ddp.ShardedExchange (optional plan, off by default): reduce-scatter -> step on the owned
    shard + the fp32-read (replicated) ranges -> all-gather of the bf16 shadow.  World size 2 over gloo, a toy optimizer step:
    both replicas end with the bf16 shadow of the all-reduce plan bit for bit, the replicated ranges' fp32 values too, each
    rank steps ~half of the buffer, gather_state() restores the whole master, and the ring-model wire bytes are ~3/4."""
    total = 64 * 300
    buckets = [("dec0", 0, 64 * 90), ("enc3", 64 * 90, 64 * 211), ("embed", 64 * 211, total)]       # 211 - 90 = 121 blocks: odd
    replicated = [(64 * 3, 64 * 4), (64 * 50, 64 * 52), (64 * 89, 64 * 92), (64 * 150, 64 * 151), (64 * 299, total)]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sharded_worker, args=(2, _free_port(), buckets, replicated, total, out), nprocs=2, join=True)
    for r in (0, 1):
        ok_shadow, rep_ok, stale, whole, frac, wire = out[r]
        assert ok_shadow and rep_ok and stale and whole, out[r]
        assert 0.45 <= frac <= 0.60, frac
        assert 0.74 <= wire <= 0.80, wire


def test_shard_bounds_cover_a_bucket_exactly():
    """tulip_amd.ddp.shard_bounds: `world` equal shards of whole 64-element blocks from the bucket's start, the rest (less than
    world x 64 elements) is the replicated remainder; every element belongs to exactly one rank's shard or to the remainder."""
    from tulip_amd.ddp import shard_bounds
    for (a, b) in [(0, 64 * 90), (64 * 90, 64 * 211), (64 * 7, 64 * 8), (64 * 5, 64 * 5 + 64 * 1000)]:
        for world in (1, 2, 3, 8):
            owned = []
            for r in range(world):
                s_, lo, hi, me = shard_bounds(a, b, world, r)
                assert s_ % 64 == 0 and hi - lo == s_ and lo == a + r * s_ and me == a + world * s_ and a <= me <= b
                assert b - me < world * 64
                owned.append((lo, hi))
            assert owned[0][0] == a and all(owned[i][1] == owned[i + 1][0] for i in range(world - 1)) and owned[-1][1] == me


# ---------------------------------------------------------------------------------------------- the exchange chooser
_BASE_BUCKETS = [("dec0", 0, 5_200_000), ("enc3", 5_200_000, 21_700_000), ("enc1", 21_700_000, 26_900_000),
                 ("embed", 26_900_000, 27_150_000)]       # tulip_base, bucket_mb = 16 (elements)


def test_comm_plan_chooser_branches():
    """tulip_amd.ddp.choose_comm_plan (what `bench.py --gpus N` runs between collective_smoke and the first capture): every
    branch -- fp32 kept when the measured bus bandwidth hides it, bf16 when fp32 would be exposed and bf16 is not, explicit
    requests win, the per-bucket optimizer follows the predicted tail -- and the ring model's arithmetic."""
    from tulip_amd.ddp import choose_comm_plan, predict_exposed_ms, close_fractions
    # ring model: one 64-MB bucket closing at 50 % of a 1-ms backward, 8 ranks, 128 GB/s -> 2*7/8*64e6/128e9 = 0.875 ms
    exp, per = predict_exposed_ms([64e6], [0.5], 1.0, 8, 128.0, 0.0)
    assert abs(per[0] - 0.875) < 1e-9 and abs(exp - 0.375) < 1e-9
    # collectives serialise on RCCL's stream: the second starts when the first ends, not when its bucket closes
    exp, per = predict_exposed_ms([64e6, 64e6], [0.1, 0.2], 1.0, 8, 128.0, 0.0)
    assert abs(exp - (0.1 + 2 * 0.875 - 1.0)) < 1e-9
    assert close_fractions(["dec0", "enc3", "enc1", "embed"]) == [0.35, 0.51, 0.77, 1.0]
    assert close_fractions(["a", "b"]) == [0.5, 1.0]
    fast = choose_comm_plan(_BASE_BUCKETS, 8, 320.0, 0.03)
    assert fast["chosen"]["grad_dtype"] == "fp32" and fast["chosen"]["predicted_exposed_exchange_ms"] < 0.1
    slow = choose_comm_plan(_BASE_BUCKETS, 8, 120.0, 0.03)
    assert slow["chosen"]["grad_dtype"] == "bf16"
    e32 = next(c for c in slow["candidates"] if c["grad_dtype"] == "fp32" and not c["bucket_adamw"])
    assert e32["predicted_exposed_exchange_ms"] > 0.1 + slow["chosen"]["predicted_exposed_exchange_ms"]
    forced = choose_comm_plan(_BASE_BUCKETS, 8, 120.0, 0.03, requested_dtype="fp32", requested_bucket_adamw=False)
    assert forced["chosen"]["grad_dtype"] == "fp32" and forced["chosen"]["bucket_adamw"] is False
    assert "requested" in forced["reason"]
    assert len(fast["candidates"]) == 4 and fast["model"]["world"] == 8
    # the per-bucket optimizer is chosen when it shortens the predicted tail (the single update waits for the last all-reduce)
    assert fast["chosen"]["bucket_adamw"] is True
    one = choose_comm_plan([("embed", 0, 1000)], 2, 100.0, 0.01)     # a single bucket: nothing to hide the update behind
    assert one["chosen"]["bucket_adamw"] is False
