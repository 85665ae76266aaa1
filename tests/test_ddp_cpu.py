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
