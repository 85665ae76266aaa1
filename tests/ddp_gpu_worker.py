"""Worker of tests/test_ddp_gpu.py: one rank of a 2-process data-parallel run that shares ONE GPU.
RCCL refuses two ranks on the same device, so the collective goes through gloo (which accepts device tensors);
everything else -- graph segments cut at the bucket points, side-stream joins, 1/world folded into AdamW -- is the
code the 8-GPU run uses."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, use_graph, steps = sys.argv[1], sys.argv[2] == "1", int(sys.argv[3])
    accum = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    diffseed = len(sys.argv) > 5 and sys.argv[5] == "diffseed"
    exchange = sys.argv[6] if len(sys.argv) > 6 else "allreduce"
    from oracle import tulip_oracle as O
    from tests.test_model_gpu import build
    from tulip_amd.trainer import Trainer
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = O.tiny_config(drop_path_rate=0.0)
    # diffseed: every rank starts from its own weights, as the reference's per-rank seeds do (main_lidar_upsampling.py:155);
    # the Trainer must broadcast rank 0's, like DistributedDataParallel's constructor (main:277)
    sd = O.key_seeded_state_dict(cfg, seed=3 + (rank if diffseed else 0))
    lo, hi = O.synthetic_batch(cfg, 2 * world, seed=77)
    m = build(cfg, sd, train=True)
    tr = Trainer(m, 2, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01, use_graph=use_graph, bucket_mb=0.05,
                 accum_iter=accum, exchange=exchange)
    assert tr.world == world and len(tr.bucketer.buckets) >= 2
    start = tr.eng.params.flat.clone()
    first = [torch.empty_like(start) for _ in range(world)]
    dist.all_gather(first, start)
    same_start = all(torch.equal(first[0], f) for f in first)
    tr.load_batch(lo[2 * rank:2 * rank + 2].cuda(), hi[2 * rank:2 * rank + 2].cuda())
    losses = [tr.step().clone() for _ in range(steps * accum)]
    torch.cuda.synchronize()
    extra = {}
    if exchange == "sharded":
        W = tr.eng.params
        sh = W.shadow.clone()
        shs = [torch.empty_like(sh) for _ in range(world)]
        dist.all_gather(shs, sh)
        stale = tr.eng.params.flat.clone()
        rep = sum(W.numel[n] for n in W.fp32_read) / W.total
        tr.gather_state()
        torch.cuda.synchronize()
        extra = {"shadow_same": all(torch.equal(shs[0], x) for x in shs), "shadow": sh.cpu(), "replicated_fraction": rep,
                 "master_was_partial": not torch.equal(stale, tr.eng.params.flat), "m": tr.m.cpu(), "v": tr.v.cpu(),
                 "wire": tr._sharded.wire_bytes_per_step()}
    else:
        extra = {"shadow": tr.eng.params.shadow.cpu(), "m": tr.m.cpu(), "v": tr.v.cpu()}
    flat = tr.eng.params.flat.clone()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save({"flat": flat.cpu(), "same_on_all_ranks": all(torch.equal(gathered[0], g) for g in gathered),
                    "losses": torch.stack(losses).cpu(), "same_start": same_start, "segments": len(tr._segments[True]) if use_graph else 0, "bucket_adamw": tr.bucket_adamw,
                    "buckets": tr.bucketer.buckets, **extra}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
