"""What the N > 1 step STRUCTURE costs on one GPU: the step graph cut into segments at the DDP bucket points, side-stream
joins there, AdamW as its own segment behind the last bucket -- with a one-rank nccl group (Trainer(force_segments=True)),
once with the (identity) all-reduces issued and once with them skipped (GradBucketer.dry)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import torch.distributed as dist
import bench
from tulip_amd.trainer import Trainer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
args = argparse.Namespace(model="tulip_base", img=[16, 1024], target=[64, 1024], batch=int(os.environ.get("B", "8")))


def run(tag, **kw):
    model = bench.make_model(args).to(dev).train()
    tr = Trainer(model, args.batch, device=dev, **kw)
    for k, v in kw.pop("_attrs", {}).items():
        setattr(tr.eng, k, v)
    lo, hi = bench.synthetic(args, 0, dev)
    tr.load_batch(lo, hi)
    out = []
    for dry in (False, True):
        tr.set_dry(dry)
        for _ in range(10):
            tr.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            tr.step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 20)
    segs = len(tr._segments[True]) if tr._segments else 1
    print(f"{tag:58s} segments {segs} ({tr.step_form})  step {out[0]:.3f} ms   collectives skipped {out[1]:.3f} ms", flush=True)


run("one graph (world 1)")
run("N > 1 DEFAULT: one graph, captured collectives, 16 MB", force_segments=True)
run("one graph, captured collectives, per-bucket AdamW", force_segments=True, bucket_adamw=True)
run("one graph, captured collectives, bf16 buckets", force_segments=True, grad_dtype="bf16")
run("one graph, captured collectives, sharded exchange?", force_segments=True, exchange="sharded")
os.environ["TULIP_GRAPH_COLLECTIVES"] = "0"
for mb in (16.0, 1000.0):
    run(f"segmented (round 5), fp32 buckets >= {mb:g} MB", force_segments=True, bucket_mb=mb)
run("segmented (round 5), per-bucket AdamW", force_segments=True, bucket_adamw=True)
dist.destroy_process_group()
