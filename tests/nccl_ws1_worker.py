"""Worker of tests/test_rccl_gpu.py: ONE rank on the `nccl` backend (= RCCL on ROCm) with the N>1 step structure
forced on (Trainer(force_segments=True)): graph segments cut at the bucket points, one real RCCL all-reduce per bucket
issued between the replays, stream-ordered work.wait(), thread-local graph capture beside the RCCL watchdog thread, the
optimizer either as its own segment behind the last all-reduce or per bucket on the optimizer stream.  A one-rank
all-reduce is the identity, so the result must equal the plain single-process step bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, steps = sys.argv[1], int(sys.argv[2])
    from oracle import tulip_oracle as O
    from tests.test_model_gpu import build
    from tulip_amd.trainer import Trainer
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    base = len(sys.argv) > 3 and sys.argv[3] == "base"
    if base:
        # the bench configuration (KITTI tulip_base, batch 8): the fused wide / deep blocks, the fused stage boundaries (csrc/glue.hip)
        # and the optimizer step in the write-outs on the plain side; the one-graph N > 1 step with its collectives as branches on the other
        cfg = O.tulip_base_config()
        sd = O.key_seeded_state_dict(cfg, seed=3)
        lo, hi = O.synthetic_batch(cfg, 8, seed=77)
        res = {}
        for name, kw in [("plain", dict()), ("captured", dict(force_segments=True)),
                         ("captured_bucket_adamw", dict(force_segments=True, bucket_adamw=True))]:
            torch.manual_seed(11)
            m = build(cfg, sd, train=True)
            os.environ["TULIP_GRAPH_COLLECTIVES"] = "1"
            tr = Trainer(m, 8, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01, **kw)
            tr.load_batch(lo.cuda(), hi.cuda())
            losses = [tr.step().clone() for _ in range(steps)]
            torch.cuda.synchronize()
            res[name] = {"flat": tr.eng.params.flat.cpu(), "m": tr.m.cpu(), "v": tr.v.cpu(), "losses": torch.stack(losses).cpu(),
                         "form": tr.step_form, "buckets": len(tr.bucketer.buckets), "glue": bool(tr.eng.fuse_glue and 0 in tr.eng.params.pk_active)}
        from tests.conftest import describe_flat_diff
        for name in ("captured", "captured_bucket_adamw"):
            res[name]["diff"] = describe_flat_diff(tr.eng, res[name]["flat"], res["plain"]["flat"])
        res["backend"] = dist.get_backend()
        torch.save(res, out_path)
        dist.destroy_process_group()
        return
    cfg = O.tiny_config()                                   # DropPath on: the counter-based draws replay identically
    sd = O.key_seeded_state_dict(cfg, seed=3)
    lo, hi = O.synthetic_batch(cfg, 4, seed=77)
    res = {"init": Trainer(build(cfg, sd, train=True), 4, use_graph=False).eng.params.flat.cpu()}
    for name, kw in [("plain", dict()), ("segments", dict(force_segments=True, bucket_mb=0.05)),
                     ("segments_joined", dict(force_segments=True, bucket_mb=0.05)),     # TULIP_DETACH_BUCKETS=0, see below
                     ("segments_bucket_adamw", dict(force_segments=True, bucket_mb=0.05, bucket_adamw=True)),
                     ("segments_bf16", dict(force_segments=True, bucket_mb=0.05, grad_dtype="bf16")),
                     ("segments_bf16_bucket_adamw", dict(force_segments=True, bucket_mb=0.05, grad_dtype="bf16",
                                                         bucket_adamw=True)),
                     ("segments_sharded", dict(force_segments=True, bucket_mb=0.05, exchange="sharded")),
                     # round 6, the N > 1 default: ONE graph, every bucket's all-reduce captured as a branch off the side queue
                     ("captured", dict(force_segments=True, bucket_mb=0.05)),
                     ("captured_bucket_adamw", dict(force_segments=True, bucket_mb=0.05, bucket_adamw=True)),
                     ("captured_bf16", dict(force_segments=True, bucket_mb=0.05, grad_dtype="bf16")),
                     # a backend that cannot capture its collective: the first one raises inside the capture -> the segmented rung
                     ("captured_fallback", dict(force_segments=True, bucket_mb=0.05)),
                     ("eager_plain", dict(use_graph=False)),
                     ("eager_segments", dict(force_segments=True, bucket_mb=0.05, use_graph=False))]:
        torch.manual_seed(11)
        m = build(cfg, sd, train=True)
        # "segments": the last side group of a bucket is a graph of its own behind the segment (Trainer.detach_buckets, the default);
        # "segments_joined": forked inside the segment and joined at the cut
        os.environ["TULIP_DETACH_BUCKETS"] = "0" if name == "segments_joined" else "1"
        os.environ["TULIP_GRAPH_COLLECTIVES"] = "1" if name.startswith("captured") else "0"
        tr = Trainer(m, 4, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01, **kw)
        if name == "captured_fallback":
            real = tr.bucketer.on_group_done

            def refusing(tag, gflat, keep=True, _real=real):
                if tr.graph_collectives and torch.cuda.is_current_stream_capturing() and tag in tr.bucketer.by_tag:
                    raise RuntimeError("this backend cannot capture its collectives (test)")
                return _real(tag, gflat, keep=keep)
            tr.bucketer.on_group_done = refusing
        tr.load_batch(lo.cuda(), hi.cuda())
        losses = [tr.step().clone() for _ in range(steps)]
        tr.gather_state()              # (exchange="sharded": the master whole again; a no-op otherwise)
        torch.cuda.synchronize()
        res[name] = {"flat": tr.eng.params.flat.cpu(), "losses": torch.stack(losses).cpu(),
                     "segments": len(tr._segments[True]) if tr.use_graph else 0, "buckets": len(tr.bucketer.buckets),
                     "segmented": tr.segmented, "bucket_adamw": tr.bucket_adamw, "detached": len(tr._det_graphs),
                     "form": tr.step_form}
    from tests.conftest import describe_flat_diff
    for name in [k for k in res if k != "init"]:
        ref = res["eager_plain"] if name.startswith("eager") else res["plain"]
        res[name]["diff"] = describe_flat_diff(tr.eng, res[name]["flat"], ref["flat"])
    res["backend"] = dist.get_backend()
    torch.save(res, out_path)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
