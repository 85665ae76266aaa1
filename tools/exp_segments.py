"""dev experiment: step time when the step graph is cut into segments at the DDP bucket points (as with
world_size > 1) but without any collective: isolates the cost of segmentation + side-stream joins."""
import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd.trainer import Trainer
args = argparse.Namespace(model=os.environ.get("MODEL", "tulip_base"), img=[int(v) for v in os.environ.get("IMG", "16,1024").split(",")],
                          target=[int(v) for v in os.environ.get("TARGET", "64,1024").split(",")], batch=8)
dev = torch.device("cuda", 0)
for fake in (1, 2):
    model = bench.make_model(args).to(dev).train()
    tr = Trainer(model, 8, device=dev)
    tr.world = fake
    class _Done:
        def wait(self): pass
    tr.bucketer.on_group_done = (lambda tag, g, keep=True, b=tr.bucketer:
                                 (_Done(), *b.by_tag[tag]) if (fake > 1 and tag in b.by_tag) else None)
    tr.bucketer.wait_all = lambda: None
    if fake > 1 and os.environ.get("TULIP_BUCKET_ADAMW", "1") != "0":
        tr.bucket_adamw, tr._opt_stream = True, torch.cuda.Stream()

    lo, hi = bench.synthetic(args, 0, dev); tr.load_batch(lo, hi)
    for _ in range(10): tr.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): tr.step()
    torch.cuda.synchronize()
    print(f"fake world {fake}: segments {len(tr._segments[True])} step {(time.perf_counter() - t0) * 10:.3f} ms")
