"""dev tool: what the edges of the step cost -- the per-step hyper-parameter upload between graph replays and the
fragment-major weight copies behind AdamW (results are wrong without them; timing only)."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd.trainer import Trainer
from tulip_amd.engine import FlatParams

a = argparse.Namespace(model="tulip_base", img=[16, 1024], target=[64, 1024], batch=8)
dev = torch.device("cuda", 0)

def run(tag, patch):
    torch.manual_seed(0)
    m = bench.make_model(a).to(dev).train()
    tr = Trainer(m, 8, device=dev)
    lo, hi = bench.synthetic(a, 0, dev); tr.load_batch(lo, hi)
    for _ in range(5): tr.step()
    undo = patch(tr)
    for _ in range(20): tr.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): tr.step()
    torch.cuda.synchronize()
    print(f"{tag:60s} {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step", flush=True)
    if undo: undo()

def no_hyper(tr):
    tr._set_hyper = lambda: None

def no_pack(tr):
    real = FlatParams.refresh_transposes
    FlatParams.refresh_transposes = lambda self: None
    tr._segments = None                       # re-capture without the pack launches
    return lambda: setattr(FlatParams, "refresh_transposes", real)

run("full step", lambda tr: None)
run("no hyper-parameter upload between replays", no_hyper)
run("no fragment-major weight copies after AdamW", no_pack)
