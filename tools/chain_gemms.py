"""Every GEMM launch of one training step on the dependent chain (the grouped weight gradients are not in this list):
shape, layout, epilogue, splits and its isolated duration (mean of 20 back-to-back launches)."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tulip_amd import ops
from tulip_amd.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
a.model, a.img, a.target = "tulip_base", [16, 1024], [64, 1024]
dev = torch.device("cuda:0")
m = bench.make_model(a).to(dev).train()
tr = Trainer(m, a.batch, use_graph=False)
lo, hi = bench.synthetic(a, 0, dev)
tr.load_batch(lo, hi)
tr.step()
rec = []
real = ops.gemm
def gemm(A, B, M, N, K, **kw):
    rec.append((M, N, K, kw, lambda: real(A, B, M, N, K, **kw)))
    real(A, B, M, N, K, **kw)
ops.gemm = gemm
try:
    tr._fwd_bwd(lambda tag: None)
finally:
    ops.gemm = real
torch.cuda.synchronize()
rows = collections.OrderedDict()
for M, N, K, kw, call in rec:
    call()
    torch.cuda.synchronize()
    # the launches are replayed from a HIP graph like the step's: eager ctypes launches are host-bound at ~10 us each
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(a.reps):
            call()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    key = (M, N, K, bool(kw.get("a_trans")), bool(kw.get("b_trans")), kw.get("epi", 0), kw.get("splits", 1))
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1; r[1] += us
tot = 0.0
print(f"{'M':>6} {'N':>5} {'K':>5} aT bT epi spl  calls   us/call  TFLOP/s  tiles")
for (M, N, K, at, bt, epi, spl), (n, us) in rows.items():
    tot += us
    tiles = -(-M // 64) * -(-N // 96) * ops.gemm_effective_splits(K, spl)
    print(f"{M:6d} {N:5d} {K:5d} {int(at):2d} {int(bt):2d} {epi:3d} {spl:3d} {n:6d} {us / n:9.2f} {2.0 * M * N * K / (us / n) / 1e6:8.1f} {tiles:6d}")
print(f"{len(rec)} launches, {tot:.1f} us per step (isolated)")
