"""The grouped weight-gradient launch of one Swin block per stage, isolated: large tiles (mode 1) against the 64 x 96 tile
(mode 0), with the engine's split policy for each, GEMM alone and GEMM + fold."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd import ops
from tulip_amd.engine import TulipEngine as Engine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[8, 64])
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--modes", type=lambda x: int(x, 0), nargs="+", default=[0, 1])
a = ap.parse_args()
dev = torch.device("cuda:0")
ws = torch.empty((Engine.WS_ELEMS + (1 << 20)), device=dev)

SMALL = False      # TULIP_WGRAD_SMALL_TILES on the launches below (mode 0)

def run(items, fold):
    ops.wgrad_group(items, [], ws, ws.numel() * 4, fold=fold, small_tiles=SMALL)

def timeit(items, fold, reps):
    run(items, fold); run(items, fold)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(items, fold)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for B in a.batch:
    for st, C in enumerate((96, 192, 384, 768)):
        tok = B * 16 * 256 >> (2 * st)
        shapes = [(3 * C, C), (C, C), (4 * C, C), (C, 4 * C)]
        bufs = []
        for Nw, Kw in shapes:
            dY = (torch.randn(tok, Nw, device=dev) * 0.5).bfloat16()
            X = torch.randn(tok, Kw, device=dev).bfloat16()
            bufs.append((dY, X, torch.zeros(Nw, Kw, device=dev), torch.zeros(Nw, device=dev)))
        fl = sum(2.0 * tok * Nw * Kw for Nw, Kw in shapes)
        line = f"B={B:3d} C={C:3d} tok={tok:6d}"
        for mode in a.modes:
            SMALL = not (mode & 1)
            items, nwg, slab = [], 0, 0
            gt = sum(ops.wgrad_tiles(Nw, Kw, SMALL) for Nw, Kw in shapes) if mode & 1 else 0
            sps = [Engine._splits(Nw, Kw, tok, group_tiles=gt) for Nw, Kw in shapes]
            while sum((Nw * Kw + Nw) * sp * 4 for (Nw, Kw), sp in zip(shapes, sps) if sp > 1) > ws.numel() * 4:
                sps = [max(1, sp // 2) for sp in sps]       # (the engine starts a second launch instead)
            for (Nw, Kw), (dY, X, dW, db), sp in zip(shapes, bufs, sps):
                nwg += ops.wgrad_tiles(Nw, Kw, SMALL) * sp
                slab += Nw * Kw * 4 * sp if sp > 1 else 0
                items.append(ops.wgrad_item(dY, Nw, X, Kw, Nw, Kw, tok, dW, db, sp))
            t0 = timeit(items, False, a.reps)
            t1 = timeit(items, True, a.reps)
            line += f" | mode {mode:#x}: {nwg:5d} wg, slabs {slab / 1e6:6.1f} MB, gemm {t0:7.1f} us ({fl / t0 / 1e6:6.1f} TF/s), +fold {t1:7.1f} us"
        print(line, flush=True)
