"""Phase stamps of the large-tile weight-gradient kernel (tulip_wgrad_group_profiled): per workgroup, s_memtime at start /
end of prologue / end of k-loop / end of write-out, for the grouped launch of one block.  The counters of different
XCDs have different bases: only differences inside a workgroup mean something; the tick is about the shader clock
(prologue + k-loop + write-out of the slowest workgroup ~ the kernel's duration x 1.7 GHz)."""
import os as _os
_os.environ.setdefault("TULIP_HIP_DEV", "1")     # the profiled twins live in libtulip_hip_dev.so (include/tulip_hip.h, conventions)
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd import ops
from tulip_amd.engine import TulipEngine as Engine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[8, 64])
ap.add_argument("--stages", type=int, nargs="+", default=[0, 1, 2])
a = ap.parse_args()
dev = torch.device("cuda:0")
ws = torch.empty((Engine.WS_ELEMS + (1 << 20)), device=dev)
for B in a.batch:
    for st, C in enumerate((96, 192, 384, 768)):
        if st not in a.stages:
            continue
        tok = B * 16 * 256 >> (2 * st)
        shapes = [(3 * C, C), (C, C), (4 * C, C), (C, 4 * C)]
        gt = sum(ops.wgrad_tiles(Nw, Kw) for Nw, Kw in shapes)
        sps = [Engine._splits(Nw, Kw, tok, group_tiles=gt) for Nw, Kw in shapes]
        while sum((Nw * Kw + Nw) * sp * 4 for (Nw, Kw), sp in zip(shapes, sps) if sp > 1) > ws.numel() * 4:
            sps = [max(1, sp // 2) for sp in sps]
        items, nwg = [], 0
        for (Nw, Kw), sp in zip(shapes, sps):
            dY = (torch.randn(tok, Nw, device=dev) * 0.5).bfloat16()
            X = torch.randn(tok, Kw, device=dev).bfloat16()
            items.append(ops.wgrad_item(dY, Nw, X, Kw, Nw, Kw, tok, torch.zeros(Nw, Kw, device=dev), torch.zeros(Nw, device=dev), sp))
            nwg += ops.wgrad_tiles(Nw, Kw) * sp
        for _ in range(3):
            ops.wgrad_group(items, [], ws, ws.numel() * 4, fold=False)
        stamps = torch.zeros(nwg, 4, dtype=torch.int64, device=dev)
        ops.wgrad_group_profiled(items, ws, ws.numel() * 4, stamps)
        torch.cuda.synchronize()
        s = stamps.cpu().double()
        rel = s
        ksteps = [tok // sp // 32 for sp in sps]
        d = lambda x: f"{x.min():8.0f} / {x.median():8.0f} / {x.max():8.0f}"
        print(f"B={B} C={C} tok={tok} wgs={nwg} k-steps per wg {ksteps}")
        print(f"   prologue      {d(rel[:, 1] - rel[:, 0])}   (min / median / max over workgroups, ticks)")
        print(f"   k-loop        {d(rel[:, 2] - rel[:, 1])}")
        print(f"   write-out     {d(rel[:, 3] - rel[:, 2])}")
        print(f"   per k-step    {(rel[:, 2] - rel[:, 1]).median() / max(ksteps):6.0f}", flush=True)
