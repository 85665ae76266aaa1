"""In-kernel phase stamps (s_memtime, shader clock) of the deep-stage split launches (csrc/swind.hip).
python tools/deep_phases.py [--model base] [--rows 16] [--cols 1024] [--batch 8] [--stage 3] [--cold]
Stamps: 0 start | 1 prologue done (LayerNorm / operand rows in LDS) | 2 barrier | 3 GEMM done | 4 end (by-heads launches: k-half
exchange barrier) | 5 end of the attention phase (by-heads launches)."""
import os as _os
_os.environ.setdefault("TULIP_HIP_DEV", "1")     # the profiled twins live in libtulip_hip_dev.so (include/tulip_hip.h, conventions)
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="base")
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--cols", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--stage", type=int, default=3)
    ap.add_argument("--cold", action="store_true")
    a = ap.parse_args()
    from tulip_amd.model import tulip as T
    from tulip_amd import ops
    import tulip_amd.ops as O_
    dev = torch.device("cuda", 0)
    f = T.tulip_base if a.model == "base" else T.tulip_large
    torch.manual_seed(0)
    m = f(img_size=(a.rows, a.cols), target_img_size=(4 * a.rows, a.cols), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
          pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).to(dev).train()
    eng = m.engine()
    eng.bind(dev)
    eng.params.refresh_shadow()
    B = a.batch
    P = eng.plan(B)
    sp = eng.enc_blocks[a.stage][1]
    M, C = B * sp.H * sp.W, sp.C
    ngrp = ops.swind_groups(C, B, sp.H, sp.W, sp.win)
    nwv = C // 128
    nwg = ngrp * 8
    xin = P[f"enc{a.stage}.in"]
    xin.copy_(torch.randn_like(xin))
    eng.draw_drop_scales(P, True, 0.5 + 0.5 * torch.rand(eng.n_drop_slots, B, device=dev))
    out = torch.empty(M, C, device=dev)
    gflat = torch.zeros(eng.params.total, device=dev)
    G = lambda name: gflat.data_ptr() + 4 * eng.params.offset[name]
    junk = torch.empty(150 << 20, device=dev) if a.cold else None
    real_f, real_b = O_.swind_block_fwd, O_.swind_block_bwd
    stamps = torch.zeros(4, nwg, nwv, 16, dtype=torch.int64, device=dev)
    O_.swind_block_fwd = lambda *aa, **kw: real_f(*aa, **{**kw, "stamps": stamps})
    O_.swind_block_bwd = lambda *aa, **kw: real_b(*aa, **{**kw, "stamps": stamps})

    def report(title):
        torch.cuda.synchronize()
        s = stamps.cpu().double()
        for l in range(4):
            st = s[l]                                   # [nwg][nwv][16]
            t0 = st[:, :, 0].min()
            print(f"  {title} launch {l}: span {(st[:, :, :6].max() - t0):.0f} cycles (first stamp to last)")
            rel = st - st[:, :, 0:1]
            for k in range(1, 6):
                sel = rel[:, :, k][st[:, :, k] > 0]
                if sel.numel():
                    print(f"      stamp {k}: mean {sel.mean():9.0f}  min {sel.min():9.0f}  max {sel.max():9.0f}   ({sel.numel()} waves)")
            starts = st[:, 0, 0] - t0
            print(f"      workgroup start skew: mean {starts.mean():.0f} max {starts.max():.0f}")

    for it in range(3):
        stamps.zero_()
        if junk is not None:
            junk.add_(1.0)
        eng._block_fwd(P, sp, xin, out)
        if it == 2:
            report("forward ")
    dx = P["enc%d.dx" % a.stage]
    for it in range(3):
        stamps.zero_()
        dx.copy_(torch.randn_like(dx))
        if junk is not None:
            junk.add_(1.0)
        eng._pending, eng._lagged_hook = [], None
        eng._block_bwd(P, sp, xin, dx, G, have_dyb=False)
        eng._pending = []
        if it == 2:
            report("backward")


if __name__ == "__main__":
    main()
