"""Isolated launch times of the deep-stage Swin block (csrc/swind.hip): the split launches against the sequences they replace.
python tools/bench_deep.py [--model base|large] [--rows 16] [--cols 1024] [--batch 8] [--stage 3] [--cold]"""
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
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cold", action="store_true", help="stream 600 MB through the caches between launches")
    a = ap.parse_args()
    from tulip_amd.model import tulip as T
    from tulip_amd import ops
    dev = torch.device("cuda", 0)
    f = T.tulip_base if a.model == "base" else T.tulip_large
    torch.manual_seed(0)
    m = f(img_size=(a.rows, a.cols), target_img_size=(4 * a.rows, a.cols), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
          pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).to(dev).train()
    eng = m.engine()
    eng.deep_min_windows, eng.deep_max_windows = 1, 1 << 30     # (measure the sliced form at every size, whatever the engine's gate)
    eng.bind(dev)
    eng.params.refresh_shadow()
    B = a.batch
    P = eng.plan(B)
    sp = eng.enc_blocks[a.stage][1]
    M, C = B * sp.H * sp.W, sp.C
    print(f"{a.model} {a.rows}x{a.cols} B={B} stage {a.stage}: C={C} tokens {M} windows {M // 16} win {sp.win} shift {sp.sft} "
          f"groups {ops.swind_groups(C, B, sp.H, sp.W, sp.win)}")
    xin = P[f"enc{a.stage}.in"]
    xin.copy_(torch.randn_like(xin))
    eng.draw_drop_scales(P, True, 0.5 + 0.5 * torch.rand(eng.n_drop_slots, B, device=dev))
    out = torch.empty(M, C, device=dev)
    dy = torch.randn(M, C, device=dev)
    eng.overlap_wgrad = True                    # weight gradients queued, never issued: the chain's launches only
    junk = torch.empty(150 << 20, device=dev) if a.cold else None
    gflat = torch.zeros(eng.params.total, device=dev)
    G = lambda name: gflat.data_ptr() + 4 * eng.params.offset[name]

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.iters):
            if junk is not None:
                junk.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2], ts[0]

    def fwd():
        eng._block_fwd(P, sp, xin, out)

    def bwd():
        dx = P["enc%d.dx" % a.stage]
        eng._pending, eng._lagged_hook = [], None
        eng._block_bwd(P, sp, xin, dx, G, have_dyb=False)
        eng._pending = []

    for fused in (False, True):
        eng.fuse_deep = fused
        fwd()
        P["enc%d.dx" % a.stage].copy_(dy.view_as(P["enc%d.dx" % a.stage]))
        tf, tb = timeit(fwd), timeit(bwd)
        print(f"{'split launches' if fused else 'sequence      '}: forward {tf[0]:7.1f} us (min {tf[1]:6.1f})   backward {tb[0]:7.1f} us (min {tb[1]:6.1f})")
    # the launches one by one
    eng.fuse_deep = True
    import tulip_amd.ops as O_
    real_f, real_b = O_.swind_block_fwd, O_.swind_block_bwd
    names = {1: "by heads       (F1 norm1+qkv+attention | B3 proj'+attention')", 2: "by out channels (F2 proj | B4 qkv')",
             4: "by hidden      (F3 norm2+fc1+GELU | B1 fc2'+GELU')", 8: "by out channels (F4 fc2 | B2 fc1')"}
    bmap = {1: 4, 2: 8, 4: 1, 8: 2}
    for ph in (1, 2, 4, 8):
        O_.swind_block_fwd = lambda *aa, **kw: real_f(*aa, **{**kw, "phases": ph})
        O_.swind_block_bwd = lambda *aa, **kw: (real_b(*aa, **{**kw, "phases": bmap[ph]}) if kw.get("phases", 15) & bmap[ph] else None)
        tf = timeit(fwd)
        real_ln = eng._ln_bwd
        eng._ln_bwd = lambda *aa, **kw: None
        tb = timeit(bwd)
        eng._ln_bwd = real_ln
        print(f"  {names[ph]:64s}: forward {tf[0]:6.1f} us (min {tf[1]:6.1f})   backward {tb[0]:6.1f} us (min {tb[1]:6.1f})")
    O_.swind_block_fwd, O_.swind_block_bwd = real_f, real_b


if __name__ == "__main__":
    main()
