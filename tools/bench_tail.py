"""dev tool: the head backward's kernels in isolation (HIP events, 20 back-to-back launches): the dz-materialising form
(tulip_tail_bwd + data-gradient GEMM + weight-gradient launch) against tulip_tail_bwd_dgrad + tulip_tail_wgrad.
usage: python tools/bench_tail.py [batch ...]      (TULIP_TAIL_WGRAD_NC=4|2 selects the weight-gradient variant)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd import ops
from tulip_amd._lib import EPI_BF16

dev = torch.device("cuda", 0)
H, W, E = 16, 256, 96


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B in [int(a) for a in sys.argv[1:]] or [8, 64]:
    M = B * H * W
    g = torch.Generator(device=dev).manual_seed(0)
    xn = torch.randn(M, E, device=dev, generator=g).bfloat16()
    We = (0.1 * torch.randn(16 * E, E, device=dev, generator=g)).bfloat16()
    be, wd = 0.1 * torch.randn(16 * E, device=dev, generator=g), 0.2 * torch.randn(E, device=dev, generator=g)
    pred, target = torch.randn(B, 1, 4 * H, 4 * W, device=dev, generator=g), torch.randn(B, 1, 4 * H, 4 * W, device=dev, generator=g)
    dz = torch.empty(M, 16 * E, dtype=torch.bfloat16, device=dev)
    dxn = torch.empty(M, E, dtype=torch.bfloat16, device=dev)
    part = torch.zeros((M + 31) // 32, 128, device=dev)
    sp = ops.tail_wgrad_splits(B, H, W, E)
    sw, sb = torch.empty(sp, 16 * E * E, device=dev), torch.empty(sp, 16 * E, device=dev)
    dW, db = torch.zeros(16 * E, E, device=dev), torch.zeros(16 * E, device=dev)
    ws = torch.empty(17 << 20, device=dev)
    kw = dict(target=target, gscale=1.0)
    t_old = timed(lambda: ops.tail_bwd(xn, We, be, wd, pred, dz, part, B, H, W, E, **kw))
    t_gemm = timed(lambda: ops.gemm(dz, We, M, E, 16 * E, lda=16 * E, ldb=E, b_trans=True, epi=EPI_BF16, out=dxn, ldo=E))
    from tulip_amd.engine import TulipEngine
    it = ops.wgrad_item(dz, 16 * E, xn, E, 16 * E, E, M, dW, db, TulipEngine._splits(16 * E, E, M, group_tiles=ops.wgrad_tiles(16 * E, E)))
    t_wg = timed(lambda: ops.wgrad_group([it], [], ws, ws.numel() * 4))
    t_dg = timed(lambda: ops.tail_bwd_dgrad(xn, We, be, wd, pred, dxn, part, B, H, W, E, **kw))
    t_tw = timed(lambda: ops.tail_wgrad(xn, We, be, wd, pred, sw, sb, B, H, W, E, **kw))
    regs = [ops.reduce_region(sw, 16 * E * E, dW, 16 * E * E, sp), ops.reduce_region(sb, 16 * E, db, 16 * E, sp)]
    t_fold = timed(lambda: ops.reduce_rows_multi(regs))
    print(f"B={B}: old tail_bwd {t_old:.1f} + dgrad GEMM {t_gemm:.1f} | wgrad_group(+fold) {t_wg:.1f}   ->   "
          f"tail_bwd_dgrad {t_dg:.1f} | tail_wgrad {t_tw:.1f} (splits {sp}, NC {os.environ.get('TULIP_TAIL_WGRAD_NC', '2')}) + fold {t_fold:.1f}  us")
