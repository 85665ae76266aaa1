"""dev tool: how much of the in-step time of the fused block kernels is cold caches?  Each recorded block launch of
one step is timed (HIP events around the single launch) warm (same launch just ran), after a 32-MB fill (L2 evicted,
Infinity Cache mostly kept) and after a 1-GB fill (both evicted).  usage: python tools/cold_probe.py [batch=8]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd import ops
from tulip_amd.trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
a = argparse.Namespace(model="tulip_base", img=[16, 1024], target=[64, 1024], batch=B)
dev = torch.device("cuda", 0)
m = bench.make_model(a).to(dev).train()
tr = Trainer(m, B, device=dev)
lo, hi = bench.synthetic(a, 0, dev); tr.load_batch(lo, hi)
for _ in range(3): tr.step()
names = ("swin96_block_fwd", "swin96_block_bwd", "swinw_block_fwd", "swinw_block_bwd", "gemm", "tail_fwd", "tail_bwd", "layernorm_fwd",
         "layernorm_bwd", "layernorm_bwd_splitk", "splitk_resid_ln", "patch_embed_fwd", "patch_embed_bwd", "window_attn_fwd",
         "window_attn_bwd", "wgrad_group", "reduce_rows_multi", "tail_wgrad")
real = {n: getattr(ops, n) for n in names}
rec = []
def wrap(n):
    def f(*a, **kw):
        tag = n + (f" C={a[0]}" if n.startswith("swinw") else "") + (f" {a[2]}x{a[3]}x{a[4]} s{kw.get('splits', 1)}" if n == "gemm" else "")
        if n == "wgrad_group":
            tag += f" #{sum(1 for t, _ in rec if t.startswith('wgrad_group'))}"
        rec.append((tag, lambda: real[n](*a, **kw)))
        real[n](*a, **kw)
    return f
for n in names: setattr(ops, n, wrap(n))
try:
    tr._fwd_bwd(lambda tag: None); torch.cuda.synchronize()
finally:
    for n in names: setattr(ops, n, real[n])
small = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
def timed(call, pre, reps=10):
    ts = []
    for _ in range(reps):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
agg = {}
for tag, call in rec:
    call()
    w = timed(call, call); l2 = timed(call, lambda: small.fill_(1)); c = timed(call, lambda: big.fill_(1))
    g = agg.setdefault(tag, [0, 0.0, 0.0, 0.0]); g[0] += 1; g[1] += w; g[2] += l2; g[3] += c
print(f"{'launch':42s} {'n':>3s} {'warm':>8s} {'L2 cold':>8s} {'all cold':>8s}   (median us per launch incl. ~2-3 us event overhead)")
tot = [0, 0, 0]
for tag, (n, w, l2, c) in agg.items():
    print(f"{tag:42s} {n:3d} {w / n:8.1f} {l2 / n:8.1f} {c / n:8.1f}")
    tot[0] += w; tot[1] += l2; tot[2] += c
print(f"sum over the step's launches of these kinds: warm {tot[0]:.0f} us, L2 cold {tot[1]:.0f} us, all cold {tot[2]:.0f} us")
