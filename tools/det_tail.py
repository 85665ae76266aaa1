"""dev tool: run each head-backward kernel repeatedly on the same inputs; outputs must be bit-identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd import ops
dev = torch.device("cuda", 0)
B, H, W, E = 8, 16, 256, 96
M = B * H * W
g = torch.Generator(device=dev).manual_seed(0)
xn = torch.randn(M, E, device=dev, generator=g).bfloat16()
x = torch.randn(M, E, device=dev, generator=g)
We = (0.1 * torch.randn(16 * E, E, device=dev, generator=g)).bfloat16()
be, wd = 0.1 * torch.randn(16 * E, device=dev, generator=g), 0.2 * torch.randn(E, device=dev, generator=g)
pred, target = torch.randn(B, 1, 4 * H, 4 * W, device=dev, generator=g), torch.randn(B, 1, 4 * H, 4 * W, device=dev, generator=g)
gam = torch.ones(E, device=dev); mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev)
kw = dict(target=target, gscale=1.0)
sp = ops.tail_wgrad_splits(B, H, W, E)
R = (M + 31) // 32
def dgrad():
    dxn = torch.full((M, E), 7.0, dtype=torch.bfloat16, device=dev); part = torch.full((R, 128), 7.0, device=dev)
    ops.tail_bwd_dgrad(xn, We, be, wd, pred, dxn, part, B, H, W, E, **kw); return [dxn, part]
def dgrad_ln():
    dx = torch.full((M, E), 7.0, device=dev); part = torch.full((R, 128), 7.0, device=dev); lnp = torch.full((R, 2 * E), 7.0, device=dev)
    ops.tail_bwd_dgrad_ln(xn, We, be, wd, pred, part, B, H, W, E, x, mean, rstd, gam, dx, lnp, **kw); return [dx, part, lnp]
def wgrad():
    sw = torch.full((sp, 16 * E * E), 7.0, device=dev); sb = torch.full((sp, 16 * E), 7.0, device=dev)
    ops.tail_wgrad(xn, We, be, wd, pred, sw, sb, B, H, W, E, **kw); return [sw, sb]
def fwd_ln():
    o = [torch.full((M, E), 7.0, dtype=torch.bfloat16, device=dev), torch.full((M,), 7.0, device=dev), torch.full((M,), 7.0, device=dev),
         torch.full((B, 1, 4 * H, 4 * W), 7.0, device=dev), torch.full((2 * R,), 7.0, device=dev)]
    ops.tail_fwd_ln(x, gam, be[:E].contiguous(), 1e-6, o[0], o[1], o[2], We, be, wd, o[3], B, H, W, E, target=target, loss_partials=o[4], log_transform=True)
    return o
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
# optional: something else keeps the chip busy on another stream while the kernel under test runs
NOISE = os.environ.get("NOISE", "0") == "1"
side = torch.cuda.Stream()
na, nb_ = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
big = torch.empty(64 << 20, device=dev)
def noisy(f):
    def g():
        if NOISE:
            with torch.cuda.stream(side):
                for _ in range(3):
                    big.add_(1.0); torch.mm(na, nb_)
        return f()
    return g
dgrad, dgrad_ln, wgrad, fwd_ln = noisy(dgrad), noisy(dgrad_ln), noisy(wgrad), noisy(fwd_ln)
for name, fn in [("tail_bwd_dgrad", dgrad), ("tail_bwd_dgrad_ln", dgrad_ln), ("tail_wgrad", wgrad), ("tail_fwd_ln", fwd_ln)]:
    ref = fn(); torch.cuda.synchronize()
    bad = 0
    for it in range(REPS):
        out = fn(); torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(out, ref)):
            if not torch.equal(a, b):
                bad += 1
                if bad <= 3:
                    d = (a.float() - b.float()).abs()
                    idx = d.reshape(-1).argmax().item()
                    print(f"  {name}: output {k} differs on repeat {it}: {int((d > 0).sum())} elements, max {d.max().item():.3e} at flat index {idx} (row {idx // a.shape[-1]})")
    print(f"{name}: {bad} mismatching outputs over {REPS} repeats")
