"""dev tool: the same forward + backward (same weights, inputs, DropPath draws) issued repeatedly; which gradient tensors differ?"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd.trainer import Trainer
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
a = argparse.Namespace(model="tulip_base", img=[16, 1024], target=[64, 1024], batch=8)
dev = torch.device("cuda", 0)
m = bench.make_model(a).to(dev).train()
tr = Trainer(m, 8, device=dev, use_graph=False)
if os.environ.get("NO_OVERLAP") == "1":
    tr.eng.overlap_wgrad = False
tr.inject_drop_u = torch.rand(tr.eng.n_drop_slots, 8, device=dev)
lo, hi = bench.synthetic(a, 0, dev); tr.load_batch(lo, hi)
W = tr.eng.params
ref = None
bad = {}
for it in range(reps):
    tr.g.zero_()
    tr._fwd_bwd(lambda tag: None)
    torch.cuda.synchronize()
    g = tr.g.clone()
    if ref is None:
        ref = g; continue
    if not torch.equal(g, ref):
        for n in W.names:
            s = slice(W.offset[n], W.offset[n] + W.numel[n])
            if not torch.equal(g[s], ref[s]):
                d = (g[s] - ref[s]).abs()
                bad.setdefault(n, []).append((it, int((d > 0).sum()), d.max().item(), ref[s].abs().max().item()))
print(f"{reps} repeats; gradient tensors that ever differed: {len(bad)}")
for n, v in list(bad.items())[:40]:
    print(f"  {n}: {len(v)} times, e.g. repeat {v[0][0]}: {v[0][1]} elements, max |d| {v[0][2]:.3e} (|g| max {v[0][3]:.3e})")
