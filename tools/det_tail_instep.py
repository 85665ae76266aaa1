"""dev tool: tulip_tail_wgrad on the training step's OWN buffers (after one forward + backward), repeated: slabs bit-identical?"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd import ops
from tulip_amd.trainer import Trainer
a = argparse.Namespace(model="tulip_base", img=[16, 1024], target=[64, 1024], batch=8)
dev = torch.device("cuda", 0)
m = bench.make_model(a).to(dev).train()
tr = Trainer(m, 8, device=dev, use_graph=False)
lo, hi = bench.synthetic(a, 0, dev); tr.load_batch(lo, hi)
tr._fwd_bwd(lambda tag: None); torch.cuda.synchronize()
P, W_ = tr.P, tr.eng.params
B, H, W, E = 8, 16, 256, 96
sp = ops.tail_wgrad_splits(B, H, W, E)
args = (P["tail.xn"], W_.p16("ps_head.conv_expand.0.weight"), W_.p32("ps_head.conv_expand.0.bias"), W_.p32("decoder_pred.weight"), P.pred)
def run():
    sw = torch.full((sp, 16 * E * E), 7.0, device=dev); sb = torch.full((sp, 16 * E), 7.0, device=dev)
    ops.tail_wgrad(*args, sw, sb, B, H, W, E, target=P.target, gscale=1.0)
    torch.cuda.synchronize(); return sw, sb
ref = run(); bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    sw, sb = run()
    if not torch.equal(sw, ref[0]):
        bad += 1
        d = (sw - ref[0]).abs()
        rows = (d.reshape(sp, 16 * E, E) > 0).any(-1).nonzero()
        if bad <= 4:
            print(f"repeat {it}: {int((d > 0).sum())} slab elements differ, max {d.max().item():.3e}; (split, oc) range: splits {sorted(set(rows[:, 0].tolist()))}, oc {rows[:, 1].min().item()}..{rows[:, 1].max().item()}")
print("mismatching repeats:", bad, "| any 7.0 left in slabs:", bool((ref[0] == 7.0).any()), bool((ref[1] == 7.0).any()))
# the fold of those slabs, repeated
nw = 16 * E * E
outs = []
for it in range(200):
    dW, db = torch.zeros(nw, device=dev), torch.zeros(16 * E, device=dev)
    ops.reduce_rows_multi([ops.reduce_region(ref[0], nw, dW, nw, sp), ops.reduce_region(ref[1], 16 * E, db, 16 * E, sp)])
    torch.cuda.synchronize()
    outs.append((dW, db))
print("fold mismatches:", sum(1 for o in outs[1:] if not (torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]))))
# the whole head section of the backward as the engine issues it, repeated (same forward state)
eng = tr.eng
name = "ps_head.conv_expand.0.weight"
sl = slice(W_.offset[name], W_.offset[name] + W_.numel[name])
gs = []
for it in range(150):
    tr.g.zero_()
    eng.run_backward(P, tr.g)
    torch.cuda.synchronize()
    gs.append(tr.g[sl].clone())
print("run_backward repeats with differing head weight gradient:", sum(1 for x in gs[1:] if not torch.equal(x, gs[0])))
# which stage differs: the slabs left in the plan's scratch after run_backward, and the folded gradient
slabbuf = P.bufs["tail.wslab"]
rec = []
for it in range(int(os.environ.get('REPS', '200'))):
    tr.g.zero_()
    eng.run_backward(P, tr.g)
    torch.cuda.synchronize()
    rec.append((slabbuf.clone(), tr.g[sl].clone()))
ds = sum(1 for x in rec[1:] if not torch.equal(x[0], rec[0][0])); dg = sum(1 for x in rec[1:] if not torch.equal(x[1], rec[0][1]))
print(f"over REPS run_backward repeats: slabs differ {ds} times, folded head weight gradient differs {dg} times")
for x in rec[1:]:
    if not torch.equal(x[0], rec[0][0]):
        d = (x[0] - rec[0][0]).abs(); idx = (d > 0).nonzero().reshape(-1)
        print("  slab elements differing:", idx.numel(), "first", idx[:4].tolist(), "last", idx[-4:].tolist(), "max", d.max().item(), "(weights region ends at", sp * nw, ")")
        break
