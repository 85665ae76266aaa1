"""dev: where exactly do two runs of the captured step differ (indices inside the first differing tensors; moments too)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import tulip_oracle as O
from tests.test_model_gpu import build
from tulip_amd.trainer import Trainer
DEV = "cuda"
cfg = O.tulip_base_config()
sd = O.key_seeded_state_dict(cfg, seed=5)
lo, hi = O.synthetic_batch(cfg, 8, seed=7)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def run(**attrs):
    torch.manual_seed(3)
    m = build(cfg, sd, train=True)
    eng = m.engine()
    for k, v in attrs.items():
        setattr(eng, k, v)
    tr = Trainer(m, 8, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01)
    tr.load_batch(lo.to(DEV), hi.to(DEV))
    for _ in range(NS):
        tr.step()
    torch.cuda.synchronize()
    return tr, tr.eng.params.flat.clone(), tr.m.clone(), tr.v.clone()
base = None
for rep in range(12):
    tr, f, m, v = run()
    if base is None:
        base = (f, m, v); continue
    W = tr.eng.params
    for nm, a, b in (("param", f, base[0]), ("exp_avg", m, base[1]), ("exp_avg_sq", v, base[2])):
        d = (a != b).nonzero().flatten().cpu()
        if d.numel() == 0:
            continue
        print(f"rep {rep} {nm}: {d.numel()} elements differ")
        shown = 0
        for n in W.names:
            o, cnt = W.offset[n], W.numel[n]
            sel = d[(d >= o) & (d < o + cnt)] - o
            if sel.numel():
                cols = W.shape[n][-1] if len(W.shape[n]) > 1 else cnt
                if len(W.shape[n]) == 4: cols = W.shape[n][1] * W.shape[n][2] * W.shape[n][3]
                rows = sorted(set((sel // cols).tolist()))
                print(f"   {n} {W.shape[n]}: {sel.numel()} differ; rows {rows[:12]}{'...' if len(rows) > 12 else ''} cols {sorted(set((sel % cols).tolist()))[:20]}"
                      f" max|d| {(a[o:o+cnt]-b[o:o+cnt]).abs().max().item():.3e} rel {((a[o:o+cnt]-b[o:o+cnt]).abs().max()/(b[o:o+cnt].abs().max()+1e-30)).item():.2e}")
                k = sel[:6] + o
                print("      this run :", [f"{x:.4e}" for x in a[k].tolist()], "\n      first run:", [f"{x:.4e}" for x in b[k].tolist()],
                      "\n      neighbours (this run, +1..+3):", [f"{x:.4e}" for x in a[k[0] + 1:k[0] + 4].tolist()])
                shown += 1
                if shown >= 6: break
