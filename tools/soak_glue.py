import os, sys
sys.path.insert(0, "/root/repo")
import torch
from oracle import tulip_oracle as O
from tests.test_model_gpu import build
from tulip_amd.trainer import Trainer
DEV = "cuda"
cfg = O.tulip_base_config()
sd = O.key_seeded_state_dict(cfg, seed=5)
res = {}
for glue in (True, False):
    torch.manual_seed(3)
    m = build(cfg, sd, train=True)
    m.engine().fuse_glue = glue
    tr = Trainer(m, 8, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01)
    ls = []
    for it in range(300):
        lo, hi = O.synthetic_batch(cfg, 8, seed=100 + it % 16)
        ls.append(tr.step(lo.to(DEV), hi.to(DEV)).clone())
    torch.cuda.synchronize()
    res[glue] = torch.stack(ls)[:, 0].cpu()
a, b = res[True], res[False]
print("loss first/last (fused boundaries):", a[0].item(), a[-1].item(), " (launch sequences):", b[0].item(), b[-1].item())
rel = ((a - b).abs() / b.abs())
print("max relative loss difference over 300 steps:", rel.max().item(), "at step", int(rel.argmax()), "; mean", rel.mean().item())
print("finite:", bool(torch.isfinite(a).all()), " decreasing:", bool(a[-20:].mean() < a[:20].mean()))
