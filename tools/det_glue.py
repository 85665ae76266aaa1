"""dev: is the training step deterministic run to run (four steps, bit-compare of the parameters) with a subset of the fused
stage-boundary forms, captured / eager, with / without the side queue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import tulip_oracle as O
from tests.test_model_gpu import build
from tests.conftest import describe_flat_diff
from tulip_amd.trainer import Trainer
DEV = "cuda"
cfg = O.tulip_base_config()
sd = O.key_seeded_state_dict(cfg, seed=5)
lo, hi = O.synthetic_batch(cfg, 8, seed=7)
def run(nsteps=4, use_graph=True, **attrs):
    torch.manual_seed(3)
    m = build(cfg, sd, train=True)
    eng = m.engine()
    for k, v in attrs.items():
        setattr(eng, k, v)
    tr = Trainer(m, 8, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01, use_graph=use_graph)
    tr.load_batch(lo.to(DEV), hi.to(DEV))
    ls = [tr.step().clone() for _ in range(nsteps)]
    torch.cuda.synchronize()
    return tr, tr.eng.params.flat.clone(), torch.stack(ls)
ALL = ("merge_fwd", "merge_bwd", "unmerge_fwd", "unmerge_bwd")
variants = [("all forms", dict()), ("no fused boundary", dict(glue_forms=frozenset()))]
for name, kw in variants:
    env = kw.pop("_env", None)
    if env: os.environ[env[0]] = env[1]
    base, bad, why = None, 0, ""
    for rep in range(9):
        tr, f, ls = run(**kw)
        if base is None:
            base = f
        elif not torch.equal(f, base):
            bad += 1
            why = describe_flat_diff(tr.eng, f, base, limit=2)
    if env: del os.environ[env[0]]
    print(f"{name}: {bad}/8 repeats differ", why if bad else "", flush=True)
