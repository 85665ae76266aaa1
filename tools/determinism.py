"""dev tool: two Trainers from the same seed, N graph-replayed steps each, parameters compared bit for bit.
usage: python tools/determinism.py [steps=30] [batch=8]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd.trainer import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a = argparse.Namespace(model="tulip_base", img=[16, 1024], target=[64, 1024], batch=B)
dev = torch.device("cuda", 0)
res = []
for run in range(2):
    m = bench.make_model(a).to(dev).train()
    tr = Trainer(m, B, device=dev)
    lo, hi = bench.synthetic(a, 0, dev)
    tr.load_batch(lo, hi)
    ls = [tr.step().clone() for _ in range(steps)]
    torch.cuda.synchronize()
    res.append((tr.eng.params.flat.clone(), torch.stack(ls).cpu()))
    del tr, m
same = torch.equal(res[0][0], res[1][0])
first = next((i for i in range(steps) if not torch.equal(res[0][1][i], res[1][1][i])), None)
print(f"parameters bit-identical after {steps} steps: {same}; first step whose loss differs: {first}; "
      f"max |d param| {(res[0][0] - res[1][0]).abs().max().item():.3e}")
