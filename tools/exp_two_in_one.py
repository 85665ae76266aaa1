"""dev experiment: two B=4 chains captured as parallel branches of ONE HIP graph vs one B=8 chain."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd.trainer import Trainer
dev = torch.device("cuda", 0)
def mk(B, graph=True):
    args = argparse.Namespace(model="tulip_base", img=[16,1024], target=[64,1024], batch=B)
    m = bench.make_model(args).to(dev).train(); tr = Trainer(m, B, device=dev, use_graph=graph)
    lo, hi = bench.synthetic(args, 0, dev); tr.load_batch(lo, hi)
    for _ in range(3): tr.step()
    torch.cuda.synchronize(); return tr
def timeit(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t8 = mk(8); print("B=8 single chain:", round(timeit(t8.step), 3), "ms / 8 images")
nch = int(os.environ.get("CHAINS", 2))
trs = [mk(8 // nch, graph=False) for _ in range(nch)]
for t in trs: t.g.zero_()
cap = torch.cuda.Stream(); ss = [torch.cuda.Stream() for _ in range(nch)]
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cap):
    for s in ss: s.wait_stream(cap)
    for s, t in zip(ss, trs):
        with torch.cuda.stream(s):
            t._fwd_bwd(lambda tag: None); t._adamw()
    for s in ss: cap.wait_stream(s)
print(f"{nch} x B={8 // nch} chains in one graph:", round(timeit(g.replay), 3), "ms / 8 images")
