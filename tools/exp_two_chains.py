"""dev experiment: one B=8 chain vs two concurrent B=4 chains (two graphs on two streams)."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd.trainer import Trainer
dev = torch.device("cuda", 0)
def mk(B):
    args = argparse.Namespace(model="tulip_base", img=[16,1024], target=[64,1024], batch=B)
    m = bench.make_model(args).to(dev).train(); tr = Trainer(m, B, device=dev)
    lo, hi = bench.synthetic(args, 0, dev); tr.load_batch(lo, hi)
    for _ in range(3): tr.step()
    torch.cuda.synchronize(); return tr
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t8 = mk(8); print("B=8 single chain:", round(timeit(t8.step), 3), "ms / 8 images")
a, b = mk(4), mk(4)
print("B=4 single chain:", round(timeit(a.step), 3), "ms / 4 images")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): a.step()
    with torch.cuda.stream(s2): b.step()
    cur.wait_stream(s1); cur.wait_stream(s2)
print("2 x B=4 concurrent chains:", round(timeit(both), 3), "ms / 8 images")
c, d = mk(2), mk(2); e, f = mk(2), mk(2)
ss = [torch.cuda.Stream() for _ in range(4)]
def four():
    cur = torch.cuda.current_stream()
    for s in ss: s.wait_stream(cur)
    for s, t in zip(ss, (c, d, e, f)):
        with torch.cuda.stream(s): t.step()
    for s in ss: cur.wait_stream(s)
print("4 x B=2 concurrent chains:", round(timeit(four), 3), "ms / 8 images")
