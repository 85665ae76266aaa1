"""dev tool: per-(kind, M,N,K, epi) GEMM timing inside one training step (eager, HIP events)."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd import ops
from tulip_amd.trainer import Trainer
args = argparse.Namespace(model="tulip_base", img=[16, 1024], target=[64, 1024], batch=int(os.environ.get("B", 8)))
dev = torch.device("cuda", 0)
model = bench.make_model(args).to(dev).train()
tr = Trainer(model, args.batch, device=dev, use_graph=False)
lo, hi = bench.synthetic(args, 0, dev); tr.load_batch(lo, hi)
rec = []; real = ops.gemm
def timed(A, B, M, N, K, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); real(A, B, M, N, K, **kw); e1.record()
    kind = "wgrad" if kw.get("a_trans") else ("dgrad" if kw.get("b_trans") else "fwd")
    rec.append((e0, e1, kind, M, N, K, kw.get("epi", 0), kw.get("splits", 1)))
ops.gemm = timed
for _ in range(3):
    rec.clear(); tr._fwd_bwd(lambda t: None); torch.cuda.synchronize()
agg = {}
for e0, e1, kind, M, N, K, epi, sp in rec:
    a = agg.setdefault((kind, M, N, K, epi, sp), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
print(f"{'kind':6s} {'M':>6s} {'N':>5s} {'K':>6s} epi sp   n   avg_us   tot_us  TFLOP/s")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    kind, M, N, K, epi, sp = k
    print(f"{kind:6s} {M:6d} {N:5d} {K:6d} {epi:3d} {sp:3d} {a[0]:3d} {a[1]/a[0]:8.1f} {a[1]:8.1f} {2.0*M*N*K*a[0]/a[1]/1e6:8.1f}")
print("total gemm us", sum(a[1] for a in agg.values()))
