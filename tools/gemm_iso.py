"""dev tool: every distinct GEMM call of one training step replayed in isolation (30 back-to-back launches,
HIP events): avg us, algorithmic bytes, GB/s and TFLOP/s per shape, sorted by time per step."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd import ops
from tulip_amd.trainer import Trainer
args = argparse.Namespace(model=os.environ.get("MODEL", "tulip_base"), img=[16, 1024], target=[64, 1024],
                          batch=int(os.environ.get("B", 8)))
dev = torch.device("cuda", 0)
model = bench.make_model(args).to(dev).train()
tr = Trainer(model, args.batch, device=dev, use_graph=False)
lo, hi = bench.synthetic(args, 0, dev); tr.load_batch(lo, hi)
calls = []; real = ops.gemm
def rec(A, B, M, N, K, **kw):
    calls.append((A, B, M, N, K, dict(kw))); real(A, B, M, N, K, **kw)
ops.gemm = rec
tr._fwd_bwd(lambda t: None); torch.cuda.synchronize()
calls.clear(); tr._fwd_bwd(lambda t: None); torch.cuda.synchronize()
ops.gemm = real
groups = {}
for A, B, M, N, K, kw in calls:
    kind = "wgrad" if kw.get("a_trans") else ("dgrad" if kw.get("b_trans") else "fwd")
    key = (kind, M, N, K, kw.get("epi", 0), kw.get("splits", 1), bool(kw.get("accumulate")))
    groups.setdefault(key, []).append((A, B, M, N, K, kw))
EPI_OUT_BYTES = {0: 2, 1: 4, 2: 2, 3: 4, 4: 4, 5: 4, 6: 4, 7: 4}   # per output element (dual = 2x bf16)
rows = []
for key, lst in groups.items():
    A, B, M, N, K, kw = lst[0]
    for _ in range(3): real(A, B, M, N, K, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(30): real(A, B, M, N, K, **kw)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    epi = key[4]
    byt = 2 * (M * K + N * K) + M * N * EPI_OUT_BYTES.get(epi, 4) * (key[5] if epi == 7 else 1)
    if epi in (2,): byt += 2 * M * N          # gelu-bwd reads the saved pre-activation
    if epi in (4,): byt += 4 * M * N          # residual read
    rows.append((us * len(lst), key, len(lst), us, byt))
tot = sum(r[0] for r in rows)
print(f"{'kind':6s} {'M':>6s} {'N':>5s} {'K':>6s} epi sp acc   n   avg_us   tot_us   GB/s  TFLOP/s")
for t, key, n, us, byt in sorted(rows, key=lambda r: -r[0]):
    kind, M, N, K, epi, sp, acc = key
    print(f"{kind:6s} {M:6d} {N:5d} {K:6d} {epi:3d} {sp:3d} {int(acc):3d} {n:3d} {us:8.1f} {t:8.1f} {byt/us/1e3:6.0f} {2.0*M*N*K/us/1e6:8.1f}")
print(f"total isolated gemm us per step {tot:.1f} over {sum(r[2] for r in rows)} launches")
