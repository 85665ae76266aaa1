"""dev tool: isolated timing of the mid-size forward / data-gradient GEMM shapes (stage 3 at batch 64, stages 3-4 of tulip_large)
(tile height chosen by csrc/gemm.hip launch()).  usage: python tools/gemm_big.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd import ops
from tulip_amd._lib import EPI_BF16, EPI_F32, load
dev = torch.device("cuda", 0)
shapes = [(4096, 2304, 768, False), (4096, 3072, 768, False), (4096, 768, 3072, False), (4096, 768, 768, False),
          (4096, 768, 2304, True), (4096, 768, 3072, True), (4096, 3072, 768, True), (2048, 6144, 1536, False),
          (2048, 1536, 6144, True), (16384, 384, 768, False), (16384, 1152, 384, False), (8192, 2304, 768, False)]
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print(f"{'M':>6s} {'N':>5s} {'K':>5s} bT |       us  TFLOP/s")
for M, N, K, bt in shapes:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    Wt = (torch.randn(K, N, device=dev, generator=g) if bt else torch.randn(N, K, device=dev, generator=g)).bfloat16()
    o = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    t = timed(lambda: ops.gemm(A, Wt, M, N, K, lda=K, ldb=N if bt else K, b_trans=bt, epi=EPI_BF16, out=o, ldo=N))
    print(f"{M:6d} {N:5d} {K:5d} {int(bt):2d} | {t:8.1f} {2.0 * M * N * K / t / 1e6:8.0f}")
