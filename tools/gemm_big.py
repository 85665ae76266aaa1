"""dev tool: isolated timing of the mid-size forward / data-gradient GEMM shapes (stage 3 at batch 64, stages 3-4 of tulip_large):
the 64 / 128 / 256 x 96 tiles of gemm_tile (TULIP_GEMM_NO_MID) against the 192 x 192 loader-wave kernel (TULIP_GEMM_MID, with the
K split that fills the chip where the output has few tiles).  usage: python tools/gemm_big.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd import ops
from tulip_amd._lib import EPI_BF16, EPI_F32, load
dev = torch.device("cuda", 0)
shapes = [(4096, 2304, 768, False), (4096, 3072, 768, False), (4096, 768, 3072, False), (4096, 768, 768, False),
          (4096, 768, 2304, True), (4096, 768, 3072, True), (4096, 3072, 768, True), (2048, 3072, 768, False), (2048, 6144, 1536, False),
          (2048, 1536, 6144, True), (2048, 1536, 6144, False), (1024, 6144, 1536, False), (1024, 1536, 6144, False),
          (16384, 384, 768, False), (16384, 1152, 384, False), (8192, 2304, 768, False), (8192, 768, 3072, False)]
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print(f"{'M':>6s} {'N':>5s} {'K':>5s} bT | gemm_tile us  TFLOP/s | 192x192 us  TFLOP/s (bits equal) | best 192x192 with a K split: splits us TFLOP/s")
for M, N, K, bt in shapes:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    Wt = (torch.randn(K, N, device=dev, generator=g) if bt else torch.randn(N, K, device=dev, generator=g)).bfloat16()
    o0 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    o1 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.zeros(8 * M * N, device=dev)
    run = lambda o, mid, sp=1: ops.gemm(A, Wt, M, N, K, lda=K, ldb=N if bt else K, b_trans=bt, epi=EPI_BF16, out=o, ldo=N, mid=mid,
                                        splits=sp, workspace=ws.data_ptr(), workspace_bytes=ws.numel() * 4)
    t0 = timed(lambda: run(o0, False))
    t1 = timed(lambda: run(o1, True))
    eq = torch.equal(o0, o1)
    fl = 2.0 * M * N * K
    best = None
    for sp in (2, 3, 4, 6, 8):
        if K % (64 * sp):
            continue
        t = timed(lambda: run(o1, True, sp))
        if best is None or t < best[1]:
            best = (sp, t)
    ref = (A.float() @ (Wt.float() if bt else Wt.float().t()))
    err = ((o1.float() - ref).norm() / ref.norm()).item()
    print(f"{M:6d} {N:5d} {K:5d} {int(bt):2d} | {t0:8.1f} {fl / t0 / 1e6:8.0f} | {t1:8.1f} {fl / t1 / 1e6:8.0f} ({eq}) | "
          f"{best[0]} {best[1]:8.1f} {fl / best[1] / 1e6:8.0f}   rel err vs fp32 {err:.1e}", flush=True)
