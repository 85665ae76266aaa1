"""dev experiment: deep-stage GEMM shapes, time of (GEMM + fold) vs split count, graph-replayed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd import ops
from tulip_amd._lib import EPI_BF16, EPI_RESID_F32
dev = "cuda"
shapes = [("fwd", 512, 2304, 768), ("fwd", 512, 768, 768), ("fwd", 512, 3072, 768), ("fwd", 512, 768, 3072),
          ("fwd", 2048, 1152, 384), ("fwd", 2048, 384, 384), ("fwd", 2048, 1536, 384), ("fwd", 2048, 384, 1536),
          ("dgrad", 512, 768, 2304), ("dgrad", 2048, 384, 1152), ("fwd", 8192, 576, 192), ("fwd", 8192, 192, 768)]
ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
for kind, M, N, K in shapes:
    A = torch.randn(M, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if kind == "fwd":
        B = torch.randn(N, K, device=dev).bfloat16(); kw = dict(lda=K, ldb=K)
    else:
        B = torch.randn(K, N, device=dev).bfloat16(); kw = dict(lda=K, ldb=N, b_trans=True)
    res = []
    for sp in (1, 2, 3, 4, 6, 8, 12):
        def run():
            ops.gemm(A, B, M, N, K, epi=EPI_BF16, out=out, splits=sp, workspace=ws, workspace_bytes=ws.numel() * 4, **kw)
        for _ in range(3): run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): run()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        res.append((sp, e0.elapsed_time(e1) / 20 * 1e3))
    print(f"{kind:5s} {M:5d} {N:5d} {K:5d} ksub_grid={os.environ.get('TULIP_GEMM_KSUB_GRID', '192'):>6s} " +
          "  ".join(f"sp{sp}:{us:5.1f}" for sp, us in res))
