"""Isolated timing of the stage-boundary GEMMs of the batch-8 step in their plain form and in the small-K form on the fragment-major
weight copy (TULIP_GEMM_B_PACKED, csrc/gemm.hip gemm_stream_kernel); 20 launches per HIP-graph replay, as tools/chain_gemms.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from tulip_amd import ops
from test_gemm_packed_gpu import packed

DEV = "cuda"


def timed(call, reps=20):
    call(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            call()
    best = 1e9
    for _ in range(5):
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


print(f"{'M':>6} {'N':>5} {'K':>5} bT spl   plain us  packed us")
tot = [0.0, 0.0]
for M, N, K, bt, splits in [(512, 1536, 768, 0, 1), (2048, 384, 768, 0, 1), (2048, 384, 384, 1, 1), (512, 768, 1536, 1, 4),
                            (512, 1536, 768, 1, 1), (32768, 96, 96, 1, 1)]:
    A = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
    W = (torch.randn(K, N, device=DEV) * 0.05).bfloat16() if bt else (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
    Wp = packed(ops, W, transpose=bool(bt))
    o = torch.zeros(M, N, device=DEV)
    ws = torch.zeros(splits * M * N, device=DEV)
    # a cold-ish weight: 256 MB of other traffic between the launches would be the honest form; the graph replays back to back
    # like chain_gemms.py so that the two tools agree
    t0 = timed(lambda: ops.gemm(A, W, M, N, K, lda=K, ldb=(N if bt else K), b_trans=bool(bt), epi=ops.EPI_F32, out=o, splits=splits,
                                workspace=ws, workspace_bytes=ws.numel() * 4))
    t1 = timed(lambda: ops.gemm(A, Wp, M, N, K, lda=K, ldb=K, epi=ops.EPI_F32, out=o, splits=splits, workspace=ws,
                                workspace_bytes=ws.numel() * 4, b_packed=True))
    extra = ""
    if splits > 1 and ops.gemm_packed_supported(M, N, K, 1):
        t2 = timed(lambda: ops.gemm(A, Wp, M, N, K, lda=K, ldb=K, epi=ops.EPI_F32, out=o, b_packed=True))
        extra = f"   packed, unsplit {t2:6.2f}"
        t1 = t2
    print(f"{M:6d} {N:5d} {K:5d} {bt:2d} {splits:3d} {t0:10.2f} {t1:10.2f}{extra}", flush=True)
    tot[0] += t0; tot[1] += t1
print(f"six launches: plain {tot[0]:.1f} us, packed {tot[1]:.1f} us")
