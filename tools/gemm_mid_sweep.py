"""dev: time of the 192 x 192 kernel against K at exactly one round of the chip (3072 x 3072 outputs = 256 tiles): fixed cost + cycles per 32-deep step"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd import ops
from tulip_amd._lib import EPI_BF16
dev = torch.device("cuda", 0)
def timed(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for bt in (False, True):
    for (M, N) in [(3072, 3072), (1536, 1536), (3072, 768)]:
        row = f"bT={int(bt)} {M}x{N}:"
        for K in (128, 256, 768, 1536, 3072, 6144):
            A = torch.randn(M, K, device=dev).bfloat16()
            Wt = (torch.randn(K, N, device=dev) if bt else torch.randn(N, K, device=dev)).bfloat16()
            o = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            t = {}
            for mid in (False, True):
                t[mid] = timed(lambda: ops.gemm(A, Wt, M, N, K, lda=K, ldb=N if bt else K, b_trans=bt, epi=EPI_BF16, out=o, ldo=N, mid=mid))
            row += f"  K={K}: {t[False]:.1f} / {t[True]:.1f} us"
        print(row, flush=True)
