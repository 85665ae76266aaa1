"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on kernels with KNOWN HBM byte counts (MI355X_MICROARCH.md, HBM
section: FETCH_SIZE reports half of a wide streaming read on gfx950; other widths and WRITE_SIZE are uncalibrated).
Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv` and again with WRITE_SIZE; tools/pmc_summary.py
prints MB per call, this script prints the expected MB per call.  Buffers are sized past the 256 MiB Infinity Cache.
  cast_flat_kernel : reads 4 B, writes 2 B per element (16-B loads)            n = 2^28
  adamw_kernel     : reads 16 B (p, g, m, v), writes 18 B (p, m, v, g=0, bf16) n = 2^27
  gemm_kernel      : the GEMM's own access pattern (16-B chunks of 64-B rows): A [M][K] bf16 read once, B [N][K]
                     re-read per row tile (cache-resident), C [M][N] bf16 written once;  M = 2^21, N = 96, K = 96
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tulip_amd import ops

dev = "cuda"
n = 1 << 28
x = torch.randn(n, device=dev)
y = torch.empty(n, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.cast_flat(x, y, n)
print(f"cast_flat_kernel: expect fetch {4 * n / 2**20:.0f} MB, write {2 * n / 2**20:.0f} MB per call")
del x, y
n = 1 << 27
p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
sh = torch.empty(n, device=dev, dtype=torch.bfloat16)
hyper = torch.tensor([1e-3, 0.9, 0.95, 1e-8, 0.01, 0.1, 0.05, 1.0], device=dev)
for _ in range(3):
    ops.adamw(p, g, m, v, sh, n, hyper, None, zero_grad=True)
print(f"adamw_kernel: expect fetch {16 * n / 2**20:.0f} MB, write {18 * n / 2**20:.0f} MB per call")
del p, g, m, v, sh
M, N, K = 1 << 21, 96, 96
A = torch.randn(M, K, device=dev).bfloat16()
B = torch.randn(N, K, device=dev).bfloat16()
Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(A, B, M, N, K, lda=K, ldb=K, epi=ops.EPI_BF16, out=Cc)
print(f"gemm_kernel<128...>: expect fetch {2 * M * K / 2**20:.0f} MB (+ weights), write {2 * M * N / 2**20:.0f} MB per call")
torch.cuda.synchronize()
