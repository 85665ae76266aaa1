"""dev tool: how much the weight-gradient kernel would gain from operands whose 32-token x 192-feature tiles are
contiguous in memory.  Same tile count, same tokens per workgroup, same MFMA work; (a) both operands are 192 features
wide (every tile spans full rows: contiguous), (b) one operand is a 192-wide slice of 768-wide rows, (c) both are."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd import ops
dev = torch.device("cuda:0")
ws = torch.empty(40 << 20, device=dev)

def timeit(items, reps=20):
    for _ in range(3): ops.wgrad_group(items, [], ws, ws.numel() * 4, fold=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.wgrad_group(items, [], ws, ws.numel() * 4, fold=False)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for tokens_per_wg in (512, 1024, 4096):
    out = []
    for name, Nw, Kw in (("both contiguous (192 x 192 of 192 / 192)", 192, 192), ("dY sliced (768 x 192)", 768, 192),
                         ("X sliced (192 x 768)", 192, 768), ("both sliced (768 x 768)", 768, 768)):
        tiles = (Nw // 192) * (Kw // 192)
        splits = 256 // tiles
        tok = tokens_per_wg * splits
        dY = (torch.randn(tok, Nw, device=dev) * 0.5).bfloat16(); X = torch.randn(tok, Kw, device=dev).bfloat16()
        dW = torch.zeros(Nw, Kw, device=dev)
        t = timeit([ops.wgrad_item(dY, Nw, X, Kw, Nw, Kw, tok, dW, None, splits)])
        out.append(f"{name}: {t:6.1f} us ({2.0 * tok * Nw * Kw / t / 1e6:5.0f} TF/s)")
        del dY, X
    print(f"{tokens_per_wg} tokens per workgroup, 256 workgroups | " + " | ".join(out), flush=True)
