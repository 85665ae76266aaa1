"""Same-box A/B of the batch-8 step with the stage-boundary GEMMs in their small-K form on the fragment-major weight copies
(TulipEngine.packed_gemm, csrc/gemm.hip gemm_stream_kernel) on / off, interleaved; identical losses expected (same bits).
usage: python tools/ab_packed_gemm.py [batch=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd.engine import TulipEngine
from tulip_amd.trainer import Trainer
from tulip_amd.model.tulip import tulip_base

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def step_ms(on, steps=300):
    TulipEngine.packed_gemm = on
    torch.manual_seed(0)
    m = tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
                   pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).to(DEV).train()
    tr = Trainer(m, B, lr=5e-4, weight_decay=0.01)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.rand(B, 1, 16, 1024, generator=g).to(DEV); y = torch.rand(B, 1, 64, 1024, generator=g).to(DEV)
    tr.load_batch(x, y)
    for _ in range(20):
        loss = tr.step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        loss = tr.step()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / steps, float(loss[0])


rows = []
for r in range(4):
    for on in (False, True):
        ms, loss = step_ms(on, 300 if B <= 16 else 60)
        rows.append((on, ms))
        print(f"batch {B}: small-K form {'on ' if on else 'off'} {ms:.4f} ms   loss {loss:.7f}", flush=True)
off = sum(ms for o, ms in rows if not o) / 4; on = sum(ms for o, ms in rows if o) / 4
print(f"mean off {off:.4f} ms, on {on:.4f} ms: {1e3 * (off - on):+.1f} us")
