"""Timing only: the batch-8 step with the end-of-step rewrite of the fragment-major weight copies SKIPPED (the copies go stale: wrong
numerics, same kernels otherwise) -- the ceiling of anything that moves those 171 MB out of the end of the step (e.g. the optimizer
write-outs emitting the packed layouts themselves).  Interleaved with the normal step.
usage: python tools/exp_nopack.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd import ops
from tulip_amd.trainer import Trainer
from tulip_amd.model.tulip import tulip_base

DEV = "cuda"
B = 8


def step_ms(skip, steps=300):
    torch.manual_seed(0)
    m = tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
                   pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).to(DEV).train()
    tr = Trainer(m, B, lr=5e-4, weight_decay=0.01)
    x = torch.rand(B, 1, 16, 1024, device=DEV); y = torch.rand(B, 1, 64, 1024, device=DEV)
    tr.load_batch(x, y)
    if skip:
        real = ops.pack_bf16_multi
        ops.pack_bf16_multi = lambda items, n: None
    try:
        for _ in range(20):
            tr.step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            tr.step()
        b.record(); b.synchronize()
    finally:
        if skip:
            ops.pack_bf16_multi = real
    return a.elapsed_time(b) / steps


for r in range(3):
    for skip in (False, True):
        print(f"weight-copy rewrite {'skipped' if skip else 'as is  '}: {step_ms(skip):.4f} ms", flush=True)
