"""dev tool: the oracle's plain-PyTorch op sequence run ON THE GPU (what stock PyTorch-ROCm eager achieves
for the same training step) -- context number only, not part of the product or the bench contract."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tulip_oracle as O
import numpy as np
dev = torch.device("cuda", 0)
cfg = O.tulip_base_config()
sd = {k: v.to(dev) for k, v in O.key_seeded_state_dict(cfg, seed=0, randomize_affine=False).items()}
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
full = dict(sd); full.update(params)
opt = torch.optim.AdamW([{"params": [p for p in params.values() if p.ndim > 1], "weight_decay": 0.01},
                         {"params": [p for p in params.values() if p.ndim <= 1], "weight_decay": 0.0}], lr=5e-4, betas=(0.9, 0.95), fused=True)
B = 8
lo, hi = O.synthetic_batch(cfg, B); lo, hi = lo.to(dev), hi.to(dev)
# the oracle builds index tensors with numpy on the host each call: cache them on device (fair to PyTorch)
_wti, _sam = O.window_token_index, O.shift_attention_mask
cache = {}
def wti(*a):
    if a not in cache: cache[a] = _wti(*a)
    return cache[a]
O.window_token_index = wti
for mode in ("fp32", "bf16-autocast"):
    ts = []
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode != "fp32")):
            _, loss, _ = O.tulip_forward(full, cfg, lo, hi)
        loss.backward(); opt.step()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts[2:])[len(ts[2:]) // 2]
    print(f"PyTorch-ROCm eager ({mode}) oracle training step B={B}: {t*1e3:.1f} ms -> {B/t:.1f} img/s", flush=True)
