"""Review item 3's measurement: two consecutive C = 96 blocks as ONE launch with tile-to-tile hand-off (tulip_swin96_pair_fwd)
against the two launches, isolated (HIP events, back to back, the training form with everything saved) at the KITTI stage-0 shape,
then the whole training step with the pairs on / off (TulipEngine.pair96), interleaved.
Kill criterion stated before the measurement: keep only if the isolated pair is <= 54 us AND the step gains >= 10 us.
usage: python tools/bench_pair96.py [batch=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from tulip_amd import ops
import test_pair96_gpu as T

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def timeit(fn, reps=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


eng = T._engine(B)
P = eng.plan(B)
for label, specs in (("encoder stage 0", eng.enc_blocks[0][:2]), ("decoder level 0", eng.dec_blocks[-1][:2])):
    if not specs:
        continue
    xin = torch.randn(B * specs[0].H * specs[0].W, 96, device=DEV)
    eng.draw_drop_scales(P, True, torch.rand(eng.n_drop_slots, B, device=DEV))
    for save in (True, False):
        bufs = T._fresh(P, specs)
        d0, d1 = T._descs(eng, P, specs, xin, bufs, save)
        sync = torch.zeros(ops.swin96_pair_sync_bytes(B, specs[0].H, specs[0].W) // 4, dtype=torch.int32, device=DEV)
        def two():
            ops.swin96_block_fwd(**d0)
            ops.swin96_block_fwd(**d1)
        t2 = [timeit(two) for _ in range(3)]
        t1 = [timeit(lambda: ops.swin96_pair_fwd(d0, d1, sync)) for _ in range(3)]
        one = timeit(lambda: ops.swin96_block_fwd(**d0))
        print(f"batch {B} {label} ({'training form' if save else 'inference form'}), us: two launches {min(t2):6.2f} "
              f"(one block alone {one:6.2f}), pair launch {min(t1):6.2f}   [{' '.join(f'{v:.1f}' for v in t2)} | {' '.join(f'{v:.1f}' for v in t1)}]", flush=True)

if len(sys.argv) > 2 and sys.argv[2] == "iso":
    sys.exit(0)
# the whole step, interleaved
from tulip_amd.trainer import Trainer
from tulip_amd.model.tulip import tulip_base


def step_ms(pair, steps=200):
    torch.manual_seed(0)
    m = tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
                   pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).to(DEV).train()
    m.engine().pair96 = pair
    tr = Trainer(m, B, lr=5e-4, weight_decay=0.01)
    x = torch.rand(B, 1, 16, 1024, device=DEV); y = torch.rand(B, 1, 64, 1024, device=DEV)
    for _ in range(20):
        loss = tr.step(x, y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        loss = tr.step(x, y)
    b.record(); b.synchronize()
    return a.elapsed_time(b) / steps, float(loss[0])


rows = []
for r in range(3):
    for pair in (False, True):
        ms, loss = step_ms(pair)
        rows.append((pair, ms, loss))
        print(f"step batch {B}: pairs {'on ' if pair else 'off'} {ms:.4f} ms  loss {loss:.6f}", flush=True)
off = min(ms for p, ms, _ in rows if not p); on = min(ms for p, ms, _ in rows if p)
print(f"step: off {off:.4f} ms, on {on:.4f} ms, gain {1e3 * (off - on):+.1f} us")
