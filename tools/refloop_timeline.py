"""dev tool / profiles: host + device timeline of ONE step of the reference's unchanged loop body on the drop-in module
(bench.reference_loop = engine_upsampling.py:69-100 + util/misc.py:292-305).  Host seconds per section from perf_counter with a
synchronize at each section boundary (so device time is charged to the section that launched it), then the same loop timed
WITHOUT the extra synchronizes, and a torch.profiler kernel summary of the optimizer / scaler part.
usage: python tools/refloop_timeline.py [steps=30]"""
import os, sys, time, argparse, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
args = argparse.Namespace(model="tulip_base", img=[16, 1024], target=[64, 1024], batch=8)
dev = torch.device("cuda", 0)
model = bench.make_model(args).to(dev).train()
decay = [p for p in model.parameters() if p.ndim > 1]
no_decay = [p for p in model.parameters() if p.ndim <= 1]
opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.01}], lr=5e-4, betas=(0.9, 0.95))
scaler = torch.amp.GradScaler("cuda")
lo, hi = bench.synthetic(args, 0, dev)
opt.zero_grad()
names = ["forward (autocast, model call)", "loss.item() x2", "scale(loss).backward()", "unscale_", "scaler.step (found_inf sync + AdamW)",
         "scaler.update", "zero_grad", "synchronize"]


def one(acc=None):
    def mark(i, t0):
        if acc is not None:
            torch.cuda.synchronize()
            acc[i] += time.perf_counter() - t0
        return time.perf_counter()
    t = time.perf_counter()
    with torch.autocast("cuda"):
        _, total_loss, pixel_loss = model(lo, hi, eval=False)
    t = mark(0, t)
    total_loss.item(); pixel_loss.item()
    t = mark(1, t)
    scaler.scale(total_loss).backward()
    t = mark(2, t)
    scaler.unscale_(opt)
    t = mark(3, t)
    scaler.step(opt)
    t = mark(4, t)
    scaler.update()
    t = mark(5, t)
    opt.zero_grad()
    t = mark(6, t)
    torch.cuda.synchronize()
    mark(7, t)


for _ in range(8):
    one()
torch.cuda.synchronize()
acc = [0.0] * len(names)
for _ in range(steps):
    one(acc)
print(f"reference loop body on the drop-in, {steps} steps, a synchronize at every section boundary:")
for n, a in zip(names, acc):
    print(f"  {n:44s} {a / steps * 1e3:8.3f} ms")
print(f"  {'sum':44s} {sum(acc) / steps * 1e3:8.3f} ms")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    one()
torch.cuda.synchronize()
print(f"the loop as the reference runs it (its own three syncs only): {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
# host-only cost of the python side: the same calls with the device idle in between are what the syncs expose
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        one()
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(((e.key, e.count, e.self_device_time_total / 5, e.self_cpu_time_total / 5) for e in ev), key=lambda r: -r[2])
print("per step, by device time (us) | host self time (us):")
for k, c, d, h in rows[:28]:
    print(f"  {k[:70]:70s} n={c / 5:6.1f}  dev {d:8.1f}  host {h:8.1f}")
print(f"  total device {sum(r[2] for r in rows):.0f} us, total host self {sum(r[3] for r in rows):.0f} us per step")
