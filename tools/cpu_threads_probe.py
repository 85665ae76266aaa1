import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tulip_oracle as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
cfg = O.tulip_base_config()
sd = O.key_seeded_state_dict(cfg, seed=0, randomize_affine=False)
B = 2
lo, hi = O.synthetic_batch(cfg, B)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    full = dict(sd); full.update(params)
    ts = []
    for it in range(2):
        t0 = time.perf_counter()
        _, loss, _ = O.tulip_forward(full, cfg, lo, hi)
        loss.backward()
        ts.append(time.perf_counter() - t0)
    print(f"threads {nt:3d}: fwd+bwd B={B}: {ts[-1]:.2f} s  -> {B/ts[-1]:.2f} img/s", flush=True)
