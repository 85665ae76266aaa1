"""Build-container tool (needs /root/reference): the reference's own CPU training step and the oracle's, timed side by side on the
same host cores, same config (KITTI tulip_base, batch 8, fp32), same optimizer -- SURVEY 8(d) asks that the stand-in CPU baseline
(`cpu_baseline.kind: "port"`) be shown not to be a slower straw man than the reference itself.  Writes profiles/r6_ref_vs_oracle_cpu.txt.
usage: python tools/ref_vs_oracle_cpu.py [steps=5] [threads=os.cpu_count()]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from oracle import tulip_oracle as O
import make_golden as MG

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
torch.set_num_threads(threads)
B = 8
cfg = O.tulip_base_config(img_size=(16, 1024), target_img_size=(64, 1024))
lo, hi = O.synthetic_batch(cfg, B, seed=1234)


def opt_for(params):
    decay = [p for p in params if p.ndim > 1]
    nodecay = [p for p in params if p.ndim <= 1]
    return torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}], lr=5e-4,
                             betas=(0.9, 0.95))


def time_loop(step):
    ts = []
    for it in range(2 + steps):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    return ts[2:]


T = MG.import_reference()
torch.manual_seed(0)
ref = MG.ref_model(T, cfg, 0.0).train()
ropt = opt_for(list(ref.parameters()))


def ref_step():
    ropt.zero_grad(set_to_none=True)
    _, loss, _ = ref(lo, hi)
    loss.backward()
    ropt.step()


sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
full = dict(sd); full.update(params)
oopt = opt_for(list(params.values()))


def ora_step():
    oopt.zero_grad(set_to_none=True)
    _, loss, _ = O.tulip_forward(full, cfg, lo, hi)
    loss.backward()
    oopt.step()


# interleaved (a noisy shared host: neither side gets the quiet minutes)
rt, ot = [], []
for rnd in range(2):
    rt += time_loop(ref_step)
    ot += time_loop(ora_step)
fmt = lambda v: f"median {statistics.median(v):.2f} s, min {min(v):.2f}, max {max(v):.2f}  ({B / statistics.median(v):.2f} img/s)"
cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
out = (f"KITTI tulip_base 16x1024 -> 64x1024, batch {B}, fp32, fwd + L1 + bwd + torch AdamW, {threads} threads ({cpu}), "
       f"{2 * steps} timed steps each, two interleaved rounds, DropPath rate 0 on both sides\n"
       f"reference (/root/reference/tulip/model/tulip.py, imported with the two stubs): {fmt(rt)}\n"
       f"oracle    (oracle/tulip_oracle.py, what bench.py's cpu_baseline times):        {fmt(ot)}\n"
       f"ratio oracle / reference (median step time): {statistics.median(ot) / statistics.median(rt):.2f}\n")
print(out)
open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r6_ref_vs_oracle_cpu.txt"), "w").write(out)
