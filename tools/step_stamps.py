"""dev tool: time-line of the CAPTURED training step with no tracer attached.  Stamp kernels (tulip_stamp_realtime: the
100 MHz device clock, one base for all XCDs) are captured into the step's graph at the block boundaries of the backward
chain and around every side-queue launch group; prints when each point was reached in the last replay.
usage: python tools/step_stamps.py [batch=8]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd import ops
from tulip_amd.trainer import Trainer
from tulip_amd.engine import TulipEngine as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
a = argparse.Namespace(model="tulip_base", img=[16, 1024], target=[64, 1024], batch=B)
dev = torch.device("cuda", 0)
buf = torch.zeros(1024, dtype=torch.int64, device=dev)
slots, seen = {}, {}

def stamp(tag):
    k = seen.get(tag, 0); seen[tag] = k + 1
    i = slots.setdefault((tag, k), len(slots))
    ops.stamp_realtime(buf[i:])

def wrap(cls, name, before=None, after=None):
    real = getattr(cls, name)
    def f(self, *a, **kw):
        if before: stamp(before(self, a, kw) if callable(before) else before)
        r = real(self, *a, **kw)
        if after: stamp(after(self, a, kw) if callable(after) else after)
        return r
    setattr(cls, name, f)

real_fwd = E.run_forward
def run_forward(self, *a, **kw):
    seen.clear(); stamp("chain forward begins")
    return real_fwd(self, *a, **kw)
E.run_forward = run_forward
wrap(E, "run_backward", before="chain backward begins", after="chain backward issued (side joined)")
wrap(E, "_block_fwd", after=lambda s, a, kw: f"chain end of block fwd {a[1].prefix}")
wrap(E, "_block_bwd", after=lambda s, a, kw: f"chain end of block bwd {a[1].prefix}")
def _grp(self, a, kw):
    pend = self._pending if (len(a) < 2 or a[1] is None) and kw.get("pending") is None else (kw.get("pending") or a[1])
    w = [x for k, x in pend if k == "w"]
    kinds = "".join(k for k, _ in pend if k != "w")
    contents.append(" ".join(f"{x[4]}x{x[5]}/{x[6]}" for x in w) + (" +" + kinds if kinds else ""))
    return "side  group begins"
contents = []
wrap(E, "_issue_pending", before=_grp, after="side  group ends")
wrap(Trainer, "_adamw", before="chain adamw begins", after="chain adamw ends")

m = bench.make_model(a).to(dev).train()
tr = Trainer(m, B, device=dev)
lo, hi = bench.synthetic(a, 0, dev); tr.load_batch(lo, hi)
for _ in range(20): tr.step()
torch.cuda.synchronize()
v = buf.cpu().tolist()
# (slots the LAST launch sequence -- the captured one -- no longer stamps keep what an earlier eager pass wrote: dropped)
ev = sorted(((v[i], tag, k) for (tag, k), i in slots.items() if v[i] and k < seen.get(tag, 0)), key=lambda e: e[0])
t0 = ev[0][0]
print(f"{len(ev)} stamps; 10-ns ticks since the first one; every stamp is a 1-thread launch of its own (~3-5 us each)")
for t, tag, k in ev:
    print(f"  +{(t - t0) / 100.0:9.1f} us  {tag}" + (f" #{k}" if tag.startswith("side") else "")
          + (f"   [{contents[len(contents) - seen['side  group begins'] + k]}]" if tag == "side  group begins" else ""))

# machine-readable twin for bench.py (`tracer_distortion`): when the side queue's first group begins, measured from the head of the
# backward chain, in the UN-traced captured step
if B == 8 and len(sys.argv) > 2:
    import json
    at = {(tag, k): (t - t0) / 100.0 for t, tag, k in ev}
    b0, s0 = at.get(("chain backward begins", 0)), at.get(("side  group begins", 0))
    json.dump({"source": "tools/step_stamps.py: stamp kernels captured into the step's graph, no tracer attached",
               "source_stamp": bench.kernel_source_stamp(), "step_us": (ev[-1][0] - t0) / 100.0,
               "side_queue_start_after_backward_begins_us": (s0 - b0) if (b0 is not None and s0 is not None) else None},
              open(sys.argv[2], "w"), indent=1)
