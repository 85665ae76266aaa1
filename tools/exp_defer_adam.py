"""dev, TIMING ONLY (the training it runs is wrong): what would deferring the optimizer step of the deep / wide stages out of the
backward's weight-gradient write-outs into the idle side queue beside the NEXT forward be worth?
  A  today's step
  B  the chosen tensors are NOT stepped in their write-outs (their gradients are stored: 4 B instead of 26 B per parameter beside the
     backward) and by nobody else either -- the upper bound of what the backward can gain
  C  B + an AdamW pass over those ranges on the side stream at the head of the forward (+ the weight-copy refresh of the deep
     stage behind it), joined in front of the first stage that reads them -- the whole deferred form, minus correctness
usage: python tools/exp_defer_adam.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch
import bench
from tulip_amd import ops
from tulip_amd.trainer import Trainer
from tulip_amd.engine import TulipEngine

args = argparse.Namespace(model="tulip_base", img=[16, 1024], target=[64, 1024], batch=8)
dev = torch.device("cuda", 0)

def run(tag, defer_prefixes=(), side_adam=False):
    model = bench.make_model(args).to(dev).train()
    real_plan = Trainer._plan_fused_adamw
    ranges = []
    def plan(self, eligible):
        W = self.eng.params
        gbase = self.g.data_ptr()
        keep = {}
        for ptr, cnt in eligible.items():
            a = (ptr - gbase) // 4
            name = next((n for n in W.names if W.offset[n] == a), None)
            if name is not None and name.startswith(tuple(defer_prefixes)) and defer_prefixes:
                ranges.append((a, cnt))
                continue
            keep[ptr] = cnt
        real_plan(self, keep)
        if defer_prefixes and self._adam_blocks is not None:
            # ... and nobody steps them at the end either (timing only)
            blk = self._adam_blocks.cpu().tolist()
            drop = set()
            for a, cnt in ranges:
                drop.update(range(a // 64, (a + cnt + 63) // 64))
            self._adam_blocks = torch.tensor([b for b in blk if b not in drop], dtype=torch.int32, device=dev)
    Trainer._plan_fused_adamw = plan
    real_fwd = TulipEngine.run_forward
    if side_adam:
        def fwd(self, P, **kw):
            tr = self._tr
            if ranges and getattr(tr, "_segments_warm", False):
                lo, hi = min(a for a, _ in ranges), max(a + c for a, c in ranges)
                ev = torch.cuda.Event(); ev.record()
                st = self._side_stream
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    W = self.params
                    ops.adamw(W.base32 + 4 * lo, tr.g.data_ptr() + 4 * lo, tr.m.data_ptr() + 4 * lo, tr.v.data_ptr() + 4 * lo,
                              W.base16 + 2 * lo, hi - lo, tr.hyper, W.decay_mask.data_ptr() + lo // 64, zero_grad=False)
                self._exp_join = True
            return real_fwd(self, P, **kw)
        TulipEngine.run_forward = fwd
        real_stage = TulipEngine._stage_fwd
        def stage(self, P, specs, xin, out_bf16=None):
            if getattr(self, "_exp_join", False) and specs[0].C >= 384:
                torch.cuda.current_stream().wait_stream(self._side_stream)
                self._exp_join = False
            return real_stage(self, P, specs, xin, out_bf16)
        TulipEngine._stage_fwd = stage
    try:
        tr = Trainer(model, args.batch, device=dev)
        tr.eng._tr = tr
        tr._segments_warm = True
        lo, hi = bench.synthetic(args, 0, dev); tr.load_batch(lo, hi)
        for _ in range(15): tr.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): tr.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 10
        n = sum(c for _, c in ranges)
        print(f"{tag:70s} {dt:.4f} ms/step   deferred parameters {n / 1e6:.1f} M", flush=True)
    finally:
        Trainer._plan_fused_adamw = real_plan
        TulipEngine.run_forward = real_fwd
        if side_adam:
            TulipEngine._stage_fwd = real_stage

DEEP = ("layers.3.", "first_patch_expanding.", "layers.2.downsample.")
WIDE = DEEP + ("layers.2.", "layers_up.0.")
for rep in range(2):
    run("A  today")
    run("B  deep stage (C = 768) not stepped at all", DEEP)
    run("C  deep stage stepped beside the next forward", DEEP, side_adam=True)
    run("B' deep + C = 384 stages not stepped at all", WIDE)
    run("C' deep + C = 384 stages stepped beside the next forward", WIDE, side_adam=True)
