"""Data-parallel gradient exchange for the flat gradient buffer (reference: DistributedDataParallel
at main_lidar_upsampling.py:277 over NCCL, misc.py:279).

One process per GPU; gradients live in ONE flat fp32 buffer ordered by backward completion
(tulip_amd.engine.FlatParams), so a bucket is a contiguous slice and is all-reduced (SUM, RCCL over
xGMI via torch.distributed backend "nccl") as soon as the backward has *launched* its last
contribution -- the collective then overlaps with the rest of the backward on RCCL's own stream.
The 1/world_size of DDP's mean is folded into the fused AdamW (hyper[7]).  No per-parameter hooks,
no gradient copies.  Works unchanged on CPU tensors with the gloo backend (tests/test_ddp_cpu.py).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def plan_buckets(groups: Sequence[Tuple[str, int]], total: int, min_elems: int) -> List[Tuple[str, int, int]]:
    """Merge consecutive completion groups into buckets of >= min_elems elements.
    Returns [(tag_that_completes_it, start, end)], covering [0,total) exactly once, in order."""
    out: List[Tuple[str, int, int]] = []
    start = 0
    for k, (tag, end) in enumerate(groups):
        last = k == len(groups) - 1
        if last:
            end = total
        if end - start >= min_elems or last:
            if end > start:
                out.append((tag, start, end))
            start = end
    if out and out[-1][2] != total:  # pragma: no cover
        raise AssertionError("bucket plan does not cover the buffer")
    return out


class _Done:
    def wait(self):
        return True


class GradBucketer:
    def __init__(self, groups: Sequence[Tuple[str, int]], total: int, bucket_mb: float = 16.0,
                 process_group: Optional[dist.ProcessGroup] = None, force: bool = False):
        """force: issue the collectives even in a one-rank group (exercises the N>1 code path -- graph segments,
        stream-ordered work.wait(), per-bucket optimizer -- on a single GPU; needs an initialised process group).
        `dry` (attribute): keep the step structure but skip the collectives themselves -- bench.py's measurement of
        what the exchange costs beyond the cuts."""
        self.dry = False
        self.buckets = plan_buckets(groups, total, int(bucket_mb * (1 << 20) / 4))
        self.by_tag = {tag: (a, b) for tag, a, b in self.buckets}
        self.pg = process_group
        self.pending: List = []
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())

    def on_group_done(self, tag: str, gflat: torch.Tensor, keep: bool = True):
        """Call after the launches producing group `tag` have been enqueued on the current stream.
        Returns (work, start, end) of the bucket's all-reduce, or None.  keep=False: the caller waits on the work
        itself (per-bucket optimizer), wait_all() will not.  gflat: the buffer the collective runs on -- the fp32
        gradients, or their bf16 copy (Trainer(grad_dtype="bf16"))."""
        if not self.active or tag not in self.by_tag:
            return None
        a, b = self.by_tag[tag]
        if self.dry:
            return _Done(), a, b
        work = dist.all_reduce(gflat[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        if keep:
            self.pending.append(work)
        return work, a, b

    def prime(self, gflat: torch.Tensor) -> None:
        """Every bucket's all-reduce once, eagerly, on the buffer and at the sizes the step will use, and waited for: the
        communicator and whatever the backend sets up on the first use of an algorithm / message size exist before the step is
        captured into a graph (a first collective INSIDE a capture would have to allocate and connect there).  The caller restores
        the buffer's contents."""
        if not self.active or self.dry:
            return
        for _tag, a, b in self.buckets:
            dist.all_reduce(gflat[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True).wait()

    def wait_all(self) -> None:
        for w in self.pending:
            w.wait()          # stream-level dependency for NCCL/RCCL; blocking for gloo
        self.pending.clear()


# ---------------------------------------------------------------------------------------------- sharded optimizer exchange
def shard_bounds(a: int, b: int, world: int, rank: int, align: int = 64) -> Tuple[int, int, int, int]:
    """Bucket [a, b) (a, b multiples of `align`) -> (s, lo, hi, me): `world` shards of s elements (a multiple of `align`) cover
    [a, me), rank r owns [a + r s, a + (r + 1) s) = [lo, hi) for this rank; [me, b) is the remainder (< world * align elements)
    that every rank keeps whole."""
    s = ((b - a) // (world * align)) * align
    return s, a + rank * s, a + (rank + 1) * s, a + world * s


class ShardedExchange:
    """Optional exchange plan (Trainer(exchange="sharded"); off by default, never measured on more than one GPU): per bucket
      reduce-scatter of the fp32 gradients  ->  AdamW on the OWNED 1/N shard (fp32 master + moments sharded: optimizer HBM traffic
      / N)  ->  all-gather of the bf16 shadow the kernels read
    instead of an all-reduce and a full optimizer step on every rank: 4 + 2 instead of 4 + 4 wire bytes per parameter.
    The kernels read ~1 % of the parameters as FP32 straight from the master buffer (biases, LayerNorm vectors, bias tables, the
    patch-embedding / decoder convolutions: `replicated`, a list of (start, end) element ranges inside the flat buffer, recorded
    by FlatParams.p32): those stay replicated -- their gradients are all-reduced through one small staging buffer per bucket and
    every rank steps them -- so that only bf16 has to travel back.  Works on CPU tensors with gloo (tests/test_ddp_cpu.py).
    After a step the fp32 master and the moments of a rank are current on its shards and on the replicated ranges only:
    gather_state() before anything reads them whole (checkpoints, state_dict, a replica check)."""

    def __init__(self, buckets: Sequence[Tuple[str, int, int]], replicated: Sequence[Tuple[int, int]], device,
                 process_group: Optional[dist.ProcessGroup] = None, align: int = 64):
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.align = align
        self.plan = {}
        smax = 0
        for tag, a, b in buckets:
            s, lo, hi, me = shard_bounds(a, b, self.world, self.rank, align)
            # replicated ranges of this bucket, clipped to its sharded part (the remainder is replicated anyway)
            rep = [(max(x, a), min(y, me)) for x, y in replicated if x < me and y > a]
            idx = (torch.cat([torch.arange(x, y, dtype=torch.int64) for x, y in rep]) if rep else
                   torch.zeros(0, dtype=torch.int64)).to(device)
            # 64-element blocks this rank steps: its shard, the replicated ranges, the remainder
            blk = set(range(lo // align, hi // align)) | set(range(me // align, -(-b // align)))
            for x, y in rep:
                blk |= set(range(x // align, -(-y // align)))
            blocks = torch.tensor(sorted(blk), dtype=torch.int32, device=device)
            self.plan[tag] = dict(a=a, b=b, s=s, lo=lo, hi=hi, me=me, idx=idx, blocks=blocks,
                                  stage=torch.zeros(max(int(idx.numel()), 1), dtype=torch.float32, device=device))
            smax = max(smax, s)
        self.rs_out = torch.zeros(max(smax, 1), dtype=torch.float32, device=device)
        self.ag_in = torch.zeros(max(smax, 1), dtype=torch.bfloat16, device=device)

    def reduce(self, tag: str, g: torch.Tensor):
        """g[a:b) holds this rank's gradients -> afterwards the SUM over ranks on this rank's shard, on the replicated ranges
        and on the remainder (the other shards keep local values nobody reads).  Blocking on gloo; stream-ordered on RCCL."""
        q = self.plan[tag]
        a, b, s, lo, hi, me = q["a"], q["b"], q["s"], q["lo"], q["hi"], q["me"]
        n = int(q["idx"].numel())
        if n:
            torch.index_select(g, 0, q["idx"], out=q["stage"][:n])          # local values, before the shard is overwritten
        if s:
            dist.reduce_scatter_tensor(self.rs_out[:s], g[a:me], op=dist.ReduceOp.SUM, group=self.pg)
            g[lo:hi].copy_(self.rs_out[:s])
        if n:
            dist.all_reduce(q["stage"][:n], op=dist.ReduceOp.SUM, group=self.pg)
            g.index_copy_(0, q["idx"], q["stage"][:n])
        if me < b:
            dist.all_reduce(g[me:b], op=dist.ReduceOp.SUM, group=self.pg)
        return q["blocks"]

    def gather_shadow(self, tag: str, shadow: torch.Tensor):
        """every rank's freshly stepped bf16 shard -> all ranks (the replicated ranges arrive as what every rank computed anyway)"""
        q = self.plan[tag]
        if q["s"]:
            self.ag_in[:q["s"]].copy_(shadow[q["lo"]:q["hi"]])
            dist.all_gather_into_tensor(shadow[q["a"]:q["me"]], self.ag_in[:q["s"]], group=self.pg)

    def gather_state(self, *flats: torch.Tensor):
        """fp32 master / moments: every shard from its owner (in place)"""
        for q in self.plan.values():
            if not q["s"]:
                continue
            for t in flats:
                mine = t[q["lo"]:q["hi"]].clone()
                dist.all_gather_into_tensor(t[q["a"]:q["me"]], mine, group=self.pg)

    def wire_bytes_per_step(self) -> dict:
        """what one rank sends per step under the ring model, against the all-reduce of the same buckets"""
        f = (self.world - 1) / self.world
        shard = sum(q["me"] - q["a"] for q in self.plan.values())
        rep = sum(int(q["idx"].numel()) + q["b"] - q["me"] for q in self.plan.values())
        total = sum(q["b"] - q["a"] for q in self.plan.values())
        return {"sharded": f * (4 * shard + 2 * shard) + 2 * f * 4 * rep, "allreduce_fp32": 2 * f * 4 * total}


# ---------------------------------------------------------------------------------------------- choosing the exchange
# Where in the backward a completion group's last gradient has been launched, as a fraction of the backward's duration
# (tools/step_stamps.py on tulip_base, KITTI, batch 8: profiles/r3_step_stamps.txt; other depths: evenly spaced).
_CLOSE_FRACTION_4 = {"head": 0.06, "dec2": 0.12, "dec1": 0.24, "dec0": 0.35, "enc3": 0.51, "enc2": 0.65, "enc1": 0.77,
                     "enc0": 0.90, "embed": 1.0}


def close_fractions(tags: Sequence[str]) -> List[float]:
    if all(t in _CLOSE_FRACTION_4 for t in tags):
        return [_CLOSE_FRACTION_4[t] for t in tags]
    return [(k + 1) / len(tags) for k in range(len(tags))]


def predict_exposed_ms(bucket_bytes: Sequence[float], close_frac: Sequence[float], backward_ms: float, world: int,
                       busbw_GBps: float, latency_ms: float) -> Tuple[float, List[float]]:
    """Ring all-reduce model (nccl-tests convention: t = latency + 2 (N-1)/N S / busbw); the collectives of a step run one
    after the other on RCCL's stream, each starting no earlier than its bucket closes.  Returns (time the last all-reduce
    ends behind the end of the backward, per-bucket milliseconds)."""
    f = 2.0 * (world - 1) / world
    end, per = 0.0, []
    for nbytes, frac in zip(bucket_bytes, close_frac):
        t = latency_ms + f * nbytes / (busbw_GBps * 1e6)
        per.append(t)
        end = max(end, frac * backward_ms) + t
    return max(0.0, end - backward_ms), per


def choose_comm_plan(buckets: Sequence[Tuple[str, int, int]], world: int, busbw_GBps: float, latency_ms: float,
                     backward_ms: float = 1.34, adamw_ms: float = 0.15, requested_dtype: str = "auto",
                     requested_bucket_adamw: Optional[bool] = None, slack_ms: float = 0.10) -> dict:
    """Pick the gradient exchange of an N-rank step BEFORE anything is captured, from the bus bandwidth bench.py's
    collective_smoke measured on this node (reference: DistributedDataParallel's fp32 all-reduce, main_lidar_upsampling.py:277).

    Candidates: gradient dtype fp32 (the reference's exchange) or bf16 (half the bytes, cast back to fp32 in front of AdamW)
    x one AdamW launch behind the last all-reduce or one per bucket behind that bucket's all-reduce (beside the rest of
    the backward).  fp32 stays unless its predicted exposed exchange exceeds `slack_ms` AND bf16 is predicted to save at
    least that much; the per-bucket optimizer is chosen when the exchange of the LAST bucket is what the step waits for
    anyway (its update then hides the other buckets' ~adamw_ms).  Explicit requests win.  Every candidate and its
    prediction is returned so that the measurement can be read against it."""
    tags = [t for t, _, _ in buckets]
    fr = close_fractions(tags)
    cands = []
    for dt, el in (("fp32", 4), ("bf16", 2)):
        nb = [(b - a) * el for _, a, b in buckets]
        exposed, per = predict_exposed_ms(nb, fr, backward_ms, world, busbw_GBps, latency_ms)
        for ba in (False, True):
            # one AdamW behind everything: exposed exchange + the whole update; per bucket: only the last bucket's share
            last_share = (buckets[-1][2] - buckets[-1][1]) / max(1, buckets[-1][2])
            tail = adamw_ms * (last_share if ba else 1.0)
            cands.append({"grad_dtype": dt, "bucket_adamw": ba, "predicted_exposed_exchange_ms": round(exposed, 4),
                          "predicted_tail_ms": round(exposed + tail, 4), "per_bucket_allreduce_ms": [round(x, 4) for x in per]})
    def pick(dt, ba):
        return next(c for c in cands if c["grad_dtype"] == dt and c["bucket_adamw"] == ba)
    reason = []
    if requested_dtype in ("fp32", "bf16"):
        dt = requested_dtype
        reason.append(f"grad dtype {dt} requested")
    else:
        e32, e16 = pick("fp32", False)["predicted_exposed_exchange_ms"], pick("bf16", False)["predicted_exposed_exchange_ms"]
        if e32 > slack_ms and e32 - e16 >= slack_ms:
            dt = "bf16"
            reason.append(f"fp32 exchange predicted {e32:.3f} ms exposed, bf16 {e16:.3f}")
        else:
            dt = "fp32"
            reason.append(f"fp32 exchange predicted {e32:.3f} ms exposed (<= {slack_ms} or bf16 saves < {slack_ms})")
    if requested_bucket_adamw is not None:
        ba = bool(requested_bucket_adamw)
        reason.append(f"per-bucket AdamW {'on' if ba else 'off'} requested")
    else:
        ba = pick(dt, True)["predicted_tail_ms"] + 0.02 < pick(dt, False)["predicted_tail_ms"]
        reason.append("per-bucket AdamW predicted to " + ("win" if ba else "make no difference"))
    chosen = dict(pick(dt, ba))
    return {"chosen": chosen, "reason": "; ".join(reason), "candidates": cands, "model": {
        "busbw_GBps": busbw_GBps, "latency_ms": latency_ms, "backward_ms": backward_ms, "adamw_ms": adamw_ms,
        "close_fractions": dict(zip(tags, fr)), "world": world, "slack_ms": slack_ms}}
