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

    def wait_all(self) -> None:
        for w in self.pending:
            w.wait()          # stream-level dependency for NCCL/RCCL; blocking for gloo
        self.pending.clear()
