"""The training step of the reference's hot loop (engine_upsampling.py:66-100, misc.py:294-308,
lr_sched.py:9-21, main_lidar_upsampling.py:282-283) as ONE fixed launch sequence on one MI355X:

    zero grads -> DropPath draws -> forward -> L1 loss -> backward -> [bucketed RCCL all-reduce,
    overlapped] -> fused AdamW (fp32 master + bf16 shadow)

captured once into HIP graphs (torch.cuda.CUDAGraph) and replayed; the host only refreshes the
8-float hyper-parameter block (lr and bias corrections) per step.  With world_size > 1 the sequence
is cut into graph segments at the bucket boundaries and the all-reduces are issued between replays
(collectives are deliberately NOT captured).  bf16 needs no GradScaler (misc.py:288-308 exists for
fp16); the non-finite-loss abort (engine:85-88) is left to the caller via `losses`.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from .ddp import GradBucketer


def cosine_lr(epoch_float: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    """Per-iteration warm-up + half-cosine (lr_sched.py:9-21)."""
    if epoch_float < warmup_epochs:
        return lr * epoch_float / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch_float - warmup_epochs)
                                                          / (epochs - warmup_epochs)))


class Trainer:
    def __init__(self, model, batch_size: int, lr: float = 5e-4, betas=(0.9, 0.95), eps: float = 1e-8,
                 weight_decay: float = 0.01, device=None, use_graph: bool = True, process_group=None,
                 bucket_mb: float = 16.0):
        self.model = model
        device = device or torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.eng = model.engine()
        self.eng.bind(device)
        self.P = self.eng.plan(batch_size)
        W = self.eng.params
        self.g = torch.zeros(W.total, dtype=torch.float32, device=device)
        self.m = torch.zeros_like(self.g)
        self.v = torch.zeros_like(self.g)
        self.hyper = torch.zeros(8, dtype=torch.float32, device=device)
        self.hyper_host = torch.zeros(8, dtype=torch.float32).pin_memory()
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.t = 0
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucketer = GradBucketer(W.groups, W.total, bucket_mb, process_group)
        self.use_graph = use_graph
        self._segments = None     # [(CUDAGraph, tag or None)]
        self._side = torch.cuda.Stream(device=device) if use_graph else None
        W.refresh_shadow()

    # ------------------------------------------------------------------ pieces
    def _set_hyper(self, lr: Optional[float]):
        self.t += 1
        b1, b2 = self.betas
        h = self.hyper_host
        h[0] = self.lr if lr is None else lr
        h[1], h[2], h[3], h[4] = b1, b2, self.eps, self.wd
        h[5], h[6] = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        h[7] = 1.0 / self.world
        self.hyper.copy_(h, non_blocking=True)

    def _fwd_bwd(self, hook):
        eng, P = self.eng, self.P
        # (the flat gradient buffer is cleared by the fused AdamW right after it consumed it)
        eng.draw_drop_scales(P, self.model.training)
        eng.run_forward(P)
        eng.run_backward(P, self.g, bucket_hook=hook,
                         join_tags=set(self.bucketer.by_tag) if self.world > 1 else None)

    def _adamw(self):
        W = self.eng.params
        ops.adamw(W.flat, self.g, self.m, self.v, W.shadow, W.total, self.hyper, W.decay_mask, zero_grad=True)

    # ------------------------------------------------------------------ graph capture
    def _capture(self):
        """Capture the step as graph segments cut at the all-reduce points."""
        segs = []
        side = self._side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            cur = torch.cuda.CUDAGraph()
            cur.capture_begin()

            def hook(tag):
                nonlocal cur
                if self.world > 1 and tag in self.bucketer.by_tag:
                    cur.capture_end()
                    segs.append((cur, tag))
                    cur = torch.cuda.CUDAGraph()
                    cur.capture_begin()

            self._fwd_bwd(hook)
            if self.world == 1:
                self._adamw()
                cur.capture_end()
                segs.append((cur, None))
            else:
                cur.capture_end()
                segs.append((cur, None))
                g2 = torch.cuda.CUDAGraph()
                g2.capture_begin()
                self._adamw()
                g2.capture_end()
                segs.append((g2, "adamw"))
        torch.cuda.current_stream().wait_stream(side)
        self._segments = segs

    # ------------------------------------------------------------------ public
    def load_batch(self, x: torch.Tensor, target: torch.Tensor):
        self.P.x_in.copy_(x.reshape(self.P.x_in.shape), non_blocking=True)
        self.P.target.copy_(target.reshape(self.P.target.shape), non_blocking=True)

    def step(self, x: Optional[torch.Tensor] = None, target: Optional[torch.Tensor] = None,
             lr: Optional[float] = None) -> torch.Tensor:
        """One optimizer step on the batch (or on the batch already resident in the plan).
        Returns the device tensor [loss, pixel_loss] (no host sync)."""
        if x is not None:
            self.load_batch(x, target)
        self._set_hyper(lr)
        if not self.use_graph:
            self._fwd_bwd(lambda tag: self.bucketer.on_group_done(tag, self.g))
            self.bucketer.wait_all()
            self._adamw()
            return self.P.losses
        if self._segments is None:
            # load every kernel once outside capture, without touching parameters or optimizer state
            self._fwd_bwd(lambda tag: None)
            scratch = torch.zeros(64, dtype=torch.float32, device=self.device)
            ops.adamw(scratch, scratch.clone(), scratch.clone(), scratch.clone(), None, 64, self.hyper, None)
            self.g.zero_()      # the warm-up pass above accumulated into g without an optimizer step
            torch.cuda.synchronize()
            self._capture()
        for graph, tag in self._segments:
            if tag == "adamw":
                self.bucketer.wait_all()
                graph.replay()
            else:
                graph.replay()
                if tag is not None:
                    self.bucketer.on_group_done(tag, self.g)
        return self.P.losses
