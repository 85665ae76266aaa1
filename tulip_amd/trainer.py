"""The training step of the reference's hot loop (engine_upsampling.py:66-100, misc.py:294-308,
lr_sched.py:9-21, main_lidar_upsampling.py:282-283) as ONE fixed launch sequence on one MI355X:

    zero grads -> DropPath draws -> forward -> L1 loss -> backward -> [bucketed RCCL all-reduce,
    overlapped] -> fused AdamW (fp32 master + bf16 shadow)

captured once into HIP graphs (torch.cuda.CUDAGraph) and replayed; the host only refreshes the
8-float hyper-parameter block (lr and bias corrections) per step.  With world_size > 1 the sequence
is cut into graph segments at the bucket boundaries and the all-reduces are issued between replays
(collectives are deliberately NOT captured).  bf16 needs no GradScaler (misc.py:288-308 exists for
fp16).  Gradient accumulation (`accum_iter`, engine:90-97) runs the forward/backward sequence with
d(loss/accum_iter) accumulating into the flat gradient buffer; only the last micro-step of a window
all-reduces (the sum of the per-micro-step all-reduces the reference issues) and applies AdamW.
`train_one_epoch` is the reference's loop around it (per-iteration LR, non-finite-loss abort).
"""
from __future__ import annotations

import math
import os
import sys
from typing import Iterable, Optional

import torch
import torch.distributed as dist

from . import knobs, ops
from .ddp import GradBucketer
from .engine import ALIGN


def cosine_lr(epoch_float: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    """Per-iteration warm-up + half-cosine (lr_sched.py:9-21)."""
    if epoch_float < warmup_epochs:
        return lr * epoch_float / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch_float - warmup_epochs)
                                                          / (epochs - warmup_epochs)))


class Trainer:
    def __init__(self, model, batch_size: int, lr: float = 5e-4, betas=(0.9, 0.95), eps: float = 1e-8,
                 weight_decay: float = 0.01, device=None, use_graph: bool = True, process_group=None,
                 bucket_mb: float = 16.0, accum_iter: int = 1, track_grad_norm: bool = False,
                 force_segments: bool = False, bucket_adamw: Optional[bool] = None, grad_dtype: str = "fp32",
                 attn_fp8: Optional[bool] = None, exchange: str = "allreduce"):
        """force_segments: run the N>1 step structure (graph segments cut at the bucket points, one all-reduce per
        bucket between replays) in a one-rank process group too -- how the RCCL path is exercised on a single GPU.
        bucket_adamw: None = environment default (TULIP_BUCKET_ADAMW, off).
        grad_dtype: "fp32" (DistributedDataParallel's exchange, main_lidar_upsampling.py:277) or "bf16": every bucket is
        cast to bf16 before its all-reduce (half the bytes on the xGMI links) and back into the fp32 gradient buffer in
        front of AdamW, whose moments and master weights stay fp32.
        exchange: "allreduce" (default) or "sharded" (tulip_amd.ddp.ShardedExchange, round 5; optional, never measured on more
        than one GPU): per bucket reduce-scatter -> AdamW on the owned shard and on the parameters the kernels read in fp32 ->
        all-gather of the bf16 shadow, on the optimizer stream beside the rest of the backward.  The fp32 master and the moments
        are then current only where this rank steps them: gather_state() (called by state_dict()) makes them whole."""
        self.model = model
        device = device or torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.eng = model.engine()
        if attn_fp8 is not None:      # BASELINE configs[4]: attention scores from e4m3 q, k (default: env TULIP_ATTN_FP8, off)
            self.eng.attn_fp8 = bool(attn_fp8)
        self.eng.bind(device)
        self.P = self.eng.plan(batch_size)
        W = self.eng.params
        self.g = torch.zeros(W.total, dtype=torch.float32, device=device)
        self.m = torch.zeros_like(self.g)
        self.v = torch.zeros_like(self.g)
        self.hyper = torch.zeros(8, dtype=torch.float32, device=device)
        # the host runs several steps ahead of the GPU: each upload of the hyper block reads its own pinned slot, and a
        # slot is rewritten only after the copy that last read it has executed
        self._hyper_slots = [torch.zeros(8, dtype=torch.float32).pin_memory() for _ in range(4)]
        self._hyper_events = [None] * 4
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.t = 0
        self.accum_iter, self.micro = int(accum_iter), 0
        if self.accum_iter < 1:
            raise ValueError("accum_iter must be >= 1")
        # misc.py:303 get_grad_norm_ (the reference computes it at every update and drops the value)
        self.track_grad_norm = track_grad_norm
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=device)
        self._norm_part = torch.zeros(1024, dtype=torch.float64, device=device)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucketer = GradBucketer(W.groups, W.total, bucket_mb, process_group, force=force_segments)
        self.segmented = self.bucketer.active          # the step is cut at the bucket points
        if grad_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_dtype must be 'fp32' or 'bf16'")
        self.grad_dtype = grad_dtype
        self.gb = (torch.zeros(W.total, dtype=torch.bfloat16, device=device)
                   if (grad_dtype == "bf16" and self.segmented) else None)
        self.use_graph = use_graph
        # N > 1: each bucket is updated (fused AdamW on its slice) on a separate stream as soon as ITS all-reduce has
        # finished, beside the rest of the backward, instead of one update after the last bucket; nothing later in
        # the step reads a finished bucket's parameters or gradients.  The grad-norm read-out needs all gradients.
        # Opt-in (TULIP_BUCKET_ADAMW=1): it relies on RCCL's stream-ordered work.wait(); the default is one update
        # after the last bucket, the order the reference's DDP + optimizer.step() has.
        if bucket_adamw is None:
            bucket_adamw = knobs.is_one("TULIP_BUCKET_ADAMW")
        if exchange not in ("allreduce", "sharded"):
            raise ValueError("exchange must be 'allreduce' or 'sharded'")
        self.exchange = exchange if self.segmented else "allreduce"
        if self.exchange == "sharded":
            if track_grad_norm or grad_dtype != "fp32" or self.accum_iter != 1:
                raise ValueError("exchange='sharded': fp32 gradients, no gradient-norm read-out, accum_iter == 1")
            bucket_adamw = True                      # the plan IS a per-bucket optimizer
        self.bucket_adamw = self.segmented and not track_grad_norm and bool(bucket_adamw)
        self._opt_stream = torch.cuda.Stream(device=device) if self.bucket_adamw else None
        self._sharded = None                         # built at the first bucket (FlatParams.fp32_read is complete by then)
        # (FlatParams groups every parameter where it is last READ in the backward -- the skip Linears sit in their
        # encoder stage's group -- so a bucket's weights are dead once the bucket's hook has fired)
        self._segments = None     # {is_update_step: [(CUDAGraph, tag or None)]}
        self._pack_epoch = -1
        # the fragment-major weight copies of the fused wide blocks (one ~25-us launch) are rewritten at the START of the
        # next step, beside the forward's first kernels, instead of on the chain behind AdamW
        self.pack_at_step_start = True
        # round 5 (TULIP_PACK_AT_END=0: off): the copies rewritten at the END of the step, on the chain's own queue while it waits for the
        # side queue's last groups (TulipEngine.run_backward) -- no forks or joins beside the forward, no pack traffic beside its
        # kernels: 1.9107 -> 1.8941 ms at batch 8, 9.054 -> 9.017 at batch 64 (four interleaved same-box runs each,
        # profiles/r5_ab_pack_at_end.txt; identical losses: the same arithmetic).  Needs every packed weight stepped beside the backward
        # (decided in _plan_fused_adamw: one GPU, captured step); otherwise the pieces beside the forward stay
        self._want_pack_at_end = knobs.on("TULIP_PACK_AT_END", True)
        self._pack_at_end = False
        # without gradient accumulation the backward WRITES every gradient (one producer per parameter and step) instead of adding
        # to a buffer AdamW has to clear: TulipEngine.grad_overwrite (TULIP_GRAD_OVERWRITE=0: accumulate + clear, as with accum_iter > 1)
        self.grad_overwrite = (self.accum_iter == 1 and knobs.on("TULIP_GRAD_OVERWRITE", True)
                               and self.eng.overwrite_supported(batch_size))
        if self.exchange == "sharded" and not self.grad_overwrite:
            # the sharded plan clears nothing outside the blocks this rank steps: an accumulating (+=) backward would feed the stale
            # running sums of the other ranks' shards into the next reduce-scatter
            raise ValueError("exchange='sharded' needs the overwriting backward (accum_iter == 1, TULIP_GRAD_OVERWRITE on, grouped "
                             "weight gradients)")
        # ... and where a weight-gradient workgroup holds a tensor's COMPLETE gradient tile (no token split: the deep stages, 90 %
        # of the parameters), the optimizer step is taken right there, in the write-out, beside the backward instead of behind it
        # (TULIP_FUSE_ADAMW=0: one AdamW launch over everything at the end of the step).  Captured steps on one GPU only: a
        # gradient all-reduce, the gradient-norm read-out and accumulation need the gradients themselves.
        self.fuse_adamw = (self.grad_overwrite and use_graph and not self.segmented and not track_grad_norm
                           and knobs.on("TULIP_FUSE_ADAMW", True))
        self._adam_mask = None
        self._adam_blocks = None
        self._adam_ctx, self._adam_fused = None, frozenset()
        self.fused_adamw_params = 0
        # parity tests: explicit DropPath uniforms [n_drop_slots][B] (device tensor) instead of the counter-based draws;
        # set before the first step (the choice is baked into the captured graphs)
        self.inject_drop_u: Optional[torch.Tensor] = None
        # (optimizer step, what the host issued last): read by bench.py's watchdog thread to say WHERE a multi-rank
        # step stopped making progress (a hung collective shows as the segment tag it followed)
        self.progress = (0, "init")
        self._side = torch.cuda.Stream(device=device) if use_graph else None
        # N > 1, captured steps: the last side group of every bucket but the final one is a graph of its own on this stream (own
        # split-K workspace), so that a cut of the step graph does not make the chain wait for it (TULIP_DETACH_BUCKETS=0: the
        # group is forked inside the segment and joined at the cut, one block late)
        self.detach_buckets = bool(use_graph and self.segmented and knobs.on("TULIP_DETACH_BUCKETS", True))
        self._det_stream = torch.cuda.Stream(device=device) if self.detach_buckets else None
        self._ws_det = (torch.empty(self.eng.WS_ELEMS + (1 << 20), dtype=torch.float32, device=device)
                        if self.detach_buckets else None)
        self._det_graphs, self._det_events = {}, {}
        # round 6 (TULIP_GRAPH_COLLECTIVES=0: off): the N > 1 step as ONE captured graph -- every bucket's all-reduce is captured as a
        # branch off the side queue behind the bucket's last weight-gradient group (TulipEngine.run_backward(bucket_on_side=True)),
        # AdamW behind the last of them -- instead of graph segments cut at the bucket points with eager collectives in between (each
        # cut costs the chain ~20 us on one GPU: tools/exp_segments.py).  RCCL only (gloo cannot be captured); a capture that raises
        # falls back to the segmented form (self.step_form says which one runs).  Never run on more than one GPU.
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self.graph_collectives = bool(use_graph and self.segmented and self.exchange == "allreduce" and backend == "nccl"
                                      and knobs.on("TULIP_GRAPH_COLLECTIVES", True))
        self.step_form = "unset"
        self.process_group = process_group
        if self.world > 1:
            # DistributedDataParallel's constructor broadcasts rank 0's parameters and buffers (main_lidar_upsampling.py:277
            # after per-rank seeds `args.seed + rank`, :155): without it every rank would start from its own init and
            # apply identical averaged gradients to different weights.
            src = 0 if process_group is None else dist.get_global_rank(process_group, 0)
            dist.broadcast(W.flat, src=src, group=process_group)
            self.eng.wgrad_ctas = self.eng.WGRAD_DDP_CTAS      # leave the CUs RCCL's channels sit on out of a round
        W.refresh_shadow()

    # ------------------------------------------------------------------ pieces
    def _set_hyper(self):
        self.t += 1
        b1, b2 = self.betas
        k = self.t % len(self._hyper_slots)
        if self._hyper_events[k] is not None:
            self._hyper_events[k].synchronize()
        h = self._hyper_slots[k]
        h[0] = self.lr
        h[1], h[2], h[3], h[4] = b1, b2, self.eps, self.wd
        h[5], h[6] = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        h[7] = 1.0 / self.world
        self.hyper.copy_(h, non_blocking=True)
        if self._hyper_events[k] is None:
            self._hyper_events[k] = torch.cuda.Event()
        self._hyper_events[k].record()

    def _fwd_bwd(self, hook, update: bool = True, apply_adamw: bool = False, bucket_on_side: bool = False):
        eng, P = self.eng, self.P
        # the fused-AdamW plan points at THIS trainer's gradient / moment / mask buffers: a second Trainer on the same model
        # (another batch size, a rebuild) must not leave its own on the engine for this one's launches or captures
        if getattr(self, "_adam_ctx", None) is not None:
            eng.adam_fused, eng.adam_ctx = self._adam_fused, self._adam_ctx
        elif eng.adam_probe is None:
            eng.adam_fused, eng.adam_ctx = frozenset(), None
        # (the flat gradient buffer is cleared by the fused AdamW right after it consumed it)
        eng.draw_drop_scales(P, self.model.training, self.inject_drop_u)
        eng.run_forward(P, pack_on_side=self.pack_at_step_start, defer_loss_final=True)
        eng.run_backward(P, self.g, gscale=1.0 / self.accum_iter, bucket_hook=hook,
                         join_tags=set(self.bucketer.by_tag) if (self.segmented and update) else None,
                         overwrite=self.grad_overwrite, apply_adamw=apply_adamw, pack_at_end=self._pack_at_end and apply_adamw,
                         bucket_on_side=bucket_on_side)

    def _adamw_range(self, lo: int, hi: int):
        W = self.eng.params
        ops.adamw(W.base32 + 4 * lo, self.g.data_ptr() + 4 * lo, self.m.data_ptr() + 4 * lo,
                  self.v.data_ptr() + 4 * lo, W.base16 + 2 * lo, hi - lo, self.hyper,
                  W.decay_mask.data_ptr() + lo // 64, zero_grad=not self.grad_overwrite)

    def _cast_bucket_down(self, tag):
        """grad_dtype bf16: the bucket's gradients -> the bf16 exchange buffer (captured at the end of the bucket's graph
        segment; issued right before the all-reduce in eager mode)."""
        if self.gb is not None and tag in self.bucketer.by_tag:
            a, b = self.bucketer.by_tag[tag]
            ops.cast_flat(self.g.data_ptr() + 4 * a, self.gb.data_ptr() + 2 * a, b - a)

    def _bucket_done(self, tag, cast: bool = False):
        """All-reduce of the bucket that `tag` completes; with bucket_adamw also its optimizer update, ordered behind
        the collective on the optimizer stream (work.wait() is a stream-level dependency for RCCL)."""
        if cast:
            self._cast_bucket_down(tag)
        if self.exchange == "sharded":
            if tag in self.bucketer.by_tag:
                self._sharded_bucket(tag)
            return
        r = self.bucketer.on_group_done(tag, self.g if self.gb is None else self.gb, keep=not self.bucket_adamw)
        if r is None or not self.bucket_adamw:
            return
        work, a, b = r
        ev = torch.cuda.Event()
        ev.record()                      # (the collective implies this order; a dry run and a capture need it said)
        with torch.cuda.stream(self._opt_stream):
            self._opt_stream.wait_event(ev)
            work.wait()
            if self.gb is not None:
                ops.cast_bf16_f32(self.gb.data_ptr() + 2 * a, self.g.data_ptr() + 4 * a, b - a)
            self._adamw_range(a, b)

    def _sharded_bucket(self, tag):
        """exchange="sharded": reduce-scatter -> AdamW on what this rank steps -> all-gather of the bf16 shadow, all on the
        optimizer stream behind the bucket's last producer (the caller's stream position)."""
        W = self.eng.params
        if self._sharded is None:
            from .ddp import ShardedExchange
            rep = sorted((W.offset[n], W.offset[n] + W.numel[n]) for n in W.fp32_read)
            self._sharded = ShardedExchange(self.bucketer.buckets, rep, self.device, self.process_group)
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._opt_stream):
            self._opt_stream.wait_event(ev)
            dry = self.bucketer.dry              # (bench.py's re-timing with the collectives skipped: the local work stays)
            blocks = self._sharded.plan[tag]["blocks"] if dry else self._sharded.reduce(tag, self.g)
            ops.adamw_blocks(W.flat, self.g, self.m, self.v, W.shadow, blocks, blocks.numel(), self.hyper, W.decay_mask,
                             zero_grad=not self.grad_overwrite)
            if not dry:
                self._sharded.gather_shadow(tag, W.shadow)
        if self.world > 1:
            # the fp32 master (what model.parameters() / state_dict() view) is now current only on the shards this rank owns:
            # nobody may re-derive the bf16 shadow from it until gather_state() has run on every rank
            W.master_partial = True

    def gather_state(self):
        """exchange="sharded": fp32 master and both moments whole on every rank again (checkpoints, state_dict(), evaluation
        through anything that re-derives the shadow from the master).  A COLLECTIVE: every rank of the process group must call it
        (and therefore Trainer.state_dict()) together -- a call guarded by `if rank == 0` deadlocks.  A no-op for the all-reduce plans.
        Until it has run, FlatParams.master_partial makes every path that would rebuild the bf16 shadow from the stale master
        (model(x) through the module, GraphedForward.weights_changed, load_state_dict -> refresh_shadow) raise instead."""
        if self._sharded is not None:
            torch.cuda.current_stream().wait_stream(self._opt_stream)
            self._sharded.gather_state(self.eng.params.flat, self.m, self.v)
            self.eng.params.master_partial = False

    def _finish_buckets(self):
        if self.bucket_adamw:
            torch.cuda.current_stream().wait_stream(self._opt_stream)
            if not self.pack_at_step_start:
                self.eng.params.refresh_transposes()
        else:
            self.bucketer.wait_all()
            self._adamw()

    def _adamw(self):
        W = self.eng.params
        if self.gb is not None:                       # the summed bf16 gradients come back into the fp32 buffer
            ops.cast_bf16_f32(self.gb, self.g, W.total)
        if self.track_grad_norm:
            # g holds the SUM over ranks here; hyper[7] = 1/world turns it into DDP's mean
            ops.grad_norm(self.g, W.total, self._norm_part, self.grad_norm, scale_dev=self.hyper[7:8])
        mask = self._adam_mask if self._adam_mask is not None else W.decay_mask       # (bit 1: stepped beside the backward)
        if self._adam_blocks is not None:
            # most tensors were stepped where their gradient was completed: the few blocks left, by index (nothing scanned)
            ops.adamw_blocks(W.flat, self.g, self.m, self.v, W.shadow, self._adam_blocks, self._adam_blocks.numel(), self.hyper,
                             mask, zero_grad=not self.grad_overwrite)
        else:
            ops.adamw(W.flat, self.g, self.m, self.v, W.shadow, W.total, self.hyper, mask, zero_grad=not self.grad_overwrite)
        if self._pack_at_end:
            if W.pk_late:
                W.refresh_transposes(late=True)
        elif not self.pack_at_step_start:
            W.refresh_transposes()

    # ------------------------------------------------------------------ checkpoint (misc.save_model / load_model keep
    # {'model', 'optimizer', 'epoch', ...}: this is the 'optimizer' entry of the fused AdamW)
    def state_dict(self) -> dict:
        W = self.eng.params
        self.gather_state()
        cut = lambda flat, n: flat[W.offset[n]:W.offset[n] + W.numel[n]].view(W.shape[n]).clone()
        return {"step": self.t, "micro": self.micro, "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps,
                "weight_decay": self.wd, "accum_iter": self.accum_iter,
                "exp_avg": {n: cut(self.m, n) for n in W.names}, "exp_avg_sq": {n: cut(self.v, n) for n in W.names},
                "grad": {n: cut(self.g, n) for n in W.names} if self.micro % self.accum_iter else None,
                "drop_seed": int(self.eng._drop_seed), "drop_counter": int(self.eng._drop_counter.item()),
                # the DropPath stream is seeded per rank (main_lidar_upsampling.py:155: args.seed + rank); a checkpoint is
                # written by rank 0 and read by every rank, which re-derives its own seed from the saver's
                "rank": dist.get_rank(self.process_group) if dist.is_initialized() else 0}

    def load_state_dict(self, sd: dict) -> None:
        W = self.eng.params
        missing = [n for n in W.names if n not in sd["exp_avg"] or n not in sd["exp_avg_sq"]]
        if missing:
            raise KeyError(f"optimizer state lacks {len(missing)} parameters, e.g. {missing[:3]}")
        put = lambda flat, n, t: flat[W.offset[n]:W.offset[n] + W.numel[n]].copy_(t.reshape(-1).to(flat.dtype))
        for n in W.names:
            put(self.m, n, sd["exp_avg"][n])
            put(self.v, n, sd["exp_avg_sq"][n])
        self.g.zero_()
        if sd.get("grad") is not None:
            for n in W.names:
                put(self.g, n, sd["grad"][n])
        self.t, self.micro, self.lr = int(sd["step"]), int(sd["micro"]), float(sd["lr"])
        self.betas, self.eps, self.wd = tuple(sd["betas"]), float(sd["eps"]), float(sd["weight_decay"])
        # DropPath stream.  Same rank as the one that saved (or a checkpoint without a rank): the saved seed -- the resume then
        # continues the uninterrupted run's draws.  Another rank: the reference seeds rank r with seed + r (main_lidar_upsampling.py:
        # 155); when this engine's own seed sits at exactly that distance from the saved one the convention holds and the own seed
        # IS the shifted one -- otherwise (one manual_seed everywhere, unseeded runs, a sub-group that this process is not a member
        # of: get_rank() == -1) the engine keeps the seed it was built with.
        my_rank = dist.get_rank(self.process_group) if dist.is_initialized() else 0
        saved_rank, own = sd.get("rank"), int(self.eng._drop_seed)
        if saved_rank is None or my_rank == int(saved_rank):
            seed = int(sd["drop_seed"])
        else:
            seed = own
        if seed != own:
            self.eng._drop_seed = seed
            self._segments = None          # the seed is a launch argument baked into the captured graphs: re-capture
        self.eng._drop_counter.fill_(int(sd["drop_counter"]))
        W.master_partial = False       # (the model's load_state_dict wrote the whole master and this call the whole moments)
        W.shadow_dirty = True          # the model's own load_state_dict normally precedes this; refresh either way

    def _plan_fused_adamw(self, eligible):
        """Which tensors take their optimizer step where their gradient is completed, beside the backward, instead of in the
        AdamW launch at the end of the step: `eligible` = {gradient address: elements} of the producers the warm-up pass saw --
        un-split large-tile weight-gradient items (step in the write-out), token-split items and partial-row folds (step in
        the fold launch).  Left out: weights the backward reads again AFTER their gradient is complete (the skip Linears: the
        x_save half of their input gradient is formed when the encoder stage is reached, see FlatParams._completion_order),
        and ranges that do not cover whole tensors."""
        W = self.eng.params
        gbase = self.g.data_ptr()
        by_off = {W.offset[n]: n for n in W.names}
        fused, mask, count = set(), W.decay_mask.clone(), 0
        cand, rejected_modules = [], set()
        module_of = lambda n: n.rsplit(".", 1)[0]
        for ptr, cnt in eligible.items():
            a = (ptr - gbase) // 4
            names, o = [], a
            while o < a + cnt and o in by_off:             # the tensors the range covers, in flat order
                names.append(by_off[o])
                o = (o + W.numel[by_off[o]] + ALIGN - 1) // ALIGN * ALIGN
            whole = bool(names) and W.offset[names[-1]] + W.numel[names[-1]] <= a + cnt <= o
            if not whole or any(n.startswith("skip_connection_layers.") for n in names):
                rejected_modules.update(module_of(n) for n in names)
                continue
            cand.append((ptr, a, cnt, names))
        for ptr, a, cnt, names in cand:
            # a token-split Linear is stepped by its fold launch only when weight AND bias are fused (TulipEngine._issue_pending):
            # a tensor whose partner of the same module was left out stays with the end-of-step launch too -- masking it here
            # would leave it stepped by nobody
            if any(module_of(n) in rejected_modules for n in names):
                continue
            fused.add(ptr)
            mask[a // 64:(a + cnt + 63) // 64] |= 2
            count += sum(W.numel[n] for n in names)
        if fused:
            self._adam_fused = frozenset(fused)
            self._adam_mask = mask
            self._adam_ctx = ops.adamw_ref(self.hyper, self.g, W.flat, self.m, self.v, W.shadow, decay_mask64=mask)
            self.eng.adam_fused, self.eng.adam_ctx = self._adam_fused, self._adam_ctx
            self.fused_adamw_params = count
            if self._want_pack_at_end and getattr(W, "pk_offset", None):
                packed = W.packed_names()
                stepped = lambda n: bool((mask[W.offset[n] // 64:(W.offset[n] + W.numel[n] + 63) // 64] & 2).all())
                late = [n for n in packed if not stepped(n)]
                # the copies of weights the END-of-step AdamW launch steps (the skip Linears: read again after their gradient is
                # complete, so never stepped beside the backward) are rewritten by a small launch of their own behind that launch
                if packed and all(n.startswith("skip_connection_layers.") for n in late):
                    self._pack_at_end, self.pack_at_step_start = True, False
                    W.pk_late = frozenset(late)
                    self.eng._pack_mark_tag_ = None
            left = torch.nonzero((mask[:(W.total + 63) // 64] & 2) == 0).flatten().to(torch.int32)
            if 0 < left.numel() * 64 <= W.total // 4:       # (a long list gains nothing over the scan)
                self._adam_blocks = left.contiguous()

    # ------------------------------------------------------------------ torch.optim.AdamW <-> fused AdamW state
    def _param_names_by_ptr(self):
        W = self.eng.params
        return {W.base32 + 4 * W.offset[n]: n for n in W.names}

    def import_torch_optimizer(self, optimizer: "torch.optim.Optimizer") -> None:
        """Continue a run of the reference's optimizer (main_lidar_upsampling.py:283: torch.optim.AdamW over the module's
        parameters; a reference checkpoint's 'optimizer' entry after `optimizer.load_state_dict`, misc.py:386-390) with the
        fused step: exp_avg / exp_avg_sq / step of every parameter, lr, betas, eps and the weight decay of the decaying
        group.  Parameters are matched by storage (the module's parameters are views of the flat buffer)."""
        W, by_ptr = self.eng.params, self._param_names_by_ptr()
        seen, steps, wds = set(), set(), set()
        for grp in optimizer.param_groups:
            for p in grp["params"]:
                n = by_ptr.get(p.data_ptr())
                if n is None:
                    raise KeyError("optimizer holds a parameter that is not one of this model's (was the model moved?)")
                st = optimizer.state.get(p, {})
                sl = slice(W.offset[n], W.offset[n] + W.numel[n])
                if st:
                    self.m[sl].copy_(st["exp_avg"].reshape(-1).to(self.m.dtype))
                    self.v[sl].copy_(st["exp_avg_sq"].reshape(-1).to(self.v.dtype))
                    steps.add(int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"]))
                else:
                    self.m[sl].zero_(); self.v[sl].zero_(); steps.add(0)
                if p.ndim > 1:
                    wds.add(float(grp["weight_decay"]))
                elif float(grp["weight_decay"]) != 0.0:
                    raise ValueError("the fused AdamW decays ndim > 1 parameters only (timm's grouping, main:282)")
                seen.add(n)
        if seen != set(W.names) or len(steps) != 1 or len(wds) > 1:
            raise ValueError(f"cannot import: {len(W.names) - len(seen)} parameters missing, steps {sorted(steps)}, decays {sorted(wds)}")
        g0 = optimizer.param_groups[0]
        self.t, self.lr, self.betas, self.eps = steps.pop(), float(g0["lr"]), tuple(g0["betas"]), float(g0["eps"])
        if wds:
            self.wd = wds.pop()
        self.g.zero_()
        self.micro = 0
        self.eng.params.shadow_dirty = True

    def export_torch_optimizer(self, optimizer: "torch.optim.Optimizer") -> None:
        """The reverse: write the fused optimizer's moments and step count into a torch.optim.AdamW built over the module's
        parameters (so that `optimizer.state_dict()` is what misc.save_model stores, misc.py:339-345)."""
        W, by_ptr = self.eng.params, self._param_names_by_ptr()
        for grp in optimizer.param_groups:
            grp["lr"], grp["betas"], grp["eps"] = self.lr, tuple(self.betas), self.eps
            for p in grp["params"]:
                n = by_ptr[p.data_ptr()]
                sl = slice(W.offset[n], W.offset[n] + W.numel[n])
                optimizer.state[p] = {"step": torch.tensor(float(self.t)), "exp_avg": self.m[sl].view(W.shape[n]).clone(),
                                      "exp_avg_sq": self.v[sl].view(W.shape[n]).clone()}

    # ------------------------------------------------------------------ graph capture
    def _abort_capture(self, g, origin):
        """Leave capture mode after an exception inside a capture (the caller captures again in another form).  HIP keeps EVERY
        stream of a capture in capture mode when hipStreamEndCapture finds forked work that was never joined (CUDA ends the
        capture there), so: back on the stream the capture began on -- the engine may have been on its side stream when the
        exception passed through --, join every stream the capture forked into, then end it.  The graph is dropped."""
        torch.cuda.set_stream(origin)
        for st in (getattr(self.eng, "_side_stream", None), self._opt_stream, self._det_stream):
            if st is None or st == origin:
                continue
            with torch.cuda.stream(st):
                forked = torch.cuda.is_current_stream_capturing()
            if forked:
                origin.wait_stream(st)
        try:
            self.bucketer.wait_all()         # (collectives captured before the failure: their stream joins through work.wait())
        except Exception:                    # noqa: BLE001
            self.bucketer.pending.clear()
        try:
            g.capture_end()
        except Exception as e:               # noqa: BLE001
            sys.stderr.write(f"tulip_amd.Trainer: ending the failed capture: {type(e).__name__}: {str(e)[:200]}\n")

    def _capture_one_graph(self):
        """The N > 1 optimizer step as ONE graph: collectives captured as branches off the side queue (graph_collectives)."""
        side = self._side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            g.capture_begin(capture_error_mode="thread_local")
            self.eng.detach_buckets = False
            try:
                self._fwd_bwd(lambda tag: self._bucket_done(tag, cast=True), True, bucket_on_side=True)
                self._finish_buckets()       # joins the collectives (and the per-bucket optimizer) / the end-of-step AdamW
            except BaseException:
                self._abort_capture(g, side)
                raise
            g.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        return [(g, None)]

    def _capture(self, update: bool):
        """Capture one step as graph segments cut at the all-reduce points.  update=False is a
        gradient-accumulation micro-step: forward + backward only, no all-reduce, no AdamW.
        Capture mode is thread-local: HIP calls of other threads (the RCCL watchdog polling its events) must not
        invalidate the capture."""
        if update and self.graph_collectives:
            try:
                segs = self._capture_one_graph()
                self.step_form = "one_graph_captured_collectives"
                return segs
            except Exception as e:      # noqa: BLE001  (a backend / runtime that cannot capture the collective: today's segmented form)
                sys.stderr.write(f"tulip_amd.Trainer: capturing the collectives failed ({type(e).__name__}: {e}); "
                                 "falling back to graph segments with eager collectives\n")
                self.graph_collectives = False
                self.bucketer.pending.clear()
                self.P.reset_exchange()
                torch.cuda.synchronize()
        if update:
            self.step_form = "segments" if self.segmented else "one_graph"
        segs = []
        if update:
            self._det_graphs = {}
        side = self._side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            cur = torch.cuda.CUDAGraph()
            cur.capture_begin(capture_error_mode="thread_local")

            def hook(tag):
                nonlocal cur
                if update and self.segmented and tag in self.bucketer.by_tag:
                    det = self.eng.take_detached()
                    if det is None:
                        self._cast_bucket_down(tag)
                    cur.capture_end()
                    if det is not None:
                        # the bucket's last side group as a graph of its own (TulipEngine.detach_buckets): replayed behind this
                        # segment on the detached stream, the all-reduce behind it; the chain's next segment does not wait for it
                        sg = torch.cuda.CUDAGraph()
                        with torch.cuda.stream(self._det_stream):
                            sg.capture_begin(capture_error_mode="thread_local")
                            self.eng.issue_detached(det, self._ws_det.data_ptr())
                            self._cast_bucket_down(tag)
                            sg.capture_end()
                        self._det_graphs[tag] = sg
                    segs.append((cur, tag))
                    cur = torch.cuda.CUDAGraph()
                    cur.capture_begin(capture_error_mode="thread_local")

            self.eng.detach_buckets = bool(update and self.segmented and self.detach_buckets)
            try:
                self._fwd_bwd(hook, update, apply_adamw=update and self._adam_mask is not None)
            except BaseException:
                self._abort_capture(cur, side)      # (the caller sees the exception; the streams are out of capture mode)
                raise
            finally:
                self.eng.detach_buckets = False
            if not update:
                cur.capture_end()
                segs.append((cur, None))
            elif not self.segmented:
                self._adamw()
                cur.capture_end()
                segs.append((cur, None))
            else:
                # the last bucket closes with the backward's final hook ("embed"): the piece opened after it is
                # empty and is not replayed
                cur.capture_end()
                if not (segs and segs[-1][1] == self.bucketer.buckets[-1][0]):
                    segs.append((cur, None))
                else:
                    self._empty_tail = cur          # keep the (empty) graph object alive
                if not self.bucket_adamw:
                    g2 = torch.cuda.CUDAGraph()
                    g2.capture_begin(capture_error_mode="thread_local")
                    self._adamw()
                    g2.capture_end()
                    segs.append((g2, "adamw"))
        torch.cuda.current_stream().wait_stream(side)
        return segs

    # ------------------------------------------------------------------ public
    def load_batch(self, x: torch.Tensor, target: torch.Tensor):
        self.P.x_in.copy_(x.reshape(self.P.x_in.shape), non_blocking=True)
        self.P.target.copy_(target.reshape(self.P.target.shape), non_blocking=True)

    def set_dry(self, dry: bool):
        """Measurement switch (tools/exp_segments.py, bench.py's comm report): keep the step structure, skip the collectives.  The
        one-graph form bakes the collectives into its capture, so the step is re-captured."""
        if self.bucketer.dry != bool(dry):
            self.bucketer.dry = bool(dry)
            if self.graph_collectives:
                self._segments = None

    def zero_grad(self):
        """optimizer.zero_grad() at the top of an epoch (engine_upsampling.py:61): drops the gradients of
        an unfinished accumulation window."""
        self.g.zero_()
        self.micro = 0

    def step(self, x: Optional[torch.Tensor] = None, target: Optional[torch.Tensor] = None,
             lr: Optional[float] = None) -> torch.Tensor:
        """One micro-step on the batch (or on the batch already resident in the plan); every
        `accum_iter`-th call is an optimizer step.  `lr` (if given) becomes the learning rate of the
        next optimizer step.  Returns the device tensor [loss, pixel_loss] (no host sync)."""
        if x is not None:
            self.load_batch(x, target)
        if lr is not None:
            self.lr = lr
        self.micro += 1
        update = self.micro % self.accum_iter == 0
        if update:
            self._set_hyper()
        if update and self.pack_at_step_start:
            # after this step the bf16 shadow is new and the wide blocks' copies are not (this Trainer's next step rewrites
            # them first thing; any other forward in between must, too)
            mark_pack_dirty = True
        else:
            mark_pack_dirty = False
        if self.eng.params.shadow_dirty:
            # parameters were written from outside (load_state_dict / misc.load_model after construction, a foreign
            # optimizer): the bf16 GEMM operands are rebuilt here, outside the captured graphs
            self.eng.params.refresh_shadow()
        if not self.use_graph:
            if update:
                self._fwd_bwd(lambda tag: self._bucket_done(tag, cast=True))
                if self.segmented:
                    self._finish_buckets()
                else:
                    self._adamw()
            else:
                self._fwd_bwd(lambda tag: None, update=False)
            self.eng.params.pack_dirty = self.eng.params.pack_dirty or mark_pack_dirty
            return self.P.losses
        if self._segments is not None and self._pack_epoch != self.eng.params.pack_epoch:
            self._segments = None       # another plan activated the weight copies of a width this capture does not rewrite
        if self._segments is None:
            # load every kernel once outside capture, without touching parameters, optimizer state, the gradients of an
            # open accumulation window or the DropPath stream (the pass below accumulates into g and draws once)
            keep_g, keep_c = self.g.clone(), self.eng._drop_counter.clone()
            if self.fuse_adamw:
                self.eng.adam_probe = {}
            self._fwd_bwd(lambda tag: None)
            if self.fuse_adamw:
                self._plan_fused_adamw(self.eng.adam_probe)
                self.eng.adam_probe = None
            if self.exchange == "allreduce":     # the collectives once outside capture too (g is restored below, gb is scratch)
                self.bucketer.prime(self.g if self.gb is None else self.gb)
            scratch = torch.zeros(64, dtype=torch.float32, device=self.device)
            ops.adamw(scratch, scratch.clone(), scratch.clone(), scratch.clone(), None, 64, self.hyper, None)
            ops.grad_norm(scratch, 64, self._norm_part, self.grad_norm)
            self.g.copy_(keep_g)
            self.eng._drop_counter.copy_(keep_c)
            del keep_g
            torch.cuda.synchronize()
            self._pack_epoch = self.eng.params.pack_epoch
            try:
                self._segments = {True: self._capture(True)}
                if self.accum_iter > 1:
                    self._segments[False] = self._capture(False)
            except BaseException:
                self._segments = None
                self.P.reset_exchange()       # a launch sequence cut short may have left an arrival ticket half-counted
                raise
        for graph, tag in self._segments[update]:
            self.progress = (self.t, f"segment:{tag}")
            if tag == "adamw":
                self.bucketer.wait_all()
                graph.replay()
            else:
                graph.replay()
                if tag is not None:
                    self.progress = (self.t, f"all_reduce:{tag}")
                    sg = self._det_graphs.get(tag) if update else None
                    if sg is None:
                        self._bucket_done(tag)
                    else:
                        ev = self._det_events.get(tag)
                        if ev is None:
                            ev = self._det_events[tag] = torch.cuda.Event()
                        ev.record()
                        with torch.cuda.stream(self._det_stream):
                            self._det_stream.wait_event(ev)
                            sg.replay()
                            self._bucket_done(tag)
        if update and self.bucket_adamw and not self.graph_collectives:
            self._finish_buckets()
        self.eng.params.pack_dirty = self.eng.params.pack_dirty or mark_pack_dirty
        return self.P.losses


def train_one_epoch(trainer: Trainer, data_loader: Iterable, epoch: int, args, log_every: int = 0) -> dict:
    """engine_upsampling.py:46-124 around the fused step: the learning rate is set from
    `data_iter_step / len(data_loader) + epoch` at the first micro-step of every accumulation window
    (engine:68-69), a non-finite loss prints both losses and exits with status 1 (engine:85-88), and
    the returned dict holds the epoch averages of `loss` and the last `lr` (engine:124).
    `args` needs lr, min_lr, warmup_epochs, epochs (lr_sched.py:9-21); batches are
    (low_res, high_res) tensors or the reference's ({'sample': ...}, {'sample': ...}) dicts.
    Like the reference, the loss is read back every iteration (one host sync per step)."""
    trainer.model.train(True)
    trainer.zero_grad()
    n = len(data_loader)
    tot, cnt, lr = 0.0, 0, trainer.lr
    for it, (lo, hi) in enumerate(data_loader):
        if it % trainer.accum_iter == 0:
            lr = cosine_lr(it / n + epoch, args.lr, args.min_lr, args.warmup_epochs, args.epochs)
        if isinstance(lo, dict):
            lo, hi = lo["sample"], hi["sample"]
        losses = trainer.step(lo.to(trainer.device, non_blocking=True), hi.to(trainer.device, non_blocking=True),
                              lr=lr).tolist()
        if not math.isfinite(losses[0]):
            print("Total Loss is {}, stopping training".format(losses[0]))
            print("Pixel Loss is {}, stopping training".format(losses[1]))
            sys.exit(1)
        if trainer.world > 1:
            # misc.all_reduce_mean on the logged values (engine_upsampling.py:107-109): one 2-float collective
            red = torch.tensor(losses, dtype=torch.float32, device=trainer.device)
            dist.all_reduce(red, group=trainer.process_group)
            losses = (red / trainer.world).tolist()
        tot, cnt = tot + losses[0], cnt + 1
        if log_every and (it + 1) % log_every == 0:
            print(f"Epoch: [{epoch}]  [{it + 1}/{n}]  lr: {lr:.6f}  loss: {losses[0]:.4f}")
    return {"loss": tot / max(cnt, 1), "lr": lr}
