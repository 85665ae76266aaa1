#!/usr/bin/env python3
"""Headline benchmark: range-images/s of a full TULIP training step on MI355X.

Workload (BASELINE.json configs[1]): tulip_base, KITTI 16x1024 -> 64x1024, per-GPU batch 8, bf16 GEMM
operands / fp32 accumulate, train mode (DropPath active), synthetic inputs of SURVEY.md 8(d) resident
in HBM.  A step = forward + L1 loss + backward + gradient all-reduce (N>1) + fused AdamW, replayed
from HIP graphs.  `python bench.py --gpus N --steps K --warmup W`.  N > 1 = one rank per GPU over RCCL: either
launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in the
environment; the reference's recipe, bash_scripts/tulip_upsampling_kitti.sh:35 + util/misc.py:253-285), or, when no
launcher set WORLD_SIZE, bench.py starts the N ranks itself through the same module (spawn_ranks).  Rank 0 prints ONE
JSON line.

Extra objects:  "roofline" -- the dominant kernel (the bf16 MFMA GEMM family): algorithmic FLOPs of
its launches in one step / their summed duration, each launch bracketed by HIP events on the launch
stream in a separate eager pass (graphs cannot be instrumented per node);  "cpu_baseline" -- the
oracle (plain PyTorch fp32 restatement of the reference, oracle/tulip_oracle.py) running the same
training step on the host cores, rank 0, N=1 only, bounded sample.
"""
import argparse
import faulthandler
import json
import os
import socket
import subprocess
import sys
import threading
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FWD_BWD_PER_IMG = 46.349e9      # tulip_base KITTI, 2*MAC, matmul/conv only
BYTES_FWD_OPLEVEL_PER_IMG = 245e6    # op-level forward traffic per image, bf16
PEAK_BF16 = 2.5e15                   # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12


def make_model(args):
    from tulip_amd.model import tulip as T
    torch.manual_seed(0)             # reference init under seed 0
    fac = T.tulip_large if args.model == "tulip_large" else T.tulip_base
    return fac(img_size=tuple(args.img), target_img_size=tuple(args.target), patch_size=(1, 4), in_chans=1,
               window_size=[2, 8], swin_v2=False, pixel_shuffle=True, circular_padding=True, log_transform=True,
               patch_unmerging=True)


def synthetic(args, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    Hh, Wh = args.target
    r = torch.rand(args.batch, 1, Hh, Wh, generator=g)
    r[torch.rand(args.batch, 1, Hh, Wh, generator=g) < 0.1] = 0
    hi = torch.log1p(r)
    lo = hi[:, :, 0::Hh // args.img[0], :].contiguous()
    return lo.to(device), hi.to(device)


def _gemm_instance(M, N, K, a_trans, b_trans, splits):
    """which gemm_kernel<BM,A_T,B_T,KSUB> tulip_gemm_bf16 launches (mirrors csrc/gemm.hip launch())"""
    from tulip_amd import ops
    eff = ops.gemm_effective_splits(K, splits)
    kchunk = -(-(-(-K // eff)) // 32) * 32
    gn = -(-N // 96)
    t128, t256 = -(-M // 128) * gn * eff, -(-M // 256) * gn * eff
    bm = 64 if (t128 < 512 or M <= 64) else 128
    if t256 >= 512 and gn >= 32:
        bm = 256
    ksub = 1
    if bm == 64:
        grid = gn * -(-M // 64) * eff
        ksub = 4 if (grid <= 400 and kchunk >= 256) else 1
    t = lambda f: "true" if f else "false"
    bks = 128 if ksub == 4 else 32
    if not a_trans and M % bm == 0 and N % 96 == 0 and kchunk % bks == 0 and K % kchunk == 0:
        return f"gemm_kernel_full<{bm}, {t(b_trans)}, {ksub}>"          # whole tiles only: no bounds tests around the loads
    return f"gemm_kernel<{bm}, {t(a_trans)}, {t(b_trans)}, {ksub}>"


def kernel_source_stamp():
    """Hash of the kernel sources + launch sequences: counter files under profiles/ carry the stamp they were measured
    at, and a stale file is not reported (the GPU box has no .git to ask for HEAD)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "tulip_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "tulip_amd", "engine.py"), "rb").read())
    return h.hexdigest()[:16]


FAMILIES = {   # kernel-name prefixes per family: shared with tools/instep_summary.py, pmc_summary.py, mfma_summary.py
    "wgrad": ("wgrad_group_kernel", "gemm_group_kernel"),
    "gemm": ("gemm_kernel", "gemm_stream_kernel"),
    "swin96_fwd": ("swin96_fwd_kernel",), "swin96_bwd": ("swin96_bwd_kernel",),
    "swinw_fwd": ("swinw_fwd_kernel",), "swinw_bwd": ("swinw_bwd_kernel",),
    "fold": ("reduce_rows_multi_kernel",), "adamw": ("adamw_kernel",),
    # round 6: what is left of the glue between the Swin blocks -- the fused stage boundaries (csrc/glue.hip), the LayerNorm
    # launches and split-K epilogues of the boundaries / deep blocks that keep their launch sequences ("gemm" above holds their GEMMs)
    "glue": ("merge_fwd_kernel", "merge_bwd_kernel", "unmerge_skip_fwd_kernel", "skip_unmerge_bwd_kernel"),
    "ln": ("ln_fwd_kernel", "ln_bwd_kernel"), "splitk": ("splitk_epilogue_kernel",)}


def _stamped(fname, family):
    """One family's entry of a committed counter / trace summary under profiles/ (each carries the hash of the kernel
    sources it was measured on: a file measured on other sources is not reported -- the GPU box has no .git)."""
    try:
        with open(os.path.join(ROOT, "profiles", fname)) as f:
            d = json.load(f)
        if d.get("source_stamp") != kernel_source_stamp():
            return None
        return d["families"][family]
    except (OSError, KeyError, ValueError):
        return None


def _pmc_traffic(family):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2-corrected + WRITE_SIZE, separate --pmc
    passes as MI355X_MICROARCH.md prescribes, tools/pmc_summary.py); counters cannot be read from inside bench.py."""
    e = _stamped("pmc_traffic.json", family)
    return round(e["hbm_bytes_per_launch"]) if e else None


def _instep(family):
    """In-step mean duration from the committed kernel trace (profiles/instep_durations.json, tools/instep_summary.py)."""
    return _stamped("instep_durations.json", family)


def _mfma_counter(family, t_launch):
    """MFMA utilisation from counters: SQ_VALU_MFMA_BUSY_CYCLES per launch (profiles/pmc_mfma.json, tools/mfma_summary.py, its
    own --pmc pass; busy cycles summed over the 1024 SIMDs) over 1024 x the family's launch duration x 2.4 GHz."""
    e = _stamped("pmc_mfma.json", family)
    return round(e["mfma_busy_cycles_per_launch"] / (1024 * t_launch * 2.4e9), 4) if e else None


def kernel_rooflines(trainer, reps=5):
    """Roofline of every kernel family that matters in the step; the top object is the DOMINANT one (largest summed
    in-step time): the grouped weight-gradient kernel.

    Durations.  `frac` / `achieved` use the mean launch duration INSIDE the step, from the committed rocprofv3 kernel
    trace of this command (profiles/instep_durations.json, kernels between consecutive AdamW launches; accepted only when
    its source stamp equals the kernel sources this process runs, `timing` says which).  Graph nodes cannot be bracketed
    by HIP events, so the LIVE measurement of this run is the `isolated` sub-object: the launches of one step are recorded
    in an eager pass and each is re-issued `reps` times back to back between one HIP-event pair on the launch stream --
    warm caches, nothing beside it, 10-40 % shorter than in the step; it is what `frac` falls back to when the committed
    trace is stale.

    Bytes.  `algorithmic_bytes_per_launch` is SURVEY 8(d)'s convention: every operand read once, every result written once
    at its final precision -- a weight gradient's operands + fp32 dW (+ db) ONCE (split-K slabs are design traffic, not
    algorithmic); a Linear's two operands + its output; a fused Swin block's fp32 stream in + out and its weights once (the
    activations it saves for the backward are design traffic too: `bytes_with_saved_activations` /
    `frac_with_saved_activations` price them in); AdamW 30 B per parameter.  `traffic` = HBM bytes per launch from the PMC
    passes (profiles/pmc_traffic.json); traffic / algorithmic = the overhead of the design.  `mfma_util_counter` =
    SQ_VALU_MFMA_BUSY_CYCLES-based utilisation (profiles/pmc_mfma.json) next to the FLOP-based `frac_mfma`."""
    from tulip_amd import ops
    rec = []                                  # (family, detail-name, callable, flops, algorithmic bytes, bytes incl. design outputs)
    real = {n: getattr(ops, n) for n in ("gemm", "wgrad_group", "swin96_block_fwd", "swin96_block_bwd", "swinw_block_fwd",
                                         "swinw_block_bwd", "reduce_rows_multi")}

    def gemm(A, B, M, N, K, **kw):
        f = 2.0 * M * N * K
        by = 2.0 * (M * K + N * K) + M * N * (4.0 if kw.get("epi", 0) in (3, 4, 5, 6, 7) else 2.0)
        if kw.get("b_packed"):        # the small-K form on the fragment-major copy (csrc/gemm.hip gemm_stream_kernel<K per split / 32>)
            from tulip_amd import ops as _o
            eff = _o.gemm_effective_splits(K, kw.get("splits", 1))
            name = f"gemm_stream_kernel<{-(-K // eff) // 32}>"
        else:
            name = _gemm_instance(M, N, K, kw.get("a_trans", False), kw.get("b_trans", False), kw.get("splits", 1))
        rec.append(("gemm", name, lambda: real["gemm"](A, B, M, N, K, **kw), f, by, by))
        real["gemm"](A, B, M, N, K, **kw)

    def wgrad_group(items, extra, ws, ws_bytes, fold=True, adam=None, small_tiles=False):
        items = list(items)
        f = by = slab = 0.0
        tiles, deep = 0, True
        for it in items:
            eff = ops.gemm_effective_splits(it.Mtok, it.splits)
            kchunk = -(-(-(-it.Mtok // eff)) // 32) * 32
            tiles += -(-it.Kw // 96) * -(-it.Nw // 64) * eff
            deep = deep and kchunk >= 256
            f += 2.0 * it.Nw * it.Kw * it.Mtok
            by += 2.0 * it.Mtok * (it.Nw + it.Kw) + 4.0 * it.Nw * it.Kw + 4.0 * it.Nw      # operands once, fp32 dW + db once
            if it.reserved_ == 1 and eff == 1:     # (a token-split item's step is taken by the fold of its slabs)
                # in the step this item's write-out takes the AdamW step (Trainer.fuse_adamw): parameter + two moments read and
                # written, bf16 shadow written, the gradient itself never stored (26 B instead of 4 B per element)
                by += 22.0 * it.Nw * it.Kw
            slab += 2.0 * it.Mtok * (it.Nw + it.Kw) + 4.0 * (it.Nw * it.Kw + it.Nw) * eff      # what the launch writes: slabs
        ksub = 4 if (tiles <= 400 and deep) else 1
        big = all(ops.wgrad_tiles(it.Nw, it.Kw, small_tiles) != -(-it.Nw // 64) * -(-it.Kw // 96) and it.Mtok % 32 == 0 for it in items)
        # the weight gradients of a stage leave as ONE grouped launch; timed without its fold (the `fold` family)
        rec.append(("wgrad", "wgrad_group_kernel (192x192 / 384x96 / 96x384 tiles)" if big
                    else f"gemm_group_kernel<64, true, true, {ksub}>",
                    lambda: real["wgrad_group"](items, [], ws, ws_bytes, fold=False, adam=adam, small_tiles=small_tiles), f, by, slab))
        real["wgrad_group"](items, extra, ws, ws_bytes, fold, adam, small_tiles)

    def block(name, fam, bwd, C_of):
        def f(*a, **kw):
            C = C_of(a)
            M = kw["B"] * kw["H"] * kw["W"]
            # four linears (data gradients only in the backward: the weight gradients are GEMM launches) + the
            # 16x16 attention core (fwd: QK^T, PV; bwd: S twice, dP twice, dQ, dK, dV)
            fl = M * (24.0 * C * C + (224.0 if bwd else 64.0) * C)
            by = 8.0 * C * M + 24.0 * C * C                                   # fp32 stream in + out, weights once
            saved = (48 if bwd else 40) * C * M + 16.0 * M + 24.0 * C * C     # + every saved / re-read activation
            rec.append((fam, name, lambda: real[name](*a, **kw), fl, by, saved))
            real[name](*a, **kw)
        return f

    patched = {"gemm": gemm, "wgrad_group": wgrad_group,
               "swin96_block_fwd": block("swin96_block_fwd", "swin96_fwd", False, lambda a: 96),
               "swin96_block_bwd": block("swin96_block_bwd", "swin96_bwd", True, lambda a: 96),
               "swinw_block_fwd": block("swinw_block_fwd", "swinw_fwd", False, lambda a: a[0]),
               "swinw_block_bwd": block("swinw_block_bwd", "swinw_bwd", True, lambda a: a[0])}
    for n, fn in patched.items():
        setattr(ops, n, fn)
    try:
        # the recording pass issues what a captured step issues, optimizer steps in the weight-gradient write-outs included (the
        # isolated re-launches below then move the bytes `algorithmic_bytes_per_launch` counts; this runs after the timed steps,
        # the parameters are not used again)
        trainer._fwd_bwd(lambda tag: None, apply_adamw=trainer._adam_mask is not None)
        torch.cuda.synchronize()
    finally:
        for n in patched:
            setattr(ops, n, real[n])
    W = trainer.eng.params
    # the end-of-step AdamW launch as the step issues it: tensors stepped in a weight-gradient write-out are masked out
    amask = trainer._adam_mask if trainer._adam_mask is not None else W.decay_mask
    n_adam = W.total - trainer.fused_adamw_params
    blk = getattr(trainer, "_adam_blocks", None)
    if blk is not None:       # the few 64-element blocks left for the end of the step, by index (Trainer._adamw)
        n_adam = blk.numel() * 64
        rec.append(("adamw", "adamw_kernel_blocks",
                    lambda: ops.adamw_blocks(W.flat, trainer.g, trainer.m, trainer.v, W.shadow, blk, blk.numel(), trainer.hyper,
                                             amask, zero_grad=not trainer.grad_overwrite), 0.0, 30.0 * n_adam, 30.0 * n_adam))
    else:
        rec.append(("adamw", "adamw_kernel",
                    lambda: ops.adamw(W.flat, trainer.g, trainer.m, trainer.v, W.shadow, W.total, trainer.hyper, amask,
                                      zero_grad=not trainer.grad_overwrite), 0.0, 30.0 * n_adam, 30.0 * n_adam))
    fams, detail = {}, {}
    for fam, name, call, fl, by, by2 in rec:
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        a = fams.setdefault(fam, [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += t; a[2] += fl; a[3] += by; a[4] += by2
        d = detail.setdefault(fam, {}).setdefault(name, [0, 0.0, 0.0])
        d[0] += 1; d[1] += t; d[2] += fl
    ridge = PEAK_BF16 / PEAK_HBM
    kernels = {
        "wgrad": "wgrad_group_kernel (every weight + bias gradient of a stage in one grouped launch; gemm_group_kernel for "
                 "shapes without a large tile)",
        "gemm": "gemm_kernel<BM,A_T,B_T,KSUB> / gemm_stream_kernel<KS> (every linear / 1x1 conv forward and data gradient outside the fused blocks; the small-K form on the fragment-major weight copies for the stage boundaries)",
        "swin96_fwd": "swin96_fwd_kernel (whole C=96 Swin block, forward)", "swin96_bwd": "swin96_bwd_kernel",
        "swinw_fwd": "swinw_fwd_kernel<C,G> (whole C=192/384 Swin block, forward)", "swinw_bwd": "swinw_bwd_kernel<C,G>",
        "adamw": "adamw_kernel (fp32 master + moments + bf16 shadow, 30 B / parameter; only the tensors whose step was not taken in "
                 "a weight-gradient write-out)"}
    out = []
    for fam, (n, t, fl, by, by2) in fams.items():
        ai = fl / by
        bound = "mfma" if ai >= ridge else "hbm"
        ins = _instep(fam)
        t_iso = t / n
        t_use = ins["mean_launch_us"] * 1e-6 if ins else t_iso
        n_step = ins["launches_per_step"] if ins else n
        fpl, bpl, b2pl = fl / n, by / n, by2 / n
        traffic = _pmc_traffic(fam)
        ach = fpl / t_use / 1e12 if bound == "mfma" else bpl / t_use / 1e9
        peak = PEAK_BF16 / 1e12 if bound == "mfma" else PEAK_HBM / 1e9
        e = {"kernel": kernels[fam], "bound": bound, "achieved": round(ach, 2), "peak": peak,
             "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": round(ach / peak, 4), "traffic": traffic,
             "timing": ("in-step: mean launch duration inside the traced step, profiles/instep_durations.json (rocprofv3 "
                        "--kernel-trace of this command, source stamp matches)") if ins else
                       (f"isolated (live): mean of {reps} back-to-back launches per recorded call, HIP events; the committed "
                        "trace was measured on other kernel sources"),
             "mean_launch_us": round(t_use * 1e6, 2), "launches_per_step": round(n_step, 1),
             "us_per_step": round(t_use * n_step * 1e6, 1),
             "arithmetic_intensity_flop_per_byte": round(ai, 1), "algorithmic_bytes_per_launch": round(bpl),
             "flops_per_launch": round(fpl), "frac_mfma": round(fpl / t_use / PEAK_BF16, 4),
             "frac_hbm_algorithmic": round(bpl / t_use / PEAK_HBM, 4),
             "traffic_over_algorithmic": round(traffic / bpl, 2) if traffic else None,
             "frac_hbm_of_traffic": round(traffic / t_use / PEAK_HBM, 4) if traffic else None,
             "mfma_util_counter": _mfma_counter(fam, t_use),
             "isolated": {"mean_launch_us": round(t_iso * 1e6, 2), "launches": n,
                          "frac_hbm_algorithmic": round(bpl / t_iso / PEAK_HBM, 4), "frac_mfma": round(fpl / t_iso / PEAK_BF16, 4),
                          "timing": f"live, this run: mean of {reps} back-to-back launches per recorded call, HIP events"}}
        if by2 != by:
            e["bytes_with_saved_activations" if fam.startswith("swin") else "bytes_with_splitk_slabs"] = round(b2pl)
            e["frac_with_saved_activations" if fam.startswith("swin") else "frac_with_splitk_slabs"] = round(b2pl / t_use / PEAK_HBM, 4)
        if fam in ("gemm", "wgrad"):
            e["by_kernel_isolated"] = {k: {"launches": c, "avg_us": round(tt / c * 1e6, 2), "tflops": round(f / tt / 1e12, 1)}
                                       for k, (c, tt, f) in sorted(detail[fam].items(), key=lambda kv: -kv[1][1])}
        out.append(e)
    fold = _instep("fold")
    if fold:      # the launches that fold the slabs + partial rows (no FLOPs, design traffic only): listed for the time they take
        out.append({"kernel": "reduce_rows_multi_kernel (folds of split-K slabs and per-workgroup partial rows)", "bound": "hbm",
                    "mean_launch_us": round(fold["mean_launch_us"], 2), "launches_per_step": round(fold["launches_per_step"], 1),
                    "us_per_step": round(fold["mean_launch_us"] * fold["launches_per_step"], 1), "traffic": _pmc_traffic("fold"),
                    "frac": None, "algorithmic_bytes_per_launch": 0, "timing": "in-step (profiles/instep_durations.json)"})
    out.sort(key=lambda e: -e["us_per_step"])
    k = next(i for i, e in enumerate(out) if e.get("frac") is not None)
    top = dict(out.pop(k))
    # the LIVE figure of this run next to the trace-derived one (round-5 review, weak #4a): `frac` is read from a committed trace of
    # this command (stamp-checked), `frac_isolated` is measured by HIP events in this very process
    top["frac_isolated"] = top["isolated"]["frac_mfma" if top["bound"] == "mfma" else "frac_hbm_algorithmic"]
    top["tracer_distortion"] = _tracer_distortion()
    top["others"] = out
    return top


def _tracer_distortion():
    """What the tracer does to the step it measures: the side queue's first kernel, measured from the head of the backward chain, in the
    traced step (profiles/instep_durations.json) and in the un-traced captured step (profiles/step_stamps.json: stamp kernels inside
    the graph); both accepted only with this process's kernel-source stamp."""
    out = {}
    for key, fname, field in (("traced", "instep_durations.json", "traced_side_queue_start_after_backward_begins_us"),
                              ("untraced", "step_stamps.json", "side_queue_start_after_backward_begins_us")):
        try:
            with open(os.path.join(ROOT, "profiles", fname)) as f:
                d = json.load(f)
            out[f"side_queue_start_{key}_us"] = (round(d[field], 1) if d.get("source_stamp") == kernel_source_stamp()
                                                  and d.get(field) is not None else None)
        except (OSError, KeyError, ValueError):
            out[f"side_queue_start_{key}_us"] = None
    out["note"] = ("microseconds from the backward chain's first kernel to the side queue's first kernel; the traced step bunches the "
                   "side work later, so in-step durations of side-queue families (`frac`) are partly a tracer artefact")
    return out


def usable_cores() -> int:
    """Host cores this process may really use: min(affinity, cgroup CPU quota).  (The GPU boxes report 256
    logical CPUs but run under a 16-CPU quota; 256 OpenMP threads there are ~200x slower than 16.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(args):
    """Oracle training step (fwd + loss + autograd bwd + torch AdamW) on the host cores."""
    from oracle import tulip_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = O.tulip_base_config(img_size=tuple(args.img), target_img_size=tuple(args.target))
    sd = O.key_seeded_state_dict(cfg, seed=0, randomize_affine=False)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    full = dict(sd)
    full.update(params)
    decay = [p for p in params.values() if p.ndim > 1]
    nodecay = [p for p in params.values() if p.ndim <= 1]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}],
                            lr=5e-4, betas=(0.9, 0.95))
    B = args.cpu_batch
    lo, hi = O.synthetic_batch(cfg, B, seed=1234)
    enc, dec = O.drop_path_rates(cfg)
    times = []
    for it in range(args.cpu_warmup + args.cpu_steps):
        drop_u = {}
        for s in range(cfg.num_layers):
            for b in range(cfg.depths[s]):
                drop_u[f"layers.{s}.blocks.{b}"] = torch.rand(2, B)
        for i in range(cfg.num_layers - 1):
            for b in range(cfg.depths[cfg.num_layers - i - 2]):
                drop_u[f"layers_up.{i}.blocks.{b}"] = torch.rand(2, B)
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        _, loss, _ = O.tulip_forward(full, cfg, lo, hi, drop_u=drop_u)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    t = sorted(times[args.cpu_warmup:])[args.cpu_steps // 2]
    cpu_name = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(B / t, 3), "unit": "range-images/s", "cores": cores, "kind": "port",
            "cpu": cpu_name, "sample": f"{args.cpu_steps} timed training steps (median) of batch {B} after {args.cpu_warmup} warm-up, "
            "fp32 eager PyTorch oracle, same model/config/optimizer"}


def comm_report(trainer, args, world, device, steps=10):
    """N > 1, after the timed region: what the gradient exchange costs beyond the cuts it needs.  The same step
    structure (graph segments at the bucket points) is timed again with the collectives skipped (every rank then applies
    its own gradients: the replicas diverge, which no longer matters here); exposed = step - that.  Also the bucket plan
    and the RCCL settings in force, for the record."""
    def timed():
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            trainer.step()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item() / steps * 1e3
    with_coll = timed()
    trainer.set_dry(True)           # (the one-graph form re-captures: its collectives are nodes of the graph)
    trainer.step(); trainer.step()
    without = timed()
    trainer.set_dry(False)
    trainer.step()
    el = 2 if trainer.grad_dtype == "bf16" else 4
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        ver = "unknown"
    return {"buckets_MB": [round((b - a) * el / 1e6, 2) for _, a, b in trainer.bucketer.buckets],
            "bucket_tags": [t for t, _, _ in trainer.bucketer.buckets], "grad_dtype": trainer.grad_dtype,
            "per_bucket_adamw": trainer.bucket_adamw, "step_ms": round(with_coll, 4),
            "step_ms_collectives_skipped": round(without, 4), "exposed_exchange_ms": round(with_coll - without, 4),
            "rccl_version": ver, "wgrad_workgroups_per_launch": trainer.eng.wgrad_ctas or trainer.eng.WGRAD_BIG_CTAS,
            # None = not set: RCCL's own choice (channel count, algorithm, protocol) is in force
            "env": {k: os.environ.get(k) for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_ALGO", "NCCL_PROTO",
                                                   "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY")}}


# ---------------------------------------------------------------------------------------------- N > 1 plumbing
def spawn_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` with no launcher in the environment: run the N ranks under torch.distributed.run on
    this node (one process per GPU, rendezvous on 127.0.0.1 -- the reference's `torchrun --nproc_per_node N`,
    bash_scripts/tulip_upsampling_kitti.sh:35).  The ranks inherit stdout: rank 0's JSON line is this process's output.
    If the captured-graph step fails the run is repeated, first with the detached bucket graphs off (TULIP_DETACH_BUCKETS=0:
    the round-2 structure), then with --no-graph (eager launches, the same kernels and collectives), so that a multi-GPU
    number exists either way; the line then carries "graph_path_failed": true and the list of failed attempts."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    env["TULIP_BENCH_SPAWNED"] = "1"
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    rc = subprocess.call(base + list(argv), env=env)
    # A failed attempt is repeated with the next more conservative step structure; every repeat carries the history of the
    # failures in TULIP_BENCH_ATTEMPTS, and the JSON line says so ("graph_path_failed", "attempts") -- a number measured on
    # a fallback is never mistaken for the captured, detached-bucket path.
    # rung 0 (default): ONE captured graph with the collectives as its nodes (round 6); rung 1: graph segments with eager collectives
    # between the replays and the detached bucket graphs (the round-5 default); rung 2: the same without the detached graphs; rung 3: eager
    ladder = [("graph_collectives_off", {"TULIP_GRAPH_COLLECTIVES": "0"}, []),
              ("detach_buckets_off", {"TULIP_GRAPH_COLLECTIVES": "0", "TULIP_DETACH_BUCKETS": "0"}, []), ("no_graph", {}, ["--no-graph"])]
    history, current = [], "default"
    for tag, extra_env, extra_argv in ladder:
        if rc == 0 or "--no-graph" in argv or os.environ.get("TULIP_BENCH_NO_RETRY", "0") == "1":
            break
        history.append({"attempt": current, "rc": rc})
        current = tag
        sys.stderr.write(f"bench.py: the {n}-rank run exited with status {rc}; retrying with {tag}\n")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            base[base.index("--master-port") + 1] = str(sk.getsockname()[1])
        env2 = dict(env, **extra_env)
        env2["TULIP_BENCH_ATTEMPTS"] = json.dumps(history)
        rc = subprocess.call(base + list(argv) + extra_argv, env=env2)
    return rc


_PHASE = ["start", 0]


def phase(name: str):
    _PHASE[0] = name
    _PHASE[1] += 1


def start_watchdog(rank: int, limit_s: float, trainer_ref: list):
    """A rank that makes no host-side progress for `limit_s` seconds (a collective that never completes blocks the
    host in the next synchronize) reports where it stopped -- bench phase, optimizer step, the graph segment or
    all-reduce it issued last -- dumps its Python stacks and exits with status 124, which takes the whole
    torch.distributed.run job down instead of hanging the node."""
    def state():
        tr = trainer_ref[0]
        return (tuple(_PHASE), tr.progress if tr is not None else None)

    def run():
        last, t_last = state(), time.time()
        while True:
            time.sleep(1.0)
            cur = state()
            if cur[0][0] == "done":
                return
            if cur != last:
                last, t_last = cur, time.time()
            elif time.time() - t_last > limit_s:
                sys.stderr.write(json.dumps({"bench_watchdog": "no progress", "rank": rank, "seconds": round(time.time() - t_last, 1),
                                             "phase": cur[0][0], "trainer_progress": cur[1]}) + "\n")
                sys.stderr.flush()
                faulthandler.dump_traceback(file=sys.stderr)
                os._exit(124)
    threading.Thread(target=run, daemon=True, name="bench-watchdog").start()


def collective_smoke(device, world: int, nbytes: int):
    """Before anything is captured: one checked all-reduce, then the largest gradient bucket's size timed (5 x, HIP
    events on the current stream, which RCCL's stream joins) and a 4-KB one for the latency floor.  busbw follows
    nccl-tests: algbw x 2 (N-1)/N."""
    chk = torch.full((1024,), float(dist.get_rank() + 1), device=device)
    dist.all_reduce(chk)
    torch.cuda.synchronize()
    want = world * (world + 1) / 2
    if not bool((chk == want).all()):
        raise SystemExit(f"collective smoke: all-reduce returned {chk[0].item()}, expected {want}")
    out = {}
    for tag, nb in (("bucket", nbytes), ("small", 4096)):
        buf = torch.ones(nb // 4, device=device)
        dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dist.all_reduce(buf)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / 5
        tt = torch.tensor([t], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = tt.item()
        out[tag] = {"bytes": nb, "ms": round(t * 1e3, 4), "algbw_GBps": round(nb / t / 1e9, 2),
                    "busbw_GBps": round(nb / t / 1e9 * 2 * (world - 1) / world, 2)}
        del buf
    return out


def replicas_identical(trainer, device) -> bool:
    """Data-parallel replicas apply the same averaged gradients to the same weights: after any number of steps the flat
    parameter buffers of all ranks are equal bit for bit (DistributedDataParallel's invariant).  Checked through two order-
    sensitive checksums (MIN == MAX over ranks)."""
    trainer.gather_state()        # (exchange="sharded": master and moments whole again first; a no-op for the all-reduce plans)
    flat = trainer.eng.params.flat
    w = torch.arange(1, 1025, device=device, dtype=torch.float64).repeat((flat.numel() + 1023) // 1024)[:flat.numel()]
    s = torch.stack([flat.double().sum(), (flat.double() * w).sum(), trainer.m.double().sum(), trainer.v.double().sum()])
    lo_, hi_ = s.clone(), s.clone()
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo_, hi_)) and bool(torch.isfinite(s).all())


def reference_loop(args, device, steps=40, warmup=5):
    """The reference's OWN calling convention on the drop-in module, unchanged (engine_upsampling.py:69-100,
    util/misc.py:292-305, main_lidar_upsampling.py:282-283): torch.autocast around model(lo, hi), GradScaler
    scale -> backward -> unscale_ -> step -> update, torch.optim.AdamW(betas=(0.9, 0.95)) over timm-style decay groups,
    the per-iteration loss read-back and torch.cuda.synchronize().  The module's forward and backward are one HIP-graph replay
    each behind the autograd bridge (TulipEngine._module_sequence; first call eager, second captures); optimizer, scaler and
    the three host synchronisations per step are PyTorch's own: what a user gets with zero edits, next to the fused
    Trainer step of the headline (host + device timeline of one step: profiles/r4_refloop_timeline.txt)."""
    model = make_model(args).to(device).train()
    decay = [p for p in model.parameters() if p.ndim > 1]
    no_decay = [p for p in model.parameters() if p.ndim <= 1]
    opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.01}],
                            lr=5e-4, betas=(0.9, 0.95))
    scaler = torch.amp.GradScaler("cuda")
    lo, hi = synthetic(args, 0, device)
    opt.zero_grad()

    def one():
        with torch.autocast("cuda"):
            _, total_loss, pixel_loss = model(lo, hi, eval=False)
        v = total_loss.item()
        pixel_loss.item()
        scaler.scale(total_loss).backward()
        scaler.unscale_(opt)
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        torch.cuda.synchronize()
        return v
    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per = []
    for _ in range(steps):
        t1 = time.perf_counter()
        v = one()                                   # (ends with the loop's own synchronize: a step's wall time is its own)
        per.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    per.sort()
    del model, opt
    torch.cuda.empty_cache()
    return {"metric": "range-images/sec training, the reference's loop body unchanged (autocast + GradScaler + "
                      "torch.optim.AdamW + loss.item() + synchronize per step) on the drop-in module",
            "value": round(args.batch * steps / dt, 2), "unit": "range-images/s", "ms_per_step": round(dt / steps * 1e3, 4),
            # (a host-bound loop of a few milliseconds: one hiccup of the box moves the mean of 20 steps by 5 %; the median says
            # what a step takes)
            "ms_per_step_median": round(per[len(per) // 2] * 1e3, 4), "ms_per_step_min": round(per[0] * 1e3, 4),
            "steps": steps, "warmup": warmup, "final_loss": round(v, 6), "grad_scale": scaler.get_scale()}


def secondary_eval_forward(args, device, iters=50, warmup=5):
    """SURVEY 8(d) "Secondary: eval-forward images/s" -- the reference's call site is engine_upsampling.py:168-171
    (`model(images_low_res, images_high_res, eval=True)` under no_grad) and the MC-dropout tile of 8 (:417-419).  The eval
    forward of the headline model at batch 8 and 64 as a HIP graph (tulip_amd.infer.GraphedForward: the fused blocks in
    their inference form, nothing a backward would read is written), HIP events around `iters` replays."""
    import copy
    from tulip_amd.infer import GraphedForward
    a = copy.copy(args)
    res = {"metric": "range-images/sec eval forward (KITTI 16->64x1024, bf16, HIP-graph replay)", "unit": "range-images/s",
           "iters": iters, "warmup": warmup}
    model = make_model(a).to(device).eval()
    for Bi in (8, 64):
        a.batch = Bi
        lo, _ = synthetic(a, 0, device)
        gf = GraphedForward(model, Bi, device)
        gf(lo)
        for _ in range(warmup):
            gf()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            gf()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / iters
        if not bool(torch.isfinite(gf.pred).all()):
            raise SystemExit("eval forward produced non-finite predictions")
        res[f"batch{Bi}"] = {"value": round(Bi / ms * 1e3, 1), "ms_per_forward": round(ms, 4),
                             "mfma_frac": round(Bi / ms * 1e3 * FLOP_FWD_BWD_PER_IMG / 3 / PEAK_BF16, 5)}
        del gf
    del model
    torch.cuda.empty_cache()
    return res


def secondary_batch64(args, device, steps=30, warmup=15, attn_fp8=False):
    """BASELINE.json configs[4]: the same step at per-GPU batch 64, where the kernels rather than the launch chain set
    the pace (secondary metric).  attn_fp8=False: bf16 operands throughout like the headline; True: the configuration as
    BASELINE names it, attention scores Q.K^T from e4m3 operands on the fp8 MFMA (Trainer(attn_fp8=True))."""
    import copy
    from tulip_amd.trainer import Trainer
    a = copy.copy(args)
    a.batch = 64
    model = make_model(a).to(device).train()
    tr = Trainer(model, 64, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01, device=device, use_graph=not args.no_graph,
                 attn_fp8=attn_fp8)
    lo, hi = synthetic(a, 0, device)
    tr.load_batch(lo, hi)
    for _ in range(warmup):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"metric": "range-images/sec training (KITTI 16->64x1024, bs=64/GPU, " +
                     ("fp8 MFMA attention scores, bf16 elsewhere)" if attn_fp8 else "bf16)"), "value": round(64 * steps / dt, 2),
           "unit": "range-images/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "warmup": warmup,
           "step_mfma_frac": round(64 * steps / dt * FLOP_FWD_BWD_PER_IMG / PEAK_BF16, 5),
           "step_oplevel_bytes_convention_over_hbm_peak": round(64 * steps / dt * 3 * BYTES_FWD_OPLEVEL_PER_IMG / PEAK_HBM, 5)}
    del tr, model
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (reference: bs 8/GPU)")
    ap.add_argument("--model", default="tulip_base")
    ap.add_argument("--img", type=int, nargs=2, default=[16, 1024])
    ap.add_argument("--target", type=int, nargs=2, default=[64, 1024])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--grad-dtype", default="auto", choices=["auto", "fp32", "bf16"],
                    help="dtype of the gradient all-reduce (N>1): fp32 = DistributedDataParallel's exchange; auto = fp32 unless "
                         "the measured bus bandwidth leaves it exposed and bf16 does not (tulip_amd.ddp.choose_comm_plan)")
    ap.add_argument("--bucket-adamw", default="auto", choices=["auto", "on", "off"],
                    help="N>1: the optimizer step per bucket behind that bucket's all-reduce (on) or once behind the last (off)")
    ap.add_argument("--exchange", default="allreduce", choices=["allreduce", "sharded"],
                    help="N>1: allreduce (default: per-bucket all-reduce, the plans choose_comm_plan picks from) or sharded (optional, "
                         "never the default: reduce-scatter -> AdamW on the owned shard -> all-gather of the bf16 shadow, "
                         "tulip_amd.ddp.ShardedExchange)")
    ap.add_argument("--bucket-mb", type=float, default=16.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--cpu-warmup", type=int, default=2)
    ap.add_argument("--no-secondary", action="store_true", help="skip the batch-64 line (BASELINE config 5, bf16)")
    ap.add_argument("--no-reference-loop", action="store_true",
                    help="skip the secondary line that times the reference's own loop body on the drop-in module")
    ap.add_argument("--watchdog", type=float, default=float(os.environ.get("TULIP_BENCH_WATCHDOG", "240")),
                    help="N>1: seconds without host-side progress before a rank reports where it hangs and exits 124")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks here (the driver's `python bench.py --gpus N`); under torch.distributed.run
        # the environment carries WORLD_SIZE and this process IS a rank
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: torch.distributed.run --nproc-per-node must equal --gpus")
    # rehearsal of the N > 1 path on a one-GPU box (dev only): TULIP_BENCH_BACKEND=gloo lets the ranks share cuda:0
    backend = os.environ.get("TULIP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    elif world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU over RCCL)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    trainer_ref = [None]
    if world > 1:
        start_watchdog(rank, args.watchdog, trainer_ref)
        phase("init_process_group")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", device_id=device)
        else:
            dist.init_process_group(backend=backend, init_method="env://")
        if dist.get_world_size() != world:
            raise SystemExit(f"process group reports {dist.get_world_size()} ranks, expected {world}")

    from tulip_amd.trainer import Trainer
    phase("model")
    model = make_model(args).to(device).train()
    smoke = None
    if world > 1:
        phase("collective_smoke")
        # the largest gradient bucket of tulip_base (stage 3: 66 MB fp32) before any graph is captured
        smoke = collective_smoke(device, world, 66 << 20)
    # ---- N > 1: the exchange is CHOSEN from the bus bandwidth collective_smoke just measured, before anything is captured
    # (tulip_amd.ddp.choose_comm_plan: every candidate with its prediction goes into `comm`), then measured; if what was
    # chosen leaves more exchange exposed than predicted the alternative is measured too and the faster one is the line.
    plan_info, plans_tried, fallback = None, [], None
    grad_dtype = "fp32" if args.grad_dtype == "auto" else args.grad_dtype
    bucket_adamw = {"auto": None, "on": True, "off": False}[args.bucket_adamw]
    if world > 1:
        from tulip_amd.ddp import choose_comm_plan, plan_buckets
        phase("plan")
        eng0 = model.engine()
        eng0.bind(device)
        buckets = plan_buckets(eng0.params.groups, eng0.params.total, int(args.bucket_mb * (1 << 20) / 4))
        bw = float(os.environ.get("TULIP_BENCH_FAKE_BUSBW_GBPS", "0")) or smoke["bucket"]["busbw_GBps"]
        lat = float(os.environ.get("TULIP_BENCH_FAKE_LATENCY_MS", "-1"))
        lat = smoke["small"]["ms"] if lat < 0 else lat
        plan_info = choose_comm_plan(buckets, world, bw, lat, requested_dtype=args.grad_dtype,
                                     requested_bucket_adamw=bucket_adamw)
        plan_info["busbw_source"] = ("TULIP_BENCH_FAKE_BUSBW_GBPS (rehearsal)" if os.environ.get("TULIP_BENCH_FAKE_BUSBW_GBPS")
                                     else "collective_smoke, largest bucket")
        grad_dtype, bucket_adamw = plan_info["chosen"]["grad_dtype"], plan_info["chosen"]["bucket_adamw"]
        if args.exchange == "sharded":           # asked for explicitly: fp32 reduce-scatter, the optimizer per bucket by construction
            grad_dtype, bucket_adamw = "fp32", True
            plan_info["reason"] += "; --exchange sharded requested (not one of the chooser's candidates)"

    def run_plan(gd, ba):
        """One Trainer on the chosen exchange: W warm-up steps, then EXACTLY K timed steps between barrier + synchronize."""
        phase(f"trainer {gd} bucket_adamw={ba}")
        tr = Trainer(model, args.batch, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01, device=device,
                     use_graph=not args.no_graph, grad_dtype=gd, bucket_mb=args.bucket_mb, bucket_adamw=ba,
                     exchange=args.exchange if world > 1 else "allreduce")
        trainer_ref[0] = tr
        lo, hi = synthetic(args, rank, device)
        tr.load_batch(lo, hi)
        phase("warmup")
        for _ in range(args.warmup):
            tr.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # wall clock brackets the K steps (the contract's number); one HIP event per step boundary on the launch stream
        # gives the per-step distribution (median and min)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        phase("timed")
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            ls = tr.step()
            marks[i + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        ps = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
        if world > 1:
            tmax = torch.tensor([dt_], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = tmax.item()
        return tr, dt_, ps, ls

    def drop(tr):
        trainer_ref[0] = None
        del tr
        import gc
        gc.collect()
        torch.cuda.empty_cache()

    trainer, dt, per_step, losses = run_plan(grad_dtype, bucket_adamw)
    comm = None
    if world > 1:
        phase("replica_check")
        same = replicas_identical(trainer, device)
        if not same:
            # the chosen structure let the replicas drift apart: not a number.  Back to the most conservative exchange (the
            # order DistributedDataParallel + optimizer.step() has), said so in the line.
            fallback = {"reason": "replicas diverged", "plan": {"grad_dtype": grad_dtype, "bucket_adamw": bucket_adamw,
                                                              "detach_buckets": trainer.detach_buckets}}
            drop(trainer)
            os.environ["TULIP_DETACH_BUCKETS"] = "0"
            grad_dtype, bucket_adamw = "fp32", False
            args.exchange = "allreduce"
            trainer, dt, per_step, losses = run_plan(grad_dtype, bucket_adamw)
            same = replicas_identical(trainer, device)
        phase("comm_report")
        comm = comm_report(trainer, args, world, device)
        comm["replicas_identical"] = same
        plans_tried.append({"grad_dtype": grad_dtype, "bucket_adamw": bool(trainer.bucket_adamw), "ms_per_step": round(dt / args.steps * 1e3, 4),
                            "exposed_exchange_ms": comm["exposed_exchange_ms"]})
        step_ms = dt / args.steps * 1e3
        adapt = (args.grad_dtype == "auto" and args.exchange == "allreduce" and fallback is None and os.environ.get("TULIP_BENCH_ADAPT", "1") != "0"
                 and grad_dtype == "fp32" and comm["exposed_exchange_ms"] > max(0.15, 0.07 * step_ms))
        if adapt:
            # the prediction said fp32 hides; the measurement says it does not: measure the bf16 exchange as well
            keep = (trainer.bucket_adamw, dt, per_step, losses, comm)
            drop(trainer)
            trainer, dt2, per2, losses2 = run_plan("bf16", bucket_adamw)
            same2 = replicas_identical(trainer, device)
            comm2 = comm_report(trainer, args, world, device)
            comm2["replicas_identical"] = same2
            plans_tried.append({"grad_dtype": "bf16", "bucket_adamw": bool(trainer.bucket_adamw),
                                "ms_per_step": round(dt2 / args.steps * 1e3, 4), "exposed_exchange_ms": comm2["exposed_exchange_ms"]})
            if same2 and dt2 < 0.97 * dt:
                grad_dtype, dt, per_step, losses, comm = "bf16", dt2, per2, losses2, comm2
            else:       # fp32 stays the line; the trainer left alive is the bf16 one, which only matters to the roofline (N = 1)
                _, dt, per_step, losses, comm = keep
    loss_val = losses[0].item()
    if not (loss_val == loss_val and abs(loss_val) < 1e9):
        raise SystemExit(f"non-finite loss {loss_val}")
    imgs = args.batch * world * args.steps
    value = imgs / dt

    headline = args.model == "tulip_base" and tuple(args.img) == (16, 1024) and tuple(args.target) == (64, 1024) and args.batch == 8
    out = {"metric": "range-images/sec training (KITTI 16->64x1024, bs=8/GPU)" if headline else
           f"range-images/sec training ({args.model} {args.img[0]}x{args.img[1]}->{args.target[0]}x{args.target[1]}, bs={args.batch}/GPU)",
           "value": round(value, 2),
           "unit": "range-images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 4), "step_ms_median": round(per_step[len(per_step) // 2], 4),
           "step_ms_min": round(per_step[0], 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"{args.model} {args.img[0]}x{args.img[1]}->{args.target[0]}x{args.target[1]} "
                                  f"training step (fwd+L1+bwd+allreduce+AdamW), per-GPU batch {args.batch}, "
                                  "window 2x8, patch 1x4, DropPath 0.1, reference init seed 0",
                      "global_batch": args.batch * world, "parallelism": f"dp{world}",
                      "hip_graph": not args.no_graph},
           "final_loss": round(loss_val, 6),
           "step_mfma_frac": round(value * FLOP_FWD_BWD_PER_IMG / world / PEAK_BF16, 5)
           if args.model == "tulip_base" and tuple(args.img) == (16, 1024) else None,
           # NOT an achieved bandwidth: SURVEY.md 8(d)'s op-level byte convention (3 x 245 MB per image: the unfused
           # forward's op traffic x3 for training) over the step time -- bytes the fused kernels never move
           "step_oplevel_bytes_convention_over_hbm_peak": round(value * 3 * BYTES_FWD_OPLEVEL_PER_IMG / world / PEAK_HBM, 5)
           if args.model == "tulip_base" and tuple(args.img) == (16, 1024) else None}
    out["tolerance"] = ("index ops bit-exact; loss within 1e-3 rel of the reference's fp32 forward; prediction inside the "
                        "reference's own bf16-autocast band (max 8e-3 abs); gradients <= 1.5e-2 rel L2 per tensor "
                        "(tests/test_model_gpu.py)")
    if world > 1:
        out["comm"] = comm
        comm["world_size_rccl"] = dist.get_world_size()
        comm["backend"] = dist.get_backend()
        # graphs per step: the chain's segments (cut at the bucket points) + the detached last side group of each bucket
        comm["graph_segments"] = len(trainer._segments[True]) if getattr(trainer, "_segments", None) else 0
        comm["detached_bucket_graphs"] = len(getattr(trainer, "_det_graphs", {}))
        comm["detach_buckets"] = bool(trainer.detach_buckets)
        comm["collective_smoke"] = smoke
        comm["launcher"] = "bench.py spawn_ranks" if os.environ.get("TULIP_BENCH_SPAWNED") else "external (torch.distributed.run)"
        # the choice, its prediction, and what was measured (plans_tried[-1] or the faster of two is the line above)
        comm["plan_chosen"] = plan_info["chosen"]
        comm["plan_reason"] = plan_info["reason"]
        comm["plan_candidates"] = plan_info["candidates"]
        comm["plan_model"] = dict(plan_info["model"], busbw_source=plan_info["busbw_source"])
        comm["predicted_exposed_exchange_ms"] = plan_info["chosen"]["predicted_exposed_exchange_ms"]
        comm["plans_tried"] = plans_tried
        comm["plan_fallback"] = fallback
        out["config"]["grad_allreduce_dtype"] = grad_dtype
    # a repeat after a failed attempt (spawn_ranks' ladder): say so in the line itself
    attempts = json.loads(os.environ.get("TULIP_BENCH_ATTEMPTS", "[]"))
    # what the timed step WAS, readable without opening `comm`: one captured graph (N = 1), graph segments cut at the bucket
    # points with each bucket's last side group as a detached graph (the N > 1 default), the same without the detached graphs
    # (first rung of the retry ladder), or eager launches (last rung / --no-graph)
    segs = len(trainer._segments[True]) if getattr(trainer, "_segments", None) else 0
    dets = len(getattr(trainer, "_det_graphs", {}))
    from tulip_amd import knobs
    out["config"]["knobs_non_default"] = knobs.non_default()      # an A/B switch in the environment shows in the line itself
    out["config"]["step_structure"] = {
        "form": ("eager" if not trainer.use_graph else
                 "one_graph_captured_collectives" if getattr(trainer, "step_form", "") == "one_graph_captured_collectives" else
                 "one_graph" if segs <= 1 and not dets else "segments+detached_buckets" if dets else "segments"),
        "graph_collectives_fell_back": bool(world > 1 and trainer.use_graph and getattr(trainer, "exchange", "") == "allreduce"
                                            and dist.get_backend() == "nccl" and not getattr(trainer, "graph_collectives", False)
                                            and not knobs.is_zero("TULIP_GRAPH_COLLECTIVES")),
        "graph_segments": segs, "detached_bucket_graphs": dets,
        "optimizer": ("in_weight_gradient_write_out" if world == 1 and getattr(trainer, "fuse_adamw", False) else
                      "sharded_per_bucket" if getattr(trainer, "exchange", "allreduce") == "sharded" else
                      "per_bucket" if getattr(trainer, "bucket_adamw", False) else "end_of_step"),
        "exchange": getattr(trainer, "exchange", "allreduce"),
        "ladder_rung": len(attempts), "after_failed": [a.get("attempt", a) if isinstance(a, dict) else a for a in attempts]}
    if attempts or fallback:
        out["graph_path_failed"] = True
        out["attempts"] = attempts
    if rank == 0 and world == 1 and not args.no_secondary and args.batch == 8 and args.model == "tulip_base":
        out["secondary"] = secondary_batch64(args, device)
        out["secondary_fp8_attention"] = secondary_batch64(args, device, attn_fp8=True)
    if rank == 0 and world == 1 and not args.no_reference_loop and headline:
        out["secondary_reference_loop"] = reference_loop(args, device)
    if rank == 0 and world == 1 and not args.no_secondary and headline:
        out["secondary_eval_forward"] = secondary_eval_forward(args, device)
    if rank == 0 and world == 1 and not args.no_roofline:
        out["roofline"] = kernel_rooflines(trainer)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    phase("done")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
