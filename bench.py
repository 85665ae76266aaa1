#!/usr/bin/env python3
"""Headline benchmark: range-images/s of a full TULIP training step on MI355X.

Workload (BASELINE.json configs[1]): tulip_base, KITTI 16x1024 -> 64x1024, per-GPU batch 8, bf16 GEMM
operands / fp32 accumulate, train mode (DropPath active), synthetic inputs of SURVEY.md 8(d) resident
in HBM.  A step = forward + L1 loss + backward + gradient all-reduce (N>1) + fused AdamW, replayed
from HIP graphs.  `python bench.py --gpus N --steps K --warmup W`; for N>1 launch one rank per GPU
with torch.distributed.run (RCCL).  Rank 0 prints ONE JSON line.

Extra objects:  "roofline" -- the dominant kernel (the bf16 MFMA GEMM family): algorithmic FLOPs of
its launches in one step / their summed duration, each launch bracketed by HIP events on the launch
stream in a separate eager pass (graphs cannot be instrumented per node);  "cpu_baseline" -- the
oracle (plain PyTorch fp32 restatement of the reference, oracle/tulip_oracle.py) running the same
training step on the host cores, rank 0, N=1 only, bounded sample.
"""
import argparse
import json
import os
import sys
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
    if -(-M // 128) * gn * eff < 2048 or M <= 64:
        grid = gn * -(-M // 64) * eff
        ksub = 4 if (grid <= 400 and kchunk >= 256) else 1
        bm = 64
    else:
        bm, ksub = 128, 1
    t = lambda f: "true" if f else "false"
    return f"gemm_kernel<{bm}, {t(a_trans)}, {t(b_trans)}, {ksub}>"


def _pmc_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2-corrected +
    WRITE_SIZE, collected as MI355X_MICROARCH.md prescribes); counters cannot be read from inside bench.py."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pmc_gemm_traffic.json")) as f:
            return round(json.load(f)["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def gemm_roofline(trainer, reps=5):
    """Roofline of the dominant kernel family (the bf16 MFMA GEMM: ~84 % of the step's FLOPs; the rest is in the two fused C=96 block kernels).
    The ~170 GEMM launches of one step are recorded in an eager pass, then each is re-issued `reps`
    times back to back between ONE HIP-event pair on the launch stream (graph nodes cannot be
    instrumented, and eager launches would include host gaps).  achieved = sum(algorithmic FLOPs) /
    sum(mean duration); per-instantiation mean durations are listed for comparison with rocprofv3."""
    from tulip_amd import ops
    calls, groups = [], []
    real, real_group = ops.gemm, ops.wgrad_group

    def record(A, B, M, N, K, **kw):
        calls.append((A, B, M, N, K, kw))
        real(A, B, M, N, K, **kw)

    def record_group(items, extra, ws, ws_bytes, fold=True):
        groups.append((list(items), ws, ws_bytes))
        real_group(items, extra, ws, ws_bytes, fold)

    ops.gemm, ops.wgrad_group = record, record_group
    try:
        trainer._fwd_bwd(lambda tag: None)
        torch.cuda.synchronize()
    finally:
        ops.gemm, ops.wgrad_group = real, real_group
    by_inst, tot_t, tot_f, alg_bytes = {}, 0.0, 0.0, 0.0
    # the weight gradients of a block leave as ONE grouped launch of the same tile code (gemm_group_kernel);
    # timed without its fold launch
    for items, ws, ws_bytes in groups:
        real_group(items, [], ws, ws_bytes, fold=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            real_group(items, [], ws, ws_bytes, fold=False)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        f, tiles, deep = 0.0, 0, True
        for it in items:
            eff = ops.gemm_effective_splits(it.Mtok, it.splits)
            kchunk = -(-(-(-it.Mtok // eff)) // 32) * 32
            tiles += -(-it.Kw // 96) * -(-it.Nw // 64) * eff
            deep = deep and kchunk >= 256
            f += 2.0 * it.Nw * it.Kw * it.Mtok
            alg_bytes += 2.0 * it.Mtok * (it.Nw + it.Kw) + 4.0 * it.Nw * it.Kw * eff
        ksub = 4 if (tiles <= 400 and deep) else 1
        d = by_inst.setdefault(f"gemm_group_kernel<64, true, true, {ksub}>", [0, 0.0, 0.0])
        d[0] += 1; d[1] += t; d[2] += f
        tot_t += t; tot_f += f
    for A, B, M, N, K, kw in calls:
        real(A, B, M, N, K, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            real(A, B, M, N, K, **kw)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        f = 2.0 * M * N * K
        alg_bytes += 2.0 * (M * K + N * K) + M * N * (4.0 if kw.get("epi", 0) in (3, 4, 5, 6, 7) else 2.0)
        name = _gemm_instance(M, N, K, kw.get("a_trans", False), kw.get("b_trans", False), kw.get("splits", 1))
        d = by_inst.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1; d[1] += t; d[2] += f
        tot_t += t; tot_f += f
    detail = {k: {"launches": n, "avg_us": round(t / n * 1e6, 2), "tflops": round(f / t / 1e12, 1)}
              for k, (n, t, f) in sorted(by_inst.items(), key=lambda kv: -kv[1][1])}
    return {"bound": "mfma", "kernel": "gemm_kernel / gemm_group_kernel<BM,A_T,B_T,KSUB> (one tile code: every linear / 1x1 conv fwd + dgrad outside the fused C=96 blocks, every wgrad)",
            "achieved": round(tot_f / tot_t / 1e12, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
            "frac": round(tot_f / tot_t / PEAK_BF16, 4), "traffic": _pmc_traffic(), "launches_per_step": len(calls) + len(groups),
            # the same launches against the other roof: PMC HBM bytes per launch / mean launch time / 8 TB/s
            "hbm_frac_of_traffic": (round(_pmc_traffic() / (tot_t / (len(calls) + len(groups))) / PEAK_HBM, 4)
                                    if _pmc_traffic() else None),
            "flops_per_launch": tot_f / (len(calls) + len(groups)),
            "operand_bytes_per_launch": round(alg_bytes / (len(calls) + len(groups))),
            "mean_launch_us": round(tot_t / (len(calls) + len(groups)) * 1e6, 2),
            "gemm_ms_per_step": round(tot_t * 1e3, 3), "flops_per_step": tot_f, "by_kernel": detail}


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
    for it in range(1 + args.cpu_steps):
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
    t = sorted(times[1:])[len(times[1:]) // 2]
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
            "cpu": cpu_name, "sample": f"{args.cpu_steps} timed training steps (median) of batch {B} after 1 warm-up, "
            "fp32 eager PyTorch oracle, same model/config/optimizer"}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=3)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://", device_id=device)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node N")

    from tulip_amd.trainer import Trainer
    model = make_model(args).to(device).train()
    trainer = Trainer(model, args.batch, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01, device=device,
                      use_graph=not args.no_graph)
    lo, hi = synthetic(args, rank, device)
    trainer.load_batch(lo, hi)

    for _ in range(args.warmup):
        trainer.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = trainer.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    loss_val = losses[0].item()
    if not (loss_val == loss_val and abs(loss_val) < 1e9):
        raise SystemExit(f"non-finite loss {loss_val}")
    imgs = args.batch * world * args.steps
    value = imgs / dt

    out = {"metric": "range-images/sec training (KITTI 16->64x1024, bs=8/GPU)", "value": round(value, 2),
           "unit": "range-images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"{args.model} {args.img[0]}x{args.img[1]}->{args.target[0]}x{args.target[1]} "
                                  f"training step (fwd+L1+bwd+allreduce+AdamW), per-GPU batch {args.batch}, "
                                  "window 2x8, patch 1x4, DropPath 0.1, reference init seed 0",
                      "global_batch": args.batch * world, "parallelism": f"dp{world}",
                      "hip_graph": not args.no_graph},
           "final_loss": round(loss_val, 6),
           "step_mfma_frac": round(value * FLOP_FWD_BWD_PER_IMG / world / PEAK_BF16, 5)
           if args.model == "tulip_base" and tuple(args.img) == (16, 1024) else None,
           # SURVEY.md 8(d) op-level convention: 3 x 245 MB per image (forward op traffic x3 for training)
           "step_hbm_frac_oplevel": round(value * 3 * BYTES_FWD_OPLEVEL_PER_IMG / world / PEAK_HBM, 5)
           if args.model == "tulip_base" and tuple(args.img) == (16, 1024) else None}
    if rank == 0 and world == 1 and not args.no_roofline:
        out["roofline"] = gemm_roofline(trainer)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
