"""dev tool: what bounds the step -- time the captured training step with parts of the side work removed
(results are wrong then; timing only).  usage: python tools/exp_chain.py [batch=8]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tulip_amd.model.tulip import tulip_base
from tulip_amd.trainer import Trainer
from tulip_amd import engine as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def run(tag, patch):
    torch.manual_seed(0)
    m = tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
                   pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).cuda().train()
    eng = m.engine()
    patch(eng)
    tr = Trainer(m, B)
    x = torch.rand(B, 1, 16, 1024, device="cuda"); y = torch.rand(B, 1, 64, 1024, device="cuda")
    tr.load_batch(x, y)
    for _ in range(10):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        tr.step()
    torch.cuda.synchronize()
    print(f"{tag:58s} {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step", flush=True)


def no_wgrad(eng):
    eng._wgrad = lambda *a, **k: None


def no_side(eng):
    eng._wgrad = lambda *a, **k: None
    eng._fold = lambda *a, **k: None
    eng._fold_bias_table = lambda *a, **k: None


def no_folds(eng):
    from tulip_amd import ops
    real = ops.wgrad_group
    ops.wgrad_group = lambda items, extra, ws, ws_bytes, fold=True, adam=None, small_tiles=False: real(items, [], ws, ws_bytes, fold=False, adam=adam, small_tiles=small_tiles)
    ops.reduce_rows_multi = lambda *a, **k: None


def no_adamw(eng):
    Trainer._adamw = lambda self: None


run("full step", lambda e: None)
run("no weight-gradient GEMMs (folds stay; every weight is then stepped by the full AdamW launch at the end)", no_wgrad)
from tulip_amd import ops as _ops
_rg, _rr = _ops.wgrad_group, _ops.reduce_rows_multi
run("weight-gradient GEMMs, nothing folded", no_folds)
_ops.wgrad_group, _ops.reduce_rows_multi = _rg, _rr
run("no side work at all (chain only)", no_side)
run("chain only, no AdamW / weight packing either", lambda e: (no_side(e), no_adamw(e)))
