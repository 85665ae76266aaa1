"""dev tool: per-phase shader-clock profile of swinw_fwd_kernel (tulip_swinw_block_fwd_profiled) + launch time.
usage: python tools/swinw_phases.py [stage=1] [batch=8]"""
import os as _os
_os.environ.setdefault("TULIP_HIP_DEV", "1")     # the profiled twins live in libtulip_hip_dev.so (include/tulip_hip.h, conventions)
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tulip_amd import ops
from tulip_amd.model.tulip import tulip_base

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(0)
m = tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
               pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).cuda().train()
eng = m.engine()
eng.bind(torch.device("cuda", 0))
eng.params.refresh_shadow()
P = eng.plan(B)
sp = eng.enc_blocks[stage][1]
M, C = B * sp.H * sp.W, sp.C
xin = P[f"enc{stage}.in"]
xin.copy_((torch.randn(M, C, device="cuda") * 1.5).view_as(xin))
eng.draw_drop_scales(P, False)
out = torch.empty(M, C, device="cuda")
nwv = C // 32
R = ops.swinw_bwd_partial_rows(C, B, sp.H, sp.W)
stamps = torch.zeros(R * nwv * 16, dtype=torch.int64, device="cuda")
real = ops.swinw_block_fwd
for it in range(3):
    ops.swinw_block_fwd = lambda C_, out_bf16=None, **kw: real(C_, out_bf16=out_bf16, stamps=stamps, **kw)
    eng._block_fwd(P, sp, xin, out)
ops.swinw_block_fwd = real
torch.cuda.synchronize()
s = stamps.view(R, nwv, 16).cpu().double()
d = s[:, :, 1:] - s[:, :, :-1]
names = ["norm1 pass", "wait barrier", "qkv gemm", "qkv epilogue", "attention", "wait barrier", "proj gemm",
         "proj epilogue+stats", "wait barrier", "norm2 write", "wait barrier", "fc1 (+gelu, stores)", "wait barrier",
         "fc2 gemm", "fc2 epilogue"]
tot = (s[:, :, 15] - s[:, :, 0])
print(f"stage {stage} C={C} B={B}: {R} workgroups x {nwv} waves; per-wave total cycles mean {tot.mean():.0f} max {tot.max():.0f}")
for k, n in enumerate(names):
    print(f"  {n:24s} mean {d[:, :, k].mean():9.0f}  max {d[:, :, k].max():9.0f}")
span = s[:, :, 15].max() - s[:, :, 0].min()
print(f"first start -> last end: {span:.0f} cycles (all workgroups); start skew {s[:, :, 0].max() - s[:, :, 0].min():.0f}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for fused in (True, False):
    eng.fuse_wide = fused
    eng._block_fwd(P, sp, xin, out)
    e0.record()
    for _ in range(20):
        eng._block_fwd(P, sp, xin, out)
    e1.record(); e1.synchronize()
    print(f"fused={fused}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per block forward")
# ---- backward: fused vs the 7-kernel chain (weight gradients + folds inline on this stream in both cases)
eng.fuse_wide = False
eng._block_fwd(P, sp, xin, out)
eng.overlap_wgrad = False
gflat = torch.zeros(eng.params.total, device="cuda")
G = lambda name: gflat.data_ptr() + 4 * eng.params.offset[name]
dx = torch.randn(M, C, device="cuda")
for fused in (True, False):
    for with_w in (True, False):
        eng.fuse_wide_bwd = fused
        real_w, real_f = eng._wgrad, eng._fold
        real_b = eng._fold_bias_table
        if not with_w:
            eng._wgrad = lambda *a, **k: None
            eng._fold = lambda *a, **k: None
            eng._fold_bias_table = lambda *a, **k: None
        eng._pending = []
        eng._block_bwd(P, sp, xin, dx, G, have_dyb=False, next_cast=None)
        e0.record()
        for _ in range(20):
            eng._block_bwd(P, sp, xin, dx, G, have_dyb=False, next_cast=None)
        e1.record(); e1.synchronize()
        eng._wgrad, eng._fold, eng._fold_bias_table = real_w, real_f, real_b
        print(f"backward fused={fused} weight-gradients={'yes' if with_w else 'no '}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per block")
e0.record()
for _ in range(20):
    eng.params.refresh_transposes()
e1.record(); e1.synchronize()
print(f"fragment-major weight copies of all fused blocks: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
