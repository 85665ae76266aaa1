"""dev tool: per-phase shader-clock profile of the fused C = 96 block kernels (tulip_swin96_block_{fwd,bwd}_profiled) + launch time.
usage: python tools/swin96_phases.py [batch=8] [recompute=1]"""
import os as _os
_os.environ.setdefault("TULIP_HIP_DEV", "1")     # the profiled twins live in libtulip_hip_dev.so (include/tulip_hip.h, conventions)
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tulip_amd import ops
from tulip_amd.model.tulip import tulip_base

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lean = (sys.argv[2] if len(sys.argv) > 2 else "0") != "0"
hgrad = (sys.argv[3] if len(sys.argv) > 3 else "1") != "0"
torch.manual_seed(0)
m = tulip_base(img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1, window_size=[2, 8],
               pixel_shuffle=True, circular_padding=True, log_transform=True, patch_unmerging=True).cuda().train()
eng = m.engine()
eng.bind(torch.device("cuda", 0))
eng.params.refresh_shadow()
eng.recompute96 = lean
eng.fc1_grad96 = hgrad
P = eng.plan(B)
sp = eng.enc_blocks[0][1]
M, C = B * sp.H * sp.W, sp.C
xin = P["enc0.in"]
xin.copy_((torch.randn(M, C, device="cuda") * 1.5).view_as(xin))
eng.draw_drop_scales(P, False)
out = torch.empty(M, C, device="cuda")
R = ops.swin96_bwd_partial_rows(B, sp.H, sp.W)
FW = ["x + bias loads, weight staging, norm1", "wait barrier", "qkv gemm (+stores)", "attention (3 heads)", "wait barrier",
      "fc1 rows -> LDS, proj, x1, norm2", "restage barriers", "fc1 / gelu / fc2 loop", "x_out stores"]
BW = ["weight staging, dy / stats loads, norm2 recompute", "wait barrier", "MLP half loop", "stage loads, x1, norm2', dyb_a",
      "wait barrier", "stage stores, x row, norm1 recompute, barrier", "qkv recompute", "proj'", "attention'",
      "dqkv stores, x row, qkv'", "norm1', dx stores", "wait barrier", "partial rows"]


def report(tag, stamps, names):
    n = len(names) + 1
    s = stamps.view(R, 8, 16)[:, :, :n].cpu().double()
    d = s[:, :, 1:] - s[:, :, :-1]
    tot = s[:, :, n - 1] - s[:, :, 0]
    print(f"{tag} B={B} lean={lean} hgrad={hgrad}: {R} workgroups x 8 waves; per-wave total cycles mean {tot.mean():.0f} max {tot.max():.0f}")
    for k, nm in enumerate(names):
        print(f"  {nm:52s} mean {d[:, :, k].mean():9.0f}  max {d[:, :, k].max():9.0f}")
    print(f"  first start -> last end: {s[:, :, n - 1].max() - s[:, :, 0].min():.0f} cycles; start skew {s[:, :, 0].max() - s[:, :, 0].min():.0f}")


st = torch.zeros(R * 8 * 16, dtype=torch.int64, device="cuda")
real_f, real_b = ops.swin96_block_fwd, ops.swin96_block_bwd
for it in range(3):
    ops.swin96_block_fwd = lambda **kw: real_f(stamps=st, **kw)
    eng._block_fwd(P, sp, xin, out)
ops.swin96_block_fwd = real_f
torch.cuda.synchronize()
report("forward", st, FW)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
eng._block_fwd(P, sp, xin, out)
e0.record()
for _ in range(20):
    eng._block_fwd(P, sp, xin, out)
e1.record(); e1.synchronize()
print(f"forward: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (20 back to back)")
# ---- backward (weight gradients / folds are queued and dropped: only the chain's launch is of interest)
gflat = torch.zeros(eng.params.total, device="cuda")
G = lambda name: gflat.data_ptr() + 4 * eng.params.offset[name]
dx = torch.randn(M, C, device="cuda")
eng._gflat = gflat
st.zero_()


def bwd(prof):
    eng._pending, eng._lagged_hook, eng._deferred, eng._carry = [], None, None, ()
    if prof:
        ops.swin96_block_bwd = lambda **kw: real_b(stamps=st, **kw)
    try:
        eng._block_bwd(P, sp, xin, dx, G, have_dyb=False)
    finally:
        ops.swin96_block_bwd = real_b
    eng._pending = []


saved = eng.overlap_wgrad
for it in range(3):
    bwd(True)
torch.cuda.synchronize()
report("backward", st, BW)
# time the launch alone
rec = []
ops.swin96_block_bwd = lambda **kw: rec.append(kw) or real_b(**kw)
bwd(False)
ops.swin96_block_bwd = real_b
torch.cuda.synchronize()
kw = rec[0]
real_b(**kw)
e0.record()
for _ in range(20):
    real_b(**kw)
e1.record(); e1.synchronize()
print(f"backward: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (20 back to back)")
