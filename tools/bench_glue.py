"""Isolated timing of the fused stage-boundary launches (csrc/glue.hip) against the launch sequences they replace, at the shapes of
the KITTI tulip_base step (batch from argv, default 8).  HIP events, 50 back-to-back repetitions after 5 warm-ups.
usage: python tools/bench_glue.py [batch=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tulip_amd import ops

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H0, W0, E = 16, 256, 96


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def packed(w, t=False):
    dst = torch.zeros(w.numel(), dtype=torch.bfloat16, device=DEV)
    it, n = ops.pack_items([(w, dst, w.shape[0], w.shape[1], int(t))])
    ops.pack_bf16_multi(it, n)
    return dst


bfr = lambda *s: (torch.randn(*s, device=DEV) * 0.05).bfloat16()
print(f"batch {B}: fused launch vs the sequence it replaces, us (isolated, warm)")
for s in range(3):                       # PatchMerging of level s
    H, W, Cin = H0 >> s, W0 >> s, E << s
    K, N, rows = 4 * Cin, 2 * Cin, B * (H // 2) * (W // 2)
    x = torch.randn(B, H, W, Cin, device=DEV); ga, be = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    w = bfr(N, K); wp = packed(w)
    xm = torch.empty(rows, K, dtype=torch.bfloat16, device=DEV); mean = torch.empty(rows, device=DEV); rstd = torch.empty(rows, device=DEV)
    y = torch.empty(rows, N, device=DEV); cat = torch.zeros(rows, 2 * N, dtype=torch.bfloat16, device=DEV)
    if ops.merge_fwd_supported(Cin, B, H, W):
        tf = timeit(lambda: ops.merge_fwd(x=x, gamma=ga, beta=be, w_packed=wp, xm=xm, mean=mean, rstd=rstd, y=y,
                                          y_bf16=cat.data_ptr() + 2 * N, ld_bf16=2 * N, B=B, H=H, W=W, Cin=Cin, eps=1e-6))
    else:
        tf = float("nan")
    def seq():
        ops.layernorm_fwd(x, ga, be, xm, mean, rstd, rows, K, 1e-6, merge=True, B=B, H=H, W=W)
        ops.gemm(xm, w, rows, N, K, lda=K, ldb=K, epi=ops.EPI_F32, out=y, out2=cat.data_ptr() + 2 * N, ldo2=2 * N)
    print(f"merge_fwd   level {s} rows {rows:6d} K {K:5d} N {N:4d}: fused {tf:7.2f}   sequence (2 launches) {timeit(seq):7.2f}")
for s in range(1, 4):                    # PatchMerging backward into level s-1 (+ the x_save half of the skip at level s)
    Cp = E << (s - 1); H, W = H0 >> (s - 1), W0 >> (s - 1)
    Cs, K4, rows = 2 * Cp, 4 * Cp, B * (H // 2) * (W // 2)
    skip = s < 3
    xprev = torch.randn(B, H, W, Cp, device=DEV); ga = torch.ones(K4, device=DEV)
    mean, rstd = torch.zeros(rows, device=DEV), torch.ones(rows, device=DEV)
    wred, wskip = bfr(Cs, K4), bfr(Cs, 2 * Cs)
    dx_in = torch.randn(rows, Cs, device=DEV); dys = bfr(rows, Cs)
    dyb = torch.zeros(rows, Cs, dtype=torch.bfloat16, device=DEV); dxm = torch.empty(rows, K4, dtype=torch.bfloat16, device=DEV)
    dxp = torch.empty_like(xprev)
    R2 = ops.layernorm_bwd_partial_rows(rows, K4); part2 = torch.zeros(max(R2, 1), 2 * K4, device=DEV)
    def seq():
        if skip:
            ops.gemm(dys, wskip.data_ptr() + 2 * Cs, rows, Cs, Cs, lda=Cs, ldb=2 * Cs, b_trans=True, epi=ops.EPI_F32, out=dx_in, ldo=Cs,
                     accumulate=True, out2=dyb, ldo2=Cs)
        ops.gemm(dyb, wred, rows, K4, Cs, lda=Cs, ldb=K4, b_trans=True, epi=ops.EPI_BF16, out=dxm, ldo=K4)
        ops.layernorm_bwd(dxm, xprev, mean, rstd, ga, None, dxp, rows, K4, merge=True, B=B, H=H, W=W, param_partials=part2)
    ts = timeit(seq)
    tf = float("nan")
    if ops.merge_bwd_supported(Cp, B, H, W):
        R = ops.merge_bwd_partial_rows(Cp, B, H, W); part = torch.zeros(R, 2 * K4, device=DEV)
        w2t, wrt = packed(wskip, True), packed(wred, True)
        tf = timeit(lambda: ops.merge_bwd(dx_in=dx_in if skip else None, dy_skip=dys if skip else None, w_skip_t_packed=w2t if skip else None,
                                          dyb=dyb, w_red_t_packed=wrt, x_prev=xprev, mean=mean, rstd=rstd, gamma=ga, dx_prev=dxp,
                                          param_partials=part, B=B, H=H, W=W, Cp=Cp))
    print(f"merge_bwd   into level {s - 1} rows {rows:6d} Cp {Cp:4d} skip {int(skip)}: fused {tf:7.2f}   sequence ({3 if skip else 2} launches) {ts:7.2f}")
for s in range(3, 0, -1):                # PatchUnmerging of level s -> skip Linear of level s-1
    C = E << s; H, W = H0 >> s, W0 >> s
    Cf, M = C // 2, B * H * W
    xb = bfr(M, C); wexp, bexp = bfr(2 * C, C), torch.zeros(2 * C, device=DEV); wskip, bskip = bfr(Cf, C), torch.zeros(Cf, device=DEV)
    cat = torch.zeros(4 * M, C, dtype=torch.bfloat16, device=DEV); out = torch.empty(4 * M, Cf, device=DEV)
    def seq():
        ops.gemm(xb, wexp, M, 2 * C, C, lda=C, ldb=C, epi=ops.EPI_PIXSHUF2_F32, bias=bexp, out=None, out2=cat, ldo2=C, psH=H, psW=W)
        ops.gemm(cat, wskip, 4 * M, Cf, C, lda=C, ldb=C, epi=ops.EPI_F32, bias=bskip, out=out)
    ts = timeit(seq); tf = float("nan")
    if ops.unmerge_skip_supported(C, B, H, W):
        we, wsk = packed(wexp), packed(wskip)
        tf = timeit(lambda: ops.unmerge_skip_fwd(x_bf16=xb, w_expand_packed=we, b_expand=bexp, cat=cat, w_skip_packed=wsk, b_skip=bskip,
                                                 out=out, B=B, H=H, W=W, C=C))
    dys = bfr(4 * M, Cf); dz = torch.zeros(M, 2 * C, dtype=torch.bfloat16, device=DEV); dx = torch.empty(M, C, device=DEV)
    cast = torch.zeros(M, C, dtype=torch.bfloat16, device=DEV)
    def seqb():
        ops.gemm(dys, wskip, 4 * M, Cf, Cf, lda=Cf, ldb=C, b_trans=True, epi=ops.EPI_UNSHUF2_BF16, out=dz, ldo=2 * C, psH=H, psW=W)
        ops.gemm(dz, wexp, M, C, 2 * C, lda=2 * C, ldb=C, b_trans=True, epi=ops.EPI_F32, out=dx, ldo=C, out2=cast, ldo2=C)
    tsb = timeit(seqb); tfb = float("nan")
    if ops.unmerge_skip_supported(C, B, H, W):
        wst, wet = packed(wskip, True), packed(wexp, True)
        tfb = timeit(lambda: ops.skip_unmerge_bwd(dy_skip=dys, w_skip_t_packed=wst, dz=dz, w_expand_t_packed=wet, dx=dx, dx_bf16=cast,
                                                  B=B, H=H, W=W, C=C))
    print(f"unmerge_skip level {s}->{s - 1} M {M:6d} C {C:4d}: fwd fused {tf:7.2f} sequence {ts:7.2f}   bwd fused {tfb:7.2f} sequence {tsb:7.2f}")
