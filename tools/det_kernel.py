import sys; sys.path.insert(0, "/root/repo")
import torch
from tulip_amd import ops
DEV="cuda"
def packed(w, t=False):
    dst = torch.zeros(w.numel(), dtype=torch.bfloat16, device=DEV)
    it, n = ops.pack_items([(w, dst, w.shape[0], w.shape[1], int(t))]); ops.pack_bf16_multi(it, n); return dst
bfr = lambda *s: (torch.randn(*s, device=DEV) * 0.05).bfloat16()
junk = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
side = torch.cuda.Stream()
for (B,H,W,C) in [(8,8,128,192),(8,4,64,384)]:
    Cf, M = C//2, B*H*W
    dys = bfr(4*M, Cf); wexp = bfr(2*C, C); wskip = bfr(Cf, C)
    wst, wet = packed(wskip, True), packed(wexp, True)
    ref = None; bad = 0
    for it in range(60):
        dz = torch.zeros(M, 2*C, dtype=torch.bfloat16, device=DEV); dx = torch.zeros(M, C, device=DEV)
        junk.fill_(it & 255)                      # evict the caches
        with torch.cuda.stream(side):             # something else on the chip
            junk2 = junk[: 64 << 20].clone()
        ops.skip_unmerge_bwd(dy_skip=dys, w_skip_t_packed=wst, dz=dz, w_expand_t_packed=wet, dx=dx, B=B, H=H, W=W, C=C)
        torch.cuda.synchronize()
        if ref is None: ref = (dz.clone(), dx.clone())
        elif not (torch.equal(dz, ref[0]) and torch.equal(dx, ref[1])):
            bad += 1
            if bad == 1:
                d = (dx != ref[1]); dzb = (dz != ref[0])
                print("  first mismatch: dx elems", int(d.sum()), "rows", d.any(1).nonzero().flatten()[:8].tolist(), "cols", d.any(0).nonzero().flatten()[:16].tolist(),
                      "| dz elems", int(dzb.sum()), "rows", dzb.any(1).nonzero().flatten()[:8].tolist(), "cols", dzb.any(0).nonzero().flatten()[:16].tolist())
    print(f"skip_unmerge_bwd C={C}: {bad}/59 repeats differ")
