"""tulip_gemm_bf16 with TULIP_GEMM_B_PACKED (csrc/gemm.hip, gemm_stream_kernel: the small-K form on the fragment-major copy of the
weight) against the plain call on the row-major matrix: the stage-boundary GEMMs of the batch-8 step, every epilogue they use --
bit for bit (same MFMA order per output element)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from tulip_amd import ops as o
    return o


def packed(ops, w, transpose):
    """fragment-major copy of w ([N][K]) or of its transpose"""
    dst = torch.zeros(w.numel(), dtype=torch.bfloat16, device=DEV)
    it, n = ops.pack_items([(w, dst, w.shape[0], w.shape[1], int(transpose))])
    ops.pack_bf16_multi(it, n)
    return dst


# (M, N, K, b_trans, epilogue, splits): the six chain GEMMs of the KITTI batch-8 step (profiles/r6_chain_gemms.txt) + variants
CASES = [
    (512, 1536, 768, False, "pixshuf", 1),
    (2048, 384, 768, False, "f32_bias_out2", 1),
    (2048, 384, 384, True, "unshuf", 1),
    (512, 768, 1536, True, "f32", 4),
    (512, 768, 1536, True, "f32_bias_out2", 1),         # the same unsplit: the whole K = 1536 in one workgroup
    (512, 1536, 768, True, "bf16", 1),
    (32768, 96, 96, True, "f32_acc", 1),
    (64, 96, 768, False, "resid", 1),
    (96, 192, 768, True, "f32_bias_out2", 2),
]


@pytest.mark.parametrize("M,N,K,bt,epi,splits", CASES)
def test_packed_form_matches_the_plain_gemm_bit_for_bit(ops, M, N, K, bt, epi, splits):
    from tulip_amd import ops as O
    torch.manual_seed(M + N + K)
    A = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
    # the weight as the reference holds it: [N][K] for a forward Linear; [K][N] where the GEMM is a data gradient (B^T = W)
    W = (torch.randn(K, N, device=DEV) * 0.05).bfloat16() if bt else (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
    Wp = packed(ops, W, transpose=bt)
    assert ops.gemm_packed_supported(M, N, K, splits)
    ws = torch.zeros(max(1, splits * M * N), device=DEV)
    bias = torch.randn(N, device=DEV)
    rs = torch.rand(M // 32, device=DEV) + 0.5
    aux = torch.randn(M, N, device=DEV)
    res = []
    for use_packed in (False, True):
        kw = dict(lda=K, ldb=(N if bt else K), b_trans=bt, splits=splits, workspace=ws, workspace_bytes=ws.numel() * 4)
        Bop = W
        if use_packed:
            kw.update(b_trans=False, ldb=K, b_packed=True)
            Bop = Wp
        outs = {}
        if epi == "pixshuf":
            psH, psW = 2, 32
            Bn = M // (psH * psW)
            o32 = torch.full((Bn * 2 * psH * 2 * psW, N // 4), float("nan"), device=DEV)
            cat = torch.zeros(Bn * 2 * psH * 2 * psW, N // 2, dtype=torch.bfloat16, device=DEV)
            ops.gemm(A, Bop, M, N, K, epi=O.EPI_PIXSHUF2_F32, bias=bias, out=o32, out2=cat, ldo2=N // 2, psH=psH, psW=psW, **kw)
            outs = dict(o32=o32, cat=cat)
        elif epi == "unshuf":
            psH, psW = 4, 32                                  # M fine tokens = B * 2psH * 2psW
            Bn = M // (4 * psH * psW)
            o = torch.zeros(Bn * psH * psW, 4 * N, dtype=torch.bfloat16, device=DEV)
            ops.gemm(A, Bop, M, N, K, epi=O.EPI_UNSHUF2_BF16, out=o, ldo=4 * N, psH=psH, psW=psW, **kw)
            outs = dict(o=o)
        elif epi == "bf16":
            o = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            ops.gemm(A, Bop, M, N, K, epi=O.EPI_BF16, bias=bias, out=o, **kw)
            outs = dict(o=o)
        elif epi in ("f32", "f32_acc", "f32_bias_out2"):
            o = torch.full((M, N), 0.25, device=DEV)
            o2 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            ops.gemm(A, Bop, M, N, K, epi=O.EPI_F32, bias=bias if epi != "f32" else None, out=o,
                     accumulate=(epi == "f32_acc"), out2=o2 if epi == "f32_bias_out2" else None,
                     ldo2=N if epi == "f32_bias_out2" else 0, rowscale=rs if epi == "f32_bias_out2" else None, rows_per_sample=32, **kw)
            outs = dict(o=o, o2=o2)
        elif epi == "resid":
            o = torch.zeros(M, N, device=DEV)
            ops.gemm(A, Bop, M, N, K, epi=O.EPI_RESID_F32, bias=bias, out=o, aux=aux, ldaux=N, rowscale=rs, rows_per_sample=32, **kw)
            outs = dict(o=o)
        torch.cuda.synchronize()
        res.append(outs)
    for k in res[0]:
        assert torch.isfinite(res[0][k].float()).all(), k
        assert torch.equal(res[0][k], res[1][k]), (epi, k, (res[0][k].float() - res[1][k].float()).abs().max().item())
    # and against fp32 arithmetic on the same bf16 operands
    if epi == "bf16":
        ref = A.float() @ (W.float() if bt else W.float().t()) + bias
        assert (res[1]["o"].float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


def test_packed_form_refuses_shapes_it_does_not_have(ops):
    assert not ops.gemm_packed_supported(48, 96, 768) and not ops.gemm_packed_supported(64, 64, 768)
    assert ops.gemm_packed_supported(64, 96, 1536) and ops.gemm_packed_supported(64, 96, 1536, 2) and not ops.gemm_packed_supported(64, 96, 3072)
    assert not ops.gemm_packed_supported(64, 96, 256)
    A = torch.zeros(64, 256, dtype=torch.bfloat16, device=DEV)
    W = torch.zeros(96 * 256, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(64, 96, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):
        ops.gemm(A, W, 64, 96, 256, lda=256, ldb=256, out=o, b_packed=True)
