"""GPU, round 6.
  * pack_at_end when the marked completion group is the backward's LAST stage (its fork is still deferred
    when the pack point is reached) or no hook carries the mark; the copies must nevertheless be rewritten (a silently skipped
    refresh trains on stale fragment-major weights for ever after);
  * the deep stages' weight copies are maintained only once a plan runs them fused, and a Trainer captured before
    such an activation re-captures."""
import pytest
import torch

from oracle import tulip_oracle as O
from tests.test_model_gpu import build

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("mark", ["enc0", "none"])
def test_pack_at_end_when_the_marked_group_is_late_or_missing(mark, monkeypatch):
    """The pack point of run_backward (pack_at_end) when the marked completion group is the backward's LAST stage -- its fork is still
    deferred when the pack point is reached ("enc0": a model whose stage 0 already had a packed width would mark it; the patch
    embedding's E <= 128 keeps that from being constructible today, so the tag is forced) -- or when no hook carries the mark at all
    ("none"): the copies are rewritten behind everything on the side queue instead of being skipped silently.  Same training as
    the regular mark (bit-identical parameters and moments), copies current after every step."""
    from tulip_amd.trainer import Trainer
    cfg = O.tulip_base_config()
    sd = O.key_seeded_state_dict(cfg, seed=5)
    lo, hi = O.synthetic_batch(cfg, 8, seed=7)
    res = {}
    for forced in (None, mark):
        torch.manual_seed(3)
        m = build(cfg, sd, train=True)
        tr = Trainer(m, 8, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01)
        if forced is not None:
            # (the tag is derived lazily, after the Trainer has planned which copies are rewritten late: a property of the class)
            monkeypatch.setattr(type(tr.eng), "_pack_mark_tag", property(lambda self: forced))
        else:
            monkeypatch.undo()
        tr.load_batch(lo.to(DEV), hi.to(DEV))
        losses = torch.stack([tr.step().clone() for _ in range(4)])
        torch.cuda.synchronize()
        assert tr._pack_at_end and (forced is not None or tr.eng._pack_mark_tag == "enc1")
        W = tr.eng.params
        pk, pkt = W.packed.clone(), W.packed_t.clone()
        W.refresh_transposes()                           # what the copies must already be: rebuilt from the current shadow
        torch.cuda.synchronize()
        assert torch.equal(pk, W.packed) and torch.equal(pkt, W.packed_t)
        res[forced] = (W.flat.clone(), tr.m.clone(), tr.v.clone(), losses)
    from tests.conftest import describe_flat_diff
    for k, (a, b) in enumerate(zip(res[None], res[mark])):
        assert torch.equal(a, b), (k, describe_flat_diff(tr.eng, a, b) if a.numel() == tr.eng.params.total else (a, b))
    assert torch.isfinite(res[None][3]).all()


def test_deep_copies_are_activated_by_the_plan_that_streams_them():
    """tulip_base at batch 64: stage 3 has 256 windows per launch -> the GEMM sequence runs, nobody streams the C = 768 copies and the
    step does not rewrite them; a GraphedForward at batch 8 (32 windows: the sliced deep form) activates them, finds them current, and
    the Trainer re-captures so that they stay current."""
    from tulip_amd.trainer import Trainer
    from tulip_amd.infer import GraphedForward
    cfg = O.tulip_base_config()
    sd = O.key_seeded_state_dict(cfg, seed=11)
    m = build(cfg, sd, train=True)
    B = 64
    lo, hi = O.synthetic_batch(cfg, B, seed=13)
    tr = Trainer(m, B, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01)
    W = tr.eng.params
    assert 768 not in W.pk_active and {192, 384} <= W.pk_active
    tr.load_batch(lo.to(DEV), hi.to(DEV))
    for _ in range(2):
        tr.step()
    epoch = W.pack_epoch
    m.eval()
    gf = GraphedForward(m, 8)
    assert 768 in W.pk_active and W.pack_epoch == epoch + 1
    a = gf(lo[:8].to(DEV)).clone()
    m.train()
    tr.step()                                                # re-captures: the step now rewrites the deep copies too
    assert tr._pack_epoch == W.pack_epoch
    tr.step()
    torch.cuda.synchronize()
    m.eval()
    got = gf(lo[:8].to(DEV)).clone()
    W.refresh_shadow()                                       # casts + repacks everything from the fp32 master
    want = gf(lo[:8].to(DEV)).clone()
    assert torch.equal(got, want) and not torch.equal(a, got)


def test_the_captured_step_is_reproducible_run_to_run():
    """Six trainings of three steps at the bench configuration, fresh module + Trainer each time: parameters and both moments
    bit-identical.  (Round 6: with a backward stage-boundary kernel small enough to share a CU with the side queue's fold launches,
    one register of one quarter-wave of `exp_avg` came out wrong a few times per thousand launches -- csrc/glue.hip WholeCU,
    tools/det_glue.py; this test is what would have caught it.)"""
    from tulip_amd.trainer import Trainer
    from tests.conftest import describe_flat_diff
    cfg = O.tulip_base_config()
    sd = O.key_seeded_state_dict(cfg, seed=5)
    lo, hi = O.synthetic_batch(cfg, 8, seed=7)
    base = None
    for rep in range(6):
        torch.manual_seed(3)
        m = build(cfg, sd, train=True)
        tr = Trainer(m, 8, lr=5e-4, betas=(0.9, 0.95), weight_decay=0.01)
        tr.load_batch(lo.to(DEV), hi.to(DEV))
        for _ in range(3):
            tr.step()
        torch.cuda.synchronize()
        got = (tr.eng.params.flat.clone(), tr.m.clone(), tr.v.clone())
        if base is None:
            base = got
            continue
        for k, (a, b) in enumerate(zip(got, base)):
            assert torch.equal(a, b), (rep, k, describe_flat_diff(tr.eng, a, b))
