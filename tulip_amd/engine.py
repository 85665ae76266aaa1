"""Host-side orchestration of the TULIP hot path on one MI355X.

``TulipEngine`` owns, for one ``tulip_amd.model.tulip.TULIP`` module:

* a **flat fp32 parameter buffer** (the module's nn.Parameters are re-pointed to views of it, so
  ``state_dict`` / optimizers / DDP keep working), ordered by *backward completion* (head first,
  patch embedding last) so gradient buckets can be all-reduced while the rest of the backward is
  still running; a **bf16 shadow** of it feeds the MFMA GEMMs;
* per-batch-size **plans**: statically allocated activation / gradient workspaces, so a whole
  forward+backward(+AdamW) is a fixed launch sequence that is captured once into a HIP graph;
* ``run_forward`` / ``run_backward``: the launch sequences (reference: TULIP.forward,
  tulip.py:702-737, and its autograd backward) expressed purely as C-ABI kernel calls;
* ``autograd_forward``: the ``loss.backward()``-compatible bridge used by ``TULIP.forward``.

Nothing here computes on the CPU and nothing falls back to PyTorch ops for the hot path; PyTorch is
used for device memory, streams, RNG draws for DropPath and graph capture.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import knobs, ops
from ._lib import (EPI_BF16, EPI_F32, EPI_GELU_BWD, EPI_GELU_DUAL, EPI_PIXSHUF2_F32, EPI_RESID_F32,
                   EPI_UNSHUF2_BF16,
                   EPI_SPLIT_F32)

ALIGN = 64  # floats; every parameter starts on a 256-byte boundary of the flat buffer


def _ceil(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def effective_window(H: int, window, shift: bool):
    """(window, shift) used for a grid of height H (tulip.py:284-287 backup window, :219-222)."""
    wh, ww = int(window[0]), int(window[1])
    L = wh * ww
    if H < wh:
        return (1, L), ((0, L // 2) if shift else (0, 0))
    return (wh, ww), ((wh // 2, ww // 2) if shift else (0, 0))


@dataclass
class BlockSpec:
    prefix: str
    H: int
    W: int
    C: int
    nh: int
    shift: bool
    rate: float
    win: Tuple[int, int] = (2, 8)
    sft: Tuple[int, int] = (0, 0)
    slot: int = -1  # row in the DropPath scale table (-1: rate 0)


class FlatParams:
    """Flat fp32 master + bf16 shadow + address book, ordered by backward completion."""

    def __init__(self, model, device):
        self.device = device
        named = dict(model.named_parameters())
        order, marks = self._completion_order(model, named)
        assert sorted(order) == sorted(named.keys())
        self.names: List[str] = order
        self.offset: Dict[str, int] = {}
        self.numel: Dict[str, int] = {}
        self.shape: Dict[str, Tuple[int, ...]] = {}
        off = 0
        for n in order:
            p = named[n]
            self.offset[n], self.numel[n], self.shape[n] = off, p.numel(), tuple(p.shape)
            off = _ceil(off + p.numel(), ALIGN)
        self.total = off
        # backward-completion groups: (tag, end offset) -- run_backward fires bucket_hook(tag) once every
        # gradient in [previous end, end) has been launched
        self.groups: List[Tuple[str, int]] = [
            (tag, self.offset[order[cnt]] if cnt < len(order) else self.total) for tag, cnt in marks]
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(self.total, dtype=torch.bfloat16, device=device)
        mask = torch.zeros(self.total // ALIGN, dtype=torch.uint8)
        for n in order:
            p = named[n]
            self.flat[self.offset[n]:self.offset[n] + p.numel()].copy_(p.data.reshape(-1).float())
            if p.ndim > 1:  # timm param groups (main_lidar_upsampling.py:282): decay only ndim > 1
                mask[self.offset[n] // ALIGN:_ceil(self.offset[n] + p.numel(), ALIGN) // ALIGN] = 1
        self.decay_mask = mask.to(device)
        self.views: Dict[str, torch.Tensor] = {}
        for n in order:
            v = self.flat[self.offset[n]:self.offset[n] + self.numel[n]].view(self.shape[n])
            self.views[n] = v
            named[n].data = v
        self.base32 = self.flat.data_ptr()
        self.base16 = self.shadow.data_ptr()
        self.fp32_read = set()
        self.shadow_dirty = True
        # the bf16 shadow is current but the fragment-major copies of the fused wide blocks are not: a Trainer rewrites them
        # at the START of its next step, beside the head of the forward (Trainer.pack_at_step_start); every other consumer of
        # the copies (run_forward outside a Trainer step, GraphedForward) refreshes them first when this is set
        self.pack_dirty = False
        # Trainer(exchange="sharded") on more than one rank: the fp32 master is current only on the shards this rank steps until
        # Trainer.gather_state() has run -- refresh_shadow() refuses to cast from it meanwhile
        self.master_partial = False
        # bumped whenever a plan activates the copies of another block width (TulipEngine.plan): a Trainer whose captured step
        # rewrites the copies re-captures (its graph holds the pack launch of the widths active at capture time)
        self.pack_epoch = 0

    @staticmethod
    def _completion_order(model, named) -> List[str]:
        nl = model.num_layers
        order: List[str] = []
        marks: List[Tuple[str, int]] = []

        def add(prefix):
            ks = [k for k in named if k.startswith(prefix)]
            order.extend(k for k in ks if k not in order)

        def blocks_rev(prefix, depth):
            for b in reversed(range(depth)):
                for sub in ("mlp.fc2", "mlp.fc1", "norm2", "attn.proj", "attn.relative_position_bias_table",
                            "attn.qkv", "norm1"):
                    add(f"{prefix}.blocks.{b}.{sub}")

        add("decoder_pred."); add("ps_head."); add("final_patch_expanding."); add("norm_up.")
        marks.append(("head", len(order)))
        for i in reversed(range(nl - 1)):
            add(f"layers_up.{i}.upsample.")
            blocks_rev(f"layers_up.{i}", model.depths[nl - i - 2])
            marks.append((f"dec{i}", len(order)))
        add("first_patch_expanding.")
        for s in reversed(range(nl)):
            blocks_rev(f"layers.{s}", model.depths[s])
            if s < nl - 1:
                # the skip Linear of level s: its gradient is complete with decoder stage s, but its weight is read
                # once more when the backward reaches encoder stage s (x_save half of its input gradient, see
                # run_backward) -- it is grouped where it is last READ, so that a group may be handed to the
                # optimizer as soon as the group's hook has fired
                add(f"skip_connection_layers.{nl - s - 2}.")
            if s > 0:
                add(f"layers.{s - 1}.downsample.")
            marks.append((f"enc{s}", len(order)))
        add("patch_embed.")
        marks.append(("embed", len(order)))
        return order, marks

    def p32(self, name: str) -> int:
        # (every launch argument that reads a parameter as FP32 from the master buffer passes through here: the sharded exchange
        # plan keeps exactly these tensors replicated, Trainer(exchange="sharded"))
        self.fp32_read.add(name)
        return self.base32 + 4 * self.offset[name]

    def p16(self, name: str) -> int:
        return self.base16 + 2 * self.offset[name]

    def still_bound(self, model) -> bool:
        return self.current_params(model) is not None

    def current_params(self, model):
        """The module's parameters in flat-buffer order, or None if one of them no longer lives in the flat buffer (replaced, or
        its storage re-created).  Called once per module forward (the reference's calling convention), so it reads the owning
        sub-modules' `_parameters` dicts directly: three `named_parameters()` traversals per call were ~0.45 ms of host time
        in front of a 0.8-ms forward (tools/refloop_timeline.py)."""
        slots = getattr(self, "_slots", None)
        if slots is None or self._slots_model is not model:
            mods = dict(model.named_modules())
            slots, edges, seen = [], [], set()
            for n in self.names:
                mn, _, pn = n.rpartition(".")
                slots.append((mods[mn]._parameters, pn, self.base32 + 4 * self.offset[n]))
                # ... and every parent -> child edge on the way to the owning module: `model.x = other_module` leaves the OLD
                # module's parameter dict intact, so the dicts alone would not notice the replacement
                while mn and mn not in seen:
                    seen.add(mn)
                    par, _, child = mn.rpartition(".")
                    edges.append((mods[par]._modules, child, mods[mn]))
                    mn = par
            self._slots, self._slot_edges, self._slots_model = slots, edges, model
        for d, k, m in self._slot_edges:
            if d.get(k) is not m:
                return None
        out = []
        for d, k, ptr in slots:
            p = d.get(k)
            if p is None or p.data_ptr() != ptr:
                return None
            out.append(p)
        return out

    def refresh_shadow(self):
        if self.master_partial:
            raise RuntimeError("tulip_amd: the fp32 master weights are partial (Trainer(exchange='sharded') stepped only this rank's "
                               "shards); call Trainer.gather_state() on EVERY rank (it is a collective) before anything re-derives "
                               "the bf16 weights from them -- model(x), load_state_dict, GraphedForward.weights_changed")
        ops.cast_flat(self.flat, self.shadow, self.total)
        self.refresh_transposes()
        self.shadow_dirty = False

    def make_packed(self, names_by_width: Dict[int, List[str]], with_transposes: bool):
        """Fragment-major bf16 copies of the given 2-D weights, and of their transposes (what the fused wide-block kernels
        stream, csrc/swinw.hip), grouped by block width; refresh_transposes() rewrites the copies of the widths some plan
        actually runs fused (`pk_active`) from the shadow."""
        self.pk_offset: Dict[str, int] = {}
        self.pk_width: Dict[str, int] = {}
        off = 0
        for width, names in names_by_width.items():
            for n in names:
                self.pk_offset[n] = off
                self.pk_width[n] = width
                off = _ceil(off + self.numel[n], ALIGN)
        self.packed = torch.zeros(max(off, ALIGN), dtype=torch.bfloat16, device=self.device)
        self.packed_t = torch.zeros(max(off, ALIGN) if with_transposes else ALIGN, dtype=torch.bfloat16, device=self.device)
        self._pk_entries, self._pk_joined = {}, {}
        # names whose copies are rewritten AFTER the end-of-step AdamW launch instead of with the others (Trainer pack_at_end: the
        # weights that launch steps -- the skip Linears -- are not current any earlier); refresh_transposes(late=...)
        self.pk_late = frozenset()

        def part_of(width, n):
            # part 0: the encoder's blocks and PatchMerging reductions (read first in a forward), part 1: the decoder's, part 2: the
            # deep stages' (C >= 768: 14 / 57 MB of weights per block, first read three stages into the forward), see refresh_transposes
            if width >= 768:
                return 2
            return int(n.startswith(("layers_up.", "skip_connection_layers.", "first_patch_expanding.")))
        for width, names in names_by_width.items():
            for part in (0, 1, 2):
                sel = [n for n in names if part_of(width, n) == part]
                ent = [(n, (self.p16(n), self.packed.data_ptr() + 2 * self.pk_offset[n], self.shape[n][0], self.shape[n][1], 0))
                       for n in sel]
                if with_transposes:
                    ent += [(n, (self.p16(n), self.packed_t.data_ptr() + 2 * self.pk_offset[n], self.shape[n][0], self.shape[n][1], 1))
                            for n in sel]
                self._pk_entries[(width, part)] = ent
        self.pk_active = set()

    def p16p(self, name: str) -> int:
        return self.packed.data_ptr() + 2 * self.pk_offset[name]

    def p16t(self, name: str) -> int:
        return self.packed_t.data_ptr() + 2 * self.pk_offset[name]

    def refresh_transposes(self, part: Optional[int] = None, also: Tuple[int, ...] = (), late: Optional[bool] = None):
        """One launch for the copies of every active width (up to TULIP_PACK_MAX matrices per launch).  part = 0 / 1 / 2: only the
        encoder's / the decoder's wide blocks / the deep stages' blocks (the Trainer's forward rewrites the pieces at different
        points, _issue_pack).  late: None = every name, False = all but `pk_late`, True = only `pk_late`."""
        widths = tuple(sorted(getattr(self, "pk_active", ())))
        if not widths:
            return
        key = (widths, part, also, late, self.pk_late if late is not None else None)
        if key not in self._pk_joined:
            parts = (0, 1, 2) if part is None else (part,) + tuple(also)
            self._pk_joined[key] = ops.pack_items([e for width in widths for q in parts for n, e in self._pk_entries[(width, q)]
                                                   if late is None or (n in self.pk_late) == late])
        items, n = self._pk_joined[key]
        if n:
            ops.pack_bf16_multi(items, n)
        if (part is None or part == 1) and late is not False:
            self.pack_dirty = False

    def packed_names(self) -> List[str]:
        """names whose fragment-major copies are being maintained (an active width)"""
        return [n for n, w in self.pk_width.items() if w in self.pk_active]


class Plan:
    """Static workspaces for one batch size."""

    def __init__(self, eng: "TulipEngine", B: int):
        self.B = B
        self.bufs: Dict[str, torch.Tensor] = {}
        self.generation = 0
        dev = eng.device
        m = eng.model
        E, nl = m.embed_dim, m.num_layers
        H0, W0 = eng.grid
        self.f32 = lambda name, *shape: self._alloc(name, shape, torch.float32, dev)
        self.b16 = lambda name, *shape: self._alloc(name, shape, torch.bfloat16, dev)
        Hh, Wh = m.target_img_size
        self.x_in = self.f32("x_in", B, m.in_chans, m.img_size[0], m.img_size[1])
        self.target = self.f32("target", B, m.in_chans, Hh, Wh)
        self.pred = self.f32("pred", B, m.in_chans, Hh, Wh)
        self.dpred = self.f32("dpred", B, m.in_chans, Hh, Wh)
        self.losses = self.f32("losses", 2)
        self.partials = self.f32("partials", max(2048, 2 * ((B * H0 * W0 + 31) // 32)))   # loss partial sums (2 per block)
        self.gscale = self.f32("gscale", 1)
        self.gscale.fill_(1.0)
        nslots = max(1, eng.n_drop_slots)
        self.drop_scale = self.f32("drop_scale", nslots, B)
        self.drop_scale.fill_(1.0)
        self.drop_u = self.f32("drop_u", nslots, B)
        maxM = B * H0 * W0
        # per-level tensors
        for s in range(nl):
            Hs, Ws, Cs = H0 >> s, W0 >> s, E << s
            Ms = B * Hs * Ws
            self.f32(f"enc{s}.in", Ms, Cs)
            self.f32(f"enc{s}.dx", Ms, Cs)
            if s < nl - 1:
                self.b16(f"enc{s}.xm", Ms // 4, 4 * Cs)
                self.f32(f"enc{s}.mmean", Ms // 4)
                self.f32(f"enc{s}.mrstd", Ms // 4)
                self.f32(f"dec{s}.in", Ms, Cs)     # skip-linear output = decoder stage input
                self.f32(f"dec{s}.dx", Ms, Cs)
                self.b16(f"dec{s}.cat", Ms, 2 * Cs)
                self.b16(f"dec{s}.dyskip", Ms, Cs)
            self.b16(f"lvl{s}.xb", Ms, Cs)         # bf16 copy of the stage output feeding a PatchUnmerging expand
            if s > 0 and not m.patch_unmerging:    # PatchExpanding (tulip.py:126-140): Linear output, statistics of the
                self.f32(f"lvl{s}.ey", Ms, 2 * Cs)                              # 4 fine tokens per row, and the
                self.f32(f"lvl{s}.emean", 4 * Ms); self.f32(f"lvl{s}.erstd", 4 * Ms)   # upstream gradient in fine order
                self.b16(f"lvl{s}.dfine", 4 * Ms, Cs // 2)
        for spec in eng.blocks:
            M, C = B * spec.H * spec.W, spec.C
            Hd = eng.hidden(C)
            p = spec.prefix
            self.b16(p + ".xn1", M, C); self.f32(p + ".mean1", M); self.f32(p + ".rstd1", M)
            self.b16(p + ".qkv", M, 3 * C); self.b16(p + ".o", M, C); self.f32(p + ".x1", M, C)
            self.b16(p + ".xn2", M, C); self.f32(p + ".mean2", M); self.f32(p + ".rstd2", M)
            self.b16(p + ".h", M, Hd); self.b16(p + ".g", M, Hd); self.f32(p + ".out", M, C)
        # gradient temporaries that feed a weight-gradient GEMM on the side stream are per block / per use
        # (never recycled inside a step), so the side stream needs no WAR synchronisation with the main chain
        for spec in eng.blocks:
            M, C = B * spec.H * spec.W, spec.C
            p = spec.prefix
            self.b16(p + ".dyb_m", M, C); self.b16(p + ".dyb_a", M, C)
            self.b16(p + ".dh", M, eng.hidden(C)); self.b16(p + ".dqkv", M, 3 * C)
        for s in range(nl):
            Ms, Cs = B * (H0 >> s) * (W0 >> s), E << s
            self.b16(f"lvl{s}.dz2", Ms, 2 * Cs)      # unshuffled grad entering the level-s PatchUnmerging
            if s > 0:
                self.b16(f"enc{s}.dyb", Ms, Cs)      # bf16 grad w.r.t. the PatchMerging output of level s-1
        Cmax_tok = max(B * sp.H * sp.W * sp.C for sp in eng.blocks)
        Hdmax_tok = max(B * sp.H * sp.W * eng.hidden(sp.C) for sp in eng.blocks)
        self.b16("t.dxn", Cmax_tok); self.b16("t.do", Cmax_tok)   # consumed on the main stream only
        big = max(4 * (B * (H0 >> s) * (W0 >> s) // 4) * (E << s) for s in range(nl))
        self.b16("t.dxm", big)                     # dgrad of a merge reduction [M/4][4C]
        self.b16("tail.xn", maxM, E); self.f32("tail.mean", maxM); self.f32("tail.rstd", maxM)
        self.b16("tail.dz", maxM, 16 * E)
        if not m.pixel_shuffle:                    # FinalPatchExpanding (tulip.py:144-159)
            r2 = m.upscale_factor ** 2
            self.f32("tail.ey", maxM, r2 * E); self.f32("tail.emean", r2 * maxM); self.f32("tail.erstd", r2 * maxM)
        self.b16("tail.dxn", maxM, E)
        # partial-row workspaces of the atomic-free reductions
        self.f32("tail.dwd_part", (maxM + 31) // 32, 128)
        W_ = eng.params
        o0 = W_.offset["patch_embed.proj.weight"]
        last = "patch_embed.norm.bias" if "patch_embed.norm.bias" in W_.offset else "patch_embed.proj.bias"
        self.embed_stride = _ceil(W_.offset[last] + W_.numel[last], ALIGN) - o0
        ep = torch.zeros(ops.patch_embed_bwd_blocks(maxM), self.embed_stride, dtype=torch.float32, device=dev)
        self.bufs["embed_part"] = ep                                          # padding columns stay zero forever
        self.embed_part_ptr = ep.data_ptr()

    def _alloc(self, name, shape, dtype, dev):
        t = torch.empty(*shape, dtype=dtype, device=dev)
        self.bufs[name] = t
        return t

    def __getitem__(self, k):
        return self.bufs[k]

    def scratch(self, name: str, numel: int) -> int:
        """Address of a lazily allocated fp32 scratch buffer that is private to one call site (so work queued
        on the side stream never races with the next user).  First use happens in the eager warm-up pass,
        never inside graph capture."""
        t = self.bufs.get(name)
        if t is None or t.numel() < numel:
            t = torch.zeros(numel, dtype=torch.float32, device=self.x_in.device)
            self.bufs[name] = t
        return t.data_ptr()

    def reset_exchange(self):
        """Zero the pair-exchange areas (partial sums + arrival tickets of the two-workgroups-per-window launches, csrc/swinw.hip):
        the kernels re-zero a ticket when its second arriver passes, so only a launch sequence that was cut short (a failed
        capture, a faulted step) can leave one non-zero -- whoever catches that calls this before the next launch."""
        for k, t in self.bufs.items():
            if k.startswith("xchg."):
                t.zero_()

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


class TulipEngine:
    def __init__(self, model):
        self.model = model
        self.device = None
        # A/B switches, carried per launch (the library keeps no state): the 64 x 96 weight-gradient tile everywhere; no L2 warm-up
        self.wgrad_small_tiles = knobs.is_zero("TULIP_WGRAD_TILES")
        self.no_warm = knobs.is_zero("TULIP_SWINW_WARM")            # (tools/cold_probe.py)
        self.params: Optional[FlatParams] = None
        self.plans: Dict[int, Plan] = {}
        m = model
        if m.in_chans != 1:
            raise NotImplementedError("tulip_amd fused head supports in_chans == 1 (range images)")
        if m.upscale_factor != 4:
            raise NotImplementedError("tulip_amd fused head supports upscale_factor == 4 (every BASELINE config)")
        ws = m.window_size if isinstance(m.window_size, (tuple, list)) else (m.window_size, m.window_size)
        self.window = (int(ws[0]), int(ws[1]))
        if self.window[0] * self.window[1] != 16:
            raise NotImplementedError("tulip_amd attention kernel supports 16-token windows (window_size 2 8)")
        ph, pw = m.patch_size
        if m.img_size[0] % ph or m.img_size[1] % pw:
            raise NotImplementedError("img_size must be divisible by patch_size")
        self.grid = (m.img_size[0] // ph, m.img_size[1] // pw)
        self.eps = m._ln_eps
        nl = m.num_layers
        self.blocks: List[BlockSpec] = []
        self.enc_blocks: List[List[BlockSpec]] = []
        self.dec_blocks: List[List[BlockSpec]] = []
        slot = 0

        def mk(prefix, s, b, rate):
            nonlocal slot
            H, W, C = self.grid[0] >> s, self.grid[1] >> s, m.embed_dim << s
            win, sft = effective_window(H, self.window, b % 2 == 1)
            if H % win[0] or W % win[1]:
                raise NotImplementedError(f"token grid {H}x{W} not divisible by window {win}")
            sp = BlockSpec(prefix, H, W, C, m.num_heads[s], b % 2 == 1, rate, win, sft)
            if rate > 0.0:
                sp.slot = slot
                slot += 2
            self.blocks.append(sp)
            return sp

        for s in range(nl):
            mods = m.layers[s].blocks
            self.enc_blocks.append([mk(f"layers.{s}.blocks.{b}", s, b, mods[b].drop_path_rate)
                                    for b in range(m.depths[s])])
        for i in range(nl - 1):
            s = nl - i - 2
            mods = m.layers_up[i].blocks
            self.dec_blocks.append([mk(f"layers_up.{i}.blocks.{b}", s, b, mods[b].drop_path_rate)
                                    for b in range(m.depths[s])])
        self.n_drop_slots = slot
        self._keep = None
        self._pending = []        # weight-gradient launches queued for the side stream
        self._side_dirty = False
        self._graphs = {}
        self._rel32 = None

    # ------------------------------------------------------------------ parameters
    def hidden(self, C: int) -> int:
        return int(C * self.model.mlp_ratio)

    def invalidate(self):
        self.params = None
        self.plans.clear()
        self._graphs.clear()

    def bind(self, device):
        """(Re-)flatten the parameters on `device` if the module's storage moved."""
        self._cur_params = None
        if self.params is not None and self.device == device:
            self._cur_params = self.params.current_params(self.model)
            if self._cur_params is not None:
                return
        self.device = device
        self.plans.clear()
        self._graphs.clear()
        self.params = FlatParams(self.model, device)
        by_width: Dict[int, List[str]] = {}
        if self.fuse_wide or self.fuse_wide_bwd or self.fuse_deep:
            for sp in self.blocks:
                if self._fusable_wide(sp) or self._fusable_deep(sp):
                    by_width.setdefault(sp.C, []).extend(sp.prefix + suffix for suffix in (
                        ".attn.qkv.weight", ".attn.proj.weight", ".mlp.fc1.weight", ".mlp.fc2.weight"))
        if self.fuse_glue and self.model.patch_unmerging:
            # the stage boundaries as one launch each (csrc/glue.hip) stream fragment-major copies too: the PatchMerging reductions,
            # the PatchUnmerging expand convs and the skip Linears, under the pseudo-width 0
            m, nl = self.model, self.model.num_layers
            glue = [f"layers.{s}.downsample.reduction.weight" for s in range(nl - 1)]
            glue += [f"skip_connection_layers.{i}.weight" for i in range(nl - 1)]
            glue += [f"layers_up.{i}.upsample.expand.weight" for i in range(nl - 2)]
            if nl > 1:
                glue.append("first_patch_expanding.expand.weight")
            # ... of the boundaries some fused form exists for (csrc/glue.hip: widths 96 / 192 / 384; tulip_large's two deepest
            # boundaries keep their GEMM launches and get no copies)
            def ok(n):
                r, c = self.params.shape[n][0], self.params.shape[n][1]
                if n.endswith("reduction.weight"):      # [2 Cin][4 Cin]: merge_fwd Cin <= 384
                    return c // 4 in (96, 192, 384)
                if n.startswith("skip_connection_layers."):     # [Cs][2 Cs]: merge_bwd (Cs = 192, 384), unmerge (Cs = 96, 192)
                    return r in (96, 192, 384)
                return c in (192, 384, 768)             # expand [2C][C]: the fused unmerge (C = 192, 384), the small-K GEMM form (C <= 768)
            by_width[0] = [n for n in glue if n in self.params.offset and ok(n)]
        self.params.make_packed(by_width, with_transposes=self.fuse_wide_bwd or self.fuse_deep or bool(by_width.get(0)))
        # the wide widths (a few MB of copies) keep their fragment-major copies fresh from the start; the DEEP widths (57 MB read +
        # 114 MB written per refresh for tulip_base) only once a plan runs them fused (plan(): a launch with 32-64 windows) -- at
        # batch 64 or under the N > 1 step nothing streams those copies.  A Trainer's captured step bakes in the
        # pack launch of the widths active at capture time: a later activation bumps params.pack_epoch and the Trainer re-captures
        self.params.pk_active = {w for w in by_width if w < 768 and by_width[w]}
        rel = self.model.layers[0].blocks[0].attn.relative_position_index
        self._rel32 = rel.to(device=device, dtype=torch.int32).contiguous()
        rates = torch.ones(max(1, self.n_drop_slots), 1)
        for sp in self.blocks:
            if sp.slot >= 0:
                rates[sp.slot] = rates[sp.slot + 1] = 1.0 - sp.rate
        self._keep = rates.to(device)
        # backward order ends with encoder stage 0, block 1 then block 0: the very last block's MLP-half weight gradients start as
        # soon as their operands exist (nothing is left to hide them behind)
        self.early_flush = frozenset(sp.prefix for sp in self.enc_blocks[0][:1])
        self._drop_seed = torch.initial_seed()      # torch.manual_seed() governs the DropPath stream
        self._drop_counter = torch.zeros(1, dtype=torch.int64, device=device)

    def plan(self, B: int) -> Plan:
        if B not in self.plans:
            self.plans[B] = Plan(self, B)
            # block widths this batch size runs fused: their weight copies are maintained from now on
            widths = {sp.C for sp in self.blocks if ((self.fuse_wide or self.fuse_wide_bwd) and self._fusable_wide(sp, B))
                      or self._fusable_deep(sp, B)}
            if not widths <= self.params.pk_active:
                self.params.pk_active |= widths
                self.params.pack_epoch += 1
                self._pack_mark_tag_ = None
                if not self.params.shadow_dirty:
                    self.params.refresh_transposes()
        if getattr(self, "_ws", None) is None or self._ws.device != self.device:
            self._ws = torch.empty(self.WS_ELEMS + (1 << 20), dtype=torch.float32, device=self.device)
            self._ws_ptr = self._ws.data_ptr()
            # weight-gradient branch: own stream + own slab workspace
            self._ws_side = torch.empty(self.WS_ELEMS + (1 << 20), dtype=torch.float32, device=self.device)
            self._ws_side_ptr = self._ws_side.data_ptr()
            self._side_stream = torch.cuda.Stream(device=self.device)
        return self.plans[B]

    # ------------------------------------------------------------------ forward
    def draw_drop_scales(self, P: Plan, train: bool, drop_u: Optional[torch.Tensor] = None):
        """DropPath multipliers floor(keep+u)/keep per (block branch, sample) (tulip.py:25-29)."""
        self._pending_draw = None
        if not train or self.n_drop_slots == 0:
            P.drop_scale.fill_(1.0)
            return
        if drop_u is None:
            # the step counter lives on the device, so a graph replay draws fresh numbers.  The draw rides in the forward's first
            # launch (tulip_patch_embed_fwd_draw: run_forward picks it up) instead of being a launch of its own at the head of the chain
            self._pending_draw = (P, (self._keep, P.drop_scale, P.drop_u, self.n_drop_slots, P.B, self._drop_seed, self._drop_counter))
            return
        P.drop_u.copy_(drop_u)                     # injected draws (parity tests against the oracle)
        torch.floor(self._keep + P.drop_u, out=P.drop_scale)
        P.drop_scale.div_(self._keep)

    _pending_draw = None

    def _ds(self, P: Plan, sp: BlockSpec, branch: int):
        if sp.slot < 0:
            return None
        return P.drop_scale.data_ptr() + 4 * (sp.slot + branch) * P.B

    infer_no_save = True   # fused blocks' inference form in with_loss=False forwards
    _no_save = False
    fuse_block96 = knobs.on("TULIP_FUSE_BLOCK96", True)
    fuse_block96_bwd = knobs.on("TULIP_FUSE_BLOCK96_BWD", True)

    # round 4: the C = 96 forward no longer writes qkv and the fc1 pre-activation (1344 of 3472 B per token); the fused backward
    # recomputes both from x / x1 with the weights it holds in LDS anyway (csrc/swin96.hip swin96_bwd_kernel<true>).
    # TULIP_SWIN96_RECOMPUTE=0: the round-3 form (everything saved and read back)
    # Measured (profiles/README.md round 4): the forward gains 1 us at batch 8 / 25 us at batch 64 (it is not store-bound), the
    # backward loses 12 / 45 us (126 more MFMAs per window, bank conflicts of the plain reads at the transpose pitch): off by
    # default, kept as a tested form (bit-identical outputs, tests/test_round4_gpu.py).
    recompute96 = knobs.on("TULIP_SWIN96_RECOMPUTE", False)
    # round 4: the fused C = 96 forward writes bf16(gelu'(h)) where it used to write h (same bytes); the only thing the backward
    # does with h is that derivative (100 of 134 vector instructions per 32 hidden channels of its MLP loop).
    # TULIP_FC1_GRAD=0: h is saved and the backward evaluates gelu' itself
    fc1_grad96 = knobs.on("TULIP_FC1_GRAD", True)

    def _hgrad96(self, sp: BlockSpec) -> bool:
        return (self.fc1_grad96 and self.fuse_block96 and self.fuse_block96_bwd and self._fusable96(sp)
                and not self._recomp96(sp))

    # ... and the same hand-off in the fused wide blocks (csrc/swinw.hip): TULIP_FC1_GRAD_WIDE=0 switches it off there
    fc1_grad_wide = knobs.on("TULIP_FC1_GRAD_WIDE", True)

    def _hgrad_wide(self, sp: BlockSpec, B: int) -> bool:
        return self.fc1_grad_wide and self.fuse_wide and self.fuse_wide_bwd and self._fusable_wide(sp, B)

    def _recomp96(self, sp: BlockSpec) -> bool:
        return self.recompute96 and self.fuse_block96 and self.fuse_block96_bwd and self._fusable96(sp)

    def _fusable96(self, sp: BlockSpec) -> bool:
        """csrc/swin96.hip covers the embed-width-96 block: 3 heads x 32, window 2x8, MLP 96 -> 384 -> 96."""
        return (sp.C == 96 and sp.nh == 3 and self.hidden(sp.C) == 384 and tuple(sp.win) == (2, 8) and sp.W % 64 == 0
                and sp.H % 2 == 0)

    # BASELINE configs[4] "fp8 MFMA attention": the attention scores Q.K^T of every block from e4m3 operands
    # (v_mfma_f32_16x16x32_fp8_fp8; q, k rounded from their bf16 values, round to nearest even; softmax, P.V and
    # everything else unchanged); the backward differentiates exactly that function (it multiplies dS with the rounded
    # q, k).  Off by default: the reference computes the scores from bf16 / fp16 operands.
    attn_fp8 = knobs.is_one("TULIP_ATTN_FP8")

    def _mask_arg(self, sp: BlockSpec, B: Optional[int] = None) -> int:
        """`masked` argument of the attention / block kernels: bit 0 shifted-window mask, bit 1 fp8 scores, bit 2 (fused
        blocks, TULIP_BLOCK_FC1_GRAD; B = the plan's batch size for the wide ones) the fc1_pre buffer carries gelu'(h) from
        the forward to the backward."""
        hg = self._hgrad96(sp) or (B is not None and (self._hgrad_wide(sp, B) or self._fusable_deep(sp, B)))
        return int(bool(sp.shift)) | (2 if self.attn_fp8 else 0) | (4 if hg else 0) | (8 if self.no_warm else 0)

    # two workgroups per window for the C = 384 blocks where one workgroup owns one window (csrc/swinw.hip, SPLIT)
    split_wide = knobs.on("TULIP_SWINW_SPLIT", True)
    # the backward's split form is 5.6 us faster per launch in isolation (50.9 -> 45.3 us) and 34 us SLOWER per step: the half of the
    # chip its 128 workgroups leave free is where the side queue's weight gradients run (profiles/README.md); off
    split_wide_bwd = knobs.on("TULIP_SWINW_SPLIT_BWD", False)
    fuse_wide = knobs.on("TULIP_FUSE_WIDE", True)
    fuse_wide_bwd = knobs.on("TULIP_FUSE_WIDE_BWD", True)
    # C = 192 always; C = 384 (stage 2: 3.5 MB of weights per block) from 128 windows per launch up (batch 8 at the KITTI
    # size).  Every workgroup streams the block's whole weight set through its own CU: with two windows per workgroup the
    # 128 windows of batch 8 were 64 workgroups and the fused form only tied the 7-kernel sequences (66 + 65 us against
    # 63 + 91 us isolated); with ONE window per workgroup below 256 windows (csrc/swinw.hip wide_g) it is 50 + 51 us and
    # 1.7 % of the step; at batch 16 / 32 / 64 it wins 3 / 5 / 4.5 %.  TULIP_FUSE_WIDE_MIN_WINDOWS overrides.
    wide_widths = (192, 384)
    wide_min_windows = knobs.integer("TULIP_FUSE_WIDE_MIN_WINDOWS", 128)

    def _fusable_wide(self, sp: BlockSpec, B: Optional[int] = None) -> bool:
        """csrc/swinw.hip covers C = 192 / 384: heads of 32, window 2x8, MLP C -> 4C -> C.  B = None: could the block ever
        be fused (weight copies are kept for it); with a batch size: is it fused in that plan."""
        ok = (sp.C in self.wide_widths and sp.C in (192, 384) and sp.nh * 32 == sp.C and self.hidden(sp.C) == 4 * sp.C
              and tuple(sp.win) == (2, 8) and sp.W % 16 == 0 and sp.H % 2 == 0)
        if ok and B is not None and sp.C == 384:
            ok = B * (sp.H // 2) * (sp.W // 8) >= self.wide_min_windows
        return ok

    # round 5: the deep stages (C = 768 / 1536: stage 3, and stage 4 of tulip_large) as four sliced launches per block and direction
    # (csrc/swind.hip; + the two LayerNorm backward launches) instead of the 15 / 16-launch sequences.  TULIP_FUSE_DEEP=0: the sequences.
    # The sliced form puts 8 workgroups on each group of windows, so it wants 32-64 windows per launch (256-512 workgroups); isolated,
    # forward / backward against the sequence (profiles/r5_bench_deep_shapes.txt): C = 768 with 32 windows 0.66 / 0.81, 64: 0.72 / 0.82,
    # 128: 1.10 / 1.05, 256 (batch 64): 1.32 / 1.18; C = 1536 with 16 windows (half the chip) 1.0 / 1.14, 32: 0.77 / 0.77 -- outside
    # [TULIP_FUSE_DEEP_MIN_WINDOWS, TULIP_FUSE_DEEP_MAX_WINDOWS] the GEMM sequence runs (its tiles fill the chip there)
    fuse_deep = knobs.on("TULIP_FUSE_DEEP", True)
    deep_min_windows = knobs.integer("TULIP_FUSE_DEEP_MIN_WINDOWS", 32)
    deep_max_windows = knobs.integer("TULIP_FUSE_DEEP_MAX_WINDOWS", 64)

    def _fusable_deep(self, sp: BlockSpec, B: Optional[int] = None) -> bool:
        ok = (self.fuse_deep and sp.C in (768, 1536) and sp.nh * 32 == sp.C and self.hidden(sp.C) == 4 * sp.C
              and tuple(sp.win) in ((2, 8), (1, 16)) and sp.H % sp.win[0] == 0 and sp.W % sp.win[1] == 0)
        if ok and B is not None:
            ok = self.deep_min_windows <= B * (sp.H // sp.win[0]) * (sp.W // sp.win[1]) <= self.deep_max_windows
        return ok

    def _fused_bwd(self, sp: BlockSpec, B: int) -> bool:
        return ((self.fuse_block96_bwd and self._fusable96(sp)) or (self.fuse_wide_bwd and self._fusable_wide(sp, B))
                or self._fusable_deep(sp, B))

    fuse_splitk_ln = True
    fuse_tail_bwd = True      # head backward without the d(expand) tensor
    fuse_tail_ln_bwd = True   # ... and norm_up's backward in its epilogue
    _tail_fused = False

    def _unfused(self, sp: BlockSpec, B: int) -> bool:
        return not ((self.fuse_wide and self._fusable_wide(sp, B)) or (self.fuse_block96 and self._fusable96(sp))
                    or self._fusable_deep(sp, B))

    def _gemm_resid_ln(self, A, Wt, M, N, K, *, lda, ldb, bias, out, aux, rowscale, tok, ln, out2=None):
        """Linear + DropPath residual (EPI_RESID_F32) followed by a LayerNorm of its output rows, ln = (gamma, beta, xn, mean,
        rstd).  Where the GEMM is split along K anyway (the M = 512..2048 GEMMs of the deep stages, see _gemm) its fold
        launch also normalises the rows it holds (tulip_splitk_resid_ln); otherwise GEMM and LayerNorm are two launches."""
        gamma, beta, xn, mean, rstd = ln
        gn = (N + 95) // 96
        blocks = ((M + 63) // 64) * gn if ((M + 127) // 128) * gn < 256 else ((M + 127) // 128) * gn
        s = 1
        if self.fuse_splitk_ln and blocks <= 96 and K >= 768 and ops.splitk_resid_ln_supported(N):
            s = ops.gemm_effective_splits(K, max(1, min(256 // blocks, K // 256, (self.WS_ELEMS * 4) // (M * N * 4))))
        if s > 1:
            ops.gemm(A, Wt, M, N, K, lda=lda, ldb=ldb, epi=EPI_SPLIT_F32, out=self._ws_ptr, ldo=N, splits=s)
            ops.splitk_resid_ln(self._ws_ptr, s, M, N, bias, aux, N, rowscale, tok, out, N, out2, N if out2 is not None else 0,
                                gamma, beta, xn, mean, rstd, self.eps)
            return
        self._gemm(A, Wt, M, N, K, lda=lda, ldb=ldb, epi=EPI_RESID_F32, bias=bias, out=out, aux=aux, ldaux=N,
                   rowscale=rowscale, rows_per_sample=tok, out2=out2, ldo2=N if out2 is not None else 0)
        ops.layernorm_fwd(out, gamma, beta, xn, mean, rstd, M, N, self.eps)

    def _block_fwd(self, P: Plan, sp: BlockSpec, xin, xout, out_bf16=None, ln1_done=False, next_sp=None):
        """out_bf16: the block output also leaves as a bf16 [M][C] copy (operand of a PatchUnmerging GEMM).
        ln1_done: the previous (unfused) block's fc2 launch already wrote this block's xn1 / mean1 / rstd1;
        next_sp: the next block of the stage if it runs unfused too -- its norm1 then rides in this block's fc2 fold."""
        W_ = self.params
        p = sp.prefix
        B, C, nh = P.B, sp.C, sp.nh
        M, Hd, tok = B * sp.H * sp.W, self.hidden(sp.C), sp.H * sp.W
        wide = self.fuse_wide and self._fusable_wide(sp, B)
        if self._fusable_deep(sp, B):
            # four sliced launches (csrc/swind.hip: by heads, by output channels, by hidden channels, by output channels); same
            # tensors as the sequence
            self._join_pack(p)
            keep = (".o", ".x1", ".g")              # pass from launch to launch: written in the inference form too
            sv = (lambda k: P[p + k] if k in keep else None) if self._no_save else (lambda k: P[p + k])
            wf = W_.p16p
            ops.swind_block_fwd(
                C, sp.win, out_bf16=out_bf16,
                x_in=xin, x1=sv(".x1"), x_out=xout, xn1=sv(".xn1"), qkv=sv(".qkv"), attn_out=sv(".o"),
                xn2=sv(".xn2"), fc1_pre=sv(".h"), fc1_act=sv(".g"), mean1=sv(".mean1"),
                rstd1=sv(".rstd1"), mean2=sv(".mean2"), rstd2=sv(".rstd2"),
                w_qkv=wf(p + ".attn.qkv.weight"), w_proj=wf(p + ".attn.proj.weight"),
                w_fc1=wf(p + ".mlp.fc1.weight"), w_fc2=wf(p + ".mlp.fc2.weight"),
                b_qkv=W_.p32(p + ".attn.qkv.bias"), b_proj=W_.p32(p + ".attn.proj.bias"),
                b_fc1=W_.p32(p + ".mlp.fc1.bias"), b_fc2=W_.p32(p + ".mlp.fc2.bias"),
                norm1_weight=W_.p32(p + ".norm1.weight"), norm1_bias=W_.p32(p + ".norm1.bias"),
                norm2_weight=W_.p32(p + ".norm2.weight"), norm2_bias=W_.p32(p + ".norm2.bias"),
                bias_table=W_.p32(p + ".attn.relative_position_bias_table"), rel_index=self._rel32,
                drop_scale_attn=self._ds(P, sp, 0), drop_scale_mlp=self._ds(P, sp, 1), B=B, H=sp.H, W=sp.W,
                shift_h=sp.sft[0], shift_w=sp.sft[1], masked=self._mask_arg(sp, B), eps=self.eps)
            return
        if wide or (self.fuse_block96 and self._fusable96(sp)):
            # the whole block in one launch (csrc/swin96.hip, csrc/swinw.hip); writes the same tensors as the sequence below
            if wide:
                self._join_pack(p)                 # the fragment-major weight copies being rewritten beside the forward
            xkw = {}
            if wide and self.split_wide and (self._no_save or self._hgrad_wide(sp, B)):
                # C = 384 with one window per workgroup: two workgroups per window (tulip_swinw_block_fwd_split)
                nb = ops.swinw_split_bytes(C, B, sp.H, sp.W)
                if nb:
                    P.scratch("xchg." + p, (nb + 3) // 4)
                    xkw["exchange"] = P.bufs["xchg." + p]
            launch = (lambda **kw: ops.swinw_block_fwd(C, out_bf16=out_bf16, **xkw, **kw)) if wide else ops.swin96_block_fwd
            wf = W_.p16p if wide else W_.p16           # the wide kernel streams fragment-major copies of the weights
            # forward without a backward behind it (run_forward(with_loss=False): eval / MC-dropout inference): the kernels'
            # inference form -- none of the activations a backward would read is written (90 % of the C = 96 kernel's traffic)
            lean = (not wide) and self._recomp96(sp)                # qkv / fc1 pre-activation: recomputed by the backward
            sv = (lambda k: None) if self._no_save else (lambda k: None if (lean and k in (".qkv", ".h")) else P[p + k])
            launch(
                x_in=xin, x1=sv(".x1"), x_out=xout, xn1=sv(".xn1"), qkv=sv(".qkv"), attn_out=sv(".o"),
                xn2=sv(".xn2"), fc1_pre=sv(".h"), fc1_act=sv(".g"), mean1=sv(".mean1"),
                rstd1=sv(".rstd1"), mean2=sv(".mean2"), rstd2=sv(".rstd2"),
                w_qkv=wf(p + ".attn.qkv.weight"), w_proj=wf(p + ".attn.proj.weight"),
                w_fc1=wf(p + ".mlp.fc1.weight"), w_fc2=wf(p + ".mlp.fc2.weight"),
                b_qkv=W_.p32(p + ".attn.qkv.bias"), b_proj=W_.p32(p + ".attn.proj.bias"),
                b_fc1=W_.p32(p + ".mlp.fc1.bias"), b_fc2=W_.p32(p + ".mlp.fc2.bias"),
                norm1_weight=W_.p32(p + ".norm1.weight"), norm1_bias=W_.p32(p + ".norm1.bias"),
                norm2_weight=W_.p32(p + ".norm2.weight"), norm2_bias=W_.p32(p + ".norm2.bias"),
                bias_table=W_.p32(p + ".attn.relative_position_bias_table"), rel_index=self._rel32,
                drop_scale_attn=self._ds(P, sp, 0), drop_scale_mlp=self._ds(P, sp, 1), B=B, H=sp.H, W=sp.W,
                shift_h=sp.sft[0], shift_w=sp.sft[1], masked=self._mask_arg(sp, B), eps=self.eps)
            if out_bf16 is not None and not wide:
                ops.cast_f32_bf16(xout, out_bf16, M, C)
            return
        if not ln1_done:
            ops.layernorm_fwd(xin, W_.p32(p + ".norm1.weight"), W_.p32(p + ".norm1.bias"), P[p + ".xn1"],
                              P[p + ".mean1"], P[p + ".rstd1"], M, C, self.eps)
        self._gemm(P[p + ".xn1"], W_.p16(p + ".attn.qkv.weight"), M, 3 * C, C, lda=C, ldb=C, epi=EPI_BF16,
                 bias=W_.p32(p + ".attn.qkv.bias"), out=P[p + ".qkv"])
        ops.window_attn_fwd(P[p + ".qkv"], W_.p32(p + ".attn.relative_position_bias_table"), self._rel32, P[p + ".o"],
                            B, sp.H, sp.W, C, nh, sp.win, sp.sft, self._mask_arg(sp))
        self._gemm_resid_ln(P[p + ".o"], W_.p16(p + ".attn.proj.weight"), M, C, C, lda=C, ldb=C,
                            bias=W_.p32(p + ".attn.proj.bias"), out=P[p + ".x1"], aux=xin, rowscale=self._ds(P, sp, 0), tok=tok,
                            ln=(W_.p32(p + ".norm2.weight"), W_.p32(p + ".norm2.bias"), P[p + ".xn2"], P[p + ".mean2"],
                                P[p + ".rstd2"]))
        self._gemm(P[p + ".xn2"], W_.p16(p + ".mlp.fc1.weight"), M, Hd, C, lda=C, ldb=C, epi=EPI_GELU_DUAL,
                 bias=W_.p32(p + ".mlp.fc1.bias"), out=P[p + ".h"], out2=P[p + ".g"], ldo2=Hd)
        if next_sp is not None:
            q = next_sp.prefix
            self._gemm_resid_ln(P[p + ".g"], W_.p16(p + ".mlp.fc2.weight"), M, C, Hd, lda=Hd, ldb=Hd,
                                bias=W_.p32(p + ".mlp.fc2.bias"), out=xout, aux=P[p + ".x1"], rowscale=self._ds(P, sp, 1),
                                tok=tok, out2=out_bf16,
                                ln=(W_.p32(q + ".norm1.weight"), W_.p32(q + ".norm1.bias"), P[q + ".xn1"], P[q + ".mean1"],
                                    P[q + ".rstd1"]))
            return
        self._gemm(P[p + ".g"], W_.p16(p + ".mlp.fc2.weight"), M, C, Hd, lda=Hd, ldb=Hd, epi=EPI_RESID_F32,
                 bias=W_.p32(p + ".mlp.fc2.bias"), out=xout, aux=P[p + ".x1"], ldaux=C,
                 rowscale=self._ds(P, sp, 1), rows_per_sample=tok, out2=out_bf16, ldo2=C if out_bf16 is not None else 0)

    # run_forward(pack_on_side=True): the weight-copy refresh in two halves, [fork event, issued, joined] each.  Half 0 (the
    # encoder's wide blocks) is forked behind the forward's first kernel and joined in front of the first wide block; half 1
    # (the decoder's) is forked in front of the LAST encoder stage -- the few-token stage whose small GEMMs leave most of the
    # chip idle -- and joined in front of the first decoder block that streams a copy: beside the 96-wide blocks at the head
    # of the forward the whole refresh cost those blocks ~15 us each (tools/step_stamps.py).
    # The deep stages' copies are a third piece forked in front of stage 2 (run_forward): batch 8 1.9426 -> 1.9315 ms, batch 64 9.022 ->
    # 9.007 (profiles/r5_ab_pack_layout2.txt).  One piece / two pieces cut the other way / the deep piece a stage earlier were all
    # measured within +-5 us or worse (profiles/r5_ab_pack_layout*.txt) and their switches are gone (round 6).
    _packs = None

    def _fork_pack(self, part, also=()):
        ev = torch.cuda.Event()
        ev.record()
        self._packs[part] = [ev, False, False, tuple(also)]

    def _issue_pack(self):
        """Enqueue the forked halves on the side stream -- called after the chain's next kernel (the graph executor keeps a
        node's first-created successor on the node's queue, see defer_side)."""
        for part, pk in (self._packs or {}).items():
            if not pk[1]:
                st = self._side_stream
                st.wait_event(pk[0])
                with torch.cuda.stream(st):
                    self.params.refresh_transposes(part, pk[3])
                pk[1] = True

    def _join_pack(self, prefix: Optional[str] = None, part: Optional[int] = None):
        """The chain waits for the halves a block with this prefix reads (None: for everything forked so far; part: that piece)."""
        if not self._packs:
            return
        mine = part if part is not None else (
            None if prefix is None else (2 if self._deep_prefix(prefix) else int(prefix.startswith("layers_up."))))
        need = [q for q in self._packs if mine is None or q is None or q == mine or mine in self._packs[q][3]]
        # (a piece this block does not read is NOT enqueued from here: it waits for _stage_fwd's _issue_pack behind the block's
        # first kernel -- enqueued in front of it, the piece becomes the first successor of the chain's last node and the graph
        # executor moves the chain to another queue and back: two ~9-us hops around the deep stage in the traced step; same-box
        # A/B 1.938 -> 1.926 ms, three pairs, profiles/r5_ab_join_issue.txt)
        if any(not self._packs[q][1] for q in need):
            self._issue_pack()
        if any(not self._packs[q][2] for q in need):
            torch.cuda.current_stream().wait_stream(self._side_stream)     # (one side stream: waits for all issued halves)
            for pk in self._packs.values():
                pk[2] = pk[2] or pk[1]

    def _deep_prefix(self, prefix: str) -> bool:
        d = getattr(self, "_deep_prefixes", None)
        if d is None:
            d = self._deep_prefixes = frozenset(sp.prefix for sp in self.blocks if sp.C >= 768)
        return prefix in d

    def _stage_fwd(self, P: Plan, specs: List[BlockSpec], xin, out_bf16=None):
        """out_bf16: bf16 copy of the stage output, written by the last block's fc2 epilogue."""
        x = xin
        ln1_done = False
        paired = False
        for k, sp in enumerate(specs):
            nxt = specs[k + 1] if k + 1 < len(specs) else None
            if paired:                             # ran in the previous block's launch
                paired = False
                x = P[sp.prefix + ".out"]
                continue
            if nxt is not None and self._pair96_ok(P, sp, nxt):
                self._pair96_fwd(P, sp, nxt, x)
                if out_bf16 is not None and k + 2 == len(specs):
                    ops.cast_f32_bf16(P[nxt.prefix + ".out"], out_bf16, P.B * nxt.H * nxt.W, nxt.C)
                paired, ln1_done = True, False
                x = P[sp.prefix + ".out"]
                continue
            chain = (nxt is not None and self._unfused(sp, P.B) and self._unfused(nxt, P.B) and nxt.C == sp.C
                     and nxt.H == sp.H and nxt.W == sp.W)
            self._block_fwd(P, sp, x, P[sp.prefix + ".out"], out_bf16 if k == len(specs) - 1 else None, ln1_done=ln1_done,
                            next_sp=nxt if chain else None)
            self._issue_pack()                     # (no-op unless a weight-copy refresh is waiting for the chain's next kernel)
            ln1_done = chain
            x = P[sp.prefix + ".out"]
        return x

    def _pair96_ok(self, P: Plan, sp: BlockSpec, nxt: BlockSpec) -> bool:
        """Two consecutive C = 96 blocks as ONE launch (tulip_swin96_pair_fwd): the un-shifted block and the shifted block behind
        it, where the stage is one round of the chip (every workgroup resident: <= 256 tiles of 8 windows)."""
        if not (self.pair96 and self.fuse_block96 and self._fusable96(sp) and self._fusable96(nxt)):
            return False
        if self._recomp96(sp) or self._recomp96(nxt) or (nxt.H, nxt.W) != (sp.H, sp.W):
            return False
        if not self._no_save and not (self._hgrad96(sp) and self._hgrad96(nxt)):     # the training form built: gelu'(h) in fc1_pre
            return False
        return P.B * (sp.H // 2) * (sp.W // 64) <= self.pair96_max_tiles

    def _desc96(self, P: Plan, sp: BlockSpec, xin, xout) -> dict:
        W_, p = self.params, sp.prefix
        sv = (lambda k: None) if self._no_save else (lambda k: P[p + k])
        return dict(
            x_in=xin, x1=sv(".x1"), x_out=xout, xn1=sv(".xn1"), qkv=sv(".qkv"), attn_out=sv(".o"),
            xn2=sv(".xn2"), fc1_pre=sv(".h"), fc1_act=sv(".g"), mean1=sv(".mean1"),
            rstd1=sv(".rstd1"), mean2=sv(".mean2"), rstd2=sv(".rstd2"),
            w_qkv=W_.p16(p + ".attn.qkv.weight"), w_proj=W_.p16(p + ".attn.proj.weight"),
            w_fc1=W_.p16(p + ".mlp.fc1.weight"), w_fc2=W_.p16(p + ".mlp.fc2.weight"),
            b_qkv=W_.p32(p + ".attn.qkv.bias"), b_proj=W_.p32(p + ".attn.proj.bias"),
            b_fc1=W_.p32(p + ".mlp.fc1.bias"), b_fc2=W_.p32(p + ".mlp.fc2.bias"),
            norm1_weight=W_.p32(p + ".norm1.weight"), norm1_bias=W_.p32(p + ".norm1.bias"),
            norm2_weight=W_.p32(p + ".norm2.weight"), norm2_bias=W_.p32(p + ".norm2.bias"),
            bias_table=W_.p32(p + ".attn.relative_position_bias_table"), rel_index=self._rel32,
            drop_scale_attn=self._ds(P, sp, 0), drop_scale_mlp=self._ds(P, sp, 1), B=P.B, H=sp.H, W=sp.W,
            shift_h=sp.sft[0], shift_w=sp.sft[1], masked=self._mask_arg(sp, P.B), eps=self.eps)

    def _pair96_fwd(self, P: Plan, sp: BlockSpec, nxt: BlockSpec, xin):
        nb = ops.swin96_pair_sync_bytes(P.B, sp.H, sp.W)
        P.scratch("xchg.pair." + sp.prefix, (nb + 3) // 4)      # zero at birth, epoch-stamped from then on (Plan.reset_exchange)
        mid = P[sp.prefix + ".out"]
        ops.swin96_pair_fwd(self._desc96(P, sp, xin, mid), self._desc96(P, nxt, mid, P[nxt.prefix + ".out"]),
                            P.bufs["xchg.pair." + sp.prefix])

    def _unmerge_fwd(self, P: Plan, prefix: str, s: int):
        """PatchUnmerging (tulip.py:117-123) of the level-s stream (its bf16 copy lvl{s}.xb, written by the producing
        stage) -> first half of the level s-1 concat buffer (tulip.py:715), bf16, straight from the PixelShuffle
        epilogue: the fp32 unmerged stream has no other reader."""
        m, W_ = self.model, self.params
        B = P.B
        H, W, C = self.grid[0] >> s, self.grid[1] >> s, m.embed_dim << s
        M = B * H * W
        if not m.patch_unmerging:
            # PatchExpanding (tulip.py:126-140): Linear C->2C (no bias), 'B H W (P1 P2 C) -> B (H P1) (W P2) C' and
            # LayerNorm(C/2); the rearrange is the output addressing of the LayerNorm kernel
            self._gemm(P[f"lvl{s}.xb"], W_.p16(prefix + ".expand.weight"), M, 2 * C, C, lda=C, ldb=C, epi=EPI_F32,
                       out=P[f"lvl{s}.ey"])
            ops.expand_norm_fwd(P[f"lvl{s}.ey"], W_.p32(prefix + ".norm.weight"), W_.p32(prefix + ".norm.bias"),
                                P[f"lvl{s}.emean"], P[f"lvl{s}.erstd"], B, H, W, 2, C // 2, self.eps,
                                out_bf16=P[f"dec{s - 1}.cat"], ld=C)
            return
        pk = self._pk(prefix + ".expand.weight")
        if pk is not None:
            self._join_pack(part=1)
        self._gemm(P[f"lvl{s}.xb"], W_.p16(prefix + ".expand.weight"), M, 2 * C, C, lda=C, ldb=C, epi=EPI_PIXSHUF2_F32,
                 bias=W_.p32(prefix + ".expand.bias"), out=None, out2=P[f"dec{s - 1}.cat"], ldo2=C, psH=H, psW=W, packed=pk)

    # round 6: the stage boundaries as one launch each (csrc/glue.hip).  TULIP_FUSE_GLUE=0: the LayerNorm / GEMM launches
    fuse_glue = knobs.on("TULIP_FUSE_GLUE", True)
    # two consecutive C = 96 blocks in one launch with tile-to-tile hand-off (csrc/swin96.hip, swin96_pair_fwd_kernel): review item 3's
    # prototype, forward only; kept or dropped on the measurement in profiles/README.md
    pair96 = knobs.on("TULIP_PAIR96", False)
    pair96_max_tiles = 256

    # Where the one-launch forms win (tools/bench_glue.py, profiles/r6_bench_glue_*.txt): every workgroup streams the boundary's whole
    # weight set through its CU, so with many row blocks the GEMM launches' tiles (weights shared by a whole tile column) take over.
    # Rows of the launch up to which the fused FORWARD forms run, by input width; the backward forms won at every size measured.
    embed_fold_on_chain = True
    glue_forms = frozenset(("merge_fwd", "merge_bwd", "unmerge_fwd", "unmerge_bwd"))      # (tests / A-B: a subset)
    glue_merge_fwd_max_rows = {96: 1 << 30, 192: 8192, 384: 2048}
    glue_unmerge_fwd_max_rows = {192: 1 << 30, 384: 8192}
    glue_unmerge_bwd_widths = (192, 384)

    def _has_copies(self, *names) -> bool:
        pk = getattr(self.params, "pk_offset", {})
        return all(n in pk for n in names)

    def _glue_merge_fwd(self, Cin: int, B: int, H: int, W: int) -> bool:
        return (self.fuse_glue and "merge_fwd" in self.glue_forms and 0 in self.params.pk_active
                and ops.merge_fwd_supported(Cin, B, H, W) and B * (H // 2) * (W // 2) <= self.glue_merge_fwd_max_rows.get(Cin, 0))

    def _glue_merge_bwd(self, Cp: int, B: int, H: int, W: int) -> bool:
        return (self.fuse_glue and "merge_bwd" in self.glue_forms and 0 in self.params.pk_active
                and ops.merge_bwd_supported(Cp, B, H, W))

    def _glue_unmerge(self, C: int, B: int, H: int, W: int, fwd: bool = False) -> bool:
        """(B,H,W,C): the coarse stage whose PatchUnmerging + the finer level's skip Linear run as one launch each way"""
        return (self.fuse_glue and ("unmerge_fwd" if fwd else "unmerge_bwd") in self.glue_forms and self.model.patch_unmerging
                and 0 in self.params.pk_active
                and ops.unmerge_skip_supported(C, B, H, W) and (fwd or C in self.glue_unmerge_bwd_widths) and (not fwd or B * H * W <= self.glue_unmerge_fwd_max_rows.get(C, 0)))

    fuse_tail_fwd = True      # norm_up + head + loss partials in one launch
    _loss_final = None

    def run_forward(self, P: Plan, with_loss: bool = True, pack_on_side: bool = False, defer_loss_final: bool = False):
        """TULIP.forward (tulip.py:702-737) on P.x_in / P.target -> P.pred, P.losses.
        pack_on_side (Trainer): the fragment-major weight copies of the fused wide blocks are rewritten from the bf16
        shadow beside the forward's first kernels (side stream, joined in front of the first wide block) instead of on
        the chain behind AdamW."""
        m, W_ = self.model, self.params
        if W_.shadow_dirty:
            W_.refresh_shadow()
        elif W_.pack_dirty and not pack_on_side:
            W_.refresh_transposes()
        self._packs = {}
        self._loss_final = None
        self._no_save = not with_loss and self.infer_no_save
        B, E, nl = P.B, m.embed_dim, m.num_layers
        H0, W0 = self.grid
        kw = 8 if m.circular_padding else m.patch_size[1]
        pend, self._pending_draw = self._pending_draw, None
        if pend is not None and pend[0] is not P:          # (drawn for another plan: a launch of its own)
            ops.drop_path_scales(*pend[1])
            pend = None
        ops.patch_embed_fwd(P.x_in, W_.p32("patch_embed.proj.weight"), W_.p32("patch_embed.proj.bias"),
                            W_.p32("patch_embed.norm.weight"), W_.p32("patch_embed.norm.bias"), P["enc0.in"], B,
                            m.in_chans, m.img_size[0], m.img_size[1], E, m.patch_size[0], m.patch_size[1], kw,
                            m.circular_padding, self.eps,
                            out_bf16=(P["dec0.cat"].data_ptr() + 2 * E) if nl > 1 else None, ld_bf16=2 * E,
                            draw=pend[1] if pend is not None else None)
        # three pieces: the encoder's wide blocks | the deep stages | the decoder's wide blocks
        two_packs = pack_on_side and W_.pk_active and nl > 2
        deep_pack = two_packs and nl >= 4 and any(w >= 768 for w in W_.pk_active)
        if pack_on_side and W_.pk_active:
            self._fork_pack(0 if two_packs else None, also=(2,) if (two_packs and not deep_pack) else ())
        # every encoder stage input is x_save[s]: its bf16 copy goes straight into the second half of the level's
        # concat buffer (tulip.py:715) from the kernel that produces it
        x = None
        # round 5: the deep stages' copies (57 MB read + 114 MB written per step for tulip_base's stage 3) as a piece of their own,
        # forked in front of stage 2 -- beside the C = 384 blocks, which stream weights from L2 and leave HBM alone -- instead of
        # riding in piece 0 beside the HBM-bound 96-wide blocks at the head of the forward (the second of them took 66 instead
        # of 32 us there in the traced step)
        for s in range(nl):
            if deep_pack and s == 2:
                self._fork_pack(2)
            if two_packs and s == nl - 1:
                self._fork_pack(1)
            x = self._stage_fwd(P, self.enc_blocks[s], P[f"enc{s}.in"],
                                out_bf16=P[f"lvl{s}.xb"] if (s == nl - 1 and nl > 1) else None)
            if s < nl - 1:  # PatchMerging (tulip.py:101-106)
                Hs, Ws, Cs = H0 >> s, W0 >> s, E << s
                rows = B * (Hs // 2) * (Ws // 2)
                pre = f"layers.{s}.downsample"
                save16 = (P[f"dec{s + 1}.cat"].data_ptr() + 2 * (2 * Cs)) if s + 1 < nl - 1 else None
                if self._glue_merge_fwd(Cs, B, Hs, Ws) and self._has_copies(pre + ".reduction.weight"):
                    # gather + LayerNorm + reduction GEMM in ONE launch (csrc/glue.hip)
                    self._join_pack(part=0)
                    ops.merge_fwd(x=x, gamma=W_.p32(pre + ".norm.weight"), beta=W_.p32(pre + ".norm.bias"),
                                  w_packed=W_.p16p(pre + ".reduction.weight"), xm=P[f"enc{s}.xm"], mean=P[f"enc{s}.mmean"],
                                  rstd=P[f"enc{s}.mrstd"], y=P[f"enc{s + 1}.in"], y_bf16=save16,
                                  ld_bf16=4 * Cs if save16 is not None else 0, B=B, H=Hs, W=Ws, Cin=Cs, eps=self.eps)
                    self._issue_pack()
                    continue
                ops.layernorm_fwd(x, W_.p32(pre + ".norm.weight"), W_.p32(pre + ".norm.bias"), P[f"enc{s}.xm"],
                                  P[f"enc{s}.mmean"], P[f"enc{s}.mrstd"], rows, 4 * Cs, self.eps, merge=True, B=B,
                                  H=Hs, W=Ws)
                self._gemm(P[f"enc{s}.xm"], W_.p16(pre + ".reduction.weight"), rows, 2 * Cs, 4 * Cs, lda=4 * Cs,
                         ldb=4 * Cs, epi=EPI_F32, out=P[f"enc{s + 1}.in"], out2=save16,
                         ldo2=4 * Cs if save16 is not None else 0)
        for i in range(nl - 1):
            s = nl - i - 2
            Cs = E << s
            Ms = B * (H0 >> s) * (W0 >> s)
            pre = f"skip_connection_layers.{i}"
            up = "first_patch_expanding" if i == 0 else f"layers_up.{i - 1}.upsample"
            if self._glue_unmerge(2 * Cs, B, H0 >> (s + 1), W0 >> (s + 1), fwd=True) and self._has_copies(up + ".expand.weight", pre + ".weight"):
                # PatchUnmerging of level s+1 -> skip Linear of level s in ONE launch (csrc/glue.hip): the x_save half of dec{s}.cat
                # was written by its producer, the unmerged half is written here (operand of the skip weight gradient)
                self._join_pack(part=1)
                ops.unmerge_skip_fwd(x_bf16=P[f"lvl{s + 1}.xb"], w_expand_packed=W_.p16p(up + ".expand.weight"),
                                     b_expand=W_.p32(up + ".expand.bias"), cat=P[f"dec{s}.cat"],
                                     w_skip_packed=W_.p16p(pre + ".weight"), b_skip=W_.p32(pre + ".bias"), out=P[f"dec{s}.in"],
                                     B=B, H=H0 >> (s + 1), W=W0 >> (s + 1), C=2 * Cs)
                self._issue_pack()
            else:
                self._unmerge_fwd(P, up, s + 1)
                # dec{s}.cat = cat[unmerged stream, x_save[s]] (tulip.py:715) was filled by its two producers
                pk = self._pk(pre + ".weight")
                if pk is not None:
                    self._join_pack(part=1)
                self._gemm(P[f"dec{s}.cat"], W_.p16(pre + ".weight"), Ms, Cs, 2 * Cs, lda=2 * Cs, ldb=2 * Cs, epi=EPI_F32,
                           bias=W_.p32(pre + ".bias"), out=P[f"dec{s}.in"], packed=pk)
            x = self._stage_fwd(P, self.dec_blocks[i], P[f"dec{s}.in"],
                                out_bf16=P[f"lvl{s}.xb"] if i < nl - 2 else None)
        M0 = B * H0 * W0
        loss_done = False
        if m.pixel_shuffle and self.fuse_tail_fwd:
            # norm_up -> expand conv -> LeakyReLU -> PixelShuffle(4) -> decoder_pred (-> L1 partial sums) in ONE launch
            ops.tail_fwd_ln(x, W_.p32("norm_up.weight"), W_.p32("norm_up.bias"), self.eps, P["tail.xn"], P["tail.mean"],
                            P["tail.rstd"], W_.p16("ps_head.conv_expand.0.weight"), W_.p32("ps_head.conv_expand.0.bias"),
                            W_.p32("decoder_pred.weight"), P.pred, B, H0, W0, E, target=P.target if with_loss else None,
                            loss_partials=P.partials if with_loss else None, log_transform=m.log_transform)
            if with_loss:
                fin = lambda: ops.l1_loss_final(P.partials, P.losses, (M0 + 31) // 32, P.pred.numel(), m.log_transform)
                if defer_loss_final:
                    self._loss_final = fin            # run_backward issues it beside the chain (nothing on the GPU reads it)
                else:
                    fin()
            loss_done = True
        elif m.pixel_shuffle:
            ops.layernorm_fwd(x, W_.p32("norm_up.weight"), W_.p32("norm_up.bias"), P["tail.xn"], P["tail.mean"],
                              P["tail.rstd"], M0, E, self.eps)
            ops.tail_fwd(P["tail.xn"], W_.p16("ps_head.conv_expand.0.weight"), W_.p32("ps_head.conv_expand.0.bias"),
                         W_.p32("decoder_pred.weight"), P.pred, B, H0, W0, E)
        else:
            # FinalPatchExpanding (tulip.py:144-159) + decoder_pred (tulip.py:731): Linear E -> r^2 E, then one kernel
            # for rearrange + LayerNorm(E) + the 1x1 conv as a per-row dot product
            ops.layernorm_fwd(x, W_.p32("norm_up.weight"), W_.p32("norm_up.bias"), P["tail.xn"], P["tail.mean"],
                              P["tail.rstd"], M0, E, self.eps)
            r = m.upscale_factor
            pre = "final_patch_expanding"
            self._gemm(P["tail.xn"], W_.p16(pre + ".expand.weight"), M0, r * r * E, E, lda=E, ldb=E, epi=EPI_F32,
                       out=P["tail.ey"])
            ops.expand_norm_fwd(P["tail.ey"], W_.p32(pre + ".norm.weight"), W_.p32(pre + ".norm.bias"), P["tail.emean"],
                                P["tail.erstd"], B, H0, W0, r, E, self.eps, dotw=W_.p32("decoder_pred.weight"),
                                pred=P.pred)
        self._join_pack()                          # (a model without fused wide blocks never asked for the copies)
        if with_loss and not loss_done:
            ops.l1_loss_fwd(P.pred, P.target, P.partials, P.losses, P.pred.numel(), m.log_transform)
        P.generation += 1
        P.last_x = x

    # ------------------------------------------------------------------ backward
    WS_ELEMS = 16 << 20  # fp32 elements of split-K slab workspace (64 MiB) (+ bias slabs behind it)

    # weight gradients run beside the latency-bound chain: what matters is how much they disturb it, not their
    # own latency.  >= 1024 tokens per split keeps the slab traffic (splits x output, written then folded) at a
    # quarter of what "fill the chip" splitting (256 tokens) produced: step 4.07 -> 3.96 ms.
    WGRAD_CTAS = 512
    WGRAD_MINK = 1024
    WGRAD_BIG_CTAS = 256
    WGRAD_BIG_MINK = 512
    # A large-tile workgroup takes a whole CU (8 waves x 256 registers, 92 KB of LDS): a group is sized to one round of
    # the CUs that are FREE.  With a gradient all-reduce running beside the backward RCCL's channel workgroups hold wave
    # slots on up to ~32 CUs for the length of a collective, and a 256-workgroup launch would need a second round for
    # the workgroups those CUs cannot take; the Trainer sets this to WGRAD_DDP_CTAS when world_size > 1: 192 leaves room
    # for up to 64 channels and costs 0.2 % of the one-GPU step (2.749 vs 2.743 ms).  Not measured on a multi-GPU node;
    # TULIP_WGRAD_DDP_CTAS overrides.
    wgrad_ctas = 0                                   # 0: WGRAD_BIG_CTAS
    WGRAD_DDP_CTAS = knobs.integer("TULIP_WGRAD_DDP_CTAS", 192)

    # Workgroups of a grouped large-tile launch as a function of the tiles it holds ("min_tiles:workgroups;..."): a stage
    # with many tiles (C = 384: ~112) fills half the chip without any token split -- no slabs, nothing to fold -- and the
    # step time is flat in the workgroup count (profiles/README.md), so the split-K traffic is what decides.  Default
    # (round 3 sweep, same box, batch 8 / 64 ms per step): all 256: 2.2755 / 9.691; this map: 2.2784 / 9.717 with the slabs
    # of a step at 70 MB instead of 220 MB; "100:112;20:28" (no split at C = 192 either): 2.360 / 10.16.
    wgrad_ctas_map = ((100, 112), (20, 128), (5, 128))

    def _group_ctas(self, group_tiles: int) -> int:
        if not self.wgrad_ctas:                      # (a DDP run pins the count: RCCL's channels hold CUs)
            for min_tiles, ctas in self.wgrad_ctas_map:
                if group_tiles >= min_tiles:
                    return ctas
        return self.wgrad_ctas

    @classmethod
    def _splits(cls, Mout: int, Nout: int, K: int, group_tiles: int = 0, ctas: int = 0) -> int:
        if group_tiles:  # large tiles (192 x 192 / 384 x 96 / 96 x 384), `group_tiles` of them per split in this launch
            s = max(1, min((ctas or cls.WGRAD_BIG_CTAS) // group_tiles, K // cls.WGRAD_BIG_MINK, cls.WS_ELEMS // (Mout * Nout)))
        else:
            tiles = ((Mout + 127) // 128) * ((Nout + 95) // 96)
            s = max(1, min(cls.WGRAD_CTAS // max(tiles, 1), K // cls.WGRAD_MINK, cls.WS_ELEMS // (Mout * Nout)))
        while True:  # the kernel cuts K in multiples of 32: iterate to the split count it really launches
            e = ops.gemm_effective_splits(K, s)
            if e == s:
                return s
            s = e

    # round 6: the stage-boundary GEMMs that no fused launch owns read the fragment-major weight copy in the small-K form
    # (TULIP_GEMM_B_PACKED, csrc/gemm.hip gemm_stream_kernel: 32 x 96 tiles, the whole K range of a split in flight) while the launch
    # is at most this many tiles -- 54 -> 41 us for the six of the batch-8 step in isolation (profiles/r6_bench_gemm_packed.txt)
    packed_gemm = True
    packed_gemm_max_tiles = 1024

    def _pk(self, name: str, transposed: bool = False, row0: int = 0, K: int = 0) -> Optional[int]:
        """Address of the fragment-major copy of weight `name` (transposed: of its transpose), from matrix row `row0` on (a multiple
        of 16; K = the copy's row length), or None where no fresh copy exists."""
        W_ = self.params
        if not (self.packed_gemm and self.fuse_glue and 0 in getattr(W_, "pk_active", ()) and W_.pk_width.get(name) == 0):
            return None
        return (W_.p16t(name) if transposed else W_.p16p(name)) + 2 * (row0 // 16) * (K // 32) * 512

    def _gemm(self, A, B, M, N, K, packed: Optional[int] = None, **kw):
        """Forward / dgrad GEMM.  Launches that cannot fill the chip and have a deep K (the M=512..2048 GEMMs
        of stages 2-3 stream megabytes of weights through a few dozen workgroups) are split along K; the
        library folds the partial slabs and applies the fused epilogue in a second kernel.
        packed: address of the fragment-major copy of the [N][K] operand (_pk) -- taken where the small-K form exists."""
        gn = (N + 95) // 96
        blocks = ((M + 63) // 64) * gn if ((M + 127) // 128) * gn < 256 else ((M + 127) // 128) * gn
        s = 1
        if blocks <= 96 and K >= 768:
            s = max(1, min(256 // blocks, K // 256, (self.WS_ELEMS * 4) // (M * N * 4)))
        if packed is not None and not kw.get("a_trans"):
            # unsplit where the form holds the whole K (<= 1536: a 96-KB panel): the fold launch and its slabs go, 128 workgroups with
            # everything in flight take what 512 + a fold took (the deepest expand conv's data gradient: 12.9 -> 8.6 us isolated)
            sp = 1 if ops.gemm_packed_supported(M, N, K, 1) else ops.gemm_effective_splits(K, s)
            if ops.gemm_packed_supported(M, N, K, sp) and (M // 32) * (N // 96) * sp <= self.packed_gemm_max_tiles:
                kw2 = dict(kw, b_trans=False, ldb=K)
                ops.gemm(A, packed, M, N, K, splits=sp, workspace=self._ws_ptr, workspace_bytes=self.WS_ELEMS * 4, b_packed=True, **kw2)
                return
        if s > 1:
            ops.gemm(A, B, M, N, K, splits=s, workspace=self._ws_ptr, workspace_bytes=self.WS_ELEMS * 4, **kw)
            return
        # round 5: the narrow-output / deep-K shapes of the mid-size stages (fc2 forward, the fc1 / qkv data gradients: N = C,
        # K = 3C .. 4C at M >= 4096 -- the deep stage at batch 64, stage 3 of tulip_large at 32 x 2048) run 330-410 TFLOP/s on
        # gemm_tile's tiles; the 192 x 192 loader-wave kernel (csrc/gemm.hip, gemm_mid_tile) with the K split that fills the chip
        # takes them at 450-680 isolated (profiles/r5_gemm_big.txt).  In the step that is worth 0.4 % on DurLAR tulip_large, nothing
        # at batch 64 and -0.7 % at M = 2048 (four slabs to fold; profiles/r5_ab_mid_gemm.txt): M >= 4096 only.  Wide outputs over
        # K = 768 stay where they are (485-710 TFLOP/s already; a 192 x 192 tile's fixed cost is not repaid by 24 k-steps).
        if self.mid_gemm and M >= 4096 and K >= 1536 and K >= 2 * N and K % 128 == 0 and kw.get("epi") not in (EPI_PIXSHUF2_F32, EPI_UNSHUF2_BF16) \
                and not kw.get("a_trans"):
            tiles = ((M + 191) // 192) * ((N + 191) // 192)
            s = max(1, min(256 // tiles, K // 768, (self.WS_ELEMS * 4) // (M * N * 4)))
            while s > 1 and K % (64 * s):
                s -= 1
            if tiles * s >= 128:
                ops.gemm(A, B, M, N, K, splits=s, workspace=self._ws_ptr, workspace_bytes=self.WS_ELEMS * 4, mid=True, **kw)
                return
        ops.gemm(A, B, M, N, K, **kw)

    mid_gemm = True

    # Side launches per STAGE, not per block (default since the side queue, not the chain, ends the backward): a grouped
    # weight-gradient launch is one round of the chip whatever it holds, so its slabs are ~one 147-KB tile per workgroup --
    # 28-37 MB written, read back by the fold -- per LAUNCH.  A stage's two blocks and its boundary linears in one launch
    # (up to 12 linears) halve the launches, the slab traffic (1.36 -> 0.7 GB per step) and the fold launches: 2.46 -> 2.33 ms
    # at batch 8, 10.41 -> 10.24 ms at batch 64 (same box; per-block launches for the backward's last stage only: 2.36).
    # (per-block side launches, a flush behind every unfused block, the patch-embedding fold riding in the last group: measured worse or
    # neutral in rounds 2-5, profiles/README.md; their switches are gone)
    wgrad_group_max = knobs.integer("TULIP_WGRAD_GROUP_MAX", 12)    # linears per grouped launch (<= 16)
    early_flush = frozenset()   # block prefixes with a mid-block side flush (set in bind())
    _lagged_hook = None
    # Side streams.  How the HIP graph executor (ROCm 7.2) turns captured branches into hardware-queue work decides what
    # a fork costs the chain: it walks the graph depth first along each node's FIRST-created successor and gives every
    # other successor a new run list on another queue.  With the side kernels enqueued right at the fork (defer_side
    # off) they are the first successor, so the chain itself changes queue at every fork (~12 us each; then 4, 8, 12...
    # side streams used round-robin are best: 1 stream 4.63 ms, 2: 4.46, 3: 4.82, 4: 4.05, 8: 4.05 at the time).  With
    # the fork deferred (_release_deferred) the chain's next kernel is created first, the chain stays on one queue for the whole
    # step, and a single side stream is best (1: 3.32 ms, 2: 3.59, 3: 3.63, 4: 3.67, 8: 3.63; 3.42 for the old layout; two side
    # streams re-measured in rounds 3 and 5: +0.36 ms).  One side stream, always deferred: the switches are gone (round 6).
    overlap_wgrad = True   # run the weight-gradient branch on a second HIP stream (forked inside the graph)

    def _wgrad(self, dY, ldy, X, ldx, Nw, Kw, Mtok, gout, gbias=None):
        """dW[Nw,Kw] += dY[Mtok,Nw]^T . X[Mtok,Kw]  and, for free, db[Nw] += sum_tokens dY (row sums of the
        transposed operand via one extra all-ones MFMA per fragment).  The token dimension is split over
        workgroups to fill the chip; partials go to fp32 slabs that one reduce kernel folds into both
        gradients (deterministic, no atomics).

        Nothing downstream in the backward depends on a weight gradient, so the whole branch (GEMM + fold)
        is issued on a side stream that forks from the main chain here and is joined at the DDP bucket
        points / end of backward: the latency-bound dgrad -> LayerNorm -> attention chain and the wgrad
        GEMMs overlap on the chip (captured as parallel branches of the HIP graph)."""
        if not self.overlap_wgrad:
            return self._wgrad_launch(dY, ldy, X, ldx, Nw, Kw, Mtok, gout, gbias, self._ws_ptr)
        # queued: forks cost a few microseconds each inside a HIP graph and every side launch has a fixed cost, so the
        # weight gradients of a whole stage share ONE fork (their inputs are per-block buffers: deferring them is hazard-free)
        self._pending.append(("w", (dY, ldy, X, ldx, Nw, Kw, Mtok, gout, gbias)))

    def _side(self, fn):
        """Queue work nothing in the backward chain waits for for the side stream."""
        if self.overlap_wgrad:
            self._pending.append(("f", fn))
        else:
            fn()

    def _side_first(self, fn):
        """Like _side, but issued BEFORE the flush's fold launch (a kernel that produces slabs the folds read)."""
        if self.overlap_wgrad:
            self._pending.append(("p", fn))
        else:
            fn()

    def _fold(self, part, stride, out, n, rows, **kw):
        """Queue a fold of per-workgroup partial rows: out[i] += sum_r part[r*stride + i] (LayerNorm affine gradients,
        relative-position-bias gradients, ...).  Folds queued by one block leave in the same launch as the folds of
        its weight-gradient slabs."""
        kw.setdefault("overwrite", self.grad_overwrite)
        if kw["overwrite"] and self.fuse_adamw_folds and self._in_gflat(out):
            # the fold produces the complete gradient of a parameter range: the optimizer step can be taken in it (Trainer.fuse_adamw)
            if self.adam_probe is not None:
                self.adam_probe[out] = max(n, self.adam_probe.get(out, 0))
            kw["adamw"] = self.adam_apply and out in self.adam_fused
        r = ops.reduce_region(part, stride, out, n, rows, **kw)
        if self.overlap_wgrad:
            self._pending.append(("r", r))
        else:
            ops.reduce_rows_multi([r], adam=self._adam_arg())

    def _in_gflat(self, ptr) -> bool:
        g = self._gflat
        return g is not None and isinstance(ptr, int) and g.data_ptr() <= ptr < g.data_ptr() + 4 * g.numel()

    def _adam_arg(self):
        return self.adam_ctx if self.adam_apply else None

    def _fold_bias_table(self, P: Plan, tag: str, part, rows: int, nh: int, gtable):
        """Relative-position-bias gradient of one block: partial rows [rows][nh*256] (dense (head, query, key) sums per
        workgroup) -> table gradient [45][nh] through the relative-position index (tulip.py:304-308 backwards).  Two
        steps, both deterministic: the dense fold runs with every other fold of the block (all columns in parallel);
        the scatter of its [nh*256] result -- one workgroup per head adding the pairs of every table entry in index
        order -- rides in the NEXT side launch (a region of one row), so no launch is added and no workgroup has to
        pull a whole partial matrix through one CU."""
        dense = P.scratch("apd." + tag, nh * 256)
        self._fold(part, nh * 256, dense, nh * 256, rows, overwrite=True)
        r = ops.reduce_region(dense, nh * 256, gtable, nh * 256, 1, overwrite=self.grad_overwrite, scatter_index=self._rel32,
                              scatter_nh=nh, scatter_len=256)
        if self.overlap_wgrad:
            self._pending.append(("s", r))
        else:
            ops.reduce_rows_multi([r])

    _carry = ()          # scatter regions waiting for the next side launch
    _glue_done = {}      # level -> arguments of the fused skip' + PatchUnmerging' launch that _unmerge_bwd issues (run_backward)

    def _flush_carry(self):
        """Launch the scatters still waiting (end of the backward / a DDP bucket point) on the side stream."""
        if self._carry:
            carry, self._carry = list(self._carry), ()
            with torch.cuda.stream(self._side_stream):
                ops.reduce_rows_multi(carry, adam=self._adam_arg())
            self._side_dirty = True

    group_wgrad = True

    def _issue_pending(self, ws: int, pending=None):
        """Launch the queued side work on the current stream: weight gradients as grouped GEMMs (<= wgrad_group_max per launch), every
        fold in the launch that folds the slabs, other closures last."""
        pending = self._pending if pending is None else pending
        for fn in [a for k, a in pending if k == "p"]:         # producers of slabs that this flush's fold launch consumes
            fn()
        full_chip = any(k == "c" for k, _ in pending)          # the backward's last stage: nothing left to run beside it
        items = [a for k, a in pending if k == "w"]
        regions = list(self._carry) + [a for k, a in pending if k == "r"]
        self._carry = tuple(a for k, a in pending if k == "s")     # their dense sums are produced by THIS launch
        fns = [a for k, a in pending if k == "f"]
        ws_bytes = (self.WS_ELEMS + (1 << 20)) * 4
        if not self.group_wgrad:
            for a in items:
                self._wgrad_launch(*a, ws)
            items = []
        from . import _lib
        while items:
            grp, used, any_adam = [], 0, False
            gmax = self.wgrad_group_max
            cand = items[:gmax]
            # (the large-tile kernel also needs whole 32-token k-steps: same test as tulip_wgrad_group, csrc/gemm.hip)
            big = all(ops.wgrad_tiles(a[4], a[5], self.wgrad_small_tiles) != ((a[4] + 63) // 64) * ((a[5] + 95) // 96) and a[6] % 32 == 0 for a in cand)
            # large tiles run one workgroup per CU: the token splits of a launch are sized so that the whole group is
            # about one round of the chip
            group_tiles = sum(ops.wgrad_tiles(a[4], a[5], self.wgrad_small_tiles) for a in cand) if big else 0
            while items and len(grp) < gmax:
                dY, ldy, X, ldx, Nw, Kw, Mtok, gout, gbias = items[0]
                sp = self._splits(Nw, Kw, Mtok, group_tiles=group_tiles, ctas=self.wgrad_ctas if full_chip else self._group_ctas(group_tiles))
                need = (Nw * Kw + Nw) * sp * 4 if sp > 1 else 0
                if grp and used + need > ws_bytes:
                    break
                used += need
                # an un-split large-tile item holds its complete gradient tile in the write-out: the optimizer step can be taken
                # there (weight only); a token-split item's step is taken by the fold of its slabs (weight and bias)
                elig = self.grad_overwrite and ((sp > 1 and self.fuse_adamw_folds) or (sp == 1 and big))
                if elig and self.adam_probe is not None:
                    self.adam_probe[gout] = Nw * Kw
                    if sp > 1 and gbias is not None:
                        self.adam_probe[gbias] = Nw
                step_here = (elig and self.adam_apply and gout in self.adam_fused
                             and (sp == 1 or gbias is None or gbias in self.adam_fused))
                any_adam = any_adam or step_here
                grp.append(ops.wgrad_item(dY, ldy, X, ldx, Nw, Kw, Mtok, gout, gbias, sp, overwrite=self.grad_overwrite,
                                          adamw=step_here))
                items.pop(0)
            room = _lib.REDUCE_REGIONS_MAX - 2 * len(grp)
            extra, regions = regions[:room], regions[room:]
            ops.wgrad_group(grp, extra, ws, ws_bytes, adam=self._adam_arg(), small_tiles=self.wgrad_small_tiles)
        while regions:
            ops.reduce_rows_multi(regions[:_lib.REDUCE_REGIONS_MAX], adam=self._adam_arg())
            regions = regions[_lib.REDUCE_REGIONS_MAX:]
        for fn in fns:
            fn()

    def _flush_wgrads(self, advance: bool = True, mark: bool = False):
        """Fork the queued side work: the fork point is here, the launches are enqueued behind the chain's next kernel
        (_release_deferred); advance=False: a mid-block flush, enqueued at once.
        mark: an event behind this group on the side stream (pack_at_end: every packable weight has been stepped once it has run)."""
        st, ws = self._side_stream, self._ws_side_ptr
        if not self._pending:
            if mark:
                # nothing queued at the marked hook: everything that steps a packed weight is already forked (or still deferred)
                self._mark_side()
            return
        main = torch.cuda.current_stream()
        if advance:                         # (advance=False: a mid-block flush, issued at once; the block's remainder follows)
            # the fork point is HERE, but the side kernels are enqueued only after the chain's next kernel (see
            # _release_deferred): the graph executor keeps a node's first-created successor on the node's queue
            self._release_deferred()
            ev = torch.cuda.Event()
            ev.record(main)
            self._deferred = (ev, st, ws, self._pending, mark)
            self._pending = []
            self._side_dirty = True
            return
        st.wait_stream(main)
        with torch.cuda.stream(st):
            self._issue_pending(ws)
        if mark:
            self._pack_ev = torch.cuda.Event()
            self._pack_ev.record(st)
        self._pending = []
        self._side_dirty = True

    _deferred = None
    pack_at_end = False           # per run_backward call (the Trainer's choice, Trainer._pack_at_end)
    _pack_ev = None

    @property
    def _pack_mark_tag(self):
        """backward hook of the LAST completion group that holds a packed weight: the lowest encoder stage with a packed width"""
        t = getattr(self, "_pack_mark_tag_", None)
        if t is None:
            W_ = self.params
            last = -1
            for n in W_.packed_names():
                if n not in W_.pk_late:
                    last = max(last, next(k for k, (_, end) in enumerate(W_.groups) if W_.offset[n] < end))
            t = self._pack_mark_tag_ = W_.groups[last][0] if last >= 0 else ""
        return t

    def _mark_side(self):
        """_pack_ev = everything forked so far has run (the deferred group is enqueued first; the side stream joins the chain's
        position, so the event is valid inside a capture even when nothing was forked yet)."""
        self._release_deferred()
        self._side_stream.wait_stream(torch.cuda.current_stream())
        self._pack_ev = torch.cuda.Event()
        self._pack_ev.record(self._side_stream)
        self._side_dirty = True

    def _release_deferred(self):
        d, self._deferred = self._deferred, None
        if d is None:
            return
        ev, st, ws, pending, mark = d
        st.wait_event(ev)
        with torch.cuda.stream(st):
            self._issue_pending(ws, pending)
        if mark:
            self._pack_ev = torch.cuda.Event()
            self._pack_ev.record(st)

    # run_backward(join_tags=...) under Trainer._capture: see hook() in run_backward
    detach_buckets = False
    _detached = None

    def take_detached(self):
        d, self._detached = self._detached, None
        return d

    def issue_detached(self, pending, ws: int):
        """The side group hook() held back, on the CURRENT stream (the capture of its own graph); the scatter regions it leaves
        for "the next side launch" are folded right behind it."""
        self._issue_pending(ws, pending)
        if self._carry:
            carry, self._carry = list(self._carry), ()
            ops.reduce_rows_multi(carry, adam=self._adam_arg())

    def _wait_side(self):
        """The current stream waits for everything issued on the side streams so far (queued work stays queued)."""
        self._release_deferred()
        self._flush_carry()
        if getattr(self, "_side_dirty", False):
            torch.cuda.current_stream().wait_stream(self._side_stream)
            self._side_dirty = False

    def _join_side(self):
        self._flush_wgrads()
        self._release_deferred()
        self._flush_carry()
        if getattr(self, "_side_dirty", False):
            torch.cuda.current_stream().wait_stream(self._side_stream)
            self._side_dirty = False

    def _wgrad_launch(self, dY, ldy, X, ldx, Nw, Kw, Mtok, gout, gbias, ws):
        splits = self._splits(Nw, Kw, Mtok)
        if splits == 1:
            ops.gemm(dY, X, Nw, Kw, Mtok, lda=ldy, ldb=ldx, a_trans=True, b_trans=True, epi=EPI_F32, out=gout, ldo=Kw,
                     accumulate=True, out2=gbias)
            return
        wsb = ws + 4 * splits * Nw * Kw if gbias is not None else None
        ops.gemm(dY, X, Nw, Kw, Mtok, lda=ldy, ldb=ldx, a_trans=True, b_trans=True, epi=EPI_SPLIT_F32, out=ws, ldo=Kw,
                 splits=splits, out2=wsb)
        ops.reduce_rows2(ws, Nw * Kw, gout, Nw * Kw, wsb, Nw, gbias, Nw if gbias is not None else 0, splits)

    def _dgrad_ln_bwd(self, P: Plan, dY, Wt, M, C, K, dxn, ln_args, ln_kw):
        """Data gradient of a Linear whose input came out of a LayerNorm (fc1 / qkv): dxn = dY . W, then the LayerNorm
        backward.  Where the GEMM is split along K its raw slabs go straight into the LayerNorm backward, which folds them
        (tulip_layernorm_bwd_splitk) -- the fold launch disappears; same bits as the two-launch form."""
        gn = (C + 95) // 96
        blocks = ((M + 63) // 64) * gn if ((M + 127) // 128) * gn < 256 else ((M + 127) // 128) * gn
        s = 1
        if self.fuse_splitk_ln and blocks <= 96 and K >= 768 and ops.layernorm_bwd_partial_rows(M, C) > 0:
            s = ops.gemm_effective_splits(K, max(1, min(256 // blocks, K // 256, (self.WS_ELEMS * 4) // (M * C * 4))))
        if s > 1:
            ops.gemm(dY, Wt, M, C, K, lda=K, ldb=C, b_trans=True, epi=EPI_SPLIT_F32, out=self._ws_ptr, ldo=C, splits=s)
            self._ln_bwd(P, None, *ln_args, slabs=(self._ws_ptr, s), **ln_kw)
            return
        self._gemm(dY, Wt, M, C, K, lda=K, ldb=C, b_trans=True, epi=EPI_BF16, out=dxn, ldo=C)
        self._ln_bwd(P, dxn, *ln_args, **ln_kw)

    def _ln_bwd(self, P: Plan, dy, x, mean, rstd, gamma, dres, dx, rows, C, gw, gb, tag, merge=False, H=0, W=0,
                cast=None, slabs=None):
        """LayerNorm backward: dx (+= dres) on the main chain; the affine gradients leave as per-workgroup
        partial rows (private buffer `tag`) that are folded on the side stream.  cast = (bf16 buffer,
        rowscale address or None, tokens per sample): the operand of the next GEMM on the chain,
        bf16(dx * DropPath scale), emitted by the same kernel."""
        cb, cs, ct = cast if cast is not None else (None, None, 1)
        nrows = ops.layernorm_bwd_partial_rows(rows, C)
        if nrows == 0:  # C > 2048 (tulip_large's deepest PatchMerging norm): stand-alone parameter pass
            if self.grad_overwrite:      # that pass ADDS: the two ranges are cleared here instead of by AdamW
                g0 = self._gflat.data_ptr()
                for ptr in (gw, gb):
                    self._gflat[(ptr - g0) // 4:(ptr - g0) // 4 + C].zero_()
            ops.layernorm_bwd_params(dy, x, mean, rstd, gw, gb, rows, C, merge=merge, B=P.B, H=H, W=W)
            ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, rows, C, merge=merge, B=P.B, H=H, W=W,
                              dx_bf16=cb, cast_rowscale=cs, cast_rows_per_sample=ct)
            return
        part = P.scratch("lnp." + tag, nrows * 2 * C)
        if slabs is not None:          # the incoming gradient is still the raw split-K slabs of the GEMM in front
            ops.layernorm_bwd_splitk(slabs[0], slabs[1], x, mean, rstd, gamma, dres, dx, rows, C, merge=merge, B=P.B, H=H,
                                     W=W, param_partials=part, dx_bf16=cb, cast_rowscale=cs, cast_rows_per_sample=ct)
        else:
            ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, rows, C, merge=merge, B=P.B, H=H, W=W,
                              param_partials=part, dx_bf16=cb, cast_rowscale=cs, cast_rows_per_sample=ct)
        self._fold(part, 2 * C, gw, C, nrows)
        self._fold(part + 4 * C, 2 * C, gb, C, nrows)

    def _mlp_cast(self, P: Plan, sp: BlockSpec):
        """What the producer of this block's incoming gradient should emit: (dyb_m, DropPath scale, tokens)."""
        if self._fused_bwd(sp, P.B):
            return None                     # the fused block backward forms its own operand from the fp32 gradient
        return (P[sp.prefix + ".dyb_m"], self._ds(P, sp, 1), sp.H * sp.W)

    def _block_bwd(self, P: Plan, sp: BlockSpec, xin, dx, G, have_dyb=False, next_cast=None):
        """In-place: dx (grad w.r.t. block output) -> grad w.r.t. block input.  G(name) = grad address.
        have_dyb: the producer of dx already wrote bf16(dx*scale) into this block's dyb_m;
        next_cast: what the consumer of this block's input gradient wants (see _ln_bwd)."""
        W_ = self.params
        p = sp.prefix
        B, C, nh = P.B, sp.C, sp.nh
        M, Hd, tok = B * sp.H * sp.W, self.hidden(sp.C), sp.H * sp.W
        dxn, dO, dh, dqkv = P["t.dxn"], P["t.do"], P[p + ".dh"], P[p + ".dqkv"]
        dyb = P[p + ".dyb_m"]
        if self._fusable_deep(sp, B):
            # csrc/swind.hip: fc2' + GELU' (by hidden channels) -> fc1' (by output channels) -> norm2' -> proj' + attention' (by heads)
            # -> qkv' (by output channels) -> norm1'; the LayerNorm backward launches are the sequence's own, fed one fp32 "slab"
            dn = P.scratch("deep.dxn", max(P.B * q.H * q.W * q.C for q in self.blocks if self._fusable_deep(q, P.B)))
            R = ops.swind_groups(C, B, sp.H, sp.W, sp.win)
            apart = P.scratch("apart." + p, R * nh * 256)
            wt = W_.p16t
            desc = dict(
                dx=dx, x_in=xin, x1=P[p + ".x1"], qkv=P[p + ".qkv"], fc1_pre=P[p + ".h"], mean1=P[p + ".mean1"],
                rstd1=P[p + ".rstd1"], mean2=P[p + ".mean2"], rstd2=P[p + ".rstd2"],
                w_qkv=wt(p + ".attn.qkv.weight"), w_proj=wt(p + ".attn.proj.weight"),
                w_fc1=wt(p + ".mlp.fc1.weight"), w_fc2=wt(p + ".mlp.fc2.weight"),
                norm1_weight=W_.p32(p + ".norm1.weight"), norm2_weight=W_.p32(p + ".norm2.weight"),
                bias_table=W_.p32(p + ".attn.relative_position_bias_table"), rel_index=self._rel32,
                drop_scale_attn=self._ds(P, sp, 0), drop_scale_mlp=self._ds(P, sp, 1), d_out_mlp=dyb, d_fc1_pre=dh,
                d_out_attn=P[p + ".dyb_a"], d_qkv=dqkv, bias_partials=apart, B=B, H=sp.H, W=sp.W, shift_h=sp.sft[0],
                shift_w=sp.sft[1], masked=self._mask_arg(sp, B))
            ops.swind_block_bwd(C, sp.win, dn, phases=3, **desc)
            self._release_deferred()
            self._wgrad(dyb, C, P[p + ".g"], Hd, C, Hd, M, G(p + ".mlp.fc2.weight"), G(p + ".mlp.fc2.bias"))
            self._wgrad(dh, Hd, P[p + ".xn2"], C, Hd, C, M, G(p + ".mlp.fc1.weight"), G(p + ".mlp.fc1.bias"))
            self._ln_bwd(P, None, P[p + ".x1"], P[p + ".mean2"], P[p + ".rstd2"], W_.p32(p + ".norm2.weight"), dx, dx, M, C,
                         G(p + ".norm2.weight"), G(p + ".norm2.bias"), p + ".2", cast=(P[p + ".dyb_a"], self._ds(P, sp, 0), tok),
                         slabs=(dn, 1))
            ops.swind_block_bwd(C, sp.win, dn, phases=12, **desc)
            self._wgrad(P[p + ".dyb_a"], C, P[p + ".o"], C, C, C, M, G(p + ".attn.proj.weight"), G(p + ".attn.proj.bias"))
            self._fold_bias_table(P, p, apart, R, nh, G(p + ".attn.relative_position_bias_table"))
            self._ln_bwd(P, None, xin, P[p + ".mean1"], P[p + ".rstd1"], W_.p32(p + ".norm1.weight"), dx, dx, M, C,
                         G(p + ".norm1.weight"), G(p + ".norm1.bias"), p + ".1", cast=next_cast, slabs=(dn, 1))
            self._wgrad(dqkv, 3 * C, P[p + ".xn1"], C, 3 * C, C, M, G(p + ".attn.qkv.weight"), G(p + ".attn.qkv.bias"))
            if self._lagged_hook is not None:
                fn, self._lagged_hook = self._lagged_hook, None
                fn()
            return
        if self._fused_bwd(sp, B):
            # the whole data-gradient chain of the block in one launch (csrc/swin96.hip, csrc/swinw.hip); the weight
            # gradients and the folds of its per-workgroup partial rows run beside the chain exactly as for the unfused
            # sequence
            wide = not (self.fuse_block96_bwd and self._fusable96(sp))
            cb, cs, ct = next_cast if next_cast is not None else (None, None, tok)
            if cs is not None and ct != tok:
                raise ValueError("fused block backward: the cast scale must be per sample")
            R = ops.swinw_bwd_partial_rows(C, B, sp.H, sp.W) if wide else ops.swin96_bwd_partial_rows(B, sp.H, sp.W)
            ln1, ln2 = P.scratch("lnp." + p + ".1", R * 2 * C), P.scratch("lnp." + p + ".2", R * 2 * C)
            apart = P.scratch("apart." + p, R * nh * 256)
            xkw = {}
            if wide and self.split_wide_bwd and self._hgrad_wide(sp, B):
                nb = ops.swinw_split_bytes(C, B, sp.H, sp.W)       # two workgroups per window (tulip_swinw_block_bwd_split)
                if nb:
                    P.scratch("xchg." + p, (nb + 3) // 4)
                    xkw["exchange"] = P.bufs["xchg." + p]
            launch = (lambda **kw: ops.swinw_block_bwd(C, **xkw, **kw)) if wide else ops.swin96_block_bwd
            wt = W_.p16t if wide else W_.p16          # the wide kernel streams fragment-major TRANSPOSED weights
            lean = (not wide) and self._recomp96(sp)
            extra = dict(b_qkv=W_.p32(p + ".attn.qkv.bias"), b_fc1=W_.p32(p + ".mlp.fc1.bias"),
                         norm1_bias=W_.p32(p + ".norm1.bias"), norm2_bias=W_.p32(p + ".norm2.bias")) if lean else {}
            launch(
                dx=dx, x_in=xin, x1=P[p + ".x1"], qkv=None if lean else P[p + ".qkv"], fc1_pre=None if lean else P[p + ".h"],
                mean1=P[p + ".mean1"],
                rstd1=P[p + ".rstd1"], mean2=P[p + ".mean2"], rstd2=P[p + ".rstd2"],
                w_qkv=wt(p + ".attn.qkv.weight"), w_proj=wt(p + ".attn.proj.weight"),
                w_fc1=wt(p + ".mlp.fc1.weight"), w_fc2=wt(p + ".mlp.fc2.weight"),
                norm1_weight=W_.p32(p + ".norm1.weight"), norm2_weight=W_.p32(p + ".norm2.weight"),
                bias_table=W_.p32(p + ".attn.relative_position_bias_table"), rel_index=self._rel32,
                drop_scale_attn=self._ds(P, sp, 0), drop_scale_mlp=self._ds(P, sp, 1), d_out_mlp=dyb, d_fc1_pre=dh,
                d_out_attn=P[p + ".dyb_a"], d_qkv=dqkv, dx_bf16=cb, dx_bf16_scale=cs, norm1_partials=ln1,
                norm2_partials=ln2, bias_partials=apart, B=B, H=sp.H, W=sp.W, shift_h=sp.sft[0], shift_w=sp.sft[1],
                masked=self._mask_arg(sp, B), **extra)
            self._release_deferred()
            self._wgrad(dyb, C, P[p + ".g"], Hd, C, Hd, M, G(p + ".mlp.fc2.weight"), G(p + ".mlp.fc2.bias"))
            self._wgrad(dh, Hd, P[p + ".xn2"], C, Hd, C, M, G(p + ".mlp.fc1.weight"), G(p + ".mlp.fc1.bias"))
            self._wgrad(P[p + ".dyb_a"], C, P[p + ".o"], C, C, C, M, G(p + ".attn.proj.weight"), G(p + ".attn.proj.bias"))
            self._wgrad(dqkv, 3 * C, P[p + ".xn1"], C, 3 * C, C, M, G(p + ".attn.qkv.weight"), G(p + ".attn.qkv.bias"))
            self._fold(ln2, 2 * C, G(p + ".norm2.weight"), C, R)
            self._fold(ln2 + 4 * C, 2 * C, G(p + ".norm2.bias"), C, R)
            self._fold(ln1, 2 * C, G(p + ".norm1.weight"), C, R)
            self._fold(ln1 + 4 * C, 2 * C, G(p + ".norm1.bias"), C, R)
            self._fold_bias_table(P, p, apart, R, nh, G(p + ".attn.relative_position_bias_table"))
            if self._lagged_hook is not None:
                fn, self._lagged_hook = self._lagged_hook, None
                fn()
            return
        # ---- MLP branch (tulip.py:346-351)
        if not have_dyb:
            ops.cast_f32_bf16(dx, dyb, M, C, self._ds(P, sp, 1), tok)
        self._gemm(dyb, W_.p16(p + ".mlp.fc2.weight"), M, Hd, C, lda=C, ldb=Hd, b_trans=True, epi=EPI_GELU_BWD, out=dh,
                 ldo=Hd, aux=P[p + ".h"], ldaux=Hd)
        self._release_deferred()
        self._wgrad(dyb, C, P[p + ".g"], Hd, C, Hd, M, G(p + ".mlp.fc2.weight"), G(p + ".mlp.fc2.bias"))
        self._dgrad_ln_bwd(P, dh, W_.p16(p + ".mlp.fc1.weight"), M, C, Hd, dxn,
                           (P[p + ".x1"], P[p + ".mean2"], P[p + ".rstd2"], W_.p32(p + ".norm2.weight"), dx, dx, M, C,
                            G(p + ".norm2.weight"), G(p + ".norm2.bias"), p + ".2"),
                           dict(cast=(P[p + ".dyb_a"], self._ds(P, sp, 0), tok)))
        self._wgrad(dh, Hd, P[p + ".xn2"], C, Hd, C, M, G(p + ".mlp.fc1.weight"), G(p + ".mlp.fc1.bias"))
        if sp.prefix in self.early_flush:
            # the last blocks of the backward: nothing is left to hide their weight gradients behind, so the MLP
            # half starts as soon as its operands exist instead of at the end of the block
            self._flush_wgrads(advance=False)
        # ---- attention branch (tulip.py:339-344); its bf16 operand came out of the LayerNorm backward above
        dyb = P[p + ".dyb_a"]
        self._gemm(dyb, W_.p16(p + ".attn.proj.weight"), M, C, C, lda=C, ldb=C, b_trans=True, epi=EPI_BF16, out=dO,
                 ldo=C)
        self._wgrad(dyb, C, P[p + ".o"], C, C, C, M, G(p + ".attn.proj.weight"), G(p + ".attn.proj.bias"))
        R = ops.window_attn_bwd_partial_rows(B, sp.H, sp.W, nh, sp.win)
        apart = P.scratch("apart." + p, R * nh * 256)
        ops.window_attn_bwd(P[p + ".qkv"], dO, W_.p32(p + ".attn.relative_position_bias_table"), self._rel32, dqkv,
                            apart, B, sp.H, sp.W, C, nh, sp.win, sp.sft, self._mask_arg(sp))
        self._fold_bias_table(P, p, apart, R, nh, G(p + ".attn.relative_position_bias_table"))
        self._dgrad_ln_bwd(P, dqkv, W_.p16(p + ".attn.qkv.weight"), M, C, 3 * C, dxn,
                           (xin, P[p + ".mean1"], P[p + ".rstd1"], W_.p32(p + ".norm1.weight"), dx, dx, M, C,
                            G(p + ".norm1.weight"), G(p + ".norm1.bias"), p + ".1"), dict(cast=next_cast))
        self._wgrad(dqkv, 3 * C, P[p + ".xn1"], C, 3 * C, C, M, G(p + ".attn.qkv.weight"), G(p + ".attn.qkv.bias"))
        if self._lagged_hook is not None:
            fn, self._lagged_hook = self._lagged_hook, None
            fn()                                    # bucket join + all-reduce of the previous group, one block late

    def _stage_bwd(self, P: Plan, specs: List[BlockSpec], stage_in, dx, G, have_dyb=False, next_cast=None):
        """have_dyb: the last block's dyb_m was already produced by whoever produced dx;
        next_cast: consumer of the stage-input gradient (skip / merge cast), or None."""
        for k in reversed(range(len(specs))):
            xin = stage_in if k == 0 else P[specs[k - 1].prefix + ".out"]
            nc = self._mlp_cast(P, specs[k - 1]) if k > 0 else next_cast
            self._block_bwd(P, specs[k], xin, dx, G, have_dyb=have_dyb, next_cast=nc)
            have_dyb = True

    def _unmerge_bwd(self, P: Plan, prefix: str, s: int, dx_out, G, cast=None):
        """Backward of PatchUnmerging: lvl{s}.dz2 (the fine-level gradient, already un-shuffled to bf16 by the epilogue
        of the GEMM that produced it) -> dx_out (level s, fp32, overwritten).  cast = (bf16 buffer, DropPath scale,
        tokens per sample): the operand of the next GEMM on the chain, written by the same epilogue."""
        m, W_ = self.model, self.params
        B = P.B
        H, W, C = self.grid[0] >> s, self.grid[1] >> s, m.embed_dim << s
        M = B * H * W
        dz = P[f"lvl{s}.dz2"]
        cb, cs, ct = cast if cast is not None else (None, None, 1)
        gbias = G(prefix + ".expand.bias") if m.patch_unmerging else None
        if not m.patch_unmerging:
            # PatchExpanding: the fine-level gradient arrived in fine-token order (lvl{s}.dfine); LayerNorm backward
            # writes it into dz in the Linear's own output layout
            R = ops.expand_norm_bwd_partial_rows(B, H, W, 2)
            Cn = C // 2
            part = P.scratch("exp." + prefix, R * 3 * Cn)
            ops.expand_norm_bwd(P[f"lvl{s}.ey"], P[f"lvl{s}.emean"], P[f"lvl{s}.erstd"], W_.p32(prefix + ".norm.weight"),
                                dz, part, B, H, W, 2, Cn, dy_fine=P[f"lvl{s}.dfine"], ld=Cn)
            self._fold(part, 3 * Cn, G(prefix + ".norm.weight"), Cn, R)
            self._fold(part + 4 * Cn, 3 * Cn, G(prefix + ".norm.bias"), Cn, R)
        self._wgrad(dz, 2 * C, P[f"lvl{s}.xb"], C, 2 * C, C, M, G(prefix + ".expand.weight"), gbias)
        if s in self._glue_done:
            # the skip Linear's data gradient (unmerged half) and this data gradient in ONE launch (csrc/glue.hip), issued HERE --
            # behind the decoder stage's hook, where the expand data-gradient GEMM used to be
            ops.skip_unmerge_bwd(**self._glue_done.pop(s))
            self._release_deferred()
            return
        self._gemm(dz, W_.p16(prefix + ".expand.weight"), M, C, 2 * C, lda=2 * C, ldb=C, b_trans=True, epi=EPI_F32,
                 out=dx_out, ldo=C, out2=cb, ldo2=C if cb is not None else 0, rowscale=cs, rows_per_sample=ct,
                 packed=self._pk(prefix + ".expand.weight", transposed=True))
        self._release_deferred()

    # Gradients are WRITTEN, not accumulated (run_backward(overwrite=True), the Trainer with accum_iter == 1): every parameter has
    # exactly one producer per backward (a weight-gradient item or a fold region), so the flat gradient buffer needs no clearing
    # -- AdamW does not write 4 B per parameter of zeros, and the un-split weight gradients of the deep stages (66 MB) are stored
    # instead of read-modify-written.  (The stand-alone LayerNorm parameter pass for C > 2048 adds: its two ranges are cleared first.)
    grad_overwrite = False
    # The optimizer step of un-split weight gradients in their write-out (Trainer.fuse_adamw): adam_ctx = ops.adamw_ref of the
    # Trainer's flat buffers, adam_fused = the gradient addresses it applies to, adam_apply = this backward is an optimizer
    # step; adam_probe (a dict: gradient address -> elements) collects the eligible ranges during the Trainer's eager warm-up pass.
    adam_ctx = None
    adam_fused = frozenset()
    adam_apply = False
    adam_probe = None
    _gflat = None
    fuse_adamw_folds = True      # (False: only the un-split write-outs step)

    def overwrite_supported(self, B: int) -> bool:
        return bool(self.group_wgrad and self.overlap_wgrad)

    def run_backward(self, P: Plan, gflat: torch.Tensor, gscale_dev=None, gscale: float = 1.0, bucket_hook=None,
                     join_tags=None, overwrite: bool = False, apply_adamw: bool = False, pack_at_end: bool = False,
                     bucket_on_side: bool = False):
        """Parameter gradients of P.losses[0] accumulated (+=) into the flat fp32 buffer `gflat`
        (same layout as the parameters).  bucket_hook(name) is called after the last gradient of
        each parameter group has been *launched* (DDP overlap)."""
        m, W_ = self.model, self.params
        B, E, nl = P.B, m.embed_dim, m.num_layers
        H0, W0 = self.grid
        self._pending, self._lagged_hook, self._deferred, self._carry = [], None, None, ()   # nothing survives an aborted call
        self._detached = None
        self._glue_done = {}
        self._pack_ev = None
        self.grad_overwrite = bool(overwrite)
        self.pack_at_end = bool(pack_at_end)
        self._gflat = gflat
        self.adam_apply = bool(apply_adamw and overwrite and self.adam_ctx is not None)
        if self._loss_final is not None:             # the loss read-out of run_forward(defer_loss_final=True): off the chain
            self._side(self._loss_final)
            self._loss_final = None
        gbase = gflat.data_ptr()
        G = lambda name: gbase + 4 * W_.offset[name]
        user_hook = bucket_hook or (lambda tag: None)

        def join_and_fire(tag):
            self._wait_side()
            user_hook(tag)

        def hook(tag):
            # a bucket is complete only once its side-stream weight gradients are in; without buckets the
            # side streams are joined once, at the end
            if self._lagged_hook is not None:           # two hooks without a block in between
                fn, self._lagged_hook = self._lagged_hook, None
                fn()
            bucket = join_tags is not None and tag in join_tags
            if bucket and bucket_on_side and tag != "embed" and self.overlap_wgrad:
                # round 6 (Trainer.graph_collectives: the whole N > 1 step is ONE captured graph): the bucket's hook -- its all-reduce --
                # is issued ON THE SIDE STREAM behind the bucket's last side group (the closure runs last in the group's flush), so
                # the collective becomes a branch of the graph off the side queue and the chain never waits for it: no cut, no join
                def fire(tag=tag):
                    if self._carry:          # the scatter regions this group left "for the next side launch" belong to this bucket
                        carry, self._carry = list(self._carry), ()
                        ops.reduce_rows_multi(carry, adam=self._adam_arg())
                    user_hook(tag)
                self._pending.append(("f", fire))
                self._flush_wgrads(mark=self.pack_at_end and self.adam_apply and tag == self._pack_mark_tag)
                return
            if bucket and self.detach_buckets and tag != "embed" and self.overlap_wgrad and self._pending:
                # The bucket's LAST side group is not forked inside this capture at all: the caller (Trainer._capture) ends the
                # chain's graph segment here, captures the group as a graph of its own (issue_detached) that is replayed on
                # another stream behind the segment, with the bucket's all-reduce behind it -- and the chain's next segment
                # starts at once instead of waiting for the group at the cut.  Older groups of the bucket were forked inside
                # the segment and are joined here (they ran beside the bucket's last blocks).
                self._release_deferred()
                self._detached, self._pending = self._pending, []
                self._wait_side()
                user_hook(tag)
                return
            self._flush_wgrads(mark=self.pack_at_end and self.adam_apply and tag == self._pack_mark_tag)
            if tag == "embed":
                join_and_fire(tag)
            elif bucket:
                # joining here would stall the chain behind the side work that was forked a moment ago; the join
                # (and with it the bucket's all-reduce) moves to the end of the NEXT block's chain work, before that
                # block's own side work is issued (see _block_bwd)
                self._lagged_hook = lambda: join_and_fire(tag)
            else:
                user_hook(tag)

        M0 = B * H0 * W0
        gdw = G("decoder_pred.weight")
        if m.pixel_shuffle:
            tpart = P["tail.dwd_part"]
            head_w, head_b = "ps_head.conv_expand.0.weight", "ps_head.conv_expand.0.bias"
            targs = (P["tail.xn"], W_.p16(head_w), W_.p32(head_b), W_.p32("decoder_pred.weight"), P.pred)
            tkw = dict(target=P.target, gscale_dev=gscale_dev, gscale=gscale)   # L1 backward (tulip.py:692-693) formed in-kernel
            self._tail_fused = self.fuse_tail_bwd and ops.tail_fused_bwd_supported(E)
            if self._tail_fused:
                # d(expand pre-activation) -- 100 MB at batch 8 -- is never written: the chain's kernel goes straight to dxn,
                # the side queue's kernel recomputes it channel-sliced for the expand conv's weight / bias gradient
                x_last = P[self.dec_blocks[-1][-1].prefix + ".out"] if nl > 1 else P[self.enc_blocks[0][-1].prefix + ".out"]
                dx0 = P["dec0.dx"] if nl > 1 else P["enc0.dx"]
                if self.fuse_tail_ln_bwd:
                    # ... and on through norm_up's backward in the same launch (dx, the next GEMM's bf16 operand, one
                    # [dgamma | dbeta] partial row per 32 tokens)
                    cb, cs, ct = self._mlp_cast(P, (self.dec_blocks[-1] if nl > 1 else self.enc_blocks[0])[-1]) or (None, None, 1)
                    R = (M0 + 31) // 32
                    lnp = P.scratch("lnp.norm_up.fused", R * 2 * E)
                    ops.tail_bwd_dgrad_ln(*targs, tpart, B, H0, W0, E, x_last, P["tail.mean"], P["tail.rstd"],
                                          W_.p32("norm_up.weight"), dx0, lnp, dx_bf16=cb, cast_rowscale=cs,
                                          cast_rows_per_sample=ct, **tkw)
                    self._fold(lnp, 2 * E, G("norm_up.weight"), E, R)
                    self._fold(lnp + 4 * E, 2 * E, G("norm_up.bias"), E, R)
                else:
                    ops.tail_bwd_dgrad(*targs, P["tail.dxn"], tpart, B, H0, W0, E, **tkw)
                sp = ops.tail_wgrad_splits(B, H0, W0, E)
                nw = 16 * E * E
                slab = P.scratch("tail.wslab", sp * (nw + 16 * E))
                self._side_first(lambda: ops.tail_wgrad(*targs, slab, slab + 4 * sp * nw, B, H0, W0, E, **tkw))
                self._fold(slab, nw, G(head_w), nw, sp)
                self._fold(slab + 4 * sp * nw, 16 * E, G(head_b), 16 * E, sp)
            else:
                ops.tail_bwd(*targs, P["tail.dz"], tpart, B, H0, W0, E, **tkw)
                self._wgrad(P["tail.dz"], 16 * E, P["tail.xn"], E, 16 * E, E, M0, G(head_w), G(head_b))
            self._fold(tpart, 128, gdw, E, (M0 + 31) // 32)
        else:
            # FinalPatchExpanding backward: d(pred) (L1, tulip.py:692-693) -> decoder_pred / LayerNorm backward per fine
            # token -> tail.dz = d(Linear output) in the Linear's layout
            r = m.upscale_factor
            pre = "final_patch_expanding"
            ops.l1_loss_bwd(P.pred, P.target, gscale_dev, gscale, P.dpred, P.pred.numel())
            R = ops.expand_norm_bwd_partial_rows(B, H0, W0, r)
            part = P.scratch("exp.final", R * 3 * E)
            ops.expand_norm_bwd(P["tail.ey"], P["tail.emean"], P["tail.erstd"], W_.p32(pre + ".norm.weight"), P["tail.dz"],
                                part, B, H0, W0, r, E, dpred=P.dpred, dotw=W_.p32("decoder_pred.weight"),
                                beta=W_.p32(pre + ".norm.bias"))
            self._fold(part, 3 * E, G(pre + ".norm.weight"), E, R)
            self._fold(part + 4 * E, 3 * E, G(pre + ".norm.bias"), E, R)
            self._fold(part + 8 * E, 3 * E, gdw, E, R)
            head_w = pre + ".expand.weight"
            self._wgrad(P["tail.dz"], 16 * E, P["tail.xn"], E, 16 * E, E, M0, G(head_w))
        if not (m.pixel_shuffle and self._tail_fused):
            self._gemm(P["tail.dz"], W_.p16(head_w), M0, E, 16 * E, lda=16 * E, ldb=E, b_trans=True,
                     epi=EPI_BF16, out=P["tail.dxn"], ldo=E)
        x_last = P[self.dec_blocks[-1][-1].prefix + ".out"] if nl > 1 else P[self.enc_blocks[0][-1].prefix + ".out"]
        dx = P["dec0.dx"] if nl > 1 else P["enc0.dx"]
        if not (m.pixel_shuffle and self._tail_fused and self.fuse_tail_ln_bwd):
            self._ln_bwd(P, P["tail.dxn"], x_last, P["tail.mean"], P["tail.rstd"], W_.p32("norm_up.weight"), None, dx, M0,
                         E, G("norm_up.weight"), G("norm_up.bias"), "norm_up",
                         cast=self._mlp_cast(P, (self.dec_blocks[-1] if nl > 1 else self.enc_blocks[0])[-1]))
        hook("head")
        # ---- decoder, fine -> coarse
        for i in reversed(range(nl - 1)):
            s = nl - i - 2
            Cs = E << s
            Ms = B * (H0 >> s) * (W0 >> s)
            dx = P[f"dec{s}.dx"]
            if i < nl - 2:
                # dx currently holds nothing for this level: pull the grad down from the finer level
                self._unmerge_bwd(P, f"layers_up.{i}.upsample", s, dx, G,
                                  cast=self._mlp_cast(P, self.dec_blocks[i][-1]))
            dys = P[f"dec{s}.dyskip"]
            self._stage_bwd(P, self.dec_blocks[i], P[f"dec{s}.in"], dx, G, have_dyb=True,
                            next_cast=(dys, None, 1))          # the skip Linear's dgrad/wgrad operand
            pre = f"skip_connection_layers.{i}"
            self._wgrad(dys, Cs, P[f"dec{s}.cat"], 2 * Cs, Cs, 2 * Cs, Ms, G(pre + ".weight"), G(pre + ".bias"))
            # grad w.r.t. the first concat half (the unmerged stream), un-shuffled to the coarser level's layout in bf16 by
            # the epilogue: it is the operand of that level's PatchUnmerging backward; the x_save half is deferred
            if self._glue_unmerge(2 * Cs, B, H0 >> (s + 1), W0 >> (s + 1)) and self._has_copies(
                    pre + ".weight", ("first_patch_expanding" if i == 0 else f"layers_up.{i - 1}.upsample") + ".expand.weight"):
                # ... and on through the PatchUnmerging's data gradient in the same launch (csrc/glue.hip): lvl{s+1}.dz2 is still
                # written (operand of the expand weight / bias gradient, queued by _unmerge_bwd below / at the next level)
                up_blocks = self.dec_blocks[i - 1] if i > 0 else self.enc_blocks[nl - 1]
                cb, cs, ct = self._mlp_cast(P, up_blocks[-1]) or (None, None, 1)
                self._glue_done[s + 1] = dict(
                    dy_skip=dys, w_skip_t_packed=W_.p16t(pre + ".weight"), dz=P[f"lvl{s + 1}.dz2"],
                    w_expand_t_packed=W_.p16t(("first_patch_expanding" if i == 0 else f"layers_up.{i - 1}.upsample") + ".expand.weight"),
                    dx=P[f"dec{s + 1}.dx"] if i > 0 else P[f"enc{nl - 1}.dx"], dx_bf16=cb, cast_rowscale=cs,
                    cast_rows_per_sample=ct, B=B, H=H0 >> (s + 1), W=W0 >> (s + 1), C=2 * Cs)
            elif m.patch_unmerging:
                self._gemm(dys, W_.p16(pre + ".weight"), Ms, Cs, Cs, lda=Cs, ldb=2 * Cs, b_trans=True,
                           epi=EPI_UNSHUF2_BF16, out=P[f"lvl{s + 1}.dz2"], ldo=4 * Cs, psH=H0 >> (s + 1), psW=W0 >> (s + 1),
                           packed=self._pk(pre + ".weight", transposed=True))          # rows 0 .. Cs-1 of W^T: the unmerged half
            else:       # PatchExpanding: fine-token order; its LayerNorm backward does the un-rearrange (_unmerge_bwd)
                self._gemm(dys, W_.p16(pre + ".weight"), Ms, Cs, Cs, lda=Cs, ldb=2 * Cs, b_trans=True, epi=EPI_BF16,
                           out=P[f"lvl{s + 1}.dfine"], ldo=Cs)
            hook(f"dec{i}")
        # ---- bottleneck unmerge
        dx = P[f"enc{nl - 1}.dx"]
        if nl > 1:
            self._unmerge_bwd(P, "first_patch_expanding", nl - 1, dx, G,
                              cast=self._mlp_cast(P, self.enc_blocks[nl - 1][-1]))
        # ---- encoder, coarse -> fine
        for s in reversed(range(nl)):
            dx = P[f"enc{s}.dx"]
            bottom_to_merge = (s == nl - 1 and s > 0)
            self._stage_bwd(P, self.enc_blocks[s], P[f"enc{s}.in"], dx, G, have_dyb=(nl > 1),
                            next_cast=(P[f"enc{s}.dyb"], None, 1) if bottom_to_merge else None)
            if s > 0 and self._glue_merge_bwd(E << (s - 1), B, H0 >> (s - 1), W0 >> (s - 1)) and self._has_copies(
                    f"layers.{s - 1}.downsample.reduction.weight", *([f"skip_connection_layers.{nl - s - 2}.weight"] if s < nl - 1 else [])):
                # [x_save half of the skip Linear's input gradient ->] PatchMerging reduction data gradient -> LayerNorm backward with
                # the 2x2 scatter, ONE launch (csrc/glue.hip) instead of three (two at the bottleneck)
                Cp, Cs = E << (s - 1), E << s
                Hp, Wp = H0 >> (s - 1), W0 >> (s - 1)
                rows = B * (Hp // 2) * (Wp // 2)
                pre = f"layers.{s - 1}.downsample"
                has_skip = s < nl - 1
                R = ops.merge_bwd_partial_rows(Cp, B, Hp, Wp)
                part = P.scratch("lnp." + pre + ".glue", R * 8 * Cp)
                cb, cs, ct = self._mlp_cast(P, self.enc_blocks[s - 1][-1]) or (None, None, 1)
                ops.merge_bwd(dx_in=dx if has_skip else None, dy_skip=P[f"dec{s}.dyskip"] if has_skip else None,
                              w_skip_t_packed=W_.p16t(f"skip_connection_layers.{nl - s - 2}.weight") if has_skip else None,
                              dyb=P[f"enc{s}.dyb"], w_red_t_packed=W_.p16t(pre + ".reduction.weight"),
                              x_prev=P[self.enc_blocks[s - 1][-1].prefix + ".out"], mean=P[f"enc{s - 1}.mmean"],
                              rstd=P[f"enc{s - 1}.mrstd"], gamma=W_.p32(pre + ".norm.weight"), dx_prev=P[f"enc{s - 1}.dx"],
                              param_partials=part, dx_bf16=cb, cast_rowscale=cs, cast_rows_per_sample=ct, B=B, H=Hp, W=Wp, Cp=Cp)
                self._wgrad(P[f"enc{s}.dyb"], 2 * Cp, P[f"enc{s - 1}.xm"], 4 * Cp, 2 * Cp, 4 * Cp, rows, G(pre + ".reduction.weight"))
                self._fold(part, 8 * Cp, G(pre + ".norm.weight"), 4 * Cp, R)
                self._fold(part + 16 * Cp, 8 * Cp, G(pre + ".norm.bias"), 4 * Cp, R)
                hook(f"enc{s}")
                continue
            if s < nl - 1:
                # deferred skip-connection gradient w.r.t. x_save[s] (second concat half, tulip.py:715)
                Cs = E << s
                Ms = B * (H0 >> s) * (W0 >> s)
                i = nl - s - 2
                # (for 0 < s the sum is also what the PatchMerging backward below consumes: bf16 copy from the epilogue)
                self._gemm(P[f"dec{s}.dyskip"], W_.p16(f"skip_connection_layers.{i}.weight") + 2 * Cs, Ms, Cs, Cs,
                         lda=Cs, ldb=2 * Cs, b_trans=True, epi=EPI_F32, out=dx, ldo=Cs, accumulate=True,
                         out2=P[f"enc{s}.dyb"] if s > 0 else None, ldo2=Cs if s > 0 else 0,
                         packed=self._pk(f"skip_connection_layers.{i}.weight", transposed=True, row0=Cs, K=Cs))    # rows Cs .. of W^T
            if s > 0:
                # PatchMerging backward of level s-1
                Cp = E << (s - 1)
                Hp, Wp = H0 >> (s - 1), W0 >> (s - 1)
                rows = B * (Hp // 2) * (Wp // 2)
                pre = f"layers.{s - 1}.downsample"
                dyb, dxm = P[f"enc{s}.dyb"], P["t.dxm"]
                self._wgrad(dyb, 2 * Cp, P[f"enc{s - 1}.xm"], 4 * Cp, 2 * Cp, 4 * Cp, rows,
                            G(pre + ".reduction.weight"))
                self._gemm(dyb, W_.p16(pre + ".reduction.weight"), rows, 4 * Cp, 2 * Cp, lda=2 * Cp, ldb=4 * Cp,
                         b_trans=True, epi=EPI_BF16, out=dxm, ldo=4 * Cp, packed=self._pk(pre + ".reduction.weight", transposed=True))
                xprev = P[self.enc_blocks[s - 1][-1].prefix + ".out"]
                self._ln_bwd(P, dxm, xprev, P[f"enc{s - 1}.mmean"], P[f"enc{s - 1}.mrstd"], W_.p32(pre + ".norm.weight"),
                             None, P[f"enc{s - 1}.dx"], rows, 4 * Cp, G(pre + ".norm.weight"), G(pre + ".norm.bias"),
                             pre, merge=True, H=Hp, W=Wp, cast=self._mlp_cast(P, self.enc_blocks[s - 1][-1]))
            if s == 0:
                # this stage's weight gradients are what the step waits for behind the chain's last kernels: one full round
                # of the chip (the tile-count map trades duration for slab traffic, which only pays while the chain runs)
                self._pending.append(("c", None))
            hook(f"enc{s}")
        kw = 8 if m.circular_padding else m.patch_size[1]
        # patch-embed parameter gradients: partial rows laid out like the flat gradient slice
        # [proj.weight | proj.bias | norm.weight | norm.bias] so ONE reduction folds all four tensors
        o0 = W_.offset["patch_embed.proj.weight"]
        rel = lambda n: 4 * (W_.offset[n] - o0)
        ep = P.embed_part_ptr
        ops.patch_embed_bwd(P.x_in, W_.p32("patch_embed.proj.weight"), W_.p32("patch_embed.proj.bias"),
                            W_.p32("patch_embed.norm.weight"), P["enc0.dx"], ep, ep + rel("patch_embed.proj.bias"),
                            ep + rel("patch_embed.norm.weight"), ep + rel("patch_embed.norm.bias"), B, m.in_chans,
                            m.img_size[0], m.img_size[1], E, m.patch_size[0], m.patch_size[1], kw, m.circular_padding,
                            self.eps, partial_stride=P.embed_stride)
        gpe, nbe = G("patch_embed.proj.weight"), ops.patch_embed_bwd_blocks(B * H0 * W0)
        embed_on_chain = self.embed_fold_on_chain and self.pack_at_end and self.adam_apply and self.overlap_wgrad
        if embed_on_chain:
            # round 6: the patch-embedding fold (+ its optimizer step) on the CHAIN's queue, in front of the weight-copy refresh: queued
            # for the side stream it was a launch group of its own behind the backward's last weight-gradient group AND behind the
            # refresh (its fork event sits behind the refresh's launches) -- ~20 us at the very end of the step with nothing beside it
            kw = dict(overwrite=self.grad_overwrite)
            if kw["overwrite"] and self.fuse_adamw_folds and self._in_gflat(gpe):
                if self.adam_probe is not None:
                    self.adam_probe[gpe] = max(P.embed_stride, self.adam_probe.get(gpe, 0))
                kw["adamw"] = self.adam_apply and gpe in self.adam_fused
            ops.reduce_rows_multi([ops.reduce_region(ep, P.embed_stride, gpe, P.embed_stride, nbe, **kw)], adam=self._adam_arg())
        if self.pack_at_end and self.adam_apply and self._pack_ev is None:
            # the marked group has not been enqueued (its fork is still deferred: the marked tag is this last stage, e.g. a model
            # whose stage 0 already has a packed width) or no hook carried the mark: enqueue what is deferred -- the chain's next
            # kernel, patch_embed_bwd, exists by now -- and order the refresh behind everything on the side queue
            self._mark_side()
        if self._pack_ev is not None:
            # pack_at_end: the chain has nothing left to do but wait for the side queue's last groups -- the fragment-major copies of
            # every wide / deep weight (all stepped by now: the marked group was the last that holds any) are rewritten HERE, on the
            # chain's queue, instead of beside the next forward (three forks and three joins there)
            torch.cuda.current_stream().wait_event(self._pack_ev)
            W_.refresh_transposes(late=False if W_.pk_late else None)     # (the late names: behind the end-of-step AdamW, Trainer._adamw)
            self._pack_ev = None
        if not embed_on_chain:
            self._fold(ep, P.embed_stride, gpe, P.embed_stride, nbe)
        hook("embed")

    # ------------------------------------------------------------------ autograd bridge
    # The reference's calling convention -- model(lo, hi) under autocast, loss.backward(), a foreign optimizer
    # (engine_upsampling.py:77-80, misc.py:295) -- issues the same ~115 launches per direction from Python through ctypes:
    # ~3 ms of host time per step against 2 ms of GPU work.  The two launch sequences are fixed per (batch size, train / eval),
    # so the module path replays them from HIP graphs as well: first call eager (loads kernels, sizes lazy buffers), second call
    # captures, later calls replay.  TULIP_GRAPH_MODULE=0: eager launches every time.
    graph_module = knobs.on("TULIP_GRAPH_MODULE", True)

    def _module_sequence(self, P: Plan, key, fn):
        graphs = P.__dict__.setdefault("_module_graphs", {})
        # what the captured launch sequence depends on besides the caller's key: the fuse switches, the DropPath seed (a launch
        # argument) and the number of draw slots
        key = key + (self.fuse_wide, self.fuse_wide_bwd, self.fuse_block96, self.fuse_block96_bwd, self.split_wide, self.split_wide_bwd,
                     self.fc1_grad_wide, self.fc1_grad96, self.recompute96, self.fuse_deep, self.fuse_tail_fwd, self.fuse_tail_bwd, self.fuse_glue, self.pair96, self.packed_gemm,
                     int(self._drop_seed), int(self.n_drop_slots))
        ent = graphs.get(key)
        if not self.graph_module or ent is None:
            fn()
            graphs[key] = "warm"
            return
        if ent == "warm":
            cur = torch.cuda.current_stream()
            cap = torch.cuda.Stream(device=self.device)
            cap.wait_stream(cur)
            with torch.cuda.stream(cap):
                ent = torch.cuda.CUDAGraph()
                try:
                    ent.capture_begin(capture_error_mode="thread_local")
                    fn()
                    ent.capture_end()
                except BaseException:
                    P.reset_exchange()
                    raise
            cur.wait_stream(cap)
            graphs[key] = ent
        ent.replay()

    def _module_forward(self, P: Plan):
        train = bool(self.model.training)

        def seq():
            self.params.shadow_dirty = True      # a foreign optimizer may have written the parameters: the cast + pack are part
            self.draw_drop_scales(P, train)      # of the sequence (and of its graph)
            self.run_forward(P)
        self._module_sequence(P, ("fwd", train, self.attn_fp8), seq)
        # the caller's optimizer owns the weights on this path and steps them AFTER this call: whoever reads the bf16 shadow /
        # packed copies next (a Trainer, a GraphedForward, the next module forward) must rebuild them
        self.params.shadow_dirty = True
        P.generation += 1

    def _module_backward(self, P: Plan, dloss):
        """The flat gradient of one module backward, in a buffer autograd may keep: `.grad` of a parameter whose gradient was None
        becomes (a view of) what this returns, so a buffer that a captured graph writes may be handed out only while nothing
        else refers to it.  Two such buffers alternate -- with `optimizer.zero_grad()` between steps (set_to_none, the reference's
        loop, engine_upsampling.py:98-99) the one handed out two calls ago is free again, and during gradient accumulation `.grad`
        keeps holding the first one while every later call writes the second; a buffer is free when its storage has no user but
        the plan (one reference count read, no walk over the parameters).  Both taken (the caller kept gradients of two calls
        alive): a third, private buffer and a copy out of it -- what every call paid before (108 MB copied per step)."""
        W_ = self.params
        if not hasattr(P, "_mod_gbufs"):
            P._mod_gbufs = [torch.zeros(W_.total, dtype=torch.float32, device=self.device) for _ in range(2)]
            P._mod_gscale = torch.ones(1, dtype=torch.float32, device=self.device)
            P._mod_gidle = _storage_users(P._mod_gbufs[0])
            if P._mod_gidle >= (1 << 30) or _storage_users(P._mod_gbufs[1]) != P._mod_gidle:
                P._mod_gidle = -1                      # no trustworthy count: every call takes the private buffer + copy
        if dloss is None:
            P._mod_gscale.fill_(1.0)
        else:
            P._mod_gscale.copy_(dloss.detach().reshape(1))
        # (EXACTLY the idle count: any other value -- a live .grad view, or a torch build that counts the temporary storage wrapper
        # differently from the baseline call -- reads as "taken" and falls to the private buffer + copy)
        which = next((i for i, b in enumerate(P._mod_gbufs[:2]) if _storage_users(b) == P._mod_gidle), None)
        if which is None:
            if len(P._mod_gbufs) == 2:
                P._mod_gbufs.append(torch.zeros(W_.total, dtype=torch.float32, device=self.device))
            which = 2
        g = P._mod_gbufs[which]
        over = self.overwrite_supported(P.B)      # every gradient element has one producer: nothing to clear between calls
        if not over:
            g.zero_()
        self._module_sequence(P, ("bwd", self.attn_fp8, which),
                              lambda: self.run_backward(P, g, gscale_dev=P._mod_gscale, gscale=1.0, overwrite=over))
        return g.clone() if which == 2 else g

    def autograd_forward(self, x, target, mc_drop: bool):
        self.bind(x.device)
        self.params.shadow_dirty = True  # parameters may have been updated by a foreign optimizer
        B = x.shape[0]
        P = self.plan(B)
        P.x_in.copy_(x.reshape(P.x_in.shape).float())
        if mc_drop or target is None:
            self.draw_drop_scales(P, self.model.training)
            self.run_forward(P, with_loss=False)
            self.params.shadow_dirty = True
            return P.pred.clone()
        P.target.copy_(target.reshape(P.target.shape).float())
        params = self._cur_params if self._cur_params is not None else self.params.current_params(self.model)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if not need_grad:
            self.draw_drop_scales(P, self.model.training)
            self.run_forward(P)
            self.params.shadow_dirty = True
            return P.pred.clone(), P.losses[0].clone(), P.losses[1].clone()
        return _TulipFn.apply(self, P, *params)            # (flat-buffer order: _TulipFn.backward returns the gradients in it)


def _storage_users(t: torch.Tensor) -> int:
    """References to the tensor's storage (torch's own count, as its CUDA-graph trees use it).  Without that private entry point a
    huge count is reported: every buffer then looks taken and _module_backward falls back to its private buffer + copy."""
    f = getattr(torch._C, "_storage_Use_Count", None)
    return f(t.untyped_storage()._cdata) if f is not None else 1 << 30


class _TulipFn(torch.autograd.Function):
    """loss.backward() support: one autograd node for the whole network."""

    @staticmethod
    def forward(ctx, eng: TulipEngine, P: Plan, *params):
        eng._module_forward(P)
        ctx.eng, ctx.P, ctx.gen = eng, P, P.generation
        pred, pixel = P.pred.clone(), P.losses[1].clone()
        # only total_loss carries a gradient (the reference back-propagates total_loss alone, misc.py:295); anything
        # else would be silently dropped by backward() below, so autograd is told to refuse it
        ctx.mark_non_differentiable(pred, pixel)
        return pred, P.losses[0].clone(), pixel

    @staticmethod
    def backward(ctx, _dpred, dloss, _dpixel):
        eng, P = ctx.eng, ctx.P
        if ctx.gen != P.generation:
            raise RuntimeError("tulip_amd: backward() must follow its own forward() (activation workspaces are "
                               "reused by the next forward of the same batch size)")
        W_ = eng.params
        gflat = eng._module_backward(P, dloss)
        grads = tuple(gflat[W_.offset[n]:W_.offset[n] + W_.numel[n]].view(W_.shape[n]) for n in W_.names)
        return (None, None) + grads
