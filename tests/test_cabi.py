"""CPU: the C-ABI library builds, loads, and exports exactly what include/tulip_hip.h declares
(no compute calls -- there is no GPU here)."""
import os
import re

from tulip_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "tulip_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(tulip_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else args.count(",") + 1
        out[name] = n
    return out


def test_library_builds_and_loads():
    from tulip_amd.csrc.build import build
    path = build(force=False, verbose=False)
    assert os.path.exists(path)
    lib = _lib.load()
    assert lib.tulip_abi_version() == _lib.ABI_VERSION == 6
    assert lib.tulip_build_arch() == b"gfx950"


def test_every_declared_symbol_is_exported_and_bound():
    decl = _header_functions()
    assert len(decl) >= 25
    lib = _lib.load()
    for name, nargs in decl.items():
        assert hasattr(lib, name), f"{name} declared in tulip_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
        assert len(_lib.SIGNATURES[name]) == nargs, (name, len(_lib.SIGNATURES[name]), nargs)
    for name in _lib.SIGNATURES:
        assert name in decl, f"{name} bound in _lib.py but not declared in the header"


def test_pure_host_entry_points():
    lib = _lib.load()
    # K is cut in multiples of 32; the reported split count is what the kernel launches
    assert lib.tulip_gemm_effective_splits(32768, 128) == 128
    assert lib.tulip_gemm_effective_splits(100, 3) == 2
    assert lib.tulip_gemm_effective_splits(512, 1) == 1
    assert lib.tulip_layernorm_bwd_partial_rows(32768, 96) == 512
    assert lib.tulip_layernorm_bwd_partial_rows(64, 1536) == 16
    assert lib.tulip_layernorm_bwd_partial_rows(16, 6144) == 0
    assert lib.tulip_patch_embed_bwd_blocks(32768) == 512
    # the split form of the fused C = 384 block exists where a workgroup owns one window: 2 x 2 x 768 lanes x 16 B + a ticket per window
    assert lib.tulip_swinw_split_bytes(384, 8, 4, 64) == 128 * (2 * 2 * 768 * 16 + 4)
    assert lib.tulip_swinw_split_bytes(384, 16, 4, 64) == 0 and lib.tulip_swinw_split_bytes(192, 8, 8, 128) == 0
    r = lib.tulip_window_attn_bwd_partial_rows(8, 16, 256, 3, 2, 8)
    assert 1 <= r <= 2048 // 3 + 1


def test_the_two_builds_export_one_symbol_set():
    """libtulip_hip.so ships what a step launches; libtulip_hip_dev.so (-DTULIP_DEV_VARIANTS=1) additionally the built-tested-off
    kernel forms and the profiled twins.  Same header, same symbols; the product library answers the
    development-only calls with TULIP_ERR_NOT_BUILT (no launch, so this is checked here without a GPU)."""
    import ctypes
    lib = _lib.load()
    assert lib.tulip_dev_variants() == 0
    with _lib.dev_library() as dev:
        assert dev.tulip_dev_variants() == 1 and dev.tulip_abi_version() == _lib.ABI_VERSION
        for name in _header_functions():
            assert hasattr(dev, name), name
        assert _lib.load() is dev
    assert _lib.load() is lib
    assert os.path.getsize(_lib.DEV_LIB_PATH) > os.path.getsize(_lib.LIB_PATH)
    it = (_lib.WgradItem * 1)()
    assert lib.tulip_wgrad_group_profiled(it, 0, None, 0, 0, None, None) == -3          # TULIP_ERR_NOT_BUILT


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.TulipHipError as e:
        assert "no CPU or PyTorch fallback" in str(e)
    else:
        raise AssertionError("loading a missing library must raise")


def test_struct_layouts_match_the_header(tmp_path):
    """The descriptor structs passed by pointer (tulip_swin96_desc, tulip_swin96_bwd_desc, tulip_reduce_region,
    tulip_wgrad_item): size and every field offset of the ctypes mirror against what a C compiler makes of the header."""
    import ctypes
    import subprocess
    structs = {"tulip_swin96_desc": _lib.Swin96Desc, "tulip_swin96_bwd_desc": _lib.Swin96BwdDesc,
               "tulip_reduce_region": _lib.ReduceRegion, "tulip_wgrad_item": _lib.WgradItem,
               "tulip_merge_fwd_desc": _lib.MergeFwdDesc, "tulip_merge_bwd_desc": _lib.MergeBwdDesc,
               "tulip_unmerge_skip_desc": _lib.UnmergeSkipDesc, "tulip_skip_unmerge_bwd_desc": _lib.SkipUnmergeBwdDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "tulip_hip.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _t in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _t in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)
    # and the compile-time limits the Python side mirrors
    hdr = open(os.path.join(ROOT, "include", "tulip_hip.h")).read()
    assert int(re.search(r"#define TULIP_REDUCE_REGIONS_MAX (\d+)", hdr).group(1)) == _lib.REDUCE_REGIONS_MAX
    assert int(re.search(r"#define TULIP_WGRAD_GROUP_MAX (\d+)", hdr).group(1)) == _lib.WGRAD_GROUP_MAX
