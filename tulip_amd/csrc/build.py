"""Build libtulip_hip.so (gfx950) in-tree with hipcc.  `python -m tulip_amd.csrc.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
SOURCES = ["gemm.hip", "norm.hip", "attention.hip", "elementwise.hip", "tail.hip", "prep.hip", "evalpost.hip", "swin96.hip", "expand.hip", "swinw.hip", "swind.hip", "glue.hip"]
LIB = os.path.join(PKG, "libtulip_hip.so")
ARCH = "gfx950"


def _stale(LIB: str = LIB) -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(HERE, "common.h"), os.path.join(HERE, "swin_stream.h"),
                                                       os.path.join(ROOT, "include", "tulip_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


ASAN_LIB = os.path.join(PKG, "libtulip_hip_asan.so")
DEV_LIB = os.path.join(PKG, "libtulip_hip_dev.so")      # -DTULIP_DEV_VARIANTS=1 (include/tulip_hip.h, conventions)


def build(force: bool = False, verbose: bool = True, asan: bool = False, dev: bool = True) -> str:
    """asan=True (`--asan`): the AddressSanitizer build, libtulip_hip_asan.so -- host code AND kernels instrumented
    (`-fsanitize=address -shared-libsan`, device side needs the xnack+ target; INTEGRATION.md, "Sanitizer build")."""
    if asan:
        return _build(ASAN_LIB, "_asan", ["-fsanitize=address", "-shared-libsan", "-g", "-O1"], f"{ARCH}:xnack+", verbose)
    # both libraries of the one source set: the product library and the development build (same objects' worth of compiles
    # again; `--no-dev` skips it)
    if dev and (force or _stale(DEV_LIB)):
        _build(DEV_LIB, "_dev", ["-O3", "-DTULIP_DEV_VARIANTS=1"], ARCH, verbose)
    if not force and not _stale():
        return LIB
    return _build(LIB, "", ["-O3"], ARCH, verbose)


def _build(LIB: str, suffix: str, flags, ARCH: str, verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, s.replace(".hip", suffix + ".o"))
        cmd = [hipcc, f"--offload-arch={ARCH}", *flags, "-std=c++17", "-fPIC", "-Wno-unused-value",
               "-I", os.path.join(ROOT, "include"), "-I", HERE, "-c", os.path.join(HERE, s), "-o", o]
        cmd += os.environ.get("TULIP_HIPCC_FLAGS", "").split()      # dev: e.g. -DTULIP_GEMM_WSK=0 for an A/B build
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            print(f"---- {s} failed:\n{out}", file=sys.stderr)
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if suffix == "_asan":
        cmd += ["-fsanitize=address", "-shared-libsan"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, asan="--asan" in sys.argv, dev="--no-dev" not in sys.argv))
