"""Every environment switch of the host side is read HERE, once, at import (engine.py / trainer.py hold the attributes and the
comments that say what each one does and what it measured).  A knob is (environment variable, kind, default); REGISTRY records
what was read, `non_default()` is what `bench.py` puts into its JSON line (`config.knobs_non_default`) so that a number
measured under an A/B switch says so itself, and `python -m tulip_amd.knobs` prints the table.  The C library reads no environment
at all (include/tulip_hip.h, conventions): these are launch-sequence and scheduling choices of the Python host only."""
import os

REGISTRY = {}        # name -> (kind, default, value)


def _note(name, kind, default, value):
    REGISTRY[name] = (kind, default, value)
    return value


def on(name: str, default: bool) -> bool:
    """A switch that is on unless set to "0" (default on) / off unless set to something other than "0" (default off)."""
    return _note(name, "on", bool(default), os.environ.get(name, "1" if default else "0") != "0")


def is_one(name: str) -> bool:
    """Off unless set to exactly "1"."""
    return _note(name, "is_one", False, os.environ.get(name, "0") == "1")


def is_zero(name: str) -> bool:
    """True only when set to exactly "0" (a default-on feature being switched off)."""
    return _note(name, "is_zero", False, os.environ.get(name, "1") == "0")


def integer(name: str, default: int) -> int:
    return _note(name, "int", int(default), int(os.environ.get(name, str(default))))


def text(name: str, default: str) -> str:
    return _note(name, "str", default, os.environ.get(name, default))


def names(name: str, default: str = "") -> tuple:
    """comma-separated list -> tuple of non-empty strings"""
    return _note(name, "names", tuple(x for x in default.split(",") if x), tuple(x for x in os.environ.get(name, default).split(",") if x))


def non_default() -> dict:
    return {k: v for k, (kind, d, v) in sorted(REGISTRY.items()) if v != d}


if __name__ == "__main__":
    import tulip_amd.engine                         # noqa: F401  (its class body registers the engine's knobs; the Trainer's
    from tulip_amd import knobs as K                # are read per instance, in Trainer.__init__)
    for k, (kind, d, v) in sorted(K.REGISTRY.items()):
        print(f"{k:34s} {kind:8s} default {d!r:28} {'' if v == d else '-> ' + repr(v)}")
