"""CPU: the host side's environment switches are read in one place (tulip_amd/knobs.py) and nowhere else; the C library reads none."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_environment_reads_outside_the_knob_table():
    for f in ("engine.py", "trainer.py", "ops.py", "ddp.py", "infer.py", "evaluation.py", "data.py"):
        src = open(os.path.join(ROOT, "tulip_amd", f)).read()
        assert "os.environ" not in src, f
    for f in os.listdir(os.path.join(ROOT, "tulip_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            assert not re.search(r"\bgetenv\s*\(", open(os.path.join(ROOT, "tulip_amd", "csrc", f)).read()), f


def test_knob_registry_and_non_default_report():
    code = ("import tulip_amd.engine; from tulip_amd import knobs; import json; "
            "print(json.dumps({'n': len(knobs.REGISTRY), 'nd': knobs.non_default()}))")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in list(env):
        if k.startswith("TULIP_"):
            del env[k]
    import json
    base = json.loads(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout)
    assert 10 <= base["n"] <= 25 and base["nd"] == {}      # (round 6: the switches whose losing side was measured are gone)
    env.update(TULIP_FUSE_DEEP="0", TULIP_WGRAD_GROUP_MAX="7", TULIP_ATTN_FP8="1")
    got = json.loads(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout)
    assert got["nd"] == {"TULIP_ATTN_FP8": True, "TULIP_FUSE_DEEP": False, "TULIP_WGRAD_GROUP_MAX": 7}
