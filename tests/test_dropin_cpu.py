"""CPU: the literal drop-in (`dropin/model/tulip.py`).  The reference does `import model.tulip as tulip`
(tulip/main_lidar_upsampling.py:29) from its script directory, where `tulip/model/` has no `__init__.py` (a namespace
portion), and then `tulip.__dict__[args.model_select](...)` (:221-230).  With `dropin/` on PYTHONPATH the regular package
`dropin/model` wins over the namespace portion although the script directory comes first on sys.path -- checked here in a
subprocess laid out like the reference (script dir with a namespace `model/` holding a sibling module and a decoy
tulip.py), with nothing edited."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_model_tulip_resolves_to_the_hip_module_without_an_edit(tmp_path):
    ref = tmp_path / "tulip"
    (ref / "model").mkdir(parents=True)                                  # namespace portion, like the reference's
    (ref / "model" / "tulip.py").write_text("WHO = 'reference'\n")       # the file the reference itself would import
    (ref / "model" / "swin_transformer_v2.py").write_text("WHO = 'reference sibling'\n")
    (ref / "main.py").write_text(textwrap.dedent("""
        import model.tulip as tulip                       # main_lidar_upsampling.py:29, verbatim
        import model.swin_transformer_v2 as sib           # the reference's other model.* module stays importable
        assert sib.WHO == 'reference sibling'
        assert not hasattr(tulip, 'WHO'), 'the reference model was imported'
        m = tulip.__dict__['tulip_base'](img_size=(16, 1024), target_img_size=(64, 1024), patch_size=(1, 4), in_chans=1,
                                         window_size=[2, 8], swin_v2=False, pixel_shuffle=True, circular_padding=True,
                                         log_transform=True, patch_unmerging=True)          # :221-230
        assert 'tulip_large' in tulip.__dict__ and 'TULIP' in tulip.__dict__
        import tulip_amd.model.tulip as T
        assert type(m) is T.TULIP and m.engine() is not None
        print('OK', sum(p.numel() for p in m.parameters()))
    """))
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = os.path.join(ROOT, "dropin")
    r = subprocess.run([sys.executable, str(ref / "main.py")], env=env, capture_output=True, text=True, timeout=600,
                       cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.split() == ["OK", "27149076"]
    # without the shim on the path the same script gets the reference's module: the shim is what makes the difference
    env.pop("PYTHONPATH")
    r = subprocess.run([sys.executable, str(ref / "main.py")], env=env, capture_output=True, text=True, timeout=600,
                       cwd=str(tmp_path))
    assert r.returncode != 0 and "the reference model was imported" in r.stderr


def test_bench_spawns_its_own_ranks_command_line():
    """`python bench.py --gpus N` without WORLD_SIZE must start N ranks itself (the driver's command line): the launch
    command is checked here without a GPU by substituting the launcher."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    calls = []
    real = bench.subprocess.call
    bench.subprocess.call = lambda cmd, env=None: (calls.append((cmd, env)), 0)[1]
    try:
        rc = bench.spawn_ranks(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"])
    finally:
        bench.subprocess.call = real
    assert rc == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["TULIP_BENCH_SPAWNED"] == "1"
