"""`model` package shim: puts the MI355X-native `model.tulip` in front of the reference's own.

The reference imports its network as `import model.tulip as tulip` (tulip/main_lidar_upsampling.py:29) from the
script's directory, where `tulip/model/` is a directory WITHOUT an `__init__.py` -- a namespace portion.  Python's
import system prefers a regular package found anywhere on `sys.path` over namespace portions found earlier, so with
this directory's parent on PYTHONPATH

    PYTHONPATH=/path/to/repo/dropin torchrun --nproc_per_node=N tulip/main_lidar_upsampling.py ...

`import model` resolves HERE and `model.tulip` is dropin/model/tulip.py (a re-export of `tulip_amd.model.tulip`), with
no edit to the reference.  The reference's other `model.*` modules (model/swin_transformer_v2.py) stay importable:
`__path__` is extended with every other `model/` directory on `sys.path`, this one first.
"""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)
