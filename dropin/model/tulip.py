"""Drop-in for the reference's `tulip/model/tulip.py`: the names `main_lidar_upsampling.py:221-230` looks up through
`tulip.__dict__[args.model_select]` (`tulip_base`, `tulip_large`; `--model_select` choices, `:39-40`) and the `TULIP`
class itself, all from the HIP implementation.  See dropin/model/__init__.py for how it gets in front."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:                      # tulip_amd lives next to dropin/ (in-tree build, not pip-installed)
    sys.path.append(_ROOT)

from tulip_amd.model.tulip import *            # noqa: F401,F403,E402
from tulip_amd.model.tulip import TULIP, tulip_base, tulip_large   # noqa: F401,E402  (explicit: they must be in __dict__)
