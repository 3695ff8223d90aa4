"""`diffuser.utils` for the MI355X-native package: what `scripts/train_libero_dp.py`, `diffuser/libero/*` and the configs take from the
reference's `diffuser/utils/__init__.py` (its seven star-imports), restated here without the simulator-side dependencies that module
pulls in at import time (gym, mujoco_py, tap, termcolor, h5py, imageio, matplotlib).

Names that are not restated here (plotting, gif / mp4 writers, renderers: everything outside the hot path) resolve LAZILY from the
user's own checkout: this package extends its `__path__` over every `diffuser/utils` directory on `sys.path`, so
`diffuser.utils.rendering`, `diffuser.utils.eval_utils`, ... import from there, and `utils.<name>` falls through to those modules
(module-level `__getattr__`) the first time it is asked for.  Nothing of the user's tree is imported until then."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from .serialization import *      # noqa: F401,F403,E402
from .config import *             # noqa: F401,F403,E402
from .textio import *             # noqa: F401,F403,E402
from .setup import *              # noqa: F401,F403,E402
from .arrays import *             # noqa: F401,F403,E402
from .timer import Timer          # noqa: F401,E402

# modules of the reference's `diffuser.utils` whose names `utils.<name>` may still be asked for (its __init__ star-imports them)
_FALLTHROUGH = ("eval_utils", "luo_utils", "file_utils", "rendering", "git_utils")


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    import importlib
    for mod in _FALLTHROUGH:
        try:
            m = importlib.import_module(f"{__name__}.{mod}")
        except ImportError:
            continue
        if hasattr(m, name):
            v = getattr(m, name)
            globals()[name] = v
            return v
    raise AttributeError(f"module {__name__!r} has no attribute {name!r} (not restated by the MI355X package and not found in a "
                         f"`diffuser/utils` directory of the user's checkout on sys.path)")
