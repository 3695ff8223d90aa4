"""Import overlay: makes `import diffuser...` / `import flowdiffusion...` resolve to this package's modules FIRST and to the user's own
video-to-action checkout for everything this package does not provide -- without editing that checkout.

Why a finder and not just `sys.path`: the reference's entry scripts put their working directory at the FRONT of `sys.path`
(scripts/train_libero_dp.py:2), and its `flowdiffusion/` is a regular package, so the checkout's copy would win whatever PYTHONPATH
says.  `install()` puts a meta-path finder ahead of the path-based one that answers for the two top-level names only, from this
package's directory; the packages found there extend their `__path__` (pkgutil.extend_path) over the same-named directories of every
`sys.path` entry, this package's first -- sub-modules that exist here win, all others (`diffuser.utils.rendering`,
`flowdiffusion.flowdiffusion.utils`, `environment`, configs ...) come from the checkout.

    python -m v2a_hip.launch scripts/train_libero_dp.py --config config/libero/lb_tk8_65to72.py       # zero edits
or, as the first import of a script / notebook:  `import v2a_hip.overlay; v2a_hip.overlay.install()`."""
import importlib.abc
import importlib.machinery
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))       # .../video-to-action-release_amd
NAMES = ("diffuser", "flowdiffusion")


class _Overlay(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname in NAMES:
            return importlib.machinery.PathFinder.find_spec(fullname, [ROOT])
        return None


def installed():
    return any(isinstance(f, _Overlay) for f in sys.meta_path)


def install():
    """Idempotent.  Raises if one of the overlaid packages was already imported from somewhere else (the overlay would be ignored)."""
    for name in NAMES:
        mod = sys.modules.get(name)
        if mod is not None:
            where = [os.path.abspath(p) for p in getattr(mod, "__path__", [])]
            if not where or not where[0].startswith(ROOT):
                raise ImportError(f"v2a_hip.overlay.install(): `{name}` is already imported from {where or mod}; install the overlay "
                                  f"before the first `import {name}`")
    if not installed():
        sys.meta_path.insert(0, _Overlay())
    if ROOT not in sys.path:
        sys.path.append(ROOT)          # `v2a_hip`, `config.*` of this package stay importable; never ahead of the user's entries
