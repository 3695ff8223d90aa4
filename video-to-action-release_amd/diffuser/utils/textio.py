"""Console helpers (reference: diffuser/utils/eval_utils.py `print_color`, `get_time`, `suppress_stdout`) on plain ANSI codes."""
import contextlib
import os
from datetime import datetime

__all__ = ["print_color", "get_time", "suppress_stdout"]

_ANSI = {"r": 31, "g": 32, "y": 33, "b": 34, "m": 35, "c": 36}


def print_color(s, *args, c="r"):
    """print(s, *args) in colour `c` ('r', 'b', 'y'; anything else cyan, as in the reference)."""
    code = _ANSI.get(c, 36) if c in ("r", "b", "y", "g", "m") else 36
    print(f"\033[{code}m" + " ".join(str(m) for m in (s,) + args) + "\033[0m", flush=True)


def get_time():
    return datetime.now().strftime("%y%m%d-%H%M%S")


@contextlib.contextmanager
def suppress_stdout():
    with open(os.devnull, "w") as sink, contextlib.redirect_stdout(sink):
        yield
