"""Stopwatch used by the trainer's log lines (reference: diffuser/utils/luo_utils.py `Timer`)."""
import time


class Timer:
    """`t()` returns the seconds since the previous call (or construction) and, unless reset=False, restarts the clock."""

    def __init__(self):
        self._t = time.time()

    def __call__(self, reset=True):
        now = time.time()
        d = now - self._t
        if reset:
            self._t = now
        return d
