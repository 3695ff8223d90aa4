"""Small host helpers the trainer needs from the reference's `diffuser.utils` / `diffuser.datasets` (restated, not imported, so
that this package works without gym / mujoco_py / tap / wandb which `diffuser.utils` pulls in at import time: SURVEY.md 8b)."""
import glob
import importlib
import os
import time
import numpy as np
import torch

LB_ACTION_MIN = np.array([-1.0] * 7, dtype=np.float32)        # diffuser/datasets/__init__.py:21-22
LB_ACTION_MAX = np.array([1.0] * 7, dtype=np.float32)

_COLORS = {"r": 31, "g": 32, "y": 33, "b": 34, "m": 35, "c": 36}


def print_color(*msg, c="c"):
    print(f"\033[{_COLORS.get(c, 36)}m" + " ".join(str(m) for m in msg) + "\033[0m", flush=True)


class Timer:
    """Callable stopwatch: `t()` returns seconds since the previous call (diffuser/utils/timer.py)."""

    def __init__(self):
        self._t = time.time()

    def __call__(self, reset=True):
        now = time.time()
        d = now - self._t
        if reset:
            self._t = now
        return d


def number_by_ratio(num, ratio):
    """num=10, ratio=[0.2, 0.8] -> [2, 8]: round(ratio * num) per entry (diffuser/utils/arrays.py:21-31)."""
    ratio = np.array(ratio)
    assert np.isclose(sum(ratio), 1), "Ratios must sum to 1"
    out = ratio * num
    assert np.isclose(out.sum(), num)
    return np.round(out).astype(np.int32).tolist()


def imgs_preproc_simple_noCrop_v1(imgs):
    """uint8 [B,H,W,3] -> float32 [B,3,H,W] in [0,1] with a true division by 255 (diffuser/datasets/img_utils.py:27-37, 62-71)."""
    assert isinstance(imgs, np.ndarray) and imgs.ndim == 4 and imgs.dtype == np.uint8
    return torch.from_numpy(imgs.copy()).permute(0, 3, 1, 2).float() / 255.0


def get_latest_epoch(loadpath):
    """Largest N among `model-N.pt` under the folder(s) (diffuser/utils/serialization.py)."""
    path = os.path.join(*[str(p) for p in loadpath]) if isinstance(loadpath, (tuple, list)) else str(loadpath)
    best = -1
    for f in glob.glob(os.path.join(path, "model-*.pt")):
        try:
            best = max(best, int(os.path.basename(f)[len("model-"):-len(".pt")]))
        except ValueError:
            pass
    return best


def report_parameters(model, topk=10):
    counts = {k: p.numel() for k, p in model.named_parameters()}
    n = sum(counts.values())
    print(f"[ utils/arrays ] Total parameters: {n / 1e6:.2f} M")
    for k in sorted(counts, key=lambda x: -counts[x])[:topk]:
        print(f"        {k}: {counts[k] / 1e6:.2f} M")
    return n


class Config(dict):
    """Deferred constructor: Config(cls_or_dotted_path, savepath=None, **kwargs)(*args, **more) -> cls(*args, **kwargs, **more)."""

    def __init__(self, _class, savepath=None, **kwargs):
        super().__init__(kwargs)
        self._class = _class if not isinstance(_class, str) else getattr(importlib.import_module(_class.rsplit(".", 1)[0]),
                                                                           _class.rsplit(".", 1)[1])
        self._dict = dict(kwargs)
        if savepath is not None:
            import pickle
            path = os.path.join(*savepath) if isinstance(savepath, tuple) else savepath
            os.makedirs(os.path.dirname(path), exist_ok=True)
            try:
                with open(path, "wb") as f:
                    pickle.dump({"_class": getattr(self._class, "__name__", str(self._class)), **self._dict}, f)
            except Exception as e:          # unpicklable members (numpy scalars are fine; lambdas are not): the config still works
                print(f"[ utils/config ] not saved: {e}")

    def __call__(self, *args, **kwargs):
        return self._class(*args, **self._dict, **kwargs)
