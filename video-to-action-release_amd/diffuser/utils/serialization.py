"""Checkpoint-folder helpers (reference: diffuser/utils/serialization.py:14-39) without its h5py / tqdm imports."""
import glob
import os
import pickle

__all__ = ["mkdir", "get_latest_epoch", "load_config"]


def mkdir(savepath):
    """Create `savepath` if it is missing; True iff it was created."""
    if os.path.exists(savepath):
        return False
    os.makedirs(savepath)
    return True


def get_latest_epoch(loadpath):
    """Largest N among `model-N.pt` in the folder given as a tuple of path parts (or a path); -1 if none.  `model-front.pt` and other
    non-numeric names are skipped."""
    path = os.path.join(*[str(p) for p in loadpath]) if isinstance(loadpath, (tuple, list)) else str(loadpath)
    best = -1
    for f in glob.glob(os.path.join(path, "model-*.pt")):
        stem = os.path.basename(f)[len("model-"):-len(".pt")]
        if stem.isdigit():
            best = max(best, int(stem))
    return best


def load_config(*loadpath):
    path = os.path.join(*loadpath)
    with open(path, "rb") as f:
        cfg = pickle.load(f)
    print(f"[ utils/serialization ] Loaded config from {path}")
    return cfg
