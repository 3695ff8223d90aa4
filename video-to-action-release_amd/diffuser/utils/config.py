"""`Config`: a deferred constructor the entry scripts build for the dataset and the trainer
(reference: diffuser/utils/config.py; used at scripts/train_libero_dp.py:35-43,65-98)."""
import collections.abc
import importlib
import os
import pickle

__all__ = ["import_class", "Config"]


def import_class(_class):
    """'pkg.mod.Name' -> the class; anything that is not a string is returned as is."""
    if not isinstance(_class, str):
        return _class
    module_name, _, class_name = _class.rpartition(".")
    cls = getattr(importlib.import_module(module_name), class_name)
    print(f"[ utils/config ] Imported {module_name}:{class_name}")
    return cls


class Config(collections.abc.Mapping):
    """Config(cls_or_dotted_path, savepath=None, device=None, **kwargs): a read-only mapping of the keyword arguments (also readable as
    attributes; `._dict` is the dict itself) that builds `cls(*args, **more, **kwargs)` when called.  With `savepath` (a path or a tuple
    of path parts) the config is pickled there on construction."""

    def __init__(self, _class, verbose=True, savepath=None, device=None, **kwargs):
        self._class = import_class(_class)
        self._device = device
        self._dict = dict(kwargs)
        if verbose:
            print(self)
        if savepath is not None:
            savepath = os.path.join(*savepath) if isinstance(savepath, tuple) else savepath
            self.savepath = savepath
            folder = os.path.dirname(savepath)
            if folder:
                os.makedirs(folder, exist_ok=True)
            try:
                with open(savepath, "wb") as f:
                    pickle.dump(self, f)
                print(f"[ utils/config ] Saved config to: {savepath}\n")
            except Exception as e:      # an unpicklable member (a lambda in a config): the config itself still works
                print(f"[ utils/config ] not saved ({type(e).__name__}: {e})")

    def __repr__(self):
        lines = [f"\n[utils/config ] Config: {self._class}"]
        for key in sorted(self._dict):
            val = self._dict[key]
            if key == "problems_dict" and val is not None:      # large arrays: a corner of each
                val = {k: (v[0, :2] if getattr(v, "ndim", 0) >= 2 else v[:2]) for k, v in val.items()}
            lines.append(f"    {key}: {val}")
        return "\n".join(lines) + "\n"

    def __iter__(self):
        return iter(self._dict)

    def __getitem__(self, item):
        return self._dict[item]

    def __len__(self):
        return len(self._dict)

    def __getattr__(self, attr):
        if attr == "_dict":                 # unpickling probes attributes before __init__ has run
            self.__dict__["_dict"] = {}
            return self.__dict__["_dict"]
        try:
            return self.__dict__["_dict"][attr]
        except KeyError:
            raise AttributeError(attr)

    def __call__(self, *args, **kwargs):
        instance = self._class(*args, **kwargs, **self._dict)
        if self._device:
            instance = instance.to(self._device)
        sp = self.__dict__.get("savepath")
        if sp and "model_config.pkl" in sp:
            with open(sp.replace("model_config.pkl", "model_config.txt"), "w") as f:
                print(instance, file=f)
        return instance
