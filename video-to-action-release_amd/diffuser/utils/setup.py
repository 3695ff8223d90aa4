"""Command line + config-file front end of the entry scripts (reference: diffuser/utils/setup.py; used as
`class Parser(utils.Parser): dataset: str = ''; config: str = ''` and `args = Parser().parse_args('diffusion')` at
scripts/train_libero_dp.py:18-23).

The reference derives `Parser` from `tap.Tap`; this one reads the same class-level annotations itself (no `tap`, no `termcolor`), with
the behaviour the scripts rely on:
  * every annotated class attribute `name: type = default` is an option `--name value`; unknown options are not an error -- their raw
    tokens collect in `args.extra_args` (Tap's `known_only=True`);
  * `parse_args(experiment)` then loads the config module named by `--config` (a dotted module path, or a path ending in `.py`), copies
    `module.base[experiment]` (updated by the per-dataset override dict, if the module has one) onto `args`, applies `--key value` pairs
    from `extra_args` to keys the config defines (value parsed with the type of the config's value), expands `'f:...{key}...'` strings,
    seeds the generators, builds `exp_name` / `savepath`, creates the folder and writes `args.json` there;
  * `args.as_dict()` / `args.save(path)` as Tap's."""
import datetime
import importlib
import importlib.util
import json
import os
import random
import subprocess
import sys

import numpy as np
import torch

from .serialization import mkdir

__all__ = ["set_seed", "watch", "lazy_fstring", "Parser", "get_git_rev"]

# keys that exist on the command line only (no config entry needed to override them)
_CLI_ONLY_KEYS = ("plan_n_maze", "diffusion_epoch", "config_2", "num_vid_pred_per_ep")


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def watch(args_to_watch):
    """[(key, label), ...] -> a function of `args` that joins `label + str(value)` of the keys present into an experiment name."""

    def _fn(args):
        parts = []
        for key, label in args_to_watch:
            if not hasattr(args, key):
                continue
            val = getattr(args, key)
            if isinstance(val, dict):
                val = "_".join(f"{k}-{v}" for k, v in val.items())
            parts.append(f"{label}{val}")
        name = "_".join(parts).replace("/_", "/").replace("(", "").replace(")", "").replace(", ", "-")
        print("exp_name", name)
        return name

    return _fn


def lazy_fstring(template, args):
    """Evaluate `template` as an f-string in which `args` is in scope (config strings of the form 'f:...{args.key}...')."""
    return eval(f"f'{template}'", {"args": args})


def get_git_rev(path=None):
    try:
        out = subprocess.run(["git", "rev-parse", "--abbrev-ref", "HEAD", "HEAD"], cwd=path or os.getcwd(), capture_output=True, text=True,
                             timeout=5)
        if out.returncode:
            return None
        branch, rev = out.stdout.split()[:2]
        return f"{branch}_{rev}"
    except Exception:
        return None


def _convert(text, typ):
    if typ is bool:
        if text.lower() in ("true", "1", "yes"):
            return True
        if text.lower() in ("false", "0", "no"):
            return False
        raise ValueError(f"not a bool: {text!r}")
    if typ in (int, float, str):
        return typ(text)
    return text


class Parser:
    extra_args: list

    def __init__(self):
        fields = {}
        for klass in reversed(type(self).__mro__):
            for name, typ in getattr(klass, "__annotations__", {}).items():
                if name.startswith("_") or name == "extra_args":
                    continue
                fields[name] = typ
        self._fields = fields
        for name in fields:
            if hasattr(type(self), name):
                setattr(self, name, getattr(type(self), name))
        self.extra_args = []
        self._dict = {}

    # ------------------------------------------------------------------ Tap's surface
    def _parse_known(self, argv):
        argv = list(sys.argv[1:] if argv is None else argv)
        extras, i = [], 0
        while i < len(argv):
            tok = argv[i]
            name = tok[2:].split("=", 1)[0] if tok.startswith("--") else None
            if name in self._fields:
                typ = self._fields[name]
                if "=" in tok:
                    val = tok.split("=", 1)[1]
                elif typ is bool and (i + 1 >= len(argv) or argv[i + 1].startswith("--")):
                    val = "true"
                else:
                    if i + 1 >= len(argv):
                        raise SystemExit(f"option --{name} needs a value")
                    i += 1
                    val = argv[i]
                setattr(self, name, _convert(val, typ))
            else:
                extras.append(tok)
            i += 1
        self.extra_args = extras
        return self

    def as_dict(self):
        out = {}
        for klass in reversed(type(self).__mro__):
            for k, v in vars(klass).items():
                if not k.startswith("_") and not callable(v) and not isinstance(v, (property, staticmethod, classmethod)):
                    out[k] = v
        for k, v in vars(self).items():
            if not k.startswith("_"):
                out[k] = v
        return out

    def save(self, path=None, skip_unpicklable=True):
        """Write the arguments as JSON (`<savepath>/args.json` by default); values JSON cannot express are written as their repr."""
        if path is None:
            path = os.path.join(self.savepath, "args.json")
            print(f"[ utils/setup ] Saved args to {path}")

        def enc(o):
            if isinstance(o, (np.integer,)):
                return int(o)
            if isinstance(o, (np.floating,)):
                return float(o)
            if isinstance(o, np.ndarray):
                return o.tolist()
            if isinstance(o, tuple):
                return list(o)
            return repr(o)

        with open(path, "w") as f:
            json.dump(self.as_dict(), f, indent=4, sort_keys=True, default=enc)

    # ------------------------------------------------------------------ the reference's pipeline
    def parse_args(self, experiment=None, from_jupyter=False, use_config_2=False, not_parse=False, input_args=None, *, argv=None):
        args = self._parse_known(argv)
        if not_parse:
            args = input_args
        if use_config_2:
            args.config = args.config_2
        if not hasattr(args, "config"):         # not configured from a config script: plain options only
            return args
        args = self.read_config(args, experiment)
        if not from_jupyter:
            self.add_extras(args)
        self.eval_fstrings(args)
        self.set_seed(args)
        self.get_commit(args)
        self.generate_exp_name(args)
        self.mkdir(args)
        self.save_diff(args)
        self.set_wandb(args)
        self.check_sinPosEmb(args)
        return args

    def read_config(self, args, experiment):
        """module.base[experiment] (+ the per-dataset overrides module.<dataset>[experiment]) -> attributes of args."""
        if args.config.endswith(".py"):         # a file path: its folder joins sys.path so that the config's own relative imports work
            args.config = args.config[:-3]
            folder = os.path.dirname(os.path.abspath(args.config))
            if folder not in sys.path:
                sys.path.append(folder)
            module = importlib.import_module(os.path.basename(args.config))
        else:
            module = importlib.import_module(args.config)
        base = getattr(module, "base")
        if "dataset" in base.keys():
            dataset = base["dataset"].replace("-", "_")
            args.dataset = dataset.replace("_", "-")
        else:
            dataset = args.dataset.replace("-", "_")
        if experiment == "plan":
            args.dataset_eval = args.dataset[:-3] + "-eval" + args.dataset[-3:]
        print(f"[ utils/setup ] Reading config: {args.config}:{dataset}")
        params = base[experiment]
        if hasattr(module, dataset) and experiment in getattr(module, dataset):
            print(f"[ utils/setup ] Using overrides | config: {args.config} | dataset: {dataset}")
            params.update(getattr(module, dataset)[experiment])
        else:
            print(f"[ utils/setup ] Not using overrides | config: {args.config} | dataset: {dataset}")
        self._dict = {}
        for key, val in params.items():
            setattr(args, key, val)
            self._dict[key] = val
        return args

    def add_extras(self, args):
        """`--key value` pairs left in extra_args override config entries (value parsed with the type of the entry it replaces)."""
        extras = args.extra_args
        if not len(extras):
            return
        print(f"[ utils/setup ] Found extras: {extras}")
        assert len(extras) % 2 == 0, f"Found odd number ({len(extras)}) of extras: {extras}"
        for i in range(0, len(extras), 2):
            key, val = extras[i].replace("--", ""), extras[i + 1]
            if key in _CLI_ONLY_KEYS:
                setattr(args, key, val)
            else:
                assert hasattr(args, key), f"[ utils/setup ] {key} not found in config: {args.config}"
            old = getattr(args, key)
            print(f"[ utils/setup ] Overriding config | {key} : {old} --> {val}")
            if val == "None":
                val = None
            elif val == "latest":
                pass
            elif type(old) in (bool, type(None)):
                try:
                    val = eval(val)
                except Exception:
                    print(f"[ utils/setup ] Warning: could not parse {val} (old: {old}, {type(old)}), using str")
            else:
                val = type(old)(val)
            setattr(args, key, val)
            self._dict[key] = val

    def eval_fstrings(self, args):
        for key, old in list(self._dict.items()):
            if isinstance(old, str) and old[:2] == "f:":
                new = lazy_fstring(old.replace("{", "{args.").replace("f:", ""), args)
                print(f"[ utils/setup ] Lazy fstring | {key} : {old} --> {new}")
                setattr(self, key, new)
                self._dict[key] = new

    def set_seed(self, args):
        if "seed" not in dir(args):
            return
        print(f"[ utils/setup ] Setting seed: {args.seed}")
        set_seed(args.seed)

    def generate_exp_name(self, args):
        if "exp_name" not in dir(args):
            return
        exp_name = getattr(args, "exp_name")
        if callable(exp_name):
            name = exp_name(args)
            print(f"[ utils/setup ] Setting exp_name to: {name}")
            setattr(args, "exp_name", name)
            self._dict["exp_name"] = name

    def mkdir(self, args):
        if "logbase" in dir(args) and "dataset" in dir(args) and "exp_name" in dir(args):
            args.savepath = os.path.join(args.logbase, args.dataset, args.exp_name)
            self.savepath = args.savepath
            self._dict["savepath"] = args.savepath
            if "suffix" in dir(args):
                args.savepath = os.path.join(args.savepath, args.suffix)
            if mkdir(args.savepath):
                print(f"[ utils/setup ] Made savepath: {args.savepath}")
            self.save()

    def get_commit(self, args):
        args.commit = get_git_rev()

    def save_diff(self, args):
        pass        # (the reference's is a no-op as well: writing the diff slowed start-up)

    def set_wandb(self, args):
        if "exp_name" not in dir(args):
            return
        name = os.path.split(args.exp_name)[1]
        date = datetime.datetime.now().strftime("%m%d")
        args.logger_name = f"{date}-{name}-{os.environ.get('SLURM_JOB_ID')}"
        args.logger_id = args.logger_name

    def check_sinPosEmb(self, args):
        if getattr(args, "wall_sinPosEmb", None):
            assert args.dataset_config["use_normed_wallLoc"] is True
