"""Random-action episode files: readers for the layout written by
environment/libero/lb_data/lb_randsam.py:84-104 --

    {task}/{episode}/agentview_image  uint8  [T+1,128,128,3]
    {task}/{episode}/action           float  [T,7]
    {task}/{episode}/ee_poses         float  [T+1,3]          attrs: env_seed, env_list_name

-- feeding the HBM-resident uint8 replay store directly (the reference re-expands every frame to an fp32 CPU tensor,
lb_online_trainer_v7.py:718-780).  Three back ends behind one interface: the HDF5 file through the library's native reader
(csrc/h5read.hip -- h5py is not needed), a directory of `.npz` episodes with the same keys, and a synthetic generator that rolls
uniform random actions in the toy environment (benchmarks / tests)."""
import os
import numpy as np


class RandSamH5:
    """The HDF5 file itself, through the library's own reader (csrc/h5read.hip: v2a_h5_*; no h5py, no libhdf5).  The file stays
    memory-mapped while the object lives; `episode` copies one episode's frames / actions out as numpy arrays (uint8 frames go on to
    the HBM replay store unchanged).  Unsupported on-disk features (compression, libver='latest' structures) raise with the reason."""

    _NP = {(0, 1, 0): np.uint8, (0, 1, 1): np.int8, (0, 2, 0): np.uint16, (0, 2, 1): np.int16, (0, 4, 0): np.uint32, (0, 4, 1): np.int32,
           (0, 8, 0): np.uint64, (0, 8, 1): np.int64, (1, 4, 1): np.float32, (1, 8, 1): np.float64}

    def __init__(self, path):
        import ctypes
        from v2a_hip._lib import lib
        self._ct, self._lib = ctypes, lib
        self.path = path
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self._h = ctypes.c_void_p()
        if lib.v2a_h5_open(os.fsencode(path), ctypes.byref(self._h)) != 0:
            msg = self._err()
            lib.v2a_h5_close(self._h)
            self._h = None
            raise OSError(f"{path}: {msg}")

    def _err(self):
        return (self._lib.v2a_h5_last_error(self._h) or b"").decode(errors="replace")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.v2a_h5_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def members(self, group):
        cap = 1 << 16
        while True:
            buf = self._ct.create_string_buffer(cap)
            n = self._lib.v2a_h5_list(self._h, group.encode(), buf, cap)
            if n == -2:
                cap *= 4
                continue
            if n < 0:
                raise KeyError(f"{group}: {self._err()}")
            return buf.value.decode().split("\n")[:n]

    def read(self, path):
        ct = self._ct
        tc, es, sg, nd, nb = ct.c_int(), ct.c_int(), ct.c_int(), ct.c_int(), ct.c_longlong()
        dims = (ct.c_longlong * 8)()
        if self._lib.v2a_h5_dataset_info(self._h, path.encode(), ct.byref(tc), ct.byref(es), ct.byref(sg), ct.byref(nd), dims, ct.byref(nb)):
            raise KeyError(f"{path}: {self._err()}")
        dt = self._NP.get((tc.value, es.value, sg.value if tc.value == 0 else 1))
        if dt is None:
            raise TypeError(f"{path}: unsupported element type (class {tc.value}, {es.value} bytes)")
        out = np.empty(tuple(dims[i] for i in range(nd.value)), dtype=dt)
        if self._lib.v2a_h5_read(self._h, path.encode(), out.ctypes.data, out.nbytes):
            raise OSError(f"{path}: {self._err()}")
        return out

    def num_episodes(self, task):
        # lb_online_trainer_v7.py:246-252 (episode groups are named by their integer index; anything else in the task group is ignored)
        return sorted(int(k) for k in self.members(task) if k.isdigit())[-1] + 1

    def has(self, task, i_ep):
        r = self._lib.v2a_h5_exists(self._h, f"{task}/{i_ep}".encode())
        if r < 0:
            raise OSError(f"{self.path}: {self._err()}")
        return bool(r)

    def episode(self, task, i_ep):
        return self.read(f"{task}/{i_ep}/agentview_image"), self.read(f"{task}/{i_ep}/action")


def check_and_clip_actions(acts, act_min, act_max, slack=0.012):
    """The loader's contract on stored actions (lb_online_trainer_v7.py:749-752): every component strictly inside the limits widened
    by `slack` (the generator's orientation noise overshoots +-0.1 by up to 0.011), then clipped to the limits; float32 out."""
    acts = np.asarray(acts)
    lo, hi = np.asarray(act_min)[None], np.asarray(act_max)[None]
    if not (acts > lo - slack).all() or not (acts < hi + slack).all():
        raise AssertionError("stored action outside the action limits by more than the loader's slack")
    return np.clip(acts, a_min=lo, a_max=hi).astype(np.float32)


class RandSamNpzDir:
    """`{root}/{task with spaces -> underscores}/{i}.npz` holding agentview_image / action / ee_poses."""

    def __init__(self, root):
        self.root = root
        if not os.path.isdir(root):
            raise FileNotFoundError(root)

    def _dir(self, task):
        return os.path.join(self.root, task.replace(' ', '_'))

    def num_episodes(self, task):
        ids = [int(os.path.splitext(f)[0]) for f in os.listdir(self._dir(task)) if f.endswith('.npz')]
        return max(ids) + 1

    def has(self, task, i_ep):
        return os.path.exists(os.path.join(self._dir(task), f'{i_ep}.npz'))

    def episode(self, task, i_ep):
        with np.load(os.path.join(self._dir(task), f'{i_ep}.npz')) as z:
            return z['agentview_image'], z['action']

    @staticmethod
    def write_episode(root, task, i_ep, imgs, acts, ee_poses=None, env_seed=0):
        d = os.path.join(root, task.replace(' ', '_'))
        os.makedirs(d, exist_ok=True)
        np.savez(os.path.join(d, f'{i_ep}.npz'), agentview_image=np.asarray(imgs, np.uint8), action=np.asarray(acts, np.float32),
                 ee_poses=np.zeros((len(imgs), 3), np.float32) if ee_poses is None else ee_poses, env_seed=env_seed)


class RandSamSynthetic:
    """Episodes generated on demand: `ep_len` uniform random delta actions (orientation components scaled to +-0.1, the range the
    loader asserts on real data) rolled in `env_list`'s toy world; deterministic in (task, episode)."""

    def __init__(self, env_list, episodes_per_task=500, ep_len=(120, 143)):
        self.env_list = env_list
        self.n = episodes_per_task
        self.ep_len = ep_len

    def num_episodes(self, task):
        return self.n

    def has(self, task, i_ep):
        return 0 <= i_ep < self.n

    def episode(self, task, i_ep):
        from .lb_synthetic_env import _ToyEnv
        tk_idx = self.env_list.task_to_task_idx[task]
        rng = np.random.RandomState((tk_idx * 100003 + i_ep) % (2 ** 31))
        env = _ToyEnv(tk_idx, 1000 + i_ep, self.env_list.image_hw)
        T = int(rng.randint(self.ep_len[0], self.ep_len[1] + 1))
        acts = rng.uniform(-1, 1, size=(T, 7)).astype(np.float32)
        acts[:, 3:6] *= 0.1
        imgs = [env.render()]
        for a in acts:
            env.step(a)
            imgs.append(env.render())
        return np.stack(imgs), acts


def open_randsam(spec, env_list=None):
    """'synthetic[:N]' | path to .hdf5 | path to an npz directory."""
    if isinstance(spec, str) and spec.startswith('synthetic'):
        n = int(spec.split(':')[1]) if ':' in spec else 500
        return RandSamSynthetic(env_list, episodes_per_task=n)
    if os.path.isdir(spec):
        return RandSamNpzDir(spec)
    return RandSamH5(spec)


def convert_h5_to_npz(h5_path, out_root):
    src = RandSamH5(h5_path)
    for task in src.members("/"):
        for ep in src.members(task):
            g = f"{task}/{ep}"
            ee = src.read(g + "/ee_poses") if "ee_poses" in src.members(g) else None
            RandSamNpzDir.write_episode(out_root, task, int(ep), src.read(g + "/agentview_image"), src.read(g + "/action"), ee)
