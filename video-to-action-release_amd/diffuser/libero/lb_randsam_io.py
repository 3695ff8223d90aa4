"""Random-action episode files: readers for the layout written by
environment/libero/lb_data/lb_randsam.py:84-104 --

    {task}/{episode}/agentview_image  uint8  [T+1,128,128,3]
    {task}/{episode}/action           float  [T,7]
    {task}/{episode}/ee_poses         float  [T+1,3]          attrs: env_seed, env_list_name

-- feeding the HBM-resident uint8 replay store directly (the reference re-expands every frame to an fp32 CPU tensor,
lb_online_trainer_v7.py:718-780).  Three back ends behind one interface: HDF5 (needs h5py, like the reference), a directory of
`.npz` episodes with the same keys (for machines without h5py; `convert_h5_to_npz` writes it), and a synthetic generator that
rolls uniform random actions in the toy environment (benchmarks / tests)."""
import os
import numpy as np


class RandSamH5:
    def __init__(self, path):
        import h5py                                  # not installed in every image: raised at construction, loudly
        self._h5py = h5py
        self.path = path
        if not os.path.exists(path):
            raise FileNotFoundError(path)

    def num_episodes(self, task):
        with self._h5py.File(self.path, 'r') as f:
            return sorted(int(k) for k in f[f'{task}'].keys())[-1] + 1      # lb_online_trainer_v7.py:246-252

    def has(self, task, i_ep):
        with self._h5py.File(self.path, 'r') as f:
            return f'{task}/{i_ep}' in f

    def episode(self, task, i_ep):
        with self._h5py.File(self.path, 'r') as f:
            g = f[f'{task}/{i_ep}']
            return g['agentview_image'][:], g['action'][:]


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
    with src._h5py.File(h5_path, 'r') as f:
        for task in f.keys():
            for ep in f[task].keys():
                g = f[f'{task}/{ep}']
                RandSamNpzDir.write_episode(out_root, task, int(ep), g['agentview_image'][:], g['action'][:],
                                            g['ee_poses'][:] if 'ee_poses' in g else None, int(g.attrs.get('env_seed', 0)))
