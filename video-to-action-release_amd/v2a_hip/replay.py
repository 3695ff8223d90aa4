"""GPU-resident replay store with the reference's bit-exact index stream.

`ReplayStore` keeps every frame once in HBM ([frames][H][W][3], uint8 when the source is uint8, else fp32) plus the action
rows, and serves `sample_random_batch_seq(B)` like Global_EnvReplayBuffer_Img (env_img_replay_buffer.py:68-116): the episode /
window indices are drawn by the native sampler from the LIVE numpy / CPython generator states (read, advanced natively, written
back), so every other consumer of np.random / random sees exactly the stream it would have seen; the payload is assembled by one
HIP gather.  `sample_mixed` reproduces the 'rand_prob' mixing of sample_from_bufs (lb_online_trainer_v7.py:826-830).
"""
import ctypes
import random
from collections import deque
import numpy as np
import torch
from ._lib import lib, check
from . import ops


def _get_np_state():
    st = np.random.get_state()
    arr = np.empty(625, dtype=np.uint32)
    arr[:624] = st[1]
    arr[624] = st[2]
    return arr, st


def _set_np_state(arr, st):
    np.random.set_state((st[0], arr[:624].copy(), int(arr[624]), st[3], st[4]))


def _get_py_state():
    st = random.getstate()
    return np.array(st[1], dtype=np.uint32), st


def _set_py_state(arr, st):
    random.setstate((st[0], tuple(int(v) for v in arr), st[2]))


def sample_indices(episode_len: np.ndarray, batch: int, act_len: int):
    """One sample_random_batch_seq worth of draws from the live generators.  Returns (episode[B], start[B]) int64."""
    if len(episode_len) == 0:
        raise ValueError("high <= 0")      # np.random.randint(0, 0, size=B) in the reference (env_img_replay_buffer.py:93): empty buffer
    npa, nps = _get_np_state()
    pya, pys = _get_py_state()
    el = np.ascontiguousarray(episode_len, dtype=np.int32)
    ep = np.zeros(batch, dtype=np.int64)
    stt = np.zeros(batch, dtype=np.int64)
    rc = lib.v2a_replay_sample_indices(npa.ctypes.data, pya.ctypes.data, el.ctypes.data, len(el), batch, act_len, ep.ctypes.data,
                                       stt.ctypes.data)
    _set_np_state(npa, nps)
    _set_py_state(pya, pys)
    if rc == -4:
        raise AssertionError("act_seq_len < cur_len violated (episode too short)")
    check(rc, "replay_sample_indices")
    return ep, stt


def count_uniform_below(batch: int, prob: float) -> int:
    npa, nps = _get_np_state()
    n = lib.v2a_replay_count_uniform_below(npa.ctypes.data, batch, float(prob))
    _set_np_state(npa, nps)
    return n


class ReplayStore:
    def __init__(self, max_episodes, max_len, min_len, image_hw=(128, 128), act_dim=7, act_len=16, device="cuda:0",
                 capacity_frames=None, dtype=torch.uint8, pool=None, pool_offset=0):
        """`pool` = (frames, acts) root tensors shared by several stores (see `ReplayStore.pair`): this store then owns rows
        [pool_offset, pool_offset + capacity_frames) of them, and one gather launch can read rows of every store in the pool."""
        self.max_episodes, self.max_len, self.min_len = max_episodes, max_len, min_len
        self.H, self.W = image_hw
        self.act_dim, self.act_len = act_dim, act_len
        self.device = torch.device(device)
        self.dtype = dtype
        cap = capacity_frames or max_episodes * 160
        if pool is None:
            self.root_frames = torch.empty((cap, self.H, self.W, 3), dtype=dtype, device=self.device)
            self.root_acts = torch.zeros((cap, act_dim), dtype=torch.float32, device=self.device)
            pool_offset = 0
        else:
            self.root_frames, self.root_acts = pool
            assert self.root_frames.dtype == dtype and pool_offset + cap <= self.root_frames.shape[0]
        self.base = int(pool_offset)                 # first pool row of this store
        self.frames = self.root_frames[self.base:self.base + cap]
        self.acts = self.root_acts[self.base:self.base + cap]
        self.episodes = deque()          # (offset, length, task, cam, env_idx) in insertion order, FIFO eviction
        self._head = 0                   # ring allocation pointer
        self.cnt_all_history_episodes = 0

    @classmethod
    def pair(cls, max_episodes_a, max_episodes_b, max_len, min_len, capacity_a=None, capacity_b=None, **kw):
        """Two stores over one HBM pool -- the reference's envBuf_rand / envBuf_vid (lb_online_trainer_v7.py:211-215) -- so that a
        mixed minibatch (sample_from_bufs, :787-851) is one gather launch over pool-absolute frame rows."""
        ca = capacity_a or max_episodes_a * 160
        cb = capacity_b or max_episodes_b * 320
        H, W = kw.get("image_hw", (128, 128))
        dev = torch.device(kw.get("device", "cuda:0"))
        dtype = kw.get("dtype", torch.uint8)
        frames = torch.empty((ca + cb, H, W, 3), dtype=dtype, device=dev)
        acts = torch.zeros((ca + cb, kw.get("act_dim", 7)), dtype=torch.float32, device=dev)
        a = cls(max_episodes_a, max_len, min_len, capacity_frames=ca, pool=(frames, acts), pool_offset=0, **kw)
        b = cls(max_episodes_b, max_len, min_len, capacity_frames=cb, pool=(frames, acts), pool_offset=ca, **kw)
        return a, b

    def __len__(self):
        return len(self.episodes)

    def pool_rows(self, ep, start):
        """Pool-absolute first-frame rows of the drawn (episode, start) pairs."""
        return np.array([self.episodes[int(e)][0] for e in ep], dtype=np.int64) + np.asarray(start, dtype=np.int64) + self.base

    def _alloc(self, n):
        cap = self.frames.shape[0]
        if n > cap:
            raise RuntimeError("episode longer than the store")
        if self._head + n > cap:
            self._head = 0
        lo, hi = self._head, self._head + n
        # evict (FIFO) anything overlapping the region we are about to overwrite.  The reference's deque(maxlen) only drops an episode
        # when max_episodes is reached; a frame pool sized below max_episodes * (longest episode) can run out earlier -- say so once
        # instead of diverging silently (size the pool with capacity_frames / frames_per_episode)
        evicted = 0
        while self.episodes and self.episodes[0][0] < hi and self.episodes[0][0] + self.episodes[0][1] > lo:
            self.episodes.popleft()
            evicted += 1
        if evicted:
            self.capacity_evictions = getattr(self, "capacity_evictions", 0) + evicted
            if len(self.episodes) + 1 < self.max_episodes and not getattr(self, "_warned_capacity", False):
                import warnings
                self._warned_capacity = True
                warnings.warn(f"ReplayStore: frame pool ({cap} frames = {cap * self.frames[0].numel() * self.frames.element_size() / 2**30:.1f} GiB) "
                              f"is full with {len(self.episodes)} of max {self.max_episodes} episodes: evicting the oldest by capacity, "
                              f"earlier than the reference's deque(maxlen) would", RuntimeWarning, stacklevel=3)
        self._head = hi
        return lo

    def add_one_episode(self, tk, cam_name, env_idx, imgs, acts, is_suc=False):
        """imgs: uint8 [T+1,H,W,3] (HDF5 layout) / float [T+1,3,H,W] tensor or list of [3,H,W]; acts: [T,act_dim]."""
        if isinstance(imgs, (list, tuple)):
            imgs = torch.stack(list(imgs), 0)
        if isinstance(acts, (list, tuple)):
            acts = torch.stack(list(acts), 0)
        imgs = torch.as_tensor(imgs)
        acts = torch.as_tensor(acts, dtype=torch.float32)
        assert len(imgs) == len(acts) + 1
        n = len(imgs)
        assert self.min_len <= n <= self.max_len
        if imgs.dim() == 4 and imgs.shape[1] == 3 and imgs.shape[-1] != 3:      # CHW float in [0,1] -> HWC store dtype
            imgs = imgs.permute(0, 2, 3, 1)
        if self.dtype == torch.uint8 and imgs.dtype != torch.uint8:
            q = torch.round(imgs.float() * 255.0)
            if not torch.equal(q / 255.0, imgs.float()):
                raise ValueError("float frames are not exactly u8/255: build the store with dtype=torch.float32")
            imgs = q.to(torch.uint8)
        if len(self.episodes) >= self.max_episodes:
            self.episodes.popleft()                                              # deque(maxlen) semantics
        off = self._alloc(n)
        self.frames[off:off + n].copy_(imgs.to(self.device, dtype=self.dtype).contiguous())
        self.acts[off:off + n - 1].copy_(acts.to(self.device))
        self.episodes.append((off, n, tk, cam_name, env_idx))
        self.cnt_all_history_episodes += 1

    def episode_lengths(self):
        return np.array([e[1] for e in self.episodes], dtype=np.int32)

    def gather(self, ep, start, normalize=False, chw=True):
        """Device gather for already drawn indices.  Returns imgs_start, imgs_goal ([B,3,H,W] if chw else [B,H,W,3]), acts [B,L,A]."""
        B = len(ep)
        offs = np.array([self.episodes[int(e)][0] for e in ep], dtype=np.int64) + np.asarray(start, dtype=np.int64)
        fs = torch.from_numpy(offs).to(self.device)
        shape = (B, 3, self.H, self.W) if chw else (B, self.H, self.W, 3)
        o0 = torch.empty(shape, dtype=torch.float32, device=self.device)
        o1 = torch.empty(shape, dtype=torch.float32, device=self.device)
        oa = torch.empty((B, self.act_len, self.act_dim), dtype=torch.float32, device=self.device)
        check(lib.v2a_replay_gather(self.frames.data_ptr(), 1 if self.dtype == torch.uint8 else 0, self.acts.data_ptr(), fs.data_ptr(),
                                    o0.data_ptr(), o1.data_ptr(), oa.data_ptr(), B, self.H, self.W, self.act_len, self.act_dim,
                                    1 if normalize else 0, 1 if chw else 0, ops._stream()), "replay_gather")
        return o0, o1, oa

    def sample_random_batch_seq(self, batch_size):
        """Same return tuple as the reference: (imgs_start, imgs_goal, acts, tasks_str, info) -- tensors stay on the GPU."""
        ep, st = sample_indices(self.episode_lengths(), batch_size, self.act_len)
        if batch_size == 0:
            raise RuntimeError("stack expects a non-empty TensorList")     # what torch.stack([]) raises in the reference
        o0, o1, oa = self.gather(ep, st)
        tasks = [self.episodes[int(e)][2] for e in ep]
        info = dict(env_idxs=np.array([self.episodes[int(e)][4] for e in ep]), cams_str=[self.episodes[int(e)][3] for e in ep],
                    episodes=ep, starts=st)
        return o0, o1, oa, tasks, info


def sample_mixed(buf_rand: ReplayStore, buf_vid: ReplayStore, batch_size: int, rand_prob: float):
    """sample_from_bufs(): rand buffer only while the video buffer is empty, else the 'rand_prob' split (rand rows first)."""
    if len(buf_vid) == 0:
        return buf_rand.sample_random_batch_seq(batch_size)
    if len(buf_rand) == 0:
        return buf_vid.sample_random_batch_seq(batch_size)
    n_rand = count_uniform_below(batch_size, rand_prob)
    r = buf_rand.sample_random_batch_seq(n_rand)
    v = buf_vid.sample_random_batch_seq(batch_size - n_rand)
    info = {k: (np.concatenate([r[4][k], v[4][k]]) if isinstance(r[4][k], np.ndarray) else r[4][k] + v[4][k]) for k in r[4]}
    return torch.cat([r[0], v[0]]), torch.cat([r[1], v[1]]), torch.cat([r[2], v[2]]), r[3] + v[3], info
