"""Replay index stream: C oracle vs numpy/CPython (pinning), vs golden vectors from the reference's own buffer, and the product's
native sampler vs both (host part runs without a GPU; the gather is covered under -m gpu)."""
import ctypes
import random
import numpy as np
import pytest
import torch


def test_oracle_pinned_against_numpy_and_cpython():
    from oracle import replay as R
    for seed in [0, 1, 123, 2 ** 31 + 5, 4294967295]:
        nps, pys = R.np_seeded(seed), R.py_seeded(seed)
        np.random.seed(seed); random.seed(seed)
        for _ in range(50):
            n_eps = random.randint(1, 1300)
            assert R.lib().py_randint(ctypes.byref(pys), 1, 1300) == n_eps
            lens = np.random.randint(17, 700, size=n_eps)
            tmp = np.zeros(n_eps, dtype=np.int64)
            R.lib().np_randint(ctypes.byref(nps), 17, 700, tmp.ctypes.data, n_eps)
            assert (tmp == lens).all()
            ref_ep = np.random.randint(0, n_eps, size=64)
            ref_st = np.array([random.randint(0, lens[e] - 16 - 1) for e in ref_ep])
            ep, st = R.sample_seq(nps, pys, lens, 64, 16)
            assert (ep == ref_ep).all() and (st == ref_st).all()
            u = np.random.uniform(low=0, high=1, size=(7,))
            u2 = np.zeros(7)
            R.lib().np_uniform(ctypes.byref(nps), 0.0, 1.0, u2.ctypes.data, 7)
            assert (u == u2).all()


def test_oracle_vs_reference_golden(golden_dir):
    """golden replay.npz was produced by the reference's Global_EnvReplayBuffer_Img itself (tools/make_golden.py g_replay)."""
    from oracle import replay as R
    g = np.load(f"{golden_dir}/replay.npz")
    for seed in (0, 123, 9001):
        nps, pys = R.np_seeded(seed), R.py_seeded(seed)
        lens = g[f"lens_{seed}"]
        for it in range(5):
            ep, st = R.sample_seq(nps, pys, lens, 64, 16)
            assert (ep == g[f"episodes_{seed}"][it]).all() and (st == g[f"starts_{seed}"][it]).all()


def test_product_sampler_consumes_live_generators_bit_exactly(golden_dir):
    """v2a_hip.replay.sample_indices must (a) reproduce the golden stream and (b) leave np.random / random in exactly the state
    the reference would have left them in."""
    pytest.importorskip("v2a_hip")
    from v2a_hip.replay import sample_indices, count_uniform_below
    g = np.load(f"{golden_dir}/replay.npz")
    for seed in (0, 123, 9001):
        lens = g[f"lens_{seed}"]
        np.random.seed(seed); random.seed(seed)
        for it in range(5):
            ep, st = sample_indices(lens, 64, 16)
            assert (ep == g[f"episodes_{seed}"][it]).all() and (st == g[f"starts_{seed}"][it]).all()
        after_np, after_py = np.random.randint(0, 10 ** 6), random.random()
        np.random.seed(seed); random.seed(seed)
        for it in range(5):
            e = np.random.randint(0, len(lens), size=64)
            [random.randint(0, lens[i] - 16 - 1) for i in e]
        assert after_np == np.random.randint(0, 10 ** 6) and after_py == random.random()
    np.random.seed(77)
    probs = np.random.uniform(low=0, high=1, size=(64,))
    np.random.seed(77)
    assert count_uniform_below(64, 0.3) == int((probs < 0.3).sum())
    # edge cases: a single episode draws nothing from numpy; too-short episode asserts like the reference
    np.random.seed(1); random.seed(1)
    s0 = np.random.get_state()[2]
    ep, st = sample_indices(np.array([40]), 4, 16)
    assert (ep == 0).all() and np.random.get_state()[2] == s0
    with pytest.raises(AssertionError):
        sample_indices(np.array([16]), 1, 16)


@pytest.mark.gpu
def test_store_gather_matches_reference_semantics(golden_dir):
    from v2a_hip.replay import ReplayStore
    g = np.load(f"{golden_dir}/replay.npz")
    seed = 123
    lens = g[f"lens_{seed}"]
    store = ReplayStore(1200, 700, 30, image_hw=(4, 4), capacity_frames=int(lens.sum()) + 8, dtype=torch.float32)
    for e, L in enumerate(lens):
        imgs = [torch.full((3, 4, 4), float(e * 1000 + i)) for i in range(L)]
        acts = [torch.full((7,), float(e * 1000 + i)) for i in range(L - 1)]
        store.add_one_episode("t", "agentview", e, imgs, acts)
    np.random.seed(seed); random.seed(seed)
    for it in range(5):
        s, gl, a, tasks, info = store.sample_random_batch_seq(64)
        ep = (s[:, 0, 0, 0] // 1000).long().cpu().numpy()
        st = (s[:, 0, 0, 0] % 1000).long().cpu().numpy()
        assert (ep == g[f"episodes_{seed}"][it]).all() and (st == g[f"starts_{seed}"][it]).all()
        assert ((gl[:, 0, 0, 0] % 1000).long().cpu().numpy() == st + 16).all()
        assert s.shape == (64, 3, 4, 4) and a.shape == (64, 16, 7)
        want = torch.from_numpy(ep * 1000 + st)[:, None] + torch.arange(16)[None]
        assert torch.equal(a[:, :, 0].cpu().long(), want)
    # uint8 store: /255 conversion and HWC->CHW
    st8 = ReplayStore(4, 100, 2, image_hw=(4, 4), capacity_frames=64)
    raw = torch.randint(0, 256, (20, 4, 4, 3), dtype=torch.uint8)
    st8.add_one_episode("t", "c", 0, raw, torch.zeros(19, 7))
    o0, o1, _ = st8.gather(np.array([0]), np.array([2]))
    assert torch.equal(o0[0].cpu(), raw[2].permute(2, 0, 1).float() / 255.0)
    assert torch.equal(o1[0].cpu(), raw[18].permute(2, 0, 1).float() / 255.0)


def _decode(codes):
    which = codes // 1000000
    return which, (codes % 1000000) // 1000, codes % 1000


def test_mixed_two_buffer_draw_vs_reference_golden(golden_dir):
    """replay_mixed.npz was produced by the reference's LB_Online_Trainer_V7.sample_from_bufs ('rand_prob', :826-851) + merge_batch
    (train_utils.py:40-74) over two of its own buffers: the oracle's C restatement and the product's native sampler must both give the
    same (n_rand, episode, start) stream -- rand rows first -- and leave numpy's / CPython's generators where the reference left them."""
    from oracle import replay as R
    from v2a_hip.replay import sample_indices, count_uniform_below
    g = np.load(f"{golden_dir}/replay_mixed.npz")
    for seed in (0, 77, 4242):
        lr, lv = g[f"lens_rand_{seed}"], g[f"lens_vid_{seed}"]
        nps, pys = R.np_seeded(seed), R.py_seeded(seed)
        np.random.seed(seed); random.seed(seed)
        for it in range(4):
            which, ep_ref, st_ref = _decode(g[f"codes_{seed}"][it])
            n_ref = int(g[f"n_rand_{seed}"][it])
            assert (which[:n_ref] == 0).all() and (which[n_ref:] == 1).all()
            n, ep, st = R.sample_mixed(nps, pys, lr, lv, 64, 16, 0.3)                    # oracle
            assert n == n_ref and (ep == ep_ref).all() and (st == st_ref).all()
            n2 = count_uniform_below(64, 0.3)                                             # product, on the live generators
            e0, s0 = sample_indices(lr, n2, 16)
            e1, s1 = sample_indices(lv, 64 - n2, 16)
            assert n2 == n_ref
            assert (np.concatenate([e0, e1]) == ep_ref).all() and (np.concatenate([s0, s1]) == st_ref).all()
        assert [np.random.randint(0, 1 << 30), random.randint(0, 1 << 30)] == g[f"tail_{seed}"].tolist()


@pytest.mark.gpu
def test_mixed_draw_payload_vs_reference_golden(golden_dir):
    """sample_mixed over two ReplayStores sharing one HBM pool returns the very rows the reference's sample_from_bufs returned."""
    from v2a_hip.replay import ReplayStore, sample_mixed
    g = np.load(f"{golden_dir}/replay_mixed.npz")
    seed = 77
    lr, lv = g[f"lens_rand_{seed}"], g[f"lens_vid_{seed}"]
    a, b = ReplayStore.pair(1200, 600, 700, 30, capacity_a=int(lr.sum()) + 4, capacity_b=int(lv.sum()) + 4, image_hw=(4, 4),
                            dtype=torch.float32)
    for which, (store, lens) in enumerate(((a, lr), (b, lv))):
        for e, L in enumerate(lens):
            base = which * 1000000 + e * 1000
            # frames carry the code modulo 2^24-exactness is not needed: float32 holds integers < 16.7 M exactly
            store.add_one_episode(f"task{which}_{e}", "agentview", e, [torch.full((3, 4, 4), float(base + i)) for i in range(L)],
                                  [torch.full((7,), float(base + i)) for i in range(L - 1)])
    np.random.seed(seed); random.seed(seed)
    for it in range(4):
        s, gl, act, tasks, info = sample_mixed(a, b, 64, 0.3)
        codes = g[f"codes_{seed}"][it]
        assert (s[:, 0, 0, 0].long().cpu().numpy() == codes).all()
        assert (gl[:, 0, 0, 0].long().cpu().numpy() == codes + 16).all()
        assert torch.equal(act[:, :, 0].cpu().long(), torch.from_numpy(codes)[:, None] + torch.arange(16)[None])
        which, ep, _ = _decode(codes)
        assert tasks == [f"task{w}_{e}" for w, e in zip(which, ep)] and len(info["cams_str"]) == 64
    assert [np.random.randint(0, 1 << 30), random.randint(0, 1 << 30)] == g[f"tail_{seed}"].tolist()


@pytest.mark.gpu
def test_reference_import_path_buffer_is_a_drop_in(golden_dir):
    """`from diffuser.datasets.env_img_replay_buffer import Global_EnvReplayBuffer_Img` with the reference's constructor keywords
    (env_img_replay_buffer.py:12-16) and call sequence, against the reference-made index fixture."""
    from diffuser.datasets.env_img_replay_buffer import Global_EnvReplayBuffer_Img

    class _EnvList:
        camera_list = ["agentview"]

    g = np.load(f"{golden_dir}/replay.npz")
    seed = 9001
    lens = g[f"lens_{seed}"]
    buf = Global_EnvReplayBuffer_Img(task_list=["t"], max_num_unitBufs=1200, max_len_uB=700, min_len_uB=30, env_list=_EnvList(),
                                     render_img_size=(4, 4), env_buf_config={"sample_act_seq_len": 16}, dtype=torch.float32,
                                     capacity_frames=int(lens.sum()) + 8)
    for e, L in enumerate(lens):
        buf.add_one_episode("t", "agentview", e, [torch.full((3, 4, 4), float(e * 1000 + i)) for i in range(L)],
                            [torch.full((7,), float(e * 1000 + i)) for i in range(L - 1)])
    assert len(buf.buffers) == len(buf.bufs_task) == len(buf.bufs_cam) == len(lens) and buf.cnt_all_history_episodes == len(lens)
    np.random.seed(seed); random.seed(seed)
    for it in range(5):
        s, gl, a, tasks, info = buf.sample_random_batch_seq(64)
        assert ((s[:, 0, 0, 0] // 1000).long().cpu().numpy() == g[f"episodes_{seed}"][it]).all()
        assert ((s[:, 0, 0, 0] % 1000).long().cpu().numpy() == g[f"starts_{seed}"][it]).all()
        assert set(info) >= {"env_idxs", "cams_str"} and tasks == ["t"] * 64
    with pytest.raises(KeyError):
        Global_EnvReplayBuffer_Img(["t"], 10, 700, 30, _EnvList(), (4, 4), env_buf_config={})


def test_capacity_eviction_before_max_episodes_warns():
    """a frame pool smaller than max_episodes x episode length evicts by capacity, which the reference's
    deque(maxlen) never does -- the store says so (once) and counts the evictions instead of diverging silently."""
    import warnings
    import torch
    from v2a_hip.replay import ReplayStore
    st = ReplayStore(max_episodes=10, max_len=40, min_len=5, image_hw=(8, 8), capacity_frames=50, device="cpu")
    imgs = torch.zeros(21, 8, 8, 3, dtype=torch.uint8)
    acts = torch.zeros(20, 7)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for i in range(4):
            st.add_one_episode("t", "cam", i, imgs, acts)
    assert len(st) == 2 and st.capacity_evictions == 2
    assert sum(issubclass(x.category, RuntimeWarning) and "evicting the oldest by capacity" in str(x.message) for x in w) == 1


def test_sampling_an_empty_buffer_raises_what_numpy_raises():
    """Global_EnvReplayBuffer_Img.sample_random_batch_seq on an empty buffer fails inside np.random.randint(0, 0, size=B)
    (env_img_replay_buffer.py:93) with ValueError; the index sampler raises the same error type and leaves the generators untouched."""
    import random
    from v2a_hip import replay
    np.random.seed(5)
    random.seed(5)
    s0, p0 = np.random.get_state()[1].copy(), random.getstate()
    with pytest.raises(ValueError):
        replay.sample_indices(np.array([], dtype=np.int32), 4, 16)
    assert (np.random.get_state()[1] == s0).all() and random.getstate() == p0
