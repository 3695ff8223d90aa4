"""The native HDF5 reader (csrc/h5read.hip, SURVEY 8f rank 2 / VERDICT r1 #7) -- host code, runs without a GPU.

Pinned three ways: (1) against a file the HDF5 LIBRARY itself wrote (tests/golden/hdf5lib_sample.mat: the MATLAB-7.3 sample of
scipy's test data, BSD-licensed -- superblock 0 behind a 512-byte user block, symbol-table root group, one float64 dataset holding
0 : pi/4 : 2 pi); (2) round trips through tools/h5_min_writer.py, which emits the structures h5py's defaults produce for the layout of
environment/libero/lb_data/lb_randsam.py:84-104 (multi-level group B-trees, attributes, float64 actions, chunked variant);
(3) refusal of what it does not implement.  Plus the loader's +-0.012 range assertion and clip (lb_online_trainer_v7.py:749-752)."""
import os
import struct
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.h5_min_writer import write_h5, write_randsam_file  # noqa: E402


def test_reads_a_file_written_by_the_hdf5_library(golden_dir):
    from diffuser.libero.lb_randsam_io import RandSamH5
    f = RandSamH5(os.path.join(golden_dir, "hdf5lib_sample.mat"))
    assert f.members("/") == ["testdouble"]
    a = f.read("testdouble")
    assert a.dtype == np.float64 and a.shape == (9, 1)
    assert np.array_equal(a[:, 0], np.arange(9) * (np.pi / 4))
    assert f._lib.v2a_h5_exists(f._h, b"nothing_here") == 0
    with pytest.raises(KeyError):
        f.read("nothing_here")


def _episodes(rng, tasks, per_task, hw=(128, 128)):
    out = {}
    for t in tasks:
        eps = []
        for i in range(per_task):
            T = int(rng.integers(30, 45))
            imgs = rng.integers(0, 256, (T + 1,) + hw + (3,), dtype=np.uint8)
            acts = rng.uniform(-1, 1, (T, 7))
            acts[:, 3:6] = rng.uniform(-0.111, 0.111, (T, 3))             # the generator's orientation range (reference :743-744)
            eps.append((imgs, acts, rng.normal(size=(T + 1, 3)), 100 + i))
        out[t] = eps
    return out


@pytest.mark.parametrize("chunked", [False, True])
def test_randsam_layout_round_trip(tmp_path, chunked):
    from diffuser.libero.lb_randsam_io import RandSamH5, open_randsam
    rng = np.random.default_rng(5)
    tasks = ["KITCHEN_SCENE10_close_the_top_drawer_of_the_cabinet", "LIVING_ROOM_SCENE5_put_the_red_mug_on_the_left_plate"]
    eps = _episodes(rng, tasks, 3, hw=(32, 32))
    path = str(tmp_path / "lb_randsam.hdf5")
    write_randsam_file(path, eps, chunked=chunked)
    rd = open_randsam(path)
    assert isinstance(rd, RandSamH5) and sorted(rd.members("/")) == sorted(tasks)
    for t in tasks:
        assert rd.num_episodes(t) == 3 and rd.has(t, 2) and not rd.has(t, 3)
        assert sorted(rd.members(f"{t}/1")) == ["action", "agentview_image", "ee_poses"]
        for i, (imgs, acts, ee, _) in enumerate(eps[t]):
            gi, ga = rd.episode(t, i)
            assert gi.dtype == np.uint8 and np.array_equal(gi, imgs)
            assert ga.dtype == np.float64 and np.array_equal(ga, acts)
            assert np.array_equal(rd.read(f"{t}/{i}/ee_poses"), ee)


def test_group_with_500_members_spans_b_tree_levels(tmp_path):
    """500 episodes per task (lb_randsam_8tk_perTk500.hdf5) = 63 symbol-table nodes under a two-level B-tree."""
    from diffuser.libero.lb_randsam_io import RandSamH5
    tree = {"task": {str(i): {"action": np.full((2, 7), float(i)), "tag": np.array([i], dtype=np.int64)} for i in range(500)}}
    path = str(tmp_path / "many.hdf5")
    write_h5(path, tree)
    rd = RandSamH5(path)
    assert rd.num_episodes("task") == 500 and len(rd.members("task")) == 500
    for i in (0, 7, 8, 255, 256, 499):
        assert float(rd.read(f"task/{i}/action")[1, 3]) == float(i) and int(rd.read(f"task/{i}/tag")[0]) == i
    assert rd.read("task/3/tag").dtype == np.int64


def test_unsupported_features_are_refused_not_guessed(tmp_path):
    from diffuser.libero.lb_randsam_io import RandSamH5
    with pytest.raises(FileNotFoundError):
        RandSamH5(str(tmp_path / "missing.hdf5"))
    p = tmp_path / "not_hdf5.bin"
    p.write_bytes(b"\0" * 4096)
    with pytest.raises(OSError, match="signature"):
        RandSamH5(str(p))
    good = tmp_path / "good.hdf5"
    write_h5(str(good), {"d": np.arange(12, dtype=np.float32).reshape(3, 4)})
    raw = bytearray(good.read_bytes())
    v2 = bytearray(raw)
    v2[8] = 2                                                  # superblock version 2 (libver='latest')
    (tmp_path / "v2.hdf5").write_bytes(bytes(v2))
    with pytest.raises(OSError, match="superblock version 2"):
        RandSamH5(str(tmp_path / "v2.hdf5"))
    # a filter-pipeline message (gzip) in the dataset header: overwrite the fill-value message (type 5) with type 0x0B, 1 filter
    i = raw.find(struct.pack("<HHB3x", 0x05, 8, 0))
    assert i > 0
    raw[i:i + 2] = struct.pack("<H", 0x0B)
    raw[i + 8:i + 10] = bytes([1, 1])
    (tmp_path / "gz.hdf5").write_bytes(bytes(raw))
    with pytest.raises(KeyError, match="filter pipeline"):
        RandSamH5(str(tmp_path / "gz.hdf5")).read("d")
    assert np.array_equal(RandSamH5(str(good)).read("d"), np.arange(12, dtype=np.float32).reshape(3, 4))
    big = tmp_path / "be.hdf5"
    raw = bytearray(good.read_bytes())
    j = raw.find(struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4))
    raw[j + 1] |= 1                                            # byte-order bit: big endian
    big.write_bytes(bytes(raw))
    with pytest.raises(KeyError, match="big-endian"):
        RandSamH5(str(big)).read("d")


def test_action_range_assert_and_clip():
    from diffuser.libero.lb_randsam_io import check_and_clip_actions
    lo, hi = -np.ones(7, np.float32), np.ones(7, np.float32)
    a = np.zeros((5, 7))
    a[0, 4], a[1, 2], a[2, 6] = 1.011, -1.0119, 0.5
    out = check_and_clip_actions(a, lo, hi)
    assert out.dtype == np.float32 and out[0, 4] == 1.0 and out[1, 2] == -1.0 and out[2, 6] == 0.5
    for bad in (1.012, 1.5):
        b = a.copy()
        b[3, 3] = bad
        with pytest.raises(AssertionError):
            check_and_clip_actions(b, lo, hi)
    b = a.copy()
    b[3, 0] = -1.0121
    with pytest.raises(AssertionError):
        check_and_clip_actions(b, lo, hi)


@pytest.mark.gpu
def test_h5_episodes_reach_the_hbm_pool_as_uint8(tmp_path):
    """File -> native reader -> range check + clip -> uint8 frames in the HBM replay pool -> one gather (the trainer's path:
    LB_Online_Trainer_V7.h5_add_rand_act_episodes_to_Buf)."""
    import torch
    from diffuser.libero.lb_randsam_io import open_randsam, check_and_clip_actions
    from v2a_hip.replay import ReplayStore
    rng = np.random.default_rng(9)
    eps = _episodes(rng, ["t0", "t1"], 2)
    eps["t1"][0][1][3, 4] = 0.1109 * 9 + 0.01                   # 1.0081: inside the slack, must come out clipped to 1.0
    path = str(tmp_path / "rs.hdf5")
    write_randsam_file(path, eps)
    rd = open_randsam(path)
    store = ReplayStore(16, 700, 30, capacity_frames=4 * 50)
    lo, hi = -np.ones(7, np.float32), np.ones(7, np.float32)
    for t in ("t0", "t1"):
        for i in range(rd.num_episodes(t)):
            imgs, acts = rd.episode(t, i)
            store.add_one_episode(t, "agentview", 0, torch.from_numpy(imgs), torch.from_numpy(check_and_clip_actions(acts, lo, hi)))
    assert store.frames.dtype == torch.uint8 and len(store) == 4
    o0, o1, oa = store.gather(np.array([2, 0]), np.array([1, 5]))
    src = eps["t1"][0]
    assert torch.equal(o0[0].cpu(), torch.from_numpy(src[0][1]).permute(2, 0, 1).float() / 255.0)
    assert torch.equal(o1[0].cpu(), torch.from_numpy(src[0][17]).permute(2, 0, 1).float() / 255.0)
    want = np.clip(src[1][1:17], -1, 1).astype(np.float32)
    assert np.array_equal(oa[0].cpu().numpy(), want) and oa[0, 2, 4].item() == 1.0
