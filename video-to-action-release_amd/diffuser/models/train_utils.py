"""Batch-merging and freezing helpers of the joint loop (reference: diffuser/models/train_utils.py; `merge_batch` is what
LB_Online_Trainer_V7.sample_from_bufs ends with, lb_online_trainer_v7.py:560-583).  The reference module imports the CLIP classes at
the top; nothing here needs them."""
import random
from typing import List

import numpy as np
import torch
import torch.nn as nn

from diffuser.utils.textio import print_color


def freeze_model(model: nn.Module):
    assert isinstance(model, nn.Module)
    model.eval()
    for prm in model.parameters():
        prm.requires_grad = False


def freeze_trainer(trainer):
    print_color(f'[freeze_trainer] {type(trainer)}')
    for name, member in trainer.__dict__.items():
        if isinstance(member, torch.nn.Module):
            freeze_model(member)
            print(f'k: {name}, v {member.training}')


def rand_switch_cls_free(video_model, g_w, cls_free_prob):
    """With probability cls_free_prob sample with guidance weight g_w, otherwise unguided (both the online and the EMA model)."""
    w = g_w if random.random() < cls_free_prob else 0
    video_model.model.guidance_weight = w
    video_model.ema.ema_model.guidance_weight = w


def identity_np(t):
    assert isinstance(t, np.ndarray)
    return t


def identity_tensor(t):
    assert torch.is_tensor(t)
    return t


def merge_dicts(dict_list: List[dict]):
    """Dicts with equal keys -> one dict whose values are the per-key concatenations along axis 0 (string arrays come back as lists)."""
    out = {}
    for key in dict_list[0]:
        cat = np.concatenate([np.asarray(d[key]) for d in dict_list], axis=0)
        out[key] = cat.tolist() if cat.dtype in ['U', 'S'] else cat
    return out


def _merge(parts, imgs_preproc_fn, n_img_cols):
    cols = [[] for _ in range(n_img_cols)]
    acts, tasks, infos = [], [], []
    img_idx = (0, 1, 5)[:n_img_cols]
    for dp in parts:
        cols[0].append(imgs_preproc_fn(dp[0]))
        cols[1].append(imgs_preproc_fn(dp[1]))
        if n_img_cols == 3:
            cols[2].append(dp[img_idx[2]])
        acts.append(dp[2])
        tasks.extend(dp[3])
        infos.append(dp[4])
    cat = (lambda xs: torch.cat(xs, dim=0)) if torch.is_tensor(cols[0][0]) else (lambda xs: np.concatenate(xs, axis=0))
    cols = [cat(c) for c in cols]
    acts = torch.cat(acts, dim=0)
    for c in cols:
        assert len(tasks) == len(c)
    return cols, acts, tasks, merge_dicts(infos)


def merge_batch(*args, imgs_preproc_fn=identity_np):
    """(imgs_start, imgs_goal, acts, task strings, info dict) tuples -> one such tuple: images and actions concatenated along the batch
    axis, task lists joined, info dicts merged; `imgs_preproc_fn` is applied to each part's images first."""
    (st, gl), acts, tasks, info = _merge(args, imgs_preproc_fn, 2)
    return st, gl, acts, tasks, info


def merge_batch_grp(*args, imgs_preproc_fn=identity_np):
    """merge_batch for six-tuples that carry a third image stack (the gripper camera) as their last element."""
    (st, gl, grp), acts, tasks, info = _merge(args, imgs_preproc_fn, 3)
    return st, gl, acts, tasks, info, grp


def weighted_uniform_sample(size: tuple, prob_0: float, low, mid, high):
    """Each element uniform in [low, mid) with probability prob_0, else uniform in [mid, high)."""
    first = np.random.rand(*size) < prob_0
    lo = np.random.uniform(low=low, high=mid, size=size)
    hi = np.random.uniform(low=mid, high=high, size=size)
    return np.where(first, lo, hi)
