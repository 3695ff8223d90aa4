"""Image pre-processing at the simulator boundary (reference: diffuser/datasets/img_utils.py): uint8 [B,H,W,3] frames -> float [B,3,H,W]
in [0, 1].  The division by 255 is a true division (not a multiplication by 1/255): bit-for-bit what the reference feeds the models."""
import os
import os.path as osp

import numpy as np
import torch


def center_crop_np(images: np.ndarray, size: tuple):
    """[B,H,W,3] -> the centred (size[0], size[1]) window, still [B,h,w,3]."""
    _, H, W, _ = images.shape
    top, left = H // 2 - size[0] // 2, W // 2 - size[1] // 2
    return images[:, top:top + size[0], left:left + size[1], :]


def img_np_toTensor(images: np.ndarray, device=None):
    """uint8 [B,H,W,C] -> float32 [B,C,H,W] in [0,1] (on `device` if given)."""
    assert images.dtype == np.uint8
    t = torch.from_numpy(images.copy()).permute(0, 3, 1, 2).float()
    if device is not None:
        t = t.to(device)
    return t / 255.0


def img_to_n1p1(images: torch.Tensor):
    """[0,1] -> [-1,1]."""
    return images * 2 - 1


def imgs_preproc_simple_v1(imgs, crop_size):
    assert type(imgs) == np.ndarray and imgs.ndim == 4
    return img_np_toTensor(center_crop_np(imgs, crop_size))


def imgs_preproc_simple_noCrop_v1(imgs):
    """Libero frames arrive at the model's resolution: no crop, only layout + scale."""
    assert type(imgs) == np.ndarray and imgs.ndim == 4
    return img_np_toTensor(imgs)


def _out_path(root_dir, sub_dir, tk, c_name, env_idx, ext):
    folder = osp.join(root_dir, sub_dir)
    os.makedirs(folder, exist_ok=True)
    return osp.join(folder, f"{tk.replace(' ', '-')}-{c_name}-{env_idx}.{ext}")


def save_img_tr(img, root_dir, sub_dir, tk: str, c_name, env_idx: int):
    import imageio
    path = _out_path(root_dir, sub_dir, tk, c_name, env_idx, "png")
    imageio.imsave(path, img)
    print(f'[Save png] to {path}')


def save_gif_tr(imgs, root_dir, sub_dir, tk: str, c_name, env_idx: int):
    import imageio
    assert imgs.dtype == np.uint8 and imgs.ndim == 4
    path = _out_path(root_dir, sub_dir, tk, c_name, env_idx, "gif")
    imageio.mimsave(path, imgs, duration=0.5)
    print(f'[Save gif] to {path}')
