"""LB_Online_Dataset (diffuser/libero/lb_online_dataset.py:13-92): the (task, camera, env seed) combinations the video-guided
explorer iterates over, plus the reference's own synthetic batch generator `sample_random_tensor`."""
import copy
import random
import numpy as np
import torch
from torch.utils.data import Dataset
from .lb_synthetic_env import SyntheticLiberoEnvList


def _load_env_list(env, target_size):
    if not isinstance(env, str):
        return env
    if env.startswith("synthetic"):
        return SyntheticLiberoEnvList(image_hw=target_size, name=env)
    import diffuser.utils as utils            # the user's checkout (needs the simulator stack): real Libero env lists
    return utils.load_environment(env)


class LB_Online_Dataset(Dataset):
    def __init__(self, env, target_size=(128, 128), dataset_config=None):
        dataset_config = dataset_config or {}
        self.env_list = env_list = _load_env_list(env, target_size)
        self.target_size = tuple(target_size)
        cam_list = env_list.camera_list
        assert list(cam_list) == ['agent']
        self.task_list = copy.deepcopy(env_list.task_list)
        self.cam_list = copy.deepcopy(cam_list)
        self.dataset_config = dataset_config
        self.act_min_max = dataset_config['act_min_max']
        self.action_dim = len(self.act_min_max[0])
        assert (np.asarray(self.act_min_max[0])[3:6] <= -0.1).all()
        assert self.action_dim == 7
        self.combo_type = dataset_config['combo_type']
        if self.combo_type != 'all':
            raise NotImplementedError
        self.combo = [(tk, cam, sd) for tk in self.task_list for cam in cam_list for sd in self.env_list.seed_sets[tk]]

    def __len__(self):
        return len(self.combo)

    def __getitem__(self, idx):
        return self.combo[idx]

    def sample_random_tensor(self, b_size, act_len, device):
        """U[0,1) images and actions of the training shapes (:79-91) -- SURVEY.md 8d's synthetic input generator."""
        task_strs = random.sample(self.env_list.task_list, b_size)
        img1 = torch.rand(b_size, 3, *self.target_size).to(device)
        img2 = torch.rand(b_size, 3, *self.target_size).to(device)
        act_shape = (b_size, act_len, self.action_dim) if act_len else (b_size, self.action_dim)
        act = torch.rand(*act_shape).to(device)
        return [img1, img2, task_strs, act]
