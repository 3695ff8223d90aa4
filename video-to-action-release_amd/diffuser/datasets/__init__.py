"""`diffuser.datasets`: the replay-buffer type of the hot path plus the action / image normalisation limits the configs import from
here (reference: diffuser/datasets/__init__.py -- a table of (min, max) arrays per environment and `*_f()` accessors that append the
broadcast shape).  Other modules of the user's `diffuser/datasets` directory stay importable (the package path is extended over it)."""
from pkgutil import extend_path

import numpy as np

__path__ = extend_path(__path__, __name__)


def _box(lo, hi):
    return np.array(lo, dtype=np.float32), np.array(hi, dtype=np.float32)


# MetaWorld Sawyer: 4-dof actions in [-1, 1]
MW_SAWYER_ACTION_MIN, MW_SAWYER_ACTION_MAX = _box([-1.] * 4, [1.] * 4)
MW_SAWYER_ACTION_LEN = 4
MW_SAWYER_ACTION_MINMAX = (MW_SAWYER_ACTION_MIN, MW_SAWYER_ACTION_MAX)
# images already scaled to [0, 1]
IMAGE_MINMAX_01 = _box([0, 0, 0], [1, 1, 1])
# Libero: 7-dof actions; the plain box, and the variant whose three orientation deltas live in [-0.1, 0.1]
LB_ACTION_LEN = 7
LB_ACTION_MIN, LB_ACTION_MAX = _box([-1.] * 7, [1.] * 7)
LB_ACTION_MINMAX = (LB_ACTION_MIN, LB_ACTION_MAX)
LB_ACTION_MIN_orn01, LB_ACTION_MAX_orn01 = _box([-1.] * 3 + [-0.1] * 3 + [-1.], [1.] * 3 + [0.1] * 3 + [1.])
LB_ACTION_MINMAX_orn01 = (LB_ACTION_MIN_orn01, LB_ACTION_MAX_orn01)
# CLIP task embedding: placeholder limits (never used to normalise)
Task_Embed_MIN, Task_Embed_MAX = _box([0.] * 512, [1.] * 512)
# iTHOR: 4-dim discrete-as-continuous actions
Thor_ACTION_LEN_Dim4 = 4
Thor_ACTION_MIN_Dim4, Thor_ACTION_MAX_Dim4 = _box([-1.] * 4, [1.] * 4)
Thor_ACTION_MINMAX_Dim4 = (Thor_ACTION_MIN_Dim4, Thor_ACTION_MAX_Dim4)
# Calvin: relative (unit box) and absolute (workspace box, widened by 0.01) 7-dof actions
CAL_ACTION_LEN = 7
CAL_ACTION_MIN, CAL_ACTION_MAX = _box([-1.] * 7, [1.] * 7)
CAL_ACTION_MINMAX = (CAL_ACTION_MIN, CAL_ACTION_MAX)
CAL_abs_ACTION_MIN = np.array([-0.20, -0.50, 0.3, -3.15, -0.50, -3.15, -1.], dtype=np.float32) - 0.01
CAL_abs_ACTION_MAX = np.array([0.36, 0.12, 0.70, 3.15, 0.30, 3.15, 1.], dtype=np.float32) + 0.01
CAL_abs_ACTION_MINMAX = (CAL_abs_ACTION_MIN, CAL_abs_ACTION_MAX)
for _lo, _hi, _n in ((LB_ACTION_MIN, LB_ACTION_MAX, LB_ACTION_LEN), (LB_ACTION_MIN_orn01, LB_ACTION_MAX_orn01, LB_ACTION_LEN),
                     (Thor_ACTION_MIN_Dim4, Thor_ACTION_MAX_Dim4, Thor_ACTION_LEN_Dim4), (CAL_ACTION_MIN, CAL_ACTION_MAX, CAL_ACTION_LEN)):
    assert len(_lo) == len(_hi) == _n


def image_minmax_01_f():
    return (*IMAGE_MINMAX_01, [1, 3, 1, 1])


def mw_sawyer_action_minmax_f():
    return (*MW_SAWYER_ACTION_MINMAX, [1, 4])


def lb_action_minmax_f():
    return (*LB_ACTION_MINMAX, [1, 7])


def lb_action_minmax_orn01_f():
    return (*LB_ACTION_MINMAX_orn01, [1, 7])


def tk_emb_minmax_f():
    return (Task_Embed_MIN, Task_Embed_MAX, [1, 512])


def thor_action_minmax_dim4_f():
    return (*Thor_ACTION_MINMAX_Dim4, [1, 4])


def cal_action_minmax_f():
    return (*CAL_ACTION_MINMAX, [1, 7])


def cal_abs_action_minmax_f():
    return (*CAL_abs_ACTION_MINMAX, [1, 7])


def __getattr__(name):
    if name == "Global_EnvReplayBuffer_Img":      # rounds 1-5 exported the class here; importing it needs the HIP library, so do it on demand
        from .env_img_replay_buffer import Global_EnvReplayBuffer_Img
        return Global_EnvReplayBuffer_Img
    raise AttributeError(name)
