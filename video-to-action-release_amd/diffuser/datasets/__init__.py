"""Import path of the reference's replay-buffer type (diffuser/datasets/env_img_replay_buffer.py)."""
from .env_img_replay_buffer import Global_EnvReplayBuffer_Img  # noqa: F401
