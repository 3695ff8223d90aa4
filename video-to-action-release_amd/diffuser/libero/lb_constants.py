"""Task table and grasp-descent ranges of the released Libero 8-task split (values: diffuser/libero/lb_constants.py:2-24)."""
from .lb_synthetic_env import LB_TASKS_65to72 as LB_65to72  # noqa: F401

LB_GRASP_actdown_value_range_1 = {idx: ((-0.99, -0.98) if idx in (69, 70) else (-0.11, -0.10)) for idx in range(65, 73)}
