"""Former home of the host helpers; they live in `diffuser.utils`, `diffuser.datasets` and `diffuser.datasets.img_utils` now (the
reference's import paths).  Kept as an alias module for code written against rounds 1-5."""
from diffuser.utils import (print_color, Timer, number_by_ratio, get_latest_epoch, report_parameters, Config, mkdir,  # noqa: F401
                            to_device_tp, set_seed)
from diffuser.datasets import LB_ACTION_MIN, LB_ACTION_MAX                                                            # noqa: F401
from diffuser.datasets.img_utils import imgs_preproc_simple_noCrop_v1                                                 # noqa: F401
