from .get_dp import Init_Diffusion_Policy  # noqa: F401
from .diffusion_unet_image_policy import DiffusionUnetImagePolicy  # noqa: F401
