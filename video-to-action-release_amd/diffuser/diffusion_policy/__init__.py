"""`diffuser.diffusion_policy` of the MI355X-native package (reference: diffuser/diffusion_policy/__init__.py re-exports the same two
names); modules not provided here come from the user's checkout."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from .get_dp import Init_Diffusion_Policy  # noqa: F401
from .diffusion_unet_image_policy import DiffusionUnetImagePolicy  # noqa: F401
