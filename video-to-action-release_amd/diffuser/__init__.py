"""Drop-in `diffuser` package surface for the hot path (MI355X-native).  Only the modules on the path named by
BASELINE.json's north_star exist here; simulator / trainer glue stays with the user's tree."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)      # the rest of the package comes from the user's checkout
