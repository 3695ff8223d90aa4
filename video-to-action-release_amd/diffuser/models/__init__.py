
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)      # the rest of the package comes from the user's checkout
