"""Overlay package: modules of the MI355X-native hot path; every other module of this package comes from the user's checkout."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
