from . import base, ilrma, iva
from .base import IterativeMethodBase

__all__ = ["IterativeMethodBase", "base", "ilrma", "iva"]
