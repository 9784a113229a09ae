from . import base, ilrma, iva, mnmf
from .base import IterativeMethodBase

__all__ = ["IterativeMethodBase", "base", "ilrma", "iva", "mnmf"]
