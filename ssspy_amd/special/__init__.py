from .flooring import add_flooring, identity, max_flooring
from .psd import to_psd

__all__ = ["identity", "max_flooring", "add_flooring", "to_psd"]
