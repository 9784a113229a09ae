from .flooring import add_flooring, identity, max_flooring

__all__ = ["identity", "max_flooring", "add_flooring"]
