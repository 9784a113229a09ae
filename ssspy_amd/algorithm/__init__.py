from .projection_back import projection_back

__all__ = ["projection_back"]
