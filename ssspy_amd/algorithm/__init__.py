from .minimal_distortion_principle import minimal_distortion_principle
from .projection_back import projection_back

__all__ = ["projection_back", "minimal_distortion_principle"]
