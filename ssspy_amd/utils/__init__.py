from .flooring import choose_flooring_fn, device_flooring
from .select_pair import combination_pair_selector, sequential_pair_selector

__all__ = ["choose_flooring_fn", "device_flooring", "sequential_pair_selector",
           "combination_pair_selector"]
