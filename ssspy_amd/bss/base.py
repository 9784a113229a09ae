"""Iterative driver shared by every separator.

Behavioural contract of the reference's ``IterativeMethodBase``
(ssspy/bss/base.py:10-89): optional initial loss/callbacks, then ``n_iter`` rounds of
``update_once()`` -> ``compute_loss()`` (if ``record_loss``) -> callbacks.
"""

from typing import Callable, List, Optional, Union

import numpy as np

__all__ = ["IterativeMethodBase"]


class IterativeMethodBase:
    """Base class of iterative methods.

    Args:
        callbacks: callable or list of callables ``f(method)``; run before the first
            iteration (when ``initial_call``) and after every iteration.
        record_loss: append ``compute_loss()`` to ``self.loss`` at the same points.
    """

    def __init__(self, callbacks: Optional[Union[Callable, List[Callable]]] = None,
                 record_loss: bool = True) -> None:
        # a single callable is wrapped; a list is kept as given (the reference keeps the object)
        self.callbacks = [callbacks] if callable(callbacks) else callbacks
        self.record_loss = bool(record_loss)
        self.loss = [] if self.record_loss else None

    def _after_step(self) -> None:
        """Loss bookkeeping, then the callbacks: the tail of every round and of the initial call."""
        if self.record_loss:
            self.loss.append(self.compute_loss())
        for hook in self.callbacks or ():
            hook(self)

    def __call__(self, *args, n_iter: int = 100, initial_call: bool = True, **kwargs) -> np.ndarray:
        rounds = int(n_iter)
        if initial_call:
            self._after_step()
        for _ in range(rounds):
            self.update_once()
            self._after_step()

    def update_once(self) -> None:
        raise NotImplementedError("Implement 'update_once' method.")

    def compute_loss(self) -> float:
        raise NotImplementedError("Implement 'compute_loss' method.")
