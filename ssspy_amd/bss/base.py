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

    def __init__(
        self,
        callbacks: Optional[Union[Callable, List[Callable]]] = None,
        record_loss: bool = True,
    ) -> None:
        if callbacks is None:
            self.callbacks = None
        elif callable(callbacks):
            self.callbacks = [callbacks]
        else:
            self.callbacks = callbacks
        self.record_loss = record_loss
        self.loss = [] if record_loss else None

    def _after_step(self) -> None:
        if self.record_loss:
            self.loss.append(self.compute_loss())
        if self.callbacks is not None:
            for callback in self.callbacks:
                callback(self)

    def __call__(self, *args, n_iter: int = 100, initial_call: bool = True, **kwargs) -> np.ndarray:
        if initial_call:
            self._after_step()
        for _ in range(n_iter):
            self.update_once()
            self._after_step()

    def update_once(self) -> None:
        raise NotImplementedError("Implement 'update_once' method.")

    def compute_loss(self) -> float:
        raise NotImplementedError("Implement 'compute_loss' method.")
