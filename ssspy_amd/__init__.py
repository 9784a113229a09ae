"""ssspy_amd: the iterative frequency-domain demixing hot path of ssspy on MI355X.

Separator classes that keep the ``ssspy.bss`` API (``GaussILRMA``, ``AuxIVA`` family,
``FastGaussMNMF``) and run every per-iteration computation in hand-written HIP kernels
(``libssspy_amd.so``, C ABI in ``include/ssspy_amd.h``).  No CPU fallback: without the
library or a HIP device the separators raise.
"""

__version__ = "0.1.0"

from . import bss  # noqa: F401

__all__ = ["bss", "__version__"]
