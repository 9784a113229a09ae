"""ssspy_amd: the iterative frequency-domain demixing hot path of ssspy on MI355X.

Separator classes that keep the ``ssspy.bss`` API (``GaussILRMA``, ``AuxIVA`` family,
``FastGaussMNMF``) and run every per-iteration computation in hand-written HIP kernels
(``libssspy_amd.so``, C ABI in ``include/ssspy_amd.h``).  No CPU fallback: without the
library or a HIP device the separators raise.
"""

__version__ = "0.1.0"

import os as _os

# Kernels that spill (the one-matrix-per-lane fallbacks for 5-8 channels: up to 7 KB of scratch per
# lane) can run into the ROCm runtime's scratch policy: measured on GaussMNMF, a separator of 8
# channels that followed one of 4 or 7 channels in the same process ran its first iterations at
# 6.9-7.9 ms instead of 3.2 (benchmarks/gmnmf_channels.py, profiles/r04_gmnmf_channels.txt); with
# HSA_SCRATCH_SINGLE_LIMIT (default 140 MB per queue, above which scratch is handed out for one
# dispatch at a time) raised the same sequence ran at 3.2 ms.  The mechanism was not pinned down
# further -- a plain update_once() loop does not show it.  The limit is read when HIP initialises,
# so this only acts when the package is imported before the first torch.cuda call; a value already
# in the environment is left alone.
_os.environ.setdefault("HSA_SCRATCH_SINGLE_LIMIT", str(8 << 30))

from . import bss  # noqa: F401,E402

__all__ = ["bss", "__version__"]
