"""ssspy_amd: the iterative frequency-domain demixing hot path of ssspy on MI355X.

Separator classes that keep the ``ssspy.bss`` API (``GaussILRMA``, ``AuxIVA`` family,
``FastGaussMNMF``) and run every per-iteration computation in hand-written HIP kernels
(``libssspy_amd.so``, C ABI in ``include/ssspy_amd.h``).  No CPU fallback: without the
library or a HIP device the separators raise.
"""

__version__ = "0.1.0"

import os as _os

# The ROCm runtime sizes a queue's scratch (register-spill) memory by the first kernel that needs
# any; a later kernel that needs more than HSA_SCRATCH_SINGLE_LIMIT (140 MB by default -- 280 bytes
# per lane on 256 CUs) then gets "use once" scratch, allocated and released around EVERY dispatch:
# measured 4.7 ms per GaussMNMF iteration at 8 channels when a 4-channel separator had run before
# it in the process (7.9 against 3.2 ms; profiles/r04_gmnmf_channels.txt).  The kernels concerned
# are the one-matrix-per-lane fallbacks for 5-8 channels (GaussMNMF, IP2, IPA, eigh: up to 7 KB per
# lane).  Raising the limit lets the queue keep the larger allocation (<= 3.7 GB).  Read by the
# runtime when HIP initialises, i.e. effective when this package is imported before the first
# torch.cuda call; a value already in the environment is left alone.
_os.environ.setdefault("HSA_SCRATCH_SINGLE_LIMIT", str(8 << 30))

from . import bss  # noqa: F401,E402

__all__ = ["bss", "__version__"]
