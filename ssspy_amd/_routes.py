"""Route selection of the separators: which of two equivalent formulations an iteration takes.

Every entry has ONE production value; the other exists because it is the fallback the production
route returns to at run time (the on-Y iterations behind the implied-filter guard, the plain passes
for shapes without a hand-over buffer) and the tests drive it directly to compare the two.  These
are not environment switches: ``override()`` is a context manager for tests and benchmarks
(process-wide: the separators are not re-entrant per instance anyway).
"""

import contextlib

_DEFAULTS = {
    # ISS / ISS2 / IPA iterations read the mixture through the filters their updates imply
    # (bss/ilrma.py: _update_spatial_model_implied, bss/iva.py: _update_once_implied); False: the
    # reference's iterations on Y -- the route the rounding guard falls back to
    "implied_filter": True,
    # ILRMA ISS2 / IPA on Y with the power normalisation folded into the update matrix; False: the
    # literal update -> mean |y|^2 -> y / psi passes (what non-stock subclasses and
    # normalization="projection_back" run)
    "folded_norm": True,
    # ILRMA ISS1 on per-bin statistics: None = by size (batches), True / False force either
    "iss1_statistics": None,
    # FastGaussMNMF keeps |Q x|^2 between the spatial and the NMF passes; False: every pass
    # recomputes it (what shapes without a hand-over buffer run)
    "handover": True,
}
VALUES = dict(_DEFAULTS)  # (tests: monkeypatch.setitem(_routes.VALUES, "implied_filter", False))


def get(name):
    return VALUES[name]


@contextlib.contextmanager
def override(**values):
    unknown = set(values) - set(_DEFAULTS)
    if unknown:
        raise KeyError("unknown route(s): {}".format(sorted(unknown)))
    before = dict(VALUES)
    VALUES.update(values)
    try:
        yield
    finally:
        VALUES.clear()
        VALUES.update(before)
