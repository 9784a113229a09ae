"""Pair selectors for the pairwise (IP2 / ISS2) updates.  ref: ssspy/utils/select_pair.py:5-76."""

import itertools


def sequential_pair_selector(n_sources, stop=None, step=1, sort=False):
    """Yield (0,1), (1,2), ..., (n_sources-1, 0), optionally strided / sorted."""
    stop = n_sources if stop is None else stop
    for start in range(0, stop, step):
        pair = (start % n_sources, (start + 1) % n_sources)
        yield tuple(sorted(pair)) if sort else pair


def combination_pair_selector(n_sources, sort=False):
    """Yield every unordered pair once, in lexicographic order."""
    for pair in itertools.combinations(range(n_sources), 2):
        yield tuple(sorted(pair)) if sort else pair


def resolve_pairs(pair_selector, n_sources):
    """Materialise a pair selector into the list of (m, n) the kernels walk; negative indices wrap
    as in the reference (ssspy/bss/_update_spatial_model.py:241-244)."""
    if pair_selector is None:
        pair_selector = sequential_pair_selector
    pairs = [(int(m) % n_sources, int(n) % n_sources) for m, n in pair_selector(n_sources)]
    for m, n in pairs:
        if m == n:
            raise ValueError("a pair must name two different sources, got ({}, {})".format(m, n))
    return pairs
