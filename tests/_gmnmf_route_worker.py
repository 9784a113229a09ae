"""Child process of test_gauss_mnmf_packed_route_equals_full_storage_route: run GaussMNMF for a few
iterations and save the state.  The route (packed per-point kernels, or the full-storage ones with
SSSPY_AMD_GMNMF_FULL=1) is read once per process by the library, hence the subprocess."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssspy_amd.bss.mnmf import GaussMNMF  # noqa: E402


def one(M, N, F, T, K, floor, out):
    rng = np.random.default_rng(7)
    B = 3
    X = rng.standard_normal((B, M, F, T)) + 1j * rng.standard_normal((B, M, F, T))
    if floor == "tiny":  # silent frames and a rank-deficient bin: the eigenvalue floor acts
        X[:, :, :, : T // 4] *= 1e-9
        X[:, 1:, F // 2] = X[:, :1, F // 2]
    from functools import partial
    from ssspy_amd.special.flooring import add_flooring, max_flooring
    fn = {"max": partial(max_flooring, eps=1e-10), "tiny": partial(max_flooring, eps=1e-10),
          "add": partial(add_flooring, eps=1e-8), "none": None}[floor]
    m = GaussMNMF(n_basis=K, n_sources=N, flooring_fn=fn, rng=np.random.default_rng(3))
    Y = m(X, n_iter=4)
    np.savez(out, Y=Y, basis=m.basis, activation=m.activation, spatial=m.spatial,
             loss=np.asarray(m.loss))


def main():
    # argv: out_pattern (with {}), then cases "M,N,F,T,K,floor"
    pattern = sys.argv[1]
    for spec in sys.argv[2:]:
        M, N, F, T, K, floor = spec.split(",")
        one(int(M), int(N), int(F), int(T), int(K), floor, pattern.format(spec.replace(",", "_")))


if __name__ == "__main__":
    main()
