#!/usr/bin/env python3
"""update_by_iss2 against the oracle WITH the phases of the 2 x 2 eigenvectors (csrc/eigh2.hpp restates
LAPACK's convention): one pair agrees to rounding; the same two sources updated twice in a row do not
-- the second problem is already diagonal, its off-diagonal entry is rounding noise and so is the
phase np.linalg.eigh derives from it (the reference's own output there depends on rounding).
python benchmarks/tools/pair_phase_check.py"""
import sys, numpy as np, warnings
sys.path.insert(0,'.')
from ssspy_amd.bss._update_spatial_model import update_by_iss2, update_by_ip2
import oracle.spatial as osp
rng=np.random.default_rng(0)
for N in (2,3):
    F,T=9,7 if N==2 else 12
    Y=rng.standard_normal((N,F,T))+1j*rng.standard_normal((N,F,T))
    w=1/(rng.random((N,F,T))+0.1)
    for pairs in ([(0,1)],[(1,0)],[(0,1),(1,0)]):
        sel=lambda n, p=pairs: iter(p)
        a=update_by_iss2(Y,w,pair_selector=sel)
        b=osp.update_by_iss2(Y,w,pairs=pairs) if 'pairs' in osp.update_by_iss2.__code__.co_varnames else None
        if b is None: print('oracle signature', osp.update_by_iss2.__code__.co_varnames); break
        r=a/b
        print(N,pairs,'rel',np.linalg.norm(a-b)/np.linalg.norm(b),'phase',np.round(r[:,0,0],3))
