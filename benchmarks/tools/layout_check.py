#!/usr/bin/env python3
"""The same mixture handed over C-ordered, Fortran-ordered, as a transposed view, as complex64: the
separators give the same output for the first three (bit for bit) and the complex64 values promoted
to complex128 for the last (1e-7: the input's own rounding).  python benchmarks/tools/layout_check.py"""
import sys, numpy as np, warnings
sys.path.insert(0,'.')
from ssspy_amd.bss.ilrma import GaussILRMA
from ssspy_amd.bss.iva import AuxLaplaceIVA
from ssspy_amd.bss.mnmf import FastGaussMNMF
from ssspy_amd.utils.dataset import nmf_mixture
warnings.simplefilter("ignore")
X=nmf_mixture(3,3,33,40)
forms={"C":X,"F":np.asfortranarray(X),"view":np.ascontiguousarray(X.transpose(2,1,0)).transpose(2,1,0),"c64":X.astype(np.complex64),
       "strided":np.concatenate([X,X],axis=-1)[..., ::2][..., :40] if False else X[:, :, ::1]}
def run(cls,Xin,**kw):
    m=cls(**kw); return np.asarray(m(Xin,n_iter=4)), np.asarray(m.loss)
for cls,kw in ((GaussILRMA,dict(n_basis=3,rng=np.random.default_rng(0))),(AuxLaplaceIVA,dict(spatial_algorithm="ISS")),(FastGaussMNMF,dict(n_basis=3,rng=np.random.default_rng(0)))):
    base=None
    for name,Xin in forms.items():
        kw2=dict(kw)
        if "rng" in kw2: kw2["rng"]=np.random.default_rng(0)
        Y,l=run(cls,Xin,**kw2)
        if base is None: base=(Y,l)
        print(cls.__name__,name,Y.dtype,"%.2e"%(np.abs(Y-base[0]).max()/np.abs(base[0]).max()))
# batch of mixtures given as a list? and 4-d
Xb=np.stack([nmf_mixture(3+i,3,33,40) for i in range(3)])
m=GaussILRMA(n_basis=3,rng=np.random.default_rng(0)); Yb=m(Xb,n_iter=4); print("batch",np.asarray(Yb).shape)
