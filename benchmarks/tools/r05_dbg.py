import sys, os, warnings, functools
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
warnings.simplefilter("ignore")
import test_gpu_parity as tp
case = sys.argv[1]
g = tp.load_golden(case)
model = (str(g["meta_model"]), float(g["meta_model_param"])) if "meta_model" in g else ("gauss", None)
for mode in ("implied", "onY"):
    if mode == "onY": os.environ["SSSPY_AMD_NO_IMPLIED_FILTER"] = "1"
    snap = tp.Snap(["output", "basis"])
    m = tp._ilrma_class(model)(n_basis=int(g["meta_n_basis"]), spatial_algorithm=str(g["meta_algo"]),
        domain=float(g["meta_domain"]), flooring_fn=tp._flooring_fn(g), callbacks=snap,
        normalization=tp._option(g["meta_normalization"]), scale_restoration=False, record_loss=True)
    kw = {k: g[k].copy() for k in ("basis", "activation") if k in g}
    kw = {"basis": g["basis0"].copy(), "activation": g["activation0"].copy()} if "basis0" in g else kw
    try:
        m(g["X"], n_iter=int(g["meta_n_iter"]), **kw)
    except Exception as e:
        print("run failed", e); print([k for k in g.keys()][:40]); break
    for key, value in snap.store.items():
        name = key.split("_", 1)[1]
        err = tp.rel_err_up_to_phase(value, g[key], name) if name == "output" else tp.rel_err(value, g[key])
        print(mode, key, "%.2e" % err)
