"""Build libssspy_amd.so (HIP, gfx950) in-tree with hipcc.

``python -m ssspy_amd._build`` or ``ssspy_amd._build.build()``.  hipcc cross-compiles
for gfx950 without a GPU.  Objects go to ``build/`` (git-ignored), the shared library to
``ssspy_amd/lib/libssspy_amd.so`` (git-ignored, but shipped to the GPU box by gpurun).
The per-N instantiations of the MFMA kernels are separate translation units so the build
runs in parallel.
"""

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ_DIR = os.path.join(ROOT, "build", "obj")
LIB_PATH = os.path.join(PKG, "lib", "libssspy_amd.so")

ARCH = "gfx950"
# --offload-compress: the code objects are stored compressed in the fat binary (45 -> ~12 MB; the HIP
# runtime inflates them at load)
CXXFLAGS = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "--offload-compress",
            "-I" + INCLUDE, "-I" + CSRC] + os.environ.get("SSSPY_AMD_EXTRA_CXXFLAGS", "").split()

ILRMA_N = list(range(2, 9))
ILRMA_FAST_N = [2, 3, 4]
MNMF_N = [2, 3, 4]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def _units():
    """(source, object name, extra flags)"""
    units = [
        ("spatial_kernels.hip", "spatial_kernels.o", []),
        ("ilrma_api.hip", "ilrma_api.o", []),
        ("iva_kernels.hip", "iva_kernels.o", []),
        ("iss_fused.hip", "iss_fused.o", []),
        ("linalg_kernels.hip", "linalg_kernels.o", []),
        ("pairwise_kernels.hip", "pairwise_kernels.o", []),
        ("gmnmf_kernels.hip", "gmnmf_kernels.o", []),
        ("gmnmf_rows.hip", "gmnmf_rows.o", []),
        ("ipa_kernels.hip", "ipa_kernels.o", []),
        ("ipa_rows.hip", "ipa_rows.o", []),
        ("ipa_rt.hip", "ipa_rt.o", []),
        ("hermitian_rt.hip", "hermitian_rt.o", []),
        ("stft_kernels.hip", "stft_kernels.o", []),
        ("hermitian_ops.hip", "hermitian_ops.o", []),
        ("hermitian_rows.hip", "hermitian_rows.o", []),
        ("fmnmf_generic.hip", "fmnmf_generic.o", []),
        ("wide_cov.hip", "wide_cov.o", []),
        ("wide_n.hip", "wide_n.o", []),
        ("wide_basis.hip", "wide_basis.o", []),
    ]
    units.append(("mnmf_api.hip", "mnmf_api.o", []))
    # Scheduling strategy per unit (measured A / B on one box, benchmarks/tools/ab_lib.sh): hipcc's
    # max-ilp strategy gains 3.6 % on the ILRMA basis pass and 4.6 % on the FastMNMF iteration, loses
    # 6 % on the ILRMA activation pass and 26 % on the fused ISS sweep -- so it is set for exactly the
    # units that gain (the basis pass is its own unit for this: SSSPY_FAST_PART).
    max_ilp = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
    for n in MNMF_N:
        units.append(("mnmf_kernels.hip", "mnmf_kernels_n{}.o".format(n),
                      ["-DSSSPY_N={}".format(n)] + max_ilp))
    for n in ILRMA_N:
        units.append(("ilrma_kernels.hip", "ilrma_kernels_n{}.o".format(n), ["-DSSSPY_N={}".format(n)]))
    for n in ILRMA_FAST_N:
        units.append(("ilrma_fast.hip", "ilrma_fast_basis_n{}.o".format(n),
                      ["-DSSSPY_N={}".format(n), "-DSSSPY_FAST_PART=1"] + max_ilp))
        units.append(("ilrma_fast.hip", "ilrma_fast_n{}.o".format(n),
                      ["-DSSSPY_N={}".format(n), "-DSSSPY_FAST_PART=2"]))
        units.append(("ilrma_small.hip", "ilrma_small_n{}.o".format(n), ["-DSSSPY_N={}".format(n)]))
    # experiments: SSSPY_AMD_UNIT_FLAGS="ilrma_fast_n=-mllvm -x=y;iss_fused=-O2" appends flags to the
    # units whose object name starts with the given prefix (benchmarks/tools/build_variant.sh)
    spec = os.environ.get("SSSPY_AMD_UNIT_FLAGS", "")
    for item in [t for t in spec.split(";") if "=" in t]:
        prefix, flags = item.split("=", 1)
        units = [(s, o, e + flags.split()) if o.startswith(prefix.strip()) else (s, o, e)
                 for s, o, e in units]
    return units


def _digest(paths, flags):
    h = hashlib.sha256(" ".join(flags).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(INCLUDE, "ssspy_amd.h"))
    return hs


def _compile(hipcc, src, obj, extra, verbose):
    src_path = os.path.join(CSRC, src)
    obj_path = os.path.join(OBJ_DIR, obj)
    stamp = obj_path + ".sha"
    digest = _digest([src_path] + _headers(), CXXFLAGS + extra)
    if os.path.exists(obj_path) and os.path.exists(stamp) and open(stamp).read() == digest:
        return obj_path
    cmd = [hipcc] + CXXFLAGS + extra + ["-c", src_path, "-o", obj_path]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for {}:\n{}".format(src, res.stderr))
    with open(stamp, "w") as f:
        f.write(digest)
    return obj_path


def build(verbose=False, jobs=None):
    """Compile every HIP translation unit for gfx950 and link the shared library."""
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    units = _units()
    jobs = jobs or min(len(units), os.cpu_count() or 4)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        futs = [pool.submit(_compile, hipcc, s, o, e, verbose) for s, o, e in units]
        objs = [f.result() for f in futs]
    link_stamp = LIB_PATH + ".sha"
    digest = _digest(objs, [])
    if os.path.exists(LIB_PATH) and os.path.exists(link_stamp) and open(link_stamp).read() == digest:
        return LIB_PATH
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n{}".format(res.stderr))
    with open(link_stamp, "w") as f:
        f.write(digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
