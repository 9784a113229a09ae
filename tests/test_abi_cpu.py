"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/ssspy_amd.h declares, and the Python binding table matches the header.  No compute."""

import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ssspy_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssspy_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = _declared_symbols()
    assert "ssspy_update_by_ip1" in syms and "ssspy_ilrma_ip1_update" in syms
    assert len(syms) >= 20


def test_library_builds_and_exports_every_symbol():
    from ssspy_amd import _build, _lib

    _build.build()
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), "libssspy_amd.so does not export {}".format(name)
    assert lib.ssspy_amd_version().startswith(b"ssspy_amd")


def test_binding_table_matches_header():
    from ssspy_amd import _lib

    assert sorted(_lib.PROTOTYPES) == _declared_symbols()


def test_argument_counts_match_header():
    from ssspy_amd import _lib

    text = open(os.path.join(ROOT, "include", "ssspy_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, (_, argtypes) in _lib.PROTOTYPES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", text, flags=re.S)
        assert m, name
        args = m.group(1).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert n == len(argtypes), "{}: header has {} args, binding {}".format(name, n, len(argtypes))


def test_product_path_fails_loudly_without_device():
    import numpy as np
    import torch

    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from ssspy_amd.bss.ilrma import GaussILRMA

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GaussILRMA(n_basis=2)(np.zeros((2, 4, 8), dtype=complex), n_iter=1)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under ssspy_amd/ may reference it."""
    pkg = os.path.join(ROOT, "ssspy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src and "oracle." not in src.replace("the oracle.", ""), f


def test_host_modules_have_no_undefined_names():
    """Every global name the host-side wrappers refer to exists (the device paths cannot run in the
    CPU suite, so a helper dropped by an edit would otherwise surface only on the GPU box)."""
    import ast
    import builtins
    import importlib
    import pkgutil

    import ssspy_amd

    def bound_names(fn):
        names = {a.arg for a in ast.walk(fn) if isinstance(a, ast.arg)}
        for node in ast.walk(fn):
            if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Store):
                names.add(node.id)
            elif isinstance(node, (ast.FunctionDef, ast.ClassDef)):
                names.add(node.name)
            elif isinstance(node, (ast.Import, ast.ImportFrom)):
                names.update((a.asname or a.name).split(".")[0] for a in node.names)
            elif isinstance(node, ast.ExceptHandler) and node.name:
                names.add(node.name)
        return names

    def visit(node, enclosing, scope, where):
        for child in ast.iter_child_nodes(node):
            if isinstance(child, (ast.FunctionDef, ast.Lambda)):
                visible = enclosing | bound_names(child)
                for sub in ast.walk(child):
                    if isinstance(sub, ast.Name) and isinstance(sub.ctx, ast.Load):
                        assert sub.id in visible or sub.id in scope, \
                            "{}: undefined name {!r} (line {})".format(where, sub.id, sub.lineno)
                visit(child, visible, scope, where)
            else:
                visit(child, enclosing, scope, where)

    for info in pkgutil.walk_packages(ssspy_amd.__path__, "ssspy_amd."):
        if info.name.endswith("_build"):
            continue
        mod = importlib.import_module(info.name)
        tree = ast.parse(open(mod.__file__).read())
        visit(tree, set(), set(dir(mod)) | set(dir(builtins)), info.name)
