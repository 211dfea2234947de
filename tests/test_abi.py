"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares."""
import ctypes
import os
import re

import pytest

from elliot_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    inc = os.path.join(REPO, "include")
    for f in sorted(os.listdir(inc)):
        if not f.endswith(".h"):
            continue
        src = open(os.path.join(inc, f)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(el_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_functions():
    names = declared_functions()
    assert "el_score_topk" in names and "el_bprmf_train_step" in names and len(names) >= 12


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libelliot_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_binding_matches_header():
    declared = set(declared_functions())
    bound = set(_lib.PROTOTYPES)
    assert declared == bound, (declared - bound, bound - declared)


def test_abi_version_and_error_string():
    lib = _lib.load()
    assert lib.el_abi_version() == 3
    assert isinstance(lib.el_last_error(), bytes)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from elliot_amd import ops
    with pytest.raises(_lib.ElliotHipError):
        ops.Context(0)


def test_product_code_never_imports_oracle():
    pkg = os.path.join(REPO, "elliot_amd")
    bad = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(root, f))
    assert not bad, f"product files import the oracle: {bad}"
