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
    assert lib.el_abi_version() == 8
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


def test_library_exports_nothing_but_the_declared_entry_points():
    """The boundary is exactly include/*.h: no el_* symbol is exported that the header does not declare."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    if not os.path.exists(nm):
        pytest.skip("no nm on this box")
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1] for line in out.splitlines() if " T el_" in line})
    assert exported == declared_functions(), (set(exported) ^ set(declared_functions()))


def test_header_is_plain_c():
    """`extern "C"`, plain pointers and sizes: the header has to compile as C99 on its own."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this box")
    inc = os.path.join(REPO, "include")
    for f in sorted(os.listdir(inc)):
        if f.endswith(".h"):
            subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, f)],
                           check=True)


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """Every state struct of include/elliot_hip.h against its ctypes mirror in elliot_amd/_lib.py: same size, every field at the
    same offset (a C program built from the header prints offsetof / sizeof; the structs grow at their end with the ABI version,
    and a field added on one side only would silently shift everything behind it)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    pairs = {"el_bprmf_state": _lib.BprmfState, "el_bprsgd_state": _lib.BprsgdState, "el_vae_state": _lib.VaeState,
             "el_nmf_state": _lib.NmfState, "el_pwmf_state": _lib.PwmfState, "el_graph_csr": _lib.GraphCsr,
             "el_mf2020_state": _lib.Mf2020State}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "elliot_hip.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'    printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'    printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["    return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(out[cname]) == ctypes.sizeof(cls), (cname, out[cname], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_python_constants_match_the_header_enums(tmp_path):
    """Every EL_* integer constant of elliot_amd/_lib.py against the enumerator / macro of the same name in include/elliot_hip.h
    (a C program built from the header prints them): a flag renumbered on one side only would select another kernel silently."""
    import re
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    names = sorted(n for n in dir(_lib) if re.fullmatch(r"EL_[A-Z0-9_]+", n) and isinstance(getattr(_lib, n), int))
    header = open(os.path.join(REPO, "include", "elliot_hip.h")).read()
    names = [n for n in names if re.search(r"\b" + n + r"\b", header)]
    assert {"EL_NMF_SCREEN", "EL_TOPK_ITEMS_UNCHANGED", "EL_TOPK_SCREEN", "EL_OPT_ADAM_TF_DENSE", "EL_ABI_VERSION"} <= set(names) | {"EL_ABI_VERSION"}
    lines = ['#include <stdio.h>', '#include "elliot_hip.h"', "int main(void) {"]
    lines += [f'    printf("{n} %lld\\n", (long long)({n}));' for n in names]
    lines += ["    return 0;", "}"]
    src = tmp_path / "consts.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "consts"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n in names:
        assert int(out[n]) == getattr(_lib, n), (n, out[n], getattr(_lib, n))
