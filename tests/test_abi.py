"""The C-ABI library builds for gfx950, loads, and exports every symbol that
include/scasr.h declares (no compute calls: no GPU needed)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "scasr.h").read_text()
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from speechcatcher_amd import _abi
    if not _abi.LIB_PATH.exists():
        _abi.build()
    lib = _abi.load()
    declared = _declared_symbols()
    assert declared, "no symbols parsed from scasr.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in scasr.h but not exported"
    assert set(declared) == set(_abi.EXPORTED_SYMBOLS)
    assert lib.sc_version() == _abi.ABI_VERSION   # = SC_ABI_VERSION of include/scasr.h
    hdr = (ROOT / 'include' / 'scasr.h').read_text()
    assert int(re.search(r'#define SC_ABI_VERSION (\d+)', hdr).group(1)) == _abi.ABI_VERSION


def test_argument_errors_do_not_need_a_gpu():
    from speechcatcher_amd import _abi
    lib = _abi.load()
    rc = lib.sc_gemm(None, None, 4, None, None, None, None, 4, 1, 1, 32, 0, 0, None)
    assert rc == -1
    assert b"null" in lib.sc_last_error()


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speechcatcher_amd import _abi
    from speechcatcher_amd.hip_backend import HipBackend
    with pytest.raises(_abi.ScasrError):
        HipBackend("cuda:0")


def test_header_is_plain_c_and_struct_layouts_match_ctypes(tmp_path):
    """include/scasr.h must be consumable by a C host (no C++ / torch types), and the ctypes mirrors of its
    structs (speechcatcher_amd/_abi.py) must have the C compiler's size and field offsets."""
    import ctypes
    import shutil
    import subprocess
    from speechcatcher_amd import _abi
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    structs = {"sc_enc_layer": _abi.EncLayer, "sc_dec_layer": _abi.DecLayer, "sc_search": _abi.Search,
               "sc_config": _abi.Config, "sc_named_tensor": _abi.NamedTensor, "sc_stream_options": _abi.StreamOptions,
               "sc_stream_info_t": _abi.StreamInfo}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "scasr.h"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(src), "-o", str(exe)],
                   check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True,
                                                       text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(out[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype of scasr.h against the ctypes signature bound for it: parameter count, and the
    class of every parameter (pointer / integer / float / double) - a drifted binding would pass garbage."""
    import ctypes
    from speechcatcher_amd import _abi
    text = re.sub(r"/\*.*?\*/", " ", (ROOT / "include" / "scasr.h").read_text(), flags=re.S)
    protos = re.findall(r"\b(?:int|long|void|double|size_t|const char \*|void \*|float \*)\s*(sc_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text)
    assert len(protos) == len(_abi.EXPORTED_SYMBOLS)

    def c_class(param):
        param = param.strip()
        if "*" in param:
            return "ptr"
        base = param.rsplit(" ", 1)[0] if " " in param else param
        if "float" in base:
            return "float"
        if "double" in base:
            return "double"
        return "int"

    def ct_class(t):
        if t in (ctypes.c_float,):
            return "float"
        if t in (ctypes.c_double,):
            return "double"
        if t in (ctypes.c_int, ctypes.c_size_t, ctypes.c_longlong, ctypes.c_int32, ctypes.c_uint, ctypes.c_long):
            return "int"
        return "ptr"

    for name, params in protos:
        params = params.strip()
        plist = [] if params in ("", "void") else [p for p in params.split(",")]
        _, argtypes = _abi._SIGS[name]
        assert len(plist) == len(argtypes), f"{name}: header has {len(plist)} parameters, binding {len(argtypes)}"
        for i, (p, t) in enumerate(zip(plist, argtypes)):
            assert c_class(p) == ct_class(t), f"{name} parameter {i}: '{p.strip()}' bound as {t}"


def test_oracle_is_only_used_as_the_checker():
    """The product never routes through the oracle: no module of the package imports it (neither do the tools that
    run on the GPU box), bench.py only inside its cpu_baseline leg and __graft_entry__ only inside smoke()."""
    import ast

    def oracle_imports(path):
        """-> names of the enclosing top-level functions (None = module level) of every oracle import"""
        tree = ast.parse(path.read_text())
        found = []

        def visit(node, owner):
            for child in ast.iter_child_nodes(node):
                o = child.name if owner is None and isinstance(child, (ast.FunctionDef, ast.ClassDef)) else owner
                if isinstance(child, ast.ImportFrom) and (child.module or "").split(".")[0] == "oracle":
                    found.append(owner)
                elif isinstance(child, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in child.names):
                    found.append(owner)
                visit(child, o)

        visit(tree, None)
        return found

    for path in sorted((ROOT / "speechcatcher_amd").rglob("*.py")):
        assert oracle_imports(path) == [], f"{path.relative_to(ROOT)} imports the oracle"
    # (the worker function of the cpu_baseline leg: it runs in spawned processes, so it is a top-level function)
    # (_oracle_model builds the oracle's model for them; cpu_fast_forward_worker brings the stream to the timed window)
    assert set(oracle_imports(ROOT / "bench.py")) == {"_oracle_model", "cpu_fast_forward_worker"}
    # ... and they are reachable from the cpu_baseline leg only
    tree = ast.parse((ROOT / "bench.py").read_text())
    refs = {node.name: {n.id for n in ast.walk(node) if isinstance(n, ast.Name)}
            for node in tree.body if isinstance(node, (ast.FunctionDef, ast.ClassDef))}
    users = lambda name: {f for f, names in refs.items() if name in names and f != name}   # noqa: E731
    assert users("_oracle_model") == {"cpu_fast_forward_worker", "cpu_baseline_worker"}
    assert users("cpu_fast_forward_worker") == users("cpu_baseline_worker") == {"cpu_baseline"}
    assert users("cpu_baseline") == {"main"}
    assert set(oracle_imports(ROOT / "__graft_entry__.py")) == {"smoke"}
    # the product's only compute backend is the HIP library: no torch arithmetic fallback in the engine
    engine_src = (ROOT / "speechcatcher_amd" / "engine.py").read_text()
    assert "SpecBackend" not in engine_src
