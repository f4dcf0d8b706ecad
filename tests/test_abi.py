"""The C-ABI library loads on a GPU-less host and exports exactly what include/spex_hip.h declares.
No compute entry point is launched here (argument validation that returns before touching the device is fine)."""
import ctypes
import os
import re

import numpy as np
import pytest

from spex_amd import _lib

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "spex_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spex_[a-z0-9_]+)\s*\(", src)) - {"spex_status"})


def test_header_declares_and_library_exports_every_symbol():
    names = declared_functions()
    assert len(names) >= 14
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libspexhip.so does not export {n}"
    assert sorted(_lib.SIGNATURES) == names, "spex_amd/_lib.py binds a different symbol set than the header declares"


def test_binding_arity_matches_header():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, argtypes) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, src, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(argtypes), f"{name}: header has {n} parameters, binding has {len(argtypes)}"


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.spex_version() == 5
    assert isinstance(lib.spex_last_error(), bytes)


def test_argument_validation_fails_loudly_before_touching_the_device():
    lib = _lib.load()
    h = ctypes.c_void_p()
    rowptr = np.array([0, 2, 1], np.int32)      # not monotone
    col = np.array([0, 1], np.int32)
    val = np.ones(2, np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.spex_graph_create(p(rowptr), p(col), p(val), None, 2, 2, 1, ctypes.byref(h))
    assert rc == -1 and b"rowptr" in lib.spex_last_error()
    rowptr = np.array([0, 1, 2], np.int32)
    col = np.array([0, 7], np.int32)            # column out of range: would be an out-of-bounds gather
    rc = lib.spex_graph_create(p(rowptr), p(col), p(val), None, 2, 2, 2, ctypes.byref(h))
    assert rc == -1 and b"out of range" in lib.spex_last_error()
    col = np.array([1, 0, 0], np.int32)
    rowptr = np.array([0, 3, 3], np.int32)      # unsorted / duplicate columns: not coalesced
    rc = lib.spex_graph_create(p(rowptr), p(col), p(np.ones(3, np.float32)), None, 2, 2, 3, ctypes.byref(h))
    assert rc == -1 and b"ascending" in lib.spex_last_error()
    with pytest.raises(_lib.SpexError):
        _lib.call("spex_spmm_f32", None, None, None, None, 1.0, None, None, 1.0, 64, None)
    with pytest.raises(_lib.SpexError):
        _lib.call("spex_adam_step_f32", None, None, None, None, 10, 1, 1e-3, 0.9, 0.999, 1e-8, None, None)


def test_missing_library_is_an_error_not_a_fallback(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SpexError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    """No product file imports, loads, links or shells out to anything under oracle/ (comments may name it)."""
    root = os.path.join(os.path.dirname(HEADER), "..", "spex_amd")
    bad = re.compile(r"(^|\s)(import\s+oracle|from\s+oracle|from\s+\.+oracle)|libspex_oracle|spex_oracle_|oracle/|oracle\.py")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                m = bad.search(txt)
                assert m is None or f == "datasets.py", f"{os.path.join(dp, f)} references the oracle: {m.group(0)!r}"


def test_header_is_plain_c(tmp_path):
    """include/spex_hip.h is the boundary a cgo / JNI / ctypes binding consumes: it must compile as C99 on its own
    (no C++, no HIP or torch types) and link against the built library."""
    import subprocess
    src = tmp_path / "use.c"
    src.write_text('#include "spex_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { spex_graph_t *g = 0; int rc = spex_graph_create(0, 0, 0, 0, 0, 0, 0, &g);\n'
                   '  printf("%d %d %s\\n", spex_version(), rc, spex_last_error()); return rc == SPEX_OK ? 0 : 1; }\n')
    exe = tmp_path / "use"
    inc = os.path.dirname(HEADER)
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe),
                        "-L", libdir, "-lspexhip", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    # NULL rowptr is refused by argument validation before any device call
    assert out.returncode == 1 and out.stdout.startswith("5 -1 spex_graph_create"), (out.stdout, out.stderr)


def test_step_descriptors_have_the_layout_of_the_header(tmp_path):
    """The ctypes mirrors of the three step descriptors (spex_amd/_lib.py) against the C compiler's view of include/spex_hip.h:
    size of each struct and the offset of every field (ABI 4 appended fields to all three)."""
    import subprocess
    structs = {"spex_lightgcn_step_t": _lib.LightGCNStepDesc, "spex_ngcf_step_t": _lib.NGCFStepDesc,
               "spex_dual_task_step_t": _lib.DualTaskStepDesc, "spex_partitioned_step_t": _lib.PartitionedStepDesc,
               "spex_partitioned_dual_step_t": _lib.PartitionedDualStepDesc, "spex_ngcf_deep_step_t": _lib.NGCFDeepStepDesc}
    lines = ['#include "spex_hip.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void) {"]
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)
    assert _lib.STEP_DETERMINISTIC == 1 and _lib.STEP_FIXED_TASK_WEIGHTS == 2 and _lib.STEP_PIPELINED == 4
    assert re.search(r"SPEX_STEP_PIPELINED = 4", open(HEADER).read())
    assert re.search(r"#define SPEX_COMM_ID_BYTES %d\b" % _lib.COMM_ID_BYTES, open(HEADER).read())
    assert re.search(r"SPEX_STEP_DETERMINISTIC = 1", open(HEADER).read()) and re.search(r"SPEX_STEP_FIXED_TASK_WEIGHTS = 2", open(HEADER).read())
