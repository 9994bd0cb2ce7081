"""CPU: the C-ABI libraries load and export every symbol their headers declare; without a GPU the HIP
library refuses to create a context (no silent CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from seeksv_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", txt)))


def test_hip_library_exports_every_declared_symbol():
    lib = _abi.hip_lib()
    names = declared_functions("seeksv_hip.h", "ssv_")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert lib.ssv_abi_version() == 9


def test_host_library_exports_every_declared_symbol():
    lib = _abi.host_lib()
    names = declared_functions("seeksv_host.h", "ssvh_")
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n


def test_struct_layouts_match_header():
    assert C.sizeof(_abi.Batch) == 8 + 4 + 4 + 14 * 8 + 16 + 8 + 8 + 8 + 8   # ... + rec + cigar_ends + tid_runs + n_tid_runs
    assert _abi.TID_RUN_DTYPE.itemsize == 16
    assert _abi.RECORD_DTYPE.itemsize == 64
    assert C.sizeof(_abi.Junction) == 28
    assert C.sizeof(_abi.ClipParams) == 40


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from seeksv_amd.device import Context, SeeksvError
    with pytest.raises(SeeksvError):
        Context(0)


def test_headers_compile_as_c_and_link(tmp_path):
    """a plain C99 program over include/*.h, linked against the two libraries (tests/native/abi_c_smoke.c)"""
    import subprocess
    exe = str(tmp_path / "abi_c_smoke")
    libdir = _abi.LIBDIR
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "abi_c_smoke.c"),
           "-o", exe, "-L" + libdir, "-lseeksv_host", "-lseeksv_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host: " in r.stdout and ("no device" in r.stdout or "gpu context ok" in r.stdout)


def integration_snippet(tag, tmp_path):
    """the fenced C block behind `<!-- snippet:<tag> -->` in INTEGRATION.md, compiled and linked as C99 -> path of the executable"""
    import re
    import subprocess
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- snippet:%s -->\s*```c\n(.*?)```" % re.escape(tag), text, re.S)
    assert m, tag
    src, exe = str(tmp_path / f"snippet_{tag}.c"), str(tmp_path / f"snippet_{tag}")
    open(src, "w").write(m.group(1))
    libdir = _abi.LIBDIR
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), src, "-o", exe, "-L" + libdir, "-lseeksv_hip",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_integration_snippet_compiles(tmp_path):
    """what INTEGRATION.md tells an integrator to write is C that this ABI version compiles and links (its run needs a GPU: tests/test_integration_snippet_gpu.py)"""
    assert os.path.exists(integration_snippet("formats", tmp_path))
