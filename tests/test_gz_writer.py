"""CPU: ssvh_gz_append(_v) - the .gz outputs of getclip.  Default: gzip members of one literal-only dynamic-Huffman block each
(seeksv_amd/host/huff_gz.h); SSV_GZ_LEVEL=n: zlib.  Whatever the coder, zlib's readers (gzip module = gzread's multi-member behaviour) must give
back exactly the bytes written."""
import ctypes as C
import gzip
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

from seeksv_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _append(path, data, append):
    lib = _abi.host_lib()
    lib.ssvh_gz_append.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_int]
    assert lib.ssvh_gz_append(path.encode(), data, len(data), append) == 0


def _cases():
    rng = np.random.default_rng(5)
    rows = open(os.path.join(ROOT, "tests", "golden", "example", "cancer.clip.txt"), "rb").read()
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    skew = b"".join(bytes([65 + k]) * min(f, 300000) for k, f in enumerate(fib[:34]))  # Fibonacci counts: the optimal code is deeper than 15 bits
    return {
        "empty": b"", "one byte": b"A", "one symbol": b"G" * 700001, "two symbols": b"AC" * 5, "all byte values": bytes(range(256)) * 2100,
        "random bytes": rng.integers(0, 256, 900000, dtype=np.uint8).tobytes(), "clip rows": rows * 12, "skewed (length limit)": skew,
        "piece boundary": rows[:1 << 18], "piece boundary + 1": (rows * 6)[:(1 << 18) + 1], "odd length": rows[:100001],
    }


@pytest.mark.parametrize("name", list(_cases()))
def test_default_coder_round_trip(tmp_path, name):
    data = _cases()[name]
    p = str(tmp_path / "t.gz")
    _append(p, data, 0)
    assert gzip.open(p, "rb").read() == data
    # members are independent: appended text follows
    _append(p, b"tail\n", 1)
    _append(p, data, 1)
    assert gzip.open(p, "rb").read() == data + b"tail\n" + data
    # every member on its own is a complete gzip stream with the right CRC and size (zlib checks both)
    raw = open(p, "rb").read()
    d = zlib.decompressobj(31)
    out = d.decompress(raw)
    assert d.eof and out == data[:1 << 18]
    if name == "clip rows":  # order-0 coding of real rows: well above 2 x
        assert len(raw) < 2 * len(data) / 2.0


def test_zlib_levels_still_selectable(tmp_path):
    """SSV_GZ_LEVEL is read once per process: a child process per level"""
    data = _cases()["clip rows"]
    sizes = {}
    for level in ("", "1", "6"):
        p = str(tmp_path / ("l%s.gz" % level))
        code = ("import sys, ctypes as C; sys.path.insert(0, %r); from seeksv_amd import _abi; lib = _abi.host_lib();"
                "lib.ssvh_gz_append.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_int]; d = open(%r, 'rb').read();"
                "sys.exit(lib.ssvh_gz_append(%r.encode(), d, len(d), 0))") % (ROOT, str(tmp_path / "in.txt"), p)
        open(str(tmp_path / "in.txt"), "wb").write(data)
        env = dict(os.environ)
        env.pop("SSV_GZ_LEVEL", None)
        if level:
            env["SSV_GZ_LEVEL"] = level
        assert subprocess.run([sys.executable, "-c", code], env=env).returncode == 0
        assert gzip.open(p, "rb").read() == data
        sizes[level] = os.path.getsize(p)
    assert sizes["6"] < sizes["1"] < sizes[""]
