"""CPU: the host junction stage of `seeksv getsv` (seeksv_amd/host/junction_stage.cpp) through tests/native/junction_check.cpp.
The rows of clip.gz are inflated member by member and parsed by several threads; both must give exactly what the serial forms give (zlib's gzread,
the reference's `fin >> ...` loop, getsv.h:441-446), and anything the fast forms cannot vouch for must end up in the serial ones."""
import gzip
import os
import struct
import subprocess
import sys
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "tests", "golden", "example")
HEAD = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3])


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from seeksv_amd import _abi
    out = str(tmp_path_factory.mktemp("jc") / "junction_check")
    flags = os.environ.get("SSV_TEST_CXXFLAGS", "-O2").split()  # (make asan: the sanitizer flags)
    subprocess.check_call(["g++"] + flags + ["-std=c++17", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "junction_check.cpp"),
                           os.path.join(ROOT, "seeksv_amd", "host", "junction_stage.cpp"), "-o", out, "-L" + _abi.LIBDIR, "-lseeksv_host", "-lz", "-lpthread", "-Wl,-rpath," + _abi.LIBDIR])
    return out


def write_gz(path, data, **env):
    """through ssvh_gz_append in a child process (its environment knobs are read once)"""
    src = path + ".in"
    open(src, "wb").write(data)
    code = ("import sys, ctypes as C; sys.path.insert(0, %r); from seeksv_amd import _abi; lib = _abi.host_lib();"
            "lib.ssvh_gz_append.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_int]; d = open(%r, 'rb').read();"
            "sys.exit(lib.ssvh_gz_append(%r.encode(), d, len(d), 0))") % (ROOT, src, path)
    e = dict(os.environ)
    e.pop("SSV_GZ_LEVEL", None)
    e.update(env)
    assert subprocess.run([sys.executable, "-c", code], env=e).returncode == 0


def run(exe, args, **env):
    e = dict(os.environ)
    for k in ("SSV_SERIAL", "SSV_ROWS_CHUNK_KB"):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([exe] + args, capture_output=True, env=e)
    assert r.returncode == 0, r.stderr
    return r.stdout


@pytest.mark.parametrize("sample", ["cancer", "normal"])
def test_join_same_with_parallel_gunzip_and_parse(exe, tmp_path, sample):
    rows = open(os.path.join(EX, sample + ".clip.txt"), "rb").read()
    bam = os.path.join(EX, sample + ".clip.bam")
    files = {}
    files["members"] = str(tmp_path / "m.gz"); write_gz(files["members"], rows, SSV_GZ_PIECE_KB="4")             # a dozen Huffman-only members
    files["zlib6"] = str(tmp_path / "z.gz"); write_gz(files["zlib6"], rows, SSV_GZ_PIECE_KB="4", SSV_GZ_LEVEL="6")  # zlib members (same ten header bytes)
    files["gzip"] = str(tmp_path / "g.gz"); open(files["gzip"], "wb").write(gzip.compress(rows))                  # somebody else's gzip: one member
    files["plain"] = str(tmp_path / "p.txt"); open(files["plain"], "wb").write(rows)
    want = run(exe, ["join", files["gzip"], bam, "20"], SSV_SERIAL="gz,rows")
    assert want.count(b"\n") >= 1
    for name, f in files.items():
        assert run(exe, ["slurp", f]) == rows, name
        assert run(exe, ["join", f, bam, "20"], SSV_ROWS_CHUNK_KB="1") == want, name    # ~50 parsing threads
        assert run(exe, ["join", f, bam, "20"]) == want, name
        assert run(exe, ["join", f, bam, "20"], SSV_SERIAL="rows") == want, name
    assert run(exe, ["join", files["members"], bam], SSV_ROWS_CHUNK_KB="1") == run(exe, ["join", files["gzip"], bam], SSV_SERIAL="gz,rows")


def test_false_member_header_inside_a_member(exe, tmp_path):
    """the ten header bytes inside a member's (stored) data: the member in front of them does not end there, so the file is read serially"""
    a = b"row one\n" + HEAD + b" not a header\n" * 3
    b = b"second member\n" * 100
    def member(data, level):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        return HEAD + c.compress(data) + c.flush() + struct.pack("<II", zlib.crc32(data), len(data))
    blob = member(a, 0) + member(b, 6)
    assert blob.count(HEAD) == 3
    p = str(tmp_path / "h.gz")
    open(p, "wb").write(blob)
    assert gzip.open(p, "rb").read() == a + b
    assert run(exe, ["slurp", p]) == a + b
    # and a member whose trailer size is wrong: serial too (gzread reports what it can; here: everything in front of the damage)
    bad = member(a, 6)[:-4] + struct.pack("<I", len(a) + 1) + member(b, 6)
    open(p, "wb").write(bad)
    assert run(exe, ["slurp", p]) == run(exe, ["slurp", p], SSV_SERIAL="gz")


@pytest.mark.parametrize("damage", ["short line", "number with a tail", "two-character side", "number out of range", "blank lines and CRLF"])
def test_rows_the_parallel_parser_does_not_vouch_for(exe, tmp_path, damage):
    """such text goes through the stream loop, whatever that makes of it (the same as with SSV_SERIAL=rows)"""
    lines = open(os.path.join(EX, "cancer.clip.txt"), "rb").read().split(b"\n")
    k = len(lines) // 2
    f = lines[k].split(b"\t")
    if damage == "short line": lines[k] = b"\t".join(f[:8])
    elif damage == "number with a tail": f[1] += b"x"; lines[k] = b"\t".join(f)
    elif damage == "two-character side": f[2] = b"55"; lines[k] = b"\t".join(f)
    elif damage == "number out of range": f[8] = b"99999999999"; lines[k] = b"\t".join(f)
    else: lines = [l + b"\r" for l in lines[:k]] + [b"", b"   ", b"\t"] + lines[k:]
    p = str(tmp_path / "d.txt")
    open(p, "wb").write(b"\n".join(lines))
    bam = os.path.join(EX, "cancer.clip.bam")
    want = run(exe, ["join", p, bam, "20"], SSV_SERIAL="rows")
    assert run(exe, ["join", p, bam, "20"], SSV_ROWS_CHUNK_KB="1") == want
    assert run(exe, ["join", p, bam, "20"]) == want
