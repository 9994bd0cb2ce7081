"""CPU: the per-lane DEFLATE decoder of the device BGZF reader (seeksv_amd/csrc/inflate_core.h, the same source the HIP kernel compiles)
against zlib on ~1100 raw deflate streams: seven data shapes x twelve sizes up to 64 KB x every level, plus the fixed-Huffman,
Huffman-only and RLE strategies, stored blocks (level 0), and refusal of a wrong output size."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inflate_core_against_zlib(tmp_path):
    exe = str(tmp_path / "inflate_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "seeksv_amd", "csrc"), os.path.join(ROOT, "tests", "native", "inflate_check.cpp"), "-lz", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 bad" in r.stdout
