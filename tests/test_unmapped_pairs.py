"""CPU: getclip's unmapped-pair side channel (seeksv_amd/host/unmapped_pairs.h: hash table over raw BAM records, FASTQ text by helper threads, a
writer thread) against a direct restatement of the reference's std::map loop (clip_reads.h:172-219) on random records, under every batch cut."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unmapped_pairs_against_the_map_loop(tmp_path):
    exe = str(tmp_path / "unmapped_pairs_check")
    flags = os.environ.get("SSV_TEST_CXXFLAGS", "-O2").split()
    subprocess.check_call(["g++"] + flags + ["-std=c++17", "-I" + os.path.join(ROOT, "seeksv_amd", "host"), os.path.join(ROOT, "tests", "native", "unmapped_pairs_check.cpp"), "-lpthread", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 bad" in r.stdout
