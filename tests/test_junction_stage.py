"""CPU: the host junction stage (clip.gz x clip.bam -> junctions -> MergeJunction) of the C++ CLI, through its -J dump hook
(no GPU is touched before the dump), against the reference's SV tables of the bundled examples."""
import os
import subprocess

import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEKSV = os.environ.get("SSV_CLI") or os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")


@pytest.mark.parametrize("sample", ["cancer", "normal"])
def test_junction_table_matches_reference(tmp_path, sample):
    if not os.path.exists(SEEKSV):
        subprocess.check_call(["make", "-C", ROOT, "cli"], stdout=subprocess.DEVNULL)
    ex = os.path.join(G.GOLDEN, "example")
    dump = str(tmp_path / "j.txt")
    r = subprocess.run([SEEKSV, "getsv", "-J", dump, os.path.join(ex, sample + ".clip.bam"), os.path.join(ex, sample + ".sort.bam"), os.path.join(ex, sample + ".clip.txt"),
                        str(tmp_path / "o.sv"), str(tmp_path / "o.fq")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [l.rstrip("\n").split("\t") for l in open(dump)]
    # with default flags every junction of the examples passes the filters, so the reference's table lists them all, in map order
    exp = [l.rstrip("\n").split("\t") for l in G.read_text("example", sample + ".sv").splitlines() if not l.startswith("@")]
    assert len(got) == len(exp) > 0
    for g, e in zip(got, exp):
        assert g[:9] == e[:9] and g[10] == e[10] and g[19:23] == e[19:23], (g, e)
