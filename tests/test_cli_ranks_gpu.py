"""-m gpu: `seeksv getclip -N n` / `seeksv getsv -N n` - the BAM file cut into n runs of records (libseeksv_host's partition), one rank per
run, every rank with its own context (on a one-GPU box they share the GPU and the exchange goes through host memory; between different
GPUs it is one RCCL all-gather) - must write the bytes the single-GPU run writes, i.e. the reference's."""
import gzip
import os
import subprocess

import pytest

import bamio
import golden_util as G
from seeksv_amd import host

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEKSV = os.environ.get("SSV_CLI") or os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")

GETCLIP = [("example", "cancer.sort.bam", "cancer", []), ("example", "normal.sort.bam", "normal", []), ("getclip", "filters.bam", "filters", []),
           ("getclip", "filters.bam", "filters.q30", ["-q", "30"]), ("getclip", "stress1.bam", "stress1.t08", ["-t", "0.8"]), ("getclip", "stress2.bam", "stress2", []),
           ("getclip", "stress3.bam", "stress3", []), ("getclip", "lone_s2.bam", "lone_s2", []),
           ("getclip", "unsorted.bam", "unsorted", [])]   # contigs that come back: the ranks give way to one GPU that ends a pass at every such visit


@pytest.mark.parametrize("n", [2, 3, 7])
@pytest.mark.parametrize("sub,bam,prefix,flags", GETCLIP, ids=[c[2] for c in GETCLIP])
def test_getclip_ranks(tmp_path, sub, bam, prefix, flags, n):
    out = str(tmp_path / "o")
    r = subprocess.run([SEEKSV, "getclip", "-N", str(n), "-H", "300"] + flags + ["-o", out, os.path.join(G.GOLDEN, sub, bam)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert gzip.open(out + ".clip.gz", "rt").read() == G.read_text(sub, prefix + ".clip.txt")
    assert gzip.open(out + ".clip.fq.gz", "rt").read() == G.read_text(sub, prefix + ".clip.fq.txt")
    assert gzip.open(out + ".unmapped_1.fq.gz", "rt").read() == "" and gzip.open(out + ".unmapped_2.fq.gz", "rt").read() == ""
    if sub == "example":
        assert r.stderr == G.read_text(sub, prefix + ".getclip.stderr")


def test_getclip_halo_too_small_is_refused(tmp_path):
    r = subprocess.run([SEEKSV, "getclip", "-N", "2", "-H", "10", "-o", str(tmp_path / "o"), os.path.join(G.GOLDEN, "example", "cancer.sort.bam")], capture_output=True, text=True)
    assert r.returncode != 0 and "-H" in r.stderr


GETSV = [("pairs1", "pairs1", []), ("pairs1", "pairs1.q0", ["-q", "0"]), ("pairs1", "pairs1.L1", ["-L", "1"]), ("pairs2", "pairs2", []), ("pairs3", "pairs3", []), ("eqx", "eqx", []),
         ("deep", "deep", [])]


@pytest.mark.parametrize("n", [2, 3, 5])
@pytest.mark.parametrize("case,prefix,flags", GETSV, ids=[c[1] for c in GETSV])
def test_getsv_ranks_junction_table(tmp_path, case, prefix, flags, n):
    """the -B harness over n ranks; `deep` holds > 8000x stacks: a rank that starts inside or behind one replays the records before it
    (ssv_getsv_prime) and must drop exactly the reads the reference's pileup drops"""
    base = os.path.join(G.GOLDEN, "getsv")
    bam = os.path.join(base, case + ".bam")
    with host.BamReader(bam) as r:
        names, lens = r.target_names, [int(x) for x in r.target_lens]
    empty_bam = str(tmp_path / "empty.clip.bam")
    bamio.write_bam(empty_bam, names, lens, [])
    empty_clip = str(tmp_path / "empty.clip")
    open(empty_clip, "w").close()
    sv = str(tmp_path / "out.sv")
    r = subprocess.run([SEEKSV, "getsv", "-N", str(n), "-d", "0", "-f", "0", "-b", "0", "-T", "100000"] + flags + ["-B", os.path.join(base, case + ".junctions.txt"), empty_bam, bam, empty_clip, sv,
                        str(tmp_path / "x.fq")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(sv).read() == G.read_text("getsv", prefix + ".sv")
    assert r.stdout == G.read_text("getsv", prefix + ".stdout")


@pytest.mark.parametrize("sample", ["cancer", "normal"])
def test_full_pipeline_example_ranks(tmp_path, sample):
    ex = os.path.join(G.GOLDEN, "example")
    out = str(tmp_path / "s")
    r = subprocess.run([SEEKSV, "getclip", "-N", "3", "-o", out, os.path.join(ex, sample + ".sort.bam")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sv = str(tmp_path / "out.sv")
    r = subprocess.run([SEEKSV, "getsv", "-N", "4", os.path.join(ex, sample + ".clip.bam"), os.path.join(ex, sample + ".sort.bam"), out + ".clip.gz", sv, str(tmp_path / "u.fq")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(sv).read() == G.read_text("example", f"{sample}.sv")
    assert r.stdout == G.read_text("example", f"{sample}.getsv.stdout")


def test_synthetic_sample_ranks_equal_single(tmp_path):
    """600 K records, 40 planted SVs: 1 rank vs 4 and 8 ranks, every output file byte for byte"""
    from seeksv_amd import synth
    w = synth.Workload(genome_frac=1 / 1024, depth=30, n_sv=40)
    bam = str(tmp_path / "s.bam")
    host.write_bam(bam, w.names, w.lens, [w.generate_host(0, w.n_total)])
    rows = "".join("\t".join(str(x) for x in (j[0], j[1], j[2], 0, j[3], j[4], j[5], 0, 0, 0, "NA", 0, 0, 0, 0, 0, 0, 0, 0, "50M", "50M", "ACGT", "ACGT")) + "\n" for j in w.junctions)
    jf = str(tmp_path / "j.txt")
    open(jf, "w").write(rows)
    empty_bam, empty_clip = str(tmp_path / "e.bam"), str(tmp_path / "e.clip")
    host.write_bam(empty_bam, w.names, w.lens, [])
    open(empty_clip, "w").close()
    outs = {}
    for n in (1, 4, 8):
        o = str(tmp_path / f"o{n}")
        r = subprocess.run([SEEKSV, "getclip", "-N", str(n), "-o", o, bam], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        sv = o + ".sv"
        r2 = subprocess.run([SEEKSV, "getsv", "-N", str(n), "-d", "0", "-f", "0", "-b", "0", "-B", jf, empty_bam, bam, empty_clip, sv, o + ".u.fq"], capture_output=True, text=True)
        assert r2.returncode == 0, r2.stderr
        outs[n] = (gzip.open(o + ".clip.gz", "rb").read(), gzip.open(o + ".clip.fq.gz", "rb").read(), open(sv, "rb").read(), r2.stdout, r.stderr)
    assert len(outs[1][0]) > 100000 and outs[1][2].count(b"\n") > 10
    assert outs[4] == outs[1] and outs[8] == outs[1]


# ---- the same with -Z: every rank's run of BGZF blocks is inflated and decoded on the GPU (ssvh_bam_raw_begin_range, ssv_bamdec_limit) ----

@pytest.mark.parametrize("n", [2, 5])
@pytest.mark.parametrize("sub,bam,prefix,flags", [c for c in GETCLIP if c[2] in ("cancer", "normal", "filters", "stress2")], ids=lambda v: v if isinstance(v, str) and "." not in v else None)
def test_getclip_ranks_device_inflate(tmp_path, sub, bam, prefix, flags, n):
    out = str(tmp_path / "o")
    r = subprocess.run([SEEKSV, "getclip", "-Z", "-N", str(n), "-H", "300"] + flags + ["-o", out, os.path.join(G.GOLDEN, sub, bam)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert gzip.open(out + ".clip.gz", "rt").read() == G.read_text(sub, prefix + ".clip.txt")
    assert gzip.open(out + ".clip.fq.gz", "rt").read() == G.read_text(sub, prefix + ".clip.fq.txt")
    if sub == "example":
        assert r.stderr == G.read_text(sub, prefix + ".getclip.stderr")


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("case,prefix,flags", [c for c in GETSV if c[1] in ("pairs1", "pairs3", "deep")], ids=lambda v: v if isinstance(v, str) else None)
def test_getsv_ranks_device_inflate(tmp_path, case, prefix, flags, n):
    base = os.path.join(G.GOLDEN, "getsv")
    bam = os.path.join(base, case + ".bam")
    with host.BamReader(bam) as r:
        names, lens = r.target_names, [int(x) for x in r.target_lens]
    empty_bam = str(tmp_path / "empty.clip.bam")
    bamio.write_bam(empty_bam, names, lens, [])
    empty_clip = str(tmp_path / "empty.clip")
    open(empty_clip, "w").close()
    sv = str(tmp_path / "out.sv")
    r = subprocess.run([SEEKSV, "getsv", "-Z", "-N", str(n), "-d", "0", "-f", "0", "-b", "0", "-T", "100000"] + flags + ["-B", os.path.join(base, case + ".junctions.txt"), empty_bam, bam, empty_clip, sv,
                        str(tmp_path / "x.fq")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(sv).read() == G.read_text("getsv", prefix + ".sv")
    assert r.stdout == G.read_text("getsv", prefix + ".stdout")


def test_synthetic_sample_ranks_device_inflate(tmp_path):
    """1.2 M records (several BGZF chunks per rank are not needed: what counts are cuts inside blocks): -Z -N 4 equals the plain single run"""
    from seeksv_amd import synth
    w = synth.Workload(genome_frac=1 / 512, depth=30, n_sv=60)
    bam = str(tmp_path / "s.bam")
    host.write_bam(bam, w.names, w.lens, [w.generate_host(0, w.n_total)])
    outs = {}
    for tag, extra in (("one", []), ("z4", ["-Z", "-N", "4"]), ("z7", ["-Z", "-N", "7"])):
        o = str(tmp_path / tag)
        r = subprocess.run([SEEKSV, "getclip"] + extra + ["-o", o, bam], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs[tag] = (gzip.open(o + ".clip.gz", "rb").read(), gzip.open(o + ".clip.fq.gz", "rb").read(), r.stderr)
    assert len(outs["one"][0]) > 200000 and outs["z4"] == outs["one"] and outs["z7"] == outs["one"]
