"""-m gpu: the C++ `seeksv` command line (seeksv_amd/bin/seeksv) against the reference's own output files, byte for byte."""
import gzip
import os
import subprocess

import pytest

import bamio
import golden_util as G
from seeksv_amd import host

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["host-inflate", "device-inflate"])
def inflate_mode(request, monkeypatch):
    """every CLI test runs twice: BAM inflated + decoded by the host threads, and on the GPU (-Z / SSV_DEVICE_INFLATE=1);
    small chunks in the second mode so that even the example BAMs span several chunks and records straddle them"""
    if request.param == "device-inflate":
        monkeypatch.setenv("SSV_DEVICE_INFLATE", "1")
        monkeypatch.setenv("SSV_CHUNK_INFLATED_MB", "1")
        monkeypatch.setenv("SSV_STAGE_MB", "1")
    else:
        monkeypatch.delenv("SSV_DEVICE_INFLATE", raising=False)
    return request.param


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEKSV = os.environ.get("SSV_CLI") or os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")

GETCLIP = [("example", "cancer.sort.bam", "cancer", []), ("example", "normal.sort.bam", "normal", []), ("getclip", "filters.bam", "filters", []),
           ("getclip", "filters.bam", "filters.s", ["-s"]), ("getclip", "filters.bam", "filters.q30", ["-q", "30"]), ("getclip", "stress1.bam", "stress1.t08", ["-t", "0.8"]),
           ("getclip", "stress2.bam", "stress2", []), ("getclip", "stress3.bam", "stress3", []), ("getclip", "stress3.bam", "stress3.t08", ["-t", "0.8"]),
           # records whose whole CIGAR is one soft clip (two rows with an empty aligned part), contigs that come back (one flush per visit)
           ("getclip", "lone_s.bam", "lone_s", []), ("getclip", "lone_s2.bam", "lone_s2", []), ("getclip", "lone_s2.bam", "lone_s2.s", ["-s"]),
           ("getclip", "lone_s2.bam", "lone_s2.q0", ["-q", "0"]), ("getclip", "unsorted.bam", "unsorted", [])]


@pytest.mark.parametrize("passes", ["one-pass", "pass-per-contig", "pass-per-contig-small-batches", "columns-by-scans"])
@pytest.mark.parametrize("sub,bam,prefix,flags", GETCLIP, ids=[c[2] for c in GETCLIP])
def test_cli_getclip(tmp_path, sub, bam, prefix, flags, passes):
    """(columns-by-scans: SSV_PACK_COLS=split - the table's columns by two words per slot and two device-wide scans over them, the form before round 6's
    scans over tiles of 256 slots, which every other test runs;
    pass-per-contig: SSV_PASS_RECORDS=1 ends the library's pass at every contig change - where the reference flushes - and the rows of a pass
    are formatted and compressed beside the reading of the next one: what a whole-genome file does every 64 M records;
    small-batches: the host reader hands out batches of 7 records, each announced one ahead (ssv_batch_prefetch), so that passes end inside a
    batch while the next one is on its way to the GPU already - the announcements must survive the ssv_clip_begin in between)"""
    out = str(tmp_path / "o")
    env = None if passes == "one-pass" else dict(os.environ, SSV_PACK_COLS="split") if passes == "columns-by-scans" else dict(os.environ, SSV_PASS_RECORDS="1")
    if passes == "pass-per-contig-small-batches":
        env["SSV_HOST_BATCH_RECORDS"] = "7"
    r = subprocess.run([SEEKSV, "getclip"] + flags + ["-o", out, os.path.join(G.GOLDEN, sub, bam)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert gzip.open(out + ".clip.gz", "rt").read() == G.read_text(sub, prefix + ".clip.txt")
    assert gzip.open(out + ".clip.fq.gz", "rt").read() == G.read_text(sub, prefix + ".clip.fq.txt")
    assert gzip.open(out + ".unmapped_1.fq.gz", "rt").read() == "" and gzip.open(out + ".unmapped_2.fq.gz", "rt").read() == ""
    if sub == "example" or prefix == "unsorted":
        assert r.stderr == G.read_text(sub, prefix + ".getclip.stderr")


@pytest.mark.parametrize("sub,bam,prefix,flags", [c for c in GETCLIP if c[2] in ("cancer", "stress3", "unsorted", "lone_s2")], ids=lambda v: v if isinstance(v, str) else "")
def test_cli_getclip_reader_maps_the_file(tmp_path, sub, bam, prefix, flags, inflate_mode, monkeypatch):
    """SSV_READER=map (-Z only): the chunks are handed to the GPU where they lie in a mapping of the file, their pages page-locked there (ssvh_bam_map_blocks +
    ssv_host_register) instead of being copied into staging buffers - not the default (slower on the boxes measured), same bytes out; chunks of
    1 MB so that a file is several chunks from all three mappings"""
    if inflate_mode != "device-inflate":
        pytest.skip("the host reader does not read chunks")
    monkeypatch.setenv("SSV_READER", "map")
    out = str(tmp_path / "o")
    r = subprocess.run([SEEKSV, "getclip"] + flags + ["-o", out, os.path.join(G.GOLDEN, sub, bam)], capture_output=True, text=True, env=dict(os.environ, SSV_TIMING="2"))
    assert r.returncode == 0, r.stderr
    assert "page-locked in the mapping" in r.stderr or "copied to staging" in r.stderr   # (a file system the runtime cannot lock falls back to the copy, chunk by chunk)
    assert gzip.open(out + ".clip.gz", "rt").read() == G.read_text(sub, prefix + ".clip.txt")
    assert gzip.open(out + ".clip.fq.gz", "rt").read() == G.read_text(sub, prefix + ".clip.fq.txt")


def test_cli_getclip_verifies_block_crcs(tmp_path, inflate_mode):
    """`-Z -C`: the GPU checks every inflated BGZF block against the CRC32 in its trailer.  A good file gives the usual outputs; a file with one quality byte
    changed inside a block whose deflate structure stays valid is refused with a CRC32 message - and, without -C, goes through like in libbam 0.1.16."""
    import struct
    import zlib
    if inflate_mode != "device-inflate":
        pytest.skip("the check belongs to the device decoder")
    src = os.path.join(G.GOLDEN, "example", "cancer.sort.bam")
    out = str(tmp_path / "o")
    r = subprocess.run([SEEKSV, "getclip", "-Z", "-C", "-o", out, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert gzip.open(out + ".clip.gz", "rt").read() == G.read_text("example", "cancer.clip.txt")
    raw = bytearray(open(src, "rb").read())
    blocks, at = [], 0
    while at < len(raw):
        bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1
        blocks.append((at, bsize))
        at += bsize
    rng = __import__("numpy").random.default_rng(3)
    for _ in range(2000):   # a flip that zlib still inflates to the block's size: only the CRC32 knows
        at, bsize = blocks[int(rng.integers(1, len(blocks) - 1))]
        i = at + 18 + int(rng.integers(0, bsize - 26))
        b = bytearray(raw)
        b[i] ^= 1 << int(rng.integers(0, 8))
        try:
            z = zlib.decompressobj(-15)
            data = z.decompress(bytes(b[at + 18:at + bsize - 8]))
        except zlib.error:
            continue
        if len(data) == struct.unpack_from("<I", b, at + bsize - 4)[0] and z.eof and zlib.crc32(data) != struct.unpack_from("<I", b, at + bsize - 8)[0]:
            break
    else:
        pytest.skip("no structure-preserving flip found")
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(b))
    r = subprocess.run([SEEKSV, "getclip", "-Z", "-C", "-o", out + "2", bad], capture_output=True, text=True)
    assert r.returncode != 0 and "CRC32" in r.stderr, r.stderr[-500:]


GETSV = [("pairs1", "pairs1", []), ("pairs1", "pairs1.q0", ["-q", "0"]), ("pairs1", "pairs1.L50", ["-L", "50"]), ("pairs1", "pairs1.L1", ["-L", "1"]),
         ("pairs2", "pairs2", []), ("pairs3", "pairs3", []), ("eqx", "eqx", []), ("deep", "deep", [])]


@pytest.mark.parametrize("case,prefix,flags", GETSV, ids=[c[1] for c in GETSV])
def test_cli_getsv_junction_table(tmp_path, case, prefix, flags):
    """The reference's -B harness: same junction file, same flags -> identical SV table and identical filtered lines on stdout."""
    base = os.path.join(G.GOLDEN, "getsv")
    bam = os.path.join(base, case + ".bam")
    with host.BamReader(bam) as r:
        names, lens = r.target_names, [int(x) for x in r.target_lens]
    empty_bam = str(tmp_path / "empty.clip.bam")
    bamio.write_bam(empty_bam, names, lens, [])
    empty_clip = str(tmp_path / "empty.clip")
    open(empty_clip, "w").close()
    sv = str(tmp_path / "out.sv")
    r = subprocess.run([SEEKSV, "getsv", "-d", "0", "-f", "0", "-b", "0", "-T", "100000"] + flags + ["-B", os.path.join(base, case + ".junctions.txt"), empty_bam, bam, empty_clip, sv,
                        str(tmp_path / "x.fq")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(sv).read() == G.read_text("getsv", prefix + ".sv")
    assert r.stdout == G.read_text("getsv", prefix + ".stdout")
    mean, sd = G.read_text("getsv", prefix + ".isize.txt").split()
    assert f"Mean insert size : {mean}\nMean deviation: {sd}\n" in r.stderr
    assert os.path.getsize(str(tmp_path / "x.fq")) == 0


FULL = [("", []), (".b40", ["-b", "40"]), (".f2", ["-f", "2"]), (".d400", ["-d", "400"]), (".e60", ["-e", "60"]), (".D", ["-D"]), (".n0", ["-n", "0"])]


@pytest.mark.parametrize("sample", ["cancer", "normal"])
@pytest.mark.parametrize("tag,flags", FULL, ids=[t[0] or "default" for t in FULL])
def test_cli_getsv_full_pipeline_example(tmp_path, sample, tag, flags):
    """BASELINE configs[0]: getclip -> (bwa mem, committed clip.bam fixture) -> getsv on the bundled example BAMs: output.sv.txt and the
    filtered-junction lines on stdout are byte-identical to seeksv v1.2.3's, for every filter flag the goldens were made with."""
    ex = os.path.join(G.GOLDEN, "example")
    out = str(tmp_path / "s")
    r = subprocess.run([SEEKSV, "getclip", "-o", out, os.path.join(ex, sample + ".sort.bam")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sv = str(tmp_path / "out.sv")
    r = subprocess.run([SEEKSV, "getsv"] + flags + [os.path.join(ex, sample + ".clip.bam"), os.path.join(ex, sample + ".sort.bam"), out + ".clip.gz", sv, str(tmp_path / "u.fq")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(sv).read() == G.read_text("example", f"{sample}{tag}.sv")
    assert r.stdout == G.read_text("example", f"{sample}{tag}.getsv.stdout")
    assert os.path.getsize(str(tmp_path / "u.fq")) == 0


SYNTH_FULL = {"synthfull": dict(genome_frac=1 / 8192, depth=40, n_sv=24),
              "hbvfull": dict(genome_frac=1 / 8192, depth=60, n_sv=8, n_integrations=10)}  # the tumor of BASELINE config 5 in small: human + HBV, 10 integrations
HBV_NORMAL = dict(genome_frac=1 / 8192, depth=30, n_sv=8, hbv=True)                         # its normal: same reference and germline SVs, no virus
SYNTH_FULL_VARIANTS = [("", []), (".l90", ["-l", "90"]), (".loose", ["-f", "0", "-b", "0", "-d", "0"]), (".strict", ["-e", "10", "-b", "20"])]
_synth_samples = {}


def _synth_sample(tmp_path_factory, name, kw):
    """A synthetic sample regenerated and written as a BAM; getclip'ed once with the CLI."""
    if name not in _synth_samples:
        from seeksv_amd import synth
        d = tmp_path_factory.mktemp(name)
        w = synth.Workload(**kw)
        bam = str(d / f"{name}.bam")
        bamio.soa_to_bam(bam, w.names, w.lens, w.generate_host(0, w.n_total))
        r = subprocess.run([SEEKSV, "getclip", "-o", str(d / "s"), bam], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        _synth_samples[name] = (bam, str(d / "s.clip.gz"), d)
    return _synth_samples[name]


@pytest.fixture(scope="module")
def synthfull_bam(tmp_path_factory):
    return _synth_sample(tmp_path_factory, "synthfull", SYNTH_FULL["synthfull"])


@pytest.mark.parametrize("tag,flags", SYNTH_FULL_VARIANTS, ids=[t[0] or "default" for t in SYNTH_FULL_VARIANTS])
@pytest.mark.parametrize("name", list(SYNTH_FULL))
def test_cli_full_pipeline_synthetic_sv_table(tmp_path_factory, name, tag, flags):
    """getclip -> (bwa mem against the hash-generated reference: committed clip.bam) -> getsv on planted DEL / INV / TRA (synthfull) and on
    the human + HBV hybrid sample with planted virus integrations (hbvfull): the SV table (reverse-strand junctions, translocations, merged
    junctions, viral ends near the HBV contig's ends included) equals the real reference's byte for byte."""
    bam, clip_gz, d = _synth_sample(tmp_path_factory, name, SYNTH_FULL[name])
    sv = str(d / f"out{tag}.sv")
    r = subprocess.run([SEEKSV, "getsv"] + flags + [os.path.join(G.GOLDEN, "synth", f"{name}.clip.bam"), bam, clip_gz, sv, str(d / "u.fq")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(sv).read() == G.read_text("synth", f"{name}{tag}.sv")
    assert r.stdout == G.read_text("synth", f"{name}{tag}.stdout")


@pytest.mark.parametrize("tag,flags", [("", []), (".n0", ["-n", "0"]), (".q0", ["-q", "0"])], ids=["default", "n0", "q0"])
def test_cli_config5_virus_integration_somatic(tmp_path_factory, tag, flags):
    """BASELINE config 5 in small: the tumor's SV table (human + HBV hybrid reference, 10 planted integrations + 8 germline SVs) filtered against
    the patient's normal sample (`seeksv getclip` on the normal, then `seeksv somatic`): every viral junction is somatic (no support in the
    control), the germline SVs carry the control's clipped-read and discordant-pair counts; table and stderr equal the real reference's."""
    bam, clip_gz, d = _synth_sample(tmp_path_factory, "hbvnormal", HBV_NORMAL)
    out = str(d / f"somatic{tag}.sv")
    r = subprocess.run([SEEKSV, "somatic"] + flags + [bam, clip_gz, os.path.join(G.GOLDEN, "synth", "hbvfull.loose.sv"), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out).read() == G.read_text("somatic", f"hbv.somatic{tag}.sv")
    assert _somatic_stderr(r.stderr) == _somatic_stderr(G.read_text("somatic", f"hbv.somatic{tag}.stderr"))
    rows = [l.split("\t") for l in open(out) if not l.startswith("@")]
    viral = [r_ for r_ in rows if "HBV" in (r_[0], r_[4])]
    assert len(viral) == 10 and all(r_[-3:] == ["0", "0", "0\n"] for r_ in viral)
    if not flags:
        assert all(int(r_[-1]) > 0 for r_ in rows if r_ not in viral)


SOMATIC_VARIANTS = [("", []), (".n0", ["-n", "0"]), (".l0", ["-l", "0"]), (".l60", ["-l", "60"]), (".t05", ["-t", "0.5"]), (".m60", ["-m", "60"]), (".q0", ["-q", "0"])]


def _somatic_stderr(text):
    """the reference prints the BAM path it was given; compare everything else"""
    return [("Bam/sam <bam>" + l[l.index("    Mean insert size"):]) if l.startswith("Bam/sam ") else l for l in text.splitlines() if l != "'CalculateInsertsizeDeviation' finished"]


@pytest.mark.parametrize("tag,flags", SOMATIC_VARIANTS, ids=[t[0] or "default" for t in SOMATIC_VARIANTS])
def test_cli_somatic_synthetic(synthfull_bam, tag, flags):
    """`seeksv somatic`: 227 tumor junction rows (the sample's own junctions and systematic variants that take every branch of
    ReadTumorFileAndOutputSomaticInfo) against the synthetic sample as the normal: output table and stderr messages equal the real reference's."""
    bam, clip_gz, d = synthfull_bam
    out = str(d / f"somatic{tag}.sv")
    r = subprocess.run([SEEKSV, "somatic"] + flags + [bam, clip_gz, os.path.join(G.GOLDEN, "somatic", "tumor.sv"), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out).read() == G.read_text("somatic", f"somatic{tag}.sv")
    assert _somatic_stderr(r.stderr) == _somatic_stderr(G.read_text("somatic", f"somatic{tag}.stderr"))


def test_cli_somatic_example(tmp_path):
    """example/seeksv.somatic.sh: tumor table of cancer.sort.bam against normal.sort.bam + its clip.gz"""
    ex = os.path.join(G.GOLDEN, "example")
    pre = str(tmp_path / "normal")
    r = subprocess.run([SEEKSV, "getclip", "-o", pre, os.path.join(ex, "normal.sort.bam")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = str(tmp_path / "somatic.sv")
    r = subprocess.run([SEEKSV, "somatic", os.path.join(ex, "normal.sort.bam"), pre + ".clip.gz", os.path.join(ex, "cancer.sv"), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out).read() == G.read_text("example", "cancer.somatic.temp.sv")


def test_cli_somatic_usage():
    r = subprocess.run([SEEKSV, "somatic", "-l", "95", "a", "b", "c", "d"], capture_output=True, text=True)
    assert r.returncode == 1 and "Error: value of -l must in range [0, 90) " in r.stderr
    r = subprocess.run([SEEKSV, "somatic", "a", "b"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("3\t1\n") and "Usage: seeksv somatic" in r.stderr


@pytest.mark.parametrize("name", list(SYNTH_FULL))
def test_cli_realign_full_pipeline(tmp_path_factory, name):
    """The whole pipeline without the external aligner: getclip -> `seeksv realign` (GPU re-aligner) -> getsv.  Every planted
    junction is found and the SV table equals, in every column, the table the real reference makes from bwa mem's clip.bam."""
    from seeksv_amd import synth
    bam, clip_gz, d = _synth_sample(tmp_path_factory, name, SYNTH_FULL[name])
    w = synth.Workload(**SYNTH_FULL[name])
    fa = str(d / "ref.fa")
    with open(fa, "w") as f:
        f.write(w.reference_fasta())
    clip_bam = str(d / "realigned.clip.bam")
    r = subprocess.run([SEEKSV, "realign", fa, clip_gz.replace(".clip.gz", ".clip.fq.gz"), clip_bam], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sv = str(d / "realigned.sv")
    r = subprocess.run([SEEKSV, "getsv", clip_bam, bam, clip_gz, sv, str(d / "u2.fq")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [l.split("\t") for l in open(sv).read().splitlines() if not l.startswith("@")]
    want = [l.split("\t") for l in G.read_text("synth", f"{name}.sv").splitlines() if not l.startswith("@")]
    assert got == want  # all 23 columns of every row: the table equals the one made from bwa mem 0.7.10's clip.bam
    planted = {(j[0], j[1], j[2], j[3], j[4], j[5]) for j in w.junctions}
    found = {(c[0], int(c[1]), c[2], c[4], int(c[5]), c[6]) for c in got}
    assert len(found) == len(planted)
    # the FASTQ reader takes the file as one text (gzip members inflated side by side): the same clip.bam from a plain file, from CRLF line ends,
    # from a file without its last newline and from somebody else's gzip; a truncated record is refused
    text = gzip.open(clip_gz.replace(".clip.gz", ".clip.fq.gz"), "rb").read()
    variants = {"plain.fq": text, "crlf.fq": text.replace(b"\n", b"\r\n"), "nonl.fq": text[:-1], "other.fq.gz": gzip.compress(text)}
    for fn, data in variants.items():
        open(str(d / fn), "wb").write(data)
        out = str(d / (fn + ".bam"))
        r = subprocess.run([SEEKSV, "realign", fa, str(d / fn), out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(out, "rb").read() == open(clip_bam, "rb").read(), fn
    open(str(d / "cut.fq"), "wb").write(b"\n".join(text.split(b"\n")[:6]) + b"\n")
    r = subprocess.run([SEEKSV, "realign", fa, str(d / "cut.fq"), str(d / "cut.bam")], capture_output=True, text=True)
    assert r.returncode != 0 and "Truncated FASTQ record" in r.stderr


@pytest.mark.parametrize("name", list(SYNTH_FULL))
@pytest.mark.parametrize("opts", [[], ["-c", "-q 5 -t 0.85", "-v", "-b 2 -L 100 -q 10"]], ids=["defaults", "options"])
@pytest.mark.parametrize("run_env", [{}, {"SSV_PASS_RECORDS": "3000", "SSV_CHUNK_INFLATED_MB": "1"}, {"SSV_RUN_PARSE_ROWS": "1"}], ids=["one-pass", "many-passes", "rows-parsed"])
def test_cli_run_equals_three_commands(tmp_path_factory, name, opts, inflate_mode, run_env):
    """`seeksv run in.bam ref.fa prefix` (one process; the BAM inflated and decoded ONCE on the GPU, its records kept in HBM for the getsv
    passes) writes, byte for byte, the files that `getclip`, `realign` and `getsv` write one after the other - and, under the default
    options, the SV table the real reference makes from bwa mem's clip.bam (tests/golden/synth).  Round 6: the aligner step runs beside getclip, pass
    by pass (many-passes: a pass per 3000 records, so that it really does), and the junction stage takes the rows as getclip's formatter kept them
    (rows-parsed: the text is parsed back instead, the form before)."""
    if inflate_mode != "device-inflate":
        pytest.skip("`seeksv run` always decodes on the GPU")
    from seeksv_amd import synth
    bam, _, d = _synth_sample(tmp_path_factory, name, SYNTH_FULL[name])
    w = synth.Workload(**SYNTH_FULL[name])
    fa = str(d / "ref_run.fa")
    with open(fa, "w") as f:
        f.write(w.reference_fasta())
    tag = ("d" if not opts else "o") + "".join(k[4] for k in sorted(run_env))
    clip_o = opts[1].split() if opts else []
    sv_o = opts[3].split() if opts else []
    a = str(d / f"three_{tag}")
    r1 = subprocess.run([SEEKSV, "getclip"] + clip_o + ["-o", a, bam], capture_output=True, text=True)
    assert r1.returncode == 0, r1.stderr
    r2 = subprocess.run([SEEKSV, "realign", fa, a + ".clip.fq.gz", a + ".clip.bam"], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    r3 = subprocess.run([SEEKSV, "getsv"] + sv_o + [a + ".clip.bam", bam, a + ".clip.gz", a + ".sv.txt", a + ".unmapped.clip.fq"], capture_output=True, text=True)
    assert r3.returncode == 0, r3.stderr
    b = str(d / f"one_{tag}")
    r = subprocess.run([SEEKSV, "run"] + opts + [bam, fa, b], capture_output=True, text=True, env=dict(os.environ, SSV_FASTA_CHECK="1", **run_env))  # (the parallel FASTA reader against the serial one)
    assert r.returncode == 0, r.stderr
    for ext in (".clip.gz", ".clip.fq.gz", ".unmapped_1.fq.gz", ".unmapped_2.fq.gz"):
        assert gzip.open(a + ext, "rb").read() == gzip.open(b + ext, "rb").read(), ext
    for ext in (".sv.txt", ".unmapped.clip.fq"):
        assert open(a + ext, "rb").read() == open(b + ext, "rb").read(), ext
    with host.BamReader(a + ".clip.bam") as x, host.BamReader(b + ".clip.bam") as y:
        ba, bb = x.read_batch(1 << 22, keep_all_seq=True), y.read_batch(1 << 22, keep_all_seq=True)
        for k in ("tid", "pos", "flag", "mapq", "n_cigar", "l_qseq", "cigar"):
            assert (ba[k] == bb[k]).all(), k
    assert r.stdout == r3.stdout                                 # the filtered junctions
    rows = [l.split("\t") for l in open(b + ".sv.txt").read().splitlines() if not l.startswith("@")]
    assert len(rows) > 0
    if not opts:
        want = [l.split("\t") for l in G.read_text("synth", f"{name}.sv").splitlines() if not l.startswith("@")]
        assert rows == want


@pytest.mark.parametrize("n_values", [1, 2, 4, 5, 8, 9, 11, 16, 17, 40, 45, 46])
def test_cli_getclip_formats_every_quality_alphabet(tmp_path, n_values):
    """the CLI's row formatter reads the compact table with every shape of quality stream (1-4 bits a quality, three or two qualities a 7-bit group, two an 11-bit
    group, bytes) and with bases outside A/C/G/T (the exception list): the rows and FASTQ records equal the oracle's on the same records (round 6: the formatter
    decodes through byte tables and dword stores)"""
    import numpy as np
    import oracle_lib
    from seeksv_amd import synth
    from test_hip_golden import _remap_qualities, _with_bases
    w = synth.Workload(genome_frac=1 / 8192, depth=40, n_sv=24)
    b = w.generate_host(0, w.n_total)
    alphabet = [(3 + 2 * k) % 94 for k in range(n_values)]
    b = _remap_qualities(b, alphabet)
    if n_values % 2 == 0:
        b = _with_bases(b, lambda r, i, c: 15 if (r * 31 + i) % 53 == 0 else c)   # scattered N
    bam = str(tmp_path / "q.bam")
    host.write_bam(bam, w.names, w.lens, [b])
    _, _, batches = host.read_bam(bam)
    rows, fq = host.format_clip_outputs(oracle_lib.getclip(batches, 0.9, 1, False), w.names)
    out = str(tmp_path / "o")
    r = subprocess.run([SEEKSV, "getclip", "-o", out, bam], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert gzip.open(out + ".clip.gz", "rt").read() == rows
    assert gzip.open(out + ".clip.fq.gz", "rt").read() == fq
