"""CPU: the host stage of `seeksv somatic` (normal-cluster look-ups for every tumor junction row), through the CLI's -J dump hook (no GPU, no
BAM pass), against the first 25 columns of the real reference's output on the 227-row branch-coverage table (tests/golden/somatic)."""
import gzip
import os
import subprocess

import numpy as np

import golden_util as G
import oracle_lib
from seeksv_amd import host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEKSV = os.environ.get("SSV_CLI") or os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")
SYNTH_FULL = dict(genome_frac=1 / 8192, depth=40, n_sv=24)


def test_somatic_lookups_match_reference(tmp_path):
    if not os.path.exists(SEEKSV):
        subprocess.check_call(["make", "-C", ROOT, "cli"], stdout=subprocess.DEVNULL)
    # the normal's clip.gz: oracle getclip on the regenerated synthetic sample (the reference's getclip output equals it, test_oracle_golden)
    w = synth.Workload(**SYNTH_FULL)
    b = w.generate_host(0, w.n_total)
    table = oracle_lib.getclip([b], 0.9, 1, False)
    rows, _ = host.format_clip_outputs(table, w.names)
    clip = str(tmp_path / "normal.clip.gz")
    with gzip.open(clip, "wt") as f:
        f.write(rows)
    for tag, flags in (("", []), (".l0", ["-l", "0"]), (".l60", ["-l", "60"]), (".t05", ["-t", "0.5"]), (".m60", ["-m", "60"])):
        dump = str(tmp_path / f"dump{tag}.txt")
        r = subprocess.run([SEEKSV, "somatic", "-J", dump] + flags + ["unused.bam", clip, os.path.join(G.GOLDEN, "somatic", "tumor.sv"), "unused.out"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = [l.split("\t") for l in open(dump).read().splitlines()]
        exp = [l.split("\t") for l in G.read_text("somatic", f"somatic{tag}.sv").splitlines()]
        assert len(got) == len(exp) == 199
        assert got[0] == exp[0]
        for g, e in zip(got[1:], exp[1:]):
            assert g == e[:25], (tag, g[:9], g[23:], e[23:])
        msgs = [l for l in G.read_text("somatic", f"somatic{tag}.stderr").splitlines() if not l.startswith(("Bam/sam", "Mean deviation"))]
        assert r.stderr.splitlines() == msgs


def _member_gz(pieces):
    """concatenated gzip members with the header bytes ssvh_gz_append writes (what junction_stage.cpp's slurp_gz_members inflates side by side)"""
    import struct
    import zlib
    out = b""
    for p in pieces:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        out += bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3]) + c.compress(p) + c.flush() + struct.pack("<II", zlib.crc32(p), len(p) & 0xffffffff)
    return out


REF = os.path.join(ROOT, "oracle", "_ref", "seeksv_ref")
BAMIDX = os.path.join(ROOT, "oracle", "_ref", "bamidx")


def test_somatic_cluster_index_equals_reference_multimap(tmp_path):
    """The normal sample's clusters as sorted arrays (round 6) against the REAL reference's std::multimap on files whose rows are NOT in getclip's
    order (shuffled: equal keys must keep file order; contigs interleaved), read as gzip members by all threads, as one stream, with a malformed last
    row (the stream loop), and with a handful of rows per thread."""
    import bamio
    import pytest
    if not (os.path.exists(REF) and os.path.exists(BAMIDX)):
        pytest.skip("the real reference binary is not built here")
    w = synth.Workload(**SYNTH_FULL)
    b = w.generate_host(0, w.n_total)
    bam = str(tmp_path / "normal.bam")
    bamio.soa_to_bam(bam, w.names, w.lens, b)
    subprocess.run([BAMIDX, bam], check=True, capture_output=True)
    rows, _ = host.format_clip_outputs(oracle_lib.getclip([b], 0.9, 1, False), w.names)
    lines = rows.splitlines(keepends=True)
    # near-duplicates of existing rows on the same key with other support values: which one a probe meets first depends on file order
    rng = np.random.RandomState(5)
    extra = []
    for l in lines[::7]:
        c = l.split("\t")
        c[8] = str(int(c[8]) + 100) + "\n"
        extra.append("\t".join(c))
    shuffled = lines + extra
    rng.shuffle(shuffled)
    tumor = os.path.join(G.GOLDEN, "somatic", "tumor.sv")
    forms = {
        "sorted.members": (_member_gz(["".join(lines[i:i + 40]).encode() for i in range(0, len(lines), 40)]), {}),
        "shuffled.members": (_member_gz(["".join(shuffled[i:i + 25]).encode() for i in range(0, len(shuffled), 25)]), {"SSV_ROWS_CHUNK_KB": "1"}),
        "shuffled.stream": (gzip.compress("".join(shuffled).encode()), {}),
        "shuffled.serial": (_member_gz(["".join(shuffled).encode()[:4000], "".join(shuffled).encode()[4000:]]), {"SSV_SERIAL": "gz,rows"}),
        "malformed": (gzip.compress(("".join(shuffled) + "chrX\tnot_a_number\t5\n").encode()), {}),
    }
    for name, (blob, env) in forms.items():
        clip = str(tmp_path / f"{name}.clip.gz")
        with open(clip, "wb") as f:
            f.write(blob)
        ref_out = str(tmp_path / f"{name}.ref.sv")
        r = subprocess.run([REF, "somatic", "-n", "0", bam, clip, tumor, ref_out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        dump = str(tmp_path / f"{name}.dump")
        g = subprocess.run([SEEKSV, "somatic", "-J", dump, "unused.bam", clip, tumor, "unused.out"], capture_output=True, text=True, env=dict(os.environ, **env))
        assert g.returncode == 0, g.stderr
        got = [l.split("\t") for l in open(dump).read().splitlines()]
        exp = [l.split("\t") for l in open(ref_out).read().splitlines()]
        assert len(got) == len(exp) > 100, name
        assert [x[:25] for x in exp[1:]] == got[1:], name
        if name.startswith("shuffled"):
            assert any(int(x[23]) >= 100 or int(x[24]) >= 100 for x in got[1:]), "the duplicated keys were never met"
