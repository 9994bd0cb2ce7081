"""CPU, only where the REAL reference binary is present (oracle/_ref/seeksv_ref: built by `make -C oracle ref` in the build container, travels
with the snapshot): the oracle against the reference itself on randomized samples - the same generator as tests/test_random_differential_gpu.py,
restricted to inputs for which the reference's behaviour is defined.  Together the two tests tie the HIP path to the reference far beyond the
committed fixtures: reference == oracle here, oracle == HIP there."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import bamio
import golden_util as G
import oracle_lib as O
from seeksv_amd import host
from test_oracle_golden import OracleBackend
from test_random_differential_gpu import OPS, random_sample

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "seeksv_ref")
BAMIDX = os.path.join(ROOT, "oracle", "_ref", "bamidx")
pytestmark = pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(BAMIDX)), reason="the real reference binary is not built here")


def write_sample(path, names, lens, recs, index=True):
    out = []
    for k, r in enumerate(recs):
        out.append(dict(qname=f"r{k}", flag=r["flag"], tid=r["tid"], pos=r["pos"], mapq=r["mapq"], cigar=[(l, OPS.index(op)) for l, op in r["ops"]],
                        mtid=r["mtid"], mpos=r["mpos"], isize=r["isize"], seq=r["seq"], qual=bytes(r["qual"].tolist()), aux=b"XCC\x01" if r["xc"] else b""))
    bamio.write_bam(path, names, lens, out)
    if index:
        subprocess.run([BAMIDX, path], check=True, capture_output=True)


@pytest.mark.parametrize("seed", range(100, 112))
def test_oracle_equals_reference_on_random_samples(tmp_path, seed):
    names, lens, b, rng, recs = random_sample(seed, safe=True, want_records=True)
    bam = str(tmp_path / "s.bam")
    write_sample(bam, names, lens, recs)
    hn, hl, batches = host.read_bam(bam, 700)
    assert hn == names
    # ---- getclip ----
    for tag, flags, kw in (("a", [], {}), ("b", ["-t", "0.8", "-q", "0"], dict(match_rate=0.8, min_mapq=0)), ("c", ["-s", "-q", "20", "-t", "1"], dict(match_rate=1.0, save_low_quality=True, min_mapq=20))):
        pre = str(tmp_path / tag)
        r = subprocess.run([REF, "getclip"] + flags + ["-o", pre, bam], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        clip, fq = host.format_clip_outputs(O.getclip(batches, **kw), names)
        assert clip == gzip.open(pre + ".clip.gz", "rt").read(), (seed, tag)
        assert fq == gzip.open(pre + ".clip.fq.gz", "rt").read(), (seed, tag)
    # ---- getsv passes through the -B harness ----
    juncs = []
    for _ in range(40):
        ta, tb = int(rng.randint(0, len(lens))), int(rng.randint(0, len(lens)))
        juncs.append((names[ta], int(rng.randint(1, lens[ta])), "+-"[int(rng.randint(0, 2))], names[tb], int(rng.randint(1, lens[tb])), "+-"[int(rng.randint(0, 2))]))
    juncs = [j for j in juncs if not (j[2] == "-" and j[5] == "-")]
    jfile = str(tmp_path / "j.txt")
    with open(jfile, "w") as f:
        for j in juncs:
            f.write("\t".join(str(x) for x in (j[0], j[1], j[2], 0, j[3], j[4], j[5], 0, 0, 0, "NA", 0, 0, 0, 0, 0, 0, 0, 0, "50M", "50M", "ACGT", "ACGT")) + "\n")
    empty_bam, empty_clip = str(tmp_path / "e.clip.bam"), str(tmp_path / "e.clip")
    bamio.write_bam(empty_bam, names, lens, [])
    open(empty_clip, "w").close()
    for q, L in ((20, 200), (0, 50)):
        sv, so = str(tmp_path / f"o{q}.sv"), str(tmp_path / f"o{q}.stdout")
        with open(so, "wb") as fo:
            rc = subprocess.call([REF, "getsv", "-d", "0", "-f", "0", "-b", "0", "-T", "100000", "-q", str(q), "-L", str(L), "-B", jfile, empty_bam, bam, empty_clip, sv, str(tmp_path / "x.fq")],
                                 stdout=fo, stderr=subprocess.DEVNULL)
        assert rc == 0
        rows = G.read_junction_file(jfile)
        stats, junctions, folded = G.run_getsv_case(bam, rows, OracleBackend(), min_mapq=q, flank_length=L, batch_records=600)
        golden = G.parse_sv_outputs(sv, so)
        assert G.check_getsv_against_golden(junctions, folded, golden) >= 3 * len(junctions)


@pytest.mark.parametrize("seed", range(300, 306))
def test_oracle_equals_reference_on_unsorted_samples(tmp_path, seed):
    """contigs that come back (a BAM that is not coordinate sorted): getclip flushes its maps at every change of contig (clip_reads.h:423-438), the
    oracle likewise - every output byte of the real reference"""
    names, lens, b, rng, recs = random_sample(seed, safe=True, want_records=True, unsorted=True)
    bam = str(tmp_path / "u.bam")
    write_sample(bam, names, lens, recs, index=False)
    hn, hl, batches = host.read_bam(bam, 700)
    for tag, flags, kw in (("a", [], {}), ("b", ["-t", "0.8", "-q", "0"], dict(match_rate=0.8, min_mapq=0)), ("c", ["-s", "-q", "20", "-t", "1"], dict(match_rate=1.0, save_low_quality=True, min_mapq=20))):
        pre = str(tmp_path / tag)
        r = subprocess.run([REF, "getclip"] + flags + ["-o", pre, bam], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        clip, fq = host.format_clip_outputs(O.getclip(batches, **kw), names)
        assert clip == gzip.open(pre + ".clip.gz", "rt").read(), (seed, tag)
        assert fq == gzip.open(pre + ".clip.fq.gz", "rt").read(), (seed, tag)
    assert r.stderr.count("Output merged soft-clipped reads of") > len(names)   # more flushes than contigs
