"""Pins the CPU oracle (oracle/seeksv_oracle.c) + the host bookkeeping (libseeksv_host.so) against outputs
of the REAL reference committed under tests/golden/ (made by tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from seeksv_amd import host

GETCLIP_CASES = [
    ("example", "cancer.sort.bam", "cancer", {}),
    ("example", "normal.sort.bam", "normal", {}),
    ("getclip", "filters.bam", "filters", {}),
    ("getclip", "filters.bam", "filters.s", dict(save_low_quality=True)),
    ("getclip", "filters.bam", "filters.q0", dict(min_mapq=0)),
    ("getclip", "filters.bam", "filters.q30", dict(min_mapq=30)),
    ("getclip", "stress1.bam", "stress1", {}),
    ("getclip", "stress1.bam", "stress1.t08", dict(match_rate=0.8)),
    ("getclip", "stress1.bam", "stress1.t1", dict(match_rate=1.0)),
    ("getclip", "stress2.bam", "stress2", {}),
    ("getclip", "stress3.bam", "stress3", {}),                          # thousands of reads per breakpoint, > 64 clusters per bin
    ("getclip", "stress3.bam", "stress3.t08", dict(match_rate=0.8)),
    ("getclip", "lone_s.bam", "lone_s", {}),                            # records whose whole CIGAR is one soft clip: two rows with an empty aligned part
    ("getclip", "lone_s2.bam", "lone_s2", {}),
    ("getclip", "lone_s2.bam", "lone_s2.s", dict(save_low_quality=True)),
    ("getclip", "lone_s2.bam", "lone_s2.q0", dict(min_mapq=0)),
    ("getclip", "unsorted.bam", "unsorted", {}),                        # contigs that come back: one flush per visit (clip_reads.h:423-438)
]

GETSV_CASES = [
    ("pairs1", "pairs1", dict()),
    ("pairs1", "pairs1.q0", dict(min_mapq=0)),
    ("pairs1", "pairs1.L50", dict(flank_length=50)),
    ("pairs1", "pairs1.L1", dict(flank_length=1)),
    ("pairs2", "pairs2", dict()),
    ("pairs3", "pairs3", dict()),
    ("eqx", "eqx", dict()),   # '=' / 'X' CIGAR operations in the depth pass (skipped by libbam 0.1.16's pileup)
    ("deep", "deep", dict()), # stacks of more than 8000 reads: the read cap of libbam 0.1.16's pileup (bam_plp_push)
]


class OracleBackend:
    def isize_stats(self, batches, min_mapq, max_pairs):
        return O.isize_stats(batches, min_mapq, max_pairs)

    def discordant_and_depth(self, batches, plan, mean, sd, min_mapq, target_lens):
        counts = O.discordant(batches, plan.junctions, mean, sd, 4, min_mapq)
        rs, pd, _ = O.depth(batches, plan.windows, plan.ranges, plan.points, min_mapq)
        return counts, rs, pd


@pytest.mark.parametrize("sub,bam,prefix,kw", GETCLIP_CASES, ids=[c[2] for c in GETCLIP_CASES])
@pytest.mark.parametrize("batch_records", [1 << 20, 1000])
def test_getclip_oracle_matches_reference(sub, bam, prefix, kw, batch_records):
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, sub, bam), batch_records)
    d = O.getclip(batches, **kw)
    clip, fq = host.format_clip_outputs(d, names)
    assert clip == G.read_text(sub, prefix + ".clip.txt")
    assert fq == G.read_text(sub, prefix + ".clip.fq.txt")


def lone_s_expected():
    """tests/golden/getclip/lone_s.*: the reference's output whole - a record whose CIGAR is one lone soft clip gives a '5' and a '3' row
    with an empty aligned part (it reads that operation as both ends of the CIGAR, clip_reads.cpp:115,150-190)"""
    rows = G.read_text("getclip", "lone_s.clip.txt")
    assert sum(1 for r in rows.splitlines() if r.split("\t")[3] == "") == 4
    return rows, G.read_text("getclip", "lone_s.clip.fq.txt")


def test_getclip_oracle_lone_soft_clip_records():
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, "getclip", "lone_s.bam"))
    clip, fq = host.format_clip_outputs(O.getclip(batches), names)
    assert (clip, fq) == lone_s_expected()


@pytest.mark.parametrize("sample", ["cancer", "normal"])
def test_isize_oracle_matches_reference_example(sample):
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, "example", sample + ".sort.bam"))
    rc, n, mean, sd = O.isize_stats(batches, 20, 5000000)
    exp = G.read_text("example", sample + ".isize.txt").split()
    assert rc == 0 and [str(mean), str(sd)] == exp


@pytest.mark.parametrize("case,prefix,kw", GETSV_CASES, ids=[c[1] for c in GETSV_CASES])
def test_getsv_passes_oracle_matches_reference(case, prefix, kw):
    base = os.path.join(G.GOLDEN, "getsv")
    rows = G.read_junction_file(os.path.join(base, case + ".junctions.txt"))
    stats, junctions, folded = G.run_getsv_case(os.path.join(base, case + ".bam"), rows, OracleBackend(), **kw)
    exp = G.read_text("getsv", prefix + ".isize.txt").split()
    assert [str(stats[2]), str(stats[3])] == exp
    golden = G.parse_sv_outputs(os.path.join(base, prefix + ".sv"), os.path.join(base, prefix + ".stdout"))
    assert sum(len(v) for v in golden.values()) == len(junctions)
    checked = G.check_getsv_against_golden(junctions, folded, golden)
    assert checked > 3 * len(junctions)


def test_example_sv_table_bam_columns():
    """Columns 10 and 12-17 of the full-pipeline table on example/cancer.sort.bam (the libbam-dependent ones)."""
    base = os.path.join(G.GOLDEN, "example")
    rows = []
    for line in G.read_text("example", "cancer.sv").splitlines():
        if line.startswith("@"):
            continue
        t = line.split("\t")
        rows.append((t[0], int(t[1]), t[2], t[4], int(t[5]), t[6], 0, int(t[3]), int(t[7]), line))
    stats, junctions, folded = G.run_getsv_case(os.path.join(base, "cancer.sort.bam"), [r[:7] for r in rows], OracleBackend())
    for i, r in enumerate(rows):
        t = r[9].split("\t")
        assert int(folded["abnormal"][i]) == int(t[9])
        assert int(folded["up_depth"][i]) + r[8] == int(t[11])      # updepth = depth(up) + down support (getsv.cpp:858)
        assert int(folded["down_depth"][i]) + r[7] == int(t[12])
        for c in range(4):
            assert int(folded["flank"][i, c]) // int(folded["flank_len"][i, c]) == int(t[13 + c])


@pytest.mark.parametrize("name", list(G.SYNTH_CASES))
def test_synthetic_workload_oracle_matches_reference(name):
    """The synthetic generator's records (regenerated here on the CPU) through the oracle == the real reference's outputs on the same records."""
    from seeksv_amd import synth
    w = synth.Workload(**G.SYNTH_CASES[name])
    b = w.generate_host(0, w.n_total)
    key = b["tid"].astype(np.int64) * (1 << 32) + b["pos"]
    assert np.all(np.diff(key) >= 0), "synthetic stream must be coordinate sorted"
    # split in uneven batches: cigar_off / seq_off must be rebased per batch
    cuts = [0, w.n_total // 3, w.n_total // 3 + 1001, w.n_total]
    batches = [split_batch(b, cuts[i], cuts[i + 1]) for i in range(3)]
    d = O.getclip(batches)
    clip, fq = host.format_clip_outputs(d, w.names)
    assert clip == G.read_gz("synth", name + ".clip.txt.gz")
    assert fq == G.read_gz("synth", name + ".clip.fq.txt.gz")
    hdr = host.Header(w.names, w.lens)
    rows = G.read_junction_file(os.path.join(G.GOLDEN, "synth", name + ".junctions.txt"))
    assert [r[:6] for r in rows] == w.junctions
    stats, junctions, folded = G.run_getsv_batches(hdr, batches, rows, OracleBackend())
    assert [str(stats[2]), str(stats[3])] == G.read_text("synth", name + ".isize.txt").split()
    golden = G.parse_sv_outputs(os.path.join(G.GOLDEN, "synth", name + ".sv"), os.path.join(G.GOLDEN, "synth", name + ".stdout"))
    assert G.check_getsv_against_golden(junctions, folded, golden) == 7 * len(junctions)
    hdr.close()


def split_batch(b, lo, hi):
    """Records [lo, hi) of a host batch as a self-contained batch."""
    out = {}
    for k in ("tid", "pos", "flag", "mapq", "n_cigar", "l_qseq", "mtid", "mpos", "isize"):
        out[k] = b[k][lo:hi].copy()
    c0 = int(b["cigar_off"][lo]) if lo < len(b["tid"]) else len(b["cigar"])
    c1 = int(b["cigar_off"][hi]) if hi < len(b["tid"]) else len(b["cigar"])
    out["cigar"] = b["cigar"][c0:c1].copy()
    out["cigar_off"] = (b["cigar_off"][lo:hi] - np.uint32(c0)).astype(np.uint32)
    so = b["seq_off"][lo:hi].copy()
    has = so != np.uint64(2 ** 64 - 1)
    if has.any():
        s0 = int(so[has].min())
        last = np.nonzero(has)[0][-1]
        lq = int(b["l_qseq"][lo + last])
        s1 = int(so[last]) + (lq + 1) // 2 + lq
        so[has] -= np.uint64(s0)
        out["seqqual"] = b["seqqual"][s0:s1].copy()
    else:
        out["seqqual"] = np.zeros(0, np.uint8)
    out["seq_off"] = so
    out["xc"] = None if b.get("xc") is None else b["xc"][lo:hi].copy()
    out["max_ref_span"] = b.get("max_ref_span", 0)
    return out
