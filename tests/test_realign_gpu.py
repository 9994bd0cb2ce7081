"""-m gpu: the clipped-sequence re-aligner (ssv_realign_*), the stand-in for the pipeline's external `bwa mem` step.
Checked against what bwa mem 0.7.10 itself produced for the same clipped sequences (the committed clip.bam fixtures of the
synthetic samples, made by tests/golden/make_golden.py with the reference's own example/bin/bwa), and by construction on sequences
cut from the hash-generated reference."""
import os

import numpy as np
import pytest

import bamio
import golden_util as G

pytestmark = pytest.mark.gpu

SAMPLES = {"synthfull": dict(genome_frac=1 / 8192, depth=40, n_sv=24), "hbvfull": dict(genome_frac=1 / 8192, depth=60, n_sv=8, n_integrations=10)}
COMP = str.maketrans("ACGTN", "TGCAN")


@pytest.fixture(scope="module")
def ctx():
    from seeksv_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _ref_string(w):
    return [("".join(l for l in part.splitlines()[1:])) for part in w.reference_fasta().split(">")[1:]]


@pytest.mark.parametrize("device_ref", [False, True], ids=["host-ref", "device-ref"])
def test_realign_constructed_queries(ctx, device_ref):
    """substrings of the reference, both strands, with mismatches, unrelated flanks, contig ends; random and too-short sequences"""
    from seeksv_amd import _abi, synth
    w = synth.Workload(genome_frac=1 / 8192, depth=40, n_sv=0, hbv=True)
    if device_ref:
        words, off = w.reference_2bit(0)
        dropped = ctx.realign_index(words.data_ptr(), off, _abi.MEM_DEVICE)
    else:
        words, off = w.reference_2bit()
        dropped = ctx.realign_index(words, off)
    assert dropped == 0
    ref = _ref_string(w)
    rng = np.random.RandomState(7)
    queries, expect = [], []
    for k in range(400):
        tid = int(rng.randint(len(ref)))
        n = int(rng.randint(80, 140))  # long enough that every variant below keeps an exact stretch of K + SAMPLE - 1 bases and a score >= 30
        p = int(rng.randint(0, len(ref[tid]) - n))
        s = list(ref[tid][p:p + n])
        kind = k % 5
        qb, qe = 0, n
        if kind == 1:      # two mismatches well inside
            for at in (n // 3, 2 * n // 3):
                s[at] = "ACGT"[("ACGT".index(s[at]) + 1) % 4]
        elif kind == 2:    # 12-base head that matches nowhere along the diagonal: soft clipped
            s[:12] = ["ACGT"[("ACGT".index(c) + 1 + int(x)) % 4] for c, x in zip(s[:12], rng.randint(0, 3, 12))]
            qb = 12
        elif kind == 3:    # mismatch 2 bases from the end: cheaper than clipping, stays aligned
            s[n - 3] = "ACGT"[("ACGT".index(s[n - 3]) + 2) % 4]
        s = "".join(s)
        rev = bool(k & 1)
        queries.append(s.translate(COMP)[::-1] if rev else s)
        expect.append((tid, p + qb, rev, qb, qe))
    hits = ctx.realign(queries)
    for h, (tid, pos, rev, qb, qe), q in zip(hits, expect, queries):
        assert h["tid"] == tid and h["pos"] == pos and bool(h["reverse"]) == rev and h["mapq"] == 60, (h, tid, pos, rev, q)
        assert (h["q_beg"], h["q_end"]) == (qb, qe), (h, qb, qe)
    # nothing to find
    junk = ["".join("ACGT"[x] for x in rng.randint(0, 4, 60)) for _ in range(50)] + ["ACGTACGTACGTACG", "", "N" * 40]
    hj = ctx.realign(junk)
    assert (hj["tid"] == -1).all()
    # a query hanging over a contig's end: 40 bases of the end of contig 3 followed by the first 60 of contig 4 -> the longer part wins,
    # the other part is soft clipped; and the mirror image (60 + 35)
    over = ctx.realign([ref[3][-40:] + ref[4][:60], ref[7][-60:] + ref[8][:35]])
    assert (over["tid"][0], over["pos"][0], over["q_beg"][0], over["q_end"][0]) == (4, 0, 40, 100)
    assert (over["tid"][1], over["pos"][1], over["q_beg"][1], over["q_end"][1]) == (7, len(ref[7]) - 60, 0, 60)
    assert over["second"][0] >= 40 and over["second"][1] >= 35  # the other contig's part is the runner-up
    # the last bases of the last contig (HBV) and the first of the first
    ends = [ref[-1][-50:], ref[0][:45], ref[-1][:40].translate(COMP)[::-1]]
    he = ctx.realign(ends)
    assert list(he["tid"]) == [len(ref) - 1, 0, len(ref) - 1] and list(he["pos"]) == [len(ref[-1]) - 50, 0, 0] and list(he["reverse"]) == [0, 0, 1]


@pytest.mark.parametrize("name", list(SAMPLES))
def test_realign_agrees_with_bwa_mem(ctx, name):
    """every clipped sequence bwa mem placed uniquely (MAPQ >= 20, >= 30 aligned bases, no indel) is placed on the same strand and
    diagonal; what bwa left unaligned or ambiguous is not reported as a confident hit"""
    from seeksv_amd import synth
    w = synth.Workload(**SAMPLES[name])
    words, off = w.reference_2bit()
    ctx.realign_index(words, off)
    names, recs = bamio.read_bam_records(os.path.join(G.GOLDEN, "synth", f"{name}.clip.bam"))
    assert names == list(w.names)
    prim = [r for r in recs if not r["flag"] & 0x900]
    hits = ctx.realign([r["qname"] for r in prim])
    n_conf = n_same = n_unal = n_unal_same = 0
    for r, h in zip(prim, hits):
        ops = "".join(op for _, op in r["cigar"])
        if r["flag"] & 4:
            n_unal += 1
            n_unal_same += int(h["tid"] == -1 or h["mapq"] < 60)
            continue
        aligned = sum(l for l, op in r["cigar"] if op == "M")
        if r["mapq"] < 20 or aligned < 30 or "I" in ops or "D" in ops:
            continue
        lead = r["cigar"][0][0] if r["cigar"][0][1] in "SH" else 0
        n_conf += 1
        n_same += int(h["tid"] == r["tid"] and bool(h["reverse"]) == bool(r["flag"] & 16) and h["pos"] - h["q_beg"] == r["pos"] - lead and h["mapq"] > 0)
    assert n_conf >= 30, n_conf  # the clusters at the planted breakpoints (random soft clips do not align anywhere)
    assert n_same >= 0.98 * n_conf, (n_same, n_conf)
    assert n_unal_same >= 0.9 * n_unal, (n_unal_same, n_unal)
    print(f"{name}: {n_same}/{n_conf} confident bwa placements reproduced, {n_unal_same}/{n_unal} unaligned agree")
