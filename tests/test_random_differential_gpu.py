"""-m gpu: randomized differential test, HIP path against the oracle (which the real reference pins on tests/golden).  Small batches with far
more variety per record than a sequencer produces: every CIGAR operation, clips of every length on either or both ends, odd read lengths,
missing qualities, XC flags, all flag bits, every MAPQ, stacks of identical starts (deep bins, near-identical and diverging sequences),
contig changes, unmapped-pair records in between - through getclip (three match rates, -s, -q), the insert-size pass, and the fused getsv
pass with random junction windows; each sample also cut into batches at random places."""
import numpy as np
import pytest

import oracle_lib as O
from seeksv_amd import host
from test_hip_golden import assert_tables_equal
from test_oracle_golden import split_batch

pytestmark = pytest.mark.gpu
OPS = "MIDNSHP=X"


@pytest.fixture(scope="module")
def ctx():
    from seeksv_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def random_sample(seed, n=1500, safe=False, want_records=False, unsorted=False):
    """safe: only inputs for which the REFERENCE's behaviour is defined (it reads out of bounds on missing qualities inside a multi-read bin).
    unsorted: the contigs' runs of records are cut in pieces and dealt out again, so that contigs come back (getclip flushes per visit,
    clip_reads.h:423-438); positions inside a piece stay in order"""
    rng = np.random.RandomState(seed)
    lens = [int(x) for x in (rng.randint(1000, 2500, 6) if unsorted else rng.randint(3000, 9000, 3))]   # (unsorted: short contigs, so that every sample holds several)
    recs = []
    tid, pos = 0, int(rng.randint(0, 50))
    templates = ["".join("ACGT"[x] for x in rng.randint(0, 4, 400)) for _ in range(4)]
    while len(recs) < n and tid < len(lens):
        pos += int(rng.choice([0, 0, 0, 1, 2, 5, 20, 60]))
        if pos > lens[tid] - 200:
            tid += 1; pos = int(rng.randint(0, 30)); continue
        lq = int(rng.choice([30, 51, 75, 100, 101, 150]))
        kind = rng.randint(0, 10)
        ops = []
        if kind <= 3:
            ops = [(lq, "M")]
        elif kind == 9 and rng.rand() < 0.4:
            ops = [(lq, "S")]   # a CIGAR that is one lone soft clip: both ends of the CIGAR at once (clip_reads.cpp:115,150-190), two rows with an empty aligned part
        else:
            left = int(rng.randint(1, lq // 2)) if rng.rand() < 0.6 else 0
            right = int(rng.randint(1, lq // 2 - 1)) if rng.rand() < 0.5 else 0
            lop = "S" if rng.rand() < 0.85 else "H"
            rop = "S" if rng.rand() < 0.85 else "H"
            mid = lq - (left if lop == "S" else 0) - (right if rop == "S" else 0)
            body = []
            remaining = mid
            while remaining > 0:
                op = str(rng.choice(list("MMMMM=XI")))
                l = int(min(remaining, rng.randint(1, 40)))
                body.append((l, op)); remaining -= l
                if rng.rand() < 0.25 and remaining > 0:
                    body.append((int(rng.randint(1, 30)), str(rng.choice(list("DNP")))))
            if safe:  # the pileup of libbam 0.1.16 asserts on CIGARs it does not expect: keep to M / I / D / N, aligned parts that start and end with M
                body = [(l, {"=": "M", "X": "M", "P": "D"}.get(op, op)) for l, op in body]
                body[0] = (body[0][0], "M")
                body[-1] = (body[-1][0], "M")
            if left: ops.append((left, lop))
            ops += body
            if right: ops.append((right, rop))
        # bases: stacks at one start share a template so that clusters absorb reads; some diverge
        t = templates[int(rng.randint(0, 4))]
        o = int(rng.randint(0, 40)) if rng.rand() < 0.3 else 7
        seq = list(t[o:o + lq])
        for _ in range(int(rng.choice([0, 0, 0, 1, 3, 12]))):
            seq[int(rng.randint(0, lq))] = "ACGTN"[int(rng.randint(0, 5))]
        qual = rng.choice([2, 11, 25, 37, 40], lq).astype(np.uint8) if (safe or rng.rand() < 0.93) else np.full(lq, 255, np.uint8)
        flag = int(rng.choice([99, 147, 83, 163, 97, 145, 65, 129, 113, 177, 73, 89, 133, 69]))
        for bit, pr in ((256, 0.03), (512, 0.03), (1024, 0.06), (2048, 0.03)):
            if rng.rand() < pr: flag |= bit
        mtid = tid if rng.rand() < 0.9 else int(rng.randint(0, len(lens)))
        mpos = max(0, pos + int(rng.randint(-600, 600)))
        isz = int(rng.choice([0, mpos - pos, 300 + int(rng.randint(-60, 60)), -(300 + int(rng.randint(-60, 60)))]))
        recs.append(dict(tid=tid, pos=pos, flag=flag, mapq=int(rng.choice([0, 1, 5, 19, 20, 30, 60, 60, 60])), ops=ops, lq=lq, seq="".join(seq), qual=qual,
                         mtid=mtid, mpos=mpos, isize=isz, xc=int(rng.rand() < 0.08)))
        # a stack on the same start: same CIGAR shape, nearly the same bases -> bins with several reads, absorbed or not depending on -t
        if len(ops) > 1:
            for _ in range(int(rng.choice([0, 0, 1, 2, 5, 9]))):
                r2 = dict(recs[-1])
                s2 = list(t[o:o + lq])
                for _k in range(int(rng.choice([0, 0, 1, 2, 8, 30]))):
                    s2[int(rng.randint(0, lq))] = "ACGT"[int(rng.randint(0, 4))]
                r2["seq"] = "".join(s2)
                r2["qual"] = rng.choice([2, 11, 25, 37, 40], lq).astype(np.uint8)
                r2["mapq"] = int(rng.choice([0, 20, 60, 60]))
                r2["flag"] = int(rng.choice([99, 147, 83, 163])) | (1024 if rng.rand() < 0.05 else 0)
                if rng.rand() < 0.4 and ops[0][1] == "S" and ops[0][0] > 3 and ops[1][1] == "M":   # a shorter / longer clip on the same breakpoint side keeps pos + 1
                    d = int(rng.randint(1, 3))
                    r2["ops"] = [(ops[0][0] - d, "S"), (ops[1][0] + d, "M")] + list(ops[2:])
                    r2["pos"] = pos - d
                recs.append(r2)
    order = sorted(range(len(recs)), key=lambda i: (recs[i]["tid"], recs[i]["pos"]))
    recs = [recs[i] for i in order if recs[i]["pos"] >= 0]
    if unsorted:
        cuts = sorted(set([0, len(recs)] + [int(x) for x in rng.randint(1, len(recs), 9)]))
        pieces = [recs[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
        rng.shuffle(pieces)
        recs = [r for piece in pieces for r in piece]
    b = dict(tid=np.array([r["tid"] for r in recs], np.int32), pos=np.array([r["pos"] for r in recs], np.int32),
             flag=np.array([r["flag"] for r in recs], np.uint16), mapq=np.array([r["mapq"] for r in recs], np.uint8),
             n_cigar=np.array([len(r["ops"]) for r in recs], np.uint16), l_qseq=np.array([r["lq"] for r in recs], np.int32),
             mtid=np.array([r["mtid"] for r in recs], np.int32), mpos=np.array([r["mpos"] for r in recs], np.int32),
             isize=np.array([r["isize"] for r in recs], np.int32), xc=np.array([r["xc"] for r in recs], np.uint8))
    cig, coff, soff, blob = [], [], [], []
    nbytes = 0
    span = 1
    code = {"=": 0, "A": 1, "C": 2, "G": 4, "T": 8, "N": 15}
    for r in recs:
        coff.append(len(cig))
        cig += [(l << 4) | OPS.index(op) for l, op in r["ops"]]
        span = max(span, sum(l for l, op in r["ops"] if op in "MDN=X"))
        if r["ops"][0][1] == "S" or r["ops"][-1][1] == "S" or seed % 3 == 0:
            soff.append(nbytes)
            lq = r["lq"]
            packed = np.zeros((lq + 1) // 2, np.uint8)
            for i, ch in enumerate(r["seq"]):
                packed[i >> 1] |= code[ch] << (0 if i & 1 else 4)
            blob += [packed, r["qual"]]
            nbytes += len(packed) + lq
        else:
            soff.append(0xFFFFFFFFFFFFFFFF)
    b.update(cigar=np.array(cig, np.uint32), cigar_off=np.array(coff, np.uint32), seq_off=np.array(soff, np.uint64),
             seqqual=np.concatenate(blob + [np.zeros(16, np.uint8)]) if blob else np.zeros(16, np.uint8), max_ref_span=int(span))
    names = [f"c{i}" for i in range(len(lens))]
    if want_records:
        return names, lens, b, rng, recs
    return names, lens, b, rng


@pytest.mark.parametrize("seed", list(range(64)) + list(range(100, 112)))
def test_random_sample_hip_equals_oracle(ctx, seed):
    # seeds 100-111 are the samples tests/test_random_oracle_vs_reference.py runs through the REAL reference (restricted to inputs it defines)
    names, lens, b, rng = random_sample(seed, safe=seed >= 100)
    n = len(b["tid"])
    cuts = sorted(set([0, n] + [int(x) for x in rng.randint(1, n, 3)]))
    parts = [split_batch(b, cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
    for kw in (dict(), dict(match_rate=0.8, min_mapq=0), dict(match_rate=1.0, save_low_quality=True, min_mapq=20)):
        want = O.getclip([b], **kw)
        assert_tables_equal(ctx.getclip([b], **kw), want)
        assert_tables_equal(ctx.getclip(parts, **kw), want)
    assert want["n_events"] > 50
    # the compact table decodes to the same strings (want: the last parameter set)
    ctx.clip_table_format(3)
    try:
        d = ctx.getclip(parts, **kw)
    finally:
        ctx.clip_table_format(0)
    assert d["n_clusters"] == want["n_clusters"]
    for k in range(0, d["n_clusters"], 3):
        assert host.cluster_strings(d, k) == host.cluster_strings(want, k), k
    for q, cap in ((20, 5000000), (0, 37)):
        assert ctx.isize_stats(parts, q, cap) == O.isize_stats([b], q, cap)
    stats = O.isize_stats([b], 20, 5000000)
    mean, sd = (stats[2], stats[3]) if stats[1] else (300, 40)
    juncs = []
    for _ in range(40):
        ta, tb = int(rng.randint(0, len(lens))), int(rng.randint(0, len(lens)))
        juncs.append((names[ta], int(rng.randint(1, lens[ta])), "+-"[int(rng.randint(0, 2))], names[tb], int(rng.randint(1, lens[tb])), "+-"[int(rng.randint(0, 2))]))
    juncs = [j for j in juncs if not (j[2] == "-" and j[5] == "-")]
    juncs.sort(key=lambda j: (j[0], j[3], j[2], j[5], j[1], j[4]))
    hdr = host.Header(names, lens)
    plan = host.Plan(hdr, juncs, mean, sd, flank_length=int(rng.choice([1, 50, 200])))
    for q in (20, 0):
        oc = O.discordant([b], plan.junctions, mean, sd, 4, q)
        ors, opd, _ = O.depth([b], plan.windows, plan.ranges, plan.points, q)
        for bs in ([b], parts):
            c, r, p = ctx.discordant_and_depth(bs, plan, mean, sd, q, hdr.target_lens)
            assert np.array_equal(c, oc) and np.array_equal(r, ors) and np.array_equal(p, opd)
    plan.close()
    hdr.close()


def dense_sample(seed, n=40000, contig_lens=(2500, 7000)):
    """tens of thousands of records on three short contigs (several hundred-fold depth), every kind of CIGAR the getsv passes tell apart: the per-candidate
    kernels' DENSE workgroups (k_getsv_cand_dense: >= 2048 candidates in a group of four scan tiles)"""
    rng = np.random.RandomState(9000 + seed)
    lens = [int(x) for x in rng.randint(contig_lens[0], contig_lens[1], 3)]
    tid = np.sort(rng.randint(0, 3, n)).astype(np.int32)
    pos = np.concatenate([np.sort(rng.randint(0, lens[t] - 160, int((tid == t).sum()))) for t in range(3)]).astype(np.int32)
    lq = rng.choice([50, 100, 150], n).astype(np.int32)
    kind = rng.choice(7, n, p=[0.62, 0.1, 0.1, 0.06, 0.05, 0.04, 0.03])
    cig, coff, ncig = [], [], []
    span = 1
    for i in range(n):
        L, k = int(lq[i]), int(kind[i])
        a = int(rng.randint(1, L - 1))
        if k == 0: ops = [(L, "M")]
        elif k == 1: ops = [(a, "S"), (L - a, "M")] if rng.rand() < 0.5 else [(a, "M"), (L - a, "S")]
        elif k == 2: ops = [(a, "M"), (int(rng.randint(1, 60)), "D"), (L - a, "M")]
        elif k == 3: ops = [(a, "M"), (int(rng.choice([5, 300, 3000, 9000])), "N"), (L - a, "M")]   # a skip that leaves the workgroup's columns
        elif k == 4: ops = [(a, "M"), (1, "I"), (L - a - 1, "M")] if L - a > 1 else [(L, "M")]
        elif k == 5:  # more operations than a record's line holds (the batch's cigar array is read)
            q = L // 8
            ops = [(q, "M"), (1, "I"), (q, "M"), (3, "D"), (q, "M"), (2, "I"), (q, "=" if rng.rand() < 0.3 else "M"), (7, "D"), (L - 4 * q - 3, "M")]
        else: ops = [(3, "H"), (L, "M")] if rng.rand() < 0.5 else [(L, "M"), (2, "H")]
        coff.append(len(cig)); ncig.append(len(ops))
        cig += [(l << 4) | OPS.index(op) for l, op in ops]
        span = max(span, sum(l for l, op in ops if op in "MDN=X"))
    flag = rng.choice([99, 147, 83, 163, 97, 145, 65, 129, 113, 177, 73, 89, 133, 69], n).astype(np.uint16)
    for bit, pr in ((256, 0.03), (512, 0.03), (1024, 0.06), (2048, 0.03)):
        flag |= (rng.rand(n) < pr).astype(np.uint16) * np.uint16(bit)
    mtid = np.where(rng.rand(n) < 0.8, tid, rng.randint(0, 3, n)).astype(np.int32)
    mpos = np.maximum(0, pos + rng.randint(-700, 700, n)).astype(np.int32)
    isz = np.choose(rng.randint(0, 4, n), [np.zeros(n, np.int64), mpos - pos, 300 + rng.randint(-60, 60, n), -(300 + rng.randint(-60, 60, n))]).astype(np.int32)
    if seed & 1:  # contigs come back: pieces of the sorted file dealt out again (positions inside a piece stay in order)
        cuts = sorted(set([0, n] + [int(x) for x in rng.randint(1, n, 5)]))
        order = np.concatenate([np.arange(cuts[i], cuts[i + 1]) for i in rng.permutation(len(cuts) - 1)])
    else:
        order = np.arange(n)
    flat, off2 = [], []
    for i in order:
        off2.append(len(flat)); flat += cig[coff[i]:coff[i] + ncig[i]]
    b = dict(tid=tid[order], pos=pos[order], flag=flag[order], mapq=rng.choice([0, 1, 5, 19, 20, 30, 60, 60, 60], n).astype(np.uint8), n_cigar=np.array(ncig, np.uint16)[order],
             l_qseq=lq[order], mtid=mtid[order], mpos=mpos[order], isize=isz[order], xc=np.zeros(n, np.uint8), cigar=np.array(flat, np.uint32), cigar_off=np.array(off2, np.uint32),
             seq_off=np.full(n, 0xFFFFFFFFFFFFFFFF, np.uint64), seqqual=np.zeros(16, np.uint8), max_ref_span=int(span))
    return [f"c{i}" for i in range(3)], lens, b, rng


@pytest.mark.parametrize("seed", range(6))
def test_dense_candidates_hip_equals_oracle(ctx, seed):
    """Workgroups whose tiles are dense with candidates keep their range's tile bits and junction windows in LDS and put the reads' depth into a difference array over
    columns there (k_getsv_cand_dense): several hundred-fold depth, up to ~70 junction windows inside one workgroup's range (more than it keeps), reads that leave
    the range (N skips), contigs that change inside a workgroup and (odd seeds) come back, CIGARs longer than the five operations of a record's line"""
    names, lens, b, rng = dense_sample(seed)
    n = len(b["tid"])
    cuts = sorted(set([0, n] + [int(x) for x in rng.randint(1, n, 2)]))
    parts = [split_batch(b, cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
    juncs = []
    for _ in range(60 if seed < 4 else 400):
        ta, tb = int(rng.randint(0, 3)), int(rng.randint(0, 3))
        juncs.append((names[ta], int(rng.randint(1, lens[ta])), "+-"[int(rng.randint(0, 2))], names[tb], int(rng.randint(1, lens[tb])), "+-"[int(rng.randint(0, 2))]))
    juncs = [j for j in juncs if not (j[2] == "-" and j[5] == "-")]
    juncs.sort(key=lambda j: (j[0], j[3], j[2], j[5], j[1], j[4]))
    hdr = host.Header(names, lens)
    plan = host.Plan(hdr, juncs, 300, 40, flank_length=int(rng.choice([1, 50, 200])))
    if seed >= 4:   # more junction windows on one contig than a dense workgroup keeps in LDS (32); seed 4 (flank 1): more depth windows inside one 4096-column range than it keeps (16)
        assert max(int((plan.junctions["up_tid"] == t).sum()) for t in range(3)) > 64
    if seed == 4:
        wb = np.sort(plan.windows["beg"][plan.windows["tid"] == 0])
        assert max(int(((wb >= x) & (wb < x + 4096)).sum()) for x in wb) > 32
    for q in (20, 0):
        oc = O.discordant([b], plan.junctions, 300, 40, 4, q)
        ors, opd, _ = O.depth([b], plan.windows, plan.ranges, plan.points, q)
        assert oc.sum() > 0 and ors.sum() > 0
        for bs in ([b], parts):
            c, r, p = ctx.discordant_and_depth(bs, plan, 300, 40, q, hdr.target_lens)
            assert np.array_equal(c, oc) and np.array_equal(r, ors) and np.array_equal(p, opd)
    plan.close()
    hdr.close()


@pytest.mark.parametrize("seed,n", [(20, 20000), (21, 4096 * 3), (22, 4096 * 2 + 1), (23, 4097)])
def test_sparse_records_with_tid_runs_hip_equals_oracle(ctx, seed, n):
    """k_getsv_scan_runs (batches of a tile or more that come with their tid column as runs): records so sparse that a wavefront's 1024 start further apart than the
    64 genome tiles of its fast path (the record-by-record path inside whole tiles), run boundaries inside tiles, contigs that come back (odd seeds), batches of a
    whole number of tiles, of one record more, of a tile and a record"""
    names, lens, b, rng = dense_sample(seed, n=n, contig_lens=(1500000, 4000000))
    from seeksv_amd import _abi
    assert len(_abi.runs_of_tid(b["tid"])) <= 48
    juncs = []
    for _ in range(300):
        ta, tb = int(rng.randint(0, 3)), int(rng.randint(0, 3))
        juncs.append((names[ta], int(rng.randint(1, lens[ta])), "+-"[int(rng.randint(0, 2))], names[tb], int(rng.randint(1, lens[tb])), "+-"[int(rng.randint(0, 2))]))
    juncs = [j for j in juncs if not (j[2] == "-" and j[5] == "-")]
    juncs.sort(key=lambda j: (j[0], j[3], j[2], j[5], j[1], j[4]))
    hdr = host.Header(names, lens)
    plan = host.Plan(hdr, juncs, 300, 40, flank_length=200)
    cuts = sorted(set([0, n] + [int(x) for x in rng.randint(1, n, 2)]))
    parts = [split_batch(b, cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
    for q in (20, 0):
        oc = O.discordant([b], plan.junctions, 300, 40, 4, q)
        ors, opd, _ = O.depth([b], plan.windows, plan.ranges, plan.points, q)
        for bs in ([b], parts):
            c, r, p = ctx.discordant_and_depth(bs, plan, 300, 40, q, hdr.target_lens)
            assert np.array_equal(c, oc) and np.array_equal(r, ors) and np.array_equal(p, opd)
    assert ors.sum() > 0
    plan.close()
    hdr.close()


def test_scan_with_an_empty_first_contig(ctx):
    """a header whose first contig has no bases (no genome tiles): the runs-only scan kernel's tile-map byte is loaded in every step, also for tiles that take
    the other path - its address must stay inside the map"""
    names, lens, b, rng = dense_sample(30, n=3 * 4096 + 7)
    names, lens = ["empty"] + names, [0] + lens
    for k in ("tid", "mtid"):
        b[k] = (b[k] + 1).astype(np.int32)
    juncs = []
    for _ in range(80):
        ta, tb = int(rng.randint(1, 4)), int(rng.randint(1, 4))
        juncs.append((names[ta], int(rng.randint(1, lens[ta])), "+-"[int(rng.randint(0, 2))], names[tb], int(rng.randint(1, lens[tb])), "+-"[int(rng.randint(0, 2))]))
    juncs = [j for j in juncs if not (j[2] == "-" and j[5] == "-")]
    juncs.sort(key=lambda j: (j[0], j[3], j[2], j[5], j[1], j[4]))
    hdr = host.Header(names, lens)
    plan = host.Plan(hdr, juncs, 300, 40, flank_length=50)
    oc = O.discordant([b], plan.junctions, 300, 40, 4, 20)
    ors, opd, _ = O.depth([b], plan.windows, plan.ranges, plan.points, 20)
    c, r, p = ctx.discordant_and_depth([b], plan, 300, 40, 20, hdr.target_lens)
    assert np.array_equal(c, oc) and np.array_equal(r, ors) and np.array_equal(p, opd) and ors.sum() > 0
    plan.close()
    hdr.close()


@pytest.mark.parametrize("seed", list(range(200, 212)) + list(range(300, 306)))
def test_random_unsorted_sample_hip_equals_oracle(ctx, seed):
    """contigs that come back: one flush per visit (clip_reads.h:423-438) - the passes end where the driver sees a contig again (ssv_clip_scan_range);
    seeds 300-305 are the samples tests/test_random_oracle_vs_reference.py runs through the REAL reference"""
    names, lens, b, rng = random_sample(seed, safe=seed >= 300, unsorted=True)
    n = len(b["tid"])
    cuts = sorted(set([0, n] + [int(x) for x in rng.randint(1, n, 3)]))
    parts = [split_batch(b, cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
    for kw in (dict(), dict(match_rate=0.8, min_mapq=0), dict(match_rate=1.0, save_low_quality=True, min_mapq=20)):
        want = O.getclip([b], **kw)
        assert_tables_equal(ctx.getclip([b], **kw), want)
        assert_tables_equal(ctx.getclip(parts, **kw), want)
    assert want["n_events"] > 50 and np.any(np.diff(want["tid"]) < 0)   # the table really holds a contig twice
