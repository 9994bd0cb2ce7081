"""-m gpu: BASELINE configs 2 and 3 at the record count bench.py runs (617 M records resident in HBM), checked by properties that do not
depend on the size - clip-event conservation, emit order, identical cluster tables under two chunkings of the input - and by the CPU
oracle on three 2 M-record slices (a contig's start, the neighbourhood of a planted junction, a contig's end): clusters inside the slice
must be the oracle's, row for row, and the discordant counts / depths of the junctions inside it too."""
import hashlib

import numpy as np
import pytest

import oracle_lib as O
from seeksv_amd import _abi, host

pytestmark = pytest.mark.gpu

SLICE = 2_000_000
MARGIN = 2000   # bp kept away from a slice's first / last record: reads outside the slice reach that far into it


@pytest.fixture(scope="module")
def ctx():
    from seeksv_amd.device import Context
    c = Context(0)
    c.clip_table_format(3)   # the bench's wire format
    yield c
    c.close()


def _sub(batch, first, n):
    """records [first, first + n) of a device batch: same CIGAR / base blobs, offsets are absolute"""
    arrays = {}
    for name, dt in _abi.BATCH_FIELDS:
        ptr = getattr(batch, name)
        arrays[name] = ptr if name in ("cigar", "seqqual", "xc") or ptr is None else ptr + first * np.dtype(dt).itemsize
    arrays["rec"] = batch.rec + first * 64 if batch.rec else None
    arrays["n_cigar_total"], arrays["seqqual_bytes"], arrays["max_ref_span"] = batch.n_cigar_total, batch.seqqual_bytes, batch.max_ref_span
    arrays["tid_runs"] = _abi.rebase_runs(batch.get_tid_runs(), first, n)
    return _abi.make_batch(arrays, mem=batch.mem, n=n)[0]


def _sha(d):
    h = hashlib.sha256()
    for k in ("tid", "pos", "side", "support", "left_len", "right_len", "qual_missing", "str_off", "str", "cigar_off", "n_cigar", "cigar"):
        h.update(np.ascontiguousarray(d[k]).tobytes())
    h.update(bytes([d["qual_bits"], d.get("base_bits", 4)]) + d["qual_alphabet"] + np.ascontiguousarray(d.get("base_exc", np.zeros(0, np.uint64))).tobytes())
    return h.hexdigest()


def _rows(d, tid, lo, hi):
    """clusters of contig tid with lo <= pos < hi as comparable tuples, in table order"""
    idx = np.nonzero((d["tid"] == tid) & (d["pos"] >= lo) & (d["pos"] < hi))[0]
    return [(int(d["pos"][k]), int(d["side"][k]), int(d["support"][k])) + host.cluster_strings(d, int(k)) for k in idx]


def _record_of(w, tid, pos):
    """index of (about) the first record starting at or after (tid, pos): the generator spaces starts evenly"""
    lin = int(w.offs[tid]) + pos
    return max(0, min(w.n_total - 1, (lin << 20) // int(w.cfg.spacing_fp)))


def _check_config(ctx, w):
    db, keep = w.generate_device(0, w.n_total, 0, soa=False, persistent=True)
    # ---- whole input at once, and cut into three batches at awkward places ----
    d = ctx.getclip([db])
    cuts = [0, (w.n_total // 3 + 12345) & ~15, (2 * (w.n_total // 3) + 777) & ~15, w.n_total]   # (device arrays are used in place: 16-byte aligned starts, also of the one-byte column)
    d3 = ctx.getclip([_sub(db, cuts[i], cuts[i + 1] - cuts[i]) for i in range(3)])
    assert d["n_events"] == d3["n_events"] and d["n_clusters"] == d3["n_clusters"]
    assert _sha(d) == _sha(d3)
    del d3
    assert int(d["support"].sum()) == d["n_events"] > 0                       # every clip event is in exactly one cluster
    key = d["tid"].astype(np.int64) << 33 | (d["side"] == ord("3")).astype(np.int64) << 32 | d["pos"].astype(np.int64)
    assert np.all(np.diff(key) >= 0)                                           # emit order: per contig the '5' rows by position, then the '3' rows
    assert d["support"].max() > 1                                              # planted breakpoints gather several reads
    # ---- the oracle on three slices ----
    hdr = host.Header(w.names, w.lens)
    jname = next(j for j in w.junctions if j[0] == j[3] and abs(j[4] - j[1]) < 100000 and 1_000_000 < j[1] < int(w.lens[w.names.index(j[0])]) - 1_000_000)
    jt = w.names.index(jname[0])
    g_j = _record_of(w, jt, jname[1])
    starts = [0, max(0, g_j - SLICE // 2), max(0, _record_of(w, 0, int(w.lens[0]) - 1) - SLICE + 1)]
    for g0 in starts:
        n = min(SLICE, w.n_total - g0)
        hb = w.generate_host(g0, n)
        t0, p0, t1, p1 = int(hb["tid"][0]), int(hb["pos"][0]), int(hb["tid"][-1]), int(hb["pos"][-1])
        o = O.getclip([hb], initial_last_tid=t0)
        tid = t0
        lo = 1 if g0 == 0 else p0 + MARGIN                                    # the file's start is a true boundary: compare from the first base
        hi = (p1 if t1 == tid else int(w.lens[tid]) + 1000) - (0 if t1 != tid else MARGIN)   # a slice that runs into the next contig holds this contig's end whole
        got, want = _rows(d, tid, lo, hi), _rows(o, tid, lo, hi)
        assert len(want) > 1000 and got == want
        # junctions inside the slice: their discordant pairs and depths come from records of the slice only
        inside = [j for j in w.junctions if j[0] == j[3] == w.names[tid] and lo + 3000 < min(j[1], j[4]) and max(j[1], j[4]) < hi - 3000]
        if inside:
            stats = O.isize_stats([hb], 20, 5000000)
            plan = host.Plan(hdr, inside, stats[2], stats[3])
            oc = O.discordant([hb], plan.junctions, stats[2], stats[3], 4, 20)
            ors, opd, _ = O.depth([hb], plan.windows, plan.ranges, plan.points, 20)
            c, r, p = ctx.discordant_and_depth([db], plan, stats[2], stats[3], 20, hdr.target_lens)
            assert np.array_equal(c, oc) and np.array_equal(r, ors) and np.array_equal(p, opd)
            assert c.sum() > 0
            plan.close()
    hdr.close()


def test_config2_30x_full_size(ctx):
    """BASELINE config 2: synthetic 30x WGS, 617,653,966 records, 1 % soft clips, 10,000 planted SVs (the bench workload)"""
    from seeksv_amd import synth
    _check_config(ctx, synth.Workload(genome_frac=1.0, depth=30, n_sv=10000))


def test_config3_300x_largest_in_a_minute(ctx):
    """BASELINE config 3's depth (300x tumor, planted DEL / INV / TRA) at the largest size that runs here in under a minute: a tenth of the
    genome, the same 617 M records; bins at planted breakpoints are ~150 reads deep"""
    from seeksv_amd import synth
    _check_config(ctx, synth.Workload(genome_frac=0.1, depth=300, n_sv=1000))


def test_config3_300x_full_size_6G_records():
    """BASELINE config 3 at its FULL size: 300x tumor WGS, 6.18 G records, 10,000 planted DEL / INV / TRA - more than fits in HBM at once, so the
    records are generated there chunk by chunk and streamed through the multi-batch API twice (getclip + insert size, then the fused getsv
    pass), once in 10 and once in 16 chunks (tools/config3_full_size.py): identical cluster tables (sha256 over every column) and identical
    tallies / depths, every clip event in exactly one cluster, every planted junction seen by the discordant tally and the depth pass - and
    the table's rows around a planted junction and at a contig's start are the CPU oracle's, row for row."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import config3_full_size as C3
    checked = []

    def probe(w, d):
        jname = next(j for j in w.junctions if j[0] == j[3] and abs(j[4] - j[1]) < 100000 and 1_000_000 < j[1] < int(w.lens[w.names.index(j[0])]) - 1_000_000)
        jt = w.names.index(jname[0])
        for g0 in (0, max(0, _record_of(w, jt, jname[1]) - SLICE // 2)):
            hb = w.generate_host(g0, SLICE)
            t0, p0, t1, p1 = int(hb["tid"][0]), int(hb["pos"][0]), int(hb["tid"][-1]), int(hb["pos"][-1])
            assert t0 == t1
            o = O.getclip([hb], initial_last_tid=t0)
            lo, hi = (1 if g0 == 0 else p0 + MARGIN), p1 - MARGIN
            got, want = _rows(d, t0, lo, hi), _rows(o, t0, lo, hi)
            assert len(want) > 1000 and got == want
            checked.append(len(want))
    out = C3.main(1.0, 300.0, 10000, 10, 16, probe=probe)
    assert out["records"] > 6_100_000_000 and out["results_identical"] and len(checked) == 2
    assert out["junctions_with_discordant_pairs"] == out["junctions"] == out["junctions_with_depth_at_both_ends"] == 10000
    assert out["table"]["support_sum"] == out["table"]["n_events"] > 50_000_000
