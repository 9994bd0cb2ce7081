"""-m gpu: the HIP path (through the C ABI of libseeksv_hip.so) against the reference's own outputs
(tests/golden/) and against the CPU oracle on the same inputs.  Bit-exact: this is integer / byte work;
the two fp64 match-rate compares are done with the same IEEE division on both sides."""
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from seeksv_amd import _abi, host
from test_oracle_golden import GETCLIP_CASES, GETSV_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from seeksv_amd.device import Context
    c = Context(0)
    yield c
    c.close()


TABLE_KEYS = ("tid", "pos", "side", "support", "left_len", "right_len", "qual_missing", "str_off", "str", "cigar_off", "n_cigar", "cigar")


def assert_tables_equal(a, b):
    assert a["n_clusters"] == b["n_clusters"] and a["n_events"] == b["n_events"]
    for k in TABLE_KEYS:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("sub,bam,prefix,kw", GETCLIP_CASES, ids=[c[2] for c in GETCLIP_CASES])
@pytest.mark.parametrize("batch_records", [1 << 20, 1000])
def test_getclip_hip_matches_reference(ctx, sub, bam, prefix, kw, batch_records):
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, sub, bam), batch_records)
    d = ctx.getclip(batches, **kw)
    clip, fq = host.format_clip_outputs(d, names)
    assert clip == G.read_text(sub, prefix + ".clip.txt")
    assert fq == G.read_text(sub, prefix + ".clip.fq.txt")
    assert_tables_equal(d, O.getclip(batches, **kw))


def _compact_checks(ctx, d, ref, t, check_size=True):
    """a format-3 table against the ASCII table of the same input: every rebuilt column, every decoded string, the C-side rebuild
    (ssv_clip_table_expand) against the numpy one, and the size of what crossed PCIe"""
    assert d["format"] == 3 and d["n_clusters"] == ref["n_clusters"] and d["n_events"] == ref["n_events"]
    for k in ("tid", "pos", "side", "support", "left_len", "right_len", "qual_missing", "cigar_off", "n_cigar", "cigar"):
        assert np.array_equal(d[k], ref[k]), k
    assert all(host.cluster_strings(d, k) == host.cluster_strings(ref, k) for k in range(d["n_clusters"]))
    n = d["n_clusters"]
    if n == 0:
        return
    ctx.clip_table_expand(t, 3)
    for name, dt in (("tid", np.int32), ("side", np.uint8), ("support", np.int32), ("left_len", np.int32), ("right_len", np.int32), ("qual_missing", np.uint8),
                     ("str_off", np.uint64), ("cigar_off", np.uint64), ("n_cigar", np.int32)):
        assert np.array_equal(np.ctypeslib.as_array(getattr(t, name), shape=(n,)), d[name].astype(dt)), name
    lib = _abi.hip_lib()
    assert all(lib.ssv_table_block_bytes3g(int(a) + int(b), d["base_bits"], d["qual_bits"], d["qual_group"]) ==
               (int(d["str_off"][k + 1]) if k + 1 < n else len(d["str"])) - int(d["str_off"][k]) for k, (a, b) in enumerate(zip(d["left_len"], d["right_len"])))
    assert bool(t.cigar) == (d["cigar_bytes"] == 4)   # 16-bit operations are read through c_cigar
    assert d["cigar_bytes"] == (2 if int(d["cigar"].max(initial=0)) < 65536 else 4)
    wire = n * (4 + 2 * d["len_bytes"] + d["support_bytes"] + d["ncig_bytes"] + 1) + len(d["str"]) + d["cigar_bytes"] * len(d["cigar"]) + 16 * len(d["runs"]) + 8 * len(d["base_exc"])
    wire_ascii = n * 42 + len(ref["str"]) + 4 * len(ref["cigar"])
    assert wire < 0.7 * wire_ascii or n < 50 or not check_size
    assert t.support_sum == d["n_events"] == int(d["support"].sum())


@pytest.mark.parametrize("sub,bam,prefix,kw", GETCLIP_CASES, ids=[c[2] for c in GETCLIP_CASES])
def test_getclip_compact_table_format(ctx, sub, bam, prefix, kw):
    """ssv_clip_table_format 3, the compact table: 12 bytes of fixed columns per cluster, contig / side as runs, no offsets, one 2-bit base
    stream and one quality stream per cluster, bases outside A/C/G/T as exceptions; rebuilt on the host it is the ASCII table (and the
    reference's rows)"""
    if prefix == "unsorted":
        pytest.skip("several passes: their tables are put together in the ASCII format (the CLI tests cover the compact one)")
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, sub, bam), 1 << 20)
    ref = ctx.getclip(batches, **kw)
    ctx.clip_table_format(3)
    try:
        ctx.clip_begin(**kw)
        for b in batches:
            ctx.clip_scan(b)
        t = ctx.clip_cluster(as_dict=False)
        d = host.table_to_dict(t)
        _compact_checks(ctx, d, ref, t)
    finally:
        ctx.clip_table_format(0)
    clip, fq = host.format_clip_outputs(d, names)
    assert clip == G.read_text(sub, prefix + ".clip.txt") and fq == G.read_text(sub, prefix + ".clip.fq.txt")


def _with_bases(b, code_of):
    """a copy of a host batch whose packed bases were rewritten: code_of(record, i, code) -> new 4-bit code"""
    b = dict(b)
    sq = b["seqqual"].copy()
    for r in np.nonzero(b["seq_off"] != np.uint64(0xFFFFFFFFFFFFFFFF))[0]:
        o, lq = int(b["seq_off"][r]), int(b["l_qseq"][r])
        for i in range(lq):
            sh = 0 if i & 1 else 4
            c = (int(sq[o + (i >> 1)]) >> sh) & 15
            sq[o + (i >> 1)] = (int(sq[o + (i >> 1)]) & ~(15 << sh)) | (code_of(int(r), i, c) << sh)
    b["seqqual"] = sq
    return b


@pytest.mark.parametrize("source", ["stress1", "stress2", "filters", "synth150", "synth400"])
@pytest.mark.parametrize("mode", ["acgt", "some_n", "iupac", "overflow"])
def test_compact_table_bases(ctx, source, mode, monkeypatch):
    """the 2-bit base streams: reads of only A/C/G/T (no exceptions), scattered N (exceptions), every IUPAC code and '=' (exceptions on nearly
    every base), and more exceptions than the list takes (SSV_EXC_CAP: the pass falls back to 4-bit streams) - decoded == ASCII == oracle"""
    if source.startswith("synth"):
        from seeksv_amd import synth
        w = synth.Workload(genome_frac=1 / 4096, depth=30, n_sv=12, read_len=int(source[5:]))
        batches = [w.generate_host(0, w.n_total)]
    else:
        batches = host.read_bam(os.path.join(G.GOLDEN, "getclip", source + ".bam"))[2]
    acgt = [1, 2, 4, 8]
    f = {"acgt": lambda r, i, c: c if c in acgt else acgt[(r + i) & 3],
         "some_n": lambda r, i, c: 15 if (r * 31 + i * 7) % 23 == 0 else (c if c in acgt else acgt[(r + i) & 3]),
         "iupac": lambda r, i, c: (r * 5 + i * 3) % 16,
         "overflow": lambda r, i, c: 15 if (r + i) % 3 == 0 else (c if c in acgt else 1)}[mode]
    batches = [_with_bases(b, f) for b in batches]
    ref = ctx.getclip(batches)
    assert_tables_equal(ref, O.getclip(batches))
    if mode == "overflow":
        monkeypatch.setenv("SSV_EXC_CAP", "64")
    ctx.clip_table_format(3)
    try:
        ctx.clip_begin()
        for b in batches:
            ctx.clip_scan(b)
        t = ctx.clip_cluster(as_dict=False)
        d = host.table_to_dict(t)
        _compact_checks(ctx, d, ref, t, check_size=mode != "iupac")   # (an exception costs 8 bytes: a small table of reads without A/C/G/T is bigger than ASCII)
    finally:
        ctx.clip_table_format(0)
    if mode == "iupac":   # nearly every base is an exception: a big enough sample overflows the list by itself
        assert (d["base_bits"] == 2 and len(d["base_exc"]) > d["n_clusters"]) or (d["base_bits"] == 4 and len(d["base_exc"]) == 0)
    else:
        assert d["base_bits"] == (4 if mode == "overflow" else 2)
        assert (len(d["base_exc"]) == 0) == (mode in ("acgt", "overflow"))


@pytest.mark.parametrize("n_values,bits,group,first", [(1, 1, 1, 3), (2, 1, 1, 3), (4, 2, 1, 3), (5, 7, 3, 3), (5, 7, 3, 47), (5, 3, 1, 3), (8, 3, 1, 3), (9, 7, 2, 3), (11, 7, 2, 3), (10, 7, 2, 40),
                                                      (11, 4, 1, 3), (12, 4, 1, 3), (16, 4, 1, 3), (17, 8, 1, 3), (17, 11, 2, 3), (17, 11, 2, -2), (40, 11, 2, 3), (40, 11, 2, -2), (41, 11, 2, -20),
                                                      (45, 11, 2, -2), (45, 11, 2, 3), (46, 8, 1, 3)])
@pytest.mark.parametrize("source", ["stress1", "filters", "synth150", "synth300", "synth400"])
def test_compact_table_quality_alphabets(ctx, monkeypatch, source, n_values, bits, group, first):
    """format 3 with every quality index width - and the grouped forms (five values: three qualities to 7 bits; nine to eleven: two to 7 bits; 17 to 45 - a
    HiSeq-style 40-value alphabet -: two to 11 bits; with SSV_QUAL_GROUPS=0 one field per quality as before; an alphabet that reaches phred 64 takes the
    LDS-staged kernel, first < 0: consecutive values from -first on, all below 64: the direct kernel) -, on deep bins (consensus storage), odd clip offsets,
    reads longer than the kernel's LDS-staged limit, missing qualities"""
    if group == 1 and n_values in (5, 9, 10, 11, 17):
        monkeypatch.setenv("SSV_QUAL_GROUPS", "0")
    if source.startswith("synth"):
        from seeksv_amd import synth
        w = synth.Workload(genome_frac=1 / 4096, depth=30, n_sv=12, read_len=int(source[5:]))
        batches = [w.generate_host(0, w.n_total)]
    else:
        batches = host.read_bam(os.path.join(G.GOLDEN, "getclip", source + ".bam"))[2]
    alphabet = [(first + 5 * k) % 94 for k in range(n_values)] if first >= 0 else [-first + k for k in range(n_values)]
    batches = [_remap_qualities(b, alphabet) for b in batches]
    ref = ctx.getclip(batches)
    ctx.clip_table_format(3)
    try:
        ctx.clip_begin()
        for b in batches:
            ctx.clip_scan(b)
        t = ctx.clip_cluster(as_dict=False)
        d = host.table_to_dict(t)
        assert (d["qual_bits"], d["qual_group"]) == (bits, group), (d["qual_bits"], d["qual_group"], bits, group)
        _compact_checks(ctx, d, ref, t)
    finally:
        ctx.clip_table_format(0)


def test_compact_table_wide_cigar(ctx):
    """a CIGAR operation of 4096 bases or more (a long N): the table's operations stay 32 bits wide; without one they cross PCIe in 16"""
    n, lq = 40, 50
    packed = np.array([0x12, 0x48] * 12 + [0x12], np.uint8)   # ACGT x 12, AC
    entry = np.concatenate([packed, np.full(lq, 30, np.uint8)])
    for skip, width in ((4095, 2), (4096, 4), (70000, 4)):
        cig = np.tile(np.array([(20 << 4) | 4, (10 << 4) | 0, (skip << 4) | 3, (20 << 4) | 0], np.uint32), n)
        b = dict(tid=np.zeros(n, np.int32), pos=(1000 + 7 * np.arange(n)).astype(np.int32), flag=np.full(n, 99, np.uint16), mapq=np.full(n, 60, np.uint8), n_cigar=np.full(n, 4, np.uint16),
                 l_qseq=np.full(n, lq, np.int32), mtid=np.zeros(n, np.int32), mpos=np.full(n, 1200, np.int32), isize=np.full(n, 250, np.int32), xc=np.zeros(n, np.uint8),
                 cigar=cig, cigar_off=(4 * np.arange(n)).astype(np.uint32),
                 seq_off=(np.arange(n) * len(entry)).astype(np.uint64), seqqual=np.concatenate([np.tile(entry, n), np.zeros(16, np.uint8)]), max_ref_span=30 + skip)
        ref = ctx.getclip([b])
        assert ref["n_clusters"] == n
        ctx.clip_table_format(3)
        try:
            ctx.clip_begin()
            ctx.clip_scan(b)
            t = ctx.clip_cluster(as_dict=False)
            d = host.table_to_dict(t)
            assert d["cigar_bytes"] == width
            _compact_checks(ctx, d, ref, t, check_size=False)
        finally:
            ctx.clip_table_format(0)


def test_compact_table_wide_support(ctx):
    """a bin with more than 65535 identical reads: the support column switches to 32 bits"""
    n, lq = 70000, 50
    packed = np.array([0x12, 0x48] * 12 + [0x12], np.uint8)   # ACGT x 12, AC
    entry = np.concatenate([packed, np.full(lq, 30, np.uint8)])
    b = dict(tid=np.zeros(n, np.int32), pos=np.full(n, 1000, np.int32), flag=np.full(n, 99, np.uint16), mapq=np.full(n, 60, np.uint8), n_cigar=np.full(n, 2, np.uint16),
             l_qseq=np.full(n, lq, np.int32), mtid=np.zeros(n, np.int32), mpos=np.full(n, 1200, np.int32), isize=np.full(n, 250, np.int32), xc=np.zeros(n, np.uint8),
             cigar=np.tile(np.array([(20 << 4) | 4, (30 << 4) | 0], np.uint32), n), cigar_off=(2 * np.arange(n)).astype(np.uint32),
             seq_off=(np.arange(n) * len(entry)).astype(np.uint64), seqqual=np.concatenate([np.tile(entry, n), np.zeros(16, np.uint8)]), max_ref_span=30)
    ref = ctx.getclip([b])
    assert ref["support"].max() == 70000
    ctx.clip_table_format(3)
    try:
        ctx.clip_begin()
        ctx.clip_scan(b)
        t = ctx.clip_cluster(as_dict=False)
        d = host.table_to_dict(t)
        assert d["support_bytes"] == 4
        _compact_checks(ctx, d, ref, t)
    finally:
        ctx.clip_table_format(0)


@pytest.mark.parametrize("sample", ["cancer", "normal"])
def test_isize_hip_matches_reference_example(ctx, sample):
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, "example", sample + ".sort.bam"), 5000)
    rc, n, mean, sd = ctx.isize_stats(batches, 20, 5000000)
    assert rc == 0 and [str(mean), str(sd)] == G.read_text("example", sample + ".isize.txt").split()
    assert (rc, n, mean, sd) == O.isize_stats(batches, 20, 5000000)
    # the cut-off after max_pairs qualifying records (cluster.cpp:68)
    for cap in (1, 1000, 4097):
        assert ctx.isize_stats(batches, 20, cap) == O.isize_stats(batches, 20, cap)


@pytest.mark.parametrize("case,prefix,kw", GETSV_CASES, ids=[c[1] for c in GETSV_CASES])
@pytest.mark.parametrize("batch_records", [1 << 20, 3000])
def test_getsv_passes_hip_matches_reference(ctx, case, prefix, kw, batch_records):
    base = os.path.join(G.GOLDEN, "getsv")
    rows = G.read_junction_file(os.path.join(base, case + ".junctions.txt"))
    stats, junctions, folded = G.run_getsv_case(os.path.join(base, case + ".bam"), rows, ctx, batch_records=batch_records, **kw)
    assert [str(stats[2]), str(stats[3])] == G.read_text("getsv", prefix + ".isize.txt").split()
    golden = G.parse_sv_outputs(os.path.join(base, prefix + ".sv"), os.path.join(base, prefix + ".stdout"))
    assert G.check_getsv_against_golden(junctions, folded, golden) > 3 * len(junctions)


def test_no_seq_and_empty_batches(ctx):
    names, lens, batches = host.read_bam(os.path.join(G.GOLDEN, "getclip", "filters.bam"))
    empty = {k: v[:0] for k, v in batches[0].items() if isinstance(v, np.ndarray)}
    d = ctx.getclip([empty, batches[0], empty])
    assert_tables_equal(d, O.getclip([batches[0]]))
    d0 = ctx.getclip([empty])
    assert d0["n_clusters"] == 0 and d0["n_events"] == 0


def _device_batch_to_host(t, n_cigar_total, seqqual_bytes):
    import torch
    out = {}
    for k, dt in (("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("mapq", np.uint8), ("n_cigar", np.uint16), ("l_qseq", np.int32),
                  ("mtid", np.int32), ("mpos", np.int32), ("isize", np.int32), ("cigar_off", np.uint32), ("seq_off", np.uint64)):
        out[k] = t[k].cpu().numpy().view(dt)
    out["cigar"] = t["cigar"].cpu().numpy().view(np.uint32)[:n_cigar_total]
    out["seqqual"] = t["seqqual"].cpu().numpy()[:seqqual_bytes]
    return out


@pytest.mark.parametrize("name", list(G.SYNTH_CASES))
def test_synthetic_workload_hip_device_resident(ctx, name):
    """Records generated straight into HBM (the bench path): identical to the CPU generator's, and the HIP path on the
    device-resident batch reproduces the real reference's clip table / discordant counts / depths."""
    from seeksv_amd import synth
    from test_oracle_golden import split_batch
    w = synth.Workload(**G.SYNTH_CASES[name])
    hb = w.generate_host(0, w.n_total)
    db, tensors = w.generate_device(0, w.n_total, 0)
    got = _device_batch_to_host(tensors, db.n_cigar_total, db.seqqual_bytes)
    for k in got:
        assert np.array_equal(got[k], hb[k]), k
    d = ctx.getclip([db])
    clip, fq = host.format_clip_outputs(d, w.names)
    assert clip == G.read_gz("synth", name + ".clip.txt.gz")
    assert fq == G.read_gz("synth", name + ".clip.fq.txt.gz")
    hdr = host.Header(w.names, w.lens)
    rows = G.read_junction_file(os.path.join(G.GOLDEN, "synth", name + ".junctions.txt"))
    stats, junctions, folded = G.run_getsv_batches(hdr, [db], rows, ctx)
    assert [str(stats[2]), str(stats[3])] == G.read_text("synth", name + ".isize.txt").split()
    golden = G.parse_sv_outputs(os.path.join(G.GOLDEN, "synth", name + ".sv"), os.path.join(G.GOLDEN, "synth", name + ".stdout"))
    assert G.check_getsv_against_golden(junctions, folded, golden) == 7 * len(junctions)
    # the same through host batches cut at awkward places
    cuts = [0, 777, w.n_total // 2 + 3, w.n_total]
    hbs = [split_batch(hb, cuts[i], cuts[i + 1]) for i in range(3)]
    assert_tables_equal(ctx.getclip(hbs), d)
    hdr.close()


def test_full_size_properties(ctx):
    """Size-independent checks on a batch too large for the reference fixtures: clip-event conservation, support sums,
    depth linearity (two passes over the same batch double every sum) and agreement with the oracle."""
    from seeksv_amd import synth
    w = synth.Workload(genome_frac=1 / 64, depth=30, n_sv=300)
    n = min(w.n_total, 6_000_000)
    db, tensors = w.generate_device(0, n, 0)
    d = ctx.getclip([db])
    assert int(d["support"].sum()) == d["n_events"]
    key = d["tid"].astype(np.int64) << 33 | (d["side"] == ord("3")).astype(np.int64) << 32 | d["pos"].astype(np.int64)
    assert np.all(np.diff(key) >= 0)
    hb = w.generate_host(0, n)
    assert_tables_equal(d, O.getclip([hb]))
    hdr = host.Header(w.names, w.lens)
    rc, npairs, mean, sd = ctx.isize_stats([db], 20, 5000000)
    assert (rc, npairs, mean, sd) == O.isize_stats([hb], 20, 5000000)
    plan = host.Plan(hdr, w.junctions, mean, sd)
    c1, r1, p1 = ctx.discordant_and_depth([db], plan, mean, sd, 20, hdr.target_lens)
    c2, r2, p2 = ctx.discordant_and_depth([db, db], plan, mean, sd, 20, hdr.target_lens)
    assert np.array_equal(2 * c1, c2) and np.array_equal(2 * r1, r2) and np.array_equal(2 * p1, p2)
    oc = O.discordant([hb], plan.junctions, mean, sd, 4, 20)
    ors, opd, _ = O.depth([hb], plan.windows, plan.ranges, plan.points, 20)
    assert np.array_equal(c1, oc) and np.array_equal(r1, ors) and np.array_equal(p1, opd)
    assert c1.sum() > 0 and r1.sum() > 0
    plan.close(), hdr.close()


def _remap_qualities(batch, alphabet):
    """a copy of the batch whose base qualities are drawn from `alphabet` (deterministically, from the old value and the position)"""
    b = dict(batch)
    sq = batch["seqqual"].copy()
    alphabet = np.asarray(alphabet, dtype=np.uint8)
    for off, lq in zip(batch["seq_off"], batch["l_qseq"]):
        if off == np.uint64(0xFFFFFFFFFFFFFFFF) or lq <= 0:
            continue
        q0 = int(off) + (int(lq) + 1) // 2
        if sq[q0] == 0xFF:
            continue  # qualities absent: stays absent
        old = sq[q0:q0 + lq].astype(np.int64)
        sq[q0:q0 + lq] = alphabet[(old * 7 + np.arange(lq)) % len(alphabet)]
    b["seqqual"] = sq
    return b


@pytest.mark.parametrize("kw", [dict(genome_frac=1 / 4096, depth=60, n_sv=30), dict(genome_frac=1 / 8192, depth=60, n_sv=8, n_integrations=10)], ids=["wgs", "hbv"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_range_partitioned_hip_equals_single(ctx, kw, world):
    """BASELINE configs 4 / 5 (the BAM range-partitioned over up to 8 GPUs) on one GPU: every rank's share - its records plus the leading
    halo for getclip, exactly its records for the getsv passes - goes through the HIP path in turn (device-resident batches, ownership by
    breakpoint position), the per-rank result vectors are merged the way the all-gather merges them, the cluster tables are concatenated:
    equal to the unpartitioned run, bin for bin."""
    from seeksv_amd import shard, synth
    from _shard_worker import cluster_rows
    w = synth.Workload(**kw)
    hdr = host.Header(w.names, w.lens)
    whole, keep = w.generate_device(0, w.n_total, 0)
    single = ctx.getclip([whole])
    stats = ctx.isize_stats([whole], 20, 100000)
    plan = host.Plan(hdr, w.junctions, stats[2], stats[3])
    c1, r1, p1 = ctx.discordant_and_depth([whole], plan, stats[2], stats[3], 20, hdr.target_lens)
    rows, vecs, per_rank = [], [], []
    for rank in range(world):
        sp = shard.shard_plan(w, rank, world)
        scan, k1 = w.generate_device(sp["scan_lo_rec"], sp["own_hi_rec"] - sp["scan_lo_rec"], 0)
        own, k2 = w.generate_device(sp["own_lo_rec"], sp["own_hi_rec"] - sp["own_lo_rec"], 0)
        d = ctx.getclip([scan], own=sp["own"], initial_last_tid=sp["initial_last_tid"])
        c, r, p = ctx.discordant_and_depth([own], plan, stats[2], stats[3], 20, hdr.target_lens)
        vecs.append(shard.pack_results(c, r, p, d["n_clusters"], d["n_events"], int(d["support"].sum())))
        rows += cluster_rows(d)
        per_rank.append(d["n_events"])
    m = shard.merge_results(np.stack(vecs), len(c1), len(r1), len(p1))
    assert np.array_equal(m[0], c1) and np.array_equal(m[1], r1) and np.array_equal(m[2], p1)
    assert m[3] == single["n_clusters"] and m[4] == single["n_events"] == m[5]
    assert sum(1 for n in per_rank if n > 0) >= min(world, 2)
    assert sorted(rows, key=lambda r: r[:3]) == sorted(cluster_rows(single), key=lambda r: r[:3])
    plan.close()
    hdr.close()


def test_config3_streaming_chunking_independent():
    """BASELINE config 3's shape (300x, planted SVs) streamed through the path in HBM-sized chunks under two chunkings (tools/config3_full_size.py
    runs this at the full 6.2 G records): identical cluster table (sha256 of every column) and identical tallies / depths wherever the
    batches are cut; event conservation; every planted junction seen by both getsv passes."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import config3_full_size as C3
    out = C3.main(genome_frac=1 / 128, depth=300, n_sv=80, chunks_a=3, chunks_b=7)
    assert out["results_identical"] and out["junctions_with_discordant_pairs"] == out["junctions"] == 80
    assert out["records"] > 40_000_000 and out["table"]["max_support"] > 50


@pytest.mark.parametrize("batch_records", [1000, 4096, 7777, 8192, 20000, 50000])
def test_pileup_read_cap_across_batch_boundaries(ctx, batch_records):
    """stacks of more than 8000 reads (tests/golden/getsv/deep.*: the read cap of libbam 0.1.16's pileup) streamed in batches of awkward
    sizes: the sweep's carried state and the look-back into the previous batch's last records give the reference's depths every time"""
    base = os.path.join(G.GOLDEN, "getsv")
    rows = G.read_junction_file(os.path.join(base, "deep.junctions.txt"))
    stats, junctions, folded = G.run_getsv_case(os.path.join(base, "deep.bam"), rows, ctx, batch_records=batch_records)
    golden = G.parse_sv_outputs(os.path.join(base, "deep.sv"), os.path.join(base, "deep.stdout"))
    assert G.check_getsv_against_golden(junctions, folded, golden) == 7 * len(junctions)
    assert int(folded["up_depth"].max()) == 7999


def test_pileup_read_cap_synthetic_12000x(ctx):
    """a synthetic sample at 12000x (the regime of a viral contig or chrM in a deep sample): every record is a "deep" record, the one-wavefront
    sweep runs over the whole batch and over batch boundaries; depths, flank sums and discordant tallies equal the oracle's (whose cap is
    pinned by the real reference on tests/golden/getsv/deep.*), and the cap really binds"""
    from seeksv_amd import synth
    w = synth.Workload(genome_frac=1 / 65536, depth=12000, n_sv=6, n_contigs=3, min_contig=12000)
    hdr = host.Header(w.names, w.lens)
    hb = w.generate_host(0, w.n_total)
    stats = O.isize_stats([hb], 20, 100000)
    plan = host.Plan(hdr, w.junctions, stats[2], stats[3])
    oc = O.discordant([hb], plan.junctions, stats[2], stats[3], 4, 20)
    ors, opd, omax = O.depth([hb], plan.windows, plan.ranges, plan.points, 20)
    assert 7800 < int(opd.max()) < 8100             # capped: ~10,400 reads per column pass the filter
    db, keep = w.generate_device(0, w.n_total, 0)
    c, r, p = ctx.discordant_and_depth([db], plan, stats[2], stats[3], 20, hdr.target_lens)
    assert np.array_equal(c, oc) and np.array_equal(r, ors) and np.array_equal(p, opd)
    # the same in host batches cut at awkward places
    from test_oracle_golden import split_batch
    cuts = [0, 5000, 5000 + 12345, w.n_total // 2 + 7, w.n_total]
    hbs = [split_batch(hb, cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
    c2, r2, p2 = ctx.discordant_and_depth(hbs, plan, stats[2], stats[3], 20, hdr.target_lens)
    assert np.array_equal(c2, oc) and np.array_equal(r2, ors) and np.array_equal(p2, opd)
    plan.close()
    hdr.close()


def _with_cigar(hb, k, ops):
    """copy of a host batch whose record k carries the CIGAR `ops` (list of (len, op)) instead of its own"""
    out = {n: (v.copy() if isinstance(v, np.ndarray) else v) for n, v in hb.items()}
    off, nc = hb["cigar_off"].astype(np.int64), hb["n_cigar"].astype(np.int64)
    new = np.array([(l << 4) | o for l, o in ops], dtype=np.uint32)
    out["cigar"] = np.concatenate([hb["cigar"][:off[k]], new, hb["cigar"][off[k] + nc[k]:]])
    out["n_cigar"][k] = len(new)
    out["cigar_off"] = np.concatenate([[0], np.cumsum(out["n_cigar"].astype(np.int64))[:-1]]).astype(np.uint32)
    out.pop("cigar_ends", None)   # (the optional hot copy of the CIGAR ends would be stale: without it the pass reads n_cigar)
    return out


def test_pileup_ring_grows_with_a_later_long_read(ctx):
    """the ring of read ends behind the pileup's read cap is sized by the longest reference span seen so far: a first batch of plain 150 bp
    reads (ring of 8192 columns), then a batch holding a read with a 20 kb `N` skip in the middle of a >8000x stack, while the cap's
    sweep is running across the batch boundary.  The pass used to fail with SSV_E_RANGE here; now the ring grows and the live entries move."""
    from seeksv_amd import synth
    from test_oracle_golden import split_batch
    w = synth.Workload(genome_frac=1 / 65536, depth=12000, n_sv=6, n_contigs=3, min_contig=12000)
    hdr = host.Header(w.names, w.lens)
    hb = w.generate_host(0, w.n_total)
    cut = w.n_total // 2 + 7
    k = next(i for i in range(cut + 5000, w.n_total) if hb["n_cigar"][i] == 1 and hb["mapq"][i] >= 20 and not (hb["flag"][i] & 0x704) and hb["tid"][i] == hb["tid"][cut])
    hb2 = _with_cigar(hb, k, [(50, 0), (20000, 3), (100, 0)])
    stats = O.isize_stats([hb2], 20, 100000)
    plan = host.Plan(hdr, w.junctions, stats[2], stats[3])
    oc = O.discordant([hb2], plan.junctions, stats[2], stats[3], 4, 20)
    ors, opd, omax = O.depth([hb2], plan.windows, plan.ranges, plan.points, 20)
    a, b = split_batch(hb2, 0, cut), split_batch(hb2, cut, w.n_total)
    a["max_ref_span"], b["max_ref_span"] = 158, 20150
    c, r, p = ctx.discordant_and_depth([a, b], plan, stats[2], stats[3], 20, hdr.target_lens)
    assert np.array_equal(c, oc) and np.array_equal(r, ors) and np.array_equal(p, opd)
    assert 7800 < int(opd.max()) < 8100
    plan.close()
    hdr.close()


@pytest.mark.parametrize("far", [False, True])
def test_right_clip_list_sort_paths(ctx, far):
    """the '3' events of a sorted BAM are nearly in key order (key = alignment end): 6000 right-clipped reads whose ends run backwards inside
    windows of a few dozen reads are repaired by the windowed rank pass (tile_sort.h); with one read whose 2 Mb `N` skip throws its event
    5000 places back the check fails and the radix sort takes over - both equal the oracle"""
    n, lq = 6000, 60
    rng = np.random.RandomState(11)
    pos = np.sort(rng.randint(1000, 3_000_000, n)).astype(np.int32)
    pos = np.maximum.accumulate(pos)
    clip = rng.randint(5, 25, n)
    m = lq - clip
    dele = rng.randint(0, 60, n)                                   # a deletion: the end moves by up to 60 bp
    cig, coff = [], []
    for i in range(n):
        coff.append(len(cig))
        if i == 7 and far:
            cig += [(20 << 4) | 0, (2_000_000 << 4) | 3, ((int(m[i]) - 20) << 4) | 0, (int(clip[i]) << 4) | 4]
        else:
            cig += [(20 << 4) | 0, (int(dele[i]) << 4) | 2, ((int(m[i]) - 20) << 4) | 0, (int(clip[i]) << 4) | 4]
    ncig = np.full(n, 4, np.uint16)
    bases = rng.randint(0, 4, (n, lq))
    packed = ((1 << bases[:, 0::2]) << 4 | (1 << bases[:, 1::2])).astype(np.uint8)
    qual = rng.choice([11, 25, 37], (n, lq)).astype(np.uint8)
    entry = np.concatenate([packed, qual], axis=1)
    b = dict(tid=np.zeros(n, np.int32), pos=pos, flag=np.full(n, 99, np.uint16), mapq=np.full(n, 60, np.uint8), n_cigar=ncig, l_qseq=np.full(n, lq, np.int32),
             mtid=np.zeros(n, np.int32), mpos=pos + 200, isize=np.full(n, 260, np.int32), xc=np.zeros(n, np.uint8), cigar=np.array(cig, np.uint32),
             cigar_off=np.array(coff, np.uint32), seq_off=(np.arange(n) * entry.shape[1]).astype(np.uint64),
             seqqual=np.concatenate([entry.reshape(-1), np.zeros(16, np.uint8)]), max_ref_span=2_000_100 if far else 200)
    want = O.getclip([b])
    assert want["n_events"] == n and np.all(want["side"] == ord("3"))
    ends = pos.astype(np.int64) + 20 + dele + (m - 20)
    assert np.any(np.diff(ends) < 0)                               # the list really is out of key order
    assert_tables_equal(ctx.getclip([b]), want)


@pytest.mark.parametrize("source", ["stress1", "filters", "synth150"])
def test_compact_table_wide_columns(ctx, source, monkeypatch):
    """the column widths of format 3 grow with the data (u32 lengths for reads over 64 kb, u32 support, u16 CIGAR counts): forced here
    (SSV_TABLE_WIDE_COLUMNS), the rebuilt table is the same"""
    if source.startswith("synth"):
        from seeksv_amd import synth
        w = synth.Workload(genome_frac=1 / 4096, depth=30, n_sv=12, read_len=int(source[5:]))
        batches = [w.generate_host(0, w.n_total)]
    else:
        batches = host.read_bam(os.path.join(G.GOLDEN, "getclip", source + ".bam"))[2]
    ref = ctx.getclip(batches)
    monkeypatch.setenv("SSV_TABLE_WIDE_COLUMNS", "1")
    ctx.clip_table_format(3)
    try:
        ctx.clip_begin()
        for b in batches:
            ctx.clip_scan(b)
        t = ctx.clip_cluster(as_dict=False)
        d = host.table_to_dict(t)
        assert (d["len_bytes"], d["support_bytes"], d["ncig_bytes"]) == (4, 4, 2)
        _compact_checks(ctx, d, ref, t, check_size=False)
    finally:
        ctx.clip_table_format(0)


def test_getsv_scan_with_and_without_tid_runs(ctx, monkeypatch):
    """ssv_batch_t.tid_runs (the tid column as runs: the streaming pass of getsv then reads pos only, 4 B/record instead of 8): same tallies and depths
    as with the column read (SSV_NO_TID_RUNS), on a device batch of several contigs whose boundaries fall inside the kernel's 4096-record tiles,
    whole and as sub-batches with rebased runs; malformed run lists are refused"""
    import ctypes as C
    from seeksv_amd import synth
    w = synth.Workload(genome_frac=1 / 2048, depth=30, n_sv=60, n_contigs=9, min_contig=20000)
    db, keep = w.generate_device(0, w.n_total, 0)
    runs = db.get_tid_runs()
    assert runs is not None and len(runs) == 9 and runs["first"][0] == 0 and np.all(runs["first"][1:] % 4096 != 0)
    hb = w.generate_host(0, w.n_total)
    hdr = host.Header(w.names, w.lens)
    stats = O.isize_stats([hb], 20, 5000000)
    plan = host.Plan(hdr, w.junctions, stats[2], stats[3])
    want = (O.discordant([hb], plan.junctions, stats[2], stats[3], 4, 20),) + O.depth([hb], plan.windows, plan.ranges, plan.points, 20)[:2]

    def run(batches):
        c, r, p = ctx.discordant_and_depth(batches, plan, stats[2], stats[3], 20, hdr.target_lens)
        assert np.array_equal(c, want[0]) and np.array_equal(r, want[1]) and np.array_equal(p, want[2])
    run([db])
    monkeypatch.setenv("SSV_NO_TID_RUNS", "1")
    run([db])
    monkeypatch.delenv("SSV_NO_TID_RUNS")
    # sub-batches (16-record aligned starts) with rebased runs
    cuts = [0, 4096 * 3 + 16, (w.n_total // 2) & ~15, w.n_total]
    parts = []
    for i in range(3):
        arrays = {name: (getattr(db, name) if name in ("cigar", "seqqual", "xc") or getattr(db, name) is None else getattr(db, name) + cuts[i] * np.dtype(dt).itemsize) for name, dt in _abi.BATCH_FIELDS}
        arrays["rec"] = db.rec + cuts[i] * 64
        arrays["n_cigar_total"], arrays["seqqual_bytes"], arrays["max_ref_span"] = db.n_cigar_total, db.seqqual_bytes, db.max_ref_span
        arrays["tid_runs"] = _abi.rebase_runs(runs, cuts[i], cuts[i + 1] - cuts[i])
        parts.append(_abi.make_batch(arrays, mem=db.mem, n=cuts[i + 1] - cuts[i])[0])
        assert parts[-1].get_tid_runs()["first"][0] == 0
    run(parts)
    # malformed lists are refused, not believed
    lib = _abi.hip_lib()
    ctx.getsv_begin(plan.junctions, plan.windows, stats[2], stats[3], hdr.target_lens, 4, 20, 20)
    for bad in (np.array([(5, 0, 0)], _abi.TID_RUN_DTYPE), np.array([(0, 0, 0), (0, 1, 0)], _abi.TID_RUN_DTYPE), np.array([(0, 0, 0), (w.n_total, 1, 0)], _abi.TID_RUN_DTYPE)):
        db.set_tid_runs(bad)
        assert lib.ssv_getsv_scan(ctx._h, C.byref(db)) == -3
    # ... and so are well-formed lists that do not say what the tid column says (SSV_VERIFY_RUNS=1, set for the whole suite in conftest.py: one
    # streaming pass holds the column against the list before the scan relies on it): a boundary one record late, one record early, a wrong
    # contig for a run, a missing run
    shifted, early, wrong, missing = runs.copy(), runs.copy(), runs.copy(), np.delete(runs, 4)
    shifted["first"][3] += 1
    early["first"][8] -= 1
    wrong["tid"][5] = int(runs["tid"][5]) + 1
    for bad in (shifted, early, wrong, missing):
        db.set_tid_runs(bad)
        assert lib.ssv_getsv_scan(ctx._h, C.byref(db)) == -3 and b"disagree with the tid column" in lib.ssv_last_error(ctx._h)
    db.set_tid_runs(runs)
    assert lib.ssv_getsv_scan(ctx._h, C.byref(db)) == 0
    plan.close()
    hdr.close()


@pytest.mark.parametrize("read_len", [31, 100, 251, 255, 256, 257, 300])
@pytest.mark.parametrize("match_rate", [0.9, 0.8, 1.0])
def test_deep_bins_at_the_edges_of_the_four_positions_per_lane_walk(ctx, read_len, match_rate):
    """k_cluster_bins4 (round 5) walks a bin with four positions per lane for reads of up to 256 bases (one round of the 64 lanes; strings as dwords; the
    match-rate compare as a table of the smallest passing count) and leaves longer reads to k_cluster_bins: deep bins (300x over a small genome: 50-150
    clipped reads per planted breakpoint, and thousands of two- and three-event bins) of reads just below, at and above that length, of lengths that are
    no multiple of four, under three match rates - tables equal to the oracle's, ASCII and compact"""
    from seeksv_amd import synth
    w = synth.Workload(genome_frac=1 / 16384, depth=300, n_sv=6, read_len=read_len)
    batches = [w.generate_host(0, w.n_total)]
    want = O.getclip(batches, match_rate=match_rate)
    assert want["n_events"] > 2000 and int(want["support"].max()) >= 20
    got = ctx.getclip(batches, match_rate=match_rate)
    assert_tables_equal(got, want)
    ctx.clip_table_format(3)
    try:
        d = ctx.getclip(batches, match_rate=match_rate)
    finally:
        ctx.clip_table_format(0)
    assert d["n_clusters"] == want["n_clusters"]
    for k in range(0, d["n_clusters"], 7):
        assert host.cluster_strings(d, k) == host.cluster_strings(got, k), k
