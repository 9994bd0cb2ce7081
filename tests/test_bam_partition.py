"""CPU: cutting a BAM file into N runs of records (libseeksv_host, bam_partition.cpp): the parts' own ranges tile the file record for record,
every boundary is a record start, a part's halo holds exactly the records that start within halo_bp before its first record on the same
contig, initial_last_tid is the contig of the last mapped-pair record before the scan start, and walking back N records lands where a
forward count says."""
import os

import numpy as np
import pytest

import golden_util as G
from seeksv_amd import host, synth

KEYS = ("tid", "pos", "flag", "mapq", "n_cigar", "l_qseq", "mtid", "mpos", "isize")


def _read_range(path, start, end, batch=50000):
    out = []
    with host.BamReader(path) as r:
        r.set_range(start, end)
        while True:
            b = r.read_batch(batch)
            if b is None:
                break
            out.append(b)
    return out


def _cat(batches, k):
    return np.concatenate([b[k] for b in batches]) if batches else np.zeros(0)


@pytest.fixture(scope="module")
def synth_bam(tmp_path_factory):
    w = synth.Workload(genome_frac=1 / 2048, depth=30, n_sv=20, n_contigs=5, min_contig=20000)
    path = str(tmp_path_factory.mktemp("part") / "s.bam")
    hb = w.generate_host(0, w.n_total)
    host.write_bam(path, w.names, w.lens, [hb])
    return path, hb


def _check(path, whole, n_parts, halo_bp):
    parts = host.partition(path, n_parts, halo_bp)
    assert len(parts) == n_parts and parts[0]["scan"] == parts[0]["own"] and parts[-1]["end"] is None
    got = []
    first_index = []
    n = 0
    for p in parts:
        bs = _read_range(path, p["own"], p["end"])
        first_index.append(n)
        n += sum(len(b["tid"]) for b in bs)
        got += bs
    for k in KEYS:
        assert np.array_equal(_cat(got, k), whole[k]), (n_parts, k)
    tid, pos, flag = whole["tid"], whole["pos"], whole["flag"]
    for r, p in enumerate(parts):
        i0 = first_index[r]
        if r and parts[r - 1]["end"] is not None:
            assert parts[r - 1]["end"] == p["own"]
        if i0 >= len(tid):
            continue
        if tid[i0] >= 0:
            assert (p["own_tid"], p["own_pos"]) == (int(tid[i0]), int(pos[i0]))
        if r == 0:
            assert p["halo_records"] == 0 and p["initial_last_tid"] == 0
            continue
        # brute force: the halo and the last mapped-pair record before it
        h = i0
        while h > 0 and tid[i0] >= 0 and tid[h - 1] == tid[i0] and pos[h - 1] >= pos[i0] - halo_bp:
            h -= 1
        assert p["halo_records"] == i0 - h
        hb = _read_range(path, p["scan"], p["own"])
        assert sum(len(b["tid"]) for b in hb) == i0 - h
        if i0 - h:
            assert np.array_equal(_cat(hb, "pos"), pos[h:i0])
        j = h - 1
        while j >= 0 and (flag[j] & 12):
            j -= 1
        assert p["initial_last_tid"] == (int(tid[j]) if j >= 0 else 0)
        j = i0 - 1
        while j >= 0 and (flag[j] & 12):
            j -= 1
        assert p["before_own_tid"] == (int(tid[j]) if j >= 0 else 0)
        # walking back from the part's first record
        for nb in (1, 7, 1000, 10 ** 9):
            at, walked = host.walk_back(path, p["own"], nb)
            assert walked == min(nb, i0)
            tail = _read_range(path, at, p["own"])
            assert sum(len(b["tid"]) for b in tail) == walked


@pytest.mark.parametrize("n_parts", [1, 2, 3, 5, 8])
def test_partition_synthetic(synth_bam, n_parts):
    path, hb = synth_bam
    _check(path, hb, n_parts, 1000)
    _check(path, hb, n_parts, 0)


@pytest.mark.parametrize("n_parts", [2, 4])
def test_partition_example_and_crafted(n_parts, tmp_path):
    for sub, name in (("example", "cancer.sort.bam"), ("getclip", "filters.bam"), ("getsv", "pairs1.bam"), ("getsv", "deep.bam")):
        path = os.path.join(G.GOLDEN, sub, name)
        names, lens, batches = host.read_bam(path)
        whole = {k: _cat(batches, k) for k in KEYS}
        _check(path, whole, n_parts, 500)


def test_partition_more_parts_than_records(tmp_path):
    path = os.path.join(G.GOLDEN, "getclip", "filters.bam")
    names, lens, batches = host.read_bam(path)
    whole = {k: _cat(batches, k) for k in KEYS}
    _check(path, whole, 64, 100)


def test_partition_false_record_start_two_bytes_early(tmp_path):
    """Equal-sized records on contig 0, built so that two bytes ahead of EVERY record start a plausible header begins (the previous record's
    last two bytes + the low half of block_size read as a 17.5 MB record with a 4-byte name and a position inside the contig) whose single
    hop lands exactly on a true record start 64,573 records later: local speculation alone takes that false start.  The boundaries must
    still be true record starts (two independently started chains have to agree), the parts must tile the file."""
    import struct
    import bamio
    S = 32 + 2 + 4 + 75 + 150 + 5                        # block_size of every record
    k = -(-((S << 16) + 2) // (S + 4))
    tail2 = k * (S + 4) - 2 - (S << 16)                  # the u16 at the end of every record: 2 + (tail2 | S << 16) + 4 - 2 = k records
    assert 0 <= tail2 < 65536
    n = k + 40000
    seq = "ACGT" * 37 + "AC"
    recs = [dict(qname="a", flag=99, tid=0, pos=0x040000 + min(i, 30000), mapq=60, cigar="150M", mtid=0, mpos=0x040000 + min(i, 30000) + 100, isize=250, seq=seq,
                 qual=bytes([30] * 150), aux=b"XXS" + struct.pack("<H", tail2)) for i in range(n)]
    path = str(tmp_path / "equal.bam")
    bamio.write_bam(path, ["c0"], [0x7fffffff], recs)
    names, lens, batches = host.read_bam(path)
    whole = {kk: _cat(batches, kk) for kk in KEYS}
    assert len(whole["tid"]) == n
    # the trap is there: the bytes two ahead of a record start pass for a record header whose hop ends on a record start
    import gzip
    raw = gzip.open(path, "rb").read()
    first = raw.index(struct.pack("<I", S) + struct.pack("<i", 0))
    t = first + 10 * (S + 4)
    bs = struct.unpack_from("<I", raw, t - 2)[0]
    assert (t - 2 + 4 + bs - first) % (S + 4) == 0 and raw[t - 2 + 4 + 8] == 4 and raw[t - 2 + 4 + 32 + 4 - 1] == 0
    for n_parts in (3, 7, 13):
        parts = host.partition(path, n_parts, 100)
        n_seen = 0
        for r, p in enumerate(parts):
            bs_ = _read_range(path, p["own"], p["end"])
            i0 = n_seen
            n_seen += sum(len(b["tid"]) for b in bs_)
            if i0 < n:
                assert (p["own_tid"], p["own_pos"]) == (0, int(whole["pos"][i0]))
            assert i0 == 0 or abs(i0 - n * r // n_parts) < 300       # and they are where the balance puts them, not a hop later
        assert n_seen == n


@pytest.mark.parametrize("n_parts", [2, 3, 5])
def test_partition_with_bai(n_parts, monkeypatch):
    """the bundled example BAMs come with their .bai (the index the reference loads, seeksv.cpp:272-280): the boundaries are then record starts of
    its linear index - no speculation - and every property of the index-free cut holds for them too; SSV_NO_BAI switches back"""
    from seeksv_amd import _abi
    lib = _abi.host_lib()
    for name in ("cancer.sort.bam", "normal.sort.bam"):
        path = os.path.join(G.GOLDEN, "example", name)
        assert os.path.exists(path + ".bai")
        names, lens, batches = host.read_bam(path)
        whole = {k: _cat(batches, k) for k in KEYS}
        monkeypatch.delenv("SSV_NO_BAI", raising=False)
        _check(path, whole, n_parts, 500)
        assert lib.ssvh_partition_used_index() == 1
        with_index = host.partition(path, n_parts, 500)
        monkeypatch.setenv("SSV_NO_BAI", "1")
        _check(path, whole, n_parts, 500)
        assert lib.ssvh_partition_used_index() == 0
        # both ways the parts tile the file; the cut points may differ (a window's first record vs the first record behind the balance point)
        assert len(with_index) == len(host.partition(path, n_parts, 500))
