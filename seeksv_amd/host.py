"""Python face of libseeksv_host.so (BAM -> SoA batches, getsv bookkeeping) plus the text formats of
`seeksv getclip` (clip.gz rows / clip.fq.gz records).  Used by tests and bench.py; the C++ `seeksv`
CLI uses the same library directly.
"""
import ctypes as C

import numpy as np

from . import _abi

CIGAR_CHARS = "MIDNSHP=X"


class BamReader:
    """samopen()/samread() stand-in (reference: sam/sam.h:59,73) producing structure-of-arrays batches."""

    def __init__(self, path, readahead=False):
        self._lib = _abi.host_lib()
        h = C.c_void_p()
        if self._lib.ssvh_bam_open(path.encode(), C.byref(h)) != 0:
            raise IOError(self._lib.ssvh_last_error().decode())
        self._h = h
        if readahead:  # decode batch k+1 on a background thread while the caller uses batch k
            self._lib.ssvh_bam_set_readahead(h, 1)
        n = self._lib.ssvh_bam_n_targets(h)
        self.target_names = [self._lib.ssvh_bam_target_name(h, i).decode() for i in range(n)]
        self.target_lens = np.array([self._lib.ssvh_bam_target_len(h, i) for i in range(n)], dtype=np.int32)

    @property
    def handle(self):
        return self._h

    def set_range(self, start, end=None):
        """read_batch then yields the records that start in [start, end): (block file offset, offset inside the block) pairs from partition()"""
        ec, eu = ((1 << 64) - 1, 0) if end is None else end
        if self._lib.ssvh_bam_set_range(self._h, start[0], start[1], ec, eu) != 0:
            raise IOError(self._lib.ssvh_last_error().decode())

    def read_batch(self, max_records=1 << 20, keep_all_seq=False):
        """Next batch as a dict of owned numpy arrays (None at EOF)."""
        b = _abi.Batch()
        if self._lib.ssvh_bam_read_batch(self._h, max_records, int(keep_all_seq), C.byref(b)) != 0:
            raise IOError(self._lib.ssvh_last_error().decode())
        if b.n == 0:
            return None
        return _abi.batch_to_arrays(b)

    def unmapped(self):
        """(qname, seq, qual, is_read1) of the UNMAP|MUNMAP records of the last batch."""
        out = []
        q, s, u, r = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_int()
        for k in range(self._lib.ssvh_bam_unmapped_count(self._h)):
            self._lib.ssvh_bam_unmapped_get(self._h, k, C.byref(q), C.byref(s), C.byref(u), C.byref(r))
            out.append((q.value.decode(), s.value.decode(), u.value.decode(), bool(r.value)))
        return out

    def close(self):
        if self._h:
            self._lib.ssvh_bam_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Header:
    """Contig names + lengths without a BAM file (synthetic inputs); usable wherever a BamReader's header is."""

    def __init__(self, names, lens):
        self._lib = _abi.host_lib()
        self.target_names = list(names)
        self.target_lens = np.asarray(lens, dtype=np.int32)
        n = len(self.target_names)
        nb = (C.c_char_p * max(n, 1))(*[s.encode() for s in self.target_names])
        lb = (C.c_int32 * max(n, 1))(*[int(x) for x in self.target_lens])
        h = C.c_void_p()
        self._lib.ssvh_bam_from_header(nb, lb, n, C.byref(h))
        self._h = h

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            self._lib.ssvh_bam_close(self._h)
            self._h = None


def partition(path, n_parts, halo_bp=1000):
    """ssvh_bam_partition: list of dicts (scan / own / end = (block file offset, offset inside the block); end None = end of file)"""
    lib = _abi.host_lib()
    parts = (_abi.BamPart * n_parts)()
    if lib.ssvh_bam_partition(path.encode(), n_parts, halo_bp, parts) != 0:
        raise IOError(lib.ssvh_partition_last_error().decode())
    return [dict(scan=(p.scan_coff, p.scan_uoff), own=(p.own_coff, p.own_uoff), end=None if p.end_coff == (1 << 64) - 1 else (p.end_coff, p.end_uoff),
                 own_tid=p.own_tid, own_pos=p.own_pos, initial_last_tid=p.initial_last_tid, before_own_tid=p.before_own_tid, halo_records=p.halo_records) for p in parts]


def walk_back(path, at, n_back):
    """ssvh_bam_walk_back: -> ((block file offset, offset inside the block), records walked)"""
    lib = _abi.host_lib()
    co, uo, n = C.c_uint64(), C.c_uint32(), C.c_int64()
    ec, eu = ((1 << 64) - 1, 0) if at is None else at
    if lib.ssvh_bam_walk_back(path.encode(), ec, eu, n_back, C.byref(co), C.byref(uo), C.byref(n)) != 0:
        raise IOError(lib.ssvh_partition_last_error().decode())
    return (co.value, uo.value), n.value


def write_bam(path, names, lens, batches, qname_prefix="s"):
    """Tooling: SoA batches (dicts of numpy arrays) -> BAM file (parallel BGZF deflate in libseeksv_host)."""
    lib = _abi.host_lib()
    n = len(names)
    nb = (C.c_char_p * max(n, 1))(*[s.encode() for s in names])
    lb = (C.c_int32 * max(n, 1))(*[int(x) for x in lens])
    first = 0
    it = iter(batches)           # streamed: a batch is written (and may be freed) before the next one is asked for
    cur = next(it, None)
    if cur is None:
        assert lib.ssvh_bam_write_batch(path.encode(), nb, lb, n, None, qname_prefix.encode(), 0, 0, 1) == 0
        return
    i = 0
    while cur is not None:
        nxt = next(it, None)
        bb, keep = _abi.make_batch(cur)
        rc = lib.ssvh_bam_write_batch(path.encode(), nb, lb, n, C.byref(bb), qname_prefix.encode(), first, int(i > 0), int(nxt is None))
        if rc != 0:
            raise IOError(lib.ssvh_last_error().decode())
        first += bb.n
        cur, i = nxt, i + 1


def read_bam(path, batch_records=1 << 20):
    """Whole BAM as (target_names, target_lens, [batch dicts])."""
    with BamReader(path) as r:
        batches = []
        while True:
            b = r.read_batch(batch_records)
            if b is None:
                break
            batches.append(b)
        return r.target_names, r.target_lens, batches


class JunctionTable:
    """ctypes form of a junction list (built once, reusable across Plan objects)."""

    def __init__(self, junctions):
        n = len(junctions)
        self.n = n
        self.arr = (_abi.JunctionIn * max(n, 1))()
        self._keep = {}
        for i, (uc, up, us, dc, dp, ds) in enumerate(junctions):
            ucb = self._keep.setdefault(uc, uc.encode())
            dcb = self._keep.setdefault(dc, dc.encode())
            a = self.arr[i]
            a.up_chr, a.down_chr, a.up_pos, a.down_pos, a.up_strand, a.down_strand = ucb, dcb, up, dp, us.encode(), ds.encode()


class Plan:
    """Junction list -> device query tables and back (ssvh_plan_*): GetBreak / MergeOverlap /
    FindDiscordantReadPairs window arithmetic / main_depth lookup rules of the reference."""

    def __init__(self, bam, junctions, mean, sd, times=4, flank_length=200, extra_points=()):
        """junctions: list of (up_chr, up_pos, up_strand, down_chr, down_pos, down_strand) in Junction order, or a JunctionTable."""
        self._lib = _abi.host_lib()
        jt = junctions if isinstance(junctions, JunctionTable) else JunctionTable(junctions)
        self._jt = jt
        n, arr = jt.n, jt.arr
        ne = len(extra_points)
        echr = (C.c_char_p * max(ne, 1))(*[c.encode() for c, _ in extra_points])
        epos = (C.c_int32 * max(ne, 1))(*[p for _, p in extra_points])
        h = C.c_void_p()
        rc = self._lib.ssvh_plan_create(bam.handle, arr, n, echr, epos, ne, mean, sd, times, flank_length, C.byref(h))
        if rc != 0:
            raise RuntimeError("ssvh_plan_create failed")
        self._h = h
        self.n_junctions = n
        self.n_extra = ne
        self.junctions = self._table("junctions", _abi.JUNCTION_DTYPE)
        self.windows = self._table("windows", _abi.INTERVAL_DTYPE)
        self.ranges = self._table("ranges", _abi.INTERVAL_DTYPE)
        self.points = self._table("points", _abi.INTERVAL_DTYPE)

    def _table(self, what, dt):
        n = C.c_int64()
        ptr = getattr(self._lib, "ssvh_plan_" + what)(self._h, C.byref(n))
        if n.value == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_uint8 * (n.value * dt.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt, count=n.value).copy()

    def update_isize(self, mean, sd, times=4):
        """New insert-size statistics -> new junction windows (the depth tables do not depend on them)."""
        self._lib.ssvh_plan_update_isize(self._h, mean, sd, times)
        self.junctions = self._table("junctions", _abi.JUNCTION_DTYPE)

    def fold(self, counts, range_sum, point_depth, prev_counts=None):
        """-> dict(abnormal, up_depth, down_depth, flank[J,4], flank_len[J,4], extra_point_depth)"""
        J = self.n_junctions
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        range_sum = np.ascontiguousarray(range_sum, dtype=np.uint64)
        point_depth = np.ascontiguousarray(point_depth, dtype=np.int32)
        prev = np.ascontiguousarray(prev_counts if prev_counts is not None else np.zeros(J), dtype=np.int32)
        out = dict(abnormal=np.zeros(J, np.int32), up_depth=np.zeros(J, np.int32), down_depth=np.zeros(J, np.int32),
                   flank=np.zeros((J, 4), np.uint64), flank_len=np.zeros((J, 4), np.uint32),
                   extra_point_depth=np.zeros(self.n_extra, np.int32))
        p = lambda a: a.ctypes.data if a.size else None
        self._lib.ssvh_plan_fold(self._h, p(counts), p(prev), p(range_sum), p(point_depth), p(out["abnormal"]),
                                 p(out["up_depth"]), p(out["down_depth"]), p(out["flank"]), p(out["flank_len"]),
                                 p(out["extra_point_depth"]))
        return out

    def close(self):
        if self._h:
            self._lib.ssvh_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- cluster table -> text (DisplaySClipReadsAndClipFq / DisplayCigarVector, clip_reads.h:300-345,489-505) ----

def table_to_dict(t):
    """ctypes ssv_cluster_table / orc_cluster_table -> dict of numpy arrays (copies)."""
    n = t.n_clusters

    def arr(ptr, cnt, dt):
        if cnt == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(ptr, shape=(cnt,)).astype(dt, copy=True)

    packed = int(getattr(t, "seq_packed", 0))
    qbits = int(getattr(t, "qual_bits", 8)) if packed else 8
    d = dict(n_clusters=n, n_events=t.n_events, seq_packed=packed, qual_bits=qbits, qual_alphabet=bytes(getattr(t, "qual_alphabet", b"")))
    if int(getattr(t, "format", 0)) == 3:
        return _compact_to_dict(t, d)
    for name, dt in (("tid", np.int32), ("pos", np.int32), ("side", np.uint8), ("support", np.int32), ("left_len", np.int32),
                     ("right_len", np.int32), ("qual_missing", np.uint8), ("str_off", np.uint64), ("cigar_off", np.uint64), ("n_cigar", np.int32)):
        d[name] = arr(getattr(t, name), n, dt)
    if n:
        sb = int(d["str_off"][-1]) + block_bytes(int(d["left_len"][-1]), int(d["right_len"][-1]))
        cb = int(d["cigar_off"][-1]) + int(d["n_cigar"][-1])
    else:
        sb = cb = 0
    d["str"] = arr(t.str, sb, np.uint8)
    d["cigar"] = arr(t.cigar, cb, np.uint32)
    return d


def host_tid_runs(b, last_tid):
    """the changes of contig among the mapped-pair records of a batch given as arrays (clip_reads.h:415-438): [(record index, contig)],
    continuing from `last_tid`, the contig of the last such record before the batch"""
    keep = np.flatnonzero((np.asarray(b["flag"]) & 12) == 0)
    if keep.size == 0:
        return []
    t = np.asarray(b["tid"])[keep]
    prev = np.concatenate(([last_tid], t[:-1]))
    ch = np.flatnonzero(t != prev)
    return [(int(keep[k]), int(t[k])) for k in ch]


def concat_tables(tables):
    """cluster tables of consecutive getclip passes (dicts of table_to_dict, ASCII format) -> one"""
    assert all(int(t.get("format", 0)) == 0 and not t["seq_packed"] for t in tables), "passes are put together in the ASCII format"
    d = dict(n_clusters=sum(t["n_clusters"] for t in tables), n_events=sum(t["n_events"] for t in tables), seq_packed=0, qual_bits=8, qual_alphabet=tables[0]["qual_alphabet"])
    for name in ("tid", "pos", "side", "support", "left_len", "right_len", "qual_missing", "n_cigar", "str", "cigar"):
        d[name] = np.concatenate([t[name] for t in tables])
    so = co = 0
    offs, coffs = [], []
    for t in tables:
        offs.append(t["str_off"] + np.uint64(so)); coffs.append(t["cigar_off"] + np.uint64(co))
        so += len(t["str"]); co += len(t["cigar"])
    d["str_off"], d["cigar_off"] = np.concatenate(offs), np.concatenate(coffs)
    return d


def _compact_to_dict(t, d):
    """the compact table (ssv_clip_table_format 3, include/seeksv_hip.h): the columns that did not cross PCIe are rebuilt here with numpy - the
    same arithmetic as ssv_clip_table_expand, written independently (the tests hold the two against each other)"""
    import ctypes as C
    from . import _abi
    n = t.n_clusters
    bb, qb, qg = int(t.base_bits), d["qual_bits"], max(1, int(t.qual_group))
    d.update(format=3, base_bits=bb, qual_group=qg, len_bytes=int(t.len_bytes), support_bytes=int(t.support_bytes), ncig_bytes=int(t.ncig_bytes))

    def raw(ptr, nbytes, dt):
        if nbytes == 0 or not ptr:
            return np.zeros(0, dtype=dt)
        return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value), dtype=dt).copy()
    ln = raw(t.c_len, n * 2 * t.len_bytes, np.uint16 if t.len_bytes == 2 else np.uint32).reshape(n, 2).astype(np.int32)
    d["left_len"], d["right_len"] = np.ascontiguousarray(ln[:, 0]), np.ascontiguousarray(ln[:, 1])
    d["support"] = raw(t.c_support, n * t.support_bytes, np.uint16 if t.support_bytes == 2 else np.uint32).astype(np.int32)
    d["n_cigar"] = raw(t.c_ncig, n * t.ncig_bytes, np.uint8 if t.ncig_bytes == 1 else np.uint16).astype(np.int32)
    d["qual_missing"] = raw(t.c_flags, n, np.uint8) & 1
    d["pos"] = raw(t.pos, n * 4, np.int32)
    runs = raw(t.runs, t.n_runs * _abi.TABLE_RUN_DTYPE.itemsize, _abi.TABLE_RUN_DTYPE)
    d["runs"] = runs
    counts = np.diff(np.append(runs["first"], n)) if len(runs) else np.zeros(0, np.int64)
    assert (counts > 0).all() and int(counts.sum()) == n and (len(runs) == 0 or runs["first"][0] == 0)
    d["tid"] = np.repeat(runs["tid"], counts).astype(np.int32)
    d["side"] = np.repeat(runs["side"], counts).astype(np.uint8)
    nn = (d["left_len"] + d["right_len"]).astype(np.int64)
    blk = 4 * ((nn * bb + 31) // 32 + (((nn + qg - 1) // qg if qg > 1 else nn) * qb + 31) // 32)
    d["str_off"] = (np.cumsum(blk) - blk).astype(np.uint64)
    nc = d["n_cigar"].astype(np.int64)
    d["cigar_off"] = (np.cumsum(nc) - nc).astype(np.uint64)
    sb, cb = int(blk.sum()), int(nc.sum())
    assert sb == t.str_bytes and cb == t.cigar_ops
    d["str"] = raw(t.str, sb, np.uint8)
    d["cigar_bytes"] = int(t.cigar_bytes)
    d["cigar"] = raw(t.c_cigar, cb * t.cigar_bytes, np.uint16 if t.cigar_bytes == 2 else np.uint32).astype(np.uint32)  # (16 bits: length << 4 | code)
    d["base_exc"] = raw(t.base_exc, t.n_base_exc * 8, np.uint64)
    return d


NT16 = "=ACMGRSVTWYHKDBN"


def block_bytes(ll, lr):
    """ssv_table_block_bytes (include/seeksv_hip.h): a cluster's string block in the ASCII table"""
    return (2 * (ll + lr) + 3) & ~3


def cluster_strings(d, k):
    """(seq_left, qual_left, seq_right, qual_right, cigar_text) of cluster k."""
    o, ll, lr = int(d["str_off"][k]), int(d["left_len"][k]), int(d["right_len"][k])
    s = d["str"]
    if d.get("format") == 3:
        n, bb, w, qg = ll + lr, int(d["base_bits"]), int(d["qual_bits"]), int(d.get("qual_group", 1))
        nb, nq = 4 * ((n * bb + 31) // 32), 4 * ((((n + qg - 1) // qg if qg > 1 else n) * w + 31) // 32)
        alphabet = d.get("qual_alphabet", b"")
        radix = sum(1 for c in alphabet if c)  # a group's number: i_0 + radix i_1 + radix^2 i_2 (include/seeksv_hip.h)

        def field(buf, i, width):  # stream bits [i * width, +width); bit b = bit b % 8 of byte b / 8
            b = (i * width) >> 3
            word = int(buf[b]) | ((int(buf[b + 1]) << 8) if b + 1 < len(buf) else 0) | ((int(buf[b + 2]) << 16) if b + 2 < len(buf) else 0)  # (11 bits can lie in three bytes)
            return (word >> ((i * width) & 7)) & ((1 << width) - 1)
        bs, qs = s[o:o + nb], s[o + nb:o + nb + nq]
        seq = ["ACGT"[field(bs, i, 2)] if bb == 2 else NT16[field(bs, i, 4)] for i in range(n)]
        if bb == 2 and len(d["base_exc"]):
            e = d["base_exc"]
            for x in e[np.searchsorted(e >> np.uint64(28), np.uint64(k), "left"):np.searchsorted(e >> np.uint64(28), np.uint64(k), "right")]:
                seq[(int(x) >> 4) & 0xFFFFFF] = NT16[int(x) & 15]
        seq = "".join(seq)
        if qg > 1:
            q = "".join(chr(alphabet[(field(qs, i // qg, w) // radix ** (i % qg)) % radix]) for i in range(n))
        else:
            q = qs[:n].tobytes().decode("latin-1") if w == 8 else "".join(chr(alphabet[field(qs, i, w)]) for i in range(n))
        sl, sr, ql, qr = seq[:ll], seq[ll:], q[:ll], q[ll:]
    else:
        sl = s[o:o + ll].tobytes().decode("latin-1")
        ql = s[o + ll:o + 2 * ll].tobytes().decode("latin-1")
        sr = s[o + 2 * ll:o + 2 * ll + lr].tobytes().decode("latin-1")
        qr = s[o + 2 * ll + lr:o + 2 * ll + 2 * lr].tobytes().decode("latin-1")
    if d["qual_missing"][k]:
        ql = qr = "*"
    co, nc = int(d["cigar_off"][k]), int(d["n_cigar"][k])
    cig = "".join(f"{int(c) >> 4}{CIGAR_CHARS[int(c) & 15]}" for c in d["cigar"][co:co + nc] if (int(c) & 15) not in (4, 5))
    return sl, ql, sr, qr, cig


def format_clip_outputs(d, target_names):
    """-> (clip text, clip.fq text): decompressed contents of prefix.clip.gz and prefix.clip.fq.gz."""
    rows, fq = [], []
    for k in range(d["n_clusters"]):
        sl, ql, sr, qr, cig = cluster_strings(d, k)
        chrom = target_names[int(d["tid"][k])]
        side = chr(int(d["side"][k]))
        if side == "5":   # aligned = seq_right, clipped = seq_left (clip_reads.h:308-311,320-321)
            rows.append(f"{chrom}\t{int(d['pos'][k])}\t5\t{cig}\t{sr}\t{qr}\t{sl}\t{ql}\t{int(d['support'][k])}\n")
            fq.append(f"@{sl}\n{sl}\n+\n{ql}\n")
        else:             # aligned = seq_left, clipped = seq_right (clip_reads.h:329-332,339-340)
            rows.append(f"{chrom}\t{int(d['pos'][k])}\t3\t{cig}\t{sl}\t{ql}\t{sr}\t{qr}\t{int(d['support'][k])}\n")
            fq.append(f"@{sr}\n{sr}\n+\n{qr}\n")
    return "".join(rows), "".join(fq)
