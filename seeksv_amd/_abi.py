"""ctypes mirror of include/seeksv_hip.h and include/seeksv_host.h (struct layouts + library loading).

Plumbing only: every computation happens in libseeksv_hip.so (HIP kernels) or libseeksv_host.so
(BAM decoding, getsv bookkeeping).  There is no Python or CPU fallback for the GPU path: loading
libseeksv_hip.so or creating a context without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.environ.get("SSV_LIBDIR") or os.path.join(_HERE, "lib")  # SSV_LIBDIR: an alternative build of the libraries (`make asan`)

NO_SEQ = (1 << 64) - 1
MEM_HOST, MEM_DEVICE, MEM_PERSISTENT = 0, 1, 2
# ssv_record: one 64-byte line per record (the cold fields + the first five CIGAR operations)
RECORD_DTYPE = np.dtype([("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("mapq", np.uint8), ("xc", np.uint8), ("n_cigar", np.uint16), ("pad", np.uint16),
                         ("l_qseq", np.int32), ("mtid", np.int32), ("mpos", np.int32), ("isize", np.int32), ("cigar_off", np.uint32), ("cigar_head", np.uint32, (5,)),
                         ("seq_off", np.uint64)])
assert RECORD_DTYPE.itemsize == 64


class Batch(C.Structure):
    """ssv_batch_t"""
    _fields_ = [
        ("n", C.c_int64), ("mem", C.c_int32), ("max_ref_span", C.c_int32),
        ("tid", C.c_void_p), ("pos", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p),
        ("n_cigar", C.c_void_p), ("l_qseq", C.c_void_p), ("mtid", C.c_void_p), ("mpos", C.c_void_p),
        ("isize", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p), ("xc", C.c_void_p),
        ("seq_off", C.c_void_p), ("seqqual", C.c_void_p),
        ("n_cigar_total", C.c_int64), ("seqqual_bytes", C.c_int64),
        ("rec", C.c_void_p), ("cigar_ends", C.c_void_p),
        ("tid_runs", C.c_void_p), ("n_tid_runs", C.c_int64),
    ]

    def set_tid_runs(self, runs):
        """runs: TID_RUN_DTYPE array (host memory, kept alive with the struct) or None"""
        if runs is None or len(runs) == 0:
            self.tid_runs, self.n_tid_runs, self._runs = None, 0, None
        else:
            self._runs = np.ascontiguousarray(runs, dtype=TID_RUN_DTYPE)
            self.tid_runs, self.n_tid_runs = self._runs.ctypes.data, len(self._runs)

    def get_tid_runs(self):
        """-> TID_RUN_DTYPE array (a copy) or None"""
        if not self.tid_runs or self.n_tid_runs <= 0:
            return None
        return np.frombuffer((C.c_uint8 * (16 * self.n_tid_runs)).from_address(self.tid_runs), dtype=TID_RUN_DTYPE).copy()


TID_RUN_DTYPE = np.dtype([("first", np.int64), ("tid", np.int32), ("pad", np.int32)])   # ssv_tid_run


def runs_of_tid(tid):
    """the tid column (numpy array) as runs: ssv_batch_t.tid_runs"""
    tid = np.asarray(tid)
    if tid.size == 0:
        return np.zeros(0, TID_RUN_DTYPE)
    starts = np.concatenate(([0], np.flatnonzero(tid[1:] != tid[:-1]) + 1))
    r = np.zeros(len(starts), TID_RUN_DTYPE)
    r["first"], r["tid"] = starts, tid[starts]
    return r


def rebase_runs(runs, first, n):
    """the runs of records [first, first + n) of a batch whose runs are `runs`"""
    if runs is None or len(runs) == 0 or n <= 0:
        return None
    k0 = int(np.searchsorted(runs["first"], first, side="right")) - 1
    k1 = int(np.searchsorted(runs["first"], first + n, side="left"))
    r = runs[k0:k1].copy()
    r["first"] = np.maximum(r["first"] - first, 0)
    return r


BATCH_FIELDS = [  # (name, numpy dtype) in struct order
    ("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("mapq", np.uint8), ("n_cigar", np.uint16),
    ("l_qseq", np.int32), ("mtid", np.int32), ("mpos", np.int32), ("isize", np.int32),
    ("cigar_off", np.uint32), ("cigar", np.uint32), ("xc", np.uint8), ("seq_off", np.uint64), ("seqqual", np.uint8),
    ("cigar_ends", np.uint8),   # optional (after `rec` in the struct)
]


class ClipParams(C.Structure):
    _fields_ = [("match_rate", C.c_double), ("min_mapq", C.c_int32), ("save_low_quality", C.c_int32),
                ("use_ownership", C.c_int32), ("initial_last_tid", C.c_int32), ("own_lo_tid", C.c_int32), ("own_lo_pos", C.c_int32),
                ("own_hi_tid", C.c_int32), ("own_hi_pos", C.c_int32)]

    @classmethod
    def make(cls, match_rate=0.9, min_mapq=1, save_low_quality=False, own=None, initial_last_tid=0):
        """own = ((lo_tid, lo_pos), (hi_tid, hi_pos)) keeps only events with lo <= (tid, pos) < hi."""
        p = cls(match_rate, min_mapq, int(save_low_quality), 0, initial_last_tid, 0, 0, 0, 0)
        if own is not None:
            p.use_ownership = 1
            (p.own_lo_tid, p.own_lo_pos), (p.own_hi_tid, p.own_hi_pos) = own
        return p


class ClusterTable(C.Structure):
    """ssv_cluster_table (and the leading part of orc_cluster_table)"""
    _fields_ = [
        ("n_clusters", C.c_int64), ("n_events", C.c_int64),
        ("tid", C.POINTER(C.c_int32)), ("pos", C.POINTER(C.c_int32)), ("side", C.POINTER(C.c_uint8)),
        ("support", C.POINTER(C.c_int32)), ("left_len", C.POINTER(C.c_int32)), ("right_len", C.POINTER(C.c_int32)),
        ("qual_missing", C.POINTER(C.c_uint8)), ("str_off", C.POINTER(C.c_uint64)), ("str", C.POINTER(C.c_uint8)),
        ("cigar_off", C.POINTER(C.c_uint64)), ("n_cigar", C.POINTER(C.c_int32)), ("cigar", C.POINTER(C.c_uint32)),
    ]


class OrcClusterTable(C.Structure):
    _fields_ = ClusterTable._fields_ + [("str_bytes", C.c_int64), ("cigar_ops", C.c_int64)]


class HipClusterTable(C.Structure):
    """ssv_cluster_table as libseeksv_hip.so hands it out: the common part + the sequence format flag"""
    _fields_ = ClusterTable._fields_ + [("seq_packed", C.c_int32), ("qual_bits", C.c_int32), ("qual_alphabet", C.c_uint8 * 64),
                                        ("format", C.c_int32), ("base_bits", C.c_int32), ("len_bytes", C.c_int32), ("support_bytes", C.c_int32), ("ncig_bytes", C.c_int32),
                                        ("qual_group", C.c_int32), ("c_len", C.c_void_p), ("c_support", C.c_void_p), ("c_ncig", C.c_void_p), ("c_flags", C.c_void_p),
                                        ("runs", C.c_void_p), ("n_runs", C.c_int64), ("base_exc", C.c_void_p), ("n_base_exc", C.c_int64), ("str_bytes", C.c_uint64),
                                        ("cigar_ops", C.c_uint64), ("support_sum", C.c_int64), ("c_cigar", C.c_void_p), ("cigar_bytes", C.c_int32), ("pad4", C.c_int32)]


TABLE_RUN_DTYPE = np.dtype([("tid", "<i4"), ("side", "u1"), ("pad", "u1", (3,)), ("first", "<i8")])  # ssv_table_run


class Junction(C.Structure):
    """ssv_junction"""
    _fields_ = [
        ("up_tid", C.c_int32), ("down_tid", C.c_int32), ("up_pos", C.c_int32), ("down_pos", C.c_int32),
        ("beg", C.c_int32), ("end", C.c_int32), ("up_strand", C.c_uint8), ("down_strand", C.c_uint8), ("pad", C.c_uint8 * 2),
    ]


JUNCTION_DTYPE = np.dtype([
    ("up_tid", np.int32), ("down_tid", np.int32), ("up_pos", np.int32), ("down_pos", np.int32),
    ("beg", np.int32), ("end", np.int32), ("up_strand", np.uint8), ("down_strand", np.uint8), ("pad", np.uint8, (2,)),
])
INTERVAL_DTYPE = np.dtype([("tid", np.int32), ("beg", np.int32), ("end", np.int32)])
assert JUNCTION_DTYPE.itemsize == C.sizeof(Junction) == 28
assert INTERVAL_DTYPE.itemsize == 12


class GetsvParams(C.Structure):
    _fields_ = [
        ("junctions", C.c_void_p), ("n_junctions", C.c_int64),
        ("mean", C.c_int32), ("sd", C.c_int32), ("times", C.c_int32), ("disc_min_mapq", C.c_int32),
        ("windows", C.c_void_p), ("n_windows", C.c_int64),
        ("depth_min_mapq", C.c_int32), ("n_targets", C.c_int32), ("target_len", C.c_void_p),
    ]


class JunctionIn(C.Structure):
    """ssvh_junction_in"""
    _fields_ = [("up_chr", C.c_char_p), ("down_chr", C.c_char_p), ("up_pos", C.c_int32), ("down_pos", C.c_int32),
                ("up_strand", C.c_char), ("down_strand", C.c_char)]


def make_batch(arrays, mem=MEM_HOST, n=None):
    """Build an ssv_batch_t over numpy arrays (host) or raw device pointers (ints). Returns (Batch, keepalive)."""
    b = Batch()
    keep = []
    for name, dt in BATCH_FIELDS:
        a = arrays.get(name)
        if a is None:
            setattr(b, name, None)
            continue
        if isinstance(a, np.ndarray):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            setattr(b, name, a.ctypes.data if a.size else None)
        else:
            setattr(b, name, int(a))
    rec = arrays.get("rec")
    if isinstance(rec, np.ndarray):
        rec = np.ascontiguousarray(rec)
        assert rec.nbytes == 64 * (len(arrays["tid"]) if n is None else n)
        keep.append(rec)
        b.rec = rec.ctypes.data if rec.size else None
    else:
        b.rec = int(rec) if rec else None
    b.mem = mem
    b.max_ref_span = int(arrays.get("max_ref_span", 0))
    if n is None:
        n = len(arrays["tid"])
    b.n = int(n)
    b.n_cigar_total = int(arrays["n_cigar_total"]) if "n_cigar_total" in arrays else int(len(arrays["cigar"]))
    b.seqqual_bytes = int(arrays["seqqual_bytes"]) if "seqqual_bytes" in arrays else (int(len(arrays["seqqual"])) if arrays.get("seqqual") is not None else 0)
    runs = arrays.get("tid_runs")
    if runs is None and isinstance(arrays.get("tid"), np.ndarray) and not arrays.get("no_tid_runs"):
        runs = runs_of_tid(arrays["tid"][:b.n])     # a host batch: the batcher (here: this function) sees the column anyway
    b.set_tid_runs(runs if runs is not None and len(runs) <= 48 else None)
    return b, keep


def batch_to_arrays(b):
    """Copy a HOST ssv_batch_t (e.g. from the BAM reader) into owned numpy arrays."""
    assert b.mem == MEM_HOST
    n = b.n
    sizes = {"cigar": b.n_cigar_total, "seqqual": b.seqqual_bytes}
    out = {}
    for name, dt in BATCH_FIELDS:
        cnt = sizes.get(name, n)
        ptr = getattr(b, name)
        if not ptr and name == "cigar_ends":
            continue                    # an optional column that is not there stays out (zeros would say "no soft clips")
        if not ptr or cnt == 0:
            out[name] = np.zeros(cnt, dtype=dt)
            continue
        buf = (C.c_uint8 * (cnt * np.dtype(dt).itemsize)).from_address(ptr)
        out[name] = np.frombuffer(buf, dtype=dt, count=cnt).copy()
    out["max_ref_span"] = int(b.max_ref_span)
    return out


class BgzfBlock(C.Structure):
    """ssv_bgzf_block (include/seeksv_hip.h)"""
    _fields_ = [("c_off", C.c_uint64), ("c_len", C.c_uint32), ("u_len", C.c_uint32)]


class BamdecInfo(C.Structure):
    """ssv_bamdec_info (include/seeksv_hip.h)"""
    _fields_ = [("n_records", C.c_int64), ("inflated_bytes", C.c_uint64), ("tail_offset", C.c_uint64), ("repaired_blocks", C.c_uint32), ("last_tid", C.c_int32),
                ("unmapped_raw", C.c_void_p), ("unmapped_bytes", C.c_uint64), ("n_tid_runs", C.c_uint32), ("tid_run_index", C.c_void_p), ("tid_run_tid", C.c_void_p)]


_libs = {}


def _load(name):
    if name not in _libs:
        path = os.path.join(LIBDIR, name)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()' or `make`)")
        _libs[name] = C.CDLL(path, mode=C.RTLD_GLOBAL)
    return _libs[name]


class BamPart(C.Structure):
    """ssvh_bam_part"""
    _fields_ = [("scan_coff", C.c_uint64), ("scan_uoff", C.c_uint32), ("own_coff", C.c_uint64), ("own_uoff", C.c_uint32), ("end_coff", C.c_uint64), ("end_uoff", C.c_uint32),
                ("own_tid", C.c_int32), ("own_pos", C.c_int32), ("initial_last_tid", C.c_int32), ("before_own_tid", C.c_int32), ("halo_records", C.c_int64)]


def host_lib():
    lib = _load("libseeksv_host.so")
    if not getattr(lib, "_typed", False):
        lib.ssvh_last_error.restype = C.c_char_p
        lib.ssvh_bam_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.ssvh_bam_from_header.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]
        lib.ssvh_bam_close.argtypes = [C.c_void_p]
        lib.ssvh_bam_write_batch.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int32, C.POINTER(Batch), C.c_char_p, C.c_int64, C.c_int, C.c_int]
        lib.ssvh_bam_n_targets.argtypes = [C.c_void_p]
        lib.ssvh_bam_target_name.argtypes = [C.c_void_p, C.c_int32]
        lib.ssvh_bam_target_name.restype = C.c_char_p
        lib.ssvh_bam_target_len.argtypes = [C.c_void_p, C.c_int32]
        lib.ssvh_bam_target_len.restype = C.c_int32
        lib.ssvh_bam_read_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.POINTER(Batch)]
        lib.ssvh_bam_set_readahead.argtypes = [C.c_void_p, C.c_int]
        lib.ssvh_bam_raw_begin.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.ssvh_bam_read_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.POINTER(BgzfBlock), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_size_t)]
        lib.ssvh_raw_record_fastq.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
        lib.ssvh_raw_record_fastq.restype = C.c_size_t
        lib.ssvh_bam_unmapped_count.argtypes = [C.c_void_p]
        lib.ssvh_bam_unmapped_count.restype = C.c_int64
        lib.ssvh_bam_unmapped_get.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
        lib.ssvh_bam_partition.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(BamPart)]
        lib.ssvh_bam_walk_back.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int64)]
        lib.ssvh_bam_set_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32]
        lib.ssvh_partition_last_error.restype = C.c_char_p
        lib.ssvh_plan_create.argtypes = [C.c_void_p, C.POINTER(JunctionIn), C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int64,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        lib.ssvh_plan_destroy.argtypes = [C.c_void_p]
        lib.ssvh_plan_update_isize.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        for f in ("junctions", "windows", "ranges", "points"):
            fn = getattr(lib, "ssvh_plan_" + f)
            fn.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
            fn.restype = C.c_void_p
        lib.ssvh_plan_fold.argtypes = [C.c_void_p] + [C.c_void_p] * 10
        lib._typed = True
    return lib


REALIGN_HIT = [("tid", "<i4"), ("pos", "<i4"), ("q_beg", "<i4"), ("q_end", "<i4"), ("score", "<i4"), ("second", "<i4"), ("n_mismatch", "<i4"),
               ("reverse", "u1"), ("mapq", "u1"), ("pad", "u1", (2,))]  # ssv_realign_hit


def hip_lib():
    """libseeksv_hip.so - raises when it is not built; creating a context raises when there is no GPU."""
    lib = _load("libseeksv_hip.so")
    if not getattr(lib, "_typed", False):
        V = C.c_void_p
        lib.ssv_ctx_create.argtypes = [C.c_int, C.POINTER(V)]
        lib.ssv_ctx_destroy.argtypes = [V]
        lib.ssv_sync.argtypes = [V]
        lib.ssv_last_error.argtypes = [V]
        lib.ssv_last_error.restype = C.c_char_p
        lib.ssv_stream.argtypes = [V]
        lib.ssv_stream.restype = V
        lib.ssv_host_alloc.argtypes = [C.c_size_t, C.POINTER(V)]
        lib.ssv_host_free.argtypes = [V]
        lib.ssv_batch_prefetch.argtypes = [V, C.POINTER(Batch)]
        lib.ssv_batch_prefetch_drop.argtypes = [V]
        lib.ssv_clip_begin.argtypes = [V, C.POINTER(ClipParams)]
        lib.ssv_clip_scan.argtypes = [V, C.POINTER(Batch)]
        lib.ssv_clip_scan_range.argtypes = [V, C.POINTER(Batch), C.c_int64, C.c_int64]
        lib.ssv_batch_retain.argtypes = [V, C.POINTER(Batch), C.POINTER(Batch)]
        lib.ssv_batch_release.argtypes = [V, C.POINTER(Batch)]
        lib.ssv_clip_event_count.argtypes = [V, C.POINTER(C.c_int64)]
        lib.ssv_clip_cluster.argtypes = [V, C.POINTER(HipClusterTable)]
        lib.ssv_clip_table_format.argtypes = [V, C.c_int]
        lib.ssv_clip_table_expand.argtypes = [V, C.POINTER(HipClusterTable), C.c_int32]
        lib.ssv_table_block_bytes3.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        lib.ssv_table_block_bytes3.restype = C.c_uint64
        lib.ssv_table_block_bytes3g.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32]
        lib.ssv_table_block_bytes3g.restype = C.c_uint64
        lib.ssv_clip_cluster_async.argtypes = [V, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.ssv_clip_table_wait.argtypes = [V, C.POINTER(HipClusterTable)]
        lib.ssv_clip_table_wait_prev.argtypes = [V, C.POINTER(HipClusterTable)]
        lib.ssv_isize_begin.argtypes = [V, C.c_int32, C.c_int64]
        lib.ssv_isize_accumulate.argtypes = [V, C.POINTER(Batch), C.POINTER(C.c_int32)]
        lib.ssv_isize_finish.argtypes = [V, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.ssv_getsv_begin.argtypes = [V, C.POINTER(GetsvParams)]
        lib.ssv_getsv_scan.argtypes = [V, C.POINTER(Batch)]
        lib.ssv_getsv_finish.argtypes = [V, V, V, C.c_int64, V, V, C.c_int64, V, C.POINTER(C.c_int32)]
        lib.ssv_bamdec_begin.argtypes = [V, C.c_int32, C.c_uint64]
        lib.ssv_bamdec_staging.argtypes = [V, C.c_int, C.c_size_t, C.POINTER(V)]
        lib.ssv_bamdec_decode.argtypes = [V, V, C.c_size_t, C.POINTER(BgzfBlock), C.c_int64, C.c_int, C.POINTER(Batch)]
        lib.ssv_bamdec_prefetch.argtypes = [V, V, C.c_size_t]
        lib.ssv_bamdec_prefetch_drop.argtypes = [V]
        lib.ssv_bamdec_expect.argtypes = [V, C.c_uint64]
        lib.ssv_bamdec_verify_crc.argtypes = [V, C.c_int]
        lib.ssv_host_register.argtypes = [V, C.c_size_t]
        lib.ssv_host_unregister.argtypes = [V]
        lib.ssv_bamdec_target_lens.argtypes = [V, C.POINTER(C.c_int32)]
        lib.ssv_bamdec_last.argtypes = [V, C.POINTER(BamdecInfo)]
        lib.ssv_batch_to_host.argtypes = [V, C.POINTER(Batch), C.POINTER(Batch)]
        lib.ssv_realign_index.argtypes = [V, V, C.c_int32, C.c_int64, V, C.c_int32, C.POINTER(C.c_int64)]
        lib.ssv_realign_query.argtypes = [V, V, V, C.c_int64, V]
        lib.ssv_realign_free.argtypes = [V]
        lib.ssv_prof_enable.argtypes = [V, C.c_int]
        lib.ssv_prof_reset.argtypes = [V]
        lib.ssv_prof_get.argtypes = [V, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.ssv_prof_names.restype = C.c_char_p
        lib._typed = True
    return lib
