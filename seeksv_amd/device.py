"""Python face of libseeksv_hip.so (include/seeksv_hip.h).  Thin ctypes plumbing: every operation runs in
hand-written HIP kernels on the GPU.  There is no fallback - without the built library or without a GPU
the constructor raises.
"""
import ctypes as C

import numpy as np

from . import _abi
from .host import concat_tables, host_tid_runs, table_to_dict


class SeeksvError(RuntimeError):
    pass


class Context:
    """One GPU, one HIP stream (ssv_ctx)."""

    def __init__(self, device=0):
        self._lib = _abi.hip_lib()
        h = C.c_void_p()
        rc = self._lib.ssv_ctx_create(device, C.byref(h))
        if rc != 0:
            raise SeeksvError(f"ssv_ctx_create({device}) failed ({rc}): {self._lib.ssv_last_error(None).decode()}")
        self._h = h
        self.device = device

    def _check(self, rc, what):
        if rc != 0:
            raise SeeksvError(f"{what} failed ({rc}): {self._lib.ssv_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ssv_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self._lib.ssv_sync(self._h), "ssv_sync")

    def prefetch(self, batch):
        """announce a host batch (an _abi.Batch over arrays that stay alive - best page-locked, see PinnedArrays): its copy to HBM starts on
        the upload stream now; the next scan call must be given this batch"""
        self._check(self._lib.ssv_batch_prefetch(self._h, C.byref(batch)), "ssv_batch_prefetch")

    def prefetch_drop(self):
        """give up the announced batches (their host arrays may be reused when this returns)"""
        self._check(self._lib.ssv_batch_prefetch_drop(self._h), "ssv_batch_prefetch_drop")

    @staticmethod
    def _as_batch(b):
        if isinstance(b, _abi.Batch):
            return b, None
        return _abi.make_batch(b)

    # ---- device-side BGZF inflate + BAM decode ----
    def bamdec_target_lens(self, lens):
        """after ssv_bamdec_begin: the contig lengths sharpen the speculation of record starts (results do not depend on it)"""
        arr = (C.c_int32 * max(1, len(lens)))(*[int(x) for x in lens])
        self._check(self._lib.ssv_bamdec_target_lens(self._h, arr), "ssv_bamdec_target_lens")

    def bam_batches(self, reader, chunk_bytes=64 << 20, max_blocks=1 << 16, keep_all_seq=False, chunk_inflated=1 << 31, prefetch=True, verify_crc=False):
        """Generator over SSV_MEM_DEVICE batches of a whole BAM file (host.BamReader), decoded on the GPU: yields (Batch, info dict).
        The batch is valid until the next iteration.  prefetch: chunk k+1 is read into the second staging buffer and announced
        (ssv_bamdec_prefetch) before chunk k is decoded, so its bytes cross PCIe while chunk k's kernels run."""
        hl = reader._lib
        first = C.c_uint64()
        if hl.ssvh_bam_raw_begin(reader.handle, C.byref(first)) != 0:
            raise IOError(hl.ssvh_last_error().decode())
        self._check(self._lib.ssv_bamdec_begin(self._h, len(reader.target_names), first.value), "ssv_bamdec_begin")
        self.bamdec_target_lens(reader.target_lens)
        if verify_crc:   # every inflated block against the CRC32 in its BGZF trailer (off by default, like libbam 0.1.16)
            self._check(self._lib.ssv_bamdec_verify_crc(self._h, 1), "ssv_bamdec_verify_crc")
        stages, blocks = [None, None], [None, None]

        def read(k):
            """chunk k into staging buffer k & 1; its compressed bytes leave for the GPU at once (ssv_bamdec_prefetch)"""
            w = k & 1
            if stages[w] is None:
                stages[w] = C.c_void_p()
                self._check(self._lib.ssv_bamdec_staging(self._h, w, chunk_bytes, C.byref(stages[w])), "ssv_bamdec_staging")
                blocks[w] = (_abi.BgzfBlock * max_blocks)()
            nb, nbytes = C.c_int64(), C.c_size_t()
            if hl.ssvh_bam_read_blocks(reader.handle, stages[w], chunk_bytes, chunk_inflated, blocks[w], max_blocks, C.byref(nb), C.byref(nbytes)) != 0:
                raise IOError(hl.ssvh_last_error().decode())
            if nb.value and prefetch:
                self._check(self._lib.ssv_bamdec_prefetch(self._h, stages[w], nbytes.value), "ssv_bamdec_prefetch")
            return nb, nbytes

        ci, cur = 0, read(0)
        while True:
            nb, nbytes = cur
            if nb.value:
                cur = read(ci + 1)  # read (and announced) before this chunk's kernels start: its upload runs beside them
            b = _abi.Batch()
            self._check(self._lib.ssv_bamdec_decode(self._h, stages[ci & 1], nbytes.value, blocks[ci & 1], nb.value, int(keep_all_seq), C.byref(b)), "ssv_bamdec_decode")
            ci += 1
            if nb.value == 0:
                return
            info = _abi.BamdecInfo()
            self._check(self._lib.ssv_bamdec_last(self._h, C.byref(info)), "ssv_bamdec_last")
            d = dict(n_records=info.n_records, inflated_bytes=info.inflated_bytes, repaired_blocks=info.repaired_blocks, last_tid=info.last_tid, n_blocks=nb.value, compressed_bytes=nbytes.value)
            k = info.n_tid_runs
            d["tid_runs"] = list(zip(np.ctypeslib.as_array((C.c_uint32 * k).from_address(info.tid_run_index)).tolist(), np.ctypeslib.as_array((C.c_int32 * k).from_address(info.tid_run_tid)).tolist())) if k else []
            unm, off = [], 0
            q, sq, ql, r1 = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_int()
            while info.unmapped_bytes:
                nxt = hl.ssvh_raw_record_fastq(info.unmapped_raw, info.unmapped_bytes, off, C.byref(q), C.byref(sq), C.byref(ql), C.byref(r1))
                if nxt == 0:
                    break
                unm.append((q.value.decode(), sq.value.decode(), ql.value.decode(), bool(r1.value)))
                off = nxt
            d["unmapped"] = unm
            yield b, d

    def batch_retain(self, dev_batch):
        """a device batch (the decoder's: valid until the next decode) copied into device memory of its own; -> Batch (SSV_MEM_DEVICE |
        SSV_MEM_PERSISTENT), to be given back with batch_release"""
        out = _abi.Batch()
        self._check(self._lib.ssv_batch_retain(self._h, C.byref(dev_batch), C.byref(out)), "ssv_batch_retain")
        return out

    def batch_release(self, batch):
        self._check(self._lib.ssv_batch_release(self._h, C.byref(batch)), "ssv_batch_release")

    def batch_to_host(self, dev_batch):
        """Device batch -> dict of owned numpy arrays (tests)."""
        h = _abi.Batch()
        self._check(self._lib.ssv_batch_to_host(self._h, C.byref(dev_batch), C.byref(h)), "ssv_batch_to_host")
        return _abi.batch_to_arrays(h)

    # ---- getclip ----
    def clip_begin(self, match_rate=0.9, min_mapq=1, save_low_quality=False, own=None, initial_last_tid=0):
        p = _abi.ClipParams.make(match_rate, min_mapq, save_low_quality, own, initial_last_tid)
        self._check(self._lib.ssv_clip_begin(self._h, C.byref(p)), "ssv_clip_begin")

    def clip_table_format(self, packed):
        """0 / False: ASCII; 3: the compact table (2-bit bases, qualities as alphabet indices, contig / side as runs: a third of the bytes over
        PCIe); host.cluster_strings decodes both"""
        self._check(self._lib.ssv_clip_table_format(self._h, int(packed)), "ssv_clip_table_format")

    def clip_scan(self, batch):
        b, keep = self._as_batch(batch)
        self._check(self._lib.ssv_clip_scan(self._h, C.byref(b)), "ssv_clip_scan")

    def clip_event_count(self):
        n = C.c_int64()
        self._check(self._lib.ssv_clip_event_count(self._h, C.byref(n)), "ssv_clip_event_count")
        return n.value

    def clip_cluster(self, as_dict=True):
        t = _abi.HipClusterTable()
        self._check(self._lib.ssv_clip_cluster(self._h, C.byref(t)), "ssv_clip_cluster")
        return table_to_dict(t) if as_dict else t

    def clip_cluster_async(self):
        """Cluster and queue the table's copy to the host; -> (n_clusters, n_events).  Collect with clip_table_wait()."""
        nc, ne = C.c_int64(), C.c_int64()
        self._check(self._lib.ssv_clip_cluster_async(self._h, C.byref(nc), C.byref(ne)), "ssv_clip_cluster_async")
        return nc.value, ne.value

    def clip_table_wait(self, prev=False, as_dict=False):
        t = _abi.HipClusterTable()
        fn = self._lib.ssv_clip_table_wait_prev if prev else self._lib.ssv_clip_table_wait
        self._check(fn(self._h, C.byref(t)), "ssv_clip_table_wait")
        return table_to_dict(t) if as_dict else t

    def clip_table_expand(self, t, n_threads=0):
        """format 3: rebuild the columns that did not cross PCIe (ssv_clip_table_expand) - sets t's pointers"""
        self._check(self._lib.ssv_clip_table_expand(self._h, C.byref(t), int(n_threads)), "ssv_clip_table_expand")
        return t

    def clip_scan_range(self, batch, rec_begin, rec_end):
        b, keep = self._as_batch(batch)
        self._check(self._lib.ssv_clip_scan_range(self._h, C.byref(b), int(rec_begin), int(rec_end)), "ssv_clip_scan_range")

    def getclip(self, batches, match_rate=0.9, min_mapq=1, save_low_quality=False, own=None, initial_last_tid=0, tid_runs=None):
        """InputBamOutputReads' record loop over a list of batches -> cluster table dict.
        Contigs that come back (unsorted input): the reference flushes its maps at every change of contig among the mapped-pair records
        (clip_reads.h:423-438); a pass of the library bins by (contig, side, position), so the pass ends in front of a visit whose contig is
        not greater than every contig the pass has seen, and the tables of the passes follow each other (see ssv_clip_scan_range).
        tid_runs: per batch the list of (record index, contig) changes - computed here for batches given as arrays; a device batch without
        it is taken as coordinate sorted."""
        tables = []
        self.clip_begin(match_rate, min_mapq, save_low_quality, own, initial_last_tid)
        last_tid = pass_max = int(initial_last_tid)
        for bi, b in enumerate(batches):
            runs = tid_runs[bi] if tid_runs is not None else (host_tid_runs(b, last_tid) if isinstance(b, dict) and b.get("flag") is not None else [])
            n = len(b["tid"]) if isinstance(b, dict) else int(b.n)
            lo = 0
            for i, tid in runs:
                if tid <= pass_max and own is None:
                    self.clip_scan_range(b, lo, i)
                    tables.append(self.clip_cluster())
                    self.clip_begin(match_rate, min_mapq, save_low_quality, own, last_tid)
                    lo, pass_max = i, tid
                pass_max = max(pass_max, tid)
                last_tid = tid
            self.clip_scan_range(b, lo, n)
        tables.append(self.clip_cluster())
        return tables[0] if len(tables) == 1 else concat_tables(tables)

    # ---- getsv pass 1 ----
    def isize_stats(self, batches, min_mapq=20, max_pairs=5000000):
        """-> (rc, n_pairs, mean, sd); rc == 1 when no pair qualifies (the reference's return value)."""
        self._check(self._lib.ssv_isize_begin(self._h, min_mapq, max_pairs), "ssv_isize_begin")
        done = C.c_int32(0)
        for bt in batches:
            b, keep = self._as_batch(bt)
            self._check(self._lib.ssv_isize_accumulate(self._h, C.byref(b), C.byref(done)), "ssv_isize_accumulate")
            if done.value:
                break
        n, mean, sd = C.c_int64(), C.c_int32(0), C.c_int32(0)
        self._check(self._lib.ssv_isize_finish(self._h, C.byref(n), C.byref(mean), C.byref(sd)), "ssv_isize_finish")
        return (1 if n.value == 0 else 0), n.value, mean.value, sd.value

    # ---- getsv passes 2+3 ----
    def getsv_begin(self, junctions, windows, mean, sd, target_lens, times=4, disc_min_mapq=20, depth_min_mapq=20):
        j = np.ascontiguousarray(junctions, dtype=_abi.JUNCTION_DTYPE)
        w = np.ascontiguousarray(windows, dtype=_abi.INTERVAL_DTYPE)
        tl = np.ascontiguousarray(target_lens, dtype=np.int32)
        p = _abi.GetsvParams()
        p.junctions = j.ctypes.data if len(j) else None
        p.n_junctions = len(j)
        p.mean, p.sd, p.times, p.disc_min_mapq = mean, sd, times, disc_min_mapq
        p.windows = w.ctypes.data if len(w) else None
        p.n_windows = len(w)
        p.depth_min_mapq = depth_min_mapq
        p.n_targets = len(tl)
        p.target_len = tl.ctypes.data if len(tl) else None
        self._gs_nj = len(j)
        self._check(self._lib.ssv_getsv_begin(self._h, C.byref(p)), "ssv_getsv_begin")

    def getsv_scan(self, batch):
        b, keep = self._as_batch(batch)
        self._check(self._lib.ssv_getsv_scan(self._h, C.byref(b)), "ssv_getsv_scan")

    def getsv_finish(self, ranges, points):
        r = np.ascontiguousarray(ranges, dtype=_abi.INTERVAL_DTYPE)
        q = np.ascontiguousarray(points, dtype=_abi.INTERVAL_DTYPE)
        counts = np.zeros(self._gs_nj, dtype=np.int32)
        rs = np.zeros(len(r), dtype=np.uint64)
        pd = np.zeros(len(q), dtype=np.int32)
        mx = C.c_int32(0)
        ptr = lambda a: a.ctypes.data if a.size else None
        self._check(self._lib.ssv_getsv_finish(self._h, ptr(counts), ptr(r), len(r), ptr(rs), ptr(q), len(q), ptr(pd), C.byref(mx)), "ssv_getsv_finish")
        return counts, rs, pd, mx.value

    def discordant_and_depth(self, batches, plan, mean, sd, min_mapq, target_lens):
        """Backend protocol shared with the oracle in the tests (see tests/golden_util.py)."""
        self.getsv_begin(plan.junctions, plan.windows, mean, sd, target_lens, 4, min_mapq, min_mapq)
        for b in batches:
            self.getsv_scan(b)
        counts, rs, pd, _ = self.getsv_finish(plan.ranges, plan.points)
        return counts, rs, pd

    # ---- clipped-sequence re-aligner (stand-in for the pipeline's external `bwa mem` step) ----
    def realign_index(self, ref2bit, target_off, mem=_abi.MEM_HOST):
        """ref2bit: numpy uint64 array (host) or a device pointer (mem=MEM_DEVICE); target_off: first base of every contig + total"""
        off = np.ascontiguousarray(target_off, np.int64)
        ptr = ref2bit.ctypes.data if isinstance(ref2bit, np.ndarray) else int(ref2bit)
        dropped = C.c_int64()
        self._check(self._lib.ssv_realign_index(self._h, ptr, mem, int(off[-1]), off.ctypes.data, len(off) - 1, C.byref(dropped)), "ssv_realign_index")
        return dropped.value

    def realign(self, seqs):
        """list of str -> numpy structured array of ssv_realign_hit"""
        blob = "".join(seqs).encode()
        off = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.uint64)
        hits = np.zeros(len(seqs), dtype=np.dtype(_abi.REALIGN_HIT))
        if len(seqs):
            self._check(self._lib.ssv_realign_query(self._h, C.c_char_p(blob), off.ctypes.data, len(seqs), hits.ctypes.data), "ssv_realign_query")
        return hits

    # ---- measurement ----
    def prof_enable(self, mode=1):
        self._check(self._lib.ssv_prof_enable(self._h, mode), "ssv_prof_enable")

    def prof_reset(self):
        self._check(self._lib.ssv_prof_reset(self._h), "ssv_prof_reset")

    def prof_get(self, name):
        ms, n, u = C.c_double(), C.c_int64(), C.c_int64()
        self._check(self._lib.ssv_prof_get(self._h, name.encode(), C.byref(ms), C.byref(n), C.byref(u)), "ssv_prof_get")
        return dict(total_ms=ms.value, launches=n.value, units=u.value)

    def prof_all(self):
        return {n: self.prof_get(n) for n in self._lib.ssv_prof_names().decode().split("\n")}


class PinnedArrays:
    """numpy arrays in page-locked host memory (ssv_host_alloc): only out of such memory do the copies of a host batch run asynchronously"""

    def __init__(self):
        self._lib = _abi.hip_lib()
        self._ptrs = []

    def empty(self, n, dtype):
        dt = np.dtype(dtype)
        p = C.c_void_p()
        if self._lib.ssv_host_alloc(max(1, int(n) * dt.itemsize), C.byref(p)) != 0:
            raise SeeksvError("ssv_host_alloc failed")
        self._ptrs.append(p)
        return np.frombuffer((C.c_uint8 * (int(n) * dt.itemsize)).from_address(p.value), dtype=dt, count=int(n))

    def copy(self, a):
        a = np.ascontiguousarray(a)
        out = self.empty(a.size * a.dtype.itemsize, np.uint8).view(a.dtype) if a.dtype.fields else self.empty(a.size, a.dtype)
        out[...] = a.reshape(-1)
        return out

    def batch(self, arrays):
        """a host batch (dict of numpy arrays, as synth.Workload.generate_host gives) copied into page-locked memory"""
        return {k: (self.copy(v) if isinstance(v, np.ndarray) else v) for k, v in arrays.items()}

    def close(self):
        for p in self._ptrs:
            self._lib.ssv_host_free(p)
        self._ptrs = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
