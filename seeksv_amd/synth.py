"""Synthetic WGS workloads: bench / test tooling around csrc/synth_core.h.

A Workload is a pure description (contigs, read count, planted structural variants); records are produced
on demand for any index range [g0, g0+n) either on the host (libseeksv_synth_cpu.so -> numpy arrays) or directly
in HBM (libseeksv_synth.so -> torch tensors), byte-identical by construction.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _abi

HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
        133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
HG38_NAMES = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"]
SEED = 0x5EE45F
MAX_CONTIGS = 64


class SyConfig(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_total", C.c_int64), ("n_contigs", C.c_int32), ("read_len", C.c_int32),
                ("contig_off", C.c_int64 * (MAX_CONTIGS + 1)), ("contig_len", C.c_int32 * MAX_CONTIGS), ("spacing_fp", C.c_uint64),
                ("clip_permille", C.c_int32), ("indel_permille", C.c_int32), ("dup_permille", C.c_int32), ("sec_permille", C.c_int32),
                ("improper_permille", C.c_int32), ("vaf_permille", C.c_int32), ("n_breakends", C.c_int32), ("qual_model", C.c_int32), ("unmap_permille", C.c_int32)]


BREAKEND_DTYPE = np.dtype([("lin", np.int64), ("tid", np.int32), ("q", np.int32), ("ptid", np.int32), ("ppos", np.int32),
                           ("side", np.int8), ("pdir", np.int8), ("mate_rev", np.int8), ("is_up", np.int8), ("mate_anchor", np.int32)])
assert BREAKEND_DTYPE.itemsize == 32

_synth_libs = {}


def _lib(gpu):
    name = "libseeksv_synth.so" if gpu else "libseeksv_synth_cpu.so"
    if name not in _synth_libs:
        path = os.path.join(_abi.LIBDIR, name)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make synth`")
        lib = C.CDLL(path)
        V = C.c_void_p
        lib.ssvs_plan.argtypes = [C.POINTER(SyConfig), V, C.c_int64, C.c_int64, V, V, V, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.ssvs_fill.argtypes = [C.POINTER(SyConfig), V, C.c_int64, C.c_int64] + [V] * 15
        lib.ssvs_last_error.restype = C.c_char_p
        if not gpu:
            lib.ssvs_fill_seq_all.argtypes = [C.POINTER(SyConfig), V, C.c_int64, C.c_int64, V]
        _synth_libs[name] = lib
    return _synth_libs[name]


class Workload:
    def __init__(self, genome_frac=1.0, depth=30.0, n_sv=10000, seed=SEED, read_len=150, clip_permille=10, indel_permille=20,
                 dup_permille=80, sec_permille=2, improper_permille=20, vaf_permille=500, n_contigs=24, min_contig=30000,
                 hbv=False, n_integrations=0, qual_model=0, unmap_permille=0):
        """hbv / n_integrations (BASELINE config 5, SURVEY 8d "C5"): a hybrid reference - the human contigs plus an `HBV` contig of 3,215 bp -
        and n_integrations planted human<->HBV junctions in all three orientations the junction model has, some of them within 200 bp of
        the HBV contig's ends (the unsigned-wrap flank windows of getsv.cpp:760-777).  A normal sample of the same patient is the same
        workload with n_integrations=0 (same seed: same reference, same germline SVs) at its own depth."""
        self.names = list(HG38_NAMES[:n_contigs])
        self.lens = np.array([max(int(l * genome_frac), min_contig) for l in HG38[:n_contigs]], dtype=np.int64)
        self.hbv_tid = -1
        if hbv or n_integrations:
            self.hbv_tid = n_contigs
            self.names.append("HBV")
            self.lens = np.concatenate([self.lens, [3215]]).astype(np.int64)
            n_contigs += 1
        self.n_integrations = n_integrations
        self.offs = np.concatenate([[0], np.cumsum(self.lens)])
        G = int(self.offs[-1])
        self.genome_len = G
        self.read_len = read_len
        self.n_total = int(depth * G / read_len)
        self.depth = depth
        cfg = SyConfig()
        cfg.seed = seed
        cfg.n_total = self.n_total
        cfg.n_contigs = n_contigs
        cfg.read_len = read_len
        for i in range(n_contigs):
            cfg.contig_off[i] = int(self.offs[i])
            cfg.contig_len[i] = int(self.lens[i])
        cfg.contig_off[n_contigs] = G
        cfg.spacing_fp = ((G - read_len - 16) << 20) // max(self.n_total, 1)
        cfg.clip_permille, cfg.indel_permille, cfg.dup_permille = clip_permille, indel_permille, dup_permille
        cfg.sec_permille, cfg.improper_permille, cfg.vaf_permille = sec_permille, improper_permille, vaf_permille
        cfg.qual_model = int(qual_model)   # 0: five binned quality values; 1: forty (HiSeq-like)
        cfg.unmap_permille = int(unmap_permille)   # pairs with one unmapped end (records 2k, 2k + 1): getclip's unmapped-pair side channel; host.write_bam gives both the name of 2k
        self.cfg = cfg
        self.max_ref_span = read_len + 8
        self._plant(n_sv, seed)
        cfg.n_breakends = len(self.breakends)
        self._be_dev = {}

    # ---- planted structural variants -> breakends (generator side) + junctions (getsv side) ----
    def _plant(self, n_sv, seed):
        rng = np.random.RandomState(seed & 0x7fffffff)
        occupied = set()
        L = self.read_len

        def free(tid, pos):
            if pos < 1500 or pos > self.lens[tid] - 1500:
                return False
            b = int(self.offs[tid] + pos) // 1000
            return not any((b + d) in occupied for d in (-1, 0, 1))

        def take(tid, pos):
            occupied.add(int(self.offs[tid] + pos) // 1000)

        be, junctions = [], []
        n_human = len(self.lens) - (1 if self.hbv_tid >= 0 else 0)
        weights = self.lens[:n_human] / self.lens[:n_human].sum()
        tries = 0
        while len(junctions) < n_sv and tries < 50 * max(n_sv, 1):
            tries += 1
            k = len(junctions) % 8
            ta = int(rng.choice(n_human, p=weights))
            A = int(rng.randint(1500, self.lens[ta] - 1500))
            if k < 4:      # DEL, log-uniform 300 bp .. 1 Mbp
                kind, tb = "DEL", ta
                B = A + int(math.exp(rng.uniform(math.log(300), math.log(min(1e6, self.lens[ta] / 4))))) + 1
            elif k < 6:    # INV, both orientations
                kind, tb = ("INVpm" if k == 4 else "INVmp"), ta
                B = A + int(rng.randint(1000, min(100000, self.lens[ta] // 4)))
            else:          # TRA
                kind = "TRA"
                tb = int(rng.choice(n_human, p=weights))
                if tb == ta:
                    continue
                B = int(rng.randint(1500, self.lens[tb] - 1500))
            if not (free(ta, A) and free(tb, B)) or (ta == tb and abs(A - B) < 300):
                continue
            take(ta, A), take(tb, B)
            la, lb = int(self.offs[ta]), int(self.offs[tb])
            if kind in ("DEL", "TRA"):     # (A,+) -> (B,+): aligned [..,A) | clip = ref[B-1..]   and   clip = ref[..A-1] | aligned [B-1,..)
                junctions.append((self.names[ta], A, "+", self.names[tb], B, "+"))
                be.append((la + A, ta, A, tb, B - 1, 0, 1, 1, 1, B - 1))
                be.append((lb + B - 1, tb, B - 1, ta, A - 1, 1, -1, 0, 0, 0))
            elif kind == "INVpm":          # (A,+) -> (B,-): both aligned parts end at their breakpoint, partner walked backwards
                junctions.append((self.names[ta], A, "+", self.names[tb], B, "-"))
                be.append((la + A, ta, A, tb, B - 1, 0, -1, 0, 1, B - 1))
                be.append((lb + B, tb, B, ta, A - 1, 0, -1, 0, 0, 0))
            else:                          # (A,-) -> (B,+): both aligned parts start at their breakpoint, partner walked forwards
                junctions.append((self.names[ta], A, "-", self.names[tb], B, "+"))
                be.append((la + A - 1, ta, A - 1, tb, B - 1, 1, 1, 1, 1, B - 1))
                be.append((lb + B - 1, tb, B - 1, ta, A - 1, 1, 1, 0, 0, 0))
        # virus integrations: human (A) <-> HBV (B), own random stream so that the human SVs above do not depend on their number.
        # HBV breakpoints are spread over the whole contig, the first ones close to its ends.
        if self.n_integrations:
            rng2 = np.random.RandomState((seed + 0x48425631) & 0x7fffffff)
            tb, lb, hl = self.hbv_tid, int(self.offs[self.hbv_tid]), int(self.lens[self.hbv_tid])
            made, tries = 0, 0
            used_b = set()
            while made < self.n_integrations and tries < 200 * self.n_integrations:
                tries += 1
                ta = int(rng2.choice(n_human, p=weights))
                A = int(rng2.randint(1500, self.lens[ta] - 1500))
                if made == 0:
                    B = int(rng2.randint(40, 190))             # flank windows wrap below the contig start
                elif made == 1:
                    B = int(rng2.randint(hl - 190, hl - 40))   # ... and run past its end
                else:
                    B = int(rng2.randint(200, hl - 200))
                if not free(ta, A) or any(abs(B - u) < 40 for u in used_b):
                    continue
                take(ta, A); used_b.add(B)
                la = int(self.offs[ta])
                kind = made % 4
                if kind == 0:      # human (A,+) -> HBV (B,+)
                    junctions.append((self.names[ta], A, "+", "HBV", B, "+"))
                    be.append((la + A, ta, A, tb, B - 1, 0, 1, 1, 1, B - 1))
                    be.append((lb + B - 1, tb, B - 1, ta, A - 1, 1, -1, 0, 0, 0))
                elif kind == 1:    # HBV (B,+) -> human (A,+): the viral side is the up end
                    junctions.append(("HBV", B, "+", self.names[ta], A, "+"))
                    be.append((lb + B, tb, B, ta, A - 1, 0, 1, 1, 1, A - 1))
                    be.append((la + A - 1, ta, A - 1, tb, B - 1, 1, -1, 0, 0, 0))
                elif kind == 2:    # human (A,+) -> HBV (B,-): virus inserted in reverse orientation
                    junctions.append((self.names[ta], A, "+", "HBV", B, "-"))
                    be.append((la + A, ta, A, tb, B - 1, 0, -1, 0, 1, B - 1))
                    be.append((lb + B, tb, B, ta, A - 1, 0, -1, 0, 0, 0))
                else:              # human (A,-) -> HBV (B,+)
                    junctions.append((self.names[ta], A, "-", "HBV", B, "+"))
                    be.append((la + A - 1, ta, A - 1, tb, B - 1, 1, 1, 1, 1, B - 1))
                    be.append((lb + B - 1, tb, B - 1, ta, A - 1, 1, 1, 0, 0, 0))
                made += 1
        arr = np.array(be, dtype=BREAKEND_DTYPE) if be else np.zeros(0, dtype=BREAKEND_DTYPE)
        self.breakends = np.sort(arr, order="lin")
        # multimap<Junction,...> order (getsv.h:187-225)
        self.junctions = sorted(junctions, key=lambda j: (j[0], j[3], j[2], j[5], j[1], j[4]))

    def reference_fasta(self):
        """The hash-generated reference genome as FASTA text (small workloads only; golden generation)."""
        lib = _lib(False)
        lib.ssvs_ref_bases.argtypes = [C.POINTER(SyConfig), C.c_int32, C.c_int64, C.c_int64, C.c_char_p]
        out = []
        for tid, (name, ln) in enumerate(zip(self.names, self.lens)):
            buf = C.create_string_buffer(int(ln))
            lib.ssvs_ref_bases(C.byref(self.cfg), tid, 0, int(ln), buf)
            seq = buf.raw.decode()
            out.append(f">{name}\n" + "\n".join(seq[i:i + 80] for i in range(0, len(seq), 80)) + "\n")
        return "".join(out)

    def write_fasta(self, path, threads=8):
        """the hash-generated reference as a FASTA file, 80 columns, generated in pieces by `threads` host threads (the same text as reference_fasta())"""
        from concurrent.futures import ThreadPoolExecutor
        lib = _lib(False)
        lib.ssvs_ref_bases.argtypes = [C.POINTER(SyConfig), C.c_int32, C.c_int64, C.c_int64, C.c_void_p]
        PIECE = 8_000_000   # a multiple of 80

        def piece(args):
            tid, start, n = args
            buf = np.empty(n, np.uint8)
            lib.ssvs_ref_bases(C.byref(self.cfg), tid, start, n, buf.ctypes.data)
            full = n // 80 * 80
            body = np.empty((full // 80, 81), np.uint8)
            body[:, :80] = buf[:full].reshape(-1, 80)
            body[:, 80] = 10
            return body.tobytes() + buf[full:].tobytes() + (b"\n" if n > full else b"")
        with open(path, "wb") as f, ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
            for tid, (name, ln) in enumerate(zip(self.names, self.lens)):
                f.write(f">{name}\n".encode())
                for chunk in ex.map(piece, [(tid, s, min(PIECE, int(ln) - s)) for s in range(0, int(ln), PIECE)]):
                    f.write(chunk)

    def reference_2bit(self, device=None):
        """The reference in the re-aligner's layout (ssv_realign_index): -> (words, target_off).  device=None: numpy uint64 array;
        device=k: torch int64 tensor resident on GPU k (SSV_MEM_DEVICE), generated there."""
        n_words = (self.genome_len + 31) // 32 + 1
        off = np.asarray(self.offs, np.int64)
        if device is None:
            lib = _lib(False)
            out = np.zeros(n_words, np.uint64)
            lib.ssvs_ref_2bit(C.byref(self.cfg), C.c_void_p(out.ctypes.data), C.c_int64(n_words))
            return out, off
        import torch
        lib = _lib(True)
        torch.cuda.set_device(device)
        out = torch.zeros(n_words, dtype=torch.int64, device=torch.device("cuda", device))
        if lib.ssvs_ref_2bit(C.byref(self.cfg), C.c_void_p(out.data_ptr()), C.c_int64(n_words)) != 0:
            raise RuntimeError("ssvs_ref_2bit: " + lib.ssvs_last_error().decode())
        return out, off

    # ---- record generation ----
    def generate_host(self, g0, n, with_rec=False, all_seq=False):
        """records [g0, g0+n) as a dict of numpy arrays (a SSV_MEM_HOST batch).  with_rec: also the records' 64-byte lines (ssv_record).
        all_seq: packed bases and qualities for EVERY record (what a BAM file holds: reference bases at the record's position with 0.2 %
        substitutions, qualities from {2, 11, 25, 37, 40}) instead of for the soft-clipped ones only - the writer of bench.py's file leg."""
        lib = _lib(False)
        be = self.breakends
        bep = be.ctypes.data if len(be) else None
        a = dict(n_cigar=np.zeros(n, np.uint16), cigar_off=np.zeros(n, np.uint32), seq_off=np.zeros(n, np.uint64))
        nct, sqb = C.c_int64(), C.c_int64()
        p = lambda x: x.ctypes.data if x.size else None
        lib.ssvs_plan(C.byref(self.cfg), bep, g0, n, p(a["n_cigar"]), p(a["cigar_off"]), p(a["seq_off"]), C.byref(nct), C.byref(sqb))
        for name, dt in (("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("mapq", np.uint8), ("l_qseq", np.int32),
                         ("mtid", np.int32), ("mpos", np.int32), ("isize", np.int32)):
            a[name] = np.zeros(n, dt)
        a["cigar"] = np.zeros(nct.value, np.uint32)
        a["seqqual"] = np.zeros(sqb.value, np.uint8)
        rec = np.zeros(n, _abi.RECORD_DTYPE) if with_rec else None
        a["cigar_ends"] = np.zeros(n, np.uint8)
        lib.ssvs_fill(C.byref(self.cfg), bep, g0, n, p(a["tid"]), p(a["pos"]), p(a["flag"]), p(a["mapq"]), p(a["n_cigar"]), p(a["l_qseq"]),
                      p(a["mtid"]), p(a["mpos"]), p(a["isize"]), p(a["cigar_off"]), p(a["cigar"]), p(a["seq_off"]), p(a["seqqual"]),
                      p(rec) if with_rec else None, p(a["cigar_ends"]))
        if with_rec:
            a["rec"] = rec
        if all_seq:
            entry = (self.read_len + 1) // 2 + self.read_len
            a["seqqual"] = np.empty(n * entry + 16, np.uint8)
            a["seq_off"] = (np.arange(n, dtype=np.uint64) * np.uint64(entry))
            lib.ssvs_fill_seq_all(C.byref(self.cfg), bep, C.c_int64(g0), C.c_int64(n), p(a["seqqual"]))
            a["seqqual_bytes"] = n * entry
        a["xc"] = None
        a["max_ref_span"] = self.max_ref_span
        return a

    def generate_device(self, g0, n, device, soa=True, persistent=False):
        """records [g0, g0+n) resident in HBM: -> (ssv_batch_t with device pointers, dict of torch tensors keeping them alive).
        The batch always carries the hot columns (tid, pos, n_cigar) and the records' 64-byte lines (ssv_record); soa=False leaves the cold
        structure-of-arrays columns out (bench.py: the kernels only read the lines).  persistent: the caller keeps the tensors alive and
        unchanged until the getclip pass has delivered its table (SSV_MEM_PERSISTENT: the table is cut straight out of the batch)."""
        import torch
        lib = _lib(True)
        torch.cuda.set_device(device)
        dev = torch.device("cuda", device)
        if device not in self._be_dev:
            be = self.breakends
            self._be_dev[device] = torch.from_numpy(be.view(np.uint8).copy()).to(dev) if len(be) else torch.zeros(32, dtype=torch.uint8, device=dev)
        bep = self._be_dev[device].data_ptr()
        t = dict(n_cigar=torch.empty(n, dtype=torch.int16, device=dev), cigar_off=torch.empty(n, dtype=torch.int32, device=dev),
                 seq_off=torch.empty(n, dtype=torch.int64, device=dev))
        nct, sqb = C.c_int64(), C.c_int64()
        torch.cuda.synchronize(dev)
        rc = lib.ssvs_plan(C.byref(self.cfg), bep, g0, n, t["n_cigar"].data_ptr(), t["cigar_off"].data_ptr(), t["seq_off"].data_ptr(), C.byref(nct), C.byref(sqb))
        if rc != 0:
            raise RuntimeError("ssvs_plan: " + lib.ssvs_last_error().decode())
        cold = (("flag", torch.int16), ("mapq", torch.uint8), ("l_qseq", torch.int32), ("mtid", torch.int32), ("mpos", torch.int32), ("isize", torch.int32))
        for name, dt in (("tid", torch.int32), ("pos", torch.int32)) + (cold if soa else ()):
            t[name] = torch.empty(n, dtype=dt, device=dev)
        t["cigar"] = torch.empty(max(nct.value, 4), dtype=torch.int32, device=dev)
        t["seqqual"] = torch.empty(sqb.value + 16, dtype=torch.uint8, device=dev)
        t["rec"] = torch.empty(max(n, 1) * 64, dtype=torch.uint8, device=dev)
        t["cigar_ends"] = torch.empty(n + 16, dtype=torch.uint8, device=dev)
        ptr = lambda k: t[k].data_ptr() if k in t else None
        rc = lib.ssvs_fill(C.byref(self.cfg), bep, g0, n, ptr("tid"), ptr("pos"), ptr("flag"), ptr("mapq"),
                           ptr("n_cigar"), ptr("l_qseq"), ptr("mtid"), ptr("mpos"), ptr("isize"),
                           ptr("cigar_off"), ptr("cigar"), ptr("seq_off"), ptr("seqqual"), ptr("rec"), ptr("cigar_ends"))
        if rc != 0:
            raise RuntimeError("ssvs_fill: " + lib.ssvs_last_error().decode())
        if not soa:  # the offsets live on in the lines
            del t["cigar_off"], t["seq_off"]
        arrays = {k: v.data_ptr() for k, v in t.items()}
        arrays["xc"] = None
        arrays["n_cigar_total"] = nct.value
        arrays["seqqual_bytes"] = sqb.value
        arrays["max_ref_span"] = self.max_ref_span
        # the tid column as runs (ssv_batch_t.tid_runs: a batcher sees the changes of contig while it writes the column)
        if n > 1:
            tt = t["tid"]
            ch = (torch.nonzero(tt[1:] != tt[:-1]).flatten() + 1).cpu().numpy()
            starts = np.concatenate(([0], ch)).astype(np.int64)
            runs = np.zeros(len(starts), _abi.TID_RUN_DTYPE)
            runs["first"], runs["tid"] = starts, tt[torch.from_numpy(starts).to(dev)].cpu().numpy()
            arrays["tid_runs"] = runs
        elif n == 1:
            runs = np.zeros(1, _abi.TID_RUN_DTYPE)
            runs["tid"] = int(t["tid"][0].item())
            arrays["tid_runs"] = runs
        b, _ = _abi.make_batch(arrays, mem=_abi.MEM_DEVICE | (_abi.MEM_PERSISTENT if persistent else 0), n=n)
        return b, t
