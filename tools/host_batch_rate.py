"""Records per second through the host-batch boundary: SoA host batches (what libseeksv_host's reader hands out) -> ssv_clip_scan /
ssv_getsv_scan, (a) pageable arrays handed over one by one, (b) page-locked arrays announced one ahead (ssv_batch_prefetch).  The batch
is the bench workload's (synthetic 30x), cut into 4 M-record host batches.  Prints one JSON line; `--trace-dir d` additionally reads the
rocprofv3 kernel / memory-copy traces of a run of this script and reports how much of the copy time ran under kernels.

    python tools/host_batch_rate.py [--frac 0.0625] [--batch 4194304]
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/h2d -- python3 tools/host_batch_rate.py --frac 0.03125
    python tools/host_batch_rate.py --trace-dir gpurun_out/h2d
"""
import argparse
import csv
import glob
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def overlap_report(d):
    def rows(pat):
        out = []
        for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
            out += list(csv.DictReader(open(f)))
        return out
    k = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows("*kernel_trace.csv"))
    m = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")) for r in rows("*memory_copy_trace.csv")]
    h2d = [(a, b) for a, b, dr in m if "HOST_TO_DEVICE" in dr.upper() or "H2D" in dr.upper()]
    merged = []
    for a, b in k:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    starts = np.array([x[0] for x in merged]); ends = np.array([x[1] for x in merged])
    tot = under = 0
    for a, b in h2d:
        tot += b - a
        i = max(0, np.searchsorted(ends, a) - 1)
        while i < len(merged) and starts[i] < b:
            under += max(0, min(b, ends[i]) - max(a, starts[i]))
            i += 1
    return {"h2d_copies": len(h2d), "h2d_ms": tot / 1e6, "h2d_ms_under_kernels": under / 1e6, "kernel_ms": float((ends - starts).sum()) / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frac", type=float, default=1 / 16)
    ap.add_argument("--batch", type=int, default=1 << 22)
    ap.add_argument("--trace-dir")
    a = ap.parse_args()
    if a.trace_dir:
        print(json.dumps(overlap_report(a.trace_dir)))
        return
    print(json.dumps(measure(a.frac, a.batch)))


def measure(frac, batch, ctx=None):
    """the three rates as a dict (bench.py's `host_batch_path` leg calls this with its own context)"""
    from seeksv_amd import _abi, synth
    from seeksv_amd.device import Context, PinnedArrays
    w = synth.Workload(genome_frac=frac, depth=30, n_sv=max(10, int(10000 * frac)))
    ctx = ctx or Context(0)
    db, keep = w.generate_device(0, w.n_total, 0, soa=True)
    hb = ctx.batch_to_host(db)     # dict of numpy arrays (pageable)
    del keep
    hb.pop("rec", None)
    n = w.n_total
    cuts = list(range(0, n, batch)) + [n]

    def cut(i0, i1):
        out = {}
        for k, v in hb.items():
            if not isinstance(v, np.ndarray):
                out[k] = v
            elif k == "cigar":
                c0, c1 = int(hb["cigar_off"][i0]), (int(hb["cigar_off"][i1]) if i1 < n else len(v))
                out[k] = v[c0:c1]
            elif k == "seqqual":
                continue
            else:
                out[k] = v[i0:i1]
        out["cigar_off"] = out["cigar_off"] - np.uint32(int(hb["cigar_off"][i0]))
        # the bases / qualities of the cut's clipped records (offsets grow with the record index)
        so = out["seq_off"]
        have = np.nonzero(so != NO_SEQ)[0]
        if len(have):
            s0 = int(so[have[0]])
            later = np.nonzero(hb["seq_off"][i1:] != NO_SEQ)[0]
            s1 = int(hb["seq_off"][i1 + later[0]]) if len(later) else len(hb["seqqual"])
            out["seqqual"] = hb["seqqual"][s0:s1]
            out["seq_off"] = np.where(so != NO_SEQ, so - np.uint64(s0), so)
            out["seqqual_bytes"] = s1 - s0
        else:
            out["seqqual"] = hb["seqqual"][:0]
            out["seqqual_bytes"] = 0
        out["n_cigar_total"] = len(out["cigar"])
        return out
    NO_SEQ = np.uint64(0xFFFFFFFFFFFFFFFF)
    parts = [cut(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
    per_record = sum(v.nbytes for p in parts for k, v in p.items() if isinstance(v, np.ndarray)) / n
    res = {"records": n, "batches": len(parts), "host_bytes_per_record": round(per_record, 1)}

    def run(batches, prefetch, scan):
        t0 = time.perf_counter()
        if prefetch:
            ctx.prefetch(batches[0])
        for k, b in enumerate(batches):
            if prefetch and k + 1 < len(batches):
                ctx.prefetch(batches[k + 1])
            scan(b)
        ctx.sync()
        return time.perf_counter() - t0

    plain = [_abi.make_batch(p) for p in parts]
    with PinnedArrays() as pin:
        pinned = [_abi.make_batch(pin.batch(p)) for p in parts]
        for name, bs, pf in (("pageable", plain, False), ("pinned", pinned, False), ("pinned_prefetch", pinned, True)):
            best = None
            for _ in range(3):
                ctx.clip_begin()
                t = run([b[0] for b in bs], pf, ctx.clip_scan)
                ev = ctx.clip_event_count()
                best = t if best is None else min(best, t)
            res["clip_scan_" + name] = {"s": round(best, 4), "records_per_s": round(n / best), "GB_per_s": round(n * per_record / best / 1e9, 2), "events": ev}
    return res


if __name__ == "__main__":
    main()
