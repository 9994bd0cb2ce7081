#!/usr/bin/env python3
"""BASELINE config 3 at its full size on one MI355X: synthetic 300x tumor WGS (6.2 G records - more than fits in HBM at once), 10,000 planted
DEL / INV / TRA.  The records are generated in HBM chunk by chunk and streamed through the path twice (getclip + insert size, then the
fused getsv pass) under two different chunkings; the results must not depend on where the batches were cut, every clip event must be
in exactly one cluster, and every planted junction must be seen by the discordant tally and the depth pass.
usage: python tools/config3_full_size.py [genome_frac] [depth] [n_sv] [chunks_a] [chunks_b]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(w, hdr, ctx, n_chunks, probe=None):
    from seeksv_amd import host
    import torch
    n = w.n_total
    per = (-(-n // n_chunks) + 7) // 8 * 8
    cuts = [(g, min(per, n - g)) for g in range(0, n, per)]
    t = {"generate": 0.0, "getclip": 0.0, "cluster": 0.0, "isize": 0.0, "getsv": 0.0}
    per_chunk = {"generate": [], "getclip_scan": [], "getsv_scan": []}   # seconds, chunk by chunk (round 6: where a chunk's time goes)
    ctx.prof_reset(); ctx.prof_enable(1)   # HIP events around every kernel group and the table's copy

    def chunk(g, m):
        t0 = time.perf_counter()
        b, keep = w.generate_device(g, m, 0)
        torch.cuda.synchronize()
        t["generate"] += time.perf_counter() - t0
        per_chunk["generate"].append(round(time.perf_counter() - t0, 4))
        return b, keep

    ctx.clip_table_format(3)
    ctx.clip_begin(0.9, 1, False, None, 0)
    stats = None
    for k, (g, m) in enumerate(cuts):
        b, keep = chunk(g, m)
        t0 = time.perf_counter()
        ctx.clip_scan(b)
        ctx.sync()
        t["getclip"] += time.perf_counter() - t0
        per_chunk["getclip_scan"].append(round(time.perf_counter() - t0, 4))
        if k == 0:
            t0 = time.perf_counter()
            pb, pkeep = w.generate_device(0, min(n, 6_500_000), 0)
            stats = ctx.isize_stats([pb], 20, 5000000)
            t["isize"] += time.perf_counter() - t0
            del pb, pkeep
        del b, keep
    t0 = time.perf_counter()
    d = ctx.clip_cluster()
    t["cluster"] += time.perf_counter() - t0
    h = hashlib.sha256()
    for k in ("tid", "pos", "side", "support", "left_len", "right_len", "n_cigar", "cigar", "str"):
        h.update(np.ascontiguousarray(d[k]).tobytes())
    table = dict(n_clusters=int(d["n_clusters"]), n_events=int(d["n_events"]), support_sum=int(d["support"].sum()), max_support=int(d["support"].max()),
                 qual_bits=int(d["qual_bits"]), table_bytes=int(sum(np.asarray(d[k]).nbytes for k in d if isinstance(d[k], np.ndarray))), sha256=h.hexdigest())
    if probe is not None:
        probe(d)   # (tests: rows of the table against the CPU oracle on slices of the input)
    del d
    plan = host.Plan(hdr, w.junctions, stats[2], stats[3])
    t0 = time.perf_counter()
    ctx.getsv_begin(plan.junctions, plan.windows, stats[2], stats[3], hdr.target_lens, 4, 20, 20)
    t["getsv"] += time.perf_counter() - t0
    for g, m in cuts:
        b, keep = chunk(g, m)
        t0 = time.perf_counter()
        ctx.getsv_scan(b)
        ctx.sync()
        t["getsv"] += time.perf_counter() - t0
        per_chunk["getsv_scan"].append(round(time.perf_counter() - t0, 4))
        del b, keep
    t0 = time.perf_counter()
    counts, rs, pd, max_depth = ctx.getsv_finish(plan.ranges, plan.points)
    t["getsv"] += time.perf_counter() - t0
    folded = plan.fold(counts, rs, pd)
    plan.close()
    prof = {k: round(v["total_ms"], 3) for k, v in ctx.prof_all().items() if v["launches"]}
    ctx.prof_enable(0)
    per_chunk["generate"] = per_chunk["generate"][:len(cuts)] + ["second pass:"] + per_chunk["generate"][len(cuts):]
    return dict(chunks=len(cuts), per_chunk_s=per_chunk, device_ms=prof, table=table, mean=stats[2], sd=stats[3], counts=np.asarray(counts), rs=np.asarray(rs), pd=np.asarray(pd), max_depth=int(max_depth),
                abnormal=np.asarray(folded["abnormal"]), up_depth=np.asarray(folded["up_depth"]), down_depth=np.asarray(folded["down_depth"]),
                seconds={k: round(v, 3) for k, v in t.items()})


def main(genome_frac=1.0, depth=300.0, n_sv=10000, chunks_a=10, chunks_b=16, unmap_permille=0, probe=None):
    from seeksv_amd import host, synth
    from seeksv_amd.device import Context
    w = synth.Workload(genome_frac=genome_frac, depth=depth, n_sv=n_sv, unmap_permille=unmap_permille)  # (unmap_permille: that share of the records in pairs with one unmapped end)
    hdr = host.Header(w.names, w.lens)
    out = {"records": w.n_total, "junctions": len(w.junctions), "depth": depth, "genome_frac": genome_frac, "unmap_permille": unmap_permille}
    with Context(0) as ctx:
        a = run(w, hdr, ctx, chunks_a, (lambda d: probe(w, d)) if probe else None)
        b = run(w, hdr, ctx, chunks_b)
    hdr.close()
    assert a["table"] == b["table"], (a["table"], b["table"])
    for k in ("counts", "rs", "pd", "abnormal", "up_depth", "down_depth"):
        assert np.array_equal(a[k], b[k]), k
    assert (a["mean"], a["sd"], a["max_depth"]) == (b["mean"], b["sd"], b["max_depth"])
    assert a["table"]["support_sum"] == a["table"]["n_events"], "every clip event is in exactly one cluster"
    seen = int((a["abnormal"] > 0).sum())
    covered = int(((a["up_depth"] > 0) & (a["down_depth"] > 0)).sum())
    out.update(chunkings=[a["chunks"], b["chunks"]], results_identical=True, table=a["table"], mean=a["mean"], sd=a["sd"], max_depth=a["max_depth"],
               junctions_with_discordant_pairs=seen, junctions_with_depth_at_both_ends=covered, discordant_pairs=int(a["abnormal"].sum()),
               median_breakpoint_depth=float(np.median(np.concatenate([a["up_depth"], a["down_depth"]]))), seconds=[a["seconds"], b["seconds"]],
               per_chunk_s=[a["per_chunk_s"], b["per_chunk_s"]], device_ms=[a["device_ms"], b["device_ms"]],
               note="seconds: wall clock per phase, summed over the chunks (generate = the synthetic records made in HBM, not part of the path; cluster = sort + bins + pack kernels AND the table's copy to the host); "
                    "device_ms: HIP events around every kernel group of the run (table_d2h = the copy); per_chunk_s: wall clock chunk by chunk")
    assert seen == len(w.junctions) and covered == len(w.junctions), (seen, covered, len(w.junctions))
    return out


if __name__ == "__main__":
    args = [float(x) for x in sys.argv[1:]]
    kw = {}
    for name, v in zip(("genome_frac", "depth", "n_sv", "chunks_a", "chunks_b", "unmap_permille"), args):
        kw[name] = v if name in ("genome_frac", "depth") else int(v)
    print(json.dumps(main(**kw)))
