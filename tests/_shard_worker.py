"""Worker of tests/test_shard_gloo.py: one rank of a range-partitioned run, CPU only (oracle backend, gloo all-gather)."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from seeksv_amd import host, shard, synth  # noqa: E402


def cluster_rows(d):
    """cluster table -> list of hashable rows (sortable by (tid, side, pos); creation order inside a key is kept by a stable sort)"""
    rows = []
    for k in range(d["n_clusters"]):
        sl, ql, sr, qr, cig = host.cluster_strings(d, k)
        rows.append((int(d["tid"][k]), int(d["side"][k]), int(d["pos"][k]), int(d["support"][k]), sl, ql, sr, qr, cig))
    return rows


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_path = sys.argv[1]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.Workload(genome_frac=1 / 4096, depth=60, n_sv=30)
    sp = shard.shard_plan(w, rank, world)
    scan = w.generate_host(sp["scan_lo_rec"], sp["own_hi_rec"] - sp["scan_lo_rec"])
    own = w.generate_host(sp["own_lo_rec"], sp["own_hi_rec"] - sp["own_lo_rec"])
    prefix = w.generate_host(0, min(w.n_total, 200000))
    d = O.getclip([scan], own=sp["own"], initial_last_tid=sp["initial_last_tid"])
    rc, n, mean, sd = O.isize_stats([prefix], 20, 100000)
    hdr = host.Header(w.names, w.lens)
    plan = host.Plan(hdr, w.junctions, mean, sd)
    counts = O.discordant([own], plan.junctions, mean, sd, 4, 20)
    rs, pd, _ = O.depth([own], plan.windows, plan.ranges, plan.points, 20)
    vec = shard.pack_results(counts, rs, pd, d["n_clusters"], d["n_events"], int(d["support"].sum()))
    stacked = shard.all_gather_vector(vec)
    merged = shard.merge_results(stacked, len(counts), len(rs), len(pd))
    rows = cluster_rows(d)
    gathered = [None] * world
    dist.all_gather_object(gathered, rows)
    if rank == 0:
        allrows = [r for part in gathered for r in part]
        json.dump(dict(counts=merged[0].tolist(), rs=[int(x) for x in merged[1]], pd=merged[2].tolist(), n_clusters=merged[3], n_events=merged[4],
                       support_sum=merged[5], rows=allrows, mean=mean, sd=sd, per_rank_events=[int(s[-2]) for s in stacked]), open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
