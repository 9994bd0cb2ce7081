"""Range partitioning of a coordinate-sorted record stream over ranks and the single exchange step.

Every rank owns a contiguous slice of the records = a contiguous reference interval.
  * getclip: a rank scans its slice plus a halo of the records that start up to ~1 kb before it and keeps the clip
    events whose breakpoint position falls inside its interval (ssv_clip_params.own_*), so every (contig, side, pos)
    bin lives on exactly one rank with its reads in BAM order - no communication, cluster tables stay sharded.
  * discordant tally and depth are sums over reads: each rank scans exactly its own records for ALL junctions and
    windows, and the per-rank vectors are added after ONE all-gather (the only collective of the path).
  * insert-size statistics need the first max_pairs qualifying records of the file: every rank computes them from the
    same global prefix (a few million records), which costs less than a round trip.
"""
import numpy as np

HALO_MIN_BP = 1000  # the halo is at least this long; it must reach the longest reference span a clipped read can carry (a '3' event's
                    # breakpoint is its read's start + span): shard_plan takes it from the workload's max_ref_span and refuses less


def shard_plan(workload, rank, world):
    n = workload.n_total
    per = -(-n // world)
    per = (per + 15) // 16 * 16                    # keeps sub-batch pointers 16-byte aligned (the narrowest column is one byte per record)
    lo, hi = min(rank * per, n), min((rank + 1) * per, n)
    spacing = workload.cfg.spacing_fp / float(1 << 20)
    halo_bp = max(HALO_MIN_BP, 2 * int(getattr(workload, "max_ref_span", 0)) + 16)
    assert halo_bp > int(getattr(workload, "max_ref_span", 0)), "the halo must cover the longest reference span of a read"
    halo = int(np.ceil(halo_bp / max(spacing, 1e-9)))
    halo = (halo + 15) // 16 * 16
    scan_lo = max(0, lo - halo) if rank > 0 else 0

    def coord(g):
        r = workload.generate_host(g, 1)
        return int(r["tid"][0]), int(r["pos"][0])

    if rank == 0:
        own_lo, last_tid = (0, 0), 0
    else:
        t, p = coord(lo)
        own_lo = (t, p)   # the smallest key a record starting at p can give: start + reference span, 0 for a lone soft clip
        last_tid = coord(scan_lo - 1)[0] if scan_lo > 0 else 0
    if rank == world - 1 or hi >= n:
        own_hi = (int(workload.cfg.n_contigs), 0)
    else:
        t, p = coord(hi)
        own_hi = (t, p)
    return dict(own_lo_rec=lo, own_hi_rec=hi, scan_lo_rec=scan_lo, own=(own_lo, own_hi), initial_last_tid=last_tid)


def pack_results(counts, range_sum, point_depth, n_clusters, n_events, support_sum):
    return np.concatenate([np.asarray(counts, np.int64), np.asarray(range_sum, np.uint64).astype(np.int64), np.asarray(point_depth, np.int64),
                           np.array([n_clusters, n_events, support_sum], np.int64)])


def merge_results(stacked, n_junctions, n_ranges, n_points):
    """stacked: [world, len] int64 -> summed (counts, range_sum, point_depth, n_clusters, n_events, support_sum)"""
    s = np.asarray(stacked, np.int64).sum(axis=0)
    a, b, c = n_junctions, n_junctions + n_ranges, n_junctions + n_ranges + n_points
    return s[:a].astype(np.int32), s[a:b].astype(np.uint64), s[b:c].astype(np.int32), int(s[c]), int(s[c + 1]), int(s[c + 2])


def all_gather_vector(vec, device=None):
    """The one collective: all-gather of the per-rank result vector (RCCL on GPU ranks, gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(vec, np.int64)[None, :]
    t = torch.from_numpy(np.ascontiguousarray(vec, np.int64))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()
