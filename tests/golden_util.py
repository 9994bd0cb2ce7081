"""Helpers shared by the golden-fixture tests: parse the reference's outputs, run a getsv BAM-pass
case through (plan -> backend -> fold) where the backend is the oracle or the HIP library."""
import os

import numpy as np

from seeksv_amd import host

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_text(*parts):
    with open(os.path.join(GOLDEN, *parts)) as f:
        return f.read()


def read_junction_file(path):
    """-B file rows -> list of (up_chr, up_pos, up_strand, down_chr, down_pos, down_strand, prev_abnormal) in multimap order."""
    rows = []
    with open(path) as f:
        for line in f:
            if line.startswith("@") or not line.strip():
                continue
            t = line.rstrip("\n").split("\t")
            rows.append((t[0], int(t[1]), t[2], t[4], int(t[5]), t[6], int(t[9])))
    # multimap<Junction,...>: ordered by Junction::operator< (getsv.h:187-225), equal keys in insertion order
    rows_sorted = sorted(rows, key=lambda j: (j[0], j[3], j[2], j[5], j[1], j[4]))
    return rows_sorted


def parse_sv_outputs(sv_path, stdout_path):
    """-> dict keyed by junction tuple -> list of dict(abnormal, updepth, downdepth, flank or None) in output order."""
    out = {}

    def add(key, rec):
        out.setdefault(key, []).append(rec)

    with open(sv_path) as f:
        for line in f:
            if line.startswith("@"):
                continue
            t = line.rstrip("\n").split("\t")
            key = (t[0], int(t[1]), t[2], t[4], int(t[5]), t[6])
            add(key, dict(abnormal=int(t[9]), updepth=int(t[11]), downdepth=int(t[12]), flank=tuple(int(x) for x in t[13:17]), where="sv"))
    with open(stdout_path) as f:
        for line in f:
            t = line.rstrip("\n").split("\t")
            if len(t) < 16:
                continue
            key = (t[1], int(t[2]), t[3], t[5], int(t[6]), t[7])
            add(key, dict(abnormal=int(t[10]), updepth=int(t[12]), downdepth=int(t[13]), flank=None, where=t[0]))
    return out


def run_getsv_case(bam_path, junction_rows, backend, min_mapq=20, flank_length=200, n_pairs=5000000, batch_records=1 << 20):
    """backend: object with isize_stats(batches, min_mapq, max_pairs) -> (rc, n, mean, sd),
    discordant_and_depth(batches, plan, mean, sd, min_mapq, target_lens) -> (counts, range_sum, point_depth)."""
    with host.BamReader(bam_path) as bam:
        batches = []
        while True:
            b = bam.read_batch(batch_records)
            if b is None:
                break
            batches.append(b)
        rc, n, mean, sd = backend.isize_stats(batches, min_mapq, n_pairs)
        if rc != 0:
            mean, sd = 0, 0  # CallGetsv's initial values stay (seeksv.cpp:243)
        junctions = [j[:6] for j in junction_rows]
        prev = np.array([j[6] for j in junction_rows], dtype=np.int32)
        plan = host.Plan(bam, junctions, mean, sd, 4, flank_length)
        counts, range_sum, point_depth = backend.discordant_and_depth(batches, plan, mean, sd, min_mapq, bam.target_lens)
        folded = plan.fold(counts, range_sum, point_depth, prev)
        plan.close()
    return (rc, n, mean, sd), junctions, folded


def check_getsv_against_golden(junctions, folded, golden):
    """Compare per-junction results with the reference's tables.  Returns number of values compared."""
    seen = {}
    checked = 0
    for idx, key in enumerate(junctions):
        k = seen.get(key, 0)
        seen[key] = k + 1
        g = golden[key][k]
        assert int(folded["abnormal"][idx]) == g["abnormal"], (key, "abnormal", int(folded["abnormal"][idx]), g)
        assert int(folded["up_depth"][idx]) == g["updepth"], (key, "updepth", int(folded["up_depth"][idx]), g)
        assert int(folded["down_depth"][idx]) == g["downdepth"], (key, "downdepth", int(folded["down_depth"][idx]), g)
        checked += 3
        if g["flank"] is not None:
            for c in range(4):
                ln = int(folded["flank_len"][idx, c])
                assert ln != 0, (key, "zero-length flank window would divide by zero in the reference")
                avg = (int(folded["flank"][idx, c]) // ln) & 0xFFFFFFFF  # unsigned int = unsigned long / unsigned int (getsv.cpp:946)
                assert avg == g["flank"][c], (key, "flank", c, avg, g)
                checked += 1
    return checked


# ---- synthetic workloads whose reference outputs are committed under golden/synth (inputs are regenerated) ----

SYNTH_CASES = {"synth30x": dict(genome_frac=1 / 2048, depth=30, n_sv=40), "synth300x": dict(genome_frac=1 / 8192, depth=300, n_sv=24),
               "synthhbv": dict(genome_frac=1 / 8192, depth=60, n_sv=8, n_integrations=10)}  # BASELINE config 5 in small (human + HBV)


def read_gz(*parts):
    import gzip
    with gzip.open(os.path.join(GOLDEN, *parts), "rt") as f:
        return f.read()


def run_getsv_batches(header, batches, junction_rows, backend, min_mapq=20, flank_length=200, n_pairs=5000000):
    """Like run_getsv_case but for batches that are already in memory (synthetic inputs)."""
    rc, n, mean, sd = backend.isize_stats(batches, min_mapq, n_pairs)
    if rc != 0:
        mean, sd = 0, 0
    junctions = [j[:6] for j in junction_rows]
    prev = np.array([j[6] for j in junction_rows], dtype=np.int32)
    plan = host.Plan(header, junctions, mean, sd, 4, flank_length)
    counts, range_sum, point_depth = backend.discordant_and_depth(batches, plan, mean, sd, min_mapq, header.target_lens)
    folded = plan.fold(counts, range_sum, point_depth, prev)
    plan.close()
    return (rc, n, mean, sd), junctions, folded
