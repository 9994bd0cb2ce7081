#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes of bench.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; the TCC block cannot hold both)
into profiles/<tag>_pmc_fetch_write.csv (per kernel) and profiles/traffic.json (HBM bytes per step: per kernel group and for the whole path).

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <records> <steps> [tag]
  each dir holds <something>_counter_collection.csv as written by
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap
  (steps = the steps that run of bench.py executed: setup 4 - warmup, timed, breakdown = 4 + 1 + 3 = 8 with --warmup 0 --steps 1: the dispatch count of k_getsv_scan in the trace)
FETCH_SIZE / WRITE_SIZE are reported in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE shows exactly half the bytes of a
wide coalesced streaming read (16 B/lane), so it is doubled for the two streaming scans, whose loads are all of that kind; other
kernels' loads are scattered 16-byte quarters of 64-byte lines or narrower and are taken as reported (uncalibrated)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel -> group of bench.py (ssv_prof_*); the device-wide scans and fills belong to whatever named kernel ran before them
GROUP_OF = {
    "k_clip_scan": "clip_scan", "k_clip_scan_ends": "clip_scan", "k_cand_place": "clip_place", "k_clip_filter": "clip_place", "k_clip_place": "clip_place", "k_event_max": "clip_place", "k_last_tid": "clip_place",
    "k_gather_sizes": "clip_gather", "k_clip_gather": "clip_gather",
    "k_check_sorted": "event_sort", "k_check_sorted_pairs": "event_sort", "k_window_rank_sort": "event_sort", "k_key_max": "event_sort", "k_qual_sample": "event_sort", "k_rs_hist": "event_sort", "k_rs_scatter": "event_sort", "k_side_bounds": "event_sort",
    "k_merge_sides": "event_sort", "k_concat_sides": "event_sort", "k_gather_lines": "event_sort", "k_iota": "event_sort",
    "k_bin_mark": "cluster_bins", "k_multi_list": "cluster_bins", "k_bin_start_flags": "cluster_bins", "k_bin_start_list": "cluster_bins", "k_cluster_bins": "cluster_bins", "k_cluster_bins4": "cluster_bins", "k_bins4_tables": "cluster_bins",
    "k_cluster_meta": "cluster_pack", "k_cluster_cols": "cluster_pack", "k_cluster_pack_ascii": "cluster_pack",
    "k_cluster_cols3": "cluster_pack", "k_cluster_tile_sums": "cluster_pack", "k_cluster_cols3_tiles": "cluster_pack", "k_pack3_direct": "cluster_pack", "k_pack3_stream": "cluster_pack", "k_pack3_slow": "cluster_pack",
    "k_isize_count": "isize_stats", "k_isize_collect": "isize_stats", "k_isize_reduce": "isize_stats",
    "k_tile_mark_windows": "getsv_scan", "k_tile_mark_junctions": "getsv_scan", "k_getsv_scan": "getsv_scan", "k_getsv_scan_runs": "getsv_scan", "k_max_span": "getsv_scan",
    "k_getsv_cand": "getsv_cand", "k_getsv_cand_dense": "getsv_cand", "k_dense_tiles": "getsv_cand", "k_cap_mark": "getsv_cand", "k_cap_sweep": "getsv_cand", "k_cap_tail": "getsv_cand", "k_cap_regrow": "getsv_cand",
    "k_depth_prefix": "depth_finish", "k_range_sum": "depth_finish", "k_point_depth": "depth_finish",
    "k_build_rec": "h2d",
}
DOUBLE_FETCH = {"k_clip_scan", "k_clip_scan_ends", "k_getsv_scan", "k_getsv_scan_runs"}


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("ssv::", "")
    return name


def base(name):
    return re.sub(r"<.*$", "", name)


def load(d, counter):
    per = collections.OrderedDict()
    rows = list(csv.DictReader(open(glob.glob(os.path.join(d, "*counter_collection.csv"))[0])))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        k = int(r["Dispatch_Id"])
        e = per.setdefault(k, [short(r["Kernel_Name"]), 0.0, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3])
        e[1] += float(r["Counter_Value"])
    return list(per.values())


def main():
    fetch_dir, write_dir, records, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    tag = sys.argv[5] if len(sys.argv) > 5 else "r02"
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")

    def per_kernel(disp):
        out, dur, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
        for name, v, d in disp:
            out[name] += v; dur[name] += d; cnt[name] += 1
        return out, dur, cnt

    def per_group(disp, fetch_side):
        g, cur = collections.defaultdict(float), None
        for name, v, _ in disp:
            b = base(name)
            if b.startswith(("k_sy_", "__amd_rocclr_copy")):
                continue  # the synthetic generator and runtime copies are not part of the path
            if b in GROUP_OF:
                cur = GROUP_OF[b]
            if cur is None:
                continue
            g[cur] += v * 1024 * (2 if fetch_side and b in DOUBLE_FETCH else 1)
        return g

    fk, fd, fc = per_kernel(fetch)
    wk, _, _ = per_kernel(write)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_write.csv"), "w") as o:
        o.write(f"# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap` ({records:,} records, {steps} steps run)\n")
        o.write("# values in KB as reported, averaged over the kernel's dispatches. gfx950: FETCH_SIZE counts a 16-B/lane coalesced stream at HALF its bytes (MI355X_MICROARCH.md, HBM); other widths are uncalibrated\n")
        o.write("kernel,dispatches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,avg_duration_us_in_pmc_run\n")
        for k in sorted(set(fk) | set(wk)):
            if k.startswith(("k_sy_", "__amd")):
                continue
            n = max(fc.get(k, 0), 1)
            o.write(f"{k},{n},{fk.get(k, 0.0) / n:.1f},{wk.get(k, 0.0) / n:.1f},{fd.get(k, 0.0) / n:.1f}\n")
    gf, gw = per_group(fetch, True), per_group(write, False)
    groups = {g: (gf.get(g, 0.0) + gw.get(g, 0.0)) / steps for g in sorted(set(gf) | set(gw)) if g != "h2d"}
    t = {"_note": f"HBM bytes per step from the rocprofv3 PMC passes in profiles/{tag}_pmc_fetch_write.csv: FETCH_SIZE (doubled for the two streaming scans, whose 16-B/lane loads gfx950 counts at "
                  "half: MI355X_MICROARCH.md) + WRITE_SIZE per kernel group of bench.py, device-wide scans and fills counted with the kernel they follow; regenerate with tools/pmc_traffic.py",
         "_round": tag, "records": records, "groups": groups, "path_bytes_per_step": sum(groups.values()),
         "path_bytes_per_record": sum(groups.values()) / records}
    json.dump(t, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(t, indent=1))


if __name__ == "__main__":
    main()
