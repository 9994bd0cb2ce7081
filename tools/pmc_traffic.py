#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes of bench.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; the TCC block cannot hold both)
into profiles/r01_pmc_fetch_write.csv (per kernel) and profiles/traffic.json (HBM bytes per unit for the kernel groups bench.py prices).

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <records> <events>
  each dir holds <something>_counter_collection.csv and <something>_kernel_trace.csv as written by
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap
FETCH_SIZE / WRITE_SIZE are reported in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE shows exactly half the bytes of a
wide coalesced streaming read (16 B/lane), so it is doubled for the two streaming scans, whose loads are all of that kind; other
kernels' loads are narrower or scattered and are taken as reported."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("ssv::", "")


def load(d, counter):
    per_dispatch = collections.defaultdict(float)
    kern = {}
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "*counter_collection.csv"))[0])):
        if r["Counter_Name"] == counter:
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
            kern[r["Dispatch_Id"]] = short(r["Kernel_Name"])
    out = collections.defaultdict(list)
    for k, v in per_dispatch.items():
        out[kern[k]].append(v)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "*kernel_trace.csv"))[0])):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out, dur


def main():
    fetch_dir, write_dir, records, events = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    fetch, dur = load(fetch_dir, "FETCH_SIZE")
    write, _ = load(write_dir, "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        if k.startswith(("k_sy_", "__amd")):
            continue  # the synthetic generator and runtime copies are not part of the path
        f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
        rows.append((k, len(f), sum(f) / len(f), sum(w) / len(w), sum(dur[k]) / max(1, len(dur[k]))))
    with open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_write.csv"), "w") as o:
        o.write(f"# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap` ({records:,} records, {events:,} clip events)\n")
        o.write("# values in KB as reported, averaged over the kernel's dispatches. gfx950: FETCH_SIZE counts a 16-B/lane coalesced stream at HALF its bytes (MI355X_MICROARCH.md, HBM); other widths are uncalibrated\n")
        o.write("kernel,dispatches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,avg_duration_us_in_pmc_run\n")
        for k, n, f, w, d in rows:
            o.write(f"{k},{n},{f:.1f},{w:.1f},{d:.1f}\n")
    by = {k: (n, f * 1024, w * 1024) for k, n, f, w, d in rows}

    def total(names, per_launch_mult=None):
        fb = wb = 0.0
        for nm in names:
            for k, (n, f, w) in by.items():
                if k == nm or k.startswith(nm + "<"):
                    m = n if per_launch_mult is None else per_launch_mult
                    fb += f * m
                    wb += w * m
        return fb, wb

    t = {"_note": "HBM bytes per unit from the rocprofv3 PMC passes in profiles/r01_pmc_fetch_write.csv: (2 x FETCH_SIZE for kernels whose streams are 16-B/lane loads, "
                  "the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, divided by the units of one launch; regenerate with tools/pmc_traffic.py"}
    f, w = total(["k_clip_scan"], 1)
    t["clip_scan"] = {"bytes_per_record": (2 * f + w) / records, "fetch_raw_bytes": f, "write_bytes": w}
    f, w = total(["k_getsv_scan"], 1)
    t["getsv_scan"] = {"bytes_per_record": (2 * f + w) / records, "fetch_raw_bytes": f, "write_bytes": w}
    f, w = total(["k_clip_gather"], 1)
    t["clip_gather"] = {"bytes_per_record": (f + w) / events, "_unit": "event", "_correction": "none (4-B/lane loads: FETCH_SIZE taken as reported)"}
    f, w = total(["k_cluster_pack_meta", "k_cluster_pack_codes", "k_cluster_pack_ascii"], 1)  # both launches of k_cluster_pack_codes (dword path + bytewise list)
    t["cluster_pack"] = {"bytes_per_record": (f + w) / events, "_unit": "event slot", "_correction": "none"}
    n_steps = by["k_clip_scan"][0]  # steps in the profiled run (the sort kernels run several passes per step)
    f, w = total(["k_rs_hist", "k_rs_scatter"])
    f, w = f / n_steps, w / n_steps
    t["event_sort"] = {"bytes_per_record": (f + w) / events, "_unit": "event", "_correction": "none; all passes of k_rs_hist + k_rs_scatter of one step, histogram scans not included"}
    json.dump(t, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps({k: v["bytes_per_record"] for k, v in t.items() if isinstance(v, dict)}))


if __name__ == "__main__":
    main()
