#!/usr/bin/env python3
"""`seeksv run` (and `seeksv getclip -Z`) on a synthetic 30x sample (genome fraction argv[1], default 1.0) under a few chunk / staging sizes of the device-inflate
reader (SSV_CHUNK_INFLATED_MB, SSV_STAGE_MB): the command's wall clock and its phases.  The defaults (1 GB inflated, 384 MB of file) were set in round 4 on
`getclip` / `getsv` alone, where a command's fixed costs counted; `run` keeps the records in HBM and reads the file once."""
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from seeksv_amd import synth  # noqa: E402


def main():
    frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    w = synth.Workload(genome_frac=frac, depth=30, n_sv=max(8, round(10000 * frac)))
    d = tempfile.mkdtemp(prefix="ssv_chunk_", dir="/dev/shm")
    exe = os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")
    try:
        bam, fa = os.path.join(d, "s.bam"), os.path.join(d, "ref.fa")
        bench.write_workload_bam(w, bam, -2)
        os.environ.pop("SSV_BGZF_LEVEL", None)
        w.write_fasta(fa, bench.effective_cpus())
        out = {"records": w.n_total, "bam_bytes": os.path.getsize(bam)}
        print(json.dumps(out), flush=True)
        variants = [("default (1024 / 384)", {}), ("2048 / 768", {"SSV_CHUNK_INFLATED_MB": "2048", "SSV_STAGE_MB": "768"}), ("4096 / 1536", {"SSV_CHUNK_INFLATED_MB": "4096", "SSV_STAGE_MB": "1536"}),
                    ("512 / 192", {"SSV_CHUNK_INFLATED_MB": "512", "SSV_STAGE_MB": "192"}), ("default, again", {})]
        if len(sys.argv) > 2:  # e.g. "SSV_HOST_THREADS=6,8,12": one variant per value (and the default before and after)
            name, values = sys.argv[2].split("=")
            variants = [("default", {})] + [(f"{name}={v}", {name: v}) for v in values.split(",")] + [("default, again", {})]
        for tag, extra in variants:
            env = dict(os.environ, SSV_TIMING="1", **extra)
            row = {}
            for cmd, argv in (("run", [exe, "run", bam, fa, os.path.join(d, "o")]), ("getclip", [exe, "getclip", "-Z", "-o", os.path.join(d, "g"), bam])):
                r, cur = bench.run_command(argv, env)
                if r.returncode != 0:
                    raise RuntimeError(r.stderr[-400:])
                row[cmd] = {k: cur.get(k) for k in ("total_s", "cpu_s", "exit_to_reaped_s")}
                row[cmd]["phases_s"] = {k: v for k, v in cur["phases_s"].items() if k.startswith("run:") or k.startswith("bam_read") or k.startswith("open")}
            print(json.dumps({tag: row}), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
