#!/usr/bin/env python3
"""Where the CPU time of `seeksv getclip -Z` goes on a box that grants a fixed amount of host time (cgroup cpu.max): a synthetic 30x sample (genome fraction argv[1],
default 0.25) as a BAM in /dev/shm, then the command under a few settings with its wall clock, its user + system seconds (bench.run_command: cpu_s) and
SSV_TIMING=2's account by thread."""
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
    frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
    w = synth.Workload(genome_frac=frac, depth=30, n_sv=max(8, round(10000 * frac)))
    d = tempfile.mkdtemp(prefix="ssv_cpu_", dir="/dev/shm")
    exe = os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")
    try:
        bam = os.path.join(d, "s.bam")
        bench.write_workload_bam(w, bam, -2)
        os.environ.pop("SSV_BGZF_LEVEL", None)
        print(json.dumps({"records": w.n_total, "bam_bytes": os.path.getsize(bam), "host_cpus": bench.effective_cpus()}), flush=True)
        variants = [("default", {}), ("ROC_ACTIVE_WAIT_TIMEOUT=0", {"ROC_ACTIVE_WAIT_TIMEOUT": "0"}), ("8 reader threads", {"SSV_HOST_THREADS": "8"}),
                    ("gzip level 1 (zlib)", {"SSV_GZ_LEVEL": "1"}), ("default, again", {})]
        for tag, extra in variants:
            env = dict(os.environ, SSV_TIMING="2", **extra)
            r, cur = bench.run_command([exe, "getclip", "-Z", "-o", os.path.join(d, "g"), bam], env)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-400:])
            cpu = [l for l in r.stderr.splitlines() if l.startswith("[timing] (cpu:")]
            print(json.dumps({tag: {"total_s": cur["total_s"], "cpu_s": cur["cpu_s"], "exit_to_reaped_s": cur.get("exit_to_reaped_s"),
                                    "phases_s": {k: v for k, v in cur["phases_s"].items() if v >= 0.05}, "cpu": cpu[-1] if cpu else None}}), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
