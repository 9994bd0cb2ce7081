#!/usr/bin/env python3
"""Where does the END of `seeksv run` go?  A synthetic 30x sample (genome fraction argv[1], default 0.25) as a BAM + its reference, then `seeksv run` three ways with the
command's own wall-clock stamps: as it is (what the caller waits for behind the last statement: exit_to_reaped_s), with SSV_EXIT_PROBE=1 (every kind of memory given back by
hand, timed) and with SSV_PINNED=malloc (page-locked memory from hipHostMalloc's 4 KB pages, the form before round 6)."""
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
    d = tempfile.mkdtemp(prefix="ssv_exit_", dir="/dev/shm")
    exe = os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")
    try:
        bam, fa = os.path.join(d, "s.bam"), os.path.join(d, "ref.fa")
        bench.write_workload_bam(w, bam, -2)
        os.environ.pop("SSV_BGZF_LEVEL", None)
        w.write_fasta(fa, bench.effective_cpus())
        out = {"records": w.n_total, "bam_bytes": os.path.getsize(bam)}
        for tag, extra in (("as it is", {}), ("exit probe", {"SSV_EXIT_PROBE": "1"}), ("hipHostMalloc", {"SSV_PINNED": "malloc"}), ("as it is, again", {})):
            env = dict(os.environ, SSV_TIMING="1", **extra)
            r, cur = bench.run_command([exe, "run", bam, fa, os.path.join(d, "o")], env)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-400:])
            out[tag] = {k: cur.get(k) for k in ("total_s", "exec_to_main_s", "exit_to_reaped_s")}
            out[tag]["phases_s"] = {k: v for k, v in cur["phases_s"].items() if k.startswith("run:")}
        print(json.dumps(out, indent=1))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
