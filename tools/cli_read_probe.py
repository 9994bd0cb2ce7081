#!/usr/bin/env python3
"""Where does the time of the file-reading side go?  Writes a synthetic 30x sample (bases and qualities for every record) as a BAM in /dev/shm and runs
`seeksv getsv -Z -B` / `seeksv getclip -Z` on it with SSV_TIMING=2 (per chunk: waited for the reader / decoded; per read_blocks call:
pread / header walk), under the environment variants given on the command line, then tools/read_rate on the same file.
usage: python tools/cli_read_probe.py [genome_frac=0.25] [VAR=VALUE,VAR=VALUE ...]   (each further argument is one variant)"""
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from seeksv_amd import host, synth  # noqa: E402

EXE = os.path.join(ROOT, "seeksv_amd", "bin", "seeksv")


def main():
    frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
    variants = [dict(kv.split("=", 1) for kv in a.split(",") if kv) for a in sys.argv[2:]] or [{}]
    cores = bench.effective_cpus()
    os.environ["SSV_BGZF_LEVEL"] = "4"
    os.environ.setdefault("SSV_WRITE_THREADS", str(cores))
    w = synth.Workload(genome_frac=frac, depth=30, n_sv=max(1, round(10000 * frac)))
    d = tempfile.mkdtemp(prefix="ssv_probe_", dir="/dev/shm")
    try:
        bam = os.path.join(d, "s.bam")
        t0 = time.perf_counter()
        chunk = 500_000
        starts = list(range(0, w.n_total, chunk))

        def batches():
            with ThreadPoolExecutor(max_workers=cores) as ex:
                for i in range(0, len(starts), 2 * cores):
                    yield from ex.map(lambda g: w.generate_host(g, min(chunk, w.n_total - g), False, True), starts[i:i + 2 * cores])
        host.write_bam(bam, w.names, w.lens, batches())
        print(f"# {w.n_total} records, {os.path.getsize(bam)} bytes, written in {time.perf_counter() - t0:.1f} s", flush=True)
        jfile = os.path.join(d, "junctions.txt")
        with open(jfile, "w") as f:
            for j in w.junctions:
                f.write("\t".join(str(x) for x in (j[0], j[1], j[2], 0, j[3], j[4], j[5], 0, 0, 0, "NA", 0, 0, 0, 0, 0, 0, 0, 0, "50M", "50M", "ACGT", "ACGT")) + "\n")
        empty_bam, empty_clip = os.path.join(d, "e.clip.bam"), os.path.join(d, "e.clip")
        host.write_bam(empty_bam, w.names, w.lens, [])
        open(empty_clip, "w").close()
        for v in variants:
            env = dict(os.environ, SSV_TIMING="2", **v)
            for name, cmd in (("getsv", [EXE, "getsv", "-Z", "-d", "0", "-f", "0", "-b", "0", "-B", jfile, empty_bam, bam, empty_clip, os.path.join(d, "o.sv"), os.path.join(d, "o.fq")]),
                              ("getclip", [EXE, "getclip", "-Z", "-o", os.path.join(d, "o"), bam])):
                for rep in range(2):
                    t, w0 = time.perf_counter(), time.time()
                    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
                    dt, w1 = time.perf_counter() - t, time.time()
                    stamps = [float(l.split(":")[1]) for l in r.stderr.splitlines() if l.startswith("[stamp] wall clock at")]
                    where = f"; exec -> main {stamps[0] - w0:.3f} s, main -> exit {stamps[1] - stamps[0]:.3f} s, exit -> reaped {w1 - stamps[1]:.3f} s" if len(stamps) == 2 else ""
                    print(f"## {name} {json.dumps(v)} rep {rep}: {dt:.3f} s = {w.n_total / dt / 1e6:.1f} M records/s (rc {r.returncode}){where}", flush=True)
                    if rep == 1 or r.returncode != 0:
                        print("\n".join(l for l in r.stderr.splitlines() if l.startswith("[timing]") or r.returncode != 0), flush=True)
        if os.environ.get("PROBE_RUN"):   # `seeksv run` on the same file: per-chunk retain timing, phases
            fa = os.path.join(d, "ref.fa")
            w.write_fasta(fa, cores)
            env = dict(os.environ, SSV_TIMING="2")
            env.pop("SSV_BGZF_LEVEL", None)
            for rep in range(2):
                t = time.perf_counter()
                r = subprocess.run([EXE, "run", bam, fa, os.path.join(d, "one")], capture_output=True, text=True, env=env)
                print(f"## run rep {rep}: {time.perf_counter() - t:.3f} s = {w.n_total / (time.perf_counter() - t) / 1e6:.1f} M records/s (rc {r.returncode})", flush=True)
                if rep == 1 or r.returncode != 0:
                    print("\n".join(l for l in r.stderr.splitlines() if l.startswith("[timing]") or r.returncode != 0), flush=True)
        rr = os.path.join(ROOT, "tools", "read_rate")
        if os.environ.get("PROBE_NO_READ_RATE"):
            rr = ""
        if os.path.exists(rr):
            print(subprocess.run([rr, "8", "/dev/shm", bam], capture_output=True, text=True).stdout, flush=True)
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
