#!/usr/bin/env python3
"""One worker of bench.py's all-cores CPU baseline: the oracle (plain-C restatement of the reference, 1 thread) on this worker's own slice
of the bench workload - records [g0, g0 + n) - for about `seconds` seconds.  Prints `records passes seconds`.  Never touches the GPU.
usage: cpu_baseline_worker.py genome_frac depth n_sv g0 n seconds"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    frac, depth, n_sv, g0, n, seconds = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
    import oracle_lib as O
    from seeksv_amd import host, synth
    w = synth.Workload(genome_frac=frac, depth=depth, n_sv=n_sv)
    b = w.generate_host(g0, n)
    hdr = host.Header(w.names, w.lens)
    # the junctions whose up end lies in this slice (what a range-partitioned CPU run would give this worker)
    lo, hi = (int(b["tid"][0]), int(b["pos"][0])), (int(b["tid"][-1]), int(b["pos"][-1]))
    mine = [j for j in w.junctions if lo <= (w.names.index(j[0]), j[1]) <= hi]
    passes, t0 = 0, time.perf_counter()
    while True:
        O.getclip([b])
        rc, npairs, mean, sd = O.isize_stats([b], 20, 5000000)
        plan = host.Plan(hdr, mine, mean, sd)
        O.discordant([b], plan.junctions, mean, sd, 4, 20)
        O.depth([b], plan.windows, plan.ranges, plan.points, 20)
        plan.close()
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or passes >= 1000:
            break
    print(n, passes, dt)


if __name__ == "__main__":
    main()
