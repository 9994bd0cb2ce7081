#!/usr/bin/env python3
"""The cluster table's copy to the host, leg by leg: bench.py's first leg (30x) and its config-3 leg (300x over a tenth of the genome) in one process, like bench.py runs them, with - behind each leg's
steps - a plain torch device-to-host copy of the table's size into torch's own pinned memory, and the library's copy timed alone (ssv_clip_cluster_async -> ssv_clip_table_wait, nothing else in flight).
On some boxes the second leg's copy takes 12-15 ms instead of 9.7: is it the table's buffers, or the process?"""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from seeksv_amd import synth  # noqa: E402
from seeksv_amd.device import Context  # noqa: E402


def torch_copy(nbytes):
    d = torch.full((nbytes,), 7, dtype=torch.uint8, device="cuda:0")
    h = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    out = []
    for _ in range(4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t) * 1e3, 2))
    del d, h
    return out


def leg(ctx, depth, frac, out, tag):
    w = synth.Workload(genome_frac=frac, depth=depth, n_sv=10000)
    b, keep = w.generate_device(0, w.n_total, 0, soa=False, persistent=True)
    ctx.clip_table_format(3)
    ms = []
    for k in range(6):
        ctx.clip_begin(0.9, 1, False, None, 0)
        ctx.clip_scan(b)
        ctx.clip_cluster_async()
        t = time.perf_counter()
        tab = ctx.clip_table_wait()
        ms.append(round((time.perf_counter() - t) * 1e3, 2))
    nbytes = int(tab.str_bytes)
    out[tag] = {"records": w.n_total, "table_str_bytes": nbytes, "library_copy_alone_ms": ms, "torch_copy_of_that_size_ms": torch_copy(nbytes)}
    del b, keep
    gc.collect()
    torch.cuda.empty_cache()


def main():
    torch.cuda.set_device(0)
    out = {"boot_id": open("/proc/sys/kernel/random/boot_id").read().strip()}
    out["torch_copy_before_anything_ms"] = torch_copy(553303952)
    with Context(0) as ctx:
        leg(ctx, 30.0, 1.0, out, "leg 1 (30x)")
        leg(ctx, 300.0, 0.1, out, "leg 2 (300x over a tenth)")
        leg(ctx, 30.0, 1.0, out, "leg 3 (30x again)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
