#!/usr/bin/env python3
"""Does a device-to-host copy pay for the time the link stood idle before it?  0.46 GB from device memory into torch's pinned memory, each copy behind an idle gap of 0 ... 100 ms (the GPU
idle too, or busy with a kernel that touches no host memory): milliseconds per copy, five each.  (the cluster table's copy takes 9.7 or 17-18 ms.)"""
import json
import time

import torch

torch.cuda.set_device(0)
n = 464930560
d = torch.full((n,), 7, dtype=torch.uint8, device="cuda:0")
h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
work = torch.empty(1 << 28, dtype=torch.float32, device="cuda:0")
out = {"boot_id": open("/proc/sys/kernel/random/boot_id").read().strip(), "idle_gap_ms": {}, "gpu_busy_gap_ms": {}}
for _ in range(3):
    h.copy_(d, non_blocking=True); torch.cuda.synchronize()
for gap in (0, 1, 2, 5, 10, 20, 50, 100):
    ms = []
    for _ in range(5):
        torch.cuda.synchronize()
        time.sleep(gap / 1e3)
        t = time.perf_counter()
        h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        ms.append(round((time.perf_counter() - t) * 1e3, 2))
    out["idle_gap_ms"][gap] = ms
for gap in (2, 5, 10, 20):
    ms = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < gap:   # kernels only: no traffic on the host link
            work.mul_(1.0001)
            torch.cuda.synchronize()
        t = time.perf_counter()
        h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        ms.append(round((time.perf_counter() - t) * 1e3, 2))
    out["gpu_busy_gap_ms"][gap] = ms
print(json.dumps(out))
