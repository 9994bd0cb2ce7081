#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of the HIP library (hipcc -Rpass-analysis=kernel-resource-usage), no GPU needed."""
import re
import subprocess
import sys

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "seeksv_amd/csrc/seeksv_hip.hip"
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Iinclude", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], cwd=ROOT, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
        continue
    m = re.search(r"remark: .*? (VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" ")[0]] = int(m.group(2))
print(f"{'kernel':70s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    print(f"{r['name'][:70]:70s} {r.get('VGPRs', 0):5d} {r.get('TotalSGPRs', 0):5d} {r.get('ScratchSize', 0):8d} {r.get('Occupancy', 0):4d} {r.get('LDS', 0):7d}")
