#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 kernel_stats.csv of bench.py: tools/kstats.py <csv> <steps incl. warmup and breakdown>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
tot = 0
for r in rows:
    n = r['Name']
    if 'k_sy_' in n or 'rocclr' in n: continue
    short = re.sub(r'\(.*', '', n).replace('void ', '').replace('ssv::', '')
    calls = int(r['Calls']); avg = float(r['AverageNs']) / 1e3
    if calls < steps: continue
    per_step = float(r['TotalDurationNs']) / 1e3 / steps
    tot += per_step
    if per_step > 8: print(f"{short:55s} calls {calls:4d} avg {avg:9.1f} us  per-step {per_step:8.1f} us")
print("sum per step (us):", round(tot, 1))
