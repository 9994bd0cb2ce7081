#!/bin/bash
# the two streaming scans as the bench times them (HIP events around every launch of the timed steps), under grid-size overrides given as arguments: VAR=VALUE ...
for kv in "${@:-X=0}"; do
  env "$kv" python3 bench.py --steps 10 --warmup 2 --file-frac 0 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['roofline']['streaming_kernels_in_timed_region']
print('$kv', 'clip_scan ms', t['clip_scan']['avg_launch_ms'], 'GB/s', t['clip_scan']['achieved_GBs'], '| getsv_scan ms', t['getsv_scan']['avg_launch_ms'], 'GB/s', t['getsv_scan']['achieved_GBs'], '| device ms per step', d['roofline']['avg_launch_ms'])"
done
