#!/bin/bash
# One GPU-box call: parity tests, the default bench line, and the rocprofv3 kernel summary of the no-overlap bench.
#   tools/gpu_check.sh <tag> [quick|full]     (run on the GPU box from the repo root; writes gpurun_out/<tag>/)
tag=${1:-chk}; mode=${2:-quick}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ "$mode" = full ]; then
  (timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > $out/tests.log
else
  (timeout 300 python -m pytest tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -5) > $out/tests.log
fi
python bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --no-overlap --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_noov.json 2> $out/bench_noov.err
find $out/prof -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/prof
cat $out/tests.log
python3 - $out <<'PY'
import json, sys
o = sys.argv[1]
for f in ("bench.json", "bench_noov.json"):
    try:
        d = json.loads(open(f"{o}/{f}").read().strip().splitlines()[-1])
        print(f, "value %.3g ms/step %.2f" % (d["value"], d["ms_per_step"]), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), "device_ms", d["roofline"].get("device_kernels_ms_per_step"))
        print("  kernel_ms", d["kernel_ms_one_step"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
grep -v "k_sy_\|rocclr" $out/kernel_stats.csv | head -14 | cut -c1-60,100-200 | awk -F'",' '{print $1 "\"," $2}' | cut -c1-150
