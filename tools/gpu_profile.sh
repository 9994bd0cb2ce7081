#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call (run from the repo root on the box):
#   tools/gpu_profile.sh <tag>      -> gpurun_out/<tag>/{bench_default.json, bench_under_rocprof.json, kernel_stats.csv, pmc_fetch/, pmc_write/}
# Counters are collected in their own passes (--pmc with --kernel-trace only), FETCH_SIZE and WRITE_SIZE separately.
tag=${1:-prof}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --no-overlap --steps 10 --warmup 2 --no-cpu-baseline --no-config3 --no-host-batch --file-frac 0 > $out/bench_under_rocprof.json 2> $out/stats.err
find $out/stats -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/stats
for ctr in FETCH_SIZE WRITE_SIZE; do
  d=$out/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-config3 --no-host-batch --no-overlap --file-frac 0 > $d.json 2> $d.err
  # keep the two csv files the traffic tool reads, flat
  find $d -name '*counter_collection.csv' -exec mv {} $d/x_counter_collection.csv \; 2>/dev/null
  find $d -name '*kernel_trace.csv' -exec mv {} $d/x_kernel_trace.csv \; 2>/dev/null
  find $d -mindepth 1 -type d -exec rm -rf {} + 2>/dev/null
  find $d -type f ! -name 'x_counter_collection.csv' ! -name 'x_kernel_trace.csv' -delete
done
ls -la $out $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
cut -c1-600 $out/bench_default.json
