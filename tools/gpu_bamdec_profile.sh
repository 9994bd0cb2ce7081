#!/bin/bash
# What profiles/ holds about the BAM-file path, in one GPU-box call (run from the repo root on the box):
#   tools/gpu_bamdec_profile.sh <tag>  -> gpurun_out/<tag>/{bench_default.json, real_1200.json, real_2000.json, synth.json, inflate_kernel_stats.csv}
#     bench_default.json       the bench line as the driver runs it (file_path leg included)
#     real_1200 / real_2000    tools/bamdec_bench.py on real reads (the bundled example BAM's records written 1200 / 2000 times): one 4.0 GB / 6.7 GB chunk
#     synth.json               the same on the synthetic 30x sample (1/32 genome)
#     inflate_kernel_stats.csv rocprofv3 --kernel-trace --stats of the real-reads run (per-kernel average durations)
tag=${1:-bamdec}; o=gpurun_out/$tag; mkdir -p $o
export TMPDIR=/tmp
python3 bench.py > $o/bench_default.json 2> $o/bench.err
export SSV_PROFILE=1
SSV_CHUNK_COMP_GB=2 python3 tools/bamdec_bench.py example 1200 2>&1 | tail -1 > $o/real_1200.json
SSV_CHUNK_COMP_GB=3 python3 tools/bamdec_bench.py example 2000 2>&1 | tail -1 > $o/real_2000.json
python3 tools/bamdec_bench.py 2>&1 | tail -1 > $o/synth.json
export SSV_CHUNK_COMP_GB=2
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o x -- python3 tools/bamdec_bench.py example 1200 > $o/rocprof_run.json 2> $o/rocprof.err
find $o/stats -name '*kernel_stats.csv' -exec cp {} $o/inflate_kernel_stats.csv \;
rm -rf $o/stats
for f in real_1200 real_2000 synth; do
  python3 -c "
import json; d=json.load(open('$o/$f.json')); print('$f', d['kernel_ms'], d['inflate_GBs_out'], d['device_records_per_s'], d['device_kernels_records_per_s'])"
done
python3 -c "
import json; d=json.load(open('$o/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); f=d['file_path']; print(f['value'], f['runs_total_s'], f['kernel_ms_per_run'])"
