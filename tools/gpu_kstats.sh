#!/bin/bash
# Per-kernel times of the bench workload (rocprofv3 --kernel-trace --stats): tools/gpu_kstats.sh <tag> [bench args]  -> gpurun_out/<tag>/kernel_stats.csv + summary
tag=${1:-ks}; shift; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-overlap --steps 10 --warmup 2 --no-cpu-baseline --no-config3 --no-host-batch --file-frac 0 "$@" > $out/bench_rocprof.json 2> $out/stats.err
cd $GRAFT_REPO_ROOT
find $out/stats -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/stats
python3 tools/kstats.py $out/kernel_stats.csv 15 | head -${KSTATS_LINES:-14}
