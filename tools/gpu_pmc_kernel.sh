#!/bin/bash
# PMC counters of the kernels matching a regex in one bench step: tools/gpu_pmc_kernel.sh <tag> <kernel regex> "<counters of pass 1>" ["<pass 2>" ...]
# (each pass is its own run with --kernel-trace only, as the pool requires)  -> gpurun_out/<tag>/pass<k>.csv + a printed per-kernel mean
tag=$1; re=$2; shift 2; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
k=0
for ctrs in "$@"; do
  k=$((k+1)); d=$out/p$k
  # PMC_CMD = the python command line to profile (default: one bench step)
  rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "$re" --output-format csv -d $d -o x -- python3 ${PMC_CMD:-$GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config3 --no-host-batch --no-overlap --file-frac 0} > $d.json 2> $d.err
  find $d -name '*counter_collection.csv' -exec cp {} $out/pass$k.csv \;
  rm -rf $d
  python3 - $out/pass$k.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "dispatches", len(next(iter(v.values()))))
PY
done
