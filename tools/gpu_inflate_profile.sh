#!/bin/bash
# The two inflate kernels (pass 1: k_bgzf_tokens_wave / k_bgzf_tokens, pass 2: k_bgzf_resolve) under rocprofv3, chunking stated:
#   tools/gpu_inflate_profile.sh <tag>  -> gpurun_out/<tag>/{real,example}_kernel_stats.csv, {real,example}_pmc.txt, *_bench.json
#   real     = tools/bamdec_bench.py real 0.03125 4   the realistic synthetic file (bases + qualities for every record, BGZF level 4 like the bench file leg), 19.3 M records,
#              5.3 GB inflated, decoded in chunks of <= 4 GB inflated (two chunks: ~62 K + ~19 K BGZF blocks)
#   example  = SSV_CHUNK_COMP_GB=2 tools/bamdec_bench.py example 1200   real reads (the bundled example BAM x 1200), ONE chunk of 4.0 GB inflated = 61 K blocks
# Time: --kernel-trace --stats; counters: one --pmc pass per counter group with --kernel-trace only (the pool refuses other combinations).
tag=${1:-inflate}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
run() { # name, env assignments..., then the command's arguments after "--"
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for e in "${envs[@]}"; do export "$e"; done
  python3 $GRAFT_REPO_ROOT/tools/bamdec_bench.py "$@" 2>/dev/null | tail -1 > $out/${name}_bench.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -o x -- python3 $GRAFT_REPO_ROOT/tools/bamdec_bench.py "$@" > /dev/null 2> $out/${name}_stats.err
  find $out/st -name '*kernel_stats.csv' -exec cp {} $out/${name}_kernel_stats.csv \;
  rm -rf $out/st
  : > $out/${name}_pmc.txt
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "k_bgzf" --output-format csv -d $out/pm -o x -- python3 $GRAFT_REPO_ROOT/tools/bamdec_bench.py "$@" > /dev/null 2> $out/${name}_pmc.err
    python3 - $out/pm >> $out/${name}_pmc.txt <<'PY'
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    n = len(next(iter(v.values())))
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "mean over", n, "dispatches; sum over them:", {c: round(sum(x), 1) for c, x in v.items()})
PY
    rm -rf $out/pm
  done
  for e in "${envs[@]}"; do unset "${e%%=*}"; done
  python3 - $out/${name}_bench.json $out/${name}_kernel_stats.csv <<'PY'
import csv, json, sys
d = json.load(open(sys.argv[1]))
print(d["input"], "| inflated", d["inflated_bytes"], "bytes in", d["chunks"], "chunk(s) | kernel_ms", {k: d["kernel_ms"][k] for k in ("bam_inflate", "bam_resolve") if k in d["kernel_ms"]}, "| inflate GB/s of output", d["inflate_GBs_out"])
for r in csv.DictReader(open(sys.argv[2])):
    if "bgzf" in r["Name"]:
        print("  ", r["Name"].split("(")[0][:50], "calls", r["Calls"], "avg ns", r["AverageNs"], "total ns", r["TotalDurationNs"])
PY
}
run real SSV_PROFILE=1 SSV_BGZF_LEVEL=4 -- real 0.03125 4
run example SSV_PROFILE=1 SSV_CHUNK_COMP_GB=2 -- example 1200
cd $GRAFT_REPO_ROOT
cat $out/real_pmc.txt $out/example_pmc.txt
