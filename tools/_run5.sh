export TMPDIR=/tmp SSV_PROFILE=1 SSV_RESOLVE=wave SSV_BGZF_LEVEL=4
R=$GRAFT_REPO_ROOT
cd /tmp
for pad in 0 65536; do
  export SSV_RESOLVE_PAD=$pad
  rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-include-regex "k_bgzf_resolve_wave" --output-format csv -d /tmp/pp$pad -o x -- python3 $R/tools/bamdec_bench.py real 0.03125 4 > /tmp/pp$pad.json 2> /tmp/pp$pad.err
  f=$(find /tmp/pp$pad -name '*counter_collection.csv' | head -1)
  python3 - $f $pad <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list); lds=set(); 
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]].append(float(r["Counter_Value"])); lds.add(r.get("LDS_Block_Size"))
print("pad", sys.argv[2], "LDS_Block_Size", lds, {c: round(sum(x) / len(x) / 1e6, 1) for c, x in acc.items()}, "M per dispatch;", len(next(iter(acc.values()))), "dispatches")
PY
done
