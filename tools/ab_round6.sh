# round 6: the one-pass columns kernel (k_cluster_cols3_chain) against the columns by scans (SSV_PACK_COLS=split), quarter-size file leg
set -x
timeout 900 python -m pytest tests/test_hip_golden.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 900 python -m pytest tests/test_cli_gpu.py -q -x -k "test_cli_getclip" 2>&1 | grep -E "passed|failed|error" | tail -3
Q="--file-frac 0.25 --no-host-batch --no-cpu-baseline --cpu-sample 0 --ref-sample 0 --config5-frac 0"
for v in default split default2 split2; do
  case $v in split*) export SSV_PACK_COLS=split;; *) unset SSV_PACK_COLS;; esac
  timeout 900 python bench.py $Q > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
done
python tools/bench_pick.py gpurun_out/ab_default.json gpurun_out/ab_split.json gpurun_out/ab_default2.json gpurun_out/ab_split2.json
