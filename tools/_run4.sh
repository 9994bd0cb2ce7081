export SSV_PROFILE=1
run() { SSV_RESOLVE=wave SSV_BGZF_LEVEL=4 python3 tools/bamdec_bench.py real 0.03125 4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$1: %.2f GB inflated in %d chunks: pass 1 %.2f ms, pass 2 %.2f ms, inflate %.1f GB/s of output' % (d['inflated_bytes']/1e9, d['chunks'], k['bam_inflate']-k['bam_resolve'], k['bam_resolve'], d['inflate_GBs_out']))"; }
SSV_RESOLVE_PAD=0 run "pad 0 (32 waves/CU)"
SSV_RESOLVE_PAD=26000 run "pad 26000 (24 waves/CU)"
SSV_RESOLVE_PAD=40000 run "pad 40000 (16 waves/CU)"
SSV_RESOLVE_PAD=65536 run "pad 65536 (8 waves/CU)"
