#!/bin/bash
# Pass 2 of the device inflate in its two global-memory forms (16 lanes per BGZF block / a wavefront per block, SSV_RESOLVE=lanes16|wave) on the realistic
# synthetic file and on real reads.  Run from the repo root on a GPU box: tools/resolve_bench.sh > out.txt
export SSV_PROFILE=1
for m in lanes16 wave; do
  SSV_RESOLVE=$m SSV_BGZF_LEVEL=4 python3 tools/bamdec_bench.py real 0.03125 4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('realistic synthetic file, %.2f GB inflated in %d chunks, SSV_RESOLVE=$m: pass 1 %.2f ms, pass 2 %.2f ms, inflate %.1f GB/s of output' % (d['inflated_bytes']/1e9, d['chunks'], k['bam_inflate']-k['bam_resolve'], k['bam_resolve'], d['inflate_GBs_out']))"
  SSV_RESOLVE=$m SSV_CHUNK_COMP_GB=2 python3 tools/bamdec_bench.py example 1200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('real reads (example x 1200), %.2f GB inflated in %d chunk(s), SSV_RESOLVE=$m: pass 1 %.2f ms, pass 2 %.2f ms, inflate %.1f GB/s of output' % (d['inflated_bytes']/1e9, d['chunks'], k['bam_inflate']-k['bam_resolve'], k['bam_resolve'], d['inflate_GBs_out']))"
  SSV_RESOLVE=$m SSV_CHUNK_COMP_GB=2 python3 tools/bamdec_bench.py example 300 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('real reads (example x 300), %.2f GB inflated in %d chunk(s), SSV_RESOLVE=$m: pass 1 %.2f ms, pass 2 %.2f ms, inflate %.1f GB/s of output' % (d['inflated_bytes']/1e9, d['chunks'], k['bam_inflate']-k['bam_resolve'], k['bam_resolve'], d['inflate_GBs_out']))"
done
