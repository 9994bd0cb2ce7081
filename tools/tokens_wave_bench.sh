#!/bin/bash
# Pass 1 of the device inflate in its two forms (a lane per BGZF block / a wavefront per block, SSV_TOKENS=lanes|wave) at four chunk sizes, plus the wavefront
# form's phase counters: what profiles/r03_tokens_wave_bench.txt holds.  Run from the repo root on a GPU box: tools/tokens_wave_bench.sh > out.txt
for k in 75 300 1200; do
  for m in lanes wave; do
    SSV_TOKENS=$m SSV_CHUNK_COMP_GB=2 python3 tools/bamdec_bench.py example $k 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('real reads (example x $k), %.2f GB inflated, SSV_TOKENS=$m: pass 1 %.2f ms, pass 2 %.2f ms, inflate %.1f GB/s of output' % (d['inflated_bytes']/1e9, k['bam_inflate']-k['bam_resolve'], k['bam_resolve'], d['inflate_GBs_out']))"
  done
done
for m in lanes wave; do
  SSV_TOKENS=$m python3 tools/bamdec_bench.py real 0.03125 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('realistic synthetic file, %.2f GB inflated in %d chunks, SSV_TOKENS=$m: pass 1 %.2f ms, pass 2 %.2f ms, inflate %.1f GB/s of output' % (d['inflated_bytes']/1e9, d['chunks'], k['bam_inflate']-k['bam_resolve'], k['bam_resolve'], d['inflate_GBs_out']))"
done
SSV_TOKENS=wave SSV_INFLATE_PHASES=1 SSV_CHUNK_COMP_GB=2 python3 tools/bamdec_bench.py example 1200 2>&1 | grep "tokens wave" | tail -1
SSV_TOKENS=wave SSV_INFLATE_PHASES=1 python3 tools/bamdec_bench.py real 0.03125 2>&1 | grep "tokens wave" | tail -2
