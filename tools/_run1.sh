timeout 300 python -m pytest tests/test_hip_golden.py -x -q -m gpu -k "compact or packed" 2>&1 | tail -3
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st1 -o bench -- python3 bench.py --no-overlap --steps 10 --warmup 2 --no-cpu-baseline --file-frac 0 > gpurun_out/st1.json 2> gpurun_out/st1.err
python3 tools/kstat.py gpurun_out/st1 10 40 2>/dev/null | grep "ssv::" | head -16
