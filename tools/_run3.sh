python -m pytest tests/test_hip_golden.py -x -q -m gpu -k "compact" 2>&1 | tail -3
python -m pytest tests/test_cli_gpu.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --file-frac 0 > gpurun_out/b2.json 2> gpurun_out/b2.err
python3 -c "
import json
d=json.loads(open('gpurun_out/b2.json').read().strip().splitlines()[-1])
print('value %.1f G/s' % (d['value']/1e9), d['ms_per_step'], d['kernel_ms_one_step'], d['table'])
"
