python -m pytest tests/test_hip_golden.py -x -q -m gpu -k "compact" 2>&1 | tail -2
python -m pytest tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --file-frac 0 > gpurun_out/b3_$i.json 2> gpurun_out/b3.err
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/b3_$i.json').read().strip().splitlines()[-1])
print('value %.1f G/s' % (d['value']/1e9), d['ms_per_step'], d['wall_ms_one_step'], d['table'], d['wall_ms_timed_steps'])
"
done
SSV_BENCH_TRACE=1 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --file-frac 0 2>&1 >/dev/null | tail -12 | cut -c1-700
