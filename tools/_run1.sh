timeout 300 python -m pytest tests/test_hip_golden.py -x -q -m gpu -k "compact or packed" 2>&1 | tail -3
B="timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --file-frac 0"
pk() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'value %.1f G/s' % (d['value']/1e9), 'pack %.3f' % d['kernel_ms_one_step']['cluster_pack'], 'all %.3f' % d['roofline']['avg_launch_ms'], d['table']['bytes_per_cluster'], d['kernel_ms_one_step'])
" $1 "$2"; }
$B > gpurun_out/d0.json 2>/dev/null; pk gpurun_out/d0.json "default"
$B > gpurun_out/d0.json 2>/dev/null; pk gpurun_out/d0.json "default"
