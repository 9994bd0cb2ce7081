python -m pytest tests/test_hip_golden.py -x -q -m gpu -k "compact" 2>&1 | tail -3
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --file-frac 0"
pk() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'value %.1f G/s' % (d['value']/1e9), 'pack %.3f' % d['kernel_ms_one_step']['cluster_pack'], 'all %.3f' % d['roofline']['avg_launch_ms'], 'd2h %.2f' % d['kernel_ms_one_step']['table_d2h'], d['table']['bytes_per_cluster'])
" $1 "$2"; }
$B > gpurun_out/d0.json 2>/dev/null; pk gpurun_out/d0.json "dense default"
SSV_PACK3=direct $B > gpurun_out/d1.json 2>/dev/null; pk gpurun_out/d1.json "direct"
SSV_PACK3_BLOCKS=1536 $B > gpurun_out/d2.json 2>/dev/null; pk gpurun_out/d2.json "dense 1536"
SSV_PACK3_BLOCKS=1792 $B > gpurun_out/d3.json 2>/dev/null; pk gpurun_out/d3.json "dense 1792"
SSV_PACK3_BLOCKS=3584 $B > gpurun_out/d4.json 2>/dev/null; pk gpurun_out/d4.json "dense 3584"
SSV_QUAL_GROUPS=0 $B > gpurun_out/d5.json 2>/dev/null; pk gpurun_out/d5.json "dense nogroups"
SSV_QUAL_GROUPS=0 SSV_PACK3=direct $B > gpurun_out/d6.json 2>/dev/null; pk gpurun_out/d6.json "direct nogroups"
