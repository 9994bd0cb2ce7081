#!/usr/bin/env python3
"""One line per bench.py output file (argv): the figures a comparison between two builds or two switches looks at - value, step, device time of a step, the
cluster_pack group, hbm_frac_measured, config3_path, the file leg's rate and inflate / resolve / records kernels, cli_path and `seeksv run`."""
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    r=d['roofline']; fp=d.get('file_path',{}); km=fp.get('kernel_ms_per_run',{})
    print(f, 'value %.2f ms/step %.2f dev %.4f pack %.4f hbm %.3f | c3 %.2f | file %.0f M inflate %.1f resolve %.1f records %.1f total_s %.3f | cli %s run %s' % (
        d['value']/1e9, d['ms_per_step'], r['avg_launch_ms'], r['groups']['cluster_pack']['ms'], r.get('hbm_frac_measured') or 0,
        (d.get('config3_path',{}).get('value') or 0)/1e9, (fp.get('value') or 0)/1e6, km.get('bam_inflate',0), km.get('bam_resolve',0), km.get('bam_records',0), fp.get('total_s',0),
        fp.get('cli_path',{}).get('total_s'), (fp.get('cli_path',{}).get('run') or {}).get('total_s')))
