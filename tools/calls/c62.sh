#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/c62_bench.json 2> gpurun_out/c62_bench.log; echo "bench rc $?"
python -c "
import json; d=json.loads(open('gpurun_out/c62_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['metric'], round(d['value']), d['unit'], round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'traffic', r['traffic'], 'cpu', round(d['cpu_baseline']['value']), 'also', d.get('also'))"
