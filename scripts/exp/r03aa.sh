#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aa; mkdir -p $O
timeout 900 python bench.py --legs c4 --no-cpu-baseline --repeats 3 2> $O/c4.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); c=d['c4_one_gpu']; r=c['roofline']
print('C4', round(c['ms_per_step'],4), round(c['value']/1e6,1), r['kernel'], round(r['achieved']), round(r['frac'],3), r['traffic'], r['valu'])"
tail -3 $O/c4.err
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -q 2>&1 | tail -3
