#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ah; mkdir -p $O
for ic in 256 128; do
EL_ICHUNK=$ic timeout 900 python bench.py --legs c4 --no-cpu-baseline --repeats 3 2> $O/c4_$ic.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); c=d['c4_one_gpu']; r=c['roofline']['kernels_ms_per_step']
print('ICHUNK $ic C4', round(c['ms_per_step'],4), round(c['value']/1e6,1), round(r['k_bpr_item_seg'],4))"
done
EL_ICHUNK=256 timeout 900 python bench.py --legs sweep --no-cpu-baseline --repeats 3 2> $O/sweep.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
for p in d['batch_sweep']['points'][:3]: print('ICHUNK 256', p['optimizer'], p['batch'], round(p['ms_per_step'],4), round(p['value']/1e6,1))"
