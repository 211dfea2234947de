#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ag; mkdir -p $O
for ic in 256 512 1024 384 128; do
EL_ICHUNK=$ic timeout 600 python bench.py --legs bpr --no-cpu-baseline --repeats 3 2> $O/bpr_$ic.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']['kernels_ms_per_step']
print('ICHUNK $ic C2', round(d['ms_per_step'],4), round(d['value']/1e6,1), round(r['k_bpr_item_seg'],4))"
done
