#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bpr.py tests/test_gpu_plugin.py tests/test_gpu_fullsize.py tests/test_gpu_cml.py -q -x > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 600 python bench.py --legs bpr --no-cpu-baseline --repeats 3 2> $O/bpr.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('C2', round(d['ms_per_step'],4), round(d['value']/1e6,1), r['kernel'], round(r['frac'],3), {k:round(v,4) for k,v in r['kernels_ms_per_step'].items()}, 'topk', round(d['topk']['ms_per_step'],3))" 2>&1 | tee $O/bpr.log
for uc in 16 8 4; do
  EL_UCHUNK=$uc timeout 900 python bench.py --legs c4 --no-cpu-baseline --repeats 3 2> $O/c4_$uc.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); c=d['c4_one_gpu']; r=c['roofline']
print('C4 uchunk=$uc', round(c['ms_per_step'],4), round(c['value']/1e6,1), r['kernel'], {k:round(v,4) for k,v in r['kernels_ms_per_step'].items()})"
done 2>&1 | tee $O/c4.log
