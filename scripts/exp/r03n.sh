#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bpr.py tests/test_gpu_plugin.py tests/test_gpu_fullsize.py tests/test_gpu_bench_contract.py -q -x > $O/pytest.log 2>&1
tail -8 $O/pytest.log
for v in "x 4" "1 4"; do
  set -- $v
  if [ $1 = x ]; then unset EL_BPR_DEFERRED; else export EL_BPR_DEFERRED=$1; fi
  EL_FUSED_RPG=$2 timeout 600 python bench.py --legs bpr --no-cpu-baseline --repeats 3 2> $O/bpr_def$1_$2.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('C2 deferred=$1 rpg=$2', round(d['ms_per_step'],4), round(d['value']/1e6,1), r['kernel'], round(r['frac'],3), r.get('user_rows_per_step'), {k:round(v,4) for k,v in r['kernels_ms_per_step'].items()}, 'topk', round(d['topk']['ms_per_step'],3))"
done 2>&1 | tee $O/bpr.log
unset EL_BPR_DEFERRED
for v in "4"; do
  EL_FUSED_RPG=$v timeout 900 python bench.py --legs c4 --no-cpu-baseline --repeats 3 2> $O/c4_$v.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); c=d['c4_one_gpu']; r=c['roofline']
print('C4 auto rpg=$v', round(c['ms_per_step'],4), round(c['value']/1e6,1), r['kernel'], round(r['frac'],3), r.get('user_rows_per_step'), {k:round(v,4) for k,v in r['kernels_ms_per_step'].items()})"
done 2>&1 | tee $O/c4.log
timeout 900 python bench.py --legs sweep --no-cpu-baseline --repeats 3 2> $O/sweep.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
for p in d['batch_sweep']['points']: print(p['optimizer'], p['batch'], round(p['ms_per_step'],4), round(p['value']/1e6,1), p.get('deferred_decay'))" 2>&1 | tee $O/sweep.log
