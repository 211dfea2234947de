#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r04h
(timeout 600 python -m pytest tests/test_gpu_bpr.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 2>&1 | tail -8) > gpurun_out/r04h/pytest.log
cat gpurun_out/r04h/pytest.log
for v in fused separate wave0 fused wave0; do
  unset EL_BPR_USER_WAVE_ROWS; export EL_BPR_USER_CATCHUP=$v; if [ $v = wave0 ]; then export EL_BPR_USER_CATCHUP=fused EL_BPR_USER_WAVE_ROWS=0; fi
  echo "=== $v" >> gpurun_out/r04h/log.txt
  timeout 300 python bench.py --legs bpr,c5 --no-cpu-baseline --repeats 1 2>> gpurun_out/r04h/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for n,x in (('main',d),('c5',d['c5_per_gpu'])):
    r=x['roofline']; print(n, round(x['ms_per_step'],4), {k: round(v,4) for k,v in sorted(r['kernels_ms_per_step'].items(), key=lambda kv:-kv[1]) if 'catchup' in k or 'flush' in k or 'seg' in k})
" >> gpurun_out/r04h/log.txt
done
cat gpurun_out/r04h/log.txt
