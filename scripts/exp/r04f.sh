#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r04f
for v in pk nopk simple pk simple; do
  unset EL_LIB_PATH
  if [ $v != pk ]; then export EL_LIB_PATH=$R/elliot_amd/csrc/variants/libelliot_hip_$v.so; fi
  echo "=== $v" >> gpurun_out/r04f/log.txt
  timeout 300 python bench.py --legs bpr,c5 --no-cpu-baseline --repeats 1 2>> gpurun_out/r04f/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for n,x in (('main',d),('c5',d['c5_per_gpu'])):
    r=x['roofline']; print(n, round(x['ms_per_step'],4), {k: round(v,4) for k,v in sorted(r['kernels_ms_per_step'].items(), key=lambda kv:-kv[1]) if 'catchup' in k or 'flush' in k or 'seg' in k})
" >> gpurun_out/r04f/log.txt
done
cat gpurun_out/r04f/log.txt
