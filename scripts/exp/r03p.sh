#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_neumf.py tests/test_gpu_gemm.py -q > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 600 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 20 > $O/nmf.log 2>&1
tail -14 $O/nmf.log
timeout 600 python bench.py --legs vae --no-cpu-baseline --repeats 3 2> $O/vae.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); v=d['vae']
print('VAE', round(v['ms_per_step'],4), round(v['value']), {k:round(x,4) for k,x in v['roofline']['kernels_ms_per_step'].items()})"
