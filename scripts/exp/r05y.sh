#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; T=${1:-r05y}; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_tf_pins.py -m gpu -q --timeout 600 2>&1 | tail -6) > gpurun_out/$T/pytest.log
cat gpurun_out/$T/pytest.log
timeout 300 python scripts/mb.py vae --iters 20 2>&1 | grep -v amdgpu.ids | tail -18 > gpurun_out/$T/log.txt
cat gpurun_out/$T/log.txt
timeout 300 python bench.py --legs bpr,vae --no-cpu-baseline --users 200000 --items 50000 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench vae', d['vae']['value'], d['vae']['ms_per_step'])"
