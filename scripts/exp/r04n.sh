#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; T=${1:-r04n}; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest tests/test_gpu_nmf_score.py -m gpu -q --timeout 600 2>&1 | tail -40) > gpurun_out/$T/pytest.log
cat gpurun_out/$T/pytest.log
EL_NMF_SCREEN=1 timeout 300 python scripts/mb.py nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 3 2>&1 | grep -v amdgpu.ids | tail -18 > gpurun_out/$T/log.txt
cat gpurun_out/$T/log.txt
