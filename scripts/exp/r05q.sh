#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; T=${1:-r05q}; mkdir -p gpurun_out/$T
(timeout 1200 python -m pytest tests/test_gpu_dense.py tests/test_gpu_neumf.py tests/test_gpu_tf_pins.py tests/test_gpu_fullsize_neumf.py tests/test_gpu_nmf_score.py -m gpu -q --timeout 900 2>&1 | tail -12) > gpurun_out/$T/pytest.log
cat gpurun_out/$T/pytest.log
EL_GEMM_SPLIT=1 timeout 300 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 6 --model NeuMF 2>&1 | grep -v amdgpu.ids | tail -14 > gpurun_out/$T/log.txt
EL_GEMM_SPLIT=1 timeout 300 python scripts/mb.py vae --iters 20 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/$T/log.txt
cat gpurun_out/$T/log.txt
