#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; T=${1:-r05h}; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest tests/test_gpu_dense.py -m gpu -q --timeout 600 -k gemm 2>&1 | tail -5) > gpurun_out/$T/pytest.log
cat gpurun_out/$T/pytest.log
EL_GEMM_SPLIT=1 timeout 300 python scripts/mb.py gemm 2>&1 | grep -v amdgpu.ids | cut -c1-120 > gpurun_out/$T/gemm.txt
cat gpurun_out/$T/gemm.txt
