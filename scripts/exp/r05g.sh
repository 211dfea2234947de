#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; T=${1:-r05g}; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_neumf.py -m gpu -q --timeout 600 2>&1 | tail -25) > gpurun_out/$T/pytest.log
cat gpurun_out/$T/pytest.log
for sp in 0 1; do echo "== EL_GEMM_SPLIT=$sp"; EL_GEMM_SPLIT=$sp timeout 300 python scripts/mb.py gemm 2>&1 | grep -v amdgpu.ids | cut -c1-120; done > gpurun_out/$T/gemm.txt
cat gpurun_out/$T/gemm.txt
