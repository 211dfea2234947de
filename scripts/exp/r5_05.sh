#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_05; cd $R; mkdir -p gpurun_out/$T
EL_GEMM_B3W=1 timeout 300 python scripts/mb.py gemm > gpurun_out/$T/gemm_w1.txt 2>&1
(timeout 600 python -m pytest -q -m gpu --timeout 600 -x tests/test_gpu_dense.py -k "gemm" 2>&1 | tail -5) > gpurun_out/$T/pytest.log
cut -c1-110 gpurun_out/$T/gemm_w1.txt; tail -3 gpurun_out/$T/pytest.log
