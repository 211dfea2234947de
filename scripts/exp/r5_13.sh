#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_13; cd $R; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest -q -m gpu --timeout 600 -x tests/test_gpu_dense.py tests/test_gpu_neumf.py 2>&1 | tail -12) > gpurun_out/$T/pytest.log
EL_GEMM_PERSIST=0 timeout 300 python scripts/mb.py gemm > gpurun_out/$T/gemm_p0.txt 2>&1
EL_GEMM_PERSIST=1 timeout 300 python scripts/mb.py gemm > gpurun_out/$T/gemm_p1.txt 2>&1
EL_GEMM_PERSIST=0 EL_NMF_SIDE=1 timeout 300 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 10 --model NeuMF > gpurun_out/$T/nmf_p0.txt 2>&1
EL_GEMM_PERSIST=1 EL_NMF_SIDE=1 timeout 300 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 10 --model NeuMF > gpurun_out/$T/nmf_p1.txt 2>&1
tail -5 gpurun_out/$T/pytest.log
paste -d'|' <(cut -c1-95 gpurun_out/$T/gemm_p0.txt) <(cut -c30-60 gpurun_out/$T/gemm_p1.txt)
for f in nmf_p0 nmf_p1; do echo == $f; tail -1 gpurun_out/$T/$f.txt; done
