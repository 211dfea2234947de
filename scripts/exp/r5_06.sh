#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_06; cd $R; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest -q -m gpu --timeout 600 tests/test_gpu_dense.py tests/test_gpu_neumf.py tests/test_gpu_fullsize_neumf.py 2>&1 | tail -30) > gpurun_out/$T/pytest.log
timeout 300 python scripts/mb.py gemm > gpurun_out/$T/gemm.txt 2>&1
timeout 300 python scripts/mb.py vae --iters 20 > gpurun_out/$T/vae.txt 2>&1
EL_NMF_FUSE_RELU_BWD=0 timeout 300 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 10 --model NeuMF > gpurun_out/$T/nmf_f0.txt 2>&1
EL_NMF_FUSE_RELU_BWD=1 timeout 300 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 10 --model NeuMF > gpurun_out/$T/nmf_f1.txt 2>&1
tail -6 gpurun_out/$T/pytest.log
cut -c1-110 gpurun_out/$T/gemm.txt
for f in vae nmf_f0 nmf_f1; do echo == $f; head -14 gpurun_out/$T/$f.txt | tail -12; tail -1 gpurun_out/$T/$f.txt; done
