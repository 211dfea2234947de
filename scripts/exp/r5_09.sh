#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_09; cd $R; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest -q -m gpu --timeout 600 tests/test_gpu_dense.py tests/test_gpu_neumf.py tests/test_gpu_fullsize_neumf.py tests/test_gpu_plugin.py tests/test_gpu_pointwise.py 2>&1 | tail -30) > gpurun_out/$T/pytest.log
timeout 300 python scripts/mb.py vae --iters 30 > gpurun_out/$T/vae.txt 2>&1
EL_NMF_SIDE=0 timeout 300 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 10 --model NeuMF > gpurun_out/$T/nmf_s0.txt 2>&1
EL_NMF_SIDE=1 timeout 300 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 10 --model NeuMF > gpurun_out/$T/nmf_s1.txt 2>&1
tail -6 gpurun_out/$T/pytest.log
for f in vae nmf_s0 nmf_s1; do echo == $f; tail -1 gpurun_out/$T/$f.txt; done
