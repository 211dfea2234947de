#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; T=${1:-r05n}; mkdir -p gpurun_out/$T
for sp in 0 1; do
echo "== EL_GEMM_SPLIT=$sp" >> gpurun_out/$T/log.txt
EL_GEMM_SPLIT=$sp timeout 300 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 6 --model NeuMF 2>&1 | grep -v amdgpu.ids | tail -16 >> gpurun_out/$T/log.txt
EL_GEMM_SPLIT=$sp timeout 300 python scripts/mb.py vae --iters 20 2>&1 | grep -v amdgpu.ids | tail -17 >> gpurun_out/$T/log.txt
done
cat gpurun_out/$T/log.txt
