#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05k
for v in nosplit noload nomfma nosl; do
  echo "=== $v" >> gpurun_out/r05k/log.txt
  EL_LIB_PATH=$R/elliot_amd/csrc/variants/libelliot_hip_$v.so timeout 300 python scripts/mb.py gemm --shape 4096,4096,4096,0,0 --shape 262144,256,512,0,1 --shape 262144,512,256,0,0 2>&1 | grep -v amdgpu.ids | cut -c1-110 >> gpurun_out/r05k/log.txt
done
cat gpurun_out/r05k/log.txt
