#!/bin/bash
# sweep of the fp32 GEMM stream-K fast path on the model shapes
cd $GRAFT_REPO_ROOT
SH="--shape 512,26744,600,0,0 --shape 600,26744,512,1,0 --shape 512,600,26744,0,1 --shape 512,400,600,0,0 --shape 512,600,200,0,0 --shape 600,400,512,1,0 --shape 512,600,400,0,1 --shape 262144,512,256,0,0 --shape 262144,256,512,0,0 --shape 262144,128,256,0,0 --shape 262144,256,512,0,1 --shape 262144,512,256,0,1 --shape 256,512,262144,1,0 --shape 512,256,262144,1,0 --shape 4096,4096,4096,0,0"
for cfg in "${@:-default}"; do
  echo "== $cfg"
  env $( [ "$cfg" = default ] || echo $cfg ) python scripts/mb.py gemm $SH 2>&1 | grep -v amdgpu.ids | cut -c1-110
done
