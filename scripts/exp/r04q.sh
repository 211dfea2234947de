#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r04z
for v in s4 same4 s14; do
  echo "=== variant '$v'" >> gpurun_out/r04z/log.txt
  export EL_LIB_PATH=$R/elliot_amd/csrc/variants/libelliot_hip_$v.so
  EL_NMF_SCREEN=1 EL_NMF_SCREEN_MAXFRAC=1.0 timeout 300 python scripts/mb.py nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 2 2>&1 | grep -E 'k_nmf_screen' >> gpurun_out/r04z/log.txt
done
cat gpurun_out/r04z/log.txt
