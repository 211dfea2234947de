#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05w
for v in ub2 ub8; do
  echo "=== $v" >> gpurun_out/r05w/log.txt
  EL_LIB_PATH=$R/elliot_amd/csrc/variants/libelliot_hip_$v.so EL_NMF_SCREEN=1 timeout 300 python scripts/mb.py nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 3 2>&1 | grep -E 'wall|k_nmf_screen|screen:' >> gpurun_out/r05w/log.txt
done
cat gpurun_out/r05w/log.txt
