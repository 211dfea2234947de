#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r04m
for v in "" nopi nostage ks4 both; do
  echo "=== variant '$v'" >> gpurun_out/r04m/log.txt
  if [ -n "$v" ]; then export EL_LIB_PATH=$R/elliot_amd/csrc/variants/libelliot_hip_$v.so; fi
  EL_NMF_SCREEN=1 EL_NMF_SCREEN_WAVES=8 EL_NMF_SCREEN_MAXFRAC=1.0 timeout 300 python scripts/mb.py nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 3 2>&1 | grep -E 'wall|k_nmf_screen|k_nmf_score|screen:' >> gpurun_out/r04m/log.txt
done
cat gpurun_out/r04m/log.txt
