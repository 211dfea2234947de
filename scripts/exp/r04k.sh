#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r04l
(timeout 900 python -m pytest tests/test_gpu_nmf_score.py -m gpu -q --timeout 600 2>&1 | tail -40) > gpurun_out/r04l/pytest.log
cat gpurun_out/r04l/pytest.log
for cfg in "1 4 1.0" "1 8 1.0"; do set -- $cfg
  echo "=== EL_NMF_SCREEN=$1 waves=$2 maxfrac=$3" >> gpurun_out/r04l/log.txt
  EL_NMF_SCREEN=$1 EL_NMF_SCREEN_WAVES=$2 EL_NMF_SCREEN_MAXFRAC=$3 timeout 300 python scripts/mb.py nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 3 2>&1 | grep -v amdgpu.ids | tail -22 >> gpurun_out/r04l/log.txt
done
cat gpurun_out/r04l/log.txt
