#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r04j
(timeout 900 python -m pytest tests/test_gpu_nmf_score.py -m gpu -q --timeout 600 -x 2>&1 | tail -40) > gpurun_out/r04j/pytest.log
cat gpurun_out/r04j/pytest.log
for v in 0 1; do for wv in 4 8; do
  echo "=== EL_NMF_SCREEN=$v waves=$wv" >> gpurun_out/r04j/log.txt
  EL_NMF_SCREEN=$v EL_NMF_SCREEN_WAVES=$wv timeout 300 python scripts/mb.py nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 3 2>&1 | grep -v amdgpu.ids | tail -12 >> gpurun_out/r04j/log.txt
done; done
cat gpurun_out/r04j/log.txt
