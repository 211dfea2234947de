#!/bin/bash
# deferred decay of the NeuMF embedding tables: parity tests, then the step with the feature off / on at the bench shape
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_neumf.py tests/test_gpu_nmf_score.py tests/test_gpu_tf_pins.py -x -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for d in 0 1; do
  EL_NMF_DEFERRED=$d timeout 600 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 20 > $O/nmf_def$d.log 2>&1
  echo "== EL_NMF_DEFERRED=$d"; tail -16 $O/nmf_def$d.log
done
