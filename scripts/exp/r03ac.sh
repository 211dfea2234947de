#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ac; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_neumf.py tests/test_gpu_nmf_score.py tests/test_gpu_tf_pins.py tests/test_gpu_fullsize_neumf.py -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 600 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 20 > $O/nmf.log 2>&1
tail -14 $O/nmf.log
