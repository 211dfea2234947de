#!/bin/bash
# round-3 GPU session d: per-item error radii in k_screen_final (trained + untrained regimes), parity of the whole top-k suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --algo screen --iters 10 --train-steps 84 > $O/topk_trained.log 2>&1
python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --algo screen --iters 10 --train-steps 0 > $O/topk_untrained.log 2>&1
python scripts/mb.py topk --users 131072 --items 1000000 --factors 128 --algo screen --iters 3 --train-steps 20 > $O/topk_c4.log 2>&1
python -m pytest tests/test_gpu_topk.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_c5.py tests/test_gpu_pointwise.py tests/test_gpu_cml.py -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log; grep "^\[" $O/topk_*.log
