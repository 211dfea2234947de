#!/bin/bash
# round-3 GPU session c: occupancy variants of k_screen_thr / k_screen_final, stride / kA sweep in the TRAINED regime, flagged-user
# counts at the c4 / c5 catalogues, NeuMF step with the tuned table placement, c5 top-k after the k_list_scores change
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
V=elliot_amd/csrc/variants
for lib in "" $V/libelliot_hip_wpe6.so $V/libelliot_hip_wpe7.so $V/libelliot_hip_wpe8.so; do
  echo "=== lib=${lib:-default}" >> $O/topk_variants.log
  EL_LIB_PATH=$lib python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --algo screen --iters 10 --train-steps 84 >> $O/topk_variants.log 2>&1
done
python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --algo screen --iters 10 --train-steps 84 --sweep 8:5,8:7,4:7,4:8,4:9,2:10,2:12 > $O/topk_sweep.log 2>&1
python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --algo screen --iters 10 --train-steps 0 --sweep 8:5,8:7,4:7,4:8,4:9 > $O/topk_sweep_untrained.log 2>&1
EL_SCREEN_PROF=1 python scripts/mb.py topk --users 131072 --items 1000000 --factors 128 --algo screen --iters 2 --train-steps 20 > $O/c4_flags.log 2>&1
python -m pytest tests/test_gpu_fullsize_c5.py tests/test_gpu_topk.py -x -q > $O/pytest_topk.log 2>&1
python bench.py --legs bpr,c5,neumf --no-cpu-baseline --repeats 2 > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest_topk.log
