#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1 2 3 4 5; do
  echo "== variant $v"
  EL_TOPK_VARIANT=$v python scripts/mb.py topk --users 131072 --iters 4 --variant $v 2>&1 | grep k_score
  EL_TOPK_VARIANT=$v timeout 120 python -m pytest tests/test_gpu_topk.py -m gpu -q -k "128 and mfma" 2>&1 | tail -1
done
