#!/bin/bash
# Development check: bench.py's N = 2 control flow (sharding, collectives' call pattern, barrier, max over ranks, one JSON
# line from rank 0) as two processes on a ONE-GPU box -- both ranks on cuda:0, gloo instead of RCCL.  Not a measurement.
cd ${GRAFT_REPO_ROOT:-.}
export EL_BENCH_SHARED_GPU=1
for extra in "" "--shard item --exchange dense --topk-shard item" "--shard item --exchange rows"; do
  echo "== bench.py --gpus 2 $extra"
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline $extra 2>&1 | grep -v "amdgpu.ids\|^W0\|^\*\*\*\*\|OMP_NUM_THREADS" | tail -4 | cut -c1-900
done
