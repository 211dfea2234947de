#!/bin/bash
# Per-kernel times (hipEvents around every launch, el_timing_enable) of the legs bench.py does not time: the sibling models
# and the reference-default epoch loop.  Output -> gpurun_out/secondary/legs.txt (copied to profiles/ by hand).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/secondary
O=gpurun_out/secondary/legs.txt
: > $O
run() { echo "## $*" >> $O; timeout 300 "$@" 2>&1 | grep -v "amdgpu.ids" >> $O; echo >> $O; }
for m in MF FunkSVD PMF LogisticMF; do run python scripts/mb.py pwmf --users 1000000 --items 100000 --factors 128 --iters 5 --model $m; done
run python scripts/cml_bench.py
run python scripts/mb.py nmf --users 1000000 --items 100000 --factors 128 --batch 262144 --iters 4 --model NeuMF
run python scripts/mb.py vae --iters 6
run python scripts/loop_bench.py 512
run python scripts/ml1m_epoch.py
tail -5 $O
