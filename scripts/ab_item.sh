#!/bin/bash
# A/B of the item side of the BPR step: two-pass form (EL_FUSED_ITEM=0) vs the fused item segments + Adam, every-step replay vs deferred,
# at the three bench shapes.   usage (through gpurun): bash scripts/ab_item.sh <tag> [shapes...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-ab_item}; shift
cd $R
mkdir -p gpurun_out/$TAG
SHAPES=${@:-c2 c4 c5}
for sh in $SHAPES; do
  case $sh in
    c2) A="--users 1000000 --items 100000 --factors 128" ;;
    c4) A="--users 10000000 --items 1000000 --factors 128" ;;
    c5) A="--users 6250000 --items 5000000 --factors 256" ;;
  esac
  for cfg in "EL_FUSED_ITEM=0" "EL_FUSED_ITEM=1 EL_BPR_ITEM_DEFERRED=0" "EL_FUSED_ITEM=1 EL_BPR_ITEM_DEFERRED=1" \
             "EL_FUSED_ITEM=1 EL_BPR_ITEM_DEFERRED=1 EL_ICHUNK=128" "EL_FUSED_ITEM=1 EL_BPR_ITEM_DEFERRED=1 EL_ICHUNK=512"; do
    echo "=== $sh $cfg" >> gpurun_out/$TAG/ab.log
    env $cfg timeout 300 python scripts/mb.py train $A --iters 10 --algo auto >> gpurun_out/$TAG/ab.log 2>&1
  done
done
cat gpurun_out/$TAG/ab.log
