#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
for gap in tuned 0 0.5 1 1.5 2 2.5 3 3.5 4 4.5 5 5.5; do
  if [ $gap = tuned ]; then unset EL_LAYOUT_GAP_MIB; else export EL_LAYOUT_GAP_MIB=$gap; fi
  EL_TUNE_DEBUG=1 python bench.py --legs bpr --no-cpu-baseline --repeats 2 2> $O/err_$gap.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('gap $gap', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['roofline']['kernels_ms_per_step'].get('k_bpr_user_adam'))"
done > $O/gaps.log 2>&1
cat $O/gaps.log; grep tune_table $O/err_tuned.log | head -3
