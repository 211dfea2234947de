#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
python -m pytest tests/test_gpu_bpr.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_c5.py tests/test_gpu_plugin.py tests/test_gpu_comm.py tests/test_gpu_bench_contract.py -x -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for rpg in 2 4 8; do EL_FUSED_RPG=$rpg python bench.py --legs bpr --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('rpg $rpg', d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'].get('k_bpr_user_adam'))"; done > $O/rpg.log 2>&1
cat $O/rpg.log
