#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
tail -6 $O/pytest.log
timeout 600 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 20 > $O/nmf.log 2>&1
tail -14 $O/nmf.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
