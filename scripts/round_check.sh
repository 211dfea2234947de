#!/bin/bash
# Full GPU check: parity tests, smoke, bench line, rocprofv3 kernel stats of the bench command.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
cd $R
mkdir -p gpurun_out/$TAG
(timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -15) > gpurun_out/$TAG/pytest.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/$TAG/smoke.log
(timeout 400 python bench.py 2>&1 | tail -2) > gpurun_out/$TAG/bench.log
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/bench_prof.log 2>&1
tail -3 gpurun_out/$TAG/pytest.log; tail -1 gpurun_out/$TAG/smoke.log; tail -1 gpurun_out/$TAG/bench.log | cut -c1-1500
