#!/bin/bash
# Whole GPU test suite (no -x: every failure in one call) + smoke + one default bench line.   usage: bash scripts/gpu_suite.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-suite}; shift
cd $R
mkdir -p gpurun_out/$TAG
free -g > gpurun_out/$TAG/host.txt; nproc >> gpurun_out/$TAG/host.txt
(timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 --durations=15 2>&1 | tail -120) > gpurun_out/$TAG/pytest.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/$TAG/smoke.log
( time (timeout 900 python bench.py --legs-file gpurun_out/$TAG/bench_legs.json "$@" 2> gpurun_out/$TAG/bench.err | tail -1 > gpurun_out/$TAG/bench_line.json) ) 2> gpurun_out/$TAG/bench_time.txt
# (bench.err carries the full report too: keep its tail only)
tail -c 4000 gpurun_out/$TAG/bench.err > gpurun_out/$TAG/bench.err.tail; rm -f gpurun_out/$TAG/bench.err
tail -40 gpurun_out/$TAG/pytest.log; tail -2 gpurun_out/$TAG/smoke.log; cat gpurun_out/$TAG/bench_time.txt; wc -c gpurun_out/$TAG/bench_line.json; cut -c1-1500 gpurun_out/$TAG/bench_line.json; cat gpurun_out/$TAG/host.txt
