#!/bin/bash
# rocprofv3 kernel stats of the default bench command -> gpurun_out/$1/prof (summarise with scripts/rocpd_summary.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-p}
cd $R
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/bench_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/$TAG/prof/bench_results.db | head -24 | cut -c1-150
