#!/bin/bash
# Full GPU check: parity tests, smoke, bench line (+ full per-leg report), rocprofv3 kernel stats of the bench command, PMC traffic.
#   usage (through gpurun): bash scripts/round_check.sh <tag> [traffic]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
cd $R
mkdir -p gpurun_out/$TAG
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -25) > gpurun_out/$TAG/pytest.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/$TAG/smoke.log
T0=$(date +%s)
(timeout 900 python bench.py --legs-file gpurun_out/$TAG/bench_legs.json 2> /dev/null | tail -1) > gpurun_out/$TAG/bench_line.json
echo "default bench.py wall seconds: $(( $(date +%s) - T0 ))" > gpurun_out/$TAG/bench_time.txt
PROF="--steps 10 --warmup 2 --no-cpu-baseline --trained-epochs 0 --neumf-trained-steps 0 --legs-file /tmp/legs_prof.json"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof -o bench -- python bench.py $PROF > gpurun_out/$TAG/bench_prof.log 2>&1
python scripts/rocpd_summary.py gpurun_out/$TAG/prof/bench_results.db "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --trained-epochs 0 --neumf-trained-steps 0 ($TAG)" > gpurun_out/$TAG/bench_kernel_stats.md 2>&1
rm -rf gpurun_out/$TAG/prof
# the headline leg alone (10M x 1M x 128): the other legs launch the same kernels on other shapes, which would blur the per-kernel averages above
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof -o bench -- python bench.py $PROF --legs bpr,metrics > gpurun_out/$TAG/bench_prof_bpr.log 2>&1
python scripts/rocpd_summary.py gpurun_out/$TAG/prof/bench_results.db "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --trained-epochs 0 --neumf-trained-steps 0 --legs bpr,metrics ($TAG; headline leg only: 10M x 1M x 128 training step + top-k block + metrics)" > gpurun_out/$TAG/bench_kernel_stats_bpr.md 2>&1
rm -rf gpurun_out/$TAG/prof
if [ "$2" = "traffic" ]; then bash scripts/collect_traffic.sh > gpurun_out/$TAG/traffic.log 2>&1; fi
tail -4 gpurun_out/$TAG/pytest.log; tail -1 gpurun_out/$TAG/smoke.log; cat gpurun_out/$TAG/bench_time.txt; wc -c gpurun_out/$TAG/bench_line.json; cut -c1-1800 gpurun_out/$TAG/bench_line.json; tail -30 gpurun_out/$TAG/traffic.log
