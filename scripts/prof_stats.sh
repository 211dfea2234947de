cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
(timeout 300 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -15) > gpurun_out/pytest2.log
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/stats -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
rocprofv3 -L > gpurun_out/prof/counters_list.txt 2>&1
ls -R gpurun_out/prof | head -30
tail -3 gpurun_out/pytest2.log
