#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_04; cd $R; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest -q -m gpu --timeout 600 -x tests/test_gpu_dense.py tests/test_gpu_neumf.py tests/test_gpu_topk.py 2>&1 | tail -60) > gpurun_out/$T/pytest.log
EL_GEMM_B3W=0 timeout 300 python scripts/mb.py gemm > gpurun_out/$T/gemm_w0.txt 2>&1
EL_GEMM_B3W=1 timeout 300 python scripts/mb.py gemm > gpurun_out/$T/gemm_w1.txt 2>&1
EL_GEMM_B3W=0 timeout 300 python scripts/mb.py vae --iters 20 > gpurun_out/$T/vae_w0.txt 2>&1
EL_GEMM_B3W=1 timeout 300 python scripts/mb.py vae --iters 20 > gpurun_out/$T/vae_w1.txt 2>&1
EL_GEMM_B3W=0 timeout 300 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 10 --model NeuMF > gpurun_out/$T/nmf_w0.txt 2>&1
EL_GEMM_B3W=1 timeout 300 python scripts/mb.py nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 10 --model NeuMF > gpurun_out/$T/nmf_w1.txt 2>&1
( time (timeout 900 python bench.py --legs-file gpurun_out/$T/bench_legs.json 2> gpurun_out/$T/bench.err | tail -1 > gpurun_out/$T/bench_line.json) ) 2> gpurun_out/$T/bench_time.txt
tail -c 3000 gpurun_out/$T/bench.err > gpurun_out/$T/bench.err.tail; rm -f gpurun_out/$T/bench.err
tail -8 gpurun_out/$T/pytest.log
paste -d'|' <(cut -c1-95 gpurun_out/$T/gemm_w0.txt) <(cut -c30-60 gpurun_out/$T/gemm_w1.txt)
for f in vae_w0 vae_w1 nmf_w0 nmf_w1; do echo == $f; head -9 gpurun_out/$T/$f.txt | tail -7; tail -1 gpurun_out/$T/$f.txt; done
cat gpurun_out/$T/bench_time.txt; wc -c gpurun_out/$T/bench_line.json; cat gpurun_out/$T/bench_line.json; tail -c 600 gpurun_out/$T/bench.err.tail
