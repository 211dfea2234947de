#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_03; cd $R; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest -q -m gpu --timeout 600 tests/test_gpu_bench_contract.py::test_one_rank_sharded_path_prints_the_same_contract tests/test_gpu_bench_contract.py::test_collectives_through_the_c_abi_one_rank tests/test_gpu_nmf_score.py tests/test_gpu_bpr.py tests/test_gpu_dense.py tests/test_gpu_neumf.py 2>&1 | tail -80) > gpurun_out/$T/pytest.log
for v in 4 8; do
  EL_BPR_USER_WAVE_ROWS=1 EL_BPR_USER_PRE=$v timeout 600 python bench.py --legs bpr --no-cpu-baseline --legs-file gpurun_out/$T/legs_wr$v.json 2> /dev/null | tail -1 > /dev/null
done
EL_GEMM_XCD=0 timeout 300 python scripts/mb.py gemm > gpurun_out/$T/gemm_xcd0.txt 2>&1
EL_GEMM_XCD=1 timeout 300 python scripts/mb.py gemm > gpurun_out/$T/gemm_xcd1.txt 2>&1
EL_GEMM_XCD=0 EL_VAE_SIDE=0 timeout 300 python scripts/mb.py vae --iters 20 > gpurun_out/$T/vae_00.txt 2>&1
EL_GEMM_XCD=1 EL_VAE_SIDE=0 timeout 300 python scripts/mb.py vae --iters 20 > gpurun_out/$T/vae_10.txt 2>&1
EL_GEMM_XCD=1 EL_VAE_SIDE=1 timeout 300 python scripts/mb.py vae --iters 20 > gpurun_out/$T/vae_11.txt 2>&1
tail -15 gpurun_out/$T/pytest.log
python - <<PY
import json
for v in (4,8):
    d=json.load(open("gpurun_out/$T/legs_wr%d.json"%v))
    r=d["roofline"]
    print("wave_rows", v, d["value"], d["ms_per_step"], r["frac"], {k:round(x,4) for k,x in r["kernels_ms_per_step"].items()})
PY
paste -d'|' <(cut -c1-95 gpurun_out/$T/gemm_xcd0.txt) <(cut -c30-60 gpurun_out/$T/gemm_xcd1.txt)
for f in vae_00 vae_10 vae_11; do echo == $f; head -8 gpurun_out/$T/$f.txt; tail -1 gpurun_out/$T/$f.txt; done
