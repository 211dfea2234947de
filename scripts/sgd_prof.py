import cProfile, pstats, os, sys, tempfile, time
from types import SimpleNamespace
import numpy as np, torch
sys.path.insert(0, '.')
from elliot_amd.dataset.dataset import DataSet, default_config
from elliot_amd.recommender import BPRMF
from elliot_amd.synthetic import zipf_csr
U, I = 6040, 3667
indptr, indices = zipf_csr(U, I, 4.45, 1.0, 16, 1800, 0.8, 0)
rs = np.random.RandomState(1)
users = np.repeat(np.arange(U), np.diff(indptr)); ratings = rs.randint(1, 6, indices.shape[0]).astype(float)
flag = rs.rand(indices.shape[0]) < 0.2
out = tempfile.mkdtemp()
cfg = default_config(top_k=10, cutoffs=[10], simple_metrics=["nDCG"], out_dir=out)
for p in (cfg.path_output_rec_result, cfg.path_output_rec_weight): os.makedirs(p, exist_ok=True)
data = DataSet(cfg, (users[~flag], indices[~flag], ratings[~flag]), (users[flag], indices[flag], ratings[flag]))
params = SimpleNamespace(meta=SimpleNamespace(save_recs=False, verbose=False), epochs=1, seed=42, factors=64, lr=0.05)
m = BPRMF(data=data, config=cfg, params=params)
n = data.transactions
step = lambda: [m._model.train_step(t) for t in m._sampler.step(n, n)]
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print("epoch %.1f ms" % ((time.perf_counter() - t0) * 1e3))
m._ctx.timing(True); step(); torch.cuda.synchronize()
rep = m._ctx.timing_report(); m._ctx.timing(False)
print({k: (c, round(ms, 2)) for k, (c, ms) in rep.items()})
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
