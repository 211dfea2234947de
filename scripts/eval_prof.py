import cProfile, pstats, os, sys, tempfile, time
from types import SimpleNamespace
import numpy as np, torch
sys.path.insert(0, '.')
from elliot_amd.dataset.dataset import DataSet, default_config
from elliot_amd.recommender import BPRMF_batch
from elliot_amd.synthetic import zipf_csr
U, I = 6040, 3667
indptr, indices = zipf_csr(U, I, 4.45, 1.0, 16, 1800, 0.8, 0)
rs = np.random.RandomState(1)
users = np.repeat(np.arange(U), np.diff(indptr)); ratings = rs.randint(1, 6, indices.shape[0]).astype(float)
flag = rs.rand(indices.shape[0]) < 0.2
out = tempfile.mkdtemp()
cfg = default_config(top_k=10, cutoffs=[10], simple_metrics=["nDCG", "Recall"], out_dir=out)
for p in (cfg.path_output_rec_result, cfg.path_output_rec_weight): os.makedirs(p, exist_ok=True)
data = DataSet(cfg, (users[~flag], indices[~flag], ratings[~flag]), (users[flag], indices[flag], ratings[flag]))
params = SimpleNamespace(meta=SimpleNamespace(save_recs=False, verbose=False), epochs=1, seed=42, factors=64, lr=0.001, l_w=0.1, l_b=0.001, batch_size=512)
m = BPRMF_batch(data=data, config=cfg, params=params)
m.train()
torch.cuda.synchronize()
for _ in range(2): m.evaluate(0, 0.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): m.evaluate(0, 0.0)
torch.cuda.synchronize()
print("evaluate: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): m.evaluate(0, 0.0)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
