"""Plugin-level timings at the ML-1M shape (BASELINE configs[0]; synthetic S-ML1M of SURVEY 8d: 6040 users x 3667 items,
~0.8 M train interactions, d = 64): one epoch of BPRMF (fp64 per-sample SGD, level-scheduled) and of BPRMF_batch (TF-dense
Adam), plus one evaluation (top-10 + metrics), driven through the plugin classes exactly as Elliot's ModelCoordinator does."""
import os
import sys
import tempfile
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elliot_amd.dataset.dataset import DataSet, default_config          # noqa: E402
from elliot_amd.recommender import BPRMF, BPRMF_batch                    # noqa: E402
from elliot_amd.synthetic import zipf_csr                                # noqa: E402


def main():
    U, I = 6040, 3667
    indptr, indices = zipf_csr(U, I, 4.45, 1.0, 16, 1800, 0.8, 0)
    rs = np.random.RandomState(1)
    users = np.repeat(np.arange(U), np.diff(indptr))
    ratings = rs.randint(1, 6, indices.shape[0]).astype(float)
    flag = rs.rand(indices.shape[0]) < 0.2                               # ~80/20 split
    out = tempfile.mkdtemp()
    cfg = default_config(top_k=10, cutoffs=[10], simple_metrics=["nDCG", "Recall"], out_dir=out)
    for p in (cfg.path_output_rec_result, cfg.path_output_rec_weight):
        os.makedirs(p, exist_ok=True)
    data = DataSet(cfg, (users[~flag], indices[~flag], ratings[~flag]), (users[flag], indices[flag], ratings[flag]))
    print(f"S-ML1M: {data.num_users} users, {data.num_items} items, {data.transactions} train interactions")

    def run(cls, **hp):
        params = SimpleNamespace(meta=SimpleNamespace(save_recs=False, verbose=False), epochs=2, seed=42, **hp)
        m = cls(data=data, config=cfg, params=params)
        torch.cuda.synchronize()
        ev = m.evaluate
        spent = {"eval": 0.0}

        def timed_eval(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev(*a, **k)
            torch.cuda.synchronize()
            spent["eval"] += time.perf_counter() - t0
        m.evaluate = timed_eval
        t0 = time.perf_counter()
        m.train()
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        tr = (tot - spent["eval"]) / 2
        print(f"{m.name[:40]:40s} train {tr * 1e3:8.1f} ms/epoch = {data.transactions / tr / 1e6:7.2f} M triplets/s | "
              f"evaluate (top-10 of {U} users + nDCG/Recall) {spent['eval'] / 2 * 1e3:6.1f} ms | nDCG@10 "
              f"{m.get_results()[10]['test_results']['nDCG']:.4f}")

    run(BPRMF, factors=64, lr=0.05)
    run(BPRMF_batch, factors=64, lr=0.001, l_w=0.1, l_b=0.001, batch_size=512)
    run(BPRMF_batch, factors=64, lr=0.001, l_w=0.1, l_b=0.001, batch_size=65536)


if __name__ == "__main__":
    main()
