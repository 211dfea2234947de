#!/usr/bin/env python
"""Host data plane at S-1M-like sizes: interaction arrays -> split flags -> DataSet (id maps, CSR) -> held-out CSR.
CPU only.  `--ref-users N` also times the reference-style route (per-user Python shuffle + dict of dicts) on the first N users
for the comparison line."""
import argparse
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elliot_amd.dataset import dataset as D       # noqa: E402
from elliot_amd.synthetic import zipf_csr        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=1_000_000)
ap.add_argument("--items", type=int, default=100_000)
ap.add_argument("--ref-users", type=int, default=20000)
a = ap.parse_args()

t = time.time()
parts = [zipf_csr(min(250_000, a.users - s), a.items, seed=1234 + s) for s in range(0, a.users, 250_000)]
deg = np.concatenate([np.diff(p[0]) for p in parts])
items = np.concatenate([p[1] for p in parts]).astype(np.int64)
users = np.repeat(np.arange(a.users, dtype=np.int64), deg)
rs = np.random.RandomState(0)
perm = rs.permutation(users.shape[0])                 # a ratings file is not sorted by user
users, items = users[perm], items[perm]
ratings = rs.randint(1, 6, size=users.shape[0]).astype(np.float64)
print(f"generated {users.shape[0]:,} interactions of {a.users:,} users x {a.items:,} items in {time.time() - t:.1f} s")

times = {}
t = time.time()
flags = D.random_subsampling(users, 0.2, 42)
times["split flags (el_host_split_flags)"] = time.time() - t
tr, te = flags == 0, flags == 1
t = time.time()
ds = D.DataSet(D.default_config(), (users[tr], items[tr], ratings[tr]), (users[te], items[te], ratings[te]))
times["DataSet: id maps (el_host_pyset_order) + train CSR"] = time.time() - t
t = time.time()
ip, cols, vals = ds.split_csr(False)
times["held-out CSR (split_csr)"] = time.time() - t
for k, v in times.items():
    print(f"  {k:55s} {v:8.2f} s")
print(f"  total {sum(times.values()):.2f} s for {users.shape[0]:,} rows = {users.shape[0] / sum(times.values()) / 1e6:.2f} M rows/s; "
      f"train nnz {ds.transactions:,}, test nnz {cols.shape[0]:,}")

if a.ref_users:
    sel = users < a.ref_users
    u, i, r = users[sel], items[sel], ratings[sel]
    t = time.time()
    rs2 = np.random.RandomState(42)
    fl = np.zeros(u.shape[0], dtype=np.int8)
    order = np.argsort(u, kind="stable")
    su = u[order]
    bounds = np.flatnonzero(np.concatenate([[True], su[1:] != su[:-1], [True]]))
    for x, y in zip(bounds[:-1], bounds[1:]):
        n = y - x
        lst = [0] * int(math.floor(n * 0.8)) + [1] * (n - int(math.floor(n * 0.8)))
        rs2.shuffle(lst)
        fl[order[x:y]] = lst
    t_split = time.time() - t
    t = time.time()
    d = {}
    for uu, ii, rr in zip(u[fl == 0].tolist(), i[fl == 0].tolist(), r[fl == 0].tolist()):
        d.setdefault(uu, {})[ii] = rr
    its = list({k for x in d.values() for k in x.keys()})
    t_dict = time.time() - t
    n = u.shape[0]
    print(f"  reference-style route on the first {a.ref_users:,} users ({n:,} rows): per-user shuffle loop {t_split:.2f} s, dict of dicts + "
          f"item set {t_dict:.2f} s = {n / (t_split + t_dict) / 1e6:.2f} M rows/s (and the reference's dataframe_to_dict filters the whole "
          f"frame once per user: O(U*T))")
