"""Synthetic interaction data (S-ML1M, S-1M, ...) used by bench.py and the tests.

There is no network and MovieLens is not on disk, so every workload is generated: per-user train
degree ~ clipped log-normal, item popularity ~ Zipf, emitted directly as CSR (int64 indptr, int32
indices, columns sorted ascending inside each row).
"""
import numpy as np


def zipf_csr(n_users, n_items, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, zipf_a=1.0, seed=1234):
    """Returns (indptr int64 [U+1], indices int32 [T]) with unique, sorted columns per row."""
    rs = np.random.RandomState(seed)
    dmax = min(dmax, n_items - 1)
    dmin = min(dmin, dmax)
    deg = np.clip(rs.lognormal(mean_log, sigma_log, size=n_users), dmin, dmax).astype(np.int64)
    # popularity CDF
    w = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), zipf_a)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    perm = rs.permutation(n_items)  # popular items are not the low ids
    total = int(deg.sum())
    # oversample, then unique per row
    over = (deg * 1.3).astype(np.int64) + 8
    starts = np.concatenate([[0], np.cumsum(over)])
    draws = perm[np.searchsorted(cdf, rs.random_sample(int(starts[-1])))]
    rows = np.repeat(np.arange(n_users, dtype=np.int64), over)
    key = rows * n_items + draws
    key = np.unique(key)  # sorted by (row, col), duplicates dropped
    r = key // n_items
    c = (key % n_items).astype(np.int32)
    counts = np.bincount(r, minlength=n_users)
    # trim rows to their target degree (keep a random subset -> keep first deg after a per-row shuffle is
    # unnecessary for a benchmark; keep the lowest `deg` columns of the sampled set)
    keep_counts = np.minimum(counts, deg)
    row_start = np.concatenate([[0], np.cumsum(counts)])[:-1]
    pos_in_row = np.arange(key.shape[0]) - np.repeat(row_start, counts)
    keep = pos_in_row < np.repeat(keep_counts, counts)
    c = c[keep]
    indptr = np.concatenate([[0], np.cumsum(keep_counts)]).astype(np.int64)
    # every user needs at least one positive
    assert (keep_counts > 0).all() and total > 0
    return indptr, c


def glorot_uniform(rows, cols, seed):
    """tf.initializers.GlorotUniform distribution (BPRMF_batch_model.py:39-42); the TF bit stream itself is
    not reproducible without TF, only the distribution is."""
    rs = np.random.RandomState(seed)
    lim = np.sqrt(6.0 / (rows + cols))
    return rs.uniform(-lim, lim, size=(rows, cols)).astype(np.float32)


def small_dataset(n_users=200, n_items=150, seed=0, mean_log=2.3, sigma_log=0.7, dmin=2, dmax=60):
    """Tiny interaction set for parity tests: returns (indptr, indices, i_train_dict) where i_train_dict is the
    reference's {user: {item: rating}} structure (dataset.py:216-217) with private ids.  The number of items is
    ``indices.max() + 1`` (<= n_items): only items that occur in train exist, as in the reference."""
    indptr, indices = zipf_csr(n_users, n_items, mean_log, sigma_log, dmin, dmax, 0.8, seed)
    # the reference defines the catalogue as the items present in train (dataset.py:202): make ids dense
    used = np.unique(indices)
    remap = np.full(n_items, -1, dtype=np.int64)
    remap[used] = np.arange(used.shape[0])
    indices = remap[indices].astype(np.int32)
    rs = np.random.RandomState(seed + 1)
    d = {}
    for u in range(n_users):
        cols = indices[indptr[u]:indptr[u + 1]]
        cols = cols[rs.permutation(cols.shape[0])]  # insertion order as a TSV would give it
        d[u] = {int(c): float(rs.randint(1, 6)) for c in cols}
    return indptr, indices, d


def zipf_csr_device(n_users, n_items, device, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, zipf_a=1.0, seed=1234):
    """Same construction as zipf_csr, generated on the GPU with torch (data plumbing only: 1e6 x 1e5 with
    ~8e7 interactions takes < 2 s there vs ~1 min in NumPy).  Returns torch tensors (int64 indptr, int32 indices)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    dmax = min(dmax, n_items - 1)
    dmin = min(dmin, dmax)
    deg = torch.exp(torch.randn(n_users, generator=g, device=device, dtype=torch.float64) * sigma_log + mean_log)
    deg = deg.clamp(dmin, dmax).to(torch.int64)
    w = 1.0 / torch.arange(1, n_items + 1, device=device, dtype=torch.float64).pow(zipf_a)
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    perm = torch.randperm(n_items, generator=g, device=device)
    over = (deg.to(torch.float64) * 1.3).to(torch.int64) + 8
    total = int(over.sum().item())
    rows = torch.repeat_interleave(torch.arange(n_users, device=device, dtype=torch.int64), over)
    keys = []
    chunk = 1 << 26
    for s in range(0, total, chunk):
        e = min(s + chunk, total)
        r = torch.rand(e - s, generator=g, device=device, dtype=torch.float64)
        d = perm[torch.searchsorted(cdf, r).clamp_(max=n_items - 1)]
        keys.append(rows[s:e] * n_items + d)
    key = torch.unique(torch.cat(keys))
    del keys, rows
    r = key // n_items
    c = (key % n_items).to(torch.int32)
    counts = torch.bincount(r, minlength=n_users)
    keep_counts = torch.minimum(counts, deg)
    row_start = torch.cumsum(counts, 0) - counts
    pos_in_row = torch.arange(key.shape[0], device=device) - torch.repeat_interleave(row_start, counts)
    keep = pos_in_row < torch.repeat_interleave(keep_counts, counts)
    c = c[keep].contiguous()
    indptr = torch.zeros(n_users + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(keep_counts, 0)
    return indptr, c
