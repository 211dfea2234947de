"""Device-side accuracy metrics (el_rec_metrics) against the reference Evaluator's golden values and the
host evaluator mirror (itself pinned to the reference Evaluator, tests/test_host_plugin.py)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from elliot_amd import ops
from elliot_amd.dataset.dataset import default_config
from elliot_amd.evaluation.evaluator import Evaluator

pytestmark = pytest.mark.gpu
NAMES = list(ops.METRIC_NAMES)


@pytest.fixture(scope="module")
def ctx():
    return ops.get_context(0)


def sorted_rows(ip, items, ratings):
    items, ratings = items.copy(), ratings.copy()
    for u in range(len(ip) - 1):
        o = np.argsort(items[ip[u]:ip[u + 1]], kind="stable")
        items[ip[u]:ip[u + 1]] = items[ip[u]:ip[u + 1]][o]
        ratings[ip[u]:ip[u + 1]] = ratings[ip[u]:ip[u + 1]][o]
    return items, ratings


def device_means(ctx, recs_idx, ip, items, ratings, thr, cutoff, per_user=False):
    test = ops.DeviceTestSet(ip, items, ratings, ctx.device)
    rec = torch.from_numpy(np.ascontiguousarray(recs_idx.astype(np.int32))).to(ctx.device)
    out = ops.rec_metrics(ctx, rec, test, thr, cutoff, per_user=per_user)
    torch.cuda.synchronize()
    sums = (out[0] if per_user else out).cpu().numpy()
    means = {n: sums[m] / sums[7] for m, n in enumerate(NAMES)} if sums[7] else {}
    return (means, sums, out[1].cpu().numpy()) if per_user else (means, sums)


def host_means(recs_idx, ip, items, ratings, thr, cutoff):
    U, k = recs_idx.shape
    test = {u: {int(i): float(r) for i, r in zip(items[ip[u]:ip[u + 1]], ratings[ip[u]:ip[u + 1]])} for u in range(U)}
    test = {u: t for u, t in test.items() if t}
    cfg = default_config(top_k=k, cutoffs=[cutoff], simple_metrics=NAMES, relevance_threshold=thr)
    data = SimpleNamespace(config=cfg, get_test=lambda: test, get_validation=lambda: None)
    ev = Evaluator(data, SimpleNamespace())
    recs = {u: [(int(it), 0.0) for it in recs_idx[u]] for u in range(U)}      # -1 = empty slot: keeps its rank, never hits
    return ev.eval((recs, recs))[cutoff]["test_results"]


def test_metrics_match_reference_evaluator_golden(ctx, golden):
    g = golden("metrics_ref.npz")
    ip, k = g["test_indptr"], int(g["k"])
    items, ratings = sorted_rows(ip, g["test_items"], g["test_ratings"])
    ref = dict(zip(g["names"].tolist(), g["values"].tolist()))
    for cutoff in (k, 5):
        means, sums = device_means(ctx, g["recs"], ip, items, ratings, 0.0, cutoff)
        for n in NAMES:
            assert abs(means[n] - ref[f"{n}@{cutoff}"]) < 1e-12, (n, cutoff, means[n], ref[f"{n}@{cutoff}"])
        assert sums[7] == g["recs"].shape[0]


@pytest.mark.parametrize("cutoff,thr,frac_ratings", [(10, 0.0, False), (50, 3.0, False), (100, 2.5, True), (7, 1.0, True)])
def test_metrics_match_host_evaluator_random(ctx, cutoff, thr, frac_ratings):
    rs = np.random.RandomState(cutoff)
    U, I, k = 700, 4000, max(cutoff, 20)
    deg = rs.randint(0, 30, size=U)
    deg[5] = 1500                                   # > LDS gain buffer: chunked selection of the IDCG gains
    deg[6] = 0
    ip = np.zeros(U + 1, np.int64)
    ip[1:] = np.cumsum(deg)
    items = np.concatenate([np.sort(rs.choice(I, size=d, replace=False)) for d in deg]).astype(np.int32)
    ratings = (rs.uniform(0.5, 5.0, size=ip[-1]) if frac_ratings else rs.randint(1, 6, size=ip[-1])).astype(np.float32)
    recs = np.stack([rs.choice(I, size=k, replace=False) for _ in range(U)]).astype(np.int32)
    for u in range(0, U, 3):                        # plant hits
        if deg[u]:
            row = items[ip[u]:ip[u + 1]]
            pos = rs.choice(k, size=min(3, len(row), k), replace=False)
            pick = rs.choice(row, size=len(pos), replace=False)
            keep = ~np.isin(recs[u], pick)
            recs[u] = np.where(keep, recs[u], (recs[u] + I) % (2 * I) + I)   # drop accidental duplicates of the planted ids
            recs[u][pos] = pick
    recs[recs >= I] = -1                            # some empty slots too
    means, sums, rows = device_means(ctx, recs, ip, items, ratings.astype(np.float64).astype(np.float32), thr, cutoff, per_user=True)
    ref = host_means(recs, ip, items, ratings.astype(np.float64), thr, cutoff)
    for n in NAMES:
        assert abs(means[n] - ref[n]) < 1e-11, (n, means[n], ref[n])
    nrel = np.array([(ratings[ip[u]:ip[u + 1]] >= thr).sum() for u in range(U)])
    assert sums[7] == (nrel > 0).sum()
    assert np.array_equal(rows[:, 7] != 0, nrel > 0)
    # the fixed-shape reduction is run-to-run identical
    _, sums2 = device_means(ctx, recs, ip, items, ratings, thr, cutoff)
    assert np.array_equal(sums, sums2)


def test_metrics_user_subrange_accumulates(ctx):
    rs = np.random.RandomState(3)
    U, I, k = 300, 500, 10
    deg = rs.randint(1, 8, size=U)
    ip = np.zeros(U + 1, np.int64)
    ip[1:] = np.cumsum(deg)
    items = np.concatenate([np.sort(rs.choice(I, size=d, replace=False)) for d in deg]).astype(np.int32)
    recs = np.stack([rs.choice(I, size=k, replace=False) for _ in range(U)]).astype(np.int32)
    test = ops.DeviceTestSet(ip, items, None, ctx.device)
    full = ops.rec_metrics(ctx, torch.from_numpy(recs).to(ctx.device), test, 0.0, k)
    acc = torch.zeros(8, dtype=torch.float64, device=ctx.device)
    for s in range(0, U, 128):
        e = min(U, s + 128)
        ops.rec_metrics(ctx, torch.from_numpy(recs[s:e].copy()).to(ctx.device), test, 0.0, k, u_start=s, sums=acc)
    torch.cuda.synchronize()
    assert np.allclose(full.cpu().numpy(), acc.cpu().numpy(), rtol=1e-13, atol=0)
