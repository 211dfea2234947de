"""Helpers for the -m gpu parity tests (diagnostics are dumped under gpurun_out/diag on failure)."""
import os

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(REPO, "gpurun_out", "diag")


def dump(name, **arrays):
    try:
        os.makedirs(DIAG, exist_ok=True)
        small = {k: np.asarray(v)[:4096] if np.asarray(v).ndim == 1 else np.asarray(v)[:256] for k, v in arrays.items()}
        np.savez_compressed(os.path.join(DIAG, name + ".npz"), **small)
    except Exception as ex:  # diagnostics must never mask the real failure
        print("diag dump failed:", ex)


def cpu(t):
    return t.detach().cpu().numpy()


def dev_csr(ops, indptr, indices, n_cols, device):
    return ops.DeviceCSR(indptr, indices, n_cols, device)


def random_excl(rs, U, I, lo=0, hi=12):
    rows = [np.sort(rs.choice(I, rs.randint(lo, hi + 1), replace=False)) for _ in range(U)]
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    indices = (np.concatenate(rows) if indptr[-1] else np.zeros(0)).astype(np.int32)
    return indptr, indices


def assert_topk_equal(name, got_idx, got_val, exp_idx, exp_val, bit_exact=True, **ctx_arrays):
    gi, gv = np.asarray(got_idx), np.asarray(got_val)
    ok_i = np.array_equal(gi, exp_idx)
    if bit_exact:
        ok_v = np.array_equal(gv.view(np.uint32) if gv.dtype == np.float32 else gv.view(np.uint64),
                              exp_val.view(np.uint32) if exp_val.dtype == np.float32 else exp_val.view(np.uint64))
    else:
        ok_v = np.allclose(gv, exp_val, rtol=1e-12, atol=1e-12)
    if not (ok_i and ok_v):
        bad_rows = np.where((gi != exp_idx).any(1) | (gv != exp_val).any(1))[0]
        dump(name, got_idx=gi, got_val=gv, exp_idx=exp_idx, exp_val=exp_val, bad_rows=bad_rows, **ctx_arrays)
        r = int(bad_rows[0]) if len(bad_rows) else 0
        raise AssertionError(
            f"{name}: {len(bad_rows)}/{gi.shape[0]} rows differ (idx_ok={ok_i}, val_ok={ok_v}); first bad row {r}:\n"
            f"  got idx {gi[r].tolist()}\n  exp idx {exp_idx[r].tolist()}\n  got val {gv[r].tolist()}\n  exp val {exp_val[r].tolist()}")
