"""Device-resident candidate masks in CSR form.

The reference materialises dense bool [U, I] masks (dataset.py:230-245): 22 MB at ML-1M, 100 GB at
1M x 100K.  The kernels take the equivalent CSR instead:
  train  : the train interactions  -> mask = NOT in row      (allunrated_mask, dataset.py:245)
  val/test: candidate rows          -> mask = in row           (val_mask / test_mask, dataset.py:228-243)
Built once per DataSet object and cached on it.
"""
from types import SimpleNamespace

import numpy as np
import scipy.sparse as sp

from .. import ops


def _csr_from_dense_mask(mask):
    m = sp.csr_matrix(np.asarray(mask, dtype=bool))
    m.sort_indices()
    return m


def device_masks(data, ctx):
    cache = getattr(data, "_elliot_amd_masks", None)
    if cache is not None and cache.device == ctx.device:
        return cache
    out = SimpleNamespace(device=ctx.device, train=None, val=None, test=None)
    train = data.sp_i_train.tocsr()
    train.sort_indices()
    out.train = ops.DeviceCSR(train.indptr, train.indices, train.shape[1], ctx.device)
    for name in ("val", "test"):
        if hasattr(data, f"{name}_cand_csr"):            # scale path: CSR given directly
            ip, ix = getattr(data, f"{name}_cand_csr")
            setattr(out, name, ops.DeviceCSR(ip, ix, train.shape[1], ctx.device))
        elif hasattr(data, f"{name}_mask"):              # Elliot DataSet: dense bool mask
            m = _csr_from_dense_mask(getattr(data, f"{name}_mask"))
            setattr(out, name, ops.DeviceCSR(m.indptr, m.indices, train.shape[1], ctx.device))
    try:
        data._elliot_amd_masks = out
    except Exception:
        pass
    return out
