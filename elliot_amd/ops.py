"""Tensor-level wrappers over the C ABI.

torch tensors are used ONLY as device-buffer holders (allocation, streams, H2D/D2H copies);
all arithmetic happens inside libelliot_hip.so.  Every function raises if the library or a
GPU is missing -- there is no CPU path here.
"""
import ctypes as C
import os
import math

import numpy as np
import torch

from . import _lib
from ._lib import (EL_OPT_ADAM_LAZY, EL_OPT_ADAM_TF_DENSE, EL_OPT_SGD, EL_TOPK_AUTO, EL_TOPK_MFMA, EL_TOPK_SCREEN,
                   EL_TOPK_SIMPLE, BprmfState, BprsgdState, check)

OPTIMIZERS = {"adam": EL_OPT_ADAM_TF_DENSE, "adam_tf_dense": EL_OPT_ADAM_TF_DENSE,
              "adam_lazy": EL_OPT_ADAM_LAZY, "sgd": EL_OPT_SGD, "sgd_dense": EL_OPT_SGD}
TOPK_ALGOS = {"auto": EL_TOPK_AUTO, "mfma": EL_TOPK_MFMA, "simple": EL_TOPK_SIMPLE, "screen": EL_TOPK_SCREEN}
BPR_ALGOS = {"auto": _lib.EL_BPR_AUTO, "atomic": _lib.EL_BPR_ATOMIC, "sorted": _lib.EL_BPR_SORTED}


def _ptr(t, dtype=None, name="tensor"):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise _lib.ElliotHipError(f"{name}: tensor must live on the GPU (got {t.device})")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


class Context:
    """One el_ctx per device (include/elliot_hip.h: el_ctx_create)."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.ElliotHipError("no GPU visible: elliot_amd needs an MI355X (gfx950)")
        self.device = torch.device("cuda", int(device))
        h = C.c_void_p()
        check(self.lib.el_ctx_create(int(device), C.byref(h)), "el_ctx_create")
        self.handle = h
        name = C.create_string_buffer(64)
        cus = C.c_int()
        hbm = C.c_int64()
        check(self.lib.el_device_info(h, name, 64, C.byref(cus), C.byref(hbm)), "el_device_info")
        self.arch, self.cus, self.hbm_bytes = name.value.decode(), cus.value, hbm.value

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_option(self, name, value):
        """A switch of the library on this context (include/elliot_hip.h: el_ctx_set_option); returns the previous value."""
        old = self.get_option(name)
        check(self.lib.el_ctx_set_option(self.handle, name.encode(), float(value)), "el_ctx_set_option")
        return old

    def get_option(self, name):
        v = C.c_double()
        check(self.lib.el_ctx_get_option(self.handle, name.encode(), C.byref(v)), "el_ctx_get_option")
        return v.value

    def option(self, name, value):
        """with ctx.option("ichunk", 64): ... -- the switch set inside the block, its previous value restored afterwards."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            old = self.set_option(name, value)
            try:
                yield
            finally:
                self.set_option(name, old)
        return scope()

    def timing(self, on, only=None):
        """Bracket kernel launches with hipEvents; `only` = a kernel name: just those launches (an event between two kernels
        costs their overlap, so timed regions keep them on the kernel of interest)."""
        check(self.lib.el_timing_filter(self.handle, only.encode() if only else None), "el_timing_filter")
        check(self.lib.el_timing_enable(self.handle, 1 if on else 0), "el_timing_enable")

    def timing_report(self):
        """{kernel_name: (launches, total_ms)} since the last report (synchronises the events)."""
        buf = C.create_string_buffer(1 << 16)
        check(self.lib.el_timing_report(self.handle, buf, len(buf)), "el_timing_report")
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            out[name] = (int(cnt), float(ms))
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.el_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_ctx_cache = {}


def get_context(device=0):
    device = int(device)
    if device not in _ctx_cache:
        _ctx_cache[device] = Context(device)
    return _ctx_cache[device]


class DeviceCSR:
    """CSR (int64 indptr, int32 indices, rows sorted ascending) resident in HBM.

    Stands in for the reference's dense bool masks (dataset.py:230-245) and the sampler's
    per-user positive lists (custom_sampler.py:19-22)."""

    def __init__(self, indptr, indices, n_cols, device):
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        if indptr.ndim != 1 or indptr[0] != 0 or indptr[-1] != indices.shape[0]:
            raise ValueError("malformed CSR")
        self.n_rows = indptr.shape[0] - 1
        self.n_cols = int(n_cols)
        self.nnz = int(indices.shape[0])
        self.indptr = torch.from_numpy(indptr).to(device)
        # never hand a zero-length buffer's null pointer to the kernels
        self.indices = torch.from_numpy(indices if self.nnz else np.zeros(1, np.int32)).to(device)

    @classmethod
    def from_tensors(cls, indptr, indices, n_cols):
        """Adopt device tensors (int64 indptr, int32 indices) without a host round trip."""
        self = cls.__new__(cls)
        assert indptr.dtype == torch.int64 and indices.dtype == torch.int32
        self.n_rows = indptr.shape[0] - 1
        self.n_cols = int(n_cols)
        self.nnz = int(indices.shape[0])
        self.indptr = indptr.contiguous()
        self.indices = indices.contiguous() if self.nnz else torch.zeros(1, dtype=torch.int32, device=indptr.device)
        return self

    @staticmethod
    def from_scipy(m, device):
        m = m.tocsr()
        m.sort_indices()
        return DeviceCSR(m.indptr, m.indices, m.shape[1], device)


def _csr_ptrs(csr):
    if csr is None:
        return None, None
    return _ptr(csr.indptr, torch.int64, "indptr"), _ptr(csr.indices, torch.int32, "indices")


# ------------------------------------------------------------------------------------------
# scoring + top-k
# ------------------------------------------------------------------------------------------
def score_topk(ctx, Gu, Gi, Bi, u_start, u_stop, k, excl=None, cand=None, item_offset=0, algo="auto",
               out_idx=None, out_val=None, items_unchanged=False):
    """BPRMF_batch_model.predict + get_top_k (BPRMF_batch_model.py:83-88) for users
    [u_start, u_stop): returns (idx int32 [n,k], val float32 [n,k]) on the device.
    items_unchanged: Gi / Bi are exactly what the previous call on this context scored against (block after block of one
    evaluation): the screened kernels keep that call's item-side image."""
    n = int(u_stop) - int(u_start)
    I_local, F = Gi.shape
    if Gu.shape[1] != F:
        raise ValueError("Gu/Gi factor mismatch")
    if out_idx is None:
        out_idx = torch.empty((n, k), dtype=torch.int32, device=ctx.device)
    if out_val is None:
        out_val = torch.empty((n, k), dtype=torch.float32, device=ctx.device)
    ep, ei = _csr_ptrs(excl)
    cp, ci = _csr_ptrs(cand)
    algo_id = TOPK_ALGOS[algo] if isinstance(algo, str) else int(algo)
    ws, need = None, 0
    if cand is None and algo_id in (EL_TOPK_AUTO, EL_TOPK_SCREEN):
        need = int(ctx.lib.el_score_topk_ws_bytes(int(n), int(I_local), int(F), int(k), int(excl.nnz) if excl is not None else 0, algo_id))
        if need:
            cached = getattr(ctx, "_topk_ws", None)
            if cached is None or cached.numel() < need:
                cached = torch.empty(need, dtype=torch.uint8, device=ctx.device)
                ctx._topk_ws = cached
            ws = C.c_void_p(cached.data_ptr())
    check(ctx.lib.el_score_topk(ctx.handle, ctx.stream(), _ptr(Gu, torch.float32, "Gu"),
                                _ptr(Gi, torch.float32, "Gi"), _ptr(Bi, torch.float32, "Bi"),
                                int(u_start), int(u_stop), int(item_offset), int(I_local), int(F),
                                ep, ei, cp, ci, int(k), _ptr(out_idx, torch.int32), _ptr(out_val, torch.float32),
                                algo_id | (_lib.EL_TOPK_ITEMS_UNCHANGED if items_unchanged else 0), ws, need),
          "el_score_topk")
    return out_idx, out_val


def score_topk_f64(ctx, P, Q, b, u_start, u_stop, k, excl=None, cand=None, item_offset=0):
    """MFModel.get_user_predictions (BPRMF_model.py:70-85), fp64 tables."""
    n = int(u_stop) - int(u_start)
    I_local, F = Q.shape
    out_idx = torch.empty((n, k), dtype=torch.int32, device=ctx.device)
    out_val = torch.empty((n, k), dtype=torch.float64, device=ctx.device)
    ep, ei = _csr_ptrs(excl)
    cp, ci = _csr_ptrs(cand)
    check(ctx.lib.el_score_topk_f64(ctx.handle, ctx.stream(), _ptr(P, torch.float64, "P"),
                                    _ptr(Q, torch.float64, "Q"), _ptr(b, torch.float64, "b"),
                                    int(u_start), int(u_stop), int(item_offset), int(I_local), int(F),
                                    ep, ei, cp, ci, int(k), _ptr(out_idx), _ptr(out_val)), "el_score_topk_f64")
    return out_idx, out_val


def dense_topk(ctx, preds, u_start, u_stop, k, excl=None, cand=None):
    """get_top_k (multi_vae_model.py:158-159 et al.) over a materialised [n_users, I] block."""
    n, I = preds.shape
    if n != int(u_stop) - int(u_start):
        raise ValueError("preds rows must equal u_stop - u_start")
    out_idx = torch.empty((n, k), dtype=torch.int32, device=ctx.device)
    out_val = torch.empty((n, k), dtype=torch.float32, device=ctx.device)
    ep, ei = _csr_ptrs(excl)
    cp, ci = _csr_ptrs(cand)
    check(ctx.lib.el_dense_topk(ctx.handle, ctx.stream(), _ptr(preds, torch.float32, "preds"), int(preds.stride(0)),
                                int(u_start), int(u_stop), int(I), ep, ei, cp, ci, int(k),
                                _ptr(out_idx), _ptr(out_val)), "el_dense_topk")
    return out_idx, out_val


def topk_merge(ctx, parts_idx, parts_val):
    """Merge [G, n_users, k] partial lists (item shards) into [n_users, k]."""
    G, n, k = parts_idx.shape
    out_idx = torch.empty((n, k), dtype=torch.int32, device=ctx.device)
    out_val = torch.empty((n, k), dtype=torch.float32, device=ctx.device)
    check(ctx.lib.el_topk_merge(ctx.handle, ctx.stream(), _ptr(parts_idx, torch.int32), _ptr(parts_val, torch.float32),
                                int(G), int(n), int(k), _ptr(out_idx), _ptr(out_val)), "el_topk_merge")
    return out_idx, out_val


# ------------------------------------------------------------------------------------------
# accuracy metrics on the device
# ------------------------------------------------------------------------------------------
METRIC_NAMES = ("nDCG", "Precision", "Recall", "HR", "MAP", "MRR", "F1")


def discount_table(cutoff, device):
    """ln 2 / ln(rank + 2) exactly as relevance.py:71-82 computes it (Python doubles), shipped to the device."""
    import math
    return torch.tensor([math.log(2) / math.log(r + 2) for r in range(cutoff)], dtype=torch.float64, device=device)


class DeviceTestSet:
    """Held-out interactions as a device CSR in PRIVATE ids: rows = users, columns ascending, float ratings."""

    def __init__(self, indptr, indices, ratings, device):
        indptr = np.asarray(indptr, dtype=np.int64)
        indices = np.asarray(indices, dtype=np.int32)
        self.nnz = int(indices.shape[0])
        self.n_rows = int(indptr.shape[0] - 1)
        self.indptr = torch.from_numpy(indptr).to(device)
        self.indices = torch.from_numpy(indices if self.nnz else np.zeros(1, np.int32)).to(device)
        self.ratings = None
        if ratings is not None:
            r = np.asarray(ratings, dtype=np.float32)
            self.ratings = torch.from_numpy(r if self.nnz else np.zeros(1, np.float32)).to(device)


    @classmethod
    def from_tensors(cls, indptr, indices, ratings):
        self = cls.__new__(cls)
        self.indptr, self.indices, self.ratings = indptr.contiguous(), indices.contiguous(), ratings
        self.nnz, self.n_rows = int(indices.shape[0]), int(indptr.shape[0] - 1)
        return self


def rec_metrics(ctx, rec_idx, test, threshold, cutoff, u_start=0, sums=None, per_user=False):
    """Sums of the seven accuracy metrics over the users of rec_idx's rows (absolute ids u_start ...) that have at
    least one relevant test item, plus their count: float64[8] on the device (ADDED to `sums` when given)."""
    n, ld = rec_idx.shape
    if sums is None:
        sums = torch.zeros(8, dtype=torch.float64, device=ctx.device)
    rows = torch.empty((n, 8), dtype=torch.float64, device=ctx.device) if per_user else None
    need = int(ctx.lib.el_rec_metrics_ws_bytes(int(n)))
    ws = getattr(ctx, "_metrics_ws", None)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 8), dtype=torch.uint8, device=ctx.device)
        ctx._metrics_ws = ws
    disc = discount_table(cutoff, ctx.device)
    check(ctx.lib.el_rec_metrics(ctx.handle, ctx.stream(), _ptr(rec_idx, torch.int32), int(ld), int(u_start), int(u_start + n),
                                 _ptr(test.indptr, torch.int64), _ptr(test.indices, torch.int32),
                                 _ptr(test.ratings, torch.float32), float(threshold), int(cutoff),
                                 C.c_void_p(disc.data_ptr()), C.c_void_p(sums.data_ptr()),
                                 C.c_void_p(rows.data_ptr()) if rows is not None else None,
                                 C.c_void_p(ws.data_ptr()), int(ws.numel())), "el_rec_metrics")
    return (sums, rows) if per_user else sums


def topk_screen_stats(ctx):
    """Diagnostics of the last screened score_topk call (el_topk_screen_stats; synchronises): users of the block, records the bf16 pass
    kept for them, users that took the exact fallback."""
    u, r, f = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    check(ctx.lib.el_topk_screen_stats(ctx.handle, ctx.stream(), C.byref(u), C.byref(r), C.byref(f)), "el_topk_screen_stats")
    return {"users": int(u.value), "records": int(r.value), "records_per_user": (r.value / u.value) if u.value else 0.0,
            "fallback_users": int(f.value)}


def fragile_users(ctx, Gu, Gi, Bi, u_start, u_stop, k, excl=None, cand=None, item_offset=0, algo="auto", flags=False):
    """Fragile-user report: how many of the users [u_start, u_stop) have a rank-k / rank-(k+1) score gap
    below the fp32 re-association bound F 2^-23 |u| max|i| -- for those (and only those) the top-k SET could differ under
    another correct fp32 summation order (TensorFlow's `tf.matmul`, BPRMF_batch_model.py:83-84).  Scores k + 1 entries per
    user with the fused kernels, then el_topk_fragile.  Returns a dict (and the uint8 flag tensor when flags=True)."""
    idx, val = score_topk(ctx, Gu, Gi, Bi, u_start, u_stop, k + 1, excl=excl, cand=cand, item_offset=item_offset, algo=algo)
    n = u_stop - u_start
    counts = torch.zeros(2, dtype=torch.int64, device=ctx.device)
    fl = torch.zeros(n, dtype=torch.uint8, device=ctx.device) if flags else None
    check(ctx.lib.el_topk_fragile(ctx.handle, ctx.stream(), _ptr(Gu, torch.float32), _ptr(Gi, torch.float32), int(Gu.shape[1]),
                                  int(u_start), int(u_stop), _ptr(idx, torch.int32), _ptr(val, torch.float32), int(idx.shape[1]),
                                  int(k), int(item_offset), C.c_void_p(fl.data_ptr()) if fl is not None else None,
                                  C.c_void_p(counts.data_ptr())), "el_topk_fragile")
    c = counts.cpu().tolist()
    rep = {"users": int(n), "k": int(k), "fragile": int(c[0]), "short_lists": int(c[1]),
           "bound": "score[k-1] - score[k] < F * 2^-23 * |u| * max(|i_k|, |i_k+1|)"}
    return (rep, fl) if flags else rep


# ------------------------------------------------------------------------------------------
# sampler
# ------------------------------------------------------------------------------------------
def sampler_meta(ctx, pos):
    """The sampler's per-user records of a positives CSR (el_bpr_sampler_meta_build), built once and cached on the CSR object:
    row start, row length and a 384-bit membership signature in one 64-byte line per user."""
    meta = getattr(pos, "_sampler_meta", None)
    if meta is None:
        nbytes = int(ctx.lib.el_bpr_sampler_meta_bytes(int(pos.n_rows)))
        meta = torch.empty(nbytes + 64, dtype=torch.uint8, device=ctx.device)
        off = (-meta.data_ptr()) % 64
        meta = meta[off:off + nbytes]                                 # 64-byte aligned view
        check(ctx.lib.el_bpr_sampler_meta_build(ctx.handle, ctx.stream(), *_csr_ptrs(pos), int(pos.n_rows),
                                                C.c_void_p(meta.data_ptr())), "el_bpr_sampler_meta_build")
        try:
            pos._sampler_meta = meta
        except Exception:
            pass
    return meta


def bpr_sample(ctx, pos, n, seed, first_sample=0, item_lo=0, item_hi=None, out=None, use_meta=True):
    """custom_sampler.Sampler.step (custom_sampler.py:31-46) on the device: n triplets.  use_meta: draw through the per-user
    records (same triplets, fewer cache lines per draw); False = the plain CSR path."""
    U, I = pos.n_rows, pos.n_cols
    if item_hi is None:
        item_hi = I
    if out is None:
        out = tuple(torch.empty((n,), dtype=torch.int32, device=ctx.device) for _ in range(3))
    meta = sampler_meta(ctx, pos) if use_meta else None
    check(ctx.lib.el_bpr_sample_meta(ctx.handle, ctx.stream(), *_csr_ptrs(pos), C.c_void_p(meta.data_ptr()) if meta is not None else None,
                                     int(U), int(I), int(item_lo), int(item_hi), int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_sample),
                                     int(n), _ptr(out[0], torch.int32), _ptr(out[1], torch.int32), _ptr(out[2], torch.int32)),
          "el_bpr_sample_meta")
    return out


def mt19937_init_state(seed):
    """np.random.seed(int) = MT19937 init_genrand: 624 words + position 624 (next draw twists)."""
    mt = np.empty(625, dtype=np.uint32)
    x = int(seed) & 0xFFFFFFFF
    mt[0] = x
    for i in range(1, 624):
        x = (1812433253 * (x ^ (x >> 30)) + i) & 0xFFFFFFFF
        mt[i] = x
    mt[624] = 624
    return mt


class MtReplaySampler:
    """Device replay of custom_sampler.Sampler's exact triplet stream (el_bpr_sample_mt19937)."""

    def __init__(self, ctx, ui_lists, pos, seed=42):
        """ui_lists: per-user positive lists in the reference's order (list(set(...)), custom_sampler.py:21);
        pos: DeviceCSR of the same rows sorted ascending."""
        self.ctx, self.pos = ctx, pos
        lp = np.concatenate([[0], np.cumsum([len(l) for l in ui_lists])]).astype(np.int64)
        li = np.concatenate([np.asarray(l, dtype=np.int32) for l in ui_lists]) if len(ui_lists) else np.zeros(0, np.int32)
        self.lists = DeviceCSR.__new__(DeviceCSR)
        self.lists.n_rows, self.lists.n_cols, self.lists.nnz = len(ui_lists), pos.n_cols, int(li.shape[0])
        self.lists.indptr = torch.from_numpy(lp).to(ctx.device)
        self.lists.indices = torch.from_numpy(li if li.size else np.zeros(1, np.int32)).to(ctx.device)
        self.state = torch.from_numpy(mt19937_init_state(seed).view(np.int32).copy()).to(ctx.device)
        self._ws = None

    def sample(self, n):
        ctx = self.ctx
        need = int(ctx.lib.el_bpr_sample_mt19937_ws_bytes(int(n)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=ctx.device)
        out = tuple(torch.empty(n, dtype=torch.int32, device=ctx.device) for _ in range(3))
        check(ctx.lib.el_bpr_sample_mt19937(ctx.handle, ctx.stream(), C.c_void_p(self.state.data_ptr()),
                                            _ptr(self.lists.indptr, torch.int64), _ptr(self.lists.indices, torch.int32),
                                            *_csr_ptrs(self.pos), int(self.pos.n_rows), int(self.pos.n_cols), int(n),
                                            _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                            C.c_void_p(self._ws.data_ptr()), self._ws.numel()), "el_bpr_sample_mt19937")
        return out


# ------------------------------------------------------------------------------------------
# BPRMF_batch (TF semantics)
# ------------------------------------------------------------------------------------------
def deterministic_item_sums(ctx):
    """True when the item segments a chunk boundary cuts are summed in a fixed order (no floating-point atomics anywhere in the
    sorted BPR step): two runs on the same batches then give the same bits, and the fused / deferred forms equal the every-row
    two-pass form bit for bit at any size."""
    fn = getattr(ctx.lib, "el_bprmf_deterministic", None)
    return bool(fn()) if fn is not None else False


def adam_lr_t(lr, step, beta1=0.9, beta2=0.999):
    """Keras Adam bias-corrected step size, computed in fp32 like TF does."""
    b1p = np.power(np.float32(beta1), np.float32(step))
    b2p = np.power(np.float32(beta2), np.float32(step))
    return float(np.float32(lr) * np.sqrt(np.float32(1.0) - b2p) / (np.float32(1.0) - b1p))


# ------------------------------------------------------------------------------------------
# HBM placement of the four streams of the dense Adam pass
# ------------------------------------------------------------------------------------------
_LAYOUT_CANDIDATES_MIB = (0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0, 5.5)


def _strided_tables(rows, F, count, gap_bytes, device):
    """`count` zeroed float32 [rows, F] tables carved from ONE allocation, consecutive tables `gap_bytes` apart."""
    n = rows * F
    stride = n + (int(gap_bytes) // 4 + 63) // 64 * 64            # in floats, 256 B granules
    big = torch.zeros(stride * count, dtype=torch.float32, device=device)
    return [big[t * stride:t * stride + n].view(rows, F) for t in range(count)], big


def tune_table_layout(ctx, rows, F, compact=False):
    """The TF-dense Adam pass streams theta, g, m and v of a table at once (4 reads + 3 writes per element; with compact
    gradient rows: theta, m, v and the rows of the batch's users).  How far apart those arrays sit in HBM decides how their
    channel / bank sequences collide: measured on MI355X, 0.61 to 0.82 ms for the same 1M x 128 table, periodic in the
    distance (about 6 MiB) -- and with one allocation per array it is the allocator's luck.  So the arrays of a large table
    are carved from one allocation and the distance is picked by timing the pass itself on a scratch copy (a dozen
    candidates, ~0.1 s, once per table shape and process).  Returns the gap in bytes."""
    key = (int(rows), int(F), bool(compact))
    cache = ctx.__dict__.setdefault("_layout_cache", {})
    if key in cache:
        return cache[key]
    if os.environ.get("EL_TUNE_LAYOUT", "1") == "0" or rows * F * 4 < (64 << 20):
        cache[key] = 0
        return 0
    dev = ctx.device
    dummy = [torch.zeros((64, F), dtype=torch.float32, device=dev) for _ in range(4)] + \
            [torch.zeros(64, dtype=torch.float32, device=dev) for _ in range(4)]
    uslot = grows = hit = slot = None
    if compact:
        # what a step looks like: ~63 % of the rows (1 - 1/e at B = U) carry a gradient row, slots ascending with the user id
        try:
            grows = torch.zeros((rows, F), dtype=torch.float32, device=dev)
        except RuntimeError:
            cache[key] = 0
            return 0
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        hit = (torch.rand(rows, generator=g, device=dev) < 0.63).to(torch.int64)
        slot = torch.arange(rows, dtype=torch.int64, device=dev) * hit
        uslot = torch.zeros(rows, dtype=torch.int64, device=dev)
    best, best_ms = 0, None
    check(ctx.lib.el_tuning_mode(ctx.handle, 1), "el_tuning_mode")      # probes under their own kernel symbols (profiles)
    for mib in _LAYOUT_CANDIDATES_MIB:
        gap = int(mib * (1 << 20))
        try:
            tabs, big = _strided_tables(rows, F, 3 if compact else 4, gap, dev)
        except RuntimeError:                                      # out of memory for the scratch copy: keep the plain layout
            break
        if compact:
            c = BprmfState(Gu=tabs[0].data_ptr(), gGu=None, mGu=tabs[1].data_ptr(), vGu=tabs[2].data_ptr(),
                           uslot=uslot.data_ptr(), gGu_rows=grows.data_ptr(), gGu_cap=rows,
                           Gi=dummy[0].data_ptr(), gGi=dummy[1].data_ptr(), mGi=dummy[2].data_ptr(), vGi=dummy[3].data_ptr(),
                           Bi=dummy[4].data_ptr(), gBi=dummy[5].data_ptr(), mBi=dummy[6].data_ptr(), vBi=dummy[7].data_ptr(),
                           tGu=None, tGi=None, tBi=None, U=rows, I=64, F=F)
        else:
            c = BprmfState(Gu=tabs[0].data_ptr(), gGu=tabs[1].data_ptr(), mGu=tabs[2].data_ptr(), vGu=tabs[3].data_ptr(),
                           Gi=dummy[0].data_ptr(), gGi=dummy[1].data_ptr(), mGi=dummy[2].data_ptr(), vGi=dummy[3].data_ptr(),
                           Bi=dummy[4].data_ptr(), gBi=dummy[5].data_ptr(), mBi=dummy[6].data_ptr(), vBi=dummy[7].data_ptr(),
                           tGu=None, tGi=None, tBi=None, U=rows, I=64, F=F)

        def run(it):
            if compact:
                torch.add(slot, hit, alpha=it << 32, out=uslot)          # stamps of step `it` on the touched rows
            check(ctx.lib.el_bprmf_apply(ctx.handle, ctx.stream(), C.byref(c), 0.001, EL_OPT_ADAM_TF_DENSE, it, 0.001), "el_bprmf_apply")

        for it in range(2):
            run(it + 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(4):
            run(it + 3)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 4
        if best_ms is None or ms < best_ms:
            best, best_ms = gap, ms
        del tabs, big
    check(ctx.lib.el_tuning_mode(ctx.handle, 0), "el_tuning_mode")
    cache[key] = best
    return best


class BprmfDeviceState:
    """Gu/Gi/Bi + gradient accumulators + Adam slots in HBM (BPRMF_batch_model.py:39-44)."""

    _LR_HIST = 1 << 16          # lr_t ring of the deferred decay (a power of two)

    def __init__(self, ctx, Gu, Gi, Bi, optimizer="adam", compact_user_grads=None, fused_user_step=None, deferred=None,
                 fused_item_step=None, item_deferred=None, replay="exact"):
        """compact_user_grads: user-row gradients as compact rows + per-user stamps (el_bprmf_state.uslot) instead of a dense
        [U,F] accumulator -- the dense Adam pass then reads a gradient only for the batch's users and re-zeroes nothing.
        None = automatic (TF-dense Adam on a user table of >= 64 MB with F % 4 == 0), True / False force it.
        deferred: Keras' every-row Adam decay of the USER table postponed per row and replayed bit for bit when a batch next
        contains the user or when the table is read (el_bprmf_state.Gu_last): a step moves only the batch's user rows.  It pays
        when a batch touches a small part of the users (10 M users, 1 M triplets: the user side of the step 7.2 -> ~2 ms) and costs
        when it touches most of them (B = U: +10 %), so None = decided at the first training call: on when 4 B <= U and the fused
        user-side step applies; reading `.Gu` syncs, `.mGu` / `.vGu` want sync() first.
        fused_item_step: the item side of the step as ONE kernel (el_bprmf_state.Gi_last): the item segments take Keras' Adam step on
        their rows in place, no dense gradient table is written, re-read and cleared.  None = whenever the fused user side applies
        (fused_item_step=False turns it off); grads() / apply() always run the two-pass form.
        item_deferred: with the fused item side, the item rows a batch leaves alone wait for their gradient-free Adam updates like the
        user rows do (replayed bit for bit when a batch next contains the item, or on sync() / reading `.Gi` / `.Bi`); False = they
        are replayed at the end of every step.  None = decided by the first batch size: on when 2 B <= I.
        replay: how a waiting row is brought forward over its gradient-free steps -- "exact": step by step, the bits of Keras'
        every-row pass; "series": in closed form from four row-level sums over the lr_t history (el_bprmf_state.replay_series:
        O(1) per element whatever the gap; as close to the exact-arithmetic recurrence as the fp32 step-by-step form, not its bits)."""
        self.ctx = ctx
        if replay not in ("exact", "series"):
            raise ValueError("replay must be 'exact' or 'series'")
        self.replay = replay
        self.opt = OPTIMIZERS[optimizer] if isinstance(optimizer, str) else int(optimizer)
        dev = ctx.device
        big = int(Gu.shape[0]) * int(Gu.shape[1]) * 4 >= (64 << 20)
        self.compact = bool(self.opt == EL_OPT_ADAM_TF_DENSE and int(Gu.shape[1]) % 4 == 0 and optimizer != "sgd_dense" and
                            (big if compact_user_grads is None else compact_user_grads))

        # fused user side (el_bprmf_state.Gu_next): segments + Keras Adam over every user row in ONE kernel, the new rows written to a
        # second table that swaps roles with Gu after every step -- whenever the compact form applies (fused_user_step=False: the
        # two-kernel form, which grads() / apply() always use)
        self.fused = bool(self.compact and fused_user_step is not False
                          and int(Gu.shape[1]) <= 512)
        self.Gu_next = None
        self._deferred_auto = bool(self.fused and deferred is None)    # decided by the first batch size (_resolve_deferred)
        self.deferred = bool(self.fused and deferred is True)
        self._pending = False
        self.item_fused = bool(self.fused and fused_item_step is not False)
        self._item_deferred_auto = bool(self.item_fused and item_deferred is None)
        self.item_deferred = bool(self.item_fused and item_deferred is True)
        self._pending_items = False

        def own(x, dt):
            if isinstance(x, np.ndarray):
                x = torch.from_numpy(np.ascontiguousarray(x))
            return x.to(device=dev, dtype=dt).contiguous().clone()

        self._Gi, self._Bi = own(Gi, torch.float32), own(Bi, torch.float32)
        self.U, self.F = int(Gu.shape[0]), int(Gu.shape[1])
        self.I = self._Gi.shape[0]
        z = torch.zeros_like
        adam = self.opt in (EL_OPT_ADAM_TF_DENSE, EL_OPT_ADAM_LAZY)
        self.uslot = self.gGu_rows = None
        if self.opt == EL_OPT_ADAM_TF_DENSE and self.U * self.F * 4 >= (64 << 20):
            # the dense Adam pass streams these at once: one allocation, tuned distance (tune_table_layout)
            gap = tune_table_layout(ctx, self.U, self.F, compact=self.compact)
            if self.compact and self.fused and not self.deferred:
                # (deferred=None: the pair is allocated; if the first batch turns the deferred decay on, Gu_next just stays unused)
                (self.Gu, self.mGu, self.vGu, self.Gu_next), self._user_block = _strided_tables(self.U, self.F, 4, gap, dev)
                self.gGu = None
            elif self.compact:
                (self.Gu, self.mGu, self.vGu), self._user_block = _strided_tables(self.U, self.F, 3, gap, dev)
                self.gGu = None
            else:
                (self.Gu, self.gGu, self.mGu, self.vGu), self._user_block = _strided_tables(self.U, self.F, 4, gap, dev)
            self.Gu.copy_(Gu if isinstance(Gu, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(Gu)))
            self.layout_gap = gap
        else:
            self.Gu = own(Gu, torch.float32)
            if self.fused and not self.deferred:
                self.Gu_next = torch.empty_like(self.Gu)
            self.gGu = None if self.compact else z(self.Gu)
            self.mGu = z(self.Gu) if adam else None
            self.vGu = z(self.Gu) if adam else None
            self.layout_gap = None
        if self.compact:
            self.uslot = torch.zeros(self.U, dtype=torch.int64, device=dev)       # (step << 32) | slot; step 0 is never used
        # item-side gradients in ONE buffer (gGi rows, then gBi): a data-parallel caller all-reduces `item_grad_flat` once
        rows_end = (self.I * self.F + 3) // 4 * 4                       # gBi starts on a 16-byte boundary
        self.item_grad_flat = torch.zeros(rows_end + self.I, dtype=torch.float32, device=dev)
        self.gGi = self.item_grad_flat[:self.I * self.F].view(self.I, self.F)
        self.gBi = self.item_grad_flat[rows_end:]
        self._item_block = None
        # (theta, m, v of the item table from ONE allocation a fixed distance apart was tried in round 5: consistent, not faster -- and the
        #  process-to-process spread of the item kernels is not about these tables at all: profiles/r06_placement_probe.md)
        self.mGi = z(self._Gi) if adam else None
        self.vGi = z(self._Gi) if adam else None
        self.mBi = z(self._Bi) if adam else None
        self.vBi = z(self._Bi) if adam else None
        rows = self.opt in (EL_OPT_ADAM_LAZY, EL_OPT_SGD) and optimizer != "sgd_dense"
        self.tGu = torch.zeros(self.U, dtype=torch.int32, device=dev) if rows else None
        self.tGi = torch.zeros(self.I, dtype=torch.int32, device=dev) if rows else None
        self.tBi = torch.zeros(self.I, dtype=torch.int32, device=dev) if rows else None
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self._step = 0
        self._ws = None
        self._c = BprmfState(
            Gu=self._Gu.data_ptr(), Gi=self._Gi.data_ptr(), Bi=self._Bi.data_ptr(),
            gGu=self.gGu.data_ptr() if self.gGu is not None else None, gGi=self.gGi.data_ptr(), gBi=self.gBi.data_ptr(),
            uslot=self.uslot.data_ptr() if self.compact else None, gGu_rows=None, gGu_cap=0,
            mGu=self.mGu.data_ptr() if adam else None, vGu=self.vGu.data_ptr() if adam else None,
            mGi=self.mGi.data_ptr() if adam else None, vGi=self.vGi.data_ptr() if adam else None,
            mBi=self.mBi.data_ptr() if adam else None, vBi=self.vBi.data_ptr() if adam else None,
            tGu=self.tGu.data_ptr() if rows else None, tGi=self.tGi.data_ptr() if rows else None,
            tBi=self.tBi.data_ptr() if rows else None, U=self.U, I=self.I, F=self.F,
            Gu_next=self.Gu_next.data_ptr() if self.Gu_next is not None else None, replay_series=int(replay == "series"))
        self.Gu_last = self.Gu_old = self.lr_hist = None
        self.Gi_last = None
        if self.deferred:
            self._enable_deferred()
        if self.item_fused:
            # fused item side: the step each item row is current at
            self.Gi_last = torch.zeros(self.I, dtype=torch.int32, device=dev)
            self._ensure_hist()
            self._c.Gi_last, self._c.Gi_defer = self.Gi_last.data_ptr(), int(self.item_deferred)

    def _ensure_hist(self):
        """The ring of the last _LR_HIST bias-corrected step sizes (what the replay of postponed row updates reads)."""
        if self.lr_hist is None:
            self.lr_hist = torch.zeros(self._LR_HIST, dtype=torch.float32, device=self.ctx.device)
            self._c.lr_hist, self._c.lr_hist_cap = self.lr_hist.data_ptr(), self._LR_HIST

    def _enable_deferred(self):
        dev = self.ctx.device
        self.deferred = True
        self.Gu_last = torch.full((self.U,), int(self._step), dtype=torch.int32, device=dev)     # every row is current now
        self._ensure_hist()
        self._c.Gu_last = self.Gu_last.data_ptr()
        if getattr(self, "_user_block", None) is None and not self._deferred_auto:
            self.Gu_next = None                                  # in-place updates: the second table is not needed
            self._c.Gu_next = None

    def _resolve_deferred(self, B):
        """A state built with deferred=None follows the batch size: the deferred decay pays when a batch leaves most user rows
        alone (4 B <= U), the every-row fused step when it touches most of them.  Switching costs one replay of the pending rows."""
        if int(B) == getattr(self, "_resolved_B", None):
            return
        self._resolved_B = int(B)
        if self._deferred_auto:
            want = 4 * int(B) <= self.U
            if want != self.deferred:
                auto = self._deferred_auto
                self.set_deferred(want)
                self._deferred_auto = auto
        if self._item_deferred_auto:
            # the item side: a batch draws B negatives uniformly over the catalogue (custom_sampler.py:39-41) and B positives: it pays
            # when those leave most item rows alone
            self.set_item_deferred(2 * int(B) <= self.I, _auto=True)

    # -- deferred decay of the user table (el_bprmf_state.Gu_last) ----------------------------------------------------------
    @property
    def Gu(self):
        """The user table, every row current (pending row updates of the deferred decay are replayed first)."""
        if self._pending:
            self.sync()
        return self._Gu

    @Gu.setter
    def Gu(self, t):
        self._Gu = t

    @property
    def Gi(self):
        """The item table, every row current (pending row updates of the deferred item decay are replayed first)."""
        if self._pending_items:
            self.sync()
        return self._Gi

    @property
    def Bi(self):
        if self._pending_items:
            self.sync()
        return self._Bi

    def sync(self):
        """Deferred decay: bring every user row and every item row (theta, m, v) to the current step; no-op otherwise."""
        if self.deferred and self._pending:
            self._pending = False
            check(self.ctx.lib.el_bprmf_sync_users(self.ctx.handle, self.ctx.stream(), C.byref(self._c), int(self._step)),
                  "el_bprmf_sync_users")
        if self.item_fused and self._pending_items:
            self._pending_items = False
            check(self.ctx.lib.el_bprmf_sync_items(self.ctx.handle, self.ctx.stream(), C.byref(self._c), int(self._step)),
                  "el_bprmf_sync_items")

    def set_item_deferred(self, on, _auto=False):
        """Fused item side: let the item rows a batch leaves alone wait for their gradient-free updates (True) or replay them at
        the end of every step (False)."""
        on = bool(on) and self.item_fused
        if not _auto:
            self._item_deferred_auto = False
        if on == self.item_deferred:
            return
        if not on:
            self.sync()
        self.item_deferred = on
        self._c.Gi_defer = int(on)

    def set_deferred(self, on):
        """Switch the deferred decay off (data-parallel owners that run the step as grads + collective + apply) or on again."""
        on = bool(on)
        self._deferred_auto = False
        if on == self.deferred:
            return
        if on:
            if not self.fused:
                raise ValueError("the deferred decay needs the fused user-side step (TF-dense Adam, compact user gradients, F % 4 == 0)")
            self._enable_deferred()
            return
        self.sync()
        self.deferred = False
        self._c.Gu_last = self._c.Gu_old = None
        self._c.Gu_old_cap = 0
        self.Gu_last = self.Gu_old = None                       # (the lr ring stays: the fused item side reads it too)
        if self.fused:                                          # the fused user-side step needs its second table from here on
            if self.Gu_next is None:
                self.Gu_next = torch.empty_like(self._Gu)
            self._c.Gu_next = self.Gu_next.data_ptr()

    def _ensure_old(self, B):
        self._resolve_deferred(B)
        if self.deferred and (self.Gu_old is None or self.Gu_old.shape[0] < B):
            self.Gu_old = torch.empty((int(B), self.F), dtype=torch.float32, device=self.ctx.device)
            self._c.Gu_old, self._c.Gu_old_cap = self.Gu_old.data_ptr(), int(B)

    def _swap_user_tables(self, steps=1):
        """After `steps` fused train steps the current user table is the other one of the pair (include/elliot_hip.h, Gu_next)."""
        if self.deferred:
            self._pending = True                                # in place; rows outside the batches wait for their replay
            return
        if self.fused and steps % 2:
            self._Gu, self.Gu_next = self.Gu_next, self._Gu
            self._c.Gu, self._c.Gu_next = self._Gu.data_ptr(), self.Gu_next.data_ptr()
            for c in getattr(self, "_c_clones", ()):
                c.Gu, c.Gu_next = self._c.Gu, self._c.Gu_next

    @property
    def step(self):
        return self._step

    @step.setter
    def step(self, value):
        """Moving the counter from outside (a checkpoint's count, a data-parallel owner's begin_step) is not an optimiser step:
        every row is first brought up to date at the old count and then stamped with the new one.  (The training calls of this
        class advance the counter through _advance.)"""
        value = int(value)
        if self.compact and value < self._step:
            self.uslot.zero_()                 # stamps of later steps must not be mistaken for this step's gradient rows
        if value != self._step:
            if getattr(self, "deferred", False):
                self.sync()
                self.Gu_last.fill_(value)
            if getattr(self, "item_fused", False):
                self.sync()
                self.Gi_last.fill_(value)
        self._step = value

    def _advance(self, n=1):
        """The counter of a training call: the library performs (or postpones, per row) exactly these optimiser steps."""
        self._step += int(n)

    def _after_steps(self, steps=1):
        self._swap_user_tables(steps)
        if self.item_fused and self.item_deferred:
            self._pending_items = True

    def _two_pass(self, on):
        """grads() / apply() run the every-row two-pass form: the per-row features stand aside (rows current first) and come back
        with every row stamped at the step the dense pass left it at."""
        if on:
            self.sync()
            self._c.Gu_last = None
            self._c.Gi_last = None
            return
        if self.deferred:
            self.Gu_last.fill_(self._step)
            self._c.Gu_last = self.Gu_last.data_ptr()
        if self.item_fused:
            self.Gi_last.fill_(self._step)
            self._c.Gi_last = self.Gi_last.data_ptr()

    def ensure_rows(self, B):
        """Compact mode: gradient rows for a batch of B triplets (slot = sorted position of a user's first occurrence)."""
        if self.compact and (self.gGu_rows is None or self.gGu_rows.shape[0] < B):
            self.gGu_rows = torch.empty((int(B), self.F), dtype=torch.float32, device=self.ctx.device)
            self._c.gGu_rows = self.gGu_rows.data_ptr()
            self._c.gGu_cap = int(B)
            for c in getattr(self, "_c_clones", ()):                      # (multi-GPU: split user / item apply states)
                c.gGu_rows, c.gGu_cap = self._c.gGu_rows, self._c.gGu_cap

    def user_grad_dense(self):
        """Dense [U,F] view of the user-row gradients of the LAST gradient pass (tests / diagnostics): the accumulator itself, or
        the compact rows scattered by their stamps."""
        if not self.compact:
            return self.gGu
        out = torch.zeros((self.U, self.F), dtype=torch.float32, device=self.ctx.device)
        ent = self.uslot
        hit = (ent >> 32) == max(int(self.uslot.max().item()) >> 32, 1)
        rows = torch.nonzero(hit).flatten()
        out[rows] = self.gGu_rows[(ent[rows] & 0xFFFFFFFF)]
        return out

    def train_step(self, u, i, j, lr, l_w, l_b, algo="auto"):
        """BPRMF_batch_model.train_step (BPRMF_batch_model.py:58-80).  u,i,j: int32 device tensors.
        The batch loss is accumulated into self.loss (device double) -- no host sync here."""
        B = u.numel()
        if self.compact:
            self._resolve_deferred(B)           # (before the step counter moves: switching the deferred decay stamps the rows with it)
        self._advance()
        lr_t = adam_lr_t(lr, self.step)
        algo = BPR_ALGOS[algo] if isinstance(algo, str) else int(algo)
        ws, ws_bytes = None, 0
        if self.compact:
            if algo == _lib.EL_BPR_ATOMIC:
                raise ValueError("compact user-gradient rows need the sorted gradient path")
            if not self.fused:
                self.ensure_rows(B)
            self._ensure_old(B)
        if algo == _lib.EL_BPR_SORTED or (algo == _lib.EL_BPR_AUTO and B >= 2048) or self.compact:
            need = int(self.ctx.lib.el_bprmf_ws_bytes(int(B), int(self.U), int(self.I), int(self.F)))
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.ctx.device)
            ws, ws_bytes = C.c_void_p(self._ws.data_ptr()), self._ws.numel()
        check(self.ctx.lib.el_bprmf_train_step(self.ctx.handle, self.ctx.stream(), C.byref(self._c),
                                               _ptr(u, torch.int32, "u"), _ptr(i, torch.int32, "i"),
                                               _ptr(j, torch.int32, "j"), int(B), float(lr), float(l_w), float(l_b),
                                               self.opt, int(self.step), float(lr_t), _ptr(self.loss, torch.float64),
                                               algo, ws, ws_bytes),
              "el_bprmf_train_step")
        self._after_steps()

    def grads(self, u, i, j, l_w, l_b):
        """First half of train_step: loss + the summed row gradients of the batch (what OptimizerV2 receives after its segment
        sum, BPRMF_batch_model.py:77) into gGu (or the compact rows) / gGi / gBi, no optimiser.  apply() consumes them."""
        B = u.numel()
        need = int(self.ctx.lib.el_bprmf_ws_bytes(int(B), int(self.U), int(self.I), int(self.F)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.ctx.device)
        self.ensure_rows(B)
        if self.deferred or self.item_fused:
            # the two-call form runs the every-row passes (apply); the rows are brought up to date and the features stand aside
            self._two_pass(True)
        check(self.ctx.lib.el_bprmf_grads(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32, "u"),
                                          _ptr(i, torch.int32, "i"), _ptr(j, torch.int32, "j"), int(B), float(l_w), float(l_b),
                                          int(self.step + 1), _ptr(self.loss, torch.float64), C.c_void_p(self._ws.data_ptr()),
                                          self._ws.numel()), "el_bprmf_grads")

    def apply(self, lr):
        """Second half: the optimiser (TF-dense Adam / dense SGD) on the accumulated gradients; accumulators come back clean."""
        if self.deferred or self.item_fused:
            self._two_pass(True)                               # (apply() without a grads() in front: the same stand-aside)
        self._advance()
        check(self.ctx.lib.el_bprmf_apply(self.ctx.handle, self.ctx.stream(), C.byref(self._c), float(lr), int(self.opt),
                                          int(self.step), float(adam_lr_t(lr, self.step))), "el_bprmf_apply")
        if self.deferred or self.item_fused:
            self._two_pass(False)                              # the every-row passes moved every row

    # -- the step in two halves for a software pipeline: ordering a batch (prep + radix sort) reads only its triplets, so the
    #    batch of step t+1 can be drawn and ordered on a side stream while step t's segment kernels and optimiser pass run
    def sort_workspace(self, B):
        """A workspace tensor for presort() / train_step_presorted() (el_bprmf_ws_bytes): one per batch in flight."""
        return torch.empty(int(self.ctx.lib.el_bprmf_ws_bytes(int(B), int(self.U), int(self.I), int(self.F))), dtype=torch.uint8, device=self.ctx.device)

    def presort(self, u, i, j, ws):
        """First half of the sorted gradient path: (row, triplet) pairs of the batch, ordered, into `ws` (current stream)."""
        check(self.ctx.lib.el_bprmf_presort(self.ctx.handle, self.ctx.stream(), _ptr(u, torch.int32, "u"), _ptr(i, torch.int32, "i"),
                                            _ptr(j, torch.int32, "j"), int(u.numel()), int(self.U), int(self.I),
                                            C.c_void_p(ws.data_ptr()), ws.numel()), "el_bprmf_presort")

    def train_step_presorted(self, u, i, j, lr, l_w, l_b, ws):
        """train_step on a batch that presort() ordered into `ws`: segment kernels + loss, then the optimiser -- the same kernels,
        the same results as train_step(algo="sorted")."""
        B = u.numel()
        if self.opt not in (EL_OPT_ADAM_TF_DENSE, EL_OPT_SGD) or self.tGu is not None:
            raise ValueError("train_step_presorted: the dense optimisers only (adam_tf_dense, sgd_dense)")
        if not self.fused:
            self.ensure_rows(B)
        self._ensure_old(B)
        self._advance()
        check(self.ctx.lib.el_bprmf_train_step_presorted(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32, "u"),
                                                         _ptr(i, torch.int32, "i"), _ptr(j, torch.int32, "j"), int(B), float(lr), float(l_w),
                                                         float(l_b), int(self.opt), int(self.step), float(adam_lr_t(lr, self.step)),
                                                         _ptr(self.loss, torch.float64), C.c_void_p(ws.data_ptr()), ws.numel()),
              "el_bprmf_train_step_presorted")
        self._after_steps()

    def train_loop(self, pos, events, B, seed, first_sample, lr, l_w, l_b, algo="auto"):
        """One epoch of `for batch in sampler.step(events, B): train_step(batch)` (BPRMF_batch.py:100-109) from a single
        library call: the same Philox stream and the same kernels as the per-batch calls, no host round trip in between."""
        steps = (int(events) + int(B) - 1) // int(B)
        if steps == 0:
            return 0
        if pos.n_rows != self.U or pos.n_cols != self.I:
            raise ValueError("train_loop: the positives' CSR must describe this state's users x items")
        algo = BPR_ALGOS[algo] if isinstance(algo, str) else int(algo)
        lr_t = self._lr_t_host = np.array([adam_lr_t(lr, self.step + 1 + k) for k in range(steps)], dtype=np.float32)
        # (kept alive on self: the library may copy the table with an asynchronous memcpy on the stream)
        if not self.fused:
            self.ensure_rows(B)
        self._ensure_old(B)
        need = int(self.ctx.lib.el_bprmf_ws_bytes(int(B), int(self.U), int(self.I), int(self.F))) if (B >= 2048 or self.compact) else 0
        if need and (self._ws is None or self._ws.numel() < need):
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.ctx.device)
        lneed = int(self.ctx.lib.el_bprmf_train_loop_ws_bytes(int(events), int(B)))
        buf = getattr(self, "_loop_ws", None)
        if buf is None or buf.numel() != lneed:                          # (a stable address keeps the captured graph valid)
            buf = self._loop_ws = torch.empty(lneed, dtype=torch.uint8, device=self.ctx.device)
        check(self.ctx.lib.el_bprmf_train_loop(
            self.ctx.handle, self.ctx.stream(), C.byref(self._c), *_csr_ptrs(pos), int(seed) & 0xFFFFFFFFFFFFFFFF,
            int(first_sample), int(events), int(B), float(lr), float(l_w), float(l_b), self.opt, int(self.step + 1),
            lr_t.ctypes.data_as(C.c_void_p), _ptr(self.loss, torch.float64), algo,
            C.c_void_p(self._ws.data_ptr()) if need else None, self._ws.numel() if need else 0,
            C.c_void_p(buf.data_ptr()), lneed, C.c_void_p(sampler_meta(self.ctx, pos).data_ptr())), "el_bprmf_train_loop")
        self._advance(steps)                                    # (consecutive steps: not a jump of the counter)
        self._after_steps(steps)
        return steps

    def pop_loss(self):
        v = float(self.loss.item())
        self.loss.zero_()
        return v



# ------------------------------------------------------------------------------------------
# graph propagation: LightGCN (graph_based/lightgcn) -- SURVEY 8f N3
# ------------------------------------------------------------------------------------------
SPMM_CHUNK = 512          # el_graph.hip: SPMM_CH


def normalized_bipartite_laplacian(train_csr_indptr, train_csr_indices, n_users, n_items):
    """LightGCN._create_adj_mat (LightGCN.py:96-118) without the dok / lil detour: the symmetric adjacency over U + I nodes (users first),
    D^-1/2 A D^-1/2 with the reference's arithmetic -- fp32 row sums + 1e-7 (added in fp32), power -1/2 in fp32, the two diagonal
    products one fp32 multiplication each: value(r, c) = fl(fl(1 * dinv[c]) * dinv[r]).  Returns CSR (indptr int64, indices int32
    ascending, vals fp32) as NumPy arrays.  (The reference's own construction is pinned against this one in tests/test_oracle_graph.py.)"""
    import scipy.sparse as sp
    indptr = np.asarray(train_csr_indptr, dtype=np.int64)
    cols = np.asarray(train_csr_indices, dtype=np.int64)
    T = cols.shape[0]
    rows = np.repeat(np.arange(n_users, dtype=np.int64), np.diff(indptr))
    N = n_users + n_items
    A = sp.csr_matrix((np.ones(2 * T, np.float32), (np.concatenate([rows, cols + n_users]), np.concatenate([cols + n_users, rows]))), shape=(N, N))
    A.sum_duplicates()
    A.data[:] = 1.0                                              # (a dok assignment of the ratings matrix: entries are the stored values;
    #                                                               sp_i_train holds ones, dataset.py:236-241)
    rowsum = np.asarray(A.sum(1), dtype=np.float32).reshape(-1)
    rowsum = rowsum + np.float32(1e-7)
    dinv = np.power(rowsum, np.float32(-0.5)).astype(np.float32)
    dinv[np.isinf(dinv)] = 0.0
    A.sort_indices()
    r = np.repeat(np.arange(N, dtype=np.int64), np.diff(A.indptr))
    vals = (A.data.astype(np.float32) * dinv[A.indices]).astype(np.float32) * dinv[r]
    return A.indptr.astype(np.int64), A.indices.astype(np.int32), vals.astype(np.float32)


def normalized_bipartite_laplacian_device(indptr, indices, n_users, n_items):
    """The same operator built on the device from a device CSR (large synthetic graphs: 1.6e8 non-zeros take a minute through SciPy).
    Same pattern and order; the values come from torch's fp32 pow, which may differ from NumPy's in the last bit -- the plugin (whose
    values are pinned to the reference's, tests/test_oracle_graph.py) uses the host builder above."""
    dev = indptr.device
    U, I = int(n_users), int(n_items)
    deg_u = (indptr[1:] - indptr[:-1])
    cols = indices.to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(U, device=dev, dtype=torch.int64), deg_u)
    deg_i = torch.bincount(cols, minlength=I)
    deg = torch.cat([deg_u, deg_i]).to(torch.float32) + torch.tensor(1e-7, dtype=torch.float32, device=dev)
    dinv = deg.pow(-0.5)
    order = torch.sort(cols * U + rows, stable=True).indices                       # item rows: (item, user) ascending
    t_rows, t_cols = cols[order], rows[order]
    lap_indptr = torch.cat([indptr, indptr[-1] + torch.cumsum(deg_i, 0)]).to(torch.int64)
    lap_indices = torch.cat([cols + U, t_cols]).to(torch.int32)
    r_all = torch.cat([rows, t_rows + U])
    vals = (dinv[lap_indices.long()] * dinv[r_all]).to(torch.float32)
    return lap_indptr, lap_indices, vals


class GraphCSR:
    """el_graph_csr: a sparse N x N operator over the stacked [users; items] table, device-resident, with the chunk decomposition
    of its product (every row cut into chunks of <= 512 non-zeros; multi-chunk rows reduce their partial rows in order)."""

    def __init__(self, ctx, indptr, indices, vals, n_users, width):
        dev = ctx.device
        self.ctx = ctx
        t = lambda a, dt: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(device=dev, dtype=dt).contiguous()
        self.indptr, self.indices, self.vals = t(indptr, torch.int64), t(indices, torch.int32), t(vals, torch.float32)
        self.N, self.n0, self.width = int(self.indptr.numel() - 1), int(n_users), int(width)
        ln = self.indptr[1:] - self.indptr[:-1]
        nch = torch.clamp((ln + SPMM_CHUNK - 1) // SPMM_CHUNK, min=1)                 # an empty row keeps one (empty) chunk
        first = torch.cumsum(nch, 0) - nch                                            # first chunk of every row
        n_chunks = int(nch.sum().item())
        row_of = torch.repeat_interleave(torch.arange(self.N, device=dev, dtype=torch.int64), nch)
        k_in_row = torch.arange(n_chunks, device=dev, dtype=torch.int64) - first[row_of]
        self.chunk_row = row_of.to(torch.int32)
        self.chunk_lo = (self.indptr[:-1][row_of] + k_in_row * SPMM_CHUNK).contiguous()
        multi = nch > 1
        mrows = torch.nonzero(multi).flatten()
        mcnt = nch[mrows]
        mfirst = torch.cumsum(mcnt, 0) - mcnt                                         # first partial slot of every multi-chunk row
        slot_of_row = torch.full((self.N,), -1, dtype=torch.int64, device=dev)
        slot_of_row[mrows] = mfirst
        cs = slot_of_row[row_of]
        self.chunk_slot = torch.where(cs >= 0, cs + k_in_row, cs).to(torch.int32)
        self.multi_row, self.multi_slot, self.multi_cnt = mrows.to(torch.int32), mfirst.to(torch.int32), mcnt.to(torch.int32)
        n_part = int(mcnt.sum().item()) if mrows.numel() else 0
        self.part = torch.empty((max(n_part, 1), self.width), dtype=torch.float32, device=dev)
        p = lambda x: x.data_ptr()
        self._c = _lib.GraphCsr(indptr=p(self.indptr), indices=p(self.indices), vals=p(self.vals), N=self.N, n0=self.n0,
                                chunk_row=p(self.chunk_row), chunk_lo=p(self.chunk_lo), chunk_slot=p(self.chunk_slot), n_chunks=n_chunks,
                                multi_row=p(self.multi_row) if mrows.numel() else None, multi_slot=p(self.multi_slot) if mrows.numel() else None,
                                multi_cnt=p(self.multi_cnt) if mrows.numel() else None, n_multi=int(mrows.numel()), part=p(self.part))

    @property
    def nnz(self):
        return int(self.indices.numel())

    def spmm(self, X0, X1, out0=None, out1=None):
        """[Y0; Y1] = L [X0; X1] (el_spmm_csr_f32)."""
        F = int(X0.shape[1])
        if F > self.width:
            raise ValueError("GraphCSR was sized for narrower tables")
        out0 = torch.empty_like(X0) if out0 is None else out0
        out1 = torch.empty_like(X1) if out1 is None else out1
        check(self.ctx.lib.el_spmm_csr_f32(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(X0, torch.float32), _ptr(X1, torch.float32),
                                           F, _ptr(out0, torch.float32), _ptr(out1, torch.float32)), "el_spmm_csr_f32")
        return out0, out1


class LightGcnDeviceState:
    """LightGCN_model (graph_based/lightgcn/LightGCN_model.py:19-172) in HBM: Gu / Gi with Keras Adam slots + the normalised adjacency.
    One train step = `_propagate_embeddings` (ASSIGNED to the variables, :93-94) + the BPR head of BPRMF_batch without an item bias and
    with the L2 term doubled (:150-158) + Adam's every-row apply -- the head runs on the segment kernels of the BPR-MF path
    (BprmfDeviceState with the every-row fused forms: the propagation reads and rewrites every row each step, nothing may wait)."""

    def __init__(self, ctx, Gu, Gi, graph, n_layers=1):
        self.ctx, self.graph, self.n_layers = ctx, graph, int(n_layers)
        self.bpr = BprmfDeviceState(ctx, Gu, Gi, np.zeros(int(Gi.shape[0]), np.float32), optimizer="adam_tf_dense", deferred=False,
                                    item_deferred=False)
        self.U, self.I, self.F = self.bpr.U, self.bpr.I, self.bpr.F
        if graph.N != self.U + self.I or graph.n0 != self.U:
            raise ValueError("the graph does not describe these tables")
        need = int(ctx.lib.el_lightgcn_ws_bytes(self.U, self.I, self.F, self.n_layers))
        self._ws = torch.empty(max(need, 16), dtype=torch.uint8, device=ctx.device)

    @property
    def Gu(self):
        return self.bpr.Gu

    @property
    def Gi(self):
        return self.bpr.Gi

    @property
    def step(self):
        return self.bpr.step

    def propagate(self):
        """_propagate_embeddings (:68-94), in place."""
        st = self.bpr
        st.sync()
        check(self.ctx.lib.el_lightgcn_propagate(self.ctx.handle, self.ctx.stream(), C.byref(self.graph._c), _ptr(st._Gu, torch.float32),
                                                 _ptr(st._Gi, torch.float32), self.F, self.n_layers, C.c_void_p(self._ws.data_ptr()),
                                                 self._ws.numel()), "el_lightgcn_propagate")

    def train_step(self, u, i, j, lr, l_w):
        """train_step (:136-167): propagate, then the BPR step on the propagated tables.  reg_loss = l_w * sum(l2_loss) * 2 is the
        BPRMF_batch head's l_w * sum(l2_loss) with twice the coefficient; there is no item bias: the bias slot of the BPR state is
        kept at zero (its gradient -- the triplet's dloss/dd -- is discarded after every step), l_b = 0."""
        self.propagate()
        st = self.bpr
        st.train_step(u, i, j, lr, 2.0 * float(l_w), 0.0)
        st.sync()
        st._Bi.zero_()
        st.mBi.zero_()
        st.vBi.zero_()

    def pop_loss(self):
        return self.bpr.pop_loss()


class NgcfDeviceState:
    """NGCFModel (graph_based/ngcf/NGCF_model.py:18-226) in HBM.  Gu / Gi are [rows, sum(weight_size_list)] wide (:88-91): the first
    embed_k columns are the trainable layer-0 embeddings, the others are REWRITTEN by every `_propagate_embeddings` (:106-142) with the
    row-normalised outputs of the propagation layers; the BPR head (bias-free, L2 doubled, :199-209) and Adam act on the full-width rows.
    The GraphLayers (W_1, b_1, W_2, b_2 per layer) see the loss through reg_loss only -- the propagated tables are ASSIGNED (:141-142) --
    so their Adam step runs on g = 2 l_w theta (el_adam_l2_dense).  layers: [{"W1": [kin, kout], "b1": [1, kout], "W2", "b2"}, ...]."""

    def __init__(self, ctx, Gu, Gi, graph, layers, embed_k, message_dropout=None, dropout_seed=42):
        self.ctx, self.graph, self.embed_k = ctx, graph, int(embed_k)
        dev = ctx.device
        self.bpr = BprmfDeviceState(ctx, Gu, Gi, np.zeros(int(Gi.shape[0]), np.float32), optimizer="adam_tf_dense", deferred=False,
                                    item_deferred=False)
        self.U, self.I, self.W = self.bpr.U, self.bpr.I, self.bpr.F
        own = lambda x: (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))).to(device=dev, dtype=torch.float32).contiguous().clone()
        self.layers = [{k: own(v) for k, v in l.items()} for l in layers]
        self.slots = [{k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in l.items()} for l in self.layers]
        sizes = [self.embed_k] + [int(l["W1"].shape[1]) for l in self.layers]
        if sum(sizes) != self.W or any(int(l["W1"].shape[0]) != sizes[n] for n, l in enumerate(self.layers)):
            raise ValueError("layer shapes do not chain to the table width")
        self.sizes = sizes
        self.message_dropout = [float(x) for x in (message_dropout or [0.0] * len(self.layers))]
        self.dropout_seed = int(dropout_seed)
        N, kmax = self.U + self.I, max(sizes)
        self._lap = torch.empty((N, kmax), dtype=torch.float32, device=dev)
        self._x2 = torch.empty((N, 2 * kmax), dtype=torch.float32, device=dev)
        self._s = torch.empty((N, kmax), dtype=torch.float32, device=dev)
        self._ego = [torch.empty((N, kmax), dtype=torch.float32, device=dev) for _ in range(2)]

    @property
    def Gu(self):
        return self.bpr.Gu

    @property
    def Gi(self):
        return self.bpr.Gi

    @property
    def step(self):
        return self.bpr.step

    def propagate(self):
        """_propagate_embeddings (:106-142), in place on the column blocks of Gu / Gi."""
        st, ctx, U, N = self.bpr, self.ctx, self.U, self.U + self.I
        st.sync()
        k0 = self.embed_k
        ego = self._ego[0].view(-1)[:N * k0].view(N, k0)
        ego[:U].copy_(st._Gu[:, :k0])
        ego[U:].copy_(st._Gi[:, :k0])
        off = k0
        for n, l in enumerate(self.layers):
            kin, kout = self.sizes[n], self.sizes[n + 1]
            lap = self._lap.view(-1)[:N * kin].view(N, kin)
            self.graph.spmm(ego[:U], ego[U:], lap[:U], lap[U:])
            x2 = self._x2.view(-1)[:N * 2 * kin].view(N, 2 * kin)
            check(ctx.lib.el_ngcf_pre(ctx.handle, ctx.stream(), _ptr(ego, torch.float32), _ptr(lap, torch.float32), N, kin, _ptr(x2, torch.float32)), "el_ngcf_pre")
            wcat = torch.cat([l["W1"], l["W2"]], 0)
            bcat = (l["b1"] + l["b2"]).reshape(-1)
            s = self._s.view(-1)[:N * kout].view(N, kout)
            gemm(ctx, x2, wcat, bias=bcat, out=s)
            nxt = self._ego[(n + 1) & 1].view(-1)[:N * kout].view(N, kout)
            check(ctx.lib.el_ngcf_post(ctx.handle, ctx.stream(), _ptr(s, torch.float32), N, U, kout, float(self.message_dropout[n]),
                                       self.dropout_seed & 0xFFFFFFFFFFFFFFFF, int(st.step * 16 + n) & 0xFFFFFFFF, _ptr(nxt, torch.float32),
                                       _ptr(st._Gu, torch.float32), _ptr(st._Gi, torch.float32), self.W, off), "el_ngcf_post")
            ego, off = nxt, off + kout

    def train_step(self, u, i, j, lr, l_w):
        """train_step (:187-217): propagate, the bias-free BPR step with the doubled L2 term on the full-width rows, the GraphLayers' L2-only
        Adam step; the batch loss includes l_w * sum ||layer parameter||^2 (:203-206)."""
        self.propagate()
        st, ctx = self.bpr, self.ctx
        st.train_step(u, i, j, lr, 2.0 * float(l_w), 0.0)
        st.sync()
        st._Bi.zero_(), st.mBi.zero_(), st.vBi.zero_()
        reg = sum((p.double() ** 2).sum() for l in self.layers for p in l.values())
        st.loss += float(l_w) * reg                                   # (device arithmetic: no synchronisation)
        lr_t = adam_lr_t(lr, st.step)
        for l, sl in zip(self.layers, self.slots):
            for k, p in l.items():
                m, v = sl[k]
                check(ctx.lib.el_adam_l2_dense(ctx.handle, ctx.stream(), _ptr(p, torch.float32), _ptr(m, torch.float32), _ptr(v, torch.float32),
                                               p.numel(), float(lr_t), 2.0 * float(l_w)), "el_adam_l2_dense")

    def pop_loss(self):
        return self.bpr.pop_loss()

# ------------------------------------------------------------------------------------------
# BPRMF (NumPy semantics, fp64)
# ------------------------------------------------------------------------------------------
class BprSgdDeviceState:
    """MFModel parameters in HBM (BPRMF_model.py:40-56), fp64."""

    def __init__(self, ctx, P, Q, b, lr, reg_bias, reg_user, reg_pos, reg_neg):
        self.ctx = ctx
        dev = ctx.device

        def own(x):
            if isinstance(x, np.ndarray):
                x = torch.from_numpy(np.ascontiguousarray(x))
            return x.to(device=dev, dtype=torch.float64).contiguous().clone()

        self.P, self.Q, self.b = own(P), own(Q), own(b)
        self.U, self.F = self.P.shape
        self.I = self.Q.shape[0]
        self._c = BprsgdState(P=self.P.data_ptr(), Q=self.Q.data_ptr(), b=self.b.data_ptr(), U=self.U, I=self.I,
                              F=self.F, lr=float(lr), reg_bias=float(reg_bias), reg_user=float(reg_user),
                              reg_pos=float(reg_pos), reg_neg=float(reg_neg))

    def apply(self, u, i, j, first=0, n=None):
        """Concurrent application of triplets [first, first+n) (Hogwild unless conflict-free)."""
        if n is None:
            n = u.numel() - first
        check(self.ctx.lib.el_bprsgd_apply(self.ctx.handle, self.ctx.stream(), C.byref(self._c),
                                           _ptr(u, torch.int32), _ptr(i, torch.int32), _ptr(j, torch.int32),
                                           int(first), int(n)), "el_bprsgd_apply")

    def apply_sequential_equivalent(self, u_host, i_host, j_host):
        """Bit-for-bit the sequential order of MFModel.train_step (BPRMF_model.py:87-89): triplets are
        grouped into dependency levels on the host, each level is one conflict-free launch."""
        order, starts = sgd_levels(u_host, i_host, j_host, self.U, self.I)
        dev = self.ctx.device
        u = torch.from_numpy(np.ascontiguousarray(u_host[order], dtype=np.int32)).to(dev)
        i = torch.from_numpy(np.ascontiguousarray(i_host[order], dtype=np.int32)).to(dev)
        j = torch.from_numpy(np.ascontiguousarray(j_host[order], dtype=np.int32)).to(dev)
        check(self.ctx.lib.el_bprsgd_apply_levels(self.ctx.handle, self.ctx.stream(), C.byref(self._c),
                                                  _ptr(u), _ptr(i), _ptr(j), starts.ctypes.data_as(C.c_void_p),
                                                  int(starts.shape[0] - 1)), "el_bprsgd_apply_levels")
        return int(starts.shape[0] - 1)


class Mf2020DeviceState:
    """MFModel of MF2020 (latent_factor_models/MF2020/MF_model.py:14-56) in HBM, fp64: user / item factors, the two bias vectors, the
    global bias.  train(samples) = MFModel.train_step (:80-113) on one batch of (user, item, rating) rows, in order."""

    def __init__(self, ctx, P, Q, bu=None, bi=None, gb=0.0, lr=0.05, reg=0.0):
        self.ctx = ctx
        dev = ctx.device
        own = lambda x: (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))).to(device=dev, dtype=torch.float64).contiguous().clone()
        self.P, self.Q = own(P), own(Q)
        self.U, self.F = int(self.P.shape[0]), int(self.P.shape[1])
        self.I = int(self.Q.shape[0])
        self.bu = own(bu) if bu is not None else torch.zeros(self.U, dtype=torch.float64, device=dev)
        self.bi = own(bi) if bi is not None else torch.zeros(self.I, dtype=torch.float64, device=dev)
        self.gb = torch.full((1,), float(gb), dtype=torch.float64, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.lr, self.reg = float(lr), float(reg)
        self._c = _lib.Mf2020State(P=self.P.data_ptr(), Q=self.Q.data_ptr(), bu=self.bu.data_ptr(), bi=self.bi.data_ptr(), gb=self.gb.data_ptr(),
                                   U=self.U, I=self.I, F=self.F, lr=self.lr, reg=self.reg)

    def train(self, samples):
        """samples: int32 [n, 3] device tensor of (user, item, rating) rows; the sum of their losses accumulates in self.loss."""
        s = samples.to(device=self.ctx.device, dtype=torch.int32).contiguous()
        check(self.ctx.lib.el_mf2020_train(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(s, torch.int32), int(s.shape[0]),
                                           _ptr(self.loss, torch.float64)), "el_mf2020_train")

    def pop_loss(self):
        v = float(self.loss.item())
        self.loss.zero_()
        return v

    def predictions(self):
        """prepare_predictions (:115-116): user_bias[:, None] + (global_bias + item_bias + P Q^T), fp64 [U, I] (small catalogues only:
        the reference materialises it too)."""
        return self.bu[:, None] + (self.gb + self.bi[None, :] + self.P @ self.Q.T)


def sgd_levels(u_host, i_host, j_host, U, I):
    """Host-only: dependency levels of a triplet sequence (el_bprsgd_levels_host)."""
    lib = _lib.load()
    u = np.ascontiguousarray(u_host, dtype=np.int32)
    i = np.ascontiguousarray(i_host, dtype=np.int32)
    j = np.ascontiguousarray(j_host, dtype=np.int32)
    n = u.shape[0]
    order = np.empty(n, dtype=np.int32)
    starts = np.empty(n + 2, dtype=np.int64)
    nl = C.c_int64()
    check(lib.el_bprsgd_levels_host(u.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p),
                                    j.ctypes.data_as(C.c_void_p), n, int(U), int(I),
                                    order.ctypes.data_as(C.c_void_p), starts.ctypes.data_as(C.c_void_p),
                                    n + 2, C.byref(nl)), "el_bprsgd_levels_host")
    return order, starts[: nl.value + 1].copy()


# ------------------------------------------------------------------------------------------
# dense layers (fp32 MFMA GEMM) and Mult-VAE
# ------------------------------------------------------------------------------------------
ACTS = {None: 0, "none": 0, "tanh": 1, "relu": 2, "sigmoid": 3}


def gemm(ctx, A, B, transA=False, transB=False, bias=None, act=None, out=None, ws=True):
    """C = act(op(A) op(B) + bias) through el_gemm_f32 (keras Dense forward / backward products).  ws=False: no workspace (the
    library then never splits K)."""
    M = A.shape[1] if transA else A.shape[0]
    K = A.shape[0] if transA else A.shape[1]
    N = B.shape[0] if transB else B.shape[1]
    Kb = B.shape[1] if transB else B.shape[0]
    if K != Kb:
        raise ValueError(f"inner dimensions differ: {K} vs {Kb}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=ctx.device)
    need = int(ctx.lib.el_gemm_ws_bytes(ctx.handle, int(M), int(N), int(K))) if ws else 0
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=ctx.device) if need else None
    check(ctx.lib.el_gemm_f32(ctx.handle, ctx.stream(), int(bool(transA)), int(bool(transB)), int(M), int(N), int(K),
                              _ptr(A, torch.float32, "A"), int(A.stride(0)), _ptr(B, torch.float32, "B"),
                              int(B.stride(0)), _ptr(out, torch.float32, "C"), int(out.stride(0)),
                              _ptr(bias, torch.float32, "bias"), ACTS[act],
                              C.c_void_p(ws.data_ptr()) if ws is not None else None, need), "el_gemm_f32")
    return out


class VaeDeviceState:
    """Variables, Adam slots and activation buffers of the Mult-VAE in HBM (multi_vae_model.py:86-109).

    weights: dict with W1[I,H] b1[H] Wm[H,L] bm[L] Wv[H,L] bv[L] W3[L,H] b3[H] W4[H,I] b4[I] (Keras Dense
    kernel layout [in, out]); the mean / log-variance heads are stored side by side as one [H, 2L] matrix."""
    ORDER = ("W1", "b1", "Wmv", "bmv", "W3", "b3", "W4", "b4")

    def __init__(self, ctx, weights, max_batch):
        self.ctx = ctx
        dev = ctx.device
        f = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev) if isinstance(x, np.ndarray) \
            else x.to(device=dev, dtype=torch.float32).contiguous().clone()
        W1, W4 = f(weights["W1"]), f(weights["W4"])
        self.I, self.H = W1.shape
        self.L = int(weights["Wm"].shape[1])
        self.Bmax = int(max_batch)
        self.dae = "Wv" not in weights          # MultiDAE: mean head only, tanh on it, no sampling / KL (multi_dae_model.py)
        if self.dae:
            wmv, bmv = f(weights["Wm"]), f(weights["bm"])
        else:
            wmv = torch.cat([f(weights["Wm"]), f(weights["Wv"])], dim=1).contiguous()
            bmv = torch.cat([f(weights["bm"]), f(weights["bv"])]).contiguous()
        t = {"W1": W1, "b1": f(weights["b1"]), "Wmv": wmv, "bmv": bmv,
             "W3": f(weights["W3"]), "b3": f(weights["b3"]), "W4": W4, "b4": f(weights["b4"])}
        self.w = [t[n] for n in self.ORDER]
        self.g = [torch.zeros_like(x) for x in self.w]
        self.m = [torch.zeros_like(x) for x in self.w]
        self.v = [torch.zeros_like(x) for x in self.w]
        B, H, L, I = self.Bmax, self.H, self.L, self.I
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.h, self.mv, self.z, self.dz, self.h2 = z(B, H), z(B, 2 * L), z(B, L), z(B, L), z(B, H)
        self.logits, self.dh2, self.dmv, self.dh, self.rnorm = z(B, I), z(B, H), z(B, 2 * L), z(B, H), z(B)
        need = max(int(ctx.lib.el_gemm_ws_bytes(ctx.handle, *mnk)) for mnk in
                   ((B, H, I), (B, L, H), (B, H, 2 * L), (H, 2 * L, B), (L, H, B), (B, 2 * L, H), (H, I, B)))
        # twice what the products (and the dW1 scratch) need: el_vae_grads then runs the weight-gradient products on the library's
        # second stream with the upper half as their workspace (el_vae.hip, vae_grads)
        w1 = ((4 * (I + 1) + 4) * 4 + 255) // 256 * 256         # dW1's index arrays: at the end of the side half
        need = max(need, w1)
        self._ws = torch.empty(2 * ((max(need, 16) + 255) // 256 * 256 + w1) + 256, dtype=torch.uint8, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.step = 0
        arr = lambda ts: (C.c_void_p * 8)(*[x.data_ptr() for x in ts])
        self._c = _lib.VaeState(I=I, H=H, L=L, Bmax=B, w=arr(self.w), g=arr(self.g), m=arr(self.m), v=arr(self.v),
                                h=self.h.data_ptr(), mv=self.mv.data_ptr(), z=self.z.data_ptr(), dz=self.dz.data_ptr(),
                                h2=self.h2.data_ptr(), logits=self.logits.data_ptr(), dh2=self.dh2.data_ptr(),
                                dmv=self.dmv.data_ptr(), dh=self.dh.data_ptr(), rnorm=self.rnorm.data_ptr(),
                                ws=self._ws.data_ptr(), ws_bytes=self._ws.numel(), dae=int(self.dae))

    def weights(self):
        """Host copies keyed like the constructor argument."""
        L = self.L
        w = {n: x.cpu().numpy() for n, x in zip(self.ORDER, self.w)}
        if self.dae:
            return {"W1": w["W1"], "b1": w["b1"], "Wm": w["Wmv"], "bm": w["bmv"], "W3": w["W3"], "b3": w["b3"],
                    "W4": w["W4"], "b4": w["b4"]}
        return {"W1": w["W1"], "b1": w["b1"], "Wm": w["Wmv"][:, :L].copy(), "Wv": w["Wmv"][:, L:].copy(),
                "bm": w["bmv"][:L].copy(), "bv": w["bmv"][L:].copy(), "W3": w["W3"], "b3": w["b3"], "W4": w["W4"],
                "b4": w["b4"]}

    def train_step(self, train_csr, rows, lr, anneal, eps=None, dropout_rate=0.0, dropout_seed=42):
        """VariationalAutoEncoder.train_step (multi_vae_model.py:125-142) on users `rows` (int32 device tensor)."""
        self.step += 1
        B = rows.numel()
        check(self.ctx.lib.el_vae_train_step(self.ctx.handle, self.ctx.stream(), C.byref(self._c), *_csr_ptrs(train_csr),
                                             _ptr(rows, torch.int32, "rows"), int(B), _ptr(eps, torch.float32, "eps"),
                                             float(anneal), float(dropout_rate), int(dropout_seed), int(self.step),
                                             float(adam_lr_t(lr, self.step)), _ptr(self.loss, torch.float64)),
              "el_vae_train_step")

    def grads(self, train_csr, rows, anneal, eps=None, dropout_rate=0.0, dropout_seed=42, n_global=None):
        """Forward + loss + backward only; the batch means run over n_global rows (multi-GPU).  dense_grads() lists the buffers a
        data-parallel caller all-reduces before apply()."""
        B = rows.numel()
        check(self.ctx.lib.el_vae_grads(self.ctx.handle, self.ctx.stream(), C.byref(self._c), *_csr_ptrs(train_csr),
                                        _ptr(rows, torch.int32, "rows"), int(B), int(B if n_global is None else n_global),
                                        _ptr(eps, torch.float32, "eps"), float(anneal), float(dropout_rate), int(dropout_seed),
                                        int(self.step + 1), _ptr(self.loss, torch.float64)), "el_vae_grads")

    def apply(self, lr):
        self.step += 1
        check(self.ctx.lib.el_vae_apply(self.ctx.handle, self.ctx.stream(), C.byref(self._c), float(adam_lr_t(lr, self.step))),
              "el_vae_apply")

    def dense_grads(self):
        return list(self.g)

    def predict(self, train_csr, rows, eps=None):
        """log_softmax(logits) [B, I] view of the activation buffer (multi_vae_model.py:144-155)."""
        B = rows.numel()
        check(self.ctx.lib.el_vae_predict(self.ctx.handle, self.ctx.stream(), C.byref(self._c), *_csr_ptrs(train_csr),
                                          _ptr(rows, torch.int32, "rows"), int(B), _ptr(eps, torch.float32, "eps")),
              "el_vae_predict")
        return self.logits[:B]

    def pop_loss(self):
        v = float(self.loss.item())
        self.loss.zero_()
        return v


# ------------------------------------------------------------------------------------------
# NeuMF / GMF
# ------------------------------------------------------------------------------------------
def pointwise_sample(ctx, pos, n, seed, first_sample=0, use_meta=True, out=None):
    """pointwise_pos_neg_sampler.Sampler.step (pointwise_pos_neg_sampler.py:26-50) on the device (use_meta: through the per-user
    sampler records -- same draws, fewer cache lines).  out: (u int32[n], i int32[n], label float32[n]) to fill."""
    if out is not None:
        u, i, y = out
    else:
        u = torch.empty(n, dtype=torch.int32, device=ctx.device)
        i = torch.empty(n, dtype=torch.int32, device=ctx.device)
        y = torch.empty(n, dtype=torch.float32, device=ctx.device)
    meta = sampler_meta(ctx, pos) if use_meta else None
    check(ctx.lib.el_pointwise_sample_meta(ctx.handle, ctx.stream(), *_csr_ptrs(pos),
                                           C.c_void_p(meta.data_ptr()) if meta is not None else None, int(pos.n_rows),
                                           int(pos.n_cols), int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_sample), int(n), _ptr(u),
                                           _ptr(i), _ptr(y)), "el_pointwise_sample_meta")
    return u, i, y


class NmfDeviceState:
    """NeuMF / GMF variables, optimiser slots and activation buffers in HBM.

    weights: dict with optional "Umf" [U,F], "Imf" [I,F] (MF branch), "Umlp" [U,E], "Imlp" [I,E], "W" list of
    [in,out] kernels and "b" list of biases (MLP branch), "hw" head weights [F + units[-1]] and optional "hb" [1]."""

    _LR_HIST = 1 << 16          # optimiser steps of lr_t history kept on the device by the deferred decay

    def __init__(self, ctx, weights, max_batch, dropout=0.0, dropout_seed=42, deferred=None, replay="exact"):
        """deferred: Keras' every-row Adam decay of the embedding tables is postponed per row and replayed bit for bit when the
        row is next needed (include/elliot_hip.h, el_nmf_state.row_last).  None = on; a data-parallel
        owner that all-reduces gtab of replicated tables turns it off (set_deferred(False); parallel.ShardedNmf does)."""
        self.ctx = ctx
        dev = ctx.device
        if replay not in ("exact", "series"):
            raise ValueError("replay must be 'exact' or 'series'")
        self.replay = replay                                         # how waiting rows are brought forward (BprmfDeviceState: the same two modes)
        self.dropout, self.dropout_seed = float(dropout), int(dropout_seed)
        self.deferred = True if deferred is None else bool(deferred)
        f = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev) if isinstance(x, np.ndarray) \
            else x.to(device=dev, dtype=torch.float32).contiguous().clone()
        self.use_mf = "Umf" in weights
        self.use_mlp = "Umlp" in weights
        self.head_bias = "hb" in weights
        names = ["Umf", "Imf", "Umlp", "Imlp"]
        self.tab = [f(weights[n]) if n in weights else None for n in names]
        ref = self.tab[0] if self.use_mf else self.tab[2]
        self.U = ref.shape[0]
        self.I = (self.tab[1] if self.use_mf else self.tab[3]).shape[0]
        self.F = self.tab[0].shape[1] if self.use_mf else 0
        self.E = self.tab[2].shape[1] if self.use_mlp else 0
        z = lambda t: None if t is None else torch.zeros_like(t)
        self.gtab, self.mtab, self.vtab, self._tab_blocks = [None] * 4, [None] * 4, [None] * 4, []
        for t in range(4):
            tb = self.tab[t]
            if tb is None:
                continue
            if tb.numel() * 4 >= (64 << 20):
                # Keras' dense Adam over an embedding table streams theta, g, m, v at once: one allocation, the distance between
                # the four arrays tuned by timing the pass (tune_table_layout -- the BPR tables' 0.82 -> 0.61 ms effect; with one
                # allocation per array the channel / bank phase of the seven streams is the allocator's luck)
                gap = tune_table_layout(ctx, int(tb.shape[0]), int(tb.shape[1]))
                (th, self.gtab[t], self.mtab[t], self.vtab[t]), blk = _strided_tables(int(tb.shape[0]), int(tb.shape[1]), 4, gap, dev)
                th.copy_(tb)
                self.tab[t] = th
                self._tab_blocks.append(blk)
                del tb
            else:
                self.gtab[t], self.mtab[t], self.vtab[t] = z(tb), z(tb), z(tb)
        self.W = [f(w) for w in weights.get("W", [])]
        self.b = [f(b) for b in weights.get("b", [])]
        self.units = [w.shape[1] for w in self.W]
        self.gW, self.mW, self.vW = [z(w) for w in self.W], [z(w) for w in self.W], [z(w) for w in self.W]
        self.gb, self.mb, self.vb = [z(b) for b in self.b], [z(b) for b in self.b], [z(b) for b in self.b]
        self.hw = f(weights["hw"])
        self.hb = f(weights["hb"]) if self.head_bias else None
        self.ghw, self.mhw, self.vhw = z(self.hw), z(self.hw), z(self.hw)
        self.ghb, self.mhb, self.vhb = z(self.hb), z(self.hb), z(self.hb)
        B = int(max_batch)
        self._alloc_activations(B)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.step = 0
        p = lambda t: None if t is None else t.data_ptr()
        a4 = lambda ts: _lib._P4(*([p(t) for t in ts] + [None] * (4 - len(ts))))
        self._c = _lib.NmfState(
            U=self.U, I=self.I, Bmax=B, F=self.F, E=self.E, n_layers=len(self.units), use_mf=int(self.use_mf),
            use_mlp=int(self.use_mlp), head_bias=int(self.head_bias),
            units=(C.c_int32 * 4)(*(self.units + [0] * (4 - len(self.units)))),
            tab=a4(self.tab), gtab=a4(self.gtab), mtab=a4(self.mtab), vtab=a4(self.vtab),
            W=a4(self.W), b=a4(self.b), gW=a4(self.gW), gb=a4(self.gb), mW=a4(self.mW), vW=a4(self.vW),
            mb=a4(self.mb), vb=a4(self.vb),
            hw=p(self.hw), hb=p(self.hb), ghw=p(self.ghw), ghb=p(self.ghb), mhw=p(self.mhw), vhw=p(self.vhw),
            mhb=p(self.mhb), vhb=p(self.vhb), X0=p(self.X0), dX0=p(self.dX0), MF=p(self.MF), dlogit=p(self.dlogit),
            act=a4(self.act), dact=a4(self.dact), ws=self._ws.data_ptr(), ws_bytes=self._ws.numel(),
            dropout=self.dropout, drop_step=0, drop_seed=self.dropout_seed & 0xFFFFFFFFFFFFFFFF)
        self._drop_calls = 0
        self._c.hist_base = 1
        self._c.replay_series = int(replay == "series")
        self._alloc_step_ws()
        if self.deferred:
            self._alloc_deferred()

    def _alloc_step_ws(self):
        """el_nmf_state.step_ws: sort buffers of the batch's (row, sample) keys, the MF factor copies, partial rows of the reductions."""
        need = int(self.ctx.lib.el_nmf_step_ws_bytes(self.ctx.handle, C.byref(self._c)))
        if need <= 0:
            raise RuntimeError("el_nmf_step_ws_bytes failed: " + (_lib.load().el_last_error() or b"").decode())
        self._step_ws = torch.empty(need + 256, dtype=torch.uint8, device=self.ctx.device)
        base = (self._step_ws.data_ptr() + 255) // 256 * 256
        self._c.step_ws, self._c.step_ws_bytes = base, need
        self._c.pre_u = self._c.pre_i = None                        # (a batch ordered ahead into the old workspace is forgotten)
        self._c.pre_n, self._c.sort_set = 0, 0

    def _alloc_deferred(self):
        dev, c = self.ctx.device, self._c
        zi = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
        self._row_last = [zi(self.U), zi(self.I)]
        if self.step:
            for t in self._row_last:
                t.fill_(self.step)                                    # switched on mid-run: every row is current
        self._lr_hist = torch.zeros(self._LR_HIST, dtype=torch.float32, device=dev)
        c.row_last = (C.c_void_p * 2)(*[t.data_ptr() for t in self._row_last])
        c.lr_hist, c.lr_hist_cap = self._lr_hist.data_ptr(), self._LR_HIST
        c.hist_base, c.claim_seq = self.step + 1, 0
        c.opt_step = c.flushed_step = self.step

    def sync(self):
        """Deferred decay: replay what is pending so that tab / mtab / vtab hold every row at the current step (no-op otherwise).
        Called by everything in this class that reads the tables; call it before reading the tensors directly."""
        if self.deferred:
            check(self.ctx.lib.el_nmf_sync_tables(self.ctx.handle, self.ctx.stream(), C.byref(self._c)), "el_nmf_sync_tables")

    def set_deferred(self, on):
        on = bool(on)
        if on == self.deferred:
            return
        if not on:
            self.sync()
            self.deferred = False
            c = self._c
            c.row_last = (C.c_void_p * 2)(None, None)
            c.lr_hist = None
            self._row_last = self._lr_hist = None
        else:
            self.deferred = True
            self._alloc_deferred()

    def _alloc_activations(self, B):
        """Activation / backward buffers for batches of up to B samples (+ the GEMM workspace sized for them)."""
        ctx, dev = self.ctx, self.ctx.device
        self.Bmax = B = int(B)
        zz = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.X0 = zz(B, 2 * self.E) if self.use_mlp else None
        self.dX0 = zz(B, 2 * self.E) if self.use_mlp else None
        self.MF = zz(B, self.F) if self.use_mf else None
        self.dlogit = zz(B)
        self.act = [zz(B, n) for n in self.units]
        self.dact = [zz(B, n) for n in self.units]
        dims = [2 * self.E] + self.units
        need = 16
        for l in range(len(self.units)):
            for mnk in ((B, dims[l + 1], dims[l]), (dims[l], dims[l + 1], B), (B, dims[l], dims[l + 1])):
                need = max(need, int(ctx.lib.el_gemm_ws_bytes(ctx.handle, *mnk)))
        # twice the products' need: the weight-gradient products of the backward pass then run on the library's second stream with the
        # upper half as their split-K workspace (el_neural.hip, nmf_grads)
        self._ws = torch.empty(2 * ((need + 255) // 256 * 256) + 256, dtype=torch.uint8, device=dev)

    def ensure_batch(self, n):
        """Grow the activation buffers to hold a batch of n samples: the reference takes ONE optimiser step per batch
        (neural_matrix_factorization_model.py:96-106), whatever its size.  Raises (torch's out-of-memory error) when HBM cannot
        hold the activations -- never splits the batch into several steps behind the caller's back."""
        if n <= self.Bmax:
            return
        self.X0 = self.dX0 = self.MF = self.dlogit = self._ws = self._step_ws = None
        self.act, self.dact = [], []
        torch.cuda.empty_cache()
        self._alloc_activations(n)
        p = lambda t: None if t is None else t.data_ptr()
        a4 = lambda ts: _lib._P4(*([p(t) for t in ts] + [None] * (4 - len(ts))))
        c = self._c
        c.Bmax, c.X0, c.dX0, c.MF, c.dlogit = self.Bmax, p(self.X0), p(self.dX0), p(self.MF), p(self.dlogit)
        c.act, c.dact, c.ws, c.ws_bytes = a4(self.act), a4(self.dact), self._ws.data_ptr(), self._ws.numel()
        self._alloc_step_ws()

    def _next_mask(self):
        self._drop_calls += 1                                        # a fresh dropout mask per gradient evaluation
        self._c.drop_step = self._drop_calls & 0x7FFFFFFF

    def weights(self):
        self.sync()
        out = {}
        for n, t in zip(["Umf", "Imf", "Umlp", "Imlp"], self.tab):
            if t is not None:
                out[n] = t.cpu().numpy()
        if self.use_mlp:
            out["W"] = [w.cpu().numpy() for w in self.W]
            out["b"] = [b.cpu().numpy() for b in self.b]
        out["hw"] = self.hw.cpu().numpy()
        if self.head_bias:
            out["hb"] = self.hb.cpu().numpy()
        return out

    def train_step(self, u, i, label, lr):
        n = u.numel()
        if n == 0:
            return                                                   # (the library takes no optimiser step on an empty batch either)
        self._margin = _PW_MARGIN
        self._next_mask()
        t = self.step + 1                                            # the counter moves only when the library has taken the step
        check(self.ctx.lib.el_nmf_train_step(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32),
                                             _ptr(i, torch.int32), _ptr(label, torch.float32), int(n), int(t),
                                             float(adam_lr_t(lr, t)), _ptr(self.loss, torch.float64)),
              "el_nmf_train_step")
        self.step = t

    def presort(self, u, i):
        """Order the batch's (embedding row, sample) keys ahead of its step, on the CURRENT stream (a side stream, while the previous
        step trains: el_nmf_presort); the train_step / grads call that follows with the same tensors skips its sort."""
        self.ensure_batch(u.numel())
        check(self.ctx.lib.el_nmf_presort(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32), _ptr(i, torch.int32),
                                          int(u.numel())), "el_nmf_presort")

    def grads(self, u, i, label, n_global=None):
        """Forward + loss + backward only (multi-GPU: the BCE mean runs over n_global samples); gradients stay in the
        state's buffers (replicated_grads() lists the ones a data-parallel caller has to all-reduce)."""
        n = u.numel()
        self._next_mask()
        if self.deferred and n_global is not None and int(n_global) != n:
            # a step shared with other ranks: the replicated tables take gradient rows of samples this rank never saw (all-reduce
            # of gtab), so which rows move is not known from the local batch -- back to the eager every-row passes
            self.set_deferred(False)
        self._batch_keep = (u, i)                                    # deferred decay: el_nmf_apply walks the batch's rows again
        check(self.ctx.lib.el_nmf_grads(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32),
                                        _ptr(i, torch.int32), _ptr(label, torch.float32), int(n),
                                        int(n if n_global is None else n_global), _ptr(self.loss, torch.float64)), "el_nmf_grads")

    def apply(self, lr):
        self._margin = _PW_MARGIN
        t = self.step + 1
        check(self.ctx.lib.el_nmf_apply(self.ctx.handle, self.ctx.stream(), C.byref(self._c), int(t),
                                        float(adam_lr_t(lr, t))), "el_nmf_apply")
        self.step = t

    def replicated_grads(self, shard="user"):
        """Gradient tensors of the variables every rank holds a full copy of: Dense layers, head, and the embedding tables
        that are NOT sharded -- the item tables with user shards (shard="user"), the user tables with item shards."""
        keep = (1, 3) if shard == "user" else (0, 2)
        out = [self.gtab[t] for t in keep if self.gtab[t] is not None]
        out += list(self.gW) + list(self.gb) + [self.ghw]
        if self.head_bias:
            out.append(self.ghb)
        return out

    def forward(self, u, i, out=None):
        n = u.numel()
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=self.ctx.device)
        check(self.ctx.lib.el_nmf_forward(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32),
                                          _ptr(i, torch.int32), int(n), _ptr(out, torch.float32)), "el_nmf_forward")
        return out

    def pop_loss(self):
        v = float(self.loss.item())
        self.loss.zero_()
        return v

    # -- full-catalogue scoring -> masked top-k --------------------------------------------------------------------
    def fused_supported(self, k):
        """True when el_nmf_score_topk takes this network and list length (three Dense layers, units <= (1024, 256, 128), ...)."""
        return bool(self.use_mlp and self.ctx.lib.el_nmf_score_supported(C.byref(self._c), int(k)))

    def score_topk_logits(self, u_start, u_stop, k, excl=None, cand=None, item_offset=0, I_local=None, items_unchanged=False, screen=None):
        """el_nmf_score_topk: the k best unmasked items of users [u_start, u_stop) by (logit desc, item asc) and their logits --
        layer 1 in its separable form, layers 2-3 and the head per (user, item) pair on fp32 MFMA tiles, selection fused.
        screen (full-catalogue calls only): layers 2-3 first on the half-precision matrix instruction with a per-pair error bound, the
        fp32 kernel on the surviving pairs -- the same lists and logit bits (EL_NMF_SCREEN in the header).  True / False force it;
        None (default) screens, and when a call had to take the unscreened
        route (the bound depends on the weights: el_nmf_screen_stats) leaves the next 15 calls unscreened before it tries again."""
        n = int(u_stop) - int(u_start)
        I_local = self.I - int(item_offset) if I_local is None else int(I_local)
        auto = screen is None
        if auto:
            screen = True
        screen = bool(screen) and cand is None
        use = screen
        if auto and screen and getattr(self, "_screen_skip", 0) > 0:
            self._screen_skip -= 1
            use = False
        need = int(self.ctx.lib.el_nmf_score_ws_bytes(self.ctx.handle, C.byref(self._c), n, I_local, int(k),
                                                      1 if cand is not None else (2 if screen else 0)))   # (auto: room for the screen
        #                                                                                   on every call, one workspace for the evaluation)
        if need == 0:
            raise _lib.ElliotHipError("el_nmf_score_topk does not take this network shape / k (NmfDeviceState.fused_supported)")
        ws = getattr(self, "_score_ws", None)
        if ws is None or ws.numel() < need:
            ws = self._score_ws = torch.empty(need, dtype=torch.uint8, device=self.ctx.device)
            items_unchanged = False
        out_idx = torch.empty((n, k), dtype=torch.int32, device=self.ctx.device)
        out_val = torch.empty((n, k), dtype=torch.float32, device=self.ctx.device)
        ep, ei = _csr_ptrs(excl)
        cp, ci = _csr_ptrs(cand)
        check(self.ctx.lib.el_nmf_score_topk(self.ctx.handle, self.ctx.stream(), C.byref(self._c), int(u_start), int(u_stop),
                                             int(item_offset), I_local, ep, ei, cp, ci, int(k), _ptr(out_idx), _ptr(out_val),
                                             (_lib.EL_TOPK_ITEMS_UNCHANGED if items_unchanged else 0) | (_lib.EL_NMF_SCREEN if use else 0),
                                             C.c_void_p(ws.data_ptr()), ws.numel()), "el_nmf_score_topk")
        if auto and use and self.screen_stats()[1]:
            self._screen_skip = 15
        return out_idx, out_val

    def screen_stats(self):
        """(pairs the exact fp32 kernel scored, a call that asked for the screen went without it) of the last score_topk_logits call."""
        pairs, fb = C.c_int64(0), C.c_int(0)
        self.ctx.lib.el_nmf_screen_stats(self.ctx.handle, C.byref(pairs), C.byref(fb))
        return int(pairs.value), bool(fb.value)

    def _dot_tables(self, items_unchanged):
        """The MF-only network (GMF; NeuMF with is_mlp_train False) is sigmoid(<Umf[u], Imf[i] * h> (+ b)): tables for the fused
        dot-product top-k kernels -- the item image Imf * h (el_gmf_item_image) and a constant bias row for the Dense(1) bias."""
        self.sync()
        img = getattr(self, "_gmf_image", None)
        if img is None or not items_unchanged:
            if img is None:
                img = self._gmf_image = (torch.empty_like(self.tab[1]),
                                         torch.empty(self.I, dtype=torch.float32, device=self.ctx.device) if self.head_bias else None)
            check(self.ctx.lib.el_gmf_item_image(self.ctx.handle, self.ctx.stream(), _ptr(self.tab[1], torch.float32),
                                                 _ptr(self.hw, torch.float32), int(self.I), int(self.F), _ptr(img[0], torch.float32)),
                  "el_gmf_item_image")
            if self.head_bias:
                img[1].copy_(self.hb.expand(self.I))
        return img

    def _pairs_topk(self, u_start, u_stop, k, excl, cand):
        """The reference's own route (index grids -> get_recs -> get_top_k, neural_matrix_factorization.py:111-119) for networks
        the fused kernel does not take: probabilities of every (user, item) pair in blocks of Bmax pairs, dense top-k."""
        nu, dev = int(u_stop) - int(u_start), self.ctx.device
        items = torch.arange(self.I, dtype=torch.int32, device=dev)
        preds = torch.empty((nu, self.I), dtype=torch.float32, device=dev)
        per = max(1, self.Bmax // self.I)
        for s in range(0, nu, per):
            e = min(s + per, nu)
            ug = torch.arange(u_start + s, u_start + e, dtype=torch.int32, device=dev).repeat_interleave(self.I)
            self.forward(ug, items.repeat(e - s), out=preds[s:e].reshape(-1))
        return dense_topk(self.ctx, preds, u_start, u_stop, k, excl=excl, cand=cand)

    def recommend(self, u_start, u_stop, k, excl=None, cand=None, items_unchanged=False):
        """get_recs + get_top_k of NeuMF / GMF for users [u_start, u_stop): (idx int32 [n, k], probabilities fp32 [n, k]).
        The fused kernels rank by the logit; sigmoid is monotone, so the ranking by probability can differ only where distinct
        logits round to one probability (tf.nn.top_k orders those by item index): the list is taken a few entries longer, linked
        (el_pwmf_link_values), re-ranked by (value desc, index asc) and cut -- the rule of PwmfDeviceState.recommend."""
        if self.use_mlp and not self.fused_supported(min(self.I, k + _PW_MARGIN)):
            return self._pairs_topk(u_start, u_stop, k, excl, cand)
        max_list = _NMF_MAX_LIST if self.use_mlp else _PW_MAX_LIST

        def score(kk):
            if self.use_mlp:
                return self.score_topk_logits(u_start, u_stop, kk, excl=excl, cand=cand, items_unchanged=items_unchanged)
            img, brow = self._dot_tables(items_unchanged)
            return score_topk(self.ctx, self.tab[0], img, brow, u_start, u_stop, kk, excl=excl, cand=cand, items_unchanged=items_unchanged)

        kk = min(self.I, k + getattr(self, "_margin", _PW_MARGIN))
        while True:
            idx, val = score(kk)
            items_unchanged = True
            check(self.ctx.lib.el_pwmf_link_values(self.ctx.handle, self.ctx.stream(), _ptr(val, torch.float32), int(val.shape[0]),
                                                   int(val.stride(0)), int(kk), _lib.EL_PW_MSE_SIGMOID, None, int(u_start)),
                  "el_pwmf_link_values")
            if kk >= self.I:
                break
            open_rows = (val[:, k - 1] == val[:, kk - 1]) & (val[:, k - 1] > float("-inf"))
            if not bool(open_rows.any()):
                break
            if kk >= max_list:
                return self._pairs_topk(u_start, u_stop, k, excl, cand)
            kk = min(self.I, max_list, k + 16 if kk < k + 16 else kk * 4)
        self._margin = kk - k
        idx, val = topk_rerank(self.ctx, idx, val)
        return idx[:, :k].contiguous(), val[:, :k].contiguous()


# ------------------------------------------------------------------------------------------
# point-wise factor models: MF, PMF, FunkSVD, LogisticMF
# ------------------------------------------------------------------------------------------
PW_KINDS = {"mse": _lib.EL_PW_MSE, "mse_sigmoid": _lib.EL_PW_MSE_SIGMOID, "logistic": _lib.EL_PW_LOGISTIC}
PW_OPTS = {"adam": _lib.EL_PW_ADAM, "adagrad": _lib.EL_PW_ADAGRAD}
PW_SIDES = {"both": _lib.EL_PW_BOTH, "items": _lib.EL_PW_ITEMS, "users": _lib.EL_PW_USERS}
# entries beyond k taken before the link: two keep k' = k + 2 <= 12 on the fastest screening policy for the usual k = 10; a row
# whose ranks k .. k' all collapse to one linked float (probability ~1e-8 per row) is redone with a 4x longer list
_PW_MARGIN, _PW_MAX_LIST, _PW_DENSE_ROWS = 2, 4032, 4096
_NMF_MAX_LIST = 448          # el_nmf_score_topk keeps k <= 448 candidates per wave in LDS
_CML_MARGIN = 16             # CML re-scores with another formula: its fp32 rounding may reorder near-ties a few ranks deep


def topk_rerank(ctx, idx, val):
    """Rows of (idx, val) re-ordered in place by (value desc, index asc) -- tf.nn.top_k's order -- after the values were
    transformed (link, CML re-score): el_topk_rerank, lists of up to 4096 entries.  Returns (idx, val)."""
    kk = idx.shape[1]
    if kk > 4096:
        raise ValueError("topk_rerank: lists of more than 4096 entries are not produced by any caller")
    check(ctx.lib.el_topk_rerank(ctx.handle, ctx.stream(), _ptr(idx, torch.int32), _ptr(val, torch.float32), int(idx.shape[0]),
                                 int(idx.stride(0)), int(kk)), "el_topk_rerank")
    return idx, val


class PwmfDeviceState:
    """Variables + optimiser slots of one point-wise factor model in HBM (include/elliot_hip.h, el_pwmf_state).

    Gu [U,F], Gi [I,F], optional biases Bu [U] / Bi [I] (both or neither); kind: "mse" (MF, FunkSVD), "mse_sigmoid"
    (PMF), "logistic" (LogisticMF, with alpha / l_w); optimizer "adam" (TF sparse-apply semantics) or "adagrad"."""

    def __init__(self, ctx, Gu, Gi, Bu=None, Bi=None, kind="mse", optimizer="adam", alpha=0.0, l_w=0.0):
        self.ctx, dev = ctx, ctx.device
        f = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
        if (Bu is None) != (Bi is None):
            raise ValueError("Bu and Bi go together")
        self.Gu, self.Gi, self.Bu, self.Bi = f(Gu), f(Gi), f(Bu), f(Bi)
        if self.Bu is not None:
            self.Bu, self.Bi = self.Bu.reshape(-1).contiguous(), self.Bi.reshape(-1).contiguous()
        self.U, self.F = self.Gu.shape
        self.I = self.Gi.shape[0]
        self.kind, self.optimizer = kind, optimizer
        self.alpha, self.l_w = float(alpha), float(l_w)
        adam = optimizer == "adam"
        z = lambda t: None if t is None else torch.zeros_like(t)
        slot = (lambda t: z(t)) if adam else (lambda t: None if t is None else torch.full_like(t, 0.1))
        names = ("Gu", "Gi", "Bu", "Bi")
        for n in names:
            t = getattr(self, n)
            if n == "Gu" and adam and self.U * self.F * 4 >= (64 << 20):
                # the dense Adam pass streams theta, g, m, v of the user table at once: one allocation, tuned distance
                gap = tune_table_layout(ctx, self.U, self.F)
                (th, self.gGu, self.mGu, self.vGu), self._user_block = _strided_tables(self.U, self.F, 4, gap, dev)
                th.copy_(self.Gu)
                self.Gu = th
                continue
            setattr(self, "g" + n, z(t))
            setattr(self, "m" + n, slot(t))
            setattr(self, "v" + n, z(t) if adam else None)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.step = 0
        self._ws = None
        p = lambda t: None if t is None else t.data_ptr()
        self._c = _lib.PwmfState(U=self.U, I=self.I, F=self.F, kind=PW_KINDS[kind], alpha=self.alpha, l_w=self.l_w,
                                 **{pre + n: p(getattr(self, pre + n)) for pre in ("", "g", "m", "v") for n in names})

    def _workspace(self, n):
        need = int(self.ctx.lib.el_pwmf_ws_bytes(int(n), int(self.U), int(self.I), int(self.F)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.ctx.device)
        return C.c_void_p(self._ws.data_ptr()), need

    def train_step(self, u, i, label, lr, side="both"):
        self.step += 1
        self._margin = _PW_MARGIN
        n = u.numel()
        ws, need = self._workspace(n)
        lr_t = adam_lr_t(lr, self.step) if self.optimizer == "adam" else float(lr)
        check(self.ctx.lib.el_pwmf_train_step(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32),
                                              _ptr(i, torch.int32), _ptr(label, torch.float32), int(n),
                                              PW_OPTS[self.optimizer], PW_SIDES[side], int(self.step), float(lr_t),
                                              _ptr(self.loss, torch.float64), ws, need), "el_pwmf_train_step")

    def train_loop(self, pos, events, B, seed, first_sample, lr, side="both"):
        """One sampler pass of an epoch -- `for batch in sampler.step(events, B): train_step(batch)` -- from a single library
        call (el_pwmf_train_loop): the same Philox stream and kernels as the per-batch calls, no host work in between."""
        steps = (int(events) + int(B) - 1) // int(B)
        if steps == 0:
            return 0
        if pos.n_rows != self.U or pos.n_cols != self.I:
            raise ValueError("train_loop: the positives' CSR must describe this state's users x items")
        adam = self.optimizer == "adam"
        lr_t = np.array([adam_lr_t(lr, self.step + 1 + k) if adam else lr for k in range(steps)], dtype=np.float32)
        ws, need = self._workspace(min(int(B), int(events)) if events < B else int(B))
        lneed = int(self.ctx.lib.el_pwmf_train_loop_ws_bytes(int(events), int(B)))
        buf = getattr(self, "_loop_ws", None)
        if buf is None or buf.numel() < lneed:
            buf = self._loop_ws = torch.empty(lneed, dtype=torch.uint8, device=self.ctx.device)
        self._margin = _PW_MARGIN
        check(self.ctx.lib.el_pwmf_train_loop(
            self.ctx.handle, self.ctx.stream(), C.byref(self._c), *_csr_ptrs(pos), C.c_void_p(sampler_meta(self.ctx, pos).data_ptr()),
            int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_sample), int(events), int(B), PW_OPTS[self.optimizer], PW_SIDES[side],
            int(self.step + 1), lr_t.ctypes.data_as(C.c_void_p), _ptr(self.loss, torch.float64), ws, need,
            C.c_void_p(buf.data_ptr()), lneed), "el_pwmf_train_loop")
        self.step += steps
        return steps

    def grads(self, u, i, label, n_global=None, side="both"):
        """Forward + loss + gradient sums only (multi-GPU: a batch-mean loss runs over n_global samples); the accumulators of
        `side` are complete on return -- item_grads() lists the replicated ones a data-parallel caller all-reduces."""
        n = u.numel()
        ws, need = self._workspace(n)
        check(self.ctx.lib.el_pwmf_grads(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32),
                                         _ptr(i, torch.int32), _ptr(label, torch.float32), int(n),
                                         int(n if n_global is None else n_global), PW_SIDES[side], _ptr(self.loss, torch.float64),
                                         ws, need), "el_pwmf_grads")

    def apply(self, lr, side="both", advance=True):
        if advance:
            self.step += 1
            self._margin = _PW_MARGIN
        lr_t = adam_lr_t(lr, self.step) if self.optimizer == "adam" else float(lr)
        check(self.ctx.lib.el_pwmf_apply(self.ctx.handle, self.ctx.stream(), C.byref(self._c), PW_OPTS[self.optimizer],
                                         PW_SIDES[side], int(self.step), float(lr_t)), "el_pwmf_apply")

    def item_grads(self):
        return [t for t in (self.gGi, self.gBi) if t is not None]

    def forward(self, u, i, out=None):
        n = u.numel()
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=self.ctx.device)
        check(self.ctx.lib.el_pwmf_forward(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32),
                                           _ptr(i, torch.int32), int(n), _ptr(out, torch.float32)), "el_pwmf_forward")
        return out

    def pop_loss(self):
        v = float(self.loss.item())
        self.loss.zero_()
        return v

    def weights(self):
        out = {"Gu": self.Gu.cpu().numpy(), "Gi": self.Gi.cpu().numpy(), "step": self.step}
        if self.Bu is not None:
            out["Bu"], out["Bi"] = self.Bu.cpu().numpy(), self.Bi.cpu().numpy()
        for pre in ("m", "v"):
            for n in ("Gu", "Gi", "Bu", "Bi"):
                t = getattr(self, pre + n)
                if t is not None:
                    out[pre + n] = t.cpu().numpy()
        return out

    def load(self, d):
        for key, val in d.items():
            if key == "step":
                self.step = int(val)
            elif getattr(self, key, None) is not None:
                getattr(self, key).copy_(torch.from_numpy(np.asarray(val, dtype=np.float32)).reshape(getattr(self, key).shape))

    # -- full-catalogue scores -> masked top-k ------------------------------------------------------------------
    def _link(self, vals, k, u_start):
        check(self.ctx.lib.el_pwmf_link_values(self.ctx.handle, self.ctx.stream(), _ptr(vals, torch.float32),
                                               int(vals.shape[0]), int(vals.stride(0)), int(k), PW_KINDS[self.kind],
                                               _ptr(self.Bu, torch.float32), int(u_start)), "el_pwmf_link_values")

    def recommend(self, u_start, u_stop, k, excl=None, cand=None):
        """get_recs + get_top_k of the reference models (e.g. matrix_factorization.py:99-113 +
        matrix_factorization_model.py:100-101) for users [u_start, u_stop): the fused scoring kernel ranks by
        Bi[i] + <Gu[u], Gi[i]>; the model's score is link(that + Bu[u]) with a monotone link, so the ranking is the same
        except where distinct raw scores round to ONE linked float -- tf.nn.top_k orders those by item index.  The list is
        therefore taken `margin` entries longer, linked, re-ranked by (value desc, index asc) and cut; a row whose k-th
        value still equals its last one is redone with a longer list (and, past 4032 entries, from dense scores)."""
        plain = self.kind != "mse_sigmoid" and self.Bu is None
        if plain:
            return score_topk(self.ctx, self.Gu, self.Gi, None, u_start, u_stop, k, excl=excl, cand=cand)
        kk = min(self.I, k + getattr(self, "_margin", _PW_MARGIN))
        while True:
            idx, val = score_topk(self.ctx, self.Gu, self.Gi, self.Bi, u_start, u_stop, kk, excl=excl, cand=cand)
            self._link(val, kk, u_start)
            if kk >= self.I:
                break
            open_rows = (val[:, k - 1] == val[:, kk - 1]) & (val[:, k - 1] > float("-inf"))
            if not bool(open_rows.any()):
                break
            if kk >= _PW_MAX_LIST:
                return self._recommend_dense(u_start, u_stop, k, excl, cand)
            kk = min(self.I, _PW_MAX_LIST, k + 16 if kk < k + 16 else kk * 4)
        self._margin = kk - k          # small-score models (untrained: everything near link(0)) collapse often: the next blocks
        #                                of this evaluation start with the list length that resolved this one (reset by train_step)
        idx, val = topk_rerank(self.ctx, idx, val)
        return idx[:, :k].contiguous(), val[:, :k].contiguous()

    def _recommend_dense(self, u_start, u_stop, k, excl, cand):
        out_i, out_v = [], []
        for s in range(u_start, u_stop, _PW_DENSE_ROWS):
            e = min(s + _PW_DENSE_ROWS, u_stop)
            preds = gemm(self.ctx, self.Gu[s:e], self.Gi, transB=True, bias=self.Bi)
            self._link(preds, self.I, s)
            bi, bv = dense_topk(self.ctx, preds, s, e, k, excl=excl, cand=cand)
            out_i.append(bi)
            out_v.append(bv)
        return torch.cat(out_i), torch.cat(out_v)


# ------------------------------------------------------------------------------------------
# Collaborative Metric Learning
# ------------------------------------------------------------------------------------------
class CmlDeviceState(BprmfDeviceState):
    """CML_model's variables (Gu, Gi, Bi) + Adam slots: the BPR state with the metric-learning step and scoring."""

    def __init__(self, ctx, Gu, Gi, Bi):
        super().__init__(ctx, Gu, Gi, Bi, optimizer="adam_tf_dense", compact_user_grads=False)
        self._cml_ws = None
        self._items2 = None

    def train_step(self, u, i, j, lr, l_w, l_b, margin):
        self.step += 1
        B = u.numel()
        need = self._cml_workspace(B, B)
        check(self.ctx.lib.el_cml_train_step(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32),
                                             _ptr(i, torch.int32), _ptr(j, torch.int32), int(B), float(l_w), float(l_b),
                                             float(margin), int(self.step), float(adam_lr_t(lr, self.step)),
                                             _ptr(self.loss, torch.float64), C.c_void_p(self._cml_ws.data_ptr()), need),
              "el_cml_train_step")
        self._items2 = None

    def _cml_workspace(self, B, B_all):
        need = int(self.ctx.lib.el_cml_ws_bytes(int(B), int(B_all), int(self.U), int(self.I), int(self.F)))
        if self._cml_ws is None or self._cml_ws.numel() < need:
            self._cml_ws = torch.empty(need, dtype=torch.uint8, device=self.ctx.device)
        return need

    def forward_de(self, u, i, j, l_w, l_b):
        """Phase 1 of the multi-GPU step: the rank's D_a, E_a (device float [B] each) + its share of the regulariser."""
        B = u.numel()
        D = torch.empty(B, dtype=torch.float32, device=self.ctx.device)
        E = torch.empty(B, dtype=torch.float32, device=self.ctx.device)
        check(self.ctx.lib.el_cml_forward(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32),
                                          _ptr(i, torch.int32), _ptr(j, torch.int32), int(B), float(l_w), float(l_b), _ptr(D), _ptr(E),
                                          _ptr(self.loss, torch.float64)), "el_cml_forward")
        return D, E

    def grads_de(self, u, i, j, l_w, l_b, margin, D, E, D_all, E_all):
        """Phase 2: the rank's triplets against the gathered D / E of the global batch; gradients stay in the accumulators."""
        B, B_all = u.numel(), D_all.numel()
        need = self._cml_workspace(B, B_all)
        check(self.ctx.lib.el_cml_grads(self.ctx.handle, self.ctx.stream(), C.byref(self._c), _ptr(u, torch.int32),
                                        _ptr(i, torch.int32), _ptr(j, torch.int32), int(B), float(l_w), float(l_b), float(margin),
                                        _ptr(D, torch.float32), _ptr(E, torch.float32), _ptr(D_all, torch.float32),
                                        _ptr(E_all, torch.float32), int(B_all), _ptr(self.loss, torch.float64),
                                        C.c_void_p(self._cml_ws.data_ptr()), need), "el_cml_grads")
        self._items2 = None

    # -- optimiser alone (multi-GPU step): the split form overlaps the item-gradient all-reduce with the user rows' update
    def item_grads(self):
        return [self.item_grad_flat]

    def begin_step(self):
        self.step += 1

    def _apply(self, c_state, lr):
        check(self.ctx.lib.el_bprmf_apply(self.ctx.handle, self.ctx.stream(), C.byref(c_state), float(lr), int(self.opt), int(self.step),
                                          float(adam_lr_t(lr, self.step))), "el_bprmf_apply")
        self._items2 = None

    def apply_users(self, lr):
        if not hasattr(self, "_c_users"):
            clone = lambda c: type(c).from_buffer_copy(c)
            self._c_users, self._c_items = clone(self._c), clone(self._c)
            self._c_users.I = 0
            self._c_items.U = 0
        self._apply(self._c_users, lr)

    def apply_items(self, lr):
        self._apply(self._c_items, lr)

    def _item_side(self):
        if self._items2 is None:
            Gi2, Bi2 = torch.empty_like(self.Gi), torch.empty_like(self.Bi)
            check(self.ctx.lib.el_cml_prepare_items(self.ctx.handle, self.ctx.stream(), _ptr(self.Gi, torch.float32),
                                                    _ptr(self.Bi, torch.float32), int(self.I), int(self.F), _ptr(Gi2), _ptr(Bi2)),
                  "el_cml_prepare_items")
            self._items2 = (Gi2, Bi2)
        return self._items2

    def rescore(self, idx, u_start):
        val = torch.empty(idx.shape, dtype=torch.float32, device=self.ctx.device)
        check(self.ctx.lib.el_cml_rescore(self.ctx.handle, self.ctx.stream(), _ptr(self.Gu, torch.float32),
                                          _ptr(self.Gi, torch.float32), _ptr(self.Bi, torch.float32), int(self.F),
                                          _ptr(idx, torch.int32), int(idx.shape[0]), int(idx.stride(0)), int(idx.shape[1]),
                                          int(u_start), _ptr(val)), "el_cml_rescore")
        return val

    def recommend(self, u_start, u_stop, k, excl=None, cand=None):
        """CML_model.predict + get_top_k (CML_model.py:97-106): the fused kernel ranks by (Bi - |Gi|^2) + <Gu, 2 Gi>, which is
        the score plus the per-user constant |Gu[u]|^2; the k + 16 best are re-scored with the reference's own formula
        -sum (u - i)^2 + b_i and re-ranked by (value desc, index asc), so values and order are those of the direct
        evaluation (the two formulas differ by fp32 rounding only; a difference that straddles rank k + 16 is not seen)."""
        Gi2, Bi2 = self._item_side()
        kk = min(self.I, k + _CML_MARGIN)
        idx, raw = score_topk(self.ctx, self.Gu, Gi2, Bi2, u_start, u_stop, kk, excl=excl, cand=cand)
        val = self.rescore(idx, u_start)
        val = torch.where(raw == float("-inf"), raw, val)           # -inf padding (masked items) stays padding
        idx, val = topk_rerank(self.ctx, idx, val)
        return idx[:, :k].contiguous(), val[:, :k].contiguous()
