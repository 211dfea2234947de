"""ctypes loader of the plain-C oracle (oracle/c/el_oracle.c).  TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libel_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "c", "el_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "c"), "-B"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _csr(csr):
    if csr is None:
        return None, None, None, None
    indptr = np.ascontiguousarray(csr[0], dtype=np.int64)
    idx = np.ascontiguousarray(csr[1], dtype=np.int32)
    if idx.size == 0:
        idx = np.zeros(1, np.int32)
    return indptr, idx, _p(indptr), _p(idx)


def scores_f32(Gu, Gi, Bi, u0, u1):
    """BPRMF_batch_model.predict (BPRMF_batch_model.py:83-84) as fp32 fma chains."""
    lib = load()
    Gu = np.ascontiguousarray(Gu, np.float32)
    Gi = np.ascontiguousarray(Gi, np.float32)
    Bi = None if Bi is None else np.ascontiguousarray(Bi, np.float32)
    I, F = Gi.shape
    out = np.empty((u1 - u0, I), np.float32)
    lib.orc_scores_f32(_p(Gu), _p(Gi), _p(Bi), C.c_int64(u0), C.c_int64(u1), C.c_int64(I), C.c_int32(F), _p(out))
    return out


def score_topk_f32(Gu, Gi, Bi, u_start, u_stop, k, excl=None, cand=None, item_offset=0):
    """predict + get_top_k (BPRMF_batch_model.py:83-88) with the CSR mask semantics of
    recommender_utils_mixin.py:102-109.  excl/cand: (indptr, indices) tuples or None."""
    lib = load()
    Gu = np.ascontiguousarray(Gu, np.float32)
    Gi = np.ascontiguousarray(Gi, np.float32)
    Bi = None if Bi is None else np.ascontiguousarray(Bi, np.float32)
    I, F = Gi.shape
    n = u_stop - u_start
    oi = np.empty((n, k), np.int32)
    ov = np.empty((n, k), np.float32)
    e = _csr(excl)
    c = _csr(cand)
    lib.orc_score_topk_f32(_p(Gu), _p(Gi), _p(Bi), C.c_int64(u_start), C.c_int64(u_stop), C.c_int64(item_offset),
                           C.c_int64(I), C.c_int32(F), e[2], e[3], c[2], c[3], C.c_int32(k), _p(oi), _p(ov))
    return oi, ov


def score_topk_f64(P, Q, b, u_start, u_stop, k, excl=None, cand=None, item_offset=0):
    """MFModel.get_user_predictions (BPRMF_model.py:70-85) with fp64 fma chains."""
    lib = load()
    P = np.ascontiguousarray(P, np.float64)
    Q = np.ascontiguousarray(Q, np.float64)
    b = None if b is None else np.ascontiguousarray(b, np.float64)
    I, F = Q.shape
    n = u_stop - u_start
    oi = np.empty((n, k), np.int32)
    ov = np.empty((n, k), np.float64)
    e = _csr(excl)
    c = _csr(cand)
    lib.orc_score_topk_f64(_p(P), _p(Q), _p(b), C.c_int64(u_start), C.c_int64(u_stop), C.c_int64(item_offset),
                           C.c_int64(I), C.c_int32(F), e[2], e[3], c[2], c[3], C.c_int32(k), _p(oi), _p(ov))
    return oi, ov


def topk_rows_f32(scores, u_start, k, excl=None, cand=None, item_offset=0):
    """get_top_k (BPRMF_batch_model.py:87-88; multi_vae_model.py:158-159) on a dense block."""
    lib = load()
    scores = np.ascontiguousarray(scores, np.float32)
    n, I = scores.shape
    oi = np.empty((n, k), np.int32)
    ov = np.empty((n, k), np.float32)
    e = _csr(excl)
    c = _csr(cand)
    lib.orc_topk_rows_f32(_p(scores), C.c_int64(I), C.c_int64(u_start), C.c_int64(n), C.c_int64(item_offset),
                          C.c_int64(I), e[2], e[3], c[2], c[3], C.c_int32(k), _p(oi), _p(ov))
    return oi, ov


def nmf_logits(w, users, i0=0, i1=None):
    """NeuMF logits of `users` x items [i0, i1) with the pinned summation order (orc_nmf_logits); w: the weight dict of
    oracle/neumf.py (Umf, Imf, Umlp, Imlp, W[3], b[3], hw, optional hb).  Returns float32 [len(users), i1 - i0]."""
    lib = load()
    f = lambda a: np.ascontiguousarray(a, np.float32)
    Umlp, Imlp = f(w["Umlp"]), f(w["Imlp"])
    has_mf = "Umf" in w
    Umf, Imf = (f(w["Umf"]), f(w["Imf"])) if has_mf else (np.zeros((1, 1), np.float32), np.zeros((1, 1), np.float32))
    F = Umf.shape[1] if has_mf else 0
    E = Umlp.shape[1]
    W = [f(x) for x in w["W"]]
    b = [f(x) for x in w["b"]]
    assert len(W) == 3
    hw = f(w["hw"])
    hb = f(w["hb"]) if "hb" in w else None
    users = np.ascontiguousarray(users, np.int64)
    if i1 is None:
        i1 = Imlp.shape[0]
    out = np.empty((users.shape[0], i1 - i0), np.float32)
    lib.orc_nmf_logits(_p(Umf), _p(Imf), _p(Umlp), _p(Imlp), C.c_int32(F), C.c_int32(E), _p(W[0]), _p(b[0]), C.c_int32(W[0].shape[1]),
                       _p(W[1]), _p(b[1]), C.c_int32(W[1].shape[1]), _p(W[2]), _p(b[2]), C.c_int32(W[2].shape[1]), _p(hw), _p(hb),
                       _p(users), C.c_int64(users.shape[0]), C.c_int64(i0), C.c_int64(i1), _p(out))
    return out
