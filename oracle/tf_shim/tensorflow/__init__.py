"""A `tensorflow` stand-in over torch-CPU autograd.  TEST INFRASTRUCTURE -- it is NOT TensorFlow.

Why it exists: tensorflow==2.3.2 (the reference's requirements.txt:3) cannot be installed in the build container, so nothing there
could EXECUTE the reference's TensorFlow model files.  This package implements exactly the symbols that
    elliot/recommender/latent_factor_models/BPRMF_batch/BPRMF_batch_model.py
    elliot/recommender/autoencoders/vae/multi_vae_model.py
    elliot/recommender/neural/NeuMF/neural_matrix_factorization_model.py
    elliot/recommender/neural/GeneralizedMF/generalized_matrix_factorization_model.py
import and call, so that oracle/gen_golden_tfshim.py can import those files UNMODIFIED, run their train_step / predict / get_recs /
get_top_k on injected weights and write fixtures (tests/golden/tfshim_*.npz).  What that buys: the FILE-LEVEL algebra of the
reference (which gathered tensors enter the L2 term, the /10 on the negative bias, the KL mean over batch AND latent, the concat
order in front of NeuMF's head, what is squeezed where) is executed from the reference's own source instead of being read.
What it does NOT buy: every LIBRARY behaviour below (clip gradient at the bound, Keras Adam's sparse / dense apply, top_k's tie
rule, l2_normalize's epsilon, BinaryCrossentropy's clip) is the same recalled reading that oracle/tf_clauses.py lists -- routed
through those switches -- so fixtures made with this package pin nothing about TensorFlow itself.  oracle/gen_golden_tf.py, run
under the real tensorflow==2.3.2, stays the real pin ("parity unpinned" until then).

Tensors are thin wrappers around torch tensors (fp32 arithmetic in the order the model files write it)."""
import contextlib
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(_HERE))))       # the repo root: oracle.tf_clauses
from oracle import tf_clauses  # noqa: E402

__version__ = "0.0-shim (torch %s; NOT TensorFlow)" % torch.__version__
float32, float64, int32, int64 = torch.float32, torch.float64, torch.int32, torch.int64
bool = torch.bool  # noqa: A001


class TensorShape(tuple):
    @property
    def rank(self):
        return len(self)

    def as_list(self):
        return list(self)


def _raw(x, like=None):
    """-> torch tensor (python floats take the dtype of `like`, default float32)."""
    if isinstance(x, (Tensor, Variable)):
        return x.t
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x))
    if isinstance(x, (list, tuple)):
        return torch.stack([_raw(e, like) for e in x])
    if isinstance(x, (float, np.floating)):
        return torch.tensor(float(x), dtype=like.dtype if (like is not None and like.dtype.is_floating_point) else torch.float32)
    if isinstance(x, (int, np.integer)) and not isinstance(x, (np.bool_,)):
        if like is not None and like.dtype.is_floating_point:
            return torch.tensor(float(x), dtype=like.dtype)
        return torch.tensor(int(x), dtype=torch.int64)
    return torch.as_tensor(x)


class Tensor:
    def __init__(self, t):
        self.t = t

    def numpy(self):
        return self.t.detach().cpu().numpy().copy()           # (a snapshot: the model's variables are updated in place)

    @property
    def shape(self):
        return TensorShape(self.t.shape)

    @property
    def dtype(self):
        return self.t.dtype

    def __len__(self):
        return self.t.shape[0]

    def __getitem__(self, k):
        return Tensor(self.t[k])

    def __neg__(self):
        return Tensor(-self.t)

    def __add__(self, o):
        return Tensor(self.t + _raw(o, self.t))

    def __radd__(self, o):
        return Tensor(_raw(o, self.t) + self.t)

    def __sub__(self, o):
        return Tensor(self.t - _raw(o, self.t))

    def __rsub__(self, o):
        return Tensor(_raw(o, self.t) - self.t)

    def __mul__(self, o):
        return Tensor(self.t * _raw(o, self.t))

    def __rmul__(self, o):
        return Tensor(_raw(o, self.t) * self.t)

    def __truediv__(self, o):
        return Tensor(self.t / _raw(o, self.t))

    def __rtruediv__(self, o):
        return Tensor(_raw(o, self.t) / self.t)

    def __float__(self):
        return float(self.t)


class Variable(Tensor):
    """tf.Variable: a trainable leaf.  Rows read through embedding_lookup inside a tape make its gradient an IndexedSlices."""
    _uid = {}

    def __init__(self, initial_value=None, name=None, dtype=None, shape=None, trainable=True):
        t = _raw(initial_value)
        if dtype is not None:
            t = t.to(dtype)
        if shape is not None and tuple(shape) != tuple(t.shape):
            raise ValueError(f"tf.Variable: initial value of shape {tuple(t.shape)} does not match shape={tuple(shape)} "
                             f"(the shim refuses what TensorFlow's behaviour for is not known here)")
        super().__init__(t.detach().clone().requires_grad_(t.dtype.is_floating_point))
        base = name or "Variable"
        n = Variable._uid.get(base, 0)
        Variable._uid[base] = n + 1
        self.name = (base if n == 0 else f"{base}_{n}") + ":0"
        self.trainable = trainable

    def assign(self, value):
        with torch.no_grad():
            self.t.copy_(_raw(value).to(self.t.dtype).reshape(self.t.shape))
        return self


class IndexedSlices:
    def __init__(self, values, indices, dense_shape):
        self.values, self.indices, self.dense_shape = values, indices, dense_shape


# --------------------------------------------------------------------------------------------------------------------- tape
_TAPES = []


class GradientTape:
    def __enter__(self):
        self.reads = []                          # (variable, flat indices, list collecting d loss / d gathered rows)
        _TAPES.append(self)
        return self

    def __exit__(self, *exc):
        _TAPES.remove(self)
        return False

    def gradient(self, target, sources):
        srcs = list(sources)
        sparse = {id(v) for v, _, _ in self.reads}
        grads = torch.autograd.grad(_raw(target), [v.t for v in srcs], allow_unused=True)      # runs the hooks of the gathers
        out = []
        for v, g in zip(srcs, grads):
            if id(v) in sparse:
                idx = torch.cat([i for vv, i, _ in self.reads if vv is v])
                vals = torch.cat([got[0].reshape(len(i), *v.t.shape[1:]) for vv, i, got in self.reads if vv is v])
                out.append(IndexedSlices(vals, idx, tuple(v.t.shape)))
            else:
                out.append(None if g is None else Tensor(g))
        return out


def _lookup(var, ids):
    """params[ids]; inside a tape the gradient w.r.t. the gathered rows is recorded per occurrence (-> IndexedSlices)."""
    idx = _raw(ids).to(torch.int64)
    rows = var.t[idx]
    if _TAPES and isinstance(var, Variable):
        got = []
        rows.register_hook(lambda g, got=got: got.append(g))
        _TAPES[-1].reads.append((var, idx.reshape(-1), got))
    return Tensor(rows)


# --------------------------------------------------------------------------------------------------------------------- ops
def function(fn=None, **kw):
    """tf.function: the model files' methods run eagerly here."""
    if fn is None:
        return lambda f: f
    return fn


def constant(value, dtype=None):
    t = _raw(value)
    return Tensor(t.to(dtype) if dtype is not None else t)


def zeros(shape, dtype=float32):
    return Tensor(torch.zeros(shape if isinstance(shape, (list, tuple)) else (int(shape),), dtype=dtype))


def shape(x):
    return TensorShape(_raw(x).shape)


def squeeze(x, axis=None):
    t = _raw(x)
    return Tensor(t.squeeze() if axis is None else t.squeeze(axis))


def reduce_sum(x, axis=None):
    t = _raw(x)
    return Tensor(t.sum() if axis is None else t.sum(dim=axis))


def reduce_mean(x, axis=None, keepdims=False):
    t = _raw(x)
    return Tensor(t.mean() if axis is None else t.mean(dim=axis, keepdim=keepdims))


def multiply(a, b):
    x = _raw(a)
    return Tensor(x * _raw(b, x))


def stack(values, axis=0):
    return Tensor(torch.stack([_raw(v) for v in values], dim=axis))


def split(value, num_or_size_splits, axis=0):
    return [Tensor(t) for t in torch.split(_raw(value), list(num_or_size_splits) if isinstance(num_or_size_splits, (list, tuple))
                                           else _raw(value).shape[axis] // int(num_or_size_splits), dim=axis)]


class SparseTensor:
    """tf.SparseTensor(indices [nnz, 2], values, dense_shape): held as a SciPy CSR (row-major, column order inside a row -- the order
    tf.sparse.sparse_dense_matmul walks a row's entries in)."""

    def __init__(self, indices, values, dense_shape):
        import scipy.sparse as sp
        idx = np.asarray(indices)
        self.m = sp.csr_matrix((np.asarray(_raw(values).detach().numpy() if not isinstance(values, np.ndarray) else values, dtype=np.float32),
                                (idx[:, 0].astype(np.int64).ravel(), idx[:, 1].astype(np.int64).ravel())), shape=tuple(int(d) for d in dense_shape))
        self.m.sort_indices()


class _Sparse:
    @staticmethod
    def sparse_dense_matmul(sp_a, b):
        # (no gradient path: the model files that use it assign the result to a variable -- Variable.assign cuts the tape)
        dense = _raw(b).detach().numpy().astype(np.float32)
        return Tensor(torch.from_numpy(np.ascontiguousarray((sp_a.m @ dense).astype(np.float32))))


sparse = _Sparse()


def exp(x):
    return Tensor(torch.exp(_raw(x)))


def square(x):
    t = _raw(x)
    return Tensor(t * t)


def concat(values, axis):
    return Tensor(torch.cat([_raw(v) for v in values], dim=axis))


def matmul(a, b, transpose_a=False, transpose_b=False):
    x, y = _raw(a), _raw(b)
    return Tensor((x.t() if transpose_a else x) @ (y.t() if transpose_b else y))


def where(cond, x, y):
    xt = _raw(x)
    return Tensor(torch.where(_raw(cond), xt, _raw(y, xt).to(xt.dtype)))


class _ClipByValue(torch.autograd.Function):
    """[clause clip_gradient_inclusive_at_bound] Minimum / Maximum pass the gradient where lo <= x <= hi (or strictly inside)."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        inside = ((x >= lo) & (x <= hi)) if tf_clauses.get("clip_gradient_inclusive_at_bound") else ((x > lo) & (x < hi))
        ctx.save_for_backward(inside)
        return torch.minimum(torch.maximum(x, lo), hi)

    @staticmethod
    def backward(ctx, g):
        (inside,) = ctx.saved_tensors
        return g * inside.to(g.dtype), None, None


def clip_by_value(x, lo, hi):
    t = _raw(x)
    return Tensor(_ClipByValue.apply(t, _raw(lo, t).to(t.dtype), _raw(hi, t).to(t.dtype)))


class _Random:
    @staticmethod
    def set_seed(seed):
        torch.manual_seed(int(seed))


random = _Random()


class _GlorotUniform:
    def __call__(self, shape, dtype=float32):
        shape = tuple(int(s) for s in shape)
        fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[0], shape[0])
        lim = (6.0 / (fan_in + fan_out)) ** 0.5
        return Tensor((torch.rand(shape, dtype=dtype) * 2 - 1) * lim)


class _GlorotNormal:
    def __call__(self, shape, dtype=float32):
        shape = tuple(int(s) for s in shape)
        std = (2.0 / (shape[0] + shape[1])) ** 0.5 / 0.87962566
        return Tensor(torch.nn.init.trunc_normal_(torch.empty(shape, dtype=dtype), std=std, a=-2 * std, b=2 * std))


class _Initializers:
    GlorotUniform = _GlorotUniform
    GlorotNormal = _GlorotNormal


initializers = _Initializers()


class _TopK(tuple):
    values = property(lambda s: s[0])
    indices = property(lambda s: s[1])


class _NN:
    @staticmethod
    def embedding_lookup(params, ids):
        return _lookup(params, ids)

    @staticmethod
    def softplus(x):
        return Tensor(torch.nn.functional.softplus(_raw(x), beta=1.0, threshold=1e9))

    @staticmethod
    def l2_loss(x):
        t = _raw(x)
        s = (t * t).sum()
        return Tensor(s / 2 if tf_clauses.get("l2_loss_is_half_sum_of_squares") else s)

    @staticmethod
    def log_softmax(x, axis=-1):
        return Tensor(torch.log_softmax(_raw(x), dim=axis))

    @staticmethod
    def top_k(x, k=1, sorted=True):  # noqa: A002
        """[clauses top_k_ties_lower_index_first, top_k_pads_with_lowest_masked_indices]"""
        a = _raw(x).detach().cpu().numpy()
        n = a.shape[-1]
        idx = np.broadcast_to(np.arange(n), a.shape)
        tie = idx if tf_clauses.get("top_k_ties_lower_index_first") else -idx
        with np.errstate(invalid="ignore"):
            key = np.where(np.isnan(a), -np.inf, a)
        order = np.lexsort((tie, -key), axis=-1)[..., :k]              # value descending, then the tie rule
        vals = np.take_along_axis(a, order, axis=-1)
        return _TopK((Tensor(torch.from_numpy(np.ascontiguousarray(vals))), Tensor(torch.from_numpy(order.astype(np.int32)))))


def _leaky_relu(x, alpha=0.2):
    return Tensor(torch.nn.functional.leaky_relu(_raw(x), negative_slope=alpha))


def _nn_dropout(x, rate, **kw):
    if float(rate) != 0.0:
        raise NotImplementedError("tf.nn.dropout with rate > 0 draws TensorFlow's random stream: the shim refuses what it cannot reproduce")
    return Tensor(_raw(x))


def _l2_normalize(x, axis=None, epsilon=1e-12):
    t = _raw(x)
    return Tensor(t * torch.rsqrt(torch.clamp((t * t).sum(dim=axis, keepdim=True), min=epsilon)))


_NN.leaky_relu = staticmethod(_leaky_relu)
_NN.dropout = staticmethod(_nn_dropout)
_NN.l2_normalize = staticmethod(_l2_normalize)
nn = _NN()


# --------------------------------------------------------------------------------------------------------------------- Adam
class _Adam:
    """Keras OptimizerV2 Adam (TF 2.3) as oracle/tf_clauses.py reads it: IndexedSlices de-duplicated by a segment sum, then
    _resource_apply_sparse (every row of m, v decays and every row of the variable moves); dense gradients through the fused
    ApplyAdam kernel (delta form).  beta1 .9, beta2 .999, epsilon 1e-7."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0
        self._slots = {}

    def get_slot(self, var, name):
        m, v = self._slots[id(var)]
        return Tensor(m if name == "m" else v)

    def _slot(self, var):
        if id(var) not in self._slots:
            self._slots[id(var)] = (torch.zeros_like(var.t), torch.zeros_like(var.t))
        return self._slots[id(var)]

    def apply_gradients(self, grads_and_vars):
        f = np.float32
        t = self.iterations + 1
        b1p, b2p = np.power(f(self.b1), f(t)), np.power(f(self.b2), f(t))
        folded = tf_clauses.get("adam_epsilon_outside_sqrt_with_folded_bias_correction")
        lr_t = float(f(self.lr) * np.sqrt(f(1) - b2p) / (f(1) - b1p))
        b1, b2, eps = f(self.b1), f(self.b2), f(self.eps)
        omb1, omb2 = float(tf_clauses.one_minus(self.b1)), float(tf_clauses.one_minus(self.b2))     # [clause adam_one_minus_beta_in_fp32]
        with torch.no_grad():
            for g, var in grads_and_vars:
                if g is None:
                    continue
                m, v = self._slot(var)
                th = var.t
                if isinstance(g, IndexedSlices):
                    if tf_clauses.get("indexed_slices_duplicates_summed_before_apply"):
                        uniq, inv = torch.unique(g.indices, return_inverse=True)
                        vals = torch.zeros((len(uniq),) + tuple(g.values.shape[1:]), dtype=g.values.dtype).index_add_(0, inv, g.values)
                        parts = [(uniq, vals)]
                    else:
                        parts = [(g.indices[n:n + 1], g.values[n:n + 1]) for n in range(len(g.indices))]
                    for idx, vals in parts:
                        if tf_clauses.get("adam_sparse_apply_moves_all_rows"):
                            m.mul_(float(b1))
                            m.index_add_(0, idx, vals * omb1)
                            v.mul_(float(b2))
                            v.index_add_(0, idx, (vals * vals) * omb2)
                            rows = slice(None)
                        else:
                            m[idx] = m[idx] * float(b1) + vals * omb1
                            v[idx] = v[idx] * float(b2) + (vals * vals) * omb2
                            rows = idx
                        self._move(th, m, v, rows, lr_t, folded, b1p, b2p)
                else:
                    gt = _raw(g)
                    if tf_clauses.get("adam_dense_uses_delta_form"):
                        m.add_((gt - m) * omb1)
                        v.add_((gt * gt - v) * omb2)
                    else:
                        m.mul_(float(b1)).add_(gt * omb1)
                        v.mul_(float(b2)).add_((gt * gt) * omb2)
                    self._move(th, m, v, slice(None), lr_t, folded, b1p, b2p)
        self.iterations = t

    def _move(self, th, m, v, rows, lr_t, folded, b1p, b2p):
        if folded:
            th[rows] = th[rows] - (lr_t * m[rows]) / (torch.sqrt(v[rows]) + self.eps)
        else:
            th[rows] = th[rows] - self.lr * (m[rows] / float(1 - b1p)) / (torch.sqrt(v[rows] / float(1 - b2p)) + self.eps)


class _Optimizers:
    Adam = _Adam


optimizers = _Optimizers()

from . import keras  # noqa: E402,F401
