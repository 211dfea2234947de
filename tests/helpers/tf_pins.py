"""Checks of the oracle against TensorFlow's own outputs (tests/golden/tf_*.npz, written by oracle/gen_golden_tf.py on a box with
tensorflow==2.3.2 and the reference checkout).  Each function takes the loaded fixture and asserts, clause by clause
(oracle/tf_clauses.py), that the NumPy / C restatements reproduce what the reference's unmodified model classes did.

Tolerances: losses 1e-5 relative (fp32 reduction order inside TF is free); variables after k Adam steps 2e-6 absolute plus, for
at most 0.2 % of the entries, up to 2 lr (Adam's m / (sqrt(v) + eps) is discontinuous where a gradient sum cancels to ~0);
top-k index lists exactly.
"""
import numpy as np

from oracle import bprmf_batch as ob
from oracle import cref
from oracle import multi_vae as ov
from oracle import neumf as on
from oracle import topk as ot


def _close_vars(name, got, exp, lr):
    err = np.abs(np.asarray(got, np.float64) - np.asarray(exp, np.float64))
    assert err.max() <= 2 * lr + 1e-6 and (err > 2e-6).mean() <= 2e-3, (name, float(err.max()), float((err > 2e-6).mean()))


def check_bprmf_batch(d):
    lr, l_w, l_b = float(d["lr"]), float(d["l_w"]), float(d["l_b"])
    orc = ob.BPRMFBatchOracle(d["Gu_init"], d["Gi_init"], d["Bi_init"], lr, l_w, l_b)
    U, I = int(d["U"]), int(d["I"])
    for s in range(3):
        u, i, j = d[f"u{s}"], d[f"i{s}"], d[f"j{s}"]
        loss = orc.train_step((u, i, j))
        exp = float(d[f"loss{s}"])
        assert abs(loss - exp) <= 1e-5 * abs(exp), ("loss", s, loss, exp)           # A.3: clip, softplus sum, l2_loss = sum/2
        for name in ("Gu", "Gi", "Bi"):
            _close_vars(f"{name} after step {s + 1}", getattr(orc, name), d[f"{name}{s}"], lr)
            m, v = orc.slots[name]
            # A.4: the slots of EVERY row (the untouched rows decay too), duplicates summed before the update
            assert np.abs(m - d[f"m{name}{s}"]).max() <= 1e-6 * max(1.0, np.abs(d[f"m{name}{s}"]).max()), ("m", name, s)
            assert np.abs(v - d[f"v{name}{s}"]).max() <= 1e-6 * max(1.0, np.abs(d[f"v{name}{s}"]).max()), ("v", name, s)
    # rows no batch ever touches have zero gradient and zero slots: they stay where they were under either Adam variant
    assert np.array_equal(d["Gu2"][U - 5:], d["Gu_init"][U - 5:]) and np.array_equal(d["Gi2"][I - 4:], d["Gi_init"][I - 4:])
    # a row touched at step 1 and NOT at step 2 keeps moving at step 2 (m decays, theta -= lr_t m / (sqrt v + eps)): the "every row
    # moves" clause of Keras' sparse apply, read off TensorFlow's own tables (Gu<s> = the table after step s + 1)
    idle = sorted(set(d["u0"].tolist()) - set(d["u1"].tolist()))
    if idle:
        assert np.abs(d["Gu1"][idle[0]] - d["Gu0"][idle[0]]).max() > 0, "TensorFlow left a row alone that the batch did not touch: lazy Adam?"
    # predict (:83-84): TF's matmul against the pinned fma chain (any fp32 order agrees to F 2^-23 |u||i|) and against fp64
    Gu, Gi, Bi = d["Gu2"], d["Gi2"], d["Bi2"]
    chain = cref.scores_f32(Gu, Gi, Bi, 0, U)
    bound = Gu.shape[1] * 2.0 ** -23 * np.linalg.norm(Gu, axis=1)[:, None] * np.linalg.norm(Gi, axis=1)[None, :] + 1e-7
    assert (np.abs(chain.astype(np.float64) - d["predict"]) <= 2 * bound).all()
    # get_top_k (:87-88): tie rule and -inf padding (A.6) -- on TF's OWN score block, so no summation-order freedom is left
    k = int(d["k"])
    val, idx = ot.get_top_k(d["predict"], d["mask"], k)
    assert np.array_equal(idx, d["topk_idx"]) and np.array_equal(val, d["topk_val"])
    val, idx = ot.get_top_k(d["tied"], d["tied_mask"], int(d["tied_idx"].shape[1]))
    assert np.array_equal(idx, d["tied_idx"]) and np.array_equal(val, d["tied_val"])
    # ... and the C oracle (the checker of the device kernels) on the same block: mask as a CSR of the masked-out items
    mask = d["mask"]
    ip = np.concatenate([[0], np.cumsum((~mask).sum(1))]).astype(np.int64)
    ix = np.concatenate([np.flatnonzero(~mask[r]) for r in range(U)]).astype(np.int32)
    ci, cv = cref.topk_rows_f32(d["predict"], 0, k, excl=(ip, ix))
    assert np.array_equal(ci, d["topk_idx"]) and np.array_equal(cv, d["topk_val"])


def check_multivae(d):
    names = [str(n) for n in d["names"]]
    w0 = {n: d[f"{n}_0"] for n in names}
    lr = float(d["lr"])
    x, eps = d["x"], d["eps"]
    c = ov.forward(w0, x, eps)
    assert np.abs(c["logits"] - d["logits_0"]).max() <= 2e-5 * max(1.0, np.abs(d["logits_0"]).max())     # l2_normalize eps row incl.
    kl = -0.5 * np.mean(c["lv"] - c["mu"] ** 2 - np.exp(c["lv"]) + 1)
    assert abs(kl - float(d["kl_0"])) <= 1e-5 * max(1.0, abs(float(d["kl_0"])))                           # mean over batch AND latent
    assert np.abs(ov.log_softmax(c["logits"]) - d["predict_0"]).max() <= 5e-5
    orc = ov.MultiVAEOracle(w0, lr)
    for s in range(3):
        loss = orc.train_step(x, eps, float(d[f"anneal{s}"]))
        exp = float(d[f"loss{s}"])
        assert abs(loss - exp) <= 2e-5 * abs(exp), ("loss", s, loss, exp)
        for n in names:
            _close_vars(f"{n} after step {s + 1}", orc.w[n], d[f"{n}_{s + 1}"], lr)


def _nmf_weights(d, step):
    """TensorFlow's trainable_weights -> the oracle's dict, by variable name (Layer.trainable_weights lists a model's own
    variables before its sub-layers' -- GMF's `h` comes first there)."""
    vn = [str(x) for x in d["var_names"]]
    get = lambda n: np.array(d[f"w{n}_{step}"])
    w = {}
    dense = []
    for n, name in enumerate(vn):
        a = get(n)
        if "U_MF" in name or "U_GMF" in name:
            w["Umf"] = a
        elif "I_MF" in name or "I_GMF" in name:
            w["Imf"] = a
        elif "U_MLP" in name:
            w["Umlp"] = a
        elif "I_MLP" in name:
            w["Imlp"] = a
        elif name.split("/")[-1].startswith("h"):
            w["hw"] = a[:, 0].copy()
        else:
            dense.append(a)
    if dense:                                   # Dense kernels / biases in layer order; the last pair is predict_layer
        ks = [a for a in dense if a.ndim == 2]
        bs = [a for a in dense if a.ndim == 1]
        w["W"], w["b"] = ks[:-1], bs[:-1]
        w["hw"], w["hb"] = ks[-1][:, 0].copy(), bs[-1]
    return w


def _check_pointwise(d, steps, lr):
    orc = on.NeuMFOracle(_nmf_weights(d, 0), lr)
    for s in range(steps):
        loss = orc.train_step(d[f"u{s}"], d[f"i{s}"], d[f"y{s}"])
        exp = float(d[f"loss{s}"])
        assert abs(loss - exp) <= 2e-5 * abs(exp), ("loss", s, loss, exp)
        tfw = _nmf_weights(d, s + 1)
        for k, v in orc.w.items():
            pairs = zip(v, tfw[k]) if isinstance(v, list) else [(v, tfw[k])]
            for a, b in pairs:
                _close_vars(f"{k} after step {s + 1}", a, b, lr)
    U, I = int(d["U"]), int(d["I"])
    ug, ig = np.meshgrid(np.arange(U), np.arange(I), indexing="ij")
    p = on.forward(_nmf_weights(d, steps), ug.reshape(-1), ig.reshape(-1))["p"].reshape(U, I)
    assert np.abs(p - d["recs"]).max() <= 2e-6
    return orc


def check_neumf(d):
    lr = float(d["lr"])
    _check_pointwise(d, 3, lr)
    # the fused scoring kernel's checker against TensorFlow's get_recs: logit(orc_nmf_logits) -> sigmoid
    w = _nmf_weights(d, 3)
    L = cref.nmf_logits(w, np.arange(int(d["U"])))
    assert np.abs(1.0 / (1.0 + np.exp(-L.astype(np.float64))) - d["recs"]).max() <= 2e-6
    # BinaryCrossentropy at saturation (A.8): loss and the (zero) gradient through clipped probabilities
    ws = _nmf_weights(d, 3)
    ws["hw"] = ws["hw"] * np.float32(200.0)
    c = on.forward(ws, d["sat_u"], d["sat_u"])
    assert np.abs(c["p"] - d["sat_p"].reshape(-1)).max() <= 1e-6
    assert abs(on.bce(c["p"], d["sat_y"]) - float(d["sat_loss"])) <= 2e-5 * abs(float(d["sat_loss"]))


def check_gmf(d):
    _check_pointwise(d, 2, float(d["lr"]))
