#!/usr/bin/env python
"""Generate the TensorFlow pins of the oracle: tests/golden/tf_*.npz.  TEST INFRASTRUCTURE.

NOT runnable in the build container (TensorFlow 2.3.2 -- requirements.txt:3 of the reference -- cannot be installed there: no
network).  Run it once on any box that has the reference checkout and that TensorFlow:

    pip install tensorflow==2.3.2 numpy
    python oracle/gen_golden_tf.py --reference /path/to/elliot [--out tests/golden]

It loads the reference's UNMODIFIED model classes BY FILE PATH (importing `elliot.recommender` as a package would pull every model
and hyperopt in; the four files below import only tensorflow / numpy):
    elliot/recommender/latent_factor_models/BPRMF_batch/BPRMF_batch_model.py   BPRMF_batch_model
    elliot/recommender/autoencoders/vae/multi_vae_model.py                     VariationalAutoEncoder
    elliot/recommender/neural/NeuMF/neural_matrix_factorization_model.py       NeuralMatrixFactorizationModel
    elliot/recommender/neural/GeneralizedMF/generalized_matrix_factorization_model.py   GeneralizedMatrixFactorizationModel
injects fixed weights (`Variable.assign` / `set_weights`: TF's GlorotUniform stream is not what is being pinned), runs
`train_step` / `predict` / `get_recs` / `get_top_k` on fixed batches, and stores inputs and TensorFlow's outputs.  The batches are
built to hit every clause oracle/tf_clauses.py lists: duplicate indices inside a batch, rows no batch touches (Keras' sparse Adam
moves them all the same), a score difference below the -80 clip bound AND one exactly on it, exact ties and rows with fewer than k
candidates in top_k, probabilities that saturate BinaryCrossentropy's 1e-7 clip, an all-zero row for l2_normalize.
Only the Mult-VAE's normal draw is replaced (keras.backend.random_normal -> a stored constant): its graph-level seed stream
cannot be reproduced outside TensorFlow, and the model draws even at inference (multi_vae_model.py:57-64).

tests/test_tf_pins.py consumes the files when present (and reports "parity unpinned" as xfail when they are not).
"""
import argparse
import importlib.util
import os
import sys

import numpy as np


def load_by_path(reference, rel, name):
    path = os.path.join(reference, rel)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def adam_slots(opt, var):
    return opt.get_slot(var, "m").numpy(), opt.get_slot(var, "v").numpy()


# ---------------------------------------------------------------------------------------------------------------------
def gen_bprmf_batch(ref, out, tf, prefix="tf_"):
    mod = load_by_path(ref, "elliot/recommender/latent_factor_models/BPRMF_batch/BPRMF_batch_model.py", "ref_bprmf_batch_model")
    U, I, F, lr, l_w, l_b = 40, 30, 8, 0.001, 0.1, 0.001
    rs = np.random.RandomState(0)
    Gu = rs.normal(scale=0.3, size=(U, F)).astype(np.float32)
    Gi = rs.normal(scale=0.3, size=(I, F)).astype(np.float32)
    Bi = rs.normal(scale=0.1, size=I).astype(np.float32)
    # clip cases: user 0 / items 0, 1 give x_ui - x_uj << -80; user 1 / items 2, 3 sit EXACTLY on -80 (Gu[1] = e_0, Gi[2] = 0,
    # Gi[3] = 80 e_0, biases 0: difference = 0 - 80)
    Gu[0] = 0
    Gu[0, 0] = 30.0
    Gi[0] = 0
    Gi[0, 0] = -2.0
    Gi[1] = 0
    Gi[1, 0] = 2.0
    Gu[1] = 0
    Gu[1, 0] = 1.0
    Gi[2] = 0
    Gi[3] = 0
    Gi[3, 0] = 80.0
    Bi[0:4] = 0
    m = mod.BPRMF_batch_model(F, lr, l_w, l_b, U, I, 42)
    m.Gu.assign(Gu)
    m.Gi.assign(Gi)
    m.Bi.assign(Bi)
    res = {"U": U, "I": I, "F": F, "lr": lr, "l_w": l_w, "l_b": l_b, "Gu_init": Gu, "Gi_init": Gi, "Bi_init": Bi}
    B = 24
    for step in range(3):
        u = rs.randint(2, U - 5, size=B)                    # users U-5.. are never touched
        i = rs.randint(4, I - 4, size=B)                    # items I-4.. are never touched
        j = rs.randint(4, I - 4, size=B)
        u[:6] = u[6]                                        # duplicates: one user six times, one item as pos AND neg
        i[:3] = i[3]
        j[10] = i[3]
        u[20], i[20], j[20] = 0, 0, 1                       # far below the clip bound: no gradient through the softplus
        u[21], i[21], j[21] = 1, 2, 3                       # exactly -80
        batch = (u.reshape(-1, 1).astype(np.int64), i.reshape(-1, 1).astype(np.int64), j.reshape(-1, 1).astype(np.int64))
        loss = m.train_step(batch)
        res[f"u{step}"], res[f"i{step}"], res[f"j{step}"] = u, i, j
        res[f"loss{step}"] = np.float32(loss.numpy())
        res[f"Gu{step}"], res[f"Gi{step}"], res[f"Bi{step}"] = m.Gu.numpy(), m.Gi.numpy(), m.Bi.numpy()
        for name, var in (("Gu", m.Gu), ("Gi", m.Gi), ("Bi", m.Bi)):
            res[f"m{name}{step}"], res[f"v{name}{step}"] = adam_slots(m.optimizer, var)
    # predict + get_top_k: the trained weights, a mask with a nearly-full row (fewer than k candidates) and an empty row
    preds = m.predict(0, U).numpy()
    mask = rs.rand(U, I) < 0.7
    mask[5] = False
    mask[5, [3, 9]] = True
    mask[6] = False
    k = 7
    v, ix = m.get_top_k(tf.constant(preds), tf.constant(mask), k=k)
    res.update({"predict": preds, "mask": mask, "k": k, "topk_val": v.numpy(), "topk_idx": ix.numpy()})
    # exact ties: a hand-made score block
    tied = np.zeros((4, 12), np.float32)
    tied[0] = [1, 3, 3, 2, 3, 0, 0, 3, 1, 1, 2, 2]
    tied[1] = 5.0
    tied[2] = np.arange(12)[::-1]
    tied[3] = [np.inf, -np.inf, 0, 0, np.inf, 1, 1, 1, -1, -1, 0, 2]
    tmask = np.ones((4, 12), bool)
    tmask[1, ::2] = False
    v, ix = m.get_top_k(tf.constant(tied), tf.constant(tmask), k=6)
    res.update({"tied": tied, "tied_mask": tmask, "tied_val": v.numpy(), "tied_idx": ix.numpy()})
    np.savez_compressed(os.path.join(out, f"{prefix}bprmf_batch.npz"), **res)
    print(f"wrote {prefix}bprmf_batch.npz")


# ---------------------------------------------------------------------------------------------------------------------
def gen_multivae(ref, out, tf, prefix="tf_"):
    mod = load_by_path(ref, "elliot/recommender/autoencoders/vae/multi_vae_model.py", "ref_multi_vae_model")
    I, H, L, B, lr = 60, 24, 8, 16, 0.001
    rs = np.random.RandomState(1)
    eps = rs.normal(size=(B, L)).astype(np.float32)
    # the only replacement: the normal draw of Sampling.call (:28) returns the stored constant
    mod.tf.keras.backend.random_normal = lambda shape, **kw: tf.constant(eps)
    m = mod.VariationalAutoEncoder(I, H, L, lr, 0.0, 0.01, 42)
    x = (rs.rand(B, I) < 0.15).astype(np.float32)
    x[3] = 0                                               # all-zero row: l2_normalize's epsilon
    m(tf.constant(x), training=False)                      # build
    names = ["W1", "b1", "Wm", "bm", "Wv", "bv", "W3", "b3", "W4", "b4"]
    tw = m.trainable_weights                               # encoder: proj, mean, log_var; decoder: proj, output
    w0 = {n: (rs.normal(scale=0.2, size=v.shape)).astype(np.float32) for n, v in zip(names, tw)}
    for n, v in zip(names, tw):
        v.assign(w0[n])
    res = {"I": I, "H": H, "L": L, "B": B, "lr": lr, "x": x, "eps": eps, "names": np.array(names),
           "var_names": np.array([v.name for v in tw])}
    res.update({f"{n}_0": w0[n] for n in names})
    logits, kl = m(tf.constant(x), training=False)
    res["logits_0"], res["kl_0"] = logits.numpy(), np.float32(kl.numpy())
    res["predict_0"] = m.predict(tf.constant(x)).numpy()
    for step, anneal in enumerate((0.0, 0.1, 0.2)):
        loss = m.train_step(tf.constant(x), anneal)
        res[f"anneal{step}"] = np.float32(anneal)
        res[f"loss{step}"] = np.float32(loss.numpy())
        for n, v in zip(names, tw):
            res[f"{n}_{step + 1}"] = v.numpy()
    np.savez_compressed(os.path.join(out, f"{prefix}multivae.npz"), **res)
    print(f"wrote {prefix}multivae.npz")


# ---------------------------------------------------------------------------------------------------------------------
def gen_neumf(ref, out, tf, prefix="tf_"):
    mod = load_by_path(ref, "elliot/recommender/neural/NeuMF/neural_matrix_factorization_model.py", "ref_neumf_model")
    U, I, F, lr = 30, 25, 8, 0.002
    units = (4 * F, 2 * F, F)
    rs = np.random.RandomState(2)
    m = mod.NeuralMatrixFactorizationModel(U, I, F, F, units, 0.0, True, True, lr, 42)
    u0 = np.arange(4, dtype=np.int64)
    m((tf.constant(u0), tf.constant(u0)), training=False)  # build the Dense layers
    tw = m.trainable_weights
    res = {"U": U, "I": I, "F": F, "lr": lr, "units": np.array(units), "var_names": np.array([v.name for v in tw])}
    for n, v in enumerate(tw):
        w = rs.normal(scale=0.5 if v.shape.rank == 2 and v.shape[0] in (U, I) else 0.3, size=v.shape).astype(np.float32)
        v.assign(w)
        res[f"w{n}_0"] = w
    B = 32
    for step in range(3):
        u = rs.randint(0, U - 4, size=B).astype(np.int64)   # the last four users / items are never touched
        i = rs.randint(0, I - 4, size=B).astype(np.int64)
        y = rs.randint(0, 2, size=B).astype(np.float32)
        u[:5] = u[5]
        i[:4] = i[4]
        loss = m.train_step((tf.constant(u), tf.constant(i), tf.constant(y)))
        res[f"u{step}"], res[f"i{step}"], res[f"y{step}"] = u, i, y
        res[f"loss{step}"] = np.float32(loss.numpy())
        for n, v in enumerate(tw):
            res[f"w{n}_{step + 1}"] = v.numpy()
    ug, ig = np.meshgrid(np.arange(U, dtype=np.int64), np.arange(I, dtype=np.int64), indexing="ij")
    res["recs"] = m.get_recs((tf.constant(ug), tf.constant(ig))).numpy()
    # saturation of BinaryCrossentropy's clip: scale the head so that probabilities hit 0 / 1 in fp32
    tw[-2].assign(tw[-2].numpy() * 200.0)
    u = np.arange(8, dtype=np.int64)
    y = np.array([0, 1, 0, 1, 1, 0, 1, 0], np.float32)
    out_p = m((tf.constant(u), tf.constant(u)), training=False).numpy()
    loss = m.train_step((tf.constant(u), tf.constant(u), tf.constant(y)))
    res.update({"sat_u": u, "sat_y": y, "sat_p": out_p, "sat_loss": np.float32(loss.numpy())})
    for n, v in enumerate(tw):
        res[f"w{n}_sat"] = v.numpy()
    np.savez_compressed(os.path.join(out, f"{prefix}neumf.npz"), **res)
    print(f"wrote {prefix}neumf.npz")


def gen_gmf(ref, out, tf, prefix="tf_"):
    mod = load_by_path(ref, "elliot/recommender/neural/GeneralizedMF/generalized_matrix_factorization_model.py", "ref_gmf_model")
    U, I, F, lr = 20, 18, 6, 0.002
    rs = np.random.RandomState(3)
    m = mod.GeneralizedMatrixFactorizationModel(U, I, F, True, lr, 42)
    tw = m.trainable_weights
    res = {"U": U, "I": I, "F": F, "lr": lr, "var_names": np.array([v.name for v in tw])}
    for n, v in enumerate(tw):
        w = rs.normal(scale=0.5, size=v.shape).astype(np.float32)
        v.assign(w)
        res[f"w{n}_0"] = w
    for step in range(2):
        u = rs.randint(0, U - 3, size=16).astype(np.int64)
        i = rs.randint(0, I - 3, size=16).astype(np.int64)
        y = rs.randint(0, 2, size=16).astype(np.float32)
        u[:3] = u[3]
        loss = m.train_step((tf.constant(u), tf.constant(i), tf.constant(y)))
        res[f"u{step}"], res[f"i{step}"], res[f"y{step}"] = u, i, y
        res[f"loss{step}"] = np.float32(loss.numpy())
        for n, v in enumerate(tw):
            res[f"w{n}_{step + 1}"] = v.numpy()
    ug, ig = np.meshgrid(np.arange(U, dtype=np.int64), np.arange(I, dtype=np.int64), indexing="ij")
    res["recs"] = m.get_recs((tf.constant(ug), tf.constant(ig))).numpy()
    np.savez_compressed(os.path.join(out, f"{prefix}gmf.npz"), **res)
    print(f"wrote {prefix}gmf.npz")


def gen_lightgcn(ref, out, tf, prefix="tf_"):
    """graph_based/lightgcn/LightGCN_model.py executed unmodified: Laplacian from the oracle's restatement of LightGCN.py:96-118 (itself
    pinned to the reference's method: tests/golden/lightgcn_laplacian.npz), injected non-zero tables (the reference initialises them to
    zero, :63-65, where every gradient vanishes), three train steps with repeated users / items, n_layers = 1 and 2, a prediction block."""
    import scipy.sparse as sp
    from oracle import lightgcn as ol
    if not hasattr(np, "mat"):
        np.mat = np.asmatrix                                     # (:59 uses np.mat, removed in NumPy 2)
    mod = load_by_path(ref, "elliot/recommender/graph_based/lightgcn/LightGCN_model.py", "ref_lightgcn_model")
    U, I, F, lr, l_w = 26, 19, 8, 0.005, 0.1
    rs = np.random.RandomState(11)
    R = sp.random(U, I, density=0.2, format="csr", random_state=rs, dtype=np.float32)
    R.data[:] = 1.0
    adj, lap = ol.create_adj_mat(R, U, I)
    res = {"U": U, "I": I, "F": F, "lr": lr, "l_w": l_w, "R_indptr": R.indptr.astype(np.int64), "R_indices": R.indices.astype(np.int32)}
    for n_layers in (1, 2):
        m = mod.LightGCNModel(U, I, lr, F, l_w, n_layers, 2, adj, lap, 42)
        Gu0 = rs.normal(scale=0.3, size=(U, F)).astype(np.float32)
        Gi0 = rs.normal(scale=0.3, size=(I, F)).astype(np.float32)
        m.Gu.assign(Gu0)
        m.Gi.assign(Gi0)
        res[f"L{n_layers}_Gu0"], res[f"L{n_layers}_Gi0"] = Gu0, Gi0
        for step in range(3):
            u = rs.randint(0, U, size=24).astype(np.int64)
            i = rs.randint(0, I, size=24).astype(np.int64)
            j = rs.randint(0, I, size=24).astype(np.int64)
            u[:4] = u[4]
            i[:3] = i[5]
            loss = m.train_step((tf.constant(u.reshape(-1, 1)), tf.constant(i.reshape(-1, 1)), tf.constant(j.reshape(-1, 1))))
            res[f"L{n_layers}_u{step}"], res[f"L{n_layers}_i{step}"], res[f"L{n_layers}_j{step}"] = u, i, j
            res[f"L{n_layers}_loss{step}"] = np.float32(loss.numpy())
            res[f"L{n_layers}_Gu{step + 1}"], res[f"L{n_layers}_Gi{step + 1}"] = m.Gu.numpy().copy(), m.Gi.numpy().copy()
        res[f"L{n_layers}_preds"] = m.predict(3, 11).numpy()
    np.savez_compressed(os.path.join(out, f"{prefix}lightgcn.npz"), **res)
    print(f"wrote {prefix}lightgcn.npz")


def gen_ngcf(ref, out, tf, prefix="tf_"):
    """graph_based/ngcf/NGCF_model.py executed unmodified: two propagation layers (8 -> 12 -> 4: widths in multiples of 4, what the device kernels take), no node dropout, message dropout 0
    (TensorFlow's stream is not reproducible), injected layer-0 embeddings, three train steps, a prediction block."""
    import scipy.sparse as sp
    from oracle import lightgcn as ol
    if not hasattr(np, "mat"):
        np.mat = np.asmatrix
    mod = load_by_path(ref, "elliot/recommender/graph_based/ngcf/NGCF_model.py", "ref_ngcf_model")
    U, I, F, lr, l_w, ws = 24, 17, 8, 0.005, 0.05, [12, 4]
    rs = np.random.RandomState(21)
    R = sp.random(U, I, density=0.25, format="csr", random_state=rs, dtype=np.float32)
    R.data[:] = 1.0
    adj, lap = ol.create_adj_mat(R, U, I)
    m = mod.NGCFModel(U, I, lr, F, l_w, ws, len(ws), [], [0.0, 0.0], 2, adj, lap, 42)
    W = F + sum(ws)
    Gu0, Gi0 = np.zeros((U, W), np.float32), np.zeros((I, W), np.float32)
    Gu0[:, :F] = rs.normal(scale=0.3, size=(U, F))
    Gi0[:, :F] = rs.normal(scale=0.3, size=(I, F))
    m.Gu.assign(Gu0)
    m.Gi.assign(Gi0)
    res = {"U": U, "I": I, "F": F, "lr": lr, "l_w": l_w, "weight_size": np.array(ws), "R_indptr": R.indptr.astype(np.int64),
           "R_indices": R.indices.astype(np.int32), "Gu0": Gu0, "Gi0": Gi0}
    for k in range(len(ws)):
        for nm in ("W_1", "b_1", "W_2", "b_2"):
            v = m.GraphLayers[f"{nm}_{k}"]
            w = rs.normal(scale=0.4, size=v.shape).astype(np.float32)
            v.assign(w)
            res[f"{nm}_{k}_0"] = w
    for step in range(3):
        u = rs.randint(0, U, size=20).astype(np.int64)
        i = rs.randint(0, I, size=20).astype(np.int64)
        j = rs.randint(0, I, size=20).astype(np.int64)
        u[:3] = u[3]
        loss = m.train_step((tf.constant(u.reshape(-1, 1)), tf.constant(i.reshape(-1, 1)), tf.constant(j.reshape(-1, 1))))
        res[f"u{step}"], res[f"i{step}"], res[f"j{step}"] = u, i, j
        res[f"loss{step}"] = np.float32(loss.numpy())
        res[f"Gu{step + 1}"], res[f"Gi{step + 1}"] = m.Gu.numpy().copy(), m.Gi.numpy().copy()
        for k in range(len(ws)):
            for nm in ("W_1", "b_1", "W_2", "b_2"):
                res[f"{nm}_{k}_{step + 1}"] = m.GraphLayers[f"{nm}_{k}"].numpy().copy()
    res["preds"] = m.predict(2, 9).numpy()
    np.savez_compressed(os.path.join(out, f"{prefix}ngcf.npz"), **res)
    print(f"wrote {prefix}ngcf.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="path of the sisinflab/elliot checkout (v0.3.1)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    ap.add_argument("--only", default="bprmf_batch,multivae,neumf,gmf,lightgcn,ngcf")
    args = ap.parse_args()
    os.environ.setdefault("CUDA_VISIBLE_DEVICES", "-1")          # the reference's default device (namespace_model.py:74)
    try:
        import tensorflow as tf
    except ImportError:
        sys.exit("gen_golden_tf.py needs tensorflow==2.3.2 (the reference's requirements.txt:3); it is not installed here")
    print("tensorflow", tf.__version__, "(the reference pins 2.3.2)")
    os.makedirs(args.out, exist_ok=True)
    todo = set(args.only.split(","))
    if "bprmf_batch" in todo:
        gen_bprmf_batch(args.reference, args.out, tf)
    if "multivae" in todo:
        gen_multivae(args.reference, args.out, tf)
    if "neumf" in todo:
        gen_neumf(args.reference, args.out, tf)
    if "gmf" in todo:
        gen_gmf(args.reference, args.out, tf)
    if "lightgcn" in todo:
        gen_lightgcn(args.reference, args.out, tf)
    if "ngcf" in todo:
        gen_ngcf(args.reference, args.out, tf)
    with open(os.path.join(args.out, "tf_VERSION.txt"), "w") as f:
        f.write(f"tensorflow {tf.__version__}\nnumpy {np.__version__}\n")


if __name__ == "__main__":
    main()
