"""Stand-in fixtures with the layout oracle/gen_golden_tf.py writes, produced by the ORACLE itself: they let the consuming checks
(tests/helpers/tf_pins.py) run in CI without TensorFlow -- proving the checks execute and are self-consistent, nothing about TF."""
import os

import numpy as np

from oracle import bprmf_batch as ob
from oracle import multi_vae as ov
from oracle import neumf as on
from oracle import topk as ot


def make(out):
    rs = np.random.RandomState(0)
    # ---- BPRMF_batch
    U, I, F, lr, l_w, l_b = 40, 30, 8, 0.001, 0.1, 0.001
    Gu = rs.normal(scale=0.3, size=(U, F)).astype(np.float32)
    Gi = rs.normal(scale=0.3, size=(I, F)).astype(np.float32)
    Bi = rs.normal(scale=0.1, size=I).astype(np.float32)
    Gu[0] = 0; Gu[0, 0] = 30.0; Gi[0] = 0; Gi[0, 0] = -2.0; Gi[1] = 0; Gi[1, 0] = 2.0
    Gu[1] = 0; Gu[1, 0] = 1.0; Gi[2] = 0; Gi[3] = 0; Gi[3, 0] = 80.0; Bi[0:4] = 0
    orc = ob.BPRMFBatchOracle(Gu, Gi, Bi, lr, l_w, l_b)
    res = {"U": U, "I": I, "F": F, "lr": lr, "l_w": l_w, "l_b": l_b, "Gu_init": Gu, "Gi_init": Gi, "Bi_init": Bi}
    for s in range(3):
        u, i, j = rs.randint(2, U - 5, 24), rs.randint(4, I - 4, 24), rs.randint(4, I - 4, 24)
        u[:6] = u[6]; i[:3] = i[3]; j[10] = i[3]
        u[20], i[20], j[20] = 0, 0, 1
        u[21], i[21], j[21] = 1, 2, 3
        res[f"u{s}"], res[f"i{s}"], res[f"j{s}"] = u, i, j
        res[f"loss{s}"] = np.float32(orc.train_step((u, i, j)))
        for n in ("Gu", "Gi", "Bi"):
            res[f"{n}{s}"] = getattr(orc, n).copy()
            res[f"m{n}{s}"], res[f"v{n}{s}"] = orc.slots[n][0].copy(), orc.slots[n][1].copy()
    preds = orc.predict(0, U).astype(np.float32)
    mask = rs.rand(U, I) < 0.7
    mask[5] = False; mask[5, [3, 9]] = True; mask[6] = False
    v, ix = ot.get_top_k(preds, mask, 7)
    res.update({"predict": preds, "mask": mask, "k": 7, "topk_val": v, "topk_idx": ix.astype(np.int32)})
    tied = np.zeros((4, 12), np.float32)
    tied[0] = [1, 3, 3, 2, 3, 0, 0, 3, 1, 1, 2, 2]; tied[1] = 5.0; tied[2] = np.arange(12)[::-1]
    tied[3] = [np.inf, -np.inf, 0, 0, np.inf, 1, 1, 1, -1, -1, 0, 2]
    tmask = np.ones((4, 12), bool); tmask[1, ::2] = False
    v, ix = ot.get_top_k(tied, tmask, 6)
    res.update({"tied": tied, "tied_mask": tmask, "tied_val": v, "tied_idx": ix.astype(np.int32)})
    np.savez_compressed(os.path.join(out, "tf_bprmf_batch.npz"), **res)
    # ---- Mult-VAE
    I, H, L, B, lr = 60, 24, 8, 16, 0.001
    rs = np.random.RandomState(1)
    eps = rs.normal(size=(B, L)).astype(np.float32)
    x = (rs.rand(B, I) < 0.15).astype(np.float32); x[3] = 0
    shapes = {"W1": (I, H), "b1": (H,), "Wm": (H, L), "bm": (L,), "Wv": (H, L), "bv": (L,), "W3": (L, H), "b3": (H,), "W4": (H, I), "b4": (I,)}
    w0 = {n: rs.normal(scale=0.2, size=s).astype(np.float32) for n, s in shapes.items()}
    res = {"I": I, "H": H, "L": L, "B": B, "lr": lr, "x": x, "eps": eps, "names": np.array(list(shapes)), "var_names": np.array(list(shapes))}
    res.update({f"{n}_0": w0[n] for n in shapes})
    c = ov.forward(w0, x, eps)
    res["logits_0"] = c["logits"]
    res["kl_0"] = np.float32(-0.5 * np.mean(c["lv"] - c["mu"] ** 2 - np.exp(c["lv"]) + 1))
    res["predict_0"] = ov.log_softmax(c["logits"])
    o = ov.MultiVAEOracle(w0, lr)
    for s, anneal in enumerate((0.0, 0.1, 0.2)):
        res[f"anneal{s}"] = np.float32(anneal)
        res[f"loss{s}"] = np.float32(o.train_step(x, eps, anneal))
        for n in shapes:
            res[f"{n}_{s + 1}"] = o.w[n].copy()
    np.savez_compressed(os.path.join(out, "tf_multivae.npz"), **res)
    # ---- NeuMF / GMF (trainable_weights order and names as Keras lists them)
    def dump(w, names_order, var_names, steps, U, I, F, lr, fname, sat):
        rs = np.random.RandomState(7)
        flat = lambda w: [w[k] if not isinstance(k, tuple) else w[k[0]][k[1]] for k in names_order]
        shape2 = lambda k, a: a.reshape(-1, 1) if k == "hw" else a
        res = {"U": U, "I": I, "F": F, "lr": lr, "var_names": np.array(var_names)}
        o = on.NeuMFOracle(w, lr)
        def snap(tag):
            for n, k in enumerate(names_order):
                a = o.w[k] if not isinstance(k, tuple) else o.w[k[0]][k[1]]
                res[f"w{n}_{tag}"] = shape2(k, a).copy()
        snap(0)
        for s in range(steps):
            u, i = rs.randint(0, U - 4, 32), rs.randint(0, I - 4, 32)
            y = rs.randint(0, 2, 32).astype(np.float32)
            u[:5] = u[5]; i[:4] = i[4]
            res[f"u{s}"], res[f"i{s}"], res[f"y{s}"] = u, i, y
            res[f"loss{s}"] = np.float32(o.train_step(u, i, y))
            snap(s + 1)
        ug, ig = np.meshgrid(np.arange(U), np.arange(I), indexing="ij")
        res["recs"] = o.predict(ug.reshape(-1), ig.reshape(-1)).reshape(U, I)
        if sat:
            ws = {k: ([a.copy() for a in v] if isinstance(v, list) else v.copy()) for k, v in o.w.items()}
            ws["hw"] = ws["hw"] * np.float32(200.0)
            u = np.arange(8)
            y = np.array([0, 1, 0, 1, 1, 0, 1, 0], np.float32)
            c = on.forward(ws, u, u)
            res.update({"sat_u": u, "sat_y": y, "sat_p": c["p"].reshape(-1, 1), "sat_loss": np.float32(on.bce(c["p"], y))})
        np.savez_compressed(os.path.join(out, fname), **res)

    U, I, F = 30, 25, 8
    w = on.init_neumf(U, I, F, 2)
    w = {k: ([a * 3 for a in v] if isinstance(v, list) else v * 3) for k, v in w.items()}
    order = ["Umf", "Imf", "Umlp", "Imlp", ("W", 0), ("b", 0), ("W", 1), ("b", 1), ("W", 2), ("b", 2), "hw", "hb"]
    names = ["U_MF/embeddings:0", "I_MF/embeddings:0", "U_MLP/embeddings:0", "I_MLP/embeddings:0", "sequential/dense/kernel:0",
             "sequential/dense/bias:0", "sequential/dense_1/kernel:0", "sequential/dense_1/bias:0", "sequential/dense_2/kernel:0",
             "sequential/dense_2/bias:0", "dense_3/kernel:0", "dense_3/bias:0"]
    dump(w, order, names, 3, U, I, F, 0.002, "tf_neumf.npz", True)
    U, I, F = 20, 18, 6
    g = on.init_gmf(U, I, F, 3)
    g = {k: v * 3 for k, v in g.items()}
    dump(g, ["hw", "Umf", "Imf"], ["h:0", "U_GMF/embeddings:0", "I_GMF/embeddings:0"], 2, U, I, F, 0.002, "tf_gmf.npz", False)
