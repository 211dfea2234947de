"""GPU end-to-end: the plugin classes (BPRMF_batch, BPRMF) driven the way ModelCoordinator.single drives them
(model_coordinator.py:100-103), checked against a full CPU replay with the oracle."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from elliot_amd.dataset.dataset import DataSet, default_config
from elliot_amd.evaluation.evaluator import Evaluator
from elliot_amd.recommender import BPRMF, BPRMF_batch
from elliot_amd.synthetic import small_dataset
from oracle import bprmf_batch as ob
from oracle import cref
from oracle import sampler as osampler
from oracle import sgd as osgd

pytestmark = pytest.mark.gpu


def make_data(tmp_path, top_k=10):
    indptr, indices, _ = small_dataset(300, 220, seed=4)
    I = int(indices.max()) + 1
    rs = np.random.RandomState(9)
    U = indptr.shape[0] - 1
    users = np.repeat(np.arange(U), np.diff(indptr))
    ratings = rs.randint(1, 6, indices.shape[0]).astype(float)
    # hold out ~20% of every user's items as test (users keep >= 1 train item)
    flag = np.zeros(indices.shape[0], bool)
    for u in range(U):
        a, b = indptr[u], indptr[u + 1]
        n_te = (b - a) // 5
        if n_te:
            flag[a + rs.choice(b - a, n_te, replace=False)] = True
    cfg = default_config(top_k=top_k, cutoffs=[top_k, 5], simple_metrics=["nDCG", "Recall"], out_dir=str(tmp_path))
    for p in (cfg.path_output_rec_result, cfg.path_output_rec_weight):
        os.makedirs(p, exist_ok=True)
    # public ids = 1000 + private user, 5000 + private item (so that the id maps are exercised)
    tr = (users[~flag] + 1000, indices[~flag] + 5000, ratings[~flag])
    te = (users[flag] + 1000, indices[flag] + 5000, ratings[flag])
    return DataSet(cfg, tr, te), cfg


def test_bprmf_batch_plugin_end_to_end_matches_cpu_replay(ctx, tmp_path):
    data, cfg = make_data(tmp_path)
    F, lr, l_w, l_b, B, epochs = 16, 0.01, 0.1, 0.001, 512, 2
    params = SimpleNamespace(meta=SimpleNamespace(save_recs=True, verbose=False), epochs=epochs, batch_size=B,
                             factors=F, lr=lr, l_w=l_w, l_b=l_b, seed=42)
    model = BPRMF_batch(data=data, config=cfg, params=params)
    st0 = model._model.state
    Gu0, Gi0, Bi0 = st0.Gu.cpu().numpy().copy(), st0.Gi.cpu().numpy().copy(), st0.Bi.cpu().numpy().copy()
    assert model.name.startswith("BPRNN_seed=42_e=2_bs=512_factors=16_lr=0$01")
    model.train()
    results = model.get_results()
    assert set(results.keys()) == {10, 5} and 0.0 <= results[10]["test_results"]["nDCG"] <= 1.0
    assert isinstance(model.get_loss(), float) and model.get_params()["name"] == model.name

    # ---- CPU replay: same Philox stream, same initial weights, TF-semantics oracle ------------------
    m = data.sp_i_train
    U, I, T = data.num_users, data.num_items, data.transactions
    orc = ob.BPRMFBatchOracle(Gu0, Gi0, Bi0, lr, l_w, l_b)
    drawn, losses = 0, []
    for it in range(epochs):
        tot = 0.0
        for start in range(0, T, B):
            n = min(start + B, T) - start
            u, i, j = osampler.philox_sample(m.indptr, m.indices, U, I, 42, drawn, n)
            drawn += n
            tot += orc.train_step((u, i, j))
        losses.append(tot / (it + 1))                       # BPRMF_batch.py:109
    for got, exp in zip(model._losses, losses):
        assert abs(got - exp) <= 1e-4 * abs(exp), (model._losses, losses)
    st = model._model.state
    assert (np.abs(st.Gu.cpu().numpy() - orc.Gu) > 2e-5).mean() < 1e-3
    assert (np.abs(st.Gi.cpu().numpy() - orc.Gi) > 2e-5).mean() < 1e-3

    # ---- recommendations: device lists == oracle lists computed from the DEVICE's final weights ------
    recs_val, recs_test = model.get_recommendations(10)
    oi, ov = cref.score_topk_f32(st.Gu.cpu().numpy(), st.Gi.cpu().numpy(), st.Bi.cpu().numpy(), 0, U, 10,
                                 excl=(m.indptr, m.indices))
    assert list(recs_test.keys()) == [data.private_users[u] for u in range(U)]
    for u in range(U):
        pub = recs_test[data.private_users[u]]
        assert [it for it, _ in pub] == [data.private_items[int(x)] for x in oi[u]]
        assert np.array_equal(np.array([s for _, s in pub], np.float32), ov[u])
    # metric = stand-alone evaluator on those lists (itself pinned to the reference Evaluator on CPU)
    again = Evaluator(data, params).eval((recs_val, recs_test))
    assert again[10]["test_results"] == model._results[-1][10]["test_results"]
    # save_recs wrote the reference's TSV format (utils/write.py:35-44)
    f = os.path.join(cfg.path_output_rec_result, f"{model.name}_it=2.tsv")
    first = open(f).readline().rstrip("\n").split("\t")
    assert len(first) == 3 and int(first[0]) == data.private_users[0]


def test_bprmf_plugin_epoch_equals_sequential_reference_semantics(ctx, tmp_path):
    data, cfg = make_data(tmp_path)
    params = SimpleNamespace(meta=SimpleNamespace(verbose=False), epochs=1, factors=12, lr=0.05, seed=42)
    model = BPRMF(data=data, config=cfg, params=params)
    U, I, T = data.num_users, data.num_items, data.transactions
    P, Q, b = osgd.initialize(U, I, 12, 42)                 # MFModel.initialize stream (BPRMF_model.py:24,40-56)
    assert np.array_equal(model._model.state.P.cpu().numpy(), P)
    model.train()
    assert model._model.levels_last > 1
    m = data.sp_i_train
    u, i, j = osampler.philox_sample(m.indptr, m.indices, U, I, 42, 0, T)
    osgd.train_sequential(P, Q, b, u, i, j, lr=0.05, reg_bias=0, reg_user=0.0025, reg_pos=0.0025, reg_neg=0.00025)
    st = model._model.state
    assert np.abs(st.P.cpu().numpy() - P).max() < 1e-12 and np.abs(st.Q.cpu().numpy() - Q).max() < 1e-12
    assert np.abs(st.b.cpu().numpy() - b).max() < 1e-12
    _, recs = model.get_recommendations(10)
    oi, ov = cref.score_topk_f64(st.P.cpu().numpy(), st.Q.cpu().numpy(), st.b.cpu().numpy(), 0, U, 10,
                                 excl=(m.indptr, m.indices))
    for uu in range(U):
        assert [it for it, _ in recs[data.private_users[uu]]] == [data.private_items[int(x)] for x in oi[uu]]
    assert model.name.startswith("BPRMF_seed=42_e=1_bs=1_f=12_lr=0$05")


def test_checkpoint_roundtrip(ctx, tmp_path):
    data, cfg = make_data(tmp_path)
    params = SimpleNamespace(meta=SimpleNamespace(save_weights=True, verbose=False), epochs=1, batch_size=256,
                             factors=8, lr=0.01, seed=1)
    model = BPRMF_batch(data=data, config=cfg, params=params)
    model.train()
    assert os.path.exists(model._saving_filepath)
    before = model.get_recommendations(10)[1]
    params2 = SimpleNamespace(meta=SimpleNamespace(restore=True, verbose=False), epochs=1, batch_size=256,
                              factors=8, lr=0.01, seed=1)
    model2 = BPRMF_batch(data=data, config=cfg, params=params2)
    model2.train()                                           # restore path: load + evaluate (BPRMF_batch.py:96-97)
    assert model2.get_recommendations(10)[1] == before


def test_mini_runner_yaml(ctx, tmp_path):
    import yaml
    from elliot_amd.run import run_experiment
    indptr, indices, _ = small_dataset(250, 200, seed=11)
    rs = np.random.RandomState(0)
    users = np.repeat(np.arange(250), np.diff(indptr))
    with open(tmp_path / "dataset.tsv", "w") as f:
        for u, i in zip(users, indices):
            f.write(f"{u + 1}\t{i + 1}\t{rs.randint(1, 6)}\t{rs.randint(0, 10 ** 6)}\n")
    cfg = {"experiment": {
        "dataset": "toy", "data_config": {"strategy": "dataset", "dataset_path": "dataset.tsv"},
        "splitting": {"test_splitting": {"strategy": "random_subsampling", "test_ratio": 0.2}},
        "top_k": 10, "evaluation": {"simple_metrics": ["nDCG"]},
        "path_output_rec_result": "out/recs/", "path_output_rec_weight": "out/weights/",
        "path_output_rec_performance": "out/perf/",
        "models": {"external.BPRMF_batch": {"meta": {"save_recs": True}, "epochs": 2, "batch_size": 512, "factors": 64,
                                            "lr": 0.001, "l_w": 0.1, "l_b": 0.001},
                   "BPRMF": {"meta": {}, "epochs": 1, "factors": 64}}}}
    with open(tmp_path / "exp.yml", "w") as f:
        yaml.safe_dump(cfg, f)
    res = run_experiment(str(tmp_path / "exp.yml"))
    assert len(res) == 2
    for name, r in res.items():
        assert 0.0 <= r[10]["test_results"]["nDCG"] <= 1.0
    assert os.listdir(tmp_path / "out" / "perf")


def test_mini_runner_folds_prefilter_and_validation_split(ctx, tmp_path):
    """prefiltering + 2-fold cross validation + a temporal leave-one-out validation split: one run per data object
    (elliot/run.py:59-75), the validation metric comes from the validation split (val_results != test_results)."""
    import yaml
    from elliot_amd.run import run_experiment
    indptr, indices, _ = small_dataset(150, 120, seed=3)
    rs = np.random.RandomState(2)
    users = np.repeat(np.arange(150), np.diff(indptr))
    with open(tmp_path / "dataset.tsv", "w") as f:
        for u, i in zip(users, indices):
            f.write(f"{u + 1}\t{i + 1}\t{rs.randint(1, 6)}\t{rs.randint(0, 10 ** 6)}\n")
    cfg = {"experiment": {
        "dataset": "toy", "data_config": {"strategy": "dataset", "dataset_path": "dataset.tsv"},
        "prefiltering": {"strategy": "user_k_core", "core": 6}, "binarize": True,
        "splitting": {"test_splitting": {"strategy": "random_cross_validation", "folds": 2},
                      "validation_splitting": {"strategy": "temporal_hold_out", "leave_n_out": 1}},
        "top_k": 10, "evaluation": {"simple_metrics": ["nDCG", "Recall"]},
        "path_output_rec_result": "out/recs/", "path_output_rec_weight": "out/weights/", "path_output_rec_performance": "out/perf/",
        "models": {"BPRMF_batch": {"meta": {}, "epochs": 2, "batch_size": 256, "factors": 16, "lr": 0.01}}}}
    with open(tmp_path / "exp.yml", "w") as f:
        yaml.safe_dump(cfg, f)
    res = run_experiment(str(tmp_path / "exp.yml"))
    assert len(res) == 2 and sum("#test1val0" in k for k in res) == 1
    for r in res.values():
        assert 0.0 <= r[10]["test_results"]["Recall"] <= 1.0 and r[10]["val_results"] != r[10]["test_results"]
    assert len(os.listdir(tmp_path / "out" / "perf")) == 2


def test_mini_runner_with_sampled_negatives(ctx, tmp_path):
    """`negative_sampling: {strategy: random, num_items: 40}` (dataset.py:221-243, recommender_utils_mixin.py:102-109): every
    recommended item is one of the user's candidates (its negatives of ../data/<dataset>/negative.tsv + its own test items);
    the dict route and the device-metric route give the same nDCG."""
    import yaml
    from elliot_amd import run as runner
    indptr, indices, _ = small_dataset(180, 260, seed=5)
    rs = np.random.RandomState(1)
    users = np.repeat(np.arange(180), np.diff(indptr))
    os.makedirs(tmp_path / "cfg")
    with open(tmp_path / "cfg" / "dataset.tsv", "w") as f:
        for u, i in zip(users, indices):
            f.write(f"{u + 1}\t{i + 1}\t{rs.randint(1, 6)}\t{rs.randint(0, 10 ** 6)}\n")
    exp = {"dataset": "toy", "data_config": {"strategy": "dataset", "dataset_path": "dataset.tsv"},
           "splitting": {"test_splitting": {"strategy": "random_subsampling", "test_ratio": 0.2}},
           "negative_sampling": {"strategy": "random", "num_items": 40},
           "top_k": 10, "evaluation": {"simple_metrics": ["nDCG", "HR"]},
           "path_output_rec_result": "out/recs/", "path_output_rec_weight": "out/weights/", "path_output_rec_performance": "out/perf/",
           "models": {"BPRMF_batch": {"meta": {"save_recs": False}, "epochs": 2, "batch_size": 512, "factors": 16, "lr": 0.01}}}
    with open(tmp_path / "cfg" / "exp.yml", "w") as f:
        yaml.safe_dump({"experiment": exp}, f)
    res = runner.run_experiment(str(tmp_path / "cfg" / "exp.yml"))
    (name, r), = res.items()
    neg_file = tmp_path / "data" / "toy" / "negative.tsv"
    assert neg_file.exists()
    # rebuild the model's data to look at the lists
    cfg = runner.build_config(exp, str(tmp_path / "cfg"))
    data = runner.load_data(exp, cfg, str(tmp_path / "cfg"))
    negs = {}
    for line in neg_file.read_text().splitlines():
        parts = line.split("\t")
        negs[int(parts[0].strip("(),"))] = {int(x) for x in parts[1:]}
    assert len(negs) == data.num_users and all(len(v) == 40 for v in negs.values())
    from elliot_amd.recommender import BPRMF_batch
    params = SimpleNamespace(meta=SimpleNamespace(verbose=False, save_recs=False), epochs=1, batch_size=512, factors=16, lr=0.01, seed=42)
    model = BPRMF_batch(data=data, config=cfg, params=params)
    model.train()
    _, recs = model.get_recommendations(10)
    test = data.test_dict
    train = data.train_dict
    for u, lst in recs.items():
        allowed = negs[u] | {i for i in test.get(u, {}) if i in data.public_items}
        got = {i for i, _ in lst}
        assert got <= allowed and not (got & set(train[u])), u
        assert len(lst) == min(10, len(allowed))
    dict_route = model.evaluator.eval(model.get_recommendations(10))
    assert abs(dict_route[10]["test_results"]["nDCG"] - model.get_results()[10]["test_results"]["nDCG"]) < 1e-9
    assert 0.0 < r[10]["test_results"]["HR"] <= 1.0


def test_multivae_plugin_end_to_end_matches_cpu_replay(ctx, tmp_path):
    import random
    from elliot_amd.recommender import MultiVAE
    from oracle import multi_vae as ov
    data, cfg = make_data(tmp_path)
    U, I = data.num_users, data.num_items
    H, L, B, epochs, lr = 32, 8, 64, 2, 0.001
    w0 = ov.init_weights(I, H, L, 7)
    params = SimpleNamespace(meta=SimpleNamespace(verbose=False), epochs=epochs, batch_size=B, intermediate_dim=H,
                             latent_dim=L, lr=lr, dropout_pkeep=1, seed=42, eps_mode="zero")
    model = MultiVAE(data=data, config=cfg, params=params, init_weights=w0)
    assert model.name.startswith("MultiVAE_seed=42_e=2_bs=64_intermediate_dim=32_latent_dim=8_reg_lambda=0$01")
    model.train()
    # CPU replay: same user shuffle (random.seed(42); random.sample, sparse_sampler.py:16,21), eps = 0
    X = data.sp_i_train.toarray().astype(np.float32)
    orc = ov.MultiVAEOracle(w0, lr)
    random.seed(42)
    losses, count = [], 0
    for it in range(epochs):
        order = random.sample(range(U), U)
        tot = 0.0
        for s in range(0, U, B):
            rows = order[s:s + B]
            anneal = min(0.2, count / 200000)
            tot += orc.train_step(X[rows], np.zeros((len(rows), L), np.float32), anneal)
            count += 1
        losses.append(tot / (it + 1))
    for got, exp in zip(model._losses, losses):
        assert abs(got - exp) <= 1e-4 * abs(exp), (model._losses, losses)
    gw = model._model.state.weights()
    for k in ov.NAMES:
        assert (np.abs(gw[k] - orc.w[k]) > 5e-5).mean() < 5e-3, k
    # recommendations = masked top-k of log_softmax from the DEVICE's weights
    _, recs = model.get_recommendations(10)
    ref = ov.log_softmax(ov.forward(gw, X, np.zeros((U, L), np.float32), dtype=np.float64)["logits"])
    m = data.sp_i_train
    hits = 0
    for u in range(U):
        got_items = [it for it, _ in recs[data.private_users[u]]]
        masked = np.where(X[u] > 0, -np.inf, ref[u])
        exp_items = [data.private_items[int(i)] for i in np.lexsort((np.arange(I), -masked))[:10]]
        hits += got_items == exp_items
        assert set(got_items).isdisjoint({data.private_items[int(i)] for i in m.indices[m.indptr[u]:m.indptr[u + 1]]})
    assert hits >= 0.97 * U          # fp32 vs fp64 scores: only near-ties at the k-th place may differ


def test_multidae_plugin_end_to_end_matches_cpu_replay(ctx, tmp_path):
    """MultiDAE as a sibling of MultiVAE on the same kernels (dae mode)."""
    import random
    from elliot_amd.recommender import MultiDAE
    from oracle import multi_dae as od
    data, cfg = make_data(tmp_path)
    U, I = data.num_users, data.num_items
    H, L, B, epochs, lr = 32, 8, 64, 2, 0.001
    w0 = od.init_weights(I, H, L, 7)
    params = SimpleNamespace(meta=SimpleNamespace(verbose=False), epochs=epochs, batch_size=B, intermediate_dim=H,
                             latent_dim=L, lr=lr, dropout_pkeep=1, seed=42)
    model = MultiDAE(data=data, config=cfg, params=params, init_weights=w0)
    assert model.name.startswith("MultiDAE_seed=42_e=2_bs=64_intermediate_dim=32_latent_dim=8_reg_lambda=0$01")
    model.train()
    X = data.sp_i_train.toarray().astype(np.float32)
    orc = od.MultiDAEOracle(w0, lr)
    random.seed(42)
    losses = []
    for it in range(epochs):
        order = random.sample(range(U), U)
        tot = 0.0
        for s in range(0, U, B):
            tot += orc.train_step(X[order[s:s + B]])
        losses.append(tot / (it + 1))
    for got, exp in zip(model._losses, losses):
        assert abs(got - exp) <= 1e-4 * abs(exp), (model._losses, losses)
    gw = model._model.state.weights()
    assert set(gw.keys()) == set(od.NAMES)
    for k in od.NAMES:
        assert (np.abs(gw[k] - orc.w[k]) > 5e-5).mean() < 5e-3, k
    res = model.get_results()
    assert set(res.keys()) == {10, 5} and 0.0 <= res[10]["test_results"]["nDCG"] <= 1.0


def test_neumf_and_gmf_plugins_end_to_end(ctx, tmp_path):
    import random
    from elliot_amd.recommender import GMF, NeuMF
    from oracle import neumf as on
    data, cfg = make_data(tmp_path)
    U, I, F, lr = data.num_users, data.num_items, 8, 0.002
    # ---- NeuMF: m = 1 negatives, CPU replay of the reference's epoch construction (custom_sampler.py:27-48)
    w0 = on.init_neumf(U, I, F, 11)
    params = SimpleNamespace(meta=SimpleNamespace(verbose=False), epochs=1, batch_size=256, mf_factors=F, lr=lr, m=1, seed=42)
    model = NeuMF(data=data, config=cfg, params=params, init_weights=w0)
    assert model.name.startswith("NeuMF_seed=42_e=1_bs=256_lr=0$002_mffactors=8_drop=0_mftrain=True_mlptrain=True_m=1")
    model.train()
    np.random.seed(42)
    random.seed(42)
    itd = data.i_train_dict
    ui = {u: list(set(itd[u])) for u in itd}
    pos = {(u, i, 1) for u, items in ui.items() for i in items}
    neg = set()
    for u, i, _ in pos:
        j = np.random.randint(I)
        while j in ui[u]:
            j = np.random.randint(I)
        neg.add((u, j, 0))
    samples = list(pos)
    samples.extend(list(neg))
    samples = np.asarray(random.sample(samples, len(samples)))
    orc = on.NeuMFOracle(w0, lr)
    tot = 0.0
    for s in range(0, len(samples), 256):
        b = samples[s:s + 256]
        tot += orc.train_step(b[:, 0], b[:, 1], b[:, 2].astype(np.float32))
    assert abs(model._losses[0] - tot) <= 1e-4 * abs(tot), (model._losses, tot)
    gw = model._model.state.weights()
    assert (np.abs(gw["Umf"] - orc.w["Umf"]) > 5e-5).mean() < 5e-3
    assert (np.abs(gw["W"][0] - orc.w["W"][0]) > 5e-5).mean() < 5e-3
    # recommendations: probabilities of every (user, item) pair from the device weights, masked top-k
    _, recs = model.get_recommendations(10)
    ug, ig = np.repeat(np.arange(U), I), np.tile(np.arange(I), U)
    ref = on.forward(gw, ug, ig, dtype=np.float64)["p"].reshape(U, I)
    m = data.sp_i_train
    agree = 0
    for u in range(U):
        masked = ref[u].copy()
        masked[m.indices[m.indptr[u]:m.indptr[u + 1]]] = -np.inf
        exp_items = [data.private_items[int(i)] for i in np.lexsort((np.arange(I), -masked))[:10]]
        agree += [it for it, _ in recs[data.private_users[u]]] == exp_items
    assert agree >= 0.95 * U
    # ---- GMF: device Philox point-wise sampler; loss decreases and results are well-formed
    params = SimpleNamespace(meta=SimpleNamespace(verbose=False), epochs=3, batch_size=512, mf_factors=16, lr=0.01, seed=42)
    g = GMF(data=data, config=cfg, params=params)
    g.train()
    assert g.name.startswith("GeneralizedMF_seed=42_e=3_bs=512_lr=0$01_mffactors=16_isedgeweighttrain=True")
    per_epoch = [l * (n + 1) for n, l in enumerate(g._losses)]
    assert per_epoch[-1] < per_epoch[0]
    assert 0.0 <= g.get_results()[10]["test_results"]["nDCG"] <= 1.0


def test_bprmf_plugin_with_replay_sampler_reproduces_the_reference_run(ctx, tmp_path, golden):
    """Seed-exact end-to-end: the reference's OWN BPRMF inner loop (MFModel + Sampler, one epoch, fixture generated by
    oracle/gen_golden.py from the reference classes) vs our plugin with `sampler: replay` on the MI355X:
    same triplet stream -> same parameters (fp64 round-off) -> the same top-10 lists."""
    g = golden("bprmf_e2e_ref.npz")
    U, I = g["P"].shape[0], g["Q"].shape[0]
    cfg = default_config(top_k=10, cutoffs=[10], simple_metrics=["nDCG"], out_dir=str(tmp_path))
    os.makedirs(cfg.path_output_rec_weight, exist_ok=True)
    test = (np.array([0]), np.array([0]), np.array([1.0]))
    data = DataSet(cfg, (g["train_u"], g["train_i"], g["train_r"]), test, public_users=np.arange(U), public_items=np.arange(I))
    params = SimpleNamespace(meta=SimpleNamespace(verbose=False), epochs=1, factors=int(g["factors"]), seed=42,
                             sampler="replay")
    model = BPRMF(data=data, config=cfg, params=params)
    model.train()
    st = model._model.state
    for name, ref in (("P", g["P"]), ("Q", g["Q"]), ("b", g["b"])):
        err = np.abs(getattr(st, name).cpu().numpy() - ref).max()
        assert err < 1e-12, (name, err)
    _, recs = model.get_recommendations(10)
    for u in range(U):
        assert [it for it, _ in recs[u]] == g["rec_idx"][u].tolist(), u
        assert np.allclose([s for _, s in recs[u]], g["rec_val"][u], rtol=0, atol=1e-12)


def test_reference_checkpoint_loads_and_recommends_identically(ctx, tmp_path, golden):
    """On-disk format: a checkpoint written by the REFERENCE's MFModel.save_weights (BPRMF_model.py:119-139,
    fixture tests/golden/bprmf_ref_weights.pkl from oracle/gen_golden.py) restores into our model and yields the
    reference's own recommendation lists; our save_weights writes the same dict layout."""
    import pickle
    g = golden("bprmf_e2e_ref.npz")
    ref_ckpt = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bprmf_ref_weights.pkl")
    U, I = g["P"].shape[0], g["Q"].shape[0]
    cfg = default_config(top_k=10, cutoffs=[10], simple_metrics=["nDCG"], out_dir=str(tmp_path))
    os.makedirs(cfg.path_output_rec_weight, exist_ok=True)
    test = (np.array([0]), np.array([0]), np.array([1.0]))
    data = DataSet(cfg, (g["train_u"], g["train_i"], g["train_r"]), test, public_users=np.arange(U), public_items=np.arange(I))
    params = SimpleNamespace(meta=SimpleNamespace(verbose=False), epochs=1, factors=int(g["factors"]), seed=7)
    model = BPRMF(data=data, config=cfg, params=params)                    # different seed: untrained, unrelated weights
    model._model.load_weights(ref_ckpt)
    _, recs = model.get_recommendations(10)
    for u in range(U):
        assert [it for it, _ in recs[u]] == g["rec_idx"][u].tolist(), u
        assert np.allclose([s for _, s in recs[u]], g["rec_val"][u], rtol=0, atol=1e-12)
    out = os.path.join(str(tmp_path), "ours.pkl")
    model._model.save_weights(out)
    ours, theirs = pickle.load(open(out, "rb")), pickle.load(open(ref_ckpt, "rb"))
    assert set(ours.keys()) == set(theirs.keys())
    for k in theirs:
        assert np.asarray(ours[k]).shape == np.asarray(theirs[k]).shape and np.asarray(ours[k]).dtype == np.asarray(theirs[k]).dtype, k
        if k != "_user_bias":
            assert np.array_equal(np.asarray(ours[k]), np.asarray(theirs[k])), k


def test_device_metrics_path_equals_host_evaluator(ctx, tmp_path):
    """Without save_recs, evaluate() computes the metrics on the device from the [users, k] index tensors
    (el_rec_metrics); the numbers must be those of the host evaluator on the same lists."""
    data, cfg = make_data(tmp_path)
    cfg.evaluation.simple_metrics = ["nDCG", "Precision", "Recall", "HR", "MAP", "MRR", "F1"]
    cfg.evaluation.relevance_threshold = 2
    params = SimpleNamespace(meta=SimpleNamespace(save_recs=False, verbose=False), epochs=1, batch_size=512,
                             factors=16, lr=0.01, l_w=0.1, l_b=0.001, seed=42)
    model = BPRMF_batch(data=data, config=cfg, params=params)
    assert model._device_metrics()
    model.train()
    dev = model._results[-1]
    host = Evaluator(data, params).eval(model.get_recommendations(10))
    assert set(dev.keys()) == set(host.keys()) == {10, 5}
    for c in dev:
        for split in ("val_results", "test_results"):
            assert set(dev[c][split].keys()) == set(host[c][split].keys())
            for m, v in host[c][split].items():
                assert abs(dev[c][split][m] - v) < 1e-12, (c, split, m, dev[c][split][m], v)
    cfg.device_metrics = False                       # the switch falls back to the dict path
    assert not model._device_metrics()


POINTWISE = {  # plugin -> (extra YAML params, file-name prefix, oracle kind / biases / optimiser / oracle kwargs)
    "MF": (dict(factors=8, lr=0.01, reg=0.1), "MF_seed=42_e=2_bs=256_factors=8_lr=0$01_reg=0$1", ("mse", False, "adam", {})),
    "PMF": (dict(factors=8, lr=0.01, reg=0.0025, gaussian_variance=0.1),
            "PMF_seed=42_e=2_bs=256_lr=0$01_factors=8_reg=0$0025_gvar=0$1", ("mse_sigmoid", False, "adam", {})),
    "FunkSVD": (dict(factors=8, lr=0.01, reg_w=0.1, reg_b=0.001),
                "FunkSVD_seed=42_e=2_bs=256_factors=8_lr=0$01_reg_w=0$1_reg_b=0$001", ("mse", True, "adam", {})),
    "LogisticMatrixFactorization": (dict(factors=8, lr=0.05, reg=0.1, alpha=0.5),
                                    "LMF_seed=42_e=2_bs=256_lr=0$05_factors=8_reg=0$1_alpha=0$5",
                                    ("logistic", True, "adagrad", dict(alpha=0.5, l_w=0.1))),
}


@pytest.mark.parametrize("plugin", list(POINTWISE))
def test_pointwise_plugins_end_to_end_match_cpu_replay(ctx, tmp_path, plugin):
    """MF / PMF / FunkSVD / LogisticMF driven as ModelCoordinator drives them; the device sampler's batches
    are replayed through oracle/pointwise_mf.py."""
    import elliot_amd.recommender as rec
    from elliot_amd.dataset.samplers import pointwise_pos_neg_sampler
    from oracle import pointwise_mf as pw
    data, cfg = make_data(tmp_path)
    U, I, F, B, epochs = data.num_users, data.num_items, 8, 256, 2
    extra, prefix, (kind, biases, opt, okw) = POINTWISE[plugin]
    rs = np.random.RandomState(3)
    w0 = {"Gu": rs.normal(scale=0.2, size=(U, F)).astype(np.float32), "Gi": rs.normal(scale=0.2, size=(I, F)).astype(np.float32)}
    if biases:
        w0["Bu"], w0["Bi"] = rs.normal(scale=0.05, size=U).astype(np.float32), rs.normal(scale=0.05, size=I).astype(np.float32)
    params = SimpleNamespace(meta=SimpleNamespace(verbose=False, save_weights=True), epochs=epochs, batch_size=B, seed=42, **extra)
    model = getattr(rec, plugin)(data=data, config=cfg, params=params, init_weights=w0)
    assert model.name == prefix, model.name
    model.train()
    # replay
    orc = pw.PointwiseOracle(w0, kind, extra["lr"], optimizer=opt, **okw)
    sampler = pointwise_pos_neg_sampler.Sampler(data.sp_i_train, ctx=ctx)
    losses = []
    for it in range(epochs):
        tot = 0.0
        for side in (("items", "users") if kind == "logistic" else ("both",)):
            for u, i, y in sampler.step(data.transactions, B):
                tot += orc.train_step((u.cpu().numpy(), i.cpu().numpy(), y.cpu().numpy()), side=side)
        losses.append(tot / (it + 1))
    for got, exp in zip(model._losses, losses):
        assert abs(got - exp) <= 2e-4 * abs(exp), (model._losses, losses)
    gw = model._model.get_model_state()
    for k, v in orc.w.items():
        assert (np.abs(gw[k].reshape(v.shape) - v) > 5e-5).mean() < 5e-3, k
    # recommendation lists = masked top-k of the oracle's score table on the device weights
    _, recs = model.get_recommendations(10)
    ref = pw.PointwiseOracle({k: gw[k] for k in orc.w}, kind, 0.0, optimizer=opt).predict_all(0, U).astype(np.float64)
    m = data.sp_i_train
    agree = 0
    for u in range(U):
        masked = ref[u].copy()
        masked[m.indices[m.indptr[u]:m.indptr[u + 1]]] = -np.inf
        top = np.lexsort((np.arange(I), -masked))[:10]
        got = recs[data.private_users[u]]
        agree += [it for it, _ in got] == [data.private_items[int(i)] for i in top]
        assert np.abs(np.array([s for _, s in got]) - masked[top]).max() < 2e-6
    assert agree >= 0.97 * U
    res = model.get_results()
    assert set(res.keys()) == {10, 5} and 0.0 <= res[10]["test_results"]["nDCG"] <= 1.0
    # checkpoint written at the best epoch restores into a fresh model
    fresh = getattr(rec, plugin)(data=data, config=cfg, params=params)
    fresh._model.load_weights(model._saving_filepath)
    a, b = fresh._model.recommend(None, 5, 0, U), model._model.recommend(None, 5, 0, U)
    if model.get_best_arg() == epochs - 1:
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_pointwise_plugins_learn(ctx, tmp_path):
    """Default initialisers + the device sampler: the epoch loss falls and ranking quality beats a random ranking."""
    from elliot_amd.recommender import MF, PMF, FunkSVD
    data, cfg = make_data(tmp_path)
    for cls, lr in ((MF, 0.02), (FunkSVD, 0.02), (PMF, 0.05)):
        params = SimpleNamespace(meta=SimpleNamespace(verbose=False), epochs=12, batch_size=512, factors=16, lr=lr, seed=42)
        m = cls(data=data, config=cfg, params=params)
        m.train()
        per_epoch = [l * (n + 1) for n, l in enumerate(m._losses)]
        assert per_epoch[-1] < per_epoch[0], (cls.__name__, per_epoch)
        assert m.get_results()[10]["test_results"]["Recall"] > 2 * 10 / data.num_items, cls.__name__


def test_mini_runner_runs_every_model_and_the_proxy_reads_its_recs(ctx, tmp_path):
    """One YAML with every plugin of the package; then ProxyRecommender re-evaluates the recs file BPRMF_batch wrote and gets
    BPRMF_batch's own metrics (write -> read -> evaluate round trip through the real models)."""
    import glob
    import yaml
    from elliot_amd.run import run_experiment
    indptr, indices, _ = small_dataset(250, 200, seed=11)
    rs = np.random.RandomState(0)
    users = np.repeat(np.arange(250), np.diff(indptr))
    with open(tmp_path / "dataset.tsv", "w") as f:
        for u, i in zip(users, indices):
            f.write(f"{u + 1}\t{i + 1}\t{rs.randint(1, 6)}\t{rs.randint(0, 10 ** 6)}\n")
    base = {"dataset": "toy", "data_config": {"strategy": "dataset", "dataset_path": "dataset.tsv"},
            "splitting": {"test_splitting": {"strategy": "random_subsampling", "test_ratio": 0.2}},
            "top_k": 10, "evaluation": {"simple_metrics": ["nDCG", "Recall"]},
            "path_output_rec_result": "out/recs/", "path_output_rec_weight": "out/weights/", "path_output_rec_performance": "out/perf/"}
    common = {"epochs": 2, "batch_size": 256}
    models = {
        "BPRMF_batch": {"meta": {"save_recs": True}, "epochs": 1, "batch_size": 256, "factors": 16, "lr": 0.01, "l_w": 0.01, "l_b": 0.001},
        "BPRMF": {"epochs": 1, "factors": 16},
        "MF": {**common, "factors": 16, "lr": 0.01}, "PMF": {**common, "factors": 16, "lr": 0.01},
        "FunkSVD": {**common, "factors": 16, "lr": 0.01},
        "LogisticMatrixFactorization": {**common, "factors": 16, "lr": 0.05, "reg": 0.01, "alpha": 0.5},
        "CML": {**common, "factors": 16, "lr": 0.01}, "GMF": {**common, "mf_factors": 8, "lr": 0.01},
        "NeuMF": {**common, "mf_factors": 8, "lr": 0.01, "m": 1, "dropout": 0.1},
        "MultiVAE": {**common, "intermediate_dim": 32, "latent_dim": 8}, "MultiDAE": {**common, "intermediate_dim": 32, "latent_dim": 8},
    }
    with open(tmp_path / "exp.yml", "w") as f:
        yaml.safe_dump({"experiment": {**base, "models": models}}, f)
    res = run_experiment(str(tmp_path / "exp.yml"))
    assert len(res) == len(models)
    for name, r in res.items():
        assert 0.0 <= r[10]["test_results"]["nDCG"] <= 1.0, name
    recs = sorted(glob.glob(str(tmp_path / "out" / "recs" / "BPRNN_*.tsv")))
    assert recs, os.listdir(tmp_path / "out" / "recs")
    with open(tmp_path / "proxy.yml", "w") as f:
        yaml.safe_dump({"experiment": {**base, "models": {"ProxyRecommender": {"path": recs[-1]}}}}, f)
    again = run_experiment(str(tmp_path / "proxy.yml"))
    (proxy_res,) = again.values()
    trained = next(r for n, r in res.items() if n.startswith("BPRNN_"))          # one epoch -> one file = the reported result
    for metric, value in trained[10]["test_results"].items():
        assert abs(proxy_res[10]["test_results"][metric] - value) < 1e-12, metric


def test_lightgcn_plugin_end_to_end_matches_cpu_replay(ctx, tmp_path):
    """external.LightGCN driven like ModelCoordinator.single: injected (non-zero) tables, two epochs of Philox triplets, against the
    oracle's replay of LightGCN_model.train_step on the same triplets with the reference's Laplacian (oracle/lightgcn.create_adj_mat);
    and the reference's own initialisation -- all-zero tables (LightGCN_model.py:63-65) -- which never learns: constant loss, lists by item id."""
    from elliot_amd.recommender import LightGCN
    from oracle import lightgcn as ol
    data, cfg = make_data(tmp_path)
    U, I, T = data.num_users, data.num_items, data.transactions
    F, lr, l_w, B, epochs, L = 16, 0.005, 0.05, 512, 2, 2
    rs = np.random.RandomState(3)
    Gu0 = rs.normal(scale=0.2, size=(U, F)).astype(np.float32)
    Gi0 = rs.normal(scale=0.2, size=(I, F)).astype(np.float32)
    params = SimpleNamespace(meta=SimpleNamespace(save_recs=False, verbose=False), epochs=epochs, batch_size=B, latent_dim=F, lr=lr, l_w=l_w,
                             n_layers=L, n_fold=3, seed=42)      # (`latent_dim` is the key LightGCN.py:72 reads; `factors` is its file-name shortcut)
    model = LightGCN(data=data, config=cfg, params=params, init_weights=(Gu0, Gi0))
    assert model.name.startswith("LightGCN_seed=42_e=2_bs=512")
    _, lap = ol.create_adj_mat(data.sp_i_train, U, I)
    mine = model._laplacian.tocsr()
    mine.sort_indices()
    lap.sort_indices()
    assert np.array_equal(mine.indices, lap.indices) and np.array_equal(mine.data.view(np.uint32), lap.data.astype(np.float32).view(np.uint32))
    model.train()
    orc = ol.LightGCNOracle(Gu0, Gi0, lap, lr, l_w, L)
    m = data.sp_i_train
    drawn = 0
    for it in range(epochs):
        tot = 0.0
        for start in range(0, T, B):
            n = min(start + B, T) - start
            u, i, j = osampler.philox_sample(m.indptr, m.indices, U, I, 42, drawn, n)
            drawn += n
            tot += orc.train_step((u, i, j))
        assert abs(model._losses[it] - tot / (it + 1)) <= 1e-4 * abs(tot / (it + 1)), (it, model._losses[it], tot / (it + 1))
    st = model._model.state
    assert np.abs(st.Gu.cpu().numpy() - orc.Gu).max() < 2e-5 and np.abs(st.Gi.cpu().numpy() - orc.Gi).max() < 2e-5
    res = model.get_results()
    assert 0.0 <= res[10]["test_results"]["nDCG"] <= 1.0
    # the reference's own initialisation
    zero = LightGCN(data=data, config=cfg, params=SimpleNamespace(meta=SimpleNamespace(save_recs=False, verbose=False), epochs=1, batch_size=B,
                                                                 latent_dim=F, lr=lr, l_w=l_w, n_layers=1, n_fold=1, seed=42))
    zero.train()
    zs = zero._model.state
    assert not bool(zs.Gu.any()) and not bool(zs.Gi.any())
    steps = -(-T // B)
    assert abs(zero._losses[0] - T * np.log(2.0)) < 1e-3 * T, (zero._losses[0], T * np.log(2.0), steps)


def test_mf2020_plugin_epoch_equals_the_oracle_on_the_reference_samplers_stream(ctx, tmp_path):
    """external.MF2020: one epoch (every positive + m = 2 uniform negatives each, the reference's shuffle) through the plugin == the
    oracle's MFModel.train_step on the oracle's restatement of the Rendle sampler (both pinned to the reference in
    tests/test_oracle_graph.py): parameters to 1e-11, the epoch loss, fp64 scores of the recommendation lists."""
    from elliot_amd.recommender import MF2020
    from oracle import mf2020 as om
    data, cfg = make_data(tmp_path)
    U, I = data.num_users, data.num_items
    F, lr, reg, m_neg = 12, 0.05, 0.003, 2
    params = SimpleNamespace(meta=SimpleNamespace(save_recs=False, verbose=False), epochs=1, factors=F, lr=lr, reg=reg, m=m_neg, seed=42)
    model = MF2020(data=data, config=cfg, params=params)
    assert model.name.startswith("MF2020_seed=42_e=1")
    model.train()
    P, Q, bu, bi, gb = om.initialize(U, I, F, 42)
    ep = om.rendle_epoch(data.sp_i_train, m_neg, 42)
    tot, nb = 0.0, 0
    for s in range(0, ep.shape[0], 100000):
        b = ep[s:s + 100000]
        l, gb = om.train_step(P, Q, bu, bi, gb, b, lr, reg)
        tot += l / len(b)
        nb += 1
    st = model._model.state
    assert np.abs(st.P.cpu().numpy() - P).max() < 1e-11 and np.abs(st.Q.cpu().numpy() - Q).max() < 1e-11
    assert np.abs(st.bu.cpu().numpy() - bu).max() < 1e-11 and abs(float(st.gb.item()) - gb) < 1e-11
    assert abs(model._losses[0] - tot) <= 1e-10 * abs(tot)
    recs_val, recs_test = model.get_recommendations(10)
    scores = om.prepare_predictions(P, Q, bu, bi, gb)
    train = data.sp_i_train.toarray() > 0
    for pub_u, lst in list(recs_test.items())[:40]:
        u = data.public_users[pub_u]
        s = np.where(train[u], -np.inf, scores[u])
        best = np.sort(s)[::-1][:10]
        assert np.allclose([v for _, v in lst], best, rtol=1e-10, atol=1e-12)


def test_ngcf_plugin_end_to_end_matches_cpu_replay(ctx, tmp_path):
    """external.NGCF: injected layer-0 embeddings and GraphLayers, two propagation layers, message dropout 0, one epoch of Philox triplets
    against the oracle's replay of NGCF_model.train_step; then the defaults (zero tables, message dropout 0.1, a node dropout) run."""
    from elliot_amd.recommender import NGCF
    from oracle import lightgcn as ol
    from oracle import ngcf as on
    data, cfg = make_data(tmp_path)
    U, I, T = data.num_users, data.num_items, data.transactions
    F, ws, lr, l_w, B = 16, [12, 8], 0.005, 0.02, 512
    rs = np.random.RandomState(8)
    W = F + sum(ws)
    Gu0, Gi0 = np.zeros((U, W), np.float32), np.zeros((I, W), np.float32)
    Gu0[:, :F] = rs.normal(scale=0.2, size=(U, F))
    Gi0[:, :F] = rs.normal(scale=0.2, size=(I, F))
    sizes = [F] + ws
    layers = [{"W1": rs.normal(scale=0.3, size=(sizes[k], sizes[k + 1])).astype(np.float32), "b1": rs.normal(scale=0.1, size=(1, sizes[k + 1])).astype(np.float32),
               "W2": rs.normal(scale=0.3, size=(sizes[k], sizes[k + 1])).astype(np.float32), "b2": rs.normal(scale=0.1, size=(1, sizes[k + 1])).astype(np.float32)}
              for k in range(len(ws))]
    params = SimpleNamespace(meta=SimpleNamespace(save_recs=False, verbose=False), epochs=1, batch_size=B, latent_dim=F, lr=lr, l_w=l_w,
                             weight_size=str(tuple(ws)), node_dropout="()", message_dropout="(0.0, 0.0)", n_fold=2, seed=42)
    model = NGCF(data=data, config=cfg, params=params, init_weights=(Gu0, Gi0, layers))
    assert model.name.startswith("NGCF_seed=42_e=1_bs=512") and "weight_size=12-8_node_dropout=_message_dropout=0$0-0$0" in model.name
    model.train()
    _, lap = ol.create_adj_mat(data.sp_i_train, U, I)
    orc = on.NGCFOracle(Gu0, Gi0, lap, layers, F, lr, l_w)
    m = data.sp_i_train
    drawn, tot = 0, 0.0
    for start in range(0, T, B):
        n = min(start + B, T) - start
        u, i, j = osampler.philox_sample(m.indptr, m.indices, U, I, 42, drawn, n)
        drawn += n
        tot += orc.train_step((u, i, j))
    assert abs(model._losses[0] - tot) <= 1e-4 * abs(tot), (model._losses[0], tot)
    st = model._model.state
    assert np.abs(st.Gu.cpu().numpy() - orc.Gu).max() < 5e-5 and np.abs(st.Gi.cpu().numpy() - orc.Gi).max() < 5e-5
    for l, ref in zip(st.layers, orc.layers):
        for k in ref:
            assert np.abs(l[k].cpu().numpy() - ref[k]).max() < 1e-5, k
    assert 0.0 <= model.get_results()[10]["test_results"]["nDCG"] <= 1.0
    dflt = NGCF(data=data, config=cfg, params=SimpleNamespace(meta=SimpleNamespace(save_recs=False, verbose=False), epochs=1, batch_size=B,
                                                              latent_dim=F, node_dropout="(0.9,)", seed=42))
    dflt.train()
    assert np.isfinite(dflt._losses[0]) and not bool(dflt._model.state.Gu[:, :F].any())           # layer-0 columns: zero, forever
