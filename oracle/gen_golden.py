#!/usr/bin/env python3
"""Mint the golden fixtures under tests/golden/ from the *reference itself*.

TEST INFRASTRUCTURE ONLY.  Runs in the build container, where /root/reference is
mounted read-only; it never runs on the GPU box and nothing in the product path
imports it.  It executes the reference's own Python (imported from where it lies,
nothing is copied) on CPU and stores inputs + expected outputs as small .npz
files.  The reference source never enters the repo: fixtures are data only.

Stages (each runs in its own interpreter so the reference's import-time argparse
singletons and same-named `utility*` packages cannot collide):

  mint      Data_process/rec/data_process_rec.py on the shipped Epinions .mat files
            -> compact tests/golden/epinion2_dataset.npz   (SURVEY.md 8c "Dataset fixture")
  lightgcn  LightGCN_SPEX/code: Loader / LightGCN / Adam / batch_test on `tiny` and
            `epinion2`  -> G1..G6, G8, G9
  ngcf      NGCF_SPEX/code: Data.create_adj_mat + Model_Wrapper.forward -> G7, G10
  trust     LightGCN_SPEX/code: model_expert_s.LightGCN dual-task forward (rec + trust-path head) -> G11

Harness-side shims (reference files untouched):
  * torch.Tensor.cuda -> identity   (dataloader.py:176,222 hard-call .cuda())
  * np.asfarray re-added            (metrics.py:50,75; removed in NumPy 2)
  * sys.argv fixed before import    (argparse at import: batch_test.py:5-6)
  * cwd = scratch `<root>/code`     (relative data paths, dataloader.py:74)

Usage:  python oracle/gen_golden.py [--stage all|mint|lightgcn|ngcf|trust|epochs] [--skip-epinion-test]
"""
import argparse
import hashlib
import importlib.util
import os
import shutil
import subprocess
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRATCH = "/tmp/spex_oracle_scratch"
GOLD = os.path.join(REPO, "tests", "golden")

sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference


def sha(a):
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# --------------------------------------------------------------------------- mint
def stage_mint():
    """Run the reference preprocessing (data_process_rec.py:563-575 order) in scratch."""
    import random
    import numpy as np

    assert os.environ.get("PYTHONHASHSEED") == "0", "mint stage needs PYTHONHASHSEED=0"
    wd = os.path.join(SCRATCH, "Data_process", "rec")
    if os.path.isdir(SCRATCH):
        shutil.rmtree(SCRATCH)
    os.makedirs(os.path.join(wd, "epinion2"))
    for f in ("rating_with_timestamp.mat", "trust_with_timestamp.mat"):
        shutil.copy(os.path.join(REF, "Data_process", "rec", "epinion2", f), os.path.join(wd, "epinion2", f))
    os.chdir(wd)
    spec = importlib.util.spec_from_file_location(
        "ref_data_process_rec", os.path.join(REF, "Data_process", "rec", "data_process_rec.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.args = types.SimpleNamespace(root="epinion2")
    random.seed(2020)
    for step in ("change_format", "select_items", "select_users", "unify_index", "sort", "split",
                 "to_NGCF", "to_NCF", "to_LightGCN"):
        getattr(mod, step)()

    # compact binary form of what to_LightGCN wrote
    rec = os.path.join(SCRATCH, "LightGCN_SPEX", "data", "epinion2", "rec")
    train = np.loadtxt(os.path.join(rec, "epinion2.train.rating"), dtype=np.int64, usecols=(0, 1))
    test_users, test_pos = [], []
    with open(os.path.join(rec, "epinion2.test.rating")) as f:
        for line in f:
            a = line.split()
            if a:
                test_users.append(int(a[0])); test_pos.append(int(a[1]))
    # the reference dict-loader keeps the LAST item per user (dataloader.py:147) -> 100 rows/user, last = positive
    neg_users, negs = [], []
    with open(os.path.join(rec, "epinion2.test.negative")) as f:
        for line in f:
            a = line.split()
            if a:
                neg_users.append(int(a[0])); negs.append([int(x) for x in a[1:]])
    assert train.max() < 65536
    # <ds>.test.rating is, per user, the 99 rows of <ds>.test.negative followed by the held-out positive
    # (data_process_rec.py:246-254, 401-416): store positives + negatives once and rebuild both files from them.
    tu = np.asarray(test_users).reshape(-1, 100)
    ti = np.asarray(test_pos).reshape(-1, 100)
    negs = np.asarray(negs)
    assert (tu == tu[:, :1]).all() and (tu[:, 0] == np.asarray(neg_users)).all() and (ti[:, :99] == negs).all()
    np.savez_compressed(
        os.path.join(GOLD, "epinion2_dataset.npz"),
        train=train.astype(np.uint16),
        test_users=tu[:, 0].astype(np.uint16), test_pos=ti[:, 99].astype(np.uint16),
        test_neg=negs.astype(np.uint16),
    )
    print("mint: train", train.shape, "test.rating rows", len(test_users), "test.negative rows", len(negs))


# --------------------------------------------------------------------------- Epinion2 trust paths (config 5 at real scale)
def stage_paths():
    """The reference's own trust-path pipeline — Data_process/path/data_process_path.py:28-229 (process_selftrust,
    built_user, build_path, data_enhancement, neg_sample, data_refine; the `__main__` step-1 order at :276-283) — run
    in scratch on the `train_trust.txt` the `mint` stage's data_process_rec.py produced from the shipped
    trust_with_timestamp.mat.  random.seed(2020) and PYTHONHASHSEED=0 make it reproducible (the script samples with the
    global `random` and iterates sets).  --user_num is the dataset's user count (the padding / negative-pool id space,
    data_process_path.py:16,85-90,199), everything else the script's defaults (m = 50 walks per user, path_len 6).
    Stored compactly: padded uint16 path arrays + lengths; the test set is cut to its first 1 024 paths (each carries
    499 negatives + the target: 16 k test paths would be 16 MB of incompressible ids)."""
    import random
    import numpy as np
    import pickle
    assert os.environ.get("PYTHONHASHSEED") == "0", "paths stage needs PYTHONHASHSEED=0"
    rec = os.path.join(SCRATCH, "Data_process", "rec", "epinion2")
    assert os.path.exists(os.path.join(rec, "train_trust.txt")), "run --stage mint first"
    wd = os.path.join(SCRATCH, "Data_process", "path")
    if os.path.isdir(wd):
        shutil.rmtree(wd)
    os.makedirs(os.path.join(wd, "epinion2"))
    shutil.copy(os.path.join(rec, "train_trust.txt"), os.path.join(wd, "epinion2", "trust.txt"))   # :272
    os.chdir(wd)
    spec = importlib.util.spec_from_file_location(
        "ref_data_process_path", os.path.join(REF, "Data_process", "path", "data_process_path.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = np.load(os.path.join(GOLD, "epinion2_dataset.npz"))
    n_users = int(d["train"][:, 0].max()) + 1
    mod.args = types.SimpleNamespace(root="epinion2", m=50, path_len=6, user_num=n_users, item_num=1000000,
                                     max_count=10000, max_time=13, fix_len=769)
    random.seed(2020)
    import warnings
    warnings.simplefilter("ignore", DeprecationWarning)     # random.sample on a set (:202), still allowed on 3.10
    for step in ("process_selftrust", "built_user", "build_path", "data_enhancement", "neg_sample", "data_refine"):
        getattr(mod, step)()
    train = pickle.load(open(os.path.join(wd, "epinion2", "train.txt"), "rb"))
    test2 = pickle.load(open(os.path.join(wd, "epinion2", "test2.txt"), "rb"))

    def pack(paths):
        lens = np.asarray([len(p) for p in paths], np.uint8)
        arr = np.full((len(paths), int(lens.max())), n_users, np.uint16)
        for k, p in enumerate(paths):
            arr[k, :len(p)] = p
        return arr, lens
    tr_arr, tr_len = pack(train[0])
    n_test = min(1024, len(test2[0]))
    te_arr, te_len = pack(test2[0][:n_test])
    assert max(max(p) for p in train[0]) < n_users and n_users < 65535
    np.savez_compressed(os.path.join(GOLD, "trust_epinion2_paths.npz"), n_users=n_users,
                        train_paths=tr_arr, train_len=tr_len, train_targets=np.asarray(train[1], np.uint16),
                        test_paths=te_arr, test_len=te_len, test_targets=np.asarray(test2[1][:n_test], np.uint16),
                        test_negs=np.asarray(test2[2][:n_test], np.uint16), n_test_total=len(test2[0]))
    print("paths: train", tr_arr.shape, "len hist", np.bincount(tr_len), "test kept", n_test, "of", len(test2[0]))


# --------------------------------------------------------------------------- common
def install_shims():
    import numpy as np
    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self
    if not hasattr(np, "asfarray"):
        np.asfarray = lambda a, dtype=float: np.asarray(a, dtype=dtype)


def write_tiny(rec_dir, name="tiny"):
    """50 users x 60 items, 400 train edges, 1 positive + 20 negatives per test user (seed 0)."""
    import numpy as np
    rng = np.random.default_rng(0)
    U, I, E = 50, 60, 400
    pairs = set()
    # every user and the max item id appear so n_user / m_item are fixed
    for u in range(U):
        pairs.add((u, int(rng.integers(I))))
    pairs.add((0, I - 1))
    while len(pairs) < E:
        pairs.add((int(rng.integers(U)), int(rng.integers(I))))
    pairs = sorted(pairs)
    os.makedirs(rec_dir, exist_ok=True)
    with open(os.path.join(rec_dir, f"{name}.train.rating"), "w") as f:
        for u, i in pairs:
            f.write(f"{u} {i} 1\n")
    seen = {}
    for u, i in pairs:
        seen.setdefault(u, set()).add(i)
    with open(os.path.join(rec_dir, f"{name}.test.rating"), "w") as fr, \
            open(os.path.join(rec_dir, f"{name}.test.negative"), "w") as fn:
        for u in range(U):
            cand = [i for i in range(I) if i not in seen[u]]
            pick = rng.permutation(cand)[:21]
            fr.write(f"{u} {int(pick[0])} 1\n")
            fn.write(str(u) + "".join(f" {int(x)}" for x in pick[1:]) + "\n")
    return np.asarray(pairs, np.int64)


def coo_to_csr_arrays(sp_tensor):
    import numpy as np
    t = sp_tensor.coalesce()
    idx = t.indices().numpy()
    val = t.values().numpy().astype(np.float32)
    n = t.shape[0]
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, idx[0] + 1, 1)
    rowptr = np.cumsum(rowptr)
    # coalesced COO is row-major sorted -> already CSR order
    return rowptr.astype(np.int32), idx[1].astype(np.int32), val


def xavier_uniform_np(rows, dim, rng):
    """The build's own seeded initialiser (mirrors spex_amd.init.xavier_uniform_np)."""
    import numpy as np
    b = np.sqrt(6.0 / (rows + dim))
    return rng.uniform(-b, b, size=(rows, dim)).astype(np.float32)


def sampled(rows_idx, a):
    return a[rows_idx].copy()


# --------------------------------------------------------------------------- lightgcn
def stage_lightgcn(skip_epinion_test=False):
    import numpy as np
    import torch
    install_shims()
    torch.set_num_threads(int(os.environ.get("SPEX_MINT_THREADS", "8")))
    code = os.path.join(SCRATCH, "LightGCN_SPEX", "code")
    os.makedirs(code, exist_ok=True)
    os.chdir(code)
    tiny_pairs = write_tiny(os.path.join(SCRATCH, "LightGCN_SPEX", "data", "tiny", "rec"))
    sys.path.insert(0, os.path.join(REF, "LightGCN_SPEX", "code"))
    sys.argv = ["main_rec.py", "--dataset", "tiny"]
    import lg_parser
    import utility1.dataloader as ref_dl
    import utility1.model as ref_model
    import utility1.utils as ref_utils
    import utility1.batch_test as ref_bt
    import utility1.metrics as ref_metrics

    # ---- G5a: metric unit vectors (batch_test.py:72-90, metrics.py:61-80), incl. ties
    mrng = np.random.default_rng(5)
    cases = []
    for c in range(12):
        n = 100 if c < 8 else 21
        scores = mrng.normal(size=n).astype(np.float32)
        if c % 3 == 1:                      # heavy ties
            scores = np.round(scores, 1)
        if c == 5:
            scores[:] = 0.25                # everything tied
        items = mrng.permutation(5000)[:n].astype(np.int64)
        pos = [int(items[-1])]
        rating = {int(items[i]): float(scores[i]) for i in range(n)}
        r = ref_bt.ranklist_by_heapq(pos, rating)
        perf = ref_bt.get_performance(pos, r)
        cases.append((scores, items, np.asarray(r, np.int8), perf["recall"], perf["ndcg"]))
    np.savez_compressed(
        os.path.join(GOLD, "g5_metric_cases.npz"),
        **{f"scores_{k}": c[0] for k, c in enumerate(cases)},
        **{f"items_{k}": c[1] for k, c in enumerate(cases)},
        **{f"r_{k}": c[2] for k, c in enumerate(cases)},
        **{f"recall_{k}": c[3] for k, c in enumerate(cases)},
        **{f"ndcg_{k}": c[4] for k, c in enumerate(cases)},
        n_cases=len(cases))

    for ds in ("tiny", "epinion2"):
        full = ds == "tiny"
        args = lg_parser.parse_args_r()
        args.dataset = ds
        cache = os.path.join(SCRATCH, "LightGCN_SPEX", "data", ds, "s_pre_adj_mat.npz")
        if os.path.exists(cache):
            os.remove(cache)
        ref_utils.set_seed(args.seed)
        dataset = ref_dl.Loader(args)
        model = ref_model.LightGCN(args, dataset)
        U1, I = dataset.n_user + 1, dataset.m_item
        N = U1 + I
        out = {"n_user": dataset.n_user, "m_item": dataset.m_item}

        # ---- G1 adjacency (dataloader.py:187-225)
        rowptr, col, val = coo_to_csr_arrays(model.Graph)
        out.update(nnz=len(col), rowptr_sha=sha(rowptr), col_sha=sha(col), val_sha=sha(val))
        grng = np.random.default_rng(11)
        if full:
            tus = list(dataset.testRatings.keys())
            out.update(rowptr=rowptr, col=col, val=val, train_pairs=tiny_pairs, test_users=np.asarray(tus),
                       test_pos=np.asarray([dataset.testRatings[u][0] for u in tus]),
                       test_neg=np.asarray([dataset.testNegatives[u] for u in tus]))
        else:
            eidx = np.sort(grng.choice(len(col), 4096, replace=False))
            out.update(edge_idx=eidx, edge_col=col[eidx], edge_val=val[eidx], rowptr=rowptr)

        # ---- G2 propagation (model.py:66-97)
        if full:
            E0 = torch.cat([model.embedding_user.weight, model.embedding_item.weight]).detach().clone()
        else:
            rng = np.random.default_rng(2020)
            uw = xavier_uniform_np(U1, args.recdim, rng)
            iw = xavier_uniform_np(I, args.recdim, rng)
            with torch.no_grad():
                model.embedding_user.weight.copy_(torch.from_numpy(uw))
                model.embedding_item.weight.copy_(torch.from_numpy(iw))
            E0 = torch.from_numpy(np.concatenate([uw, iw]))
        layers = [E0]
        for _ in range(args.layer):
            layers.append(torch.sparse.mm(model.Graph, layers[-1]))
        model.eval()
        with torch.no_grad():
            users_out, items_out = model.computer()
        light_out = torch.cat([users_out, items_out]).numpy()
        srow = np.sort(grng.choice(N, min(512, N), replace=False))
        out["sample_rows"] = srow
        for l, E in enumerate(layers):
            e = E.numpy()
            if full:
                out[f"E{l}"] = e
            else:
                out[f"E{l}_rows"] = sampled(srow, e)
            out[f"E{l}_colsum"] = e.astype(np.float64).sum(0)
            out[f"E{l}_fro"] = np.sqrt((e.astype(np.float64) ** 2).sum())
        if full:
            out["light_out"] = light_out
        else:
            out["light_out_rows"] = sampled(srow, light_out)
        out["light_out_colsum"] = light_out.astype(np.float64).sum(0)
        out["light_out_fro"] = np.sqrt((light_out.astype(np.float64) ** 2).sum())

        # ---- G3 scoring + loss + grads (model.py:111-121), train mode, no dropout
        brng = np.random.default_rng(2020)
        B = 256
        nb = 5
        bu = brng.integers(0, dataset.n_user, size=(nb, B)).astype(np.int64)
        bi = brng.integers(0, dataset.m_item, size=(nb, B)).astype(np.int64)
        bl = (brng.random((nb, B)) < 1.0 / 6.0).astype(np.int64)
        out.update(batch_users=bu, batch_items=bi, batch_labels=bl)
        model.train()
        model.zero_grad()
        gamma = model(torch.from_numpy(bu[0]), torch.from_numpy(bi[0]), torch.from_numpy(bl[0]), flag=1)
        loss = model(torch.from_numpy(bu[0]), torch.from_numpy(bi[0]), torch.from_numpy(bl[0]), flag=0)
        loss.backward()
        gu = model.embedding_user.weight.grad.numpy().copy()
        gi = model.embedding_item.weight.grad.numpy().copy()
        gall = np.concatenate([gu, gi])
        out.update(g3_gamma=gamma.detach().numpy(), g3_loss=np.float32(loss.item()))
        if full:
            out["g3_grad"] = gall
        else:
            out["g3_grad_rows"] = sampled(srow, gall)
        out["g3_grad_colsum"] = gall.astype(np.float64).sum(0)
        out["g3_grad_fro"] = np.sqrt((gall.astype(np.float64) ** 2).sum())

        # ---- G4 Adam (main_rec.py:23,30-37): tables after 1, 2, 5 steps
        model.zero_grad()
        opt = torch.optim.Adam(model.parameters(), lr=args.lr)
        losses = []
        for s in range(nb):
            opt.zero_grad()
            loss = model(torch.from_numpy(bu[s]), torch.from_numpy(bi[s]), torch.from_numpy(bl[s]), flag=0)
            loss.backward()
            opt.step()
            losses.append(loss.item())
            if s + 1 in (1, 2, 5):
                w = torch.cat([model.embedding_user.weight, model.embedding_item.weight]).detach().numpy()
                if full:
                    out[f"g4_w_step{s + 1}"] = w.copy()
                else:
                    out[f"g4_w_step{s + 1}_rows"] = sampled(srow, w)
                out[f"g4_w_step{s + 1}_colsum"] = w.astype(np.float64).sum(0)
        out["g4_losses"] = np.asarray(losses, np.float32)

        # ---- G5b end-to-end test() (batch_test.py:12-40) with the post-Adam tables
        wfin = torch.cat([model.embedding_user.weight, model.embedding_item.weight]).detach().numpy()
        out["g5_w_sha"] = sha(wfin)
        if full or not skip_epinion_test:
            model.eval()
            ret = ref_bt.test(model, dataset.testRatings, dataset.testNegatives)
            out.update(g5_recall=ret["recall"], g5_ndcg=ret["ndcg"])
            # and per-user scores for the first 64 test users
            tu = list(dataset.testRatings.keys())[:64]
            sc = []
            with torch.no_grad():
                for u in tu:
                    its = dataset.testNegatives[u] + dataset.testRatings[u]
                    sc.append(model(torch.full((len(its),), u).long(), torch.tensor(its).long(), None, flag=1).numpy())
            out.update(g5_users=np.asarray(tu), g5_scores=np.stack(sc))
            print(ds, "test():", ret)

        # ---- G9 dropout with injected mask (model.py:46-55)
        keep = 0.6
        drng = np.random.default_rng(9)
        G = model.Graph
        rnd = drng.random(G._nnz()).astype(np.float32)
        keepmask = torch.from_numpy((rnd + np.float32(keep)).astype(np.int32).astype(bool))
        index = G.indices().t()[keepmask]
        values = G.values()[keepmask] / keep
        g = torch.sparse_coo_tensor(index.t(), values, G.size())
        cur = E0
        dl = [E0]
        for _ in range(args.layer):
            cur = torch.sparse.mm(g, cur)
            dl.append(cur)
        dmean = torch.mean(torch.stack(dl, dim=1), dim=1).numpy()
        out["g9_keep"] = np.float32(keep)
        out["g9_rand"] = rnd if full else np.zeros(0, np.float32)
        out["g9_seed"] = 9
        if full:
            out["g9_light_out"] = dmean
        else:
            out["g9_light_out_rows"] = sampled(srow, dmean)
        out["g9_light_out_colsum"] = dmean.astype(np.float64).sum(0)

        # ---- G6 sampler (dataloader.py:250-265) — tiny only (pure-Python loop is slow on epinion2)
        if full:
            td = ref_dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
            np.random.seed(2020)
            td.ng_sample()
            out["g6_neg"] = np.asarray(td.features_ng, np.int64)
            out["g6_len"] = len(td)
            out["g6_item0"] = np.asarray(td[0], np.int64)
            out["g6_item_last"] = np.asarray(td[len(td) - 1], np.int64)

        # ---- G8 expert gating (model_expert_s.py:154-161), flag=1 scores
        import utility1.model_expert_s as ref_ex
        ref_utils.set_seed(args.seed)
        ex = ref_ex.LightGCN(args, dataset)
        with torch.no_grad():
            ex.embedding_user.weight.copy_(E0[:U1]); ex.embedding_item.weight.copy_(E0[U1:])
        ex.eval()
        with torch.no_grad():
            gx = ex(torch.from_numpy(bu[0]), torch.from_numpy(bi[0]), None, None, None, flag=1)
        out.update(g8_att_exp1=ex.att_exp1.detach().numpy(), g8_att_exp2=ex.att_exp2.detach().numpy(),
                   g8_gamma=gx.numpy())

        np.savez_compressed(os.path.join(GOLD, f"lightgcn_{ds}.npz"), **out)
        print(ds, "written; N", N, "nnz", len(col), "loss0", float(out["g3_loss"]))


# --------------------------------------------------------------------------- ngcf
def stage_ngcf():
    import numpy as np
    import torch
    install_shims()
    torch.set_num_threads(int(os.environ.get("SPEX_MINT_THREADS", "8")))
    code = os.path.join(SCRATCH, "NGCF_SPEX", "code")
    os.makedirs(code, exist_ok=True)
    os.chdir(code)
    # tiny dataset in NGCF's format (load_data.py:29-31): train.txt `u i1 i2 ...`, test.txt `u i`, negative.txt
    tiny_rec = os.path.join(SCRATCH, "NGCF_SPEX", "data", "tiny", "rec")
    os.makedirs(tiny_rec, exist_ok=True)
    lg_tiny = os.path.join(SCRATCH, "LightGCN_SPEX", "data", "tiny", "rec")
    pairs = np.loadtxt(os.path.join(lg_tiny, "tiny.train.rating"), dtype=np.int64, usecols=(0, 1))
    with open(os.path.join(tiny_rec, "train.txt"), "w") as f:
        for u in np.unique(pairs[:, 0]):
            f.write(str(u) + "".join(f" {i}" for i in pairs[pairs[:, 0] == u, 1]) + "\n")
    shutil.copy(os.path.join(lg_tiny, "tiny.test.negative"), os.path.join(tiny_rec, "negative.txt"))
    with open(os.path.join(lg_tiny, "tiny.test.rating")) as f, open(os.path.join(tiny_rec, "test.txt"), "w") as g:
        for line in f:
            a = line.split()
            g.write(f"{a[0]} {a[1]}\n")

    sys.path.insert(0, os.path.join(REF, "NGCF_SPEX", "code"))
    sys.argv = ["main_rec.py"]
    import utility.load_data as ref_ld

    # Model_Wrapper lives inside main_rec.py next to import-time side effects (log dir, batch_test singletons,
    # main_rec.py:13-34); exec only the class body's source text out of the file, in a namespace that provides
    # what it closes over.  Nothing is written to the repo.
    src = open(os.path.join(REF, "NGCF_SPEX", "code", "main_rec.py")).read()
    start = src.index("class Model_Wrapper")
    end = src.index("def train(model, optimizer)")
    import torch.nn as nn
    import torch.nn.functional as F
    margs = ref_ld.args
    ns = {"nn": nn, "torch": torch, "F": F, "np": np, "args": margs,
          "trans_to_cuda": lambda v: v}
    exec(compile(src[start:end], "<ref Model_Wrapper>", "exec"), ns)
    Model_Wrapper = ns["Model_Wrapper"]

    for ds in ("tiny", "epinion2"):
        full = ds == "tiny"
        path = os.path.join(SCRATCH, "NGCF_SPEX", "data", ds)
        for c in ("s_adj_mat.npz", "s_norm_adj_mat.npz", "s_mean_adj_mat.npz"):
            p = os.path.join(path, "rec", c)
            if os.path.exists(p):
                os.remove(p)
        data = ref_ld.Data(path=path, batch_size=256)
        plain, norm, mean = data.get_adj_mat()
        norm = norm.tocsr().astype(np.float32)
        norm.sort_indices()
        out = {"n_users": data.n_users, "n_items": data.n_items, "nnz": norm.nnz,
               "rowptr_sha": sha(norm.indptr.astype(np.int32)), "col_sha": sha(norm.indices.astype(np.int32)),
               "val_sha": sha(norm.data.astype(np.float32)), "rowptr": norm.indptr.astype(np.int32)}
        grng = np.random.default_rng(11)
        if full:
            out.update(col=norm.indices.astype(np.int32), val=norm.data.astype(np.float32), train_pairs=pairs)
        else:
            eidx = np.sort(grng.choice(norm.nnz, 4096, replace=False))
            out.update(edge_idx=eidx, edge_col=norm.indices[eidx].astype(np.int32), edge_val=norm.data[eidx].astype(np.float32))

        torch.manual_seed(2020)
        m = Model_Wrapper(data_config={"n_users": data.n_users, "n_items": data.n_items, "norm_adj": norm},
                          device=torch.device("cpu"))
        N = data.n_users + data.n_items
        rng = np.random.default_rng(2020)
        if not full:
            uw = xavier_uniform_np(data.n_users + 1, 64, rng)
            iw = xavier_uniform_np(data.n_items, 64, rng)
            with torch.no_grad():
                m.user_embedding.weight.copy_(torch.from_numpy(uw)); m.item_embedding.weight.copy_(torch.from_numpy(iw))
        else:
            out["user_w"] = m.user_embedding.weight.detach().numpy().copy()
            out["item_w"] = m.item_embedding.weight.detach().numpy().copy()
        # the small dense weights are always stored (4 x 16 KB)
        out.update(W_gc=m.GC_Linear_list[0].weight.detach().numpy().copy(), b_gc=m.GC_Linear_list[0].bias.detach().numpy().copy(),
                   W_bi=m.Bi_Linear_list[0].weight.detach().numpy().copy(), b_bi=m.Bi_Linear_list[0].bias.detach().numpy().copy())
        m.eval()  # dropout off (main_rec.py:81)
        with torch.no_grad():
            ua, ia = m(None, None, None, flag=1)
        allemb = torch.cat([ua, ia]).numpy()
        srow = np.sort(grng.choice(N, min(512, N), replace=False))
        out["sample_rows"] = srow
        if full:
            out["all_emb"] = allemb
        else:
            out["all_emb_rows"] = allemb[srow]
        out["all_emb_colsum"] = allemb.astype(np.float64).sum(0)
        out["all_emb_fro"] = np.sqrt((allemb.astype(np.float64) ** 2).sum())

        brng = np.random.default_rng(2020)
        B = 256
        bu = brng.integers(0, data.n_users, size=B).astype(np.int64)
        bi = brng.integers(0, data.n_items, size=B).astype(np.int64)
        bl = (brng.random(B) < 1.0 / 6.0).astype(np.float32)
        m.zero_grad()
        loss = m(torch.from_numpy(bu), torch.from_numpy(bi), torch.from_numpy(bl), flag=0)
        loss.backward()
        gu = m.user_embedding.weight.grad.numpy(); gi = m.item_embedding.weight.grad.numpy()
        gall = np.concatenate([gu, gi])
        out.update(batch_users=bu, batch_items=bi, batch_labels=bl, loss=np.float32(loss.item()),
                   grad_W_gc=m.GC_Linear_list[0].weight.grad.numpy().copy(), grad_b_gc=m.GC_Linear_list[0].bias.grad.numpy().copy(),
                   grad_W_bi=m.Bi_Linear_list[0].weight.grad.numpy().copy(), grad_b_bi=m.Bi_Linear_list[0].bias.grad.numpy().copy(),
                   grad_emb_colsum=gall.astype(np.float64).sum(0), grad_emb_fro=np.sqrt((gall.astype(np.float64) ** 2).sum()))
        srow_g = np.sort(grng.choice(N + 1, min(512, N + 1), replace=False))
        out["grad_sample_rows"] = srow_g
        if full:
            out["grad_emb"] = gall
        else:
            out["grad_emb_rows"] = gall[srow_g]
        np.savez_compressed(os.path.join(GOLD, f"ngcf_{ds}.npz"), **out)
        print("ngcf", ds, "N", N, "nnz", norm.nnz, "loss", float(loss.item()))


# --------------------------------------------------------------------------- trust head
def stage_trust():
    """Dual-task model (model_expert_s.py) on `tiny` with synthetic trust paths: parameters by seed, rec + trust
    losses (flag=0), trust scores (flag=2), gradients, trust_test5 metrics."""
    import numpy as np
    import torch
    install_shims()
    torch.set_num_threads(int(os.environ.get("SPEX_MINT_THREADS", "8")))
    code = os.path.join(SCRATCH, "LightGCN_SPEX", "code")
    os.makedirs(code, exist_ok=True)
    os.chdir(code)
    write_tiny(os.path.join(SCRATCH, "LightGCN_SPEX", "data", "tiny", "rec"))
    sys.path.insert(0, os.path.join(REF, "LightGCN_SPEX", "code"))
    sys.argv = ["main_auto_expert_s.py", "--dataset", "tiny"]
    import lg_parser
    import utility1.dataloader as ref_dl
    import utility1.utils as ref_utils
    import utility1.model_expert_s as ref_ex
    from utility2.utils import Data
    from utility2.batch_test_gnn import trust_test5

    args = lg_parser.parse_args_r()
    args.dataset = "tiny"
    cache = os.path.join(SCRATCH, "LightGCN_SPEX", "data", "tiny", "s_pre_adj_mat.npz")
    if os.path.exists(cache):
        os.remove(cache)
    rng = np.random.default_rng(77)
    n_users = 50
    def rand_paths(n):
        paths, targets = [], []
        for _ in range(n):
            l = int(rng.integers(2, 7))                      # 2..6 nodes (data_process_path.py --path_len 6)
            p = rng.choice(n_users, size=l, replace=False).tolist()
            paths.append(p)
            targets.append(int(rng.integers(n_users)))
        return paths, targets
    tr_paths, tr_targets = rand_paths(40)
    te_paths, te_targets = rand_paths(12)
    te_negs = []
    for t in te_targets:
        cand = [u for u in range(n_users) if u != t]
        te_negs.append(rng.permutation(cand)[:49].tolist() + [t])   # 49 negatives first, target last (trust_test5 :36-37 takes topk(50))
    ref_utils.set_seed(args.seed)
    dataset = ref_dl.Loader(args)
    train_data2 = Data((tr_paths, tr_targets), dataset.n_users, shuffle=False)
    test_data2 = Data((te_paths, te_targets, te_negs), dataset.n_users, shuffle=False, test=True)
    model = ref_ex.LightGCN(args, dataset)
    out = {"state_" + k.replace(".", "__"): v.detach().numpy().copy() for k, v in model.state_dict().items()}
    out.update(train_inputs=train_data2.inputs, train_mask=train_data2.mask, train_targets=train_data2.targets,
               test_inputs=test_data2.inputs, test_mask=test_data2.mask, test_targets=test_data2.targets,
               test_negs=test_data2.neg, n_users=dataset.n_users)
    g = np.load(os.path.join(GOLD, "lightgcn_tiny.npz"))
    bu, bi, bl = (torch.from_numpy(g[k][0]) for k in ("batch_users", "batch_items", "batch_labels"))
    sl = np.arange(0, 30)
    model.train()
    model.zero_grad()
    loss1, loss2 = model(bu, bi, bl, sl, train_data2, flag=0)
    (loss1 + loss2).backward()
    out.update(slice_indices=sl, loss1=np.float32(loss1.item()), loss2=np.float32(loss2.item()))
    for name, p in model.named_parameters():
        if p.grad is not None:
            out["grad_" + name.replace(".", "__")] = p.grad.numpy().copy()
    model.eval()
    with torch.no_grad():
        scores, negs = model(None, None, None, np.arange(12), test_data2, flag=2)
        out.update(trust_scores=scores.numpy(), trust_negs=negs.numpy())
        model.batch_size = 5                                  # exercises generate_batch's ragged last slice
        out["trust_test5"] = np.asarray(trust_test5(model, test_data2), np.float64)
    np.savez_compressed(os.path.join(GOLD, "trust_tiny.npz"), **out)
    print("trust tiny: loss1 %.6f loss2 %.6f test5 %s" % (loss1.item(), loss2.item(), out["trust_test5"]))


# --------------------------------------------------------------------------- whole training run (G12)
def stage_epochs(ds="tiny", n_epochs=3, dropout=None, max_steps=None, n_layers=None):
    """G12: the reference's own training run — main_rec.py:15-37,50 executed with the reference's modules (set_seed,
    Loader, LightTrainData.ng_sample, DataLoader(256, shuffle=True), model.LightGCN, torch Adam, test()) for three
    epochs on `tiny`: per-epoch loss sums, per-epoch recall / ndcg, the trained tables.  main_rec.py itself runs at
    import and writes logs, so its Train() / Test() bodies are driven from here, line for line."""
    import numpy as np
    import torch
    from torch.utils.data import DataLoader
    install_shims()
    torch.set_num_threads(int(os.environ.get("SPEX_MINT_THREADS", "8")))
    code = os.path.join(SCRATCH, "LightGCN_SPEX", "code")
    os.makedirs(code, exist_ok=True)
    os.chdir(code)
    if ds == "tiny":
        write_tiny(os.path.join(SCRATCH, "LightGCN_SPEX", "data", "tiny", "rec"))
    cache = os.path.join(SCRATCH, "LightGCN_SPEX", "data", ds, "s_pre_adj_mat.npz")
    if os.path.exists(cache):
        os.remove(cache)
    sys.path.insert(0, os.path.join(REF, "LightGCN_SPEX", "code"))
    sys.argv = ["main_rec.py", "--dataset", ds]
    if dropout is not None:                # README.md:119-123: "--dropout=1 --keepprob=0.3"
        sys.argv += ["--dropout", "1", "--keepprob", str(dropout)]
    if n_layers is not None:               # lg_parser.py:10 "--layer" (default 3): a run of another depth -> lightgcn_{ds}_L{n}.npz,
        sys.argv += ["--layer", str(n_layers)]      # with every step's loss
    import lg_parser
    import utility1.dataloader as ref_dl
    import utility1.model as ref_model
    import utility1.utils as ref_utils
    from utility1.batch_test import test as ref_test
    args = lg_parser.parse_args_r()
    ref_utils.set_seed(args.seed)                                                   # main_rec.py:15
    device = torch.device("cpu")
    dataset = ref_dl.Loader(args)                                                   # :18
    train_dataset = ref_dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)   # :19
    train_loader = DataLoader(train_dataset, batch_size=256, shuffle=True)          # :20
    Recmodel = ref_model.LightGCN(args, dataset).to(device)                         # :22
    optimizer = torch.optim.Adam(Recmodel.parameters(), lr=args.lr)                 # :23
    losses, recalls, ndcgs, first_batch = [], [], [], None
    step_losses, mask0 = [], {}
    nnz = Recmodel.Graph._nnz()
    for epoch in range(n_epochs):
        train_loader.dataset.ng_sample()                                            # :26
        Recmodel.train()
        total_loss = 0.0
        for step, data in enumerate(train_loader):                                  # :30-37
            if max_steps is not None and step == max_steps:
                break
            optimizer.zero_grad()
            user, item, label = data
            if first_batch is None:
                first_batch = np.stack([user.numpy(), item.numpy(), label.numpy()])
            if dropout is not None and not mask0:
                # what model.py:50-51 is about to draw for this step (the generator is put back where it was)
                st = torch.get_rng_state()
                keep = (torch.rand(nnz) + args.keepprob).int().bool().numpy()
                torch.set_rng_state(st)
                mask0 = dict(mask0_sha=sha(keep.astype(np.uint8)), mask0_kept=int(keep.sum()), mask0_head=keep[:4096].copy())
            loss = Recmodel(users=user.to(device), items=item.to(device), labels=label.to(device), flag=0)
            loss.backward()
            total_loss += loss.item()
            if dropout is not None or n_layers is not None:
                step_losses.append(loss.item())
            optimizer.step()
        losses.append(total_loss)
        Recmodel.eval()
        with torch.no_grad():                                                       # :49-50
            ret = ref_test(Recmodel, dataset.testRatings, dataset.testNegatives)
        recalls.append(ret["recall"]); ndcgs.append(ret["ndcg"])
    uw, iw = Recmodel.embedding_user.weight.detach().numpy(), Recmodel.embedding_item.weight.detach().numpy()
    extra = {}
    if ds != "tiny":                       # full-size run: keep the fixture small (sampled rows + column sums)
        rows_u = np.sort(np.random.default_rng(1).choice(uw.shape[0], 256, replace=False))
        rows_i = np.sort(np.random.default_rng(2).choice(iw.shape[0], 256, replace=False))
        extra = dict(rows_u=rows_u, rows_i=rows_i, user_w_colsum=uw.astype(np.float64).sum(0),
                     item_w_colsum=iw.astype(np.float64).sum(0))
        uw, iw = uw[rows_u], iw[rows_i]
    name = f"lightgcn_{ds}_epochs.npz"
    if dropout is not None:                # G12-dropout: per-step losses of the whole (capped) run, the first step's keep mask
        name = f"lightgcn_{ds}_dropout.npz"
        extra.update(mask0, keepprob=args.keepprob, step_losses=np.asarray(step_losses, np.float64),
                     max_steps=-1 if max_steps is None else max_steps, nnz=nnz)
    if n_layers is not None:
        assert dropout is None and int(args.layer) == n_layers
        name = f"lightgcn_{ds}_L{n_layers}.npz"
        extra.update(n_layers=n_layers, step_losses=np.asarray(step_losses, np.float64), max_steps=-1 if max_steps is None else max_steps)
    np.savez_compressed(os.path.join(GOLD, name), seed=args.seed, lr=args.lr,
                        losses=np.asarray(losses, np.float64), recall=np.asarray(recalls, np.float64),
                        ndcg=np.asarray(ndcgs, np.float64), first_batch=first_batch, user_w=uw, item_w=iw, **extra)
    print("epochs: losses", losses, "recall", recalls[-1], "ndcg", ndcgs[-1])


def stage_epochs_dual():
    """G13: the reference's dual-task training run — main_auto_expert_s.py:22-120 executed with the reference's
    modules (rec loader + shuffled DataLoader, the trust paths of G11 wrapped in utility2.utils.Data, model_expert_s,
    path selection per batch incl. random.sample, the uncertainty-weighted loss, torch Adam, rec_test + trust_test5)
    for two epochs on `tiny`: per-epoch loss sums of both tasks, the learned task weights, both tasks' metrics."""
    import random
    from collections import defaultdict
    import numpy as np
    import torch
    from torch.utils.data import DataLoader
    install_shims()
    torch.set_num_threads(int(os.environ.get("SPEX_MINT_THREADS", "8")))
    code = os.path.join(SCRATCH, "LightGCN_SPEX", "code")
    os.makedirs(code, exist_ok=True)
    os.chdir(code)
    write_tiny(os.path.join(SCRATCH, "LightGCN_SPEX", "data", "tiny", "rec"))
    cache = os.path.join(SCRATCH, "LightGCN_SPEX", "data", "tiny", "s_pre_adj_mat.npz")
    if os.path.exists(cache):
        os.remove(cache)
    sys.path.insert(0, os.path.join(REF, "LightGCN_SPEX", "code"))
    sys.argv = ["main_auto_expert_s.py", "--dataset", "tiny"]
    import lg_parser
    import utility1.dataloader as ref_dl
    import utility1.utils as ref_utils
    import utility1.model_expert_s as ref_ex
    from utility1.batch_test import rec_test
    from utility2.utils import Data
    from utility2.batch_test_gnn import trust_test5
    args = lg_parser.parse_args_r()
    g11 = np.load(os.path.join(GOLD, "trust_tiny.npz"))
    lens, tl = g11["train_mask"].sum(1), g11["test_mask"].sum(1)
    raw_train = ([r[:l].tolist() for r, l in zip(g11["train_inputs"], lens)], g11["train_targets"].tolist())
    raw_test = ([r[:l].tolist() for r, l in zip(g11["test_inputs"], tl)], g11["test_targets"].tolist(), g11["test_negs"].tolist())
    ref_utils.set_seed(args.seed)                                                   # main_auto_expert_s.py:22
    device = torch.device("cpu")
    dataset = ref_dl.Loader(args)                                                   # :34
    train_dataset = ref_dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
    train_loader = DataLoader(train_dataset, batch_size=256, shuffle=True)          # :36
    user_path_indx = defaultdict(list)                                              # :42-46
    path = raw_train[0]
    for i, p in zip(range(len(path)), path):
        user_path_indx[p[0]].append(i)
    train_data2 = Data(raw_train, dataset.n_users, shuffle=False)                   # :47-48
    test_data2 = Data(raw_test, dataset.n_users, shuffle=False, test=True)
    trust_batch_size = len(path) // len(train_loader)                               # :49
    Recmodel = ref_ex.LightGCN(args, dataset).to(device)                            # :51-52
    optimizer = torch.optim.Adam(Recmodel.parameters(), lr=args.lr)
    out = dict(loss1=[], loss2=[], task_weights=[], rec_recall=[], rec_ndcg=[], trust=[], n_paths=[])
    for epoch in range(2):
        train_loader.dataset.ng_sample()                                            # :56
        Recmodel.train()
        t1 = t2 = 0.0
        for data in train_loader:                                                   # :60-87
            optimizer.zero_grad()
            user, item, label = data
            unique_user = set(user.numpy().tolist())
            path_index = []
            for u in unique_user:
                path_index.extend(user_path_indx[u])
            if len(path_index) > trust_batch_size * 3:
                path_index = random.sample(path_index, trust_batch_size * 3)
            out["n_paths"].append(len(path_index))
            loss1, loss2 = Recmodel(users=user.to(device), items=item.to(device), labels=label.to(device),
                                    slice_indices=np.array(list(path_index), dtype=int), trust_data=train_data2, flag=0)
            T, n_rec, T_rec = len(path_index), 5, len(user)
            precision1 = torch.exp(-2 * Recmodel.task_weights[0])
            precision2 = torch.exp(-2 * Recmodel.task_weights[1])
            loss = precision1 * loss1 + precision2 * loss2 + 2 * (n_rec + 1) * T_rec * Recmodel.task_weights[0] \
                + T * Recmodel.task_weights[1]
            loss.backward()
            t1 += loss1.item()
            t2 += loss2.item()
            optimizer.step()
        out["loss1"].append(t1); out["loss2"].append(t2)
        out["task_weights"].append(Recmodel.task_weights.detach().numpy().copy())
        Recmodel.eval()
        with torch.no_grad():                                                       # :98-114
            ret = rec_test(Recmodel, dataset.testRatings, dataset.testNegatives)
            out["rec_recall"].append(ret["recall"]); out["rec_ndcg"].append(ret["ndcg"])
            out["trust"].append(np.asarray(trust_test5(Recmodel, test_data2), np.float64))
    np.savez_compressed(os.path.join(GOLD, "dual_tiny_epochs.npz"), seed=args.seed,
                        **{k: np.asarray(v, np.float64) for k, v in out.items()},
                        user_w=Recmodel.embedding_user.weight.detach().numpy(), w=Recmodel.w.detach().numpy())
    print("dual epochs: loss1", out["loss1"], "loss2", out["loss2"], "tw", out["task_weights"][-1], "trust", out["trust"][-1])


# --------------------------------------------------------------------------- dual-task at Epinion2 scale (config 5)
def _epinion2_trust_raw():
    """(train paths, targets), (test paths, targets, negatives) as the lists the reference pickles
    (data_process_path.py:184-188,207-208), rebuilt from the compact fixture the `paths` stage wrote."""
    import numpy as np
    t = np.load(os.path.join(GOLD, "trust_epinion2_paths.npz"))
    tr = ([r[:l].tolist() for r, l in zip(t["train_paths"].astype(np.int64), t["train_len"])],
          t["train_targets"].astype(np.int64).tolist())
    te = ([r[:l].tolist() for r, l in zip(t["test_paths"].astype(np.int64), t["test_len"])],
          t["test_targets"].astype(np.int64).tolist(), t["test_negs"].astype(np.int64).tolist())
    return tr, te


def _dual_setup(argv0="main_auto_expert_s.py"):
    import torch
    install_shims()
    torch.set_num_threads(int(os.environ.get("SPEX_MINT_THREADS", "8")))
    code = os.path.join(SCRATCH, "LightGCN_SPEX", "code")
    os.makedirs(code, exist_ok=True)
    os.chdir(code)
    cache = os.path.join(SCRATCH, "LightGCN_SPEX", "data", "epinion2", "s_pre_adj_mat.npz")
    if os.path.exists(cache):
        os.remove(cache)
    sys.path.insert(0, os.path.join(REF, "LightGCN_SPEX", "code"))
    sys.argv = [argv0, "--dataset", "epinion2"]


def stage_trust_epinion2():
    """G11 at Epinion2 scale: model_expert_s.LightGCN (seed 2020) on the Epinion2 graph and the reference-minted trust
    paths: both losses at flag=0 for a fixed rec batch + the paths of its users, every parameter's gradient (tables:
    sampled rows + column sums + norm), trust scores at the test negatives, trust_test5 on the 1 024 kept test paths."""
    import numpy as np
    import torch
    _dual_setup()
    import lg_parser
    import utility1.dataloader as ref_dl
    import utility1.utils as ref_utils
    import utility1.model_expert_s as ref_ex
    from utility2.utils import Data
    from utility2.batch_test_gnn import trust_test5
    from collections import defaultdict
    args = lg_parser.parse_args_r()
    raw_train, raw_test = _epinion2_trust_raw()
    ref_utils.set_seed(args.seed)
    dataset = ref_dl.Loader(args)
    train2 = Data(raw_train, dataset.n_users, shuffle=False)
    test2 = Data(raw_test, dataset.n_users, shuffle=False, test=True)
    model = ref_ex.LightGCN(args, dataset)
    g = np.load(os.path.join(GOLD, "lightgcn_epinion2.npz"))
    bu, bi, bl = (torch.from_numpy(g[k][0]) for k in ("batch_users", "batch_items", "batch_labels"))
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    sl = []
    for u in sorted(set(bu.numpy().tolist())):
        sl.extend(by_user[u])
    sl = np.asarray(sl[:192], dtype=int)              # a batch's worth of paths, longer than the driver's cap of 15
    model.train()
    model.zero_grad()
    loss1, loss2 = model(bu, bi, bl, sl, train2, flag=0)
    (loss1 + loss2).backward()
    out = dict(slice_indices=sl, loss1=np.float32(loss1.item()), loss2=np.float32(loss2.item()), seed=args.seed)
    rng = np.random.default_rng(11)
    for name, p in model.named_parameters():
        key = name.replace(".", "__")
        if p.grad is None:
            continue
        gnp = p.grad.numpy()
        if gnp.shape[0] > 1024:       # the two embedding tables
            rows = np.sort(rng.choice(gnp.shape[0], 512, replace=False))
            nz = np.flatnonzero(np.abs(gnp).sum(1) > 0)
            rows = np.unique(np.concatenate([rows, nz[:256]]))
            out["gradrows_" + key] = rows
            out["grad_" + key] = gnp[rows].copy()
            out["gradcolsum_" + key] = gnp.astype(np.float64).sum(0)
            out["gradfro_" + key] = np.sqrt((gnp.astype(np.float64) ** 2).sum())
        else:
            out["grad_" + key] = gnp.copy()
    # small parameters by value (the tables are pinned by the seed; their hash is enough)
    for name, p in model.state_dict().items():
        if p.numel() <= 64 * 256:
            out["state_" + name.replace(".", "__")] = p.detach().numpy().copy()
    out["user_w_sha"] = sha(model.embedding_user.weight.detach().numpy())
    out["item_w_sha"] = sha(model.embedding_item.weight.detach().numpy())
    model.eval()
    with torch.no_grad():
        scores, negs = model(None, None, None, np.arange(64), test2, flag=2)
        out["trust_scores_at_negs"] = torch.gather(scores, 1, negs).numpy()
        out["trust_test5"] = np.asarray(trust_test5(model, test2), np.float64)
    np.savez_compressed(os.path.join(GOLD, "trust_epinion2.npz"), **out)
    print("trust epinion2: loss1 %.6f loss2 %.6f test5 %s" % (loss1.item(), loss2.item(), out["trust_test5"]))


def stage_epochs_dual_epinion2(n_steps=600, full_epoch=False, fixed_weights=False, n_layers=None):
    """G13 at Epinion2 scale: main_auto_expert_s.py:22-91 executed with the reference's modules on the Epinion2 graph and
    the reference-minted trust paths, for the first `n_steps` batches of epoch 0 (a full epoch is 4 906 batches; the
    dual-task step costs ~1.5 s of reference CPU time), then Test() (:98-114: rec_test over all 3 185 test users +
    trust_test5).  Stored: per-step path counts, both losses per step for the first 16 steps, running loss sums every
    100 steps, the task weights, both tasks' metrics, sampled rows of the trained user table, `w`.
    n_layers: the reference's --layer (lg_parser.py:10; default 3) — a run at another depth goes to dual_epinion2_L{n}_epochs.npz
    (no checkpoint file)."""
    import random
    from collections import defaultdict
    import numpy as np
    import torch
    from torch.utils.data import DataLoader
    _dual_setup()
    if n_layers is not None:
        assert not full_epoch and not fixed_weights
        sys.argv += ["--layer", str(n_layers)]
    import lg_parser
    import utility1.dataloader as ref_dl
    import utility1.utils as ref_utils
    import utility1.model_expert_s as ref_ex
    from utility1.batch_test import rec_test
    from utility2.utils import Data
    from utility2.batch_test_gnn import trust_test5
    args = lg_parser.parse_args_r()
    assert n_layers is None or int(args.layer) == n_layers
    raw_train, raw_test = _epinion2_trust_raw()
    ref_utils.set_seed(args.seed)                                                   # main_auto_expert_s.py:22
    device = torch.device("cpu")
    dataset = ref_dl.Loader(args)                                                   # :34
    train_dataset = ref_dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
    train_loader = DataLoader(train_dataset, batch_size=256, shuffle=True)          # :36
    user_path_indx = defaultdict(list)                                              # :42-46
    path = raw_train[0]
    for i, p in zip(range(len(path)), path):
        user_path_indx[p[0]].append(i)
    train_data2 = Data(raw_train, dataset.n_users, shuffle=False)                   # :47-48
    test_data2 = Data(raw_test, dataset.n_users, shuffle=False, test=True)
    trust_batch_size = len(path) // len(train_loader)                               # :49
    # fixed_weights: the same run as main_11.py drives it (:56-69): at most trust_batch_size paths per batch (not 3 x) and
    # loss = loss1 + loss2 — task_weights get no gradient and stay where they are
    cap_paths = trust_batch_size if fixed_weights else trust_batch_size * 3
    out_name = "dual11_epinion2_epochs.npz" if fixed_weights else "dual_epinion2_epochs.npz"
    if n_layers is not None:
        out_name = f"dual_epinion2_L{n_layers}_epochs.npz"
    Recmodel = ref_ex.LightGCN(args, dataset).to(device)                            # :51-52
    optimizer = torch.optim.Adam(Recmodel.parameters(), lr=args.lr)
    out = dict(n_paths=[], loss1_first=[], loss2_first=[], loss1_cum=[], loss2_cum=[])
    train_loader.dataset.ng_sample()                                                # :56
    Recmodel.train()
    t1 = t2 = full1 = full2 = 0.0
    full_cum1, full_cum2 = [], []
    import time
    t0 = time.time()
    ckpt = {}

    def table_grad_summary(prefix, gnp, rng):
        rows = np.sort(rng.choice(gnp.shape[0], 512, replace=False))
        nz = np.flatnonzero(np.abs(gnp).sum(1) > 0)
        rows = np.unique(np.concatenate([rows, nz[:256]]))
        ckpt[prefix + "_rows"] = rows
        ckpt[prefix] = gnp[rows].copy()
        ckpt[prefix + "_colsum"] = gnp.astype(np.float64).sum(0)
        ckpt[prefix + "_fro"] = np.sqrt((gnp.astype(np.float64) ** 2).sum())

    def take_checkpoint(tag, batch, path_index, l1, l2):
        """Teacher forcing: the reference's FULL parameter state in front of a step, the step's rec batch and path indices,
        both losses and every gradient of the weighted loss (called between loss.backward() and optimizer.step())."""
        rng = np.random.default_rng(31)
        ckpt[f"{tag}_batch"] = batch
        ckpt[f"{tag}_path_index"] = np.asarray(path_index, np.int64)
        ckpt[f"{tag}_loss1"] = np.float64(l1)
        ckpt[f"{tag}_loss2"] = np.float64(l2)
        for name, p in Recmodel.named_parameters():
            key = name.replace(".", "__")
            ckpt[f"{tag}_state_{key}"] = p.detach().numpy().copy()
            if p.grad is None:
                continue
            g = p.grad.numpy()
            if g.shape[0] > 1024:
                table_grad_summary(f"{tag}_grad_{key}", g, rng)
            else:
                ckpt[f"{tag}_grad_{key}"] = g.copy()

    def evaluate():
        Recmodel.eval()
        with torch.no_grad():                                                       # :98-114
            r = rec_test(Recmodel, dataset.testRatings, dataset.testNegatives)
            t = np.asarray(trust_test5(Recmodel, test_data2), np.float64)
        Recmodel.train()
        return r, t

    def write_600(tw, ret, trust):
        uw = Recmodel.embedding_user.weight.detach().numpy()
        iw = Recmodel.embedding_item.weight.detach().numpy()
        rows_u = np.sort(np.random.default_rng(1).choice(uw.shape[0], 256, replace=False))
        rows_i = np.sort(np.random.default_rng(2).choice(iw.shape[0], 256, replace=False))
        np.savez_compressed(os.path.join(GOLD, out_name), seed=args.seed, n_steps=n_steps,
                            trust_batch_size=trust_batch_size, steps_per_epoch=len(train_loader),
                            **{k: np.asarray(v, np.float64) for k, v in out.items()}, task_weights=tw,
                            rec_recall=ret["recall"], rec_ndcg=ret["ndcg"], trust=trust,
                            rows_u=rows_u, rows_i=rows_i, user_w=uw[rows_u], item_w=iw[rows_i],
                            user_w_colsum=uw.astype(np.float64).sum(0), item_w_colsum=iw.astype(np.float64).sum(0),
                            w=Recmodel.w.detach().numpy(), att_exp1=Recmodel.att_exp1.detach().numpy(),
                            att_t=Recmodel.att_t.detach().numpy())

    first = None
    for step, data in enumerate(train_loader):                                      # :60-87
        if step == n_steps:
            # (evaluation touches no RNG: the run goes on unchanged behind it when the full epoch is minted)
            tw600 = Recmodel.task_weights.detach().numpy().copy()
            ret600, trust600 = evaluate()
            write_600(tw600, ret600, trust600)
            print("dual epinion2 @%d: loss1" % n_steps, t1, "loss2", t2, "tw", tw600, "rec", ret600, "trust", trust600, flush=True)
        optimizer.zero_grad()
        user, item, label = data
        unique_user = set(user.numpy().tolist())
        path_index = []
        for u in unique_user:
            path_index.extend(user_path_indx[u])
        if len(path_index) > cap_paths:
            path_index = random.sample(path_index, cap_paths)
        out["n_paths"].append(len(path_index))
        loss1, loss2 = Recmodel(users=user.to(device), items=item.to(device), labels=label.to(device),
                                slice_indices=np.array(list(path_index), dtype=int), trust_data=train_data2, flag=0)
        T, n_rec, T_rec = len(path_index), 5, len(user)
        precision1 = torch.exp(-2 * Recmodel.task_weights[0])
        precision2 = torch.exp(-2 * Recmodel.task_weights[1])
        loss = precision1 * loss1 + precision2 * loss2 + 2 * (n_rec + 1) * T_rec * Recmodel.task_weights[0] \
            + T * Recmodel.task_weights[1]
        if fixed_weights:
            loss = loss1 + loss2                                                      # main_11.py:69
        loss.backward()
        if first is None:
            first = (np.stack([user.numpy(), item.numpy(), label.numpy()]), list(path_index))
        if step == n_steps:
            take_checkpoint("ckpt%d" % step, np.stack([user.numpy(), item.numpy(), label.numpy()]), path_index,
                            loss1.item(), loss2.item())
            if not fixed_weights and n_layers is None:
                np.savez_compressed(os.path.join(GOLD, "dual_epinion2_ckpt.npz"), seed=args.seed, ckpt_step=n_steps,
                                    metrics_rec=np.concatenate([ret600["recall"], ret600["ndcg"]]), metrics_trust=trust600, **ckpt)
            if not full_epoch:
                break
        if step < n_steps:
            t1 += loss1.item()
            t2 += loss2.item()
        full1 += loss1.item()
        full2 += loss2.item()
        if step < 16:
            out["loss1_first"].append(loss1.item()); out["loss2_first"].append(loss2.item())
        optimizer.step()
        if (step + 1) % 100 == 0:
            if step < n_steps:
                out["loss1_cum"].append(t1); out["loss2_cum"].append(t2)
            full_cum1.append(full1); full_cum2.append(full2)
            print("step", step + 1, "loss1", full1, "loss2", full2, "%.0f s" % (time.time() - t0), flush=True)
    if full_epoch:
        # G13 full epoch: running loss sums every 100 steps, the task weights, both tasks' metrics, and the state the epoch ends
        # in (teacher forcing: probed with the epoch's first batch and first path selection, no optimizer step)
        tw = Recmodel.task_weights.detach().numpy().copy()
        ret, trust = evaluate()
        optimizer.zero_grad()
        fb, fp = first
        loss1, loss2 = Recmodel(users=torch.from_numpy(fb[0]), items=torch.from_numpy(fb[1]), labels=torch.from_numpy(fb[2]),
                                slice_indices=np.array(fp, dtype=int), trust_data=train_data2, flag=0)
        T, n_rec, T_rec = len(fp), 5, fb.shape[1]
        loss = torch.exp(-2 * Recmodel.task_weights[0]) * loss1 + torch.exp(-2 * Recmodel.task_weights[1]) * loss2 \
            + 2 * (n_rec + 1) * T_rec * Recmodel.task_weights[0] + T * Recmodel.task_weights[1]
        loss.backward()
        ckpt.clear()
        take_checkpoint("ckptend", fb, fp, loss1.item(), loss2.item())
        np.savez_compressed(os.path.join(GOLD, "dual_epinion2_full_epoch.npz"), seed=args.seed, n_steps=step + 1,
                            trust_batch_size=trust_batch_size, n_paths=np.asarray(out["n_paths"], np.int64),
                            loss1_cum=np.asarray(full_cum1), loss2_cum=np.asarray(full_cum2), loss1=full1, loss2=full2,
                            task_weights=tw, rec_recall=ret["recall"], rec_ndcg=ret["ndcg"], trust=trust, **ckpt)
        print("dual epinion2 full epoch: loss1", full1, "loss2", full2, "tw", tw, "rec", ret, "trust", trust)


def stage_dual_dropout_epinion2(n_steps=150, keep_prob=0.3):
    """G13-dropout: main_auto_expert_s.py:22-91 with the reference's recommended `--dropout 1 --keepprob 0.3` (README.md:119-123) on
    Epinion2 with the reference-minted trust paths: the first `n_steps` batches of epoch 0, then Test().  The rec branch's edge mask
    is drawn per step with torch.rand(nnz) from the global CPU generator (model_expert_s.py:75-92,104-109); the trust branch reads
    the raw user table.  Stored: every step's two losses and path count, the first mask's digest, the task weights, both tasks'
    metrics, sampled rows + column sums of the trained tables."""
    import hashlib
    import random
    from collections import defaultdict
    import numpy as np
    import torch
    from torch.utils.data import DataLoader
    _dual_setup()
    sys.argv += ["--dropout", "1", "--keepprob", str(keep_prob)]
    import lg_parser
    import utility1.dataloader as ref_dl
    import utility1.utils as ref_utils
    import utility1.model_expert_s as ref_ex
    from utility1.batch_test import rec_test
    from utility2.utils import Data
    from utility2.batch_test_gnn import trust_test5
    args = lg_parser.parse_args_r()
    assert args.dropout == 1 and abs(args.keepprob - keep_prob) < 1e-12
    raw_train, raw_test = _epinion2_trust_raw()
    ref_utils.set_seed(args.seed)
    device = torch.device("cpu")
    dataset = ref_dl.Loader(args)
    train_dataset = ref_dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
    train_loader = DataLoader(train_dataset, batch_size=256, shuffle=True)
    user_path_indx = defaultdict(list)
    path = raw_train[0]
    for i, p in zip(range(len(path)), path):
        user_path_indx[p[0]].append(i)
    train_data2 = Data(raw_train, dataset.n_users, shuffle=False)
    test_data2 = Data(raw_test, dataset.n_users, shuffle=False, test=True)
    trust_batch_size = len(path) // len(train_loader)
    cap_paths = trust_batch_size * 3
    Recmodel = ref_ex.LightGCN(args, dataset).to(device)
    optimizer = torch.optim.Adam(Recmodel.parameters(), lr=args.lr)
    nnz = int(Recmodel.Graph._nnz())
    # digest of the first mask: what torch.rand(nnz) + keep_prob gives at this point of the stream, then the generator is put back
    state = torch.get_rng_state()
    train_loader.dataset.ng_sample()                                                # :56 (NumPy RNG)
    it = iter(train_loader)                                                         # the shuffle's two seed draws
    first_batch = next(it)
    mask0 = (torch.rand(nnz) + keep_prob).int().bool().numpy()
    torch.set_rng_state(state)
    losses1, losses2, n_paths = [], [], []
    Recmodel.train()
    import time
    t0 = time.time()
    for step, data in enumerate(train_loader):
        if step == n_steps:
            break
        optimizer.zero_grad()
        user, item, label = data
        if step == 0:
            assert torch.equal(user, first_batch[0])
        unique_user = set(user.numpy().tolist())
        path_index = []
        for u in unique_user:
            path_index.extend(user_path_indx[u])
        if len(path_index) > cap_paths:
            path_index = random.sample(path_index, cap_paths)
        n_paths.append(len(path_index))
        loss1, loss2 = Recmodel(users=user.to(device), items=item.to(device), labels=label.to(device),
                                slice_indices=np.array(list(path_index), dtype=int), trust_data=train_data2, flag=0)
        T, n_rec, T_rec = len(path_index), 5, len(user)
        loss = torch.exp(-2 * Recmodel.task_weights[0]) * loss1 + torch.exp(-2 * Recmodel.task_weights[1]) * loss2 \
            + 2 * (n_rec + 1) * T_rec * Recmodel.task_weights[0] + T * Recmodel.task_weights[1]
        loss.backward()
        optimizer.step()
        losses1.append(loss1.item()); losses2.append(loss2.item())
        if (step + 1) % 50 == 0:
            print("step", step + 1, "loss1", sum(losses1), "loss2", sum(losses2), "%.0f s" % (time.time() - t0), flush=True)
    Recmodel.eval()
    with torch.no_grad():
        ret = rec_test(Recmodel, dataset.testRatings, dataset.testNegatives)
        trust = np.asarray(trust_test5(Recmodel, test_data2), np.float64)
    uw = Recmodel.embedding_user.weight.detach().numpy()
    iw = Recmodel.embedding_item.weight.detach().numpy()
    rows_u = np.sort(np.random.default_rng(1).choice(uw.shape[0], 256, replace=False))
    rows_i = np.sort(np.random.default_rng(2).choice(iw.shape[0], 256, replace=False))
    np.savez_compressed(os.path.join(GOLD, "dual_epinion2_dropout.npz"), seed=args.seed, n_steps=n_steps, keepprob=keep_prob, nnz=nnz,
                        trust_batch_size=trust_batch_size, n_paths=np.asarray(n_paths, np.int64),
                        first_batch=np.stack([first_batch[0].numpy(), first_batch[1].numpy(), first_batch[2].numpy()]),
                        mask0_sha=np.frombuffer(hashlib.sha256(np.packbits(mask0).tobytes()).digest(), np.uint8), mask0_kept=int(mask0.sum()),
                        step_loss1=np.asarray(losses1, np.float64), step_loss2=np.asarray(losses2, np.float64),
                        task_weights=Recmodel.task_weights.detach().numpy().copy(), rec_recall=ret["recall"], rec_ndcg=ret["ndcg"],
                        trust=trust, rows_u=rows_u, rows_i=rows_i, user_w=uw[rows_u], item_w=iw[rows_i],
                        user_w_colsum=uw.astype(np.float64).sum(0), item_w_colsum=iw.astype(np.float64).sum(0),
                        w=Recmodel.w.detach().numpy(), att_exp1=Recmodel.att_exp1.detach().numpy())
    print("dual epinion2 dropout @%d: loss1" % n_steps, sum(losses1), "loss2", sum(losses2), "tw", Recmodel.task_weights.detach().numpy(),
          "rec", ret, "trust", trust)


# --------------------------------------------------------------------------- NGCF whole-run goldens (config 4)
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """philox4x32-10 on uint32 arrays (Salmon et al. 2011; the generator libspexhip's dropout masks use).  Returns the
    four output words.  Harness copy of oracle/oracle.py:philox4x32_10 (the harness stays importable on its own)."""
    import numpy as np
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    c0, c1, c2, c3 = (np.asarray(x, np.uint32).astype(np.uint64) for x in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def message_keep_mask(n_rows, d, p_drop, seed, step, layer):
    """The counter-based message-dropout mask of libspexhip's NGCF kernels (include/spex_hip.h, spex_ngcf_layer_*):
    element e = row * d + col keeps iff u_e >= p_drop, u_e = (word[e & 3] of philox4x32-10(counter = (e >> 2, step,
    layer, 0), key = seed) >> 8) * 2^-24."""
    import numpy as np
    n = n_rows * d
    assert n % 4 == 0
    w = philox4x32_10(np.arange(n // 4, dtype=np.uint32), np.uint32(step), np.uint32(layer), np.uint32(0),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = (np.stack(w, 1).reshape(-1) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (u >= np.float32(p_drop)).reshape(n_rows, d)


def write_small_ngcf(rec_dir):
    """300 users x 200 items, ~3 000 train pairs, 1 held-out item + 30 negatives per user (seed 4), in NGCF's file
    format (load_data.py:29-31).  load_train_data only trains whole blocks of 256 users (:177,181), so the 50-user
    `tiny` graph would train on nothing.  Returns the arrays the fixture stores."""
    import numpy as np
    rng = np.random.default_rng(4)
    U, I = 300, 200
    os.makedirs(rec_dir, exist_ok=True)
    train, test_pos, test_neg = [], [], []
    for u in range(U):
        k = int(rng.integers(3, 18))
        its = rng.permutation(I)[:k + 31]
        train.append(np.sort(its[:k]))
        test_pos.append(int(its[k]))
        test_neg.append(its[k + 1:k + 31])
    train[0] = np.unique(np.append(train[0], I - 1))          # the largest item id occurs in the train file
    if test_pos[0] == I - 1 or (I - 1) in test_neg[0]:
        train[0] = train[0][train[0] != I - 1]; train[1] = np.unique(np.append(train[1], I - 1))
    with open(os.path.join(rec_dir, "train.txt"), "w") as f:
        for u in range(U):
            f.write(str(u) + "".join(f" {int(i)}" for i in train[u]) + "\n")
    with open(os.path.join(rec_dir, "test.txt"), "w") as f:
        for u in range(U):
            f.write(f"{u} {test_pos[u]}\n")
    with open(os.path.join(rec_dir, "negative.txt"), "w") as f:
        for u in range(U):
            f.write(str(u) + "".join(f" {int(i)}" for i in test_neg[u]) + "\n")
    pairs = np.asarray([(u, int(i)) for u in range(U) for i in train[u]], np.int64)
    return dict(train_pairs=pairs, test_pos=np.asarray(test_pos, np.int64), test_neg=np.stack(test_neg).astype(np.int64))


def stage_ngcf_epochs(ds, n_epochs, ckpt_steps=(), native_dropout=False, max_steps=None, extra_argv=(), suffix=""):
    """G12-NGCF: the reference's training run — NGCF_SPEX/code/main_rec.py:18-27,116-148 (setup_seed, the module-level
    Data singleton of utility/batch_test.py, Model_Wrapper, Adam, train(): load_train_data per epoch, shuffled
    DataLoader, loss.backward, step; test()) — driven from the reference's modules on CPU.
    Two harness-side substitutions, both so that the run is a function of the seeds:
      * load_data's multiprocessing.Pool -> a serial map (the pool's workers all fork from the same `random` state and
        take chunks in timing order, so the reference's negatives are not reproducible; a serial map draws the same
        per-user samples from one stream);
      * each nn.Dropout(p) of the model -> the same arithmetic (x * (keep / (1 - p))) with the keep mask taken from a
        counter-based generator (message_keep_mask: seed, step, layer) instead of torch's global RNG.
    Also stored: the sampler's first epoch (G6-NGCF), the three adjacency matrices' hashes (G10+), test() after every
    epoch, the learned weights.
    native_dropout: the second substitution is NOT made — the model keeps its nn.Dropout modules and their noise comes from
    torch's global generator (at::dropout: empty_like(x).bernoulli_(1 - p), one draw of [N, 64] per layer and step) — and every
    step's loss is stored; max_steps stops the run inside epoch 0.  Written to ngcf_<ds>_native_dropout.npz.
    extra_argv: further flags of the reference's parser (e.g. --layer_size [64,64] --mess_dropout [0.1,0.1]: a two-layer model,
    NGCF_SPEX/code/ngcf_parser.py:12); suffix: appended to the output files' stem."""
    import random
    import numpy as np
    import torch
    install_shims()
    torch.set_num_threads(int(os.environ.get("SPEX_MINT_THREADS", "8")))
    code = os.path.join(SCRATCH, "NGCF_SPEX", "code")
    os.makedirs(code, exist_ok=True)
    os.chdir(code)
    data_root = os.path.join(SCRATCH, "NGCF_SPEX", "data") + "/"
    small = write_small_ngcf(os.path.join(data_root, "small", "rec")) if ds == "small" else {}
    for c in ("s_adj_mat.npz", "s_norm_adj_mat.npz", "s_mean_adj_mat.npz"):
        pth = os.path.join(data_root, ds, "rec", c)
        if os.path.exists(pth):
            os.remove(pth)
    sys.path.insert(0, os.path.join(REF, "NGCF_SPEX", "code"))
    sys.argv = ["main_rec.py", "--data_path", data_root, "--dataset", ds] + list(extra_argv)
    import utility.load_data as ref_ld

    class SerialPool:
        def __init__(self, *a, **k):
            pass

        def map(self, fn, it):
            return [fn(x) for x in it]

        def close(self):
            pass
    real_pool = ref_ld.multiprocessing.Pool
    import utility.batch_test as ref_bt                        # builds data_generator from sys.argv (batch_test.py:9-16)
    data_generator, margs = ref_bt.data_generator, ref_bt.args
    src = open(os.path.join(REF, "NGCF_SPEX", "code", "main_rec.py")).read()
    import torch.nn as nn
    import torch.nn.functional as F
    ns = {"nn": nn, "torch": torch, "F": F, "np": np, "args": margs, "trans_to_cuda": lambda v: v}
    exec(compile(src[src.index("class Model_Wrapper"):src.index("def train(model, optimizer)")], "<ref Model_Wrapper>", "exec"), ns)
    Model_Wrapper = ns["Model_Wrapper"]

    def setup_seed(seed):                                       # main_rec.py:18-27
        torch.manual_seed(seed)
        random.seed(seed)
        np.random.seed(seed)
    setup_seed(2020)
    out = {"n_users": data_generator.n_users, "n_items": data_generator.n_items, "n_train": data_generator.n_train}
    plain, norm, mean = data_generator.get_adj_mat()            # main_rec.py:160
    for name, m in (("plain", plain), ("norm", norm), ("mean", mean)):
        m = m.tocsr(); m.sort_indices()
        out[f"{name}_sha"] = np.asarray([sha(m.indptr.astype(np.int32)), sha(m.indices.astype(np.int32)),
                                         sha(m.data.astype(np.float32))])
    device = torch.device("cpu")
    model = Model_Wrapper(data_config={"n_users": data_generator.n_users, "n_items": data_generator.n_items,
                                       "norm_adj": norm}, device=device).to(device)
    out.update({"init_" + k.replace(".", "__"): v.detach().numpy().copy() for k, v in model.state_dict().items()
                if v.numel() <= 64 * 64})
    out["init_user_sha"] = sha(model.user_embedding.weight.detach().numpy())
    out["init_item_sha"] = sha(model.item_embedding.weight.detach().numpy())
    optimizer = torch.optim.Adam(model.parameters(), lr=margs.lr)
    N = data_generator.n_users + data_generator.n_items
    p_drop = eval(margs.mess_dropout)
    DROP_SEED = 2020
    state = {"step": 0}

    class InjectedDropout(nn.Module):
        def __init__(self, p, layer):
            super().__init__()
            self.p, self.layer = p, layer

        def forward(self, x):
            if not self.training or self.p == 0.0:
                return x
            keep = message_keep_mask(x.shape[0], x.shape[1], self.p, DROP_SEED, state["step"], self.layer)
            noise = torch.from_numpy(keep.astype(np.float32))
            noise.div_(1 - self.p)                               # what at::dropout does to its Bernoulli noise
            return x * noise
    for i in range(len(model.dropout_list)):
        if not native_dropout:
            model.dropout_list[i] = InjectedDropout(p_drop[i], i)

    losses, recalls, ndcgs, first_batch, step_losses = [], [], [], None, []
    import time
    t0 = time.time()
    # evaluation in mid-run (does not touch any RNG): at the seeded initial weights — pins the evaluation path alone — and
    # after 500 / 1 500 steps, to show how fast two fp32 runs with different summation orders drift apart
    eval_at = {0: None, 500: None, 1500: None}

    def maybe_eval():
        if state["step"] in eval_at and eval_at[state["step"]] is None:
            model.eval()
            r = ref_bt.test(model, list(data_generator.test_set.keys()), drop_flag=True)
            eval_at[state["step"]] = np.concatenate([r["recall"], r["ndcg"]])
    maybe_eval()
    ckpt = {}

    def table_grad_summary(prefix, gnp, rng):
        rows = np.sort(rng.choice(gnp.shape[0], 512, replace=False))
        nz = np.flatnonzero(np.abs(gnp).sum(1) > 0)
        rows = np.unique(np.concatenate([rows, nz[:256]]))
        ckpt[prefix + "_rows"] = rows
        ckpt[prefix] = gnp[rows].copy()
        ckpt[prefix + "_colsum"] = gnp.astype(np.float64).sum(0)
        ckpt[prefix + "_fro"] = np.sqrt((gnp.astype(np.float64) ** 2).sum())

    def take_checkpoint(tag, batch, loss_value):
        """Teacher forcing: the reference's FULL parameter state in front of a step, that step's batch and dropout step,
        its loss and its gradients (model.parameters() hold them: called between loss.backward() and optimizer.step())."""
        rng = np.random.default_rng(31)
        ckpt[f"{tag}_drop_step"] = state["step"]
        ckpt[f"{tag}_batch"] = batch
        ckpt[f"{tag}_loss"] = np.float64(loss_value)
        for name, p in model.named_parameters():
            key = name.replace(".", "__")
            ckpt[f"{tag}_state_{key}"] = p.detach().numpy().copy()
            g = p.grad.numpy()
            if g.shape[0] > 1024:
                table_grad_summary(f"{tag}_grad_{key}", g, rng)
            else:
                ckpt[f"{tag}_grad_{key}"] = g.copy()
    for epoch in range(n_epochs):
        ref_ld.multiprocessing.Pool = SerialPool
        data_loader = data_generator.load_train_data()           # main_rec.py:121
        ref_ld.multiprocessing.Pool = real_pool
        if epoch == 0:
            u, v, r = data_loader.dataset.tensors
            out.update(sample_sha=np.asarray([sha(u.numpy()), sha(v.numpy()), sha(r.numpy())]), sample_len=len(u),
                       sample_head=np.stack([u[:4096].numpy(), v[:4096].numpy(), r[:4096].numpy().astype(np.int64)]))
        total_loss = 0.0
        for data in data_loader:                                 # :122-129
            model.train()
            optimizer.zero_grad()
            user, item, labels_list = data
            if first_batch is None:
                first_batch = np.stack([user.numpy(), item.numpy(), labels_list.numpy().astype(np.int64)])
            loss = model(user=user, item=item, labels_list=labels_list, flag=0)
            loss.backward(retain_graph=True)
            if state["step"] in ckpt_steps:
                take_checkpoint("ckpt%d" % state["step"],
                                np.stack([user.numpy(), item.numpy(), labels_list.numpy().astype(np.int64)]), loss.item())
            optimizer.step()
            total_loss += loss.item()
            if state["step"] < 32 or native_dropout:
                step_losses.append(loss.item())
            state["step"] += 1
            if max_steps is not None and state["step"] >= max_steps:
                break
            maybe_eval()
            if state["step"] % 500 == 0:
                print("step", state["step"], "loss sum", total_loss, "%.0f s" % (time.time() - t0), flush=True)
        losses.append(total_loss)
        model.eval()
        ret = ref_bt.test(model, list(data_generator.test_set.keys()), drop_flag=True)   # :134-135
        recalls.append(ret["recall"]); ndcgs.append(ret["ndcg"])
        print("epoch", epoch, "loss", total_loss, ret, flush=True)
    if ckpt_steps:
        # the state the run ends in, probed with the run's first batch under the next step's dropout mask (no optimizer step)
        model.train()
        optimizer.zero_grad()
        fb = [torch.from_numpy(first_batch[0]), torch.from_numpy(first_batch[1]), torch.from_numpy(first_batch[2]).float()]
        loss = model(user=fb[0], item=fb[1], labels_list=fb[2], flag=0)
        loss.backward()
        take_checkpoint("ckptend", first_batch, loss.item())
        optimizer.zero_grad()
        model.eval()
        np.savez_compressed(os.path.join(GOLD, f"ngcf_{ds}{suffix}_ckpt.npz"), seed=2020, drop_seed=DROP_SEED, lr=margs.lr,
                            mess_dropout=np.asarray(p_drop), n_steps=state["step"], ckpt_steps=np.asarray(sorted(ckpt_steps)),
                            eval_steps=np.asarray([k for k, v in eval_at.items() if v is not None] + [state["step"]]),
                            eval_metrics=np.asarray([v for v in eval_at.values() if v is not None]
                                                    + [np.concatenate([recalls[-1], ndcgs[-1]])]), **ckpt)
    uw, iw = model.user_embedding.weight.detach().numpy(), model.item_embedding.weight.detach().numpy()
    if ds != "small":
        rows_u = np.sort(np.random.default_rng(1).choice(uw.shape[0], 256, replace=False))
        rows_i = np.sort(np.random.default_rng(2).choice(iw.shape[0], 256, replace=False))
        out.update(rows_u=rows_u, rows_i=rows_i, user_w_colsum=uw.astype(np.float64).sum(0),
                   item_w_colsum=iw.astype(np.float64).sum(0))
        uw, iw = uw[rows_u], iw[rows_i]
    out.update(small)
    out.update({"final_" + k.replace(".", "__"): v.detach().numpy().copy() for k, v in model.state_dict().items()
                if v.numel() <= 64 * 64})
    np.savez_compressed(os.path.join(GOLD, f"ngcf_{ds}{suffix}_native_dropout.npz" if native_dropout else f"ngcf_{ds}{suffix}_epochs.npz"), seed=2020,
                        drop_seed=DROP_SEED, lr=margs.lr,
                        mess_dropout=np.asarray(p_drop), n_steps=state["step"],
                        losses=np.asarray(losses, np.float64), step_losses=np.asarray(step_losses, np.float64),
                        recall=np.asarray(recalls, np.float64), ndcg=np.asarray(ndcgs, np.float64),
                        first_batch=first_batch, user_w=uw, item_w=iw,
                        eval_steps=np.asarray([k for k, v in eval_at.items() if v is not None]),
                        eval_metrics=np.asarray([v for v in eval_at.values() if v is not None]), **out)
    print("ngcf epochs", ds, "losses", losses, "recall", recalls[-1], "ndcg", ndcgs[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="all")
    ap.add_argument("--skip-epinion-test", action="store_true")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    if a.stage == "all":
        env = dict(os.environ, PYTHONHASHSEED="0", PYTHONDONTWRITEBYTECODE="1")
        for st in ("mint", "lightgcn", "ngcf", "trust", "epochs", "epochs-dual"):
            cmd = [sys.executable, os.path.abspath(__file__), "--stage", st]
            if a.skip_epinion_test:
                cmd.append("--skip-epinion-test")
            subprocess.run(cmd, check=True, env=env)
    elif a.stage == "mint":
        stage_mint()
    elif a.stage == "paths":
        stage_paths()
    elif a.stage == "lightgcn":
        stage_lightgcn(a.skip_epinion_test)
    elif a.stage == "ngcf":
        stage_ngcf()
    elif a.stage == "trust":
        stage_trust()
    elif a.stage == "epochs":
        stage_epochs()
    elif a.stage == "epochs-dual":
        stage_epochs_dual()
    elif a.stage == "ngcf-epochs":
        stage_ngcf_epochs("small", 3)
    elif a.stage == "ngcf-epochs-epinion2":  # ~15 min of CPU: one full NGCF epoch (4.7 k steps) + test() through the reference
        stage_ngcf_epochs("epinion2", 1, ckpt_steps=(500, 1500))   # + teacher-forced checkpoints (ngcf_epinion2_ckpt.npz)
    elif a.stage == "ngcf-native-dropout-epinion2":   # ~1 min of CPU: 300 steps of the UNMODIFIED model (torch's own dropout stream) + test()
        # (300 steps: the window in which two fp32 runs of this model stay together — replayed on the GPU the per-step losses agree
        #  to 6e-7 there; minted to 1 500 steps the same replay drifts 8e-5 / 2e-3 / 5e-3 per 300-step window, the growth the
        #  reference's own two mints show, ngcf_epinion2_ref_spread.npz)
        stage_ngcf_epochs("epinion2", 1, native_dropout=True, max_steps=300)
    elif a.stage == "ngcf-2layer-epinion2":   # ~1 min of CPU: 60 steps of the TWO-layer model (--layer_size [64,64]) + test(), with
        # teacher-forced checkpoints in front of step 40 and at the end (ngcf_epinion2_2layer_ckpt.npz)
        stage_ngcf_epochs("epinion2", 1, ckpt_steps=(40,), max_steps=60,
                          extra_argv=["--layer_size", "[64,64]", "--mess_dropout", "[0.1,0.1]"], suffix="_2layer")
    elif a.stage == "trust-epinion2":
        stage_trust_epinion2()
    elif a.stage == "epochs-dual-epinion2":  # ~20 min of CPU: 600 dual-task steps + both evaluations through the reference
        stage_epochs_dual_epinion2()
    elif a.stage in ("epochs-dual-L2-epinion2", "epochs-dual-L4-epinion2"):   # ~3 min of CPU each: 100 steps of main_auto_expert_s.py
        stage_epochs_dual_epinion2(n_steps=100, n_layers=int(a.stage[13]))       # --layer 2 / 4 + both evaluations
    elif a.stage == "epochs-dual11-epinion2":      # ~2 min of CPU: 300 steps of the fixed-weights driver (main_11.py) + both evaluations
        stage_epochs_dual_epinion2(n_steps=300, fixed_weights=True)
    elif a.stage == "epochs-dual-epinion2-full":  # ~2.5 h of CPU: the same run continued to the end of epoch 0 (4 906 steps)
        stage_epochs_dual_epinion2(full_epoch=True)
    elif a.stage == "dual-dropout-epinion2":      # ~5 min of CPU: 150 dual-task steps under --dropout 1 --keepprob 0.3 + both evaluations
        stage_dual_dropout_epinion2()
    elif a.stage == "epochs-epinion2":      # ~25 min of CPU: one full Epinion2 epoch + test() through the reference
        stage_epochs("epinion2", 1)
    elif a.stage == "epochs-dropout":       # G12-dropout on tiny: three epochs of main_rec.py --dropout 1 --keepprob 0.3
        stage_epochs("tiny", 3, dropout=0.3)
    elif a.stage == "epochs-dropout-epinion2":   # ~6 min of CPU: the first 300 steps of that run on Epinion2 + test()
        stage_epochs("epinion2", 1, dropout=0.3, max_steps=300)
    elif a.stage in ("epochs-L2-epinion2", "epochs-L4-epinion2"):   # ~2 min of CPU each: the first 120 steps of main_rec.py --layer 2 / 4 + test()
        stage_epochs("epinion2", 1, max_steps=120, n_layers=int(a.stage[8]))


if __name__ == "__main__":
    main()
