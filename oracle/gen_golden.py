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
    torch.set_num_threads(8)
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
    torch.set_num_threads(8)
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
    torch.set_num_threads(8)
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
def stage_epochs(ds="tiny", n_epochs=3):
    """G12: the reference's own training run — main_rec.py:15-37,50 executed with the reference's modules (set_seed,
    Loader, LightTrainData.ng_sample, DataLoader(256, shuffle=True), model.LightGCN, torch Adam, test()) for three
    epochs on `tiny`: per-epoch loss sums, per-epoch recall / ndcg, the trained tables.  main_rec.py itself runs at
    import and writes logs, so its Train() / Test() bodies are driven from here, line for line."""
    import numpy as np
    import torch
    from torch.utils.data import DataLoader
    install_shims()
    torch.set_num_threads(8)
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
    for epoch in range(n_epochs):
        train_loader.dataset.ng_sample()                                            # :26
        Recmodel.train()
        total_loss = 0.0
        for data in train_loader:                                                   # :30-37
            optimizer.zero_grad()
            user, item, label = data
            if first_batch is None:
                first_batch = np.stack([user.numpy(), item.numpy(), label.numpy()])
            loss = Recmodel(users=user.to(device), items=item.to(device), labels=label.to(device), flag=0)
            loss.backward()
            total_loss += loss.item()
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
    np.savez_compressed(os.path.join(GOLD, f"lightgcn_{ds}_epochs.npz"), seed=args.seed, lr=args.lr,
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
    torch.set_num_threads(8)
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
    elif a.stage == "epochs-epinion2":      # ~25 min of CPU: one full Epinion2 epoch + test() through the reference
        stage_epochs("epinion2", 1)


if __name__ == "__main__":
    main()
