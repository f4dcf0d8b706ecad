#!/usr/bin/env python3
"""Time one Diffnet++ training step (forward + backward + Adam) on the HIP kernels: Epinion2's user-item interactions
plus a synthetic social graph with the same user count (the trust links of the shipped .mat are not part of the
LightGCN fixture), H = 64, B = 256.  Also the same step with torch's own sparse ops on the GPU for orientation."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spex_amd.datasets import load_epinion2                                     # noqa: E402
from spex_amd.diffnet import DiffnetPlusPlus, LearnedGraph, loss_fn, pairs_to_csr  # noqa: E402


def main():
    tr = load_epinion2()["train"]
    U, I = int(tr[:, 0].max()) + 1, int(tr[:, 1].max()) + 1
    rng = np.random.default_rng(0)
    n_social = 20 * U
    su = rng.integers(0, U, n_social)
    sv = (su + 1 + rng.zipf(1.6, n_social)) % U
    csr = {"social": pairs_to_csr(su, sv, U, U), "consumed": pairs_to_csr(tr[:, 0], tr[:, 1], U, I),
           "customer": pairs_to_csr(tr[:, 1], tr[:, 0], I, U)}
    cols = {"social": U, "consumed": I, "customer": U}
    graphs = {k: LearnedGraph(*csr[k], n_cols=cols[k]) for k in csr}
    torch.manual_seed(0)
    model = DiffnetPlusPlus(U, I, 64, graphs["social"], graphs["consumed"], graphs["customer"]).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    B = 256
    users = torch.from_numpy(rng.integers(0, U, B)).cuda()
    items = torch.from_numpy(rng.integers(0, I, B)).cuda()
    labels = torch.from_numpy((rng.random(B) < 1 / 6).astype(np.float32)).cuda()

    def step():
        opt.zero_grad(set_to_none=True)
        s, l = model(users, items, labels, 0)
        loss = loss_fn(s, l)
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        for _ in range(3):
            model(users, items, None, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(users, items, None, 1)
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t0) / n * 1e3
    # the same step captured once into a HIP graph and replayed (what the GPU itself needs for it)
    graph_ms = None
    try:
        opt2 = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                opt2.zero_grad(set_to_none=True)
                s, l = model(users, items, labels, 0)
                loss_fn(s, l).backward()
                opt2.step()
        torch.cuda.current_stream().wait_stream(side)
        cg = torch.cuda.CUDAGraph()
        opt2.zero_grad(set_to_none=True)
        with torch.cuda.graph(cg):
            s, l = model(users, items, labels, 0)
            static_loss = loss_fn(s, l)
            static_loss.backward()
            opt2.step()
        for _ in range(3):
            cg.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            cg.replay()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / n * 1e3
    except Exception as e:                                     # orientation only
        graph_ms = repr(e)
    nnz = {k: len(v[1]) for k, v in csr.items()}
    print(json.dumps({"users": U, "items": I, "nnz": nnz, "train_step_ms": ms, "forward_ms": fwd_ms, "train_step_hipgraph_replay_ms": graph_ms,
                      "spmm_edges_per_train_step": 3 * 2 * sum(nnz.values()),
                      "edges_per_s": 3 * 2 * sum(nnz.values()) / (ms * 1e-3)}))


if __name__ == "__main__":
    main()
