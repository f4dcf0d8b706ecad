#!/usr/bin/env python3
"""NGCF training step (BCE, backward, torch Adam; B = 256) on Epinion2 through spex_amd.ngcf.NGCF."""
import os, sys, time, types
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd.datasets import load_epinion2
from spex_amd.graph import ngcf_norm_adj
from spex_amd.ngcf import NGCF
tr = load_epinion2()["train"]
U, I = 3185, 12407
rp, c, v = ngcf_norm_adj(tr[:, 0], tr[:, 1], U, I)
adj = sp.csr_matrix((v, c, rp), shape=(U + I, U + I))
for layers in ("[64]", "[64,64,64]"):
    args = types.SimpleNamespace(embed_size=64, layer_size=layers, mess_dropout="[0.1,0.1,0.1]", regs="[1e-5]")
    m = NGCF({"n_users": U, "n_items": I, "norm_adj": adj}, "cuda", args).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    rng = np.random.default_rng(0)
    u = torch.from_numpy(rng.integers(0, U, 256)); i = torch.from_numpy(rng.integers(0, I, 256))
    y = torch.from_numpy((rng.random(256) < 1 / 6).astype(np.int64))
    def step():
        opt.zero_grad(); loss = m(u, i, y, 0); loss.backward(); opt.step()
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): step()
    torch.cuda.synchronize()
    print("layers %-12s %.2f ms/step" % (layers, (time.perf_counter() - t0) / 100 * 1e3))

