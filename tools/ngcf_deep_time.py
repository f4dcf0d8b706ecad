"""A two-layer NGCF training step (`--layer_size [64,64]`) on Epinion2, B = 256: the one-call native step
(spex_ngcf_deep_step_bce_f32) against NGCFStepper's launch-by-launch path — us per step by HIP events."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scipy.sparse as sp
from spex_amd.datasets import load_epinion2
from spex_amd.graph import ngcf_norm_adj
from spex_amd.ngcf import NGCF
from spex_amd.trainer import NGCFStepper
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
n_u, n_i = 3185, 12407
csr = ngcf_norm_adj(tr[:, 0], tr[:, 1], n_u, n_i)
rng = np.random.default_rng(2)
ub = torch.from_numpy(rng.integers(0, n_u, 256)).to(dev); ib = torch.from_numpy(rng.integers(0, n_i, 256)).to(dev)
yb = torch.from_numpy((rng.random(256) < 1 / 6).astype(np.float32)).to(dev)


def timed(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for layers in ("[64,64]", "[64,64,64]"):
    L = layers.count("64")
    for native in (True, False):
        torch.manual_seed(0)
        a = argparse.Namespace(embed_size=64, layer_size=layers, mess_dropout=str([0.1] * L), regs="[1e-5]")
        net = NGCF({"n_users": n_u, "n_items": n_i, "norm_adj": sp.csr_matrix((csr[2], csr[1], csr[0]), shape=(n_u + n_i,) * 2)}, dev, a).to(dev)
        net.train()
        st = NGCFStepper(net, lr=1e-3)
        if not native:
            st._one_call_ok = lambda *x: False
        acc = torch.zeros(1, device=dev)
        print("NGCF %d layers, %s: %.1f us per step" % (L, "one native call     " if native else "launch by launch    ", timed(lambda: st.step(ub, ib, yb, loss_acc=acc))), flush=True)
