"""The dense NGCF layer backward (spex_ngcf_layer_bwd_f32: multi-layer models' earlier layers) on Epinion2's 15 592-row table:
us per launch with (a) a dense upstream gradient and (b) the gradient pattern behind a 256-sample batch one layer up (g_norm at the
batch's rows, g_next at their neighbours)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spex_amd import ops
from spex_amd.datasets import load_epinion2
from spex_amd.graph import SpexGraph, ngcf_norm_adj
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
csr = ngcf_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
n, d = len(csr[0]) - 1, 64
g = SpexGraph(*csr, device=dev)
torch.manual_seed(0)
ego = torch.rand(n, d, device=dev) - 0.5
side = torch.empty_like(ego); g.spmm(ego, Y=side)
W = [torch.rand(d, d, device=dev) - 0.5, torch.rand(d, device=dev) - 0.5, torch.rand(d, d, device=dev) - 0.5, torch.rand(d, device=dev) - 0.5]
gW = [torch.zeros_like(w) for w in W]
g_all = torch.zeros(n, 3 * d, device=dev)
g_side, g_ego = torch.empty_like(ego), torch.empty_like(ego)


def timed(fn, k=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3


g_next = torch.randn(n, d, device=dev)
g_all.normal_()
print("dense upstream gradient      : %.1f us" % timed(lambda: ops.ngcf_layer_bwd(ego, side, *W, g_all, 0, g_next, g_side, g_ego, *gW, drop=(0.1, 7, 3), pad_row=3185)))
rng = np.random.default_rng(1)
rows = np.concatenate([rng.integers(0, 3185, 256), 3186 + rng.integers(0, 12407, 256)])
g_all.zero_(); g_all[torch.from_numpy(rows).to(dev)] = torch.randn(512, 3 * d, device=dev)
nb = np.unique(np.concatenate([csr[1][csr[0][r]:csr[0][r + 1]] for r in rows]))
g_next.zero_(); g_next[torch.from_numpy(nb).to(dev)] = torch.randn(len(nb), d, device=dev)
tiles = len(np.unique(np.concatenate([rows, nb]) // 16))
print("behind a 256-sample batch     : %.1f us  (%d of %d tiles carry a gradient)" % (
    timed(lambda: ops.ngcf_layer_bwd(ego, side, *W, g_all, 0, g_next, g_side, g_ego, *gW, drop=(0.1, 7, 3), pad_row=3185)), tiles, (n + 15) // 16))
