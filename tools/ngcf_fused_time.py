"""NGCF layer forward on Epinion2: SpMM + layer as two launches vs the fused launch on a tile-mode handle (us per call)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd import ops
from spex_amd.datasets import load_epinion2
from spex_amd.graph import SpexGraph, ngcf_norm_adj
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
if len(sys.argv) > 1 and sys.argv[1] == "lightgcn":       # a graph whose task table fits one dispatch round (496 workgroups)
    from spex_amd.graph import lightgcn_norm_adj
    rowptr, col, val = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
else:
    rowptr, col, val = ngcf_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
n = len(rowptr) - 1
g_plain, g_tile = SpexGraph(rowptr, col, val, device=dev), SpexGraph(rowptr, col, val, device=dev, tile_rows=True)
rng = np.random.default_rng(3)
ego = torch.from_numpy((rng.normal(size=(n, 64)) * 0.1).astype(np.float32)).to(dev)
W_gc, W_bi = (torch.from_numpy(rng.normal(size=(64, 64)).astype(np.float32) * 0.2).to(dev) for _ in range(2))
b_gc, b_bi = (torch.from_numpy(rng.normal(size=64).astype(np.float32) * 0.1).to(dev) for _ in range(2))
drop = (0.1, 12345, 7)
out, side = torch.zeros(n, 128, device=dev), torch.zeros(n, 64, device=dev)


def two():
    g_plain.spmm(ego, Y=side)
    ops.ngcf_layer_fwd(ego, side, W_gc, b_gc, W_bi, b_bi, out, 0, True, drop=drop, pad_row=3185)


def two_tile():
    g_tile.spmm(ego, Y=side)
    ops.ngcf_layer_fwd(ego, side, W_gc, b_gc, W_bi, b_bi, out, 0, True, drop=drop, pad_row=3185)


def fused():
    ops.ngcf_spmm_layer_fwd(g_tile, ego, W_gc, b_gc, W_bi, b_bi, out, side, drop=drop, pad_row=3185)


for name, fn in (("two launches (ordinary handle)", two), ("two launches (tile-mode handle)", two_tile), ("fused launch", fused)):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(500):
        fn()
    b.record(); torch.cuda.synchronize()
    print("%-34s %.2f us" % (name, a.elapsed_time(b) / 500 * 1e3))
