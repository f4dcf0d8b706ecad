import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd import ops
from spex_amd.datasets import load_epinion2
from spex_amd.graph import SpexGraph, ngcf_norm_adj
tr = load_epinion2()["train"]
ncsr = ngcf_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
dev = torch.device("cuda:0")
gn = SpexGraph(*ncsr, device=dev)
n = 3185 + 12407
ego = torch.rand(n, 64, device=dev) - 0.5
Wg, Wb = (torch.rand(64, 64, device=dev) - 0.5 for _ in range(2))
bg, bb = (torch.rand(64, device=dev) - 0.5 for _ in range(2))
side = torch.empty_like(ego)
def t(fn, it=200):
    for _ in range(10): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3
print("spmm %.1f us   layer epilogue %.1f us" % (t(lambda: gn.spmm(ego, Y=side)), t(lambda: ops.ngcf_layer(ego, side, Wg, bg, Wb, bb))))
