"""(run under rocprofv3 --kernel-trace --stats; PUSH_CAP=none|256|128|64|32 selects the variant: the host loop here is launch-bound)
Is the push-form product bound by atomics piling up on hub rows?  Time spex_spmm_push_batch_f32 on Epinion2 and on Epinion2
with every item's degree capped (hub items lose their excess interactions): same batch, same launch shape."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd.datasets import load_epinion2
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd import ops
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
rng = np.random.default_rng(0)
u = torch.from_numpy(rng.integers(0, 3185, 256)).to(dev); i = torch.from_numpy(rng.integers(0, 12407, 256)).to(dev)
slots = torch.randn(512, 64, device=dev)
caps = {'none': None}
caps.update({str(c): c for c in (256, 128, 64, 32)})
for cap in [caps[os.environ.get('PUSH_CAP', 'none')]]:
    pairs = tr
    if cap is not None:
        order = rng.permutation(len(tr)); t2 = tr[order]
        rank = np.zeros(len(t2), np.int64); cnt = {}
        keep = np.ones(len(t2), bool)
        c = np.zeros(12407, np.int64)
        for k, it in enumerate(t2[:, 1]):
            c[it] += 1
            keep[k] = c[it] <= cap
        pairs = t2[keep]
    g = SpexGraph(*lightgcn_norm_adj(pairs[:, 0], pairs[:, 1], 3185, 12407), device=dev)
    out = torch.zeros(15593, 64, device=dev)
    for _ in range(20): ops.spmm_push_batch(g, u, i, 3186, slots, out, add=slots, scale=0.25)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(200): ops.spmm_push_batch(g, u, i, 3186, slots, out, add=slots, scale=0.25)
    ev1.record(); torch.cuda.synchronize()
    print("item degree cap %s: nnz %d, push %.1f us" % (cap, 2 * len(pairs), ev0.elapsed_time(ev1) / 200 * 1e3))
