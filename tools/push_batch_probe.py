"""Push-form product over many different batches (run under rocprofv3 --kernel-trace): does its duration depend on the batch?
Prints per batch: total entries pushed, the largest slot row, and (from the trace, tools/push_batch_report.py) the duration."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd.datasets import load_epinion2
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd import ops
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
deg = np.diff(csr[0])
g = SpexGraph(*csr, device=dev)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
slots = torch.randn(512, 64, device=dev)
out = torch.zeros(15593, 64, device=dev)
meta = []
for b in range(40):
    u_h, i_h = rng.integers(0, 3185, 256), rng.integers(0, 12407, 256)
    u, i = torch.from_numpy(u_h).to(dev), torch.from_numpy(i_h).to(dev)
    for _ in range(3):
        ops.spmm_push_batch(g, u, i, 3186, slots, out, add=slots, scale=0.25)
    d = np.concatenate([deg[u_h], deg[3186 + i_h]])
    meta.append({"entries": int(d.sum()), "max_row": int(d.max()), "rows_over_256": int((d > 256).sum())})
torch.cuda.synchronize()
json.dump(meta, open(os.environ.get("META", "gpurun_out/push_meta.json"), "w"))
