"""BPR fused step on Epinion2's tables: atomic form vs LDS-bucketed (grouped) form, random and sampler order, per T."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd import ops
from spex_amd.datasets import load_epinion2


def timed(fn, iters=20):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dev = torch.device("cuda:0")
    n_u, n_i = 3186, 12407
    lo = torch.rand(n_u + n_i, 64, device=dev) - 0.5
    E0 = torch.rand(n_u + n_i, 64, device=dev) - 0.5
    tr = load_epinion2()["train"]
    order = np.lexsort((tr[:, 1], tr[:, 0]))
    out = []
    for logT in (14, 16, 18, 20):
        T = 1 << logT
        cases = {"random": (torch.randint(0, 3185, (T,), device=dev), torch.randint(0, n_i, (T,), device=dev))}
        if T <= 5 * len(tr):
            k = T // 5
            cases["sampler"] = (torch.from_numpy(np.repeat(tr[order][:k, 0], 5)).to(dev), torch.from_numpy(np.repeat(tr[order][:k, 1], 5)).to(dev))
        for name, (u, p) in cases.items():
            n = torch.randint(0, n_i, (u.numel(),), device=dev)
            row = {"T": int(u.numel()), "order": name}
            for grouped in (False, True):
                ms = timed(lambda: ops.bpr_sgd_step(lo[:n_u], lo[n_u:], E0[:n_u], E0[n_u:], u, p, n, 1e-6, 0.0, grouped=grouped))
                row["grouped" if grouped else "atomic"] = {"ms": ms, "Gtriples_s": u.numel() / ms / 1e6, "algorithmic_TBs": u.numel() * 1548 / ms / 1e9}
            out.append(row)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
