#!/usr/bin/env python3
"""Time the learned-edge-value kernels (SDDMM, row softmax fwd/bwd, set_values) next to the SpMM on Epinion2 and on an
HBM-resident Epinion2 x K graph.  python tools/edge_ops_time.py [--log2 22]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spex_amd.datasets import scaled_graph        # noqa: E402
from spex_amd.graph import SpexGraph              # noqa: E402


def ev(fn, iters):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, nargs="*", default=[14, 22])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for lg in a.log2:
        rp, cc, vv, _ = scaled_graph(lg, device=dev)
        n, nnz = len(rp) - 1, len(cc)
        g = SpexGraph(rp, cc, vv, device=dev)
        X = torch.rand(n, 64, device=dev) - 0.5
        Gm = torch.rand(n, 64, device=dev) - 0.5
        v = torch.randn(nnz, device=dev)
        out = torch.empty_like(v)
        it = 200 if lg <= 16 else 10
        r = {"log2_nodes": lg, "n": n, "nnz": nnz,
             "spmm_us": ev(lambda: g.spmm(X), it),
             "sddmm_us": ev(lambda: g.sddmm(Gm, X, out=out), it),
             "edge_softmax_us": ev(lambda: g.edge_softmax(v, out=out), it),
             "edge_softmax_bwd_us": ev(lambda: g.edge_softmax_bwd(out, v, grad_in=out), it),
             "set_values_us": ev(lambda: g.set_values(v), it)}
        r["sddmm_alg_GBs"] = (nnz * (256 + 12) + n * 256) / r["sddmm_us"] / 1e3
        r["spmm_alg_GBs"] = (nnz * 264 + n * 260) / r["spmm_us"] / 1e3
        r["softmax_alg_GBs"] = (nnz * 8 + n * 4) / r["edge_softmax_us"] / 1e3
        print(json.dumps(r), flush=True)
        del g, X, Gm, v, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
