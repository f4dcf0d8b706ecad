#!/usr/bin/env python3
"""SpMM launch time against graph size: "Epinion2 x K" for N ~ 2^14 ... 2^24 nodes, d = 64, the
forward-layer form of the launch (Y = A X, running layer sum updated).  Shows the latency-bound -> cache-bound ->
HBM-bound transition.  One JSON line per size on stdout and in gpurun_out/size_sweep.jsonl.

    python tools/size_sweep.py [--lo 14] [--hi 24]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from spex_amd.datasets import scaled_graph           # noqa: E402
from spex_amd.graph import SpexGraph                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lo", type=int, default=14)
    ap.add_argument("--hi", type=int, default=24)
    ap.add_argument("--d", type=int, default=64)
    a = ap.parse_args()
    D = a.d
    dev = torch.device("cuda:0")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    log = open(os.path.join(ROOT, "gpurun_out", "size_sweep.jsonl"), "a")
    for lg in range(a.lo, a.hi + 1):
        t0 = time.perf_counter()
        rp, cc, vv, _ = scaled_graph(lg, device=dev)
        n, nnz = len(rp) - 1, len(cc)
        g = SpexGraph(rp, cc, vv, device=dev)
        del rp, cc, vv
        build_s = time.perf_counter() - t0
        X = torch.rand(n, D, device=dev) - 0.5
        Y, A = torch.empty_like(X), torch.zeros_like(X)
        iters = 200 if lg <= 18 else (50 if lg <= 21 else 10)
        for _ in range(3):
            g.spmm(X, Y=Y, acc_in=A, acc_out=A)
        g.attach_timer(iters)
        for _ in range(iters):
            g.spmm(X, Y=Y, acc_in=A, acc_out=A)
        ms = g.read_timer()
        g.detach_timer()
        ms_avg, ms_min = float(ms.mean()), float(ms.min())
        alg = nnz * (8 + 4 * D) + n * (4 + 4 * D)                  # SURVEY.md 8d gather model
        comp = nnz * 8 + n * (4 + 8 * D)                           # compulsory: every table row read once
        row = {"d": D, "log2_nodes": lg, "replicas": max(1, round((1 << lg) / 15593)), "n_nodes": n, "nnz": nnz,
               "table_MB": n * D * 4 / 1e6, "launch_us_avg": ms_avg * 1e3, "launch_us_min": ms_min * 1e3,
               "edges_per_s": nnz / (ms_avg * 1e-3), "algorithmic_GBs": alg / (ms_avg * 1e-3) / 1e9,
               "compulsory_GBs": comp / (ms_avg * 1e-3) / 1e9, "frac_of_8TBs": alg / (ms_avg * 1e-3) / 8e12,
               "graph_build_s": build_s}
        line = json.dumps(row)
        print(line, flush=True)
        log.write(line + "\n")
        log.flush()
        del g, X, Y, A
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
