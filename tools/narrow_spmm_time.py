import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spex_amd.datasets import scaled_graph
from spex_amd.graph import SpexGraph
dev = torch.device("cuda:0")
rp, cc, vv, _ = scaled_graph(22, device=dev)
g = SpexGraph(rp, cc, vv, device=dev)
n, nnz = len(rp) - 1, len(cc)
for d in (32, 16, 8):
    X = torch.rand(n, d, device=dev) - 0.5
    Y = torch.empty_like(X)
    for _ in range(3):
        g.spmm(X, Y=Y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        g.spmm(X, Y=Y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    alg = nnz * (8 + 4 * d) + n * (4 + 4 * d)
    print("d=%d: %.2f ms per launch, %.2f TB/s algorithmic (N=%d nnz=%d)" % (d, ms, alg / ms / 1e9, n, nnz))
