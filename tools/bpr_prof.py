"""One grouped + one atomic BPR step at T = 2^20 on Epinion2-sized tables, for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd import ops

dev = torch.device("cuda:0")
n_u, n_i, T = 3186, 12407, 1 << 20
lo = torch.rand(n_u + n_i, 64, device=dev) - 0.5
E0 = torch.rand(n_u + n_i, 64, device=dev) - 0.5
u, p, n = torch.randint(0, 3185, (T,), device=dev), torch.randint(0, n_i, (T,), device=dev), torch.randint(0, n_i, (T,), device=dev)
for grouped in (True, False):
    for _ in range(6):
        ops.bpr_sgd_step(lo[:n_u], lo[n_u:], E0[:n_u], E0[n_u:], u, p, n, 1e-6, 0.0, grouped=grouped)
torch.cuda.synchronize()
