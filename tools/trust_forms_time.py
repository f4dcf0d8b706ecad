#!/usr/bin/env python3
"""The trust head's training forms against each other (spex_trust_head_train_f32): the FUSED kernel (one 16-wave workgroup per
path: forward chain, logits over the whole user table, backward chain), its SPLIT form (S workgroups per path, each sweeping a
share of the table; the path's last one folds the shares and runs the backward chain) for several (users, paths) sizes
(SPEX_TRUST_SPLIT caps the workgroups per path; every (form, size) runs in its own subprocess).  Prints us per call (forward +
backward + reductions).  (Round 3's table also had the five-launch TILED form, removed in round 4: profiles/r03/trust_forms.txt.)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import ctypes
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from spex_amd import _lib, ops
    if os.environ.get("SPEX_LIB"):
        _lib.LIB_PATH = os.environ["SPEX_LIB"]          # A/B against another build of the library
    from spex_amd.graph import _launch, _ptr
    n_users, T, L, H = int(sys.argv[2]), int(sys.argv[3]), 6, 3
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu"); g.manual_seed(1)
    table = (torch.rand(n_users + 1, 64, generator=g) * 0.2 - 0.1).to(dev)
    params = (torch.rand(ops.trust_param_count(H, 64), generator=g) * 0.2 - 0.1).to(dev)
    rng = np.random.default_rng(0)
    lens = rng.integers(1, L + 1, T)
    seq = np.full((T, L), n_users, np.int64)
    for k in range(T):
        seq[k, :lens[k]] = rng.integers(0, n_users, lens[k])
    seq_d, len_d, tgt = torch.from_numpy(seq).to(dev), torch.from_numpy(lens.astype(np.int64)).to(dev), torch.from_numpy(rng.integers(0, n_users, T)).to(dev)
    n_ws = int(_lib.load().spex_trust_workspace_floats(T, L, 64, H, n_users + 1))
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    a2, ws, ds, lb, loss, gp, gt = z(T, 64), z(n_ws), z(T, n_users), z(T), z(1), z(params.numel()), z(n_users + 1, 64)
    def call():
        _launch(dev, "spex_trust_head_train_f32", _ptr(table), n_users + 1, _ptr(params), _ptr(seq_d), _ptr(len_d), _ptr(tgt), T, L, 64, H, 1,
                1.0, None, _ptr(a2), _ptr(ds), _ptr(lb), _ptr(ws), _ptr(loss), 0, _ptr(gp), _ptr(gt))
    for _ in range(30): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    print("%.2f %.6f" % (e0.elapsed_time(e1) / n * 1e3, loss.item()))
    sys.exit(0)
print("%8s %6s %12s %12s   (us per call; loss agreement)" % ("users", "paths", "fused, S=1", "fused, split"))
for n_users, T in ((3185, 15), (3185, 45), (6812, 15), (6812, 45), (3185, 192), (26000, 15), (26000, 60), (60000, 15), (100000, 15)):
    res = []
    for env in ({"SPEX_TRUST_SPLIT": "1"}, {}):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(n_users), str(T)], capture_output=True, text=True,
                             env=dict(os.environ, **env))
        res.append(out.stdout.strip().split() if out.returncode == 0 else ["nan", out.stderr[-200:]])
    print("%8d %6d %12s %12s   losses %s %s" % (n_users, T, res[0][0], res[1][0], res[0][1], res[1][1]))
