#!/usr/bin/env python3
"""Phase stamps of the fused trust kernel in its ONE-workgroup-per-path form (trust_path_train_kernel; SPEX_TRUST_SPLIT=1 is set
here — in the split form no single workgroup walks every phase), workgroup 0 / thread 0, wall_clock64 (10 ns).  Needs the
debug library spex_amd/lib/libspexhip_stamps.so (trust.hip compiled with -DSPEX_STAMPS and linked with the other objects):

    cd spex_amd/csrc && hipcc $FLAGS -DSPEX_STAMPS -c trust.hip -o /tmp/trust_stamps.o && \
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libspexhip_stamps.so $(ls *.o | grep -v '^trust.o') /tmp/trust_stamps.o -ldl

usage: trust_stamps.py [n_users] [paths] [path_len of workgroup 0]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SPEX_TRUST_SPLIT"] = "1"
from spex_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "spex_amd", "lib", "libspexhip_stamps.so")
from spex_amd import ops
from spex_amd.graph import _launch, _ptr
n_users = int(sys.argv[1]) if len(sys.argv) > 1 else 3185
T = int(sys.argv[2]) if len(sys.argv) > 2 else 15
l0 = int(sys.argv[3]) if len(sys.argv) > 3 else 6
L, H = 6, 3
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(1)
table = (torch.rand(n_users + 1, 64, generator=g) * 0.2 - 0.1).to(dev)
params = (torch.rand(ops.trust_param_count(H, 64), generator=g) * 0.2 - 0.1).to(dev)
rng = np.random.default_rng(0)
lens = rng.integers(1, L + 1, T)
lens[0] = l0
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
lib = ctypes.CDLL(_lib.LIB_PATH)
names = {0: "start", 1: "F0 rows + one input head per wave", 2: "F1 M @ w on 4 waves", 16: "F2 partials, ELU, output attention", 17: "F3 readout mat-vecs as tasks",
         18: "F4 sigmoids, alphas, p_a", 3: "F4 max-pool, gate, a2", 4: "logits sweep (thread 0)", 5: "block max / sum-exp (waits for the slowest wave)",
         6: "loss + barrier", 7: "d score pass", 8: "d a2 fold + barrier", 9: "P0 gate", 10: "P1 Wt^T on 2 waves", 11: "P2 readout backward",
         12: "P3 W2^T per position / W1^T", 13: "P4 output attention + ELU backward", 14: "P5 one head per wave: w rows -> dM, head backward",
         15: "P6 row gradients"}
order = [0, 1, 2, 16, 17, 18, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]
rows = []
for k in range(40):
    call()
    out = (ctypes.c_ulonglong * 32)()
    assert lib.spex_debug_trust_stamps(out) == 0
    t = np.array([out[i] for i in order], np.float64)
    if k >= 10:
        rows.append(np.diff(t) * 0.01)
rows = np.array(rows)
print("fused trust kernel, %d users, %d paths, workgroup 0: path of %d positions, %d heads" % (n_users, T, lens[0], H))
for i, m, lo, hi in zip(order[1:], rows.mean(0), rows.min(0), rows.max(0)):
    print("%-44s %6.2f us (%.2f-%.2f)" % (names[i], m, lo, hi))
print("total stamped %.2f us" % rows.sum(1).mean())
