"""The C restatement (oracle/spex_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU.

GPU AddressSanitizer is not available on the MI355X pool, so memory-safety checking happens where it can: the oracle is
what every GPU parity test trusts, and it walks CSR arrays, index lists and edge masks with raw pointers.  The sanitized
build (`make -C oracle sanitize`) re-runs the golden checks plus the ragged cases (empty rows, an empty graph, a batch of
one, a single layer) in a child process with libasan preloaded; any report aborts the child (-fno-sanitize-recover)."""
import os
import subprocess
import sys

from conftest import REPO

CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SPEX_REPO"])
from oracle import oracle as O
g = np.load(os.path.join(os.environ["SPEX_REPO"], "tests", "golden", "lightgcn_tiny.npz"))
rowptr, col, val, E0 = g["rowptr"], g["col"], g["val"], g["E0"]
assert O.build().endswith("libspex_oracle_san.so")
for threads in (1, 2):
    out, layers = O.propagate_mean(rowptr, col, val, E0, 3, n_threads=threads, return_layers=True)
    assert np.array_equal(out, g["light_out"]) and np.array_equal(layers[2], g["E3"]), "the sanitized build must reproduce the golden bit for bit"
# ragged shapes: a matrix without entries, one layer, zero layers, an empty (0-row) matrix
n, d = E0.shape
empty_ptr = np.zeros(n + 1, np.int32)
z = O.propagate_mean(empty_ptr, col[:0], val[:0], E0, 2)
assert np.allclose(z, E0 / 3)
O.propagate_mean(rowptr, col, val, E0, 1); O.propagate_mean(rowptr, col, val, E0, 0)
assert O.spmm(np.zeros(1, np.int32), col[:0], val[:0], E0).shape == (0, d)
# scoring + its gradient: a batch of one, repeated indices at the table ends, the golden batch
n_u = int(g["n_user"]) + 1
U, I = np.ascontiguousarray(out[:n_u]), np.ascontiguousarray(out[n_u:])
for u, i in (([0], [0]), ([n_u - 1] * 4, [I.shape[0] - 1] * 4), (g["batch_users"][0], g["batch_items"][0])):
    O.score_bce(U, I, np.asarray(u), np.asarray(i), np.ones(len(u), np.float32), want_grad=True)
gamma, loss, G = O.lightgcn_loss_and_grad(rowptr, col, val, E0, n_u, 3, g["batch_users"][0], g["batch_items"][0], g["batch_labels"][0])
assert abs(float(loss) - float(g["g3_loss"])) < 1e-6
# masked SpMM: the golden mask, an all-dropped and an all-kept one
for keep in (O.dropout_keep_mask(g["g9_rand"], float(g["g9_keep"])), np.zeros(len(col), np.uint8), np.ones(len(col), np.uint8)):
    O.spmm_masked(rowptr, col, val, keep, 0.6, E0)
# Adam on a length that is not a multiple of any vector width; the BPR closed form on a tiny batch
p = np.ones(13, np.float32)
O.adam_step(p, np.full(13, .5, np.float32), np.zeros(13, np.float32), np.zeros(13, np.float32), 1)
u3, ip, im = np.array([0, 1, 0]), np.array([1, 2, 3]), np.array([4, 5, 1])
O.bpr_loss(U, I, U, I, u3, ip, im)
O.bpr_sgd(U, I, U.astype(np.float64), I.astype(np.float64), u3, ip, im, 0.05, 1e-4)
print("sanitized-ok")
"""


def test_oracle_c_code_is_clean_under_asan_and_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.sep in libasan and os.path.exists(libasan), "gcc's libasan.so not found"
    env = dict(os.environ, SPEX_ORACLE_SANITIZE="1", SPEX_REPO=REPO, LD_PRELOAD=libasan, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sanitized-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
