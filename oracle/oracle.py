"""CPU oracle for the SPEX LightGCN / NGCF hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the checker
(or the timed CPU baseline).  Nothing under spex_amd/ imports it.

It restates, in NumPy (small / elementwise parts) and plain C (oracle/spex_oracle.c, the sparse product), what the
reference computes on this path.  Every function cites the reference lines it follows (paths relative to
/root/reference).  Parity pin: tests/test_oracle_golden.py checks each function against tests/golden/*.npz, which
oracle/gen_golden.py minted by running the reference itself (torch 2.10.0 CPU) in the build container.
Unpinned (no reference counterpart exists): bpr_* — closed form only, see SURVEY.md 0.3 / 8c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Build (if stale) and return the oracle library.  SPEX_ORACLE_SANITIZE=1 selects the AddressSanitizer +
    UndefinedBehaviorSanitizer build (`make sanitize`; run under LD_PRELOAD=libasan — tests/test_oracle_sanitized.py)."""
    target = "libspex_oracle_san.so" if os.environ.get("SPEX_ORACLE_SANITIZE") == "1" else "libspex_oracle.so"
    so = os.path.join(_HERE, target)
    src = os.path.join(_HERE, "spex_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", target], check=True, stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ----------------------------------------------------------------------------------------------- adjacency (a1, a10)
def _bipartite_csr(u, i, n_u, n_i, self_loops=False):
    """Structure of [[0,R],[R^T,0]] (+I) as sorted CSR; duplicates in (u,i) collapse (dok assignment semantics,
    LightGCN dataloader.py:98-100,110; NGCF load_data.py:86-88)."""
    pairs = np.unique(np.stack([np.asarray(u, np.int64), np.asarray(i, np.int64)], 1), axis=0)
    uu, ii = pairs[:, 0], pairs[:, 1]
    n = n_u + n_i
    rows = np.concatenate([uu, ii + n_u])
    cols = np.concatenate([ii + n_u, uu])
    if self_loops:
        rows = np.concatenate([rows, np.arange(n)])
        cols = np.concatenate([cols, np.arange(n)])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), rows.astype(np.int32), cols.astype(np.int32)


def build_norm_adj_lightgcn(train_u, train_i, n_user, m_item):
    """A_hat = D^-1/2 [[0,R],[R^T,0]] D^-1/2 over n_user+1 (pad row) + m_item nodes, fp32 throughout, zero-degree
    rows -> 0, no self loops.  LightGCN_SPEX/code/utility1/dataloader.py:197-212 then 179-185,221-222 (coalesce:
    row-major, ascending column).  Each stored value is one fp32 product chain (d_r * 1) * d_c: scipy's
    diag.dot(adj).dot(diag) never sums two terms into one entry."""
    rowptr, rows, cols = _bipartite_csr(train_u, train_i, n_user + 1, m_item)
    deg = np.diff(rowptr).astype(np.float32)                      # rowsum of a 0/1 fp32 matrix (:205)
    with np.errstate(divide="ignore"):
        d_inv = np.power(deg, -0.5).astype(np.float32)            # :206
    d_inv[np.isinf(d_inv)] = 0.0                                  # :207
    val = (d_inv[rows] * np.float32(1.0)) * d_inv[cols]           # :210-211
    return rowptr, cols, val.astype(np.float32)


def build_norm_adj_ngcf(train_u, train_i, n_users, n_items):
    """norm_adj = D^-1 (A + I), NGCF_SPEX/code/utility/load_data.py:122-166 (normalized_adj_single of adj + sp.eye,
    :162).  sp.eye is float64, so the row sums, their reciprocal and the product are float64; the fp32 cast happens
    when the model converts the matrix (NGCF main_rec.py:104).  Not symmetric."""
    rowptr, rows, cols = _bipartite_csr(train_u, train_i, n_users, n_items, self_loops=True)
    deg = np.diff(rowptr).astype(np.float64)
    with np.errstate(divide="ignore"):
        d_inv = np.power(deg, -1.0)
    d_inv[np.isinf(d_inv)] = 0.0
    val = (d_inv[rows] * 1.0).astype(np.float32)
    return rowptr, cols, val


def csr_transpose(rowptr, col, val, n_cols):
    """CSR of A^T with ascending-column rows (what autograd's A^T g iterates over for sparse.mm backward)."""
    n_rows = len(rowptr) - 1
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr))
    order = np.lexsort((rows, col))
    t_rowptr = np.zeros(n_cols + 1, np.int64)
    np.add.at(t_rowptr, np.asarray(col, np.int64) + 1, 1)
    return np.cumsum(t_rowptr).astype(np.int32), rows[order].astype(np.int32), np.asarray(val)[order]


# ----------------------------------------------------------------------------------------------- propagation (a2, a3)
def spmm(rowptr, col, val, X, n_threads=1):
    """torch.sparse.mm(Graph, X): LightGCN model.py:91, NGCF main_rec.py:76."""
    rowptr, col, val, X = _c(rowptr, np.int32), _c(col, np.int32), _c(val, np.float32), _c(X, np.float32)
    n, d = len(rowptr) - 1, X.shape[1]
    Y = np.empty((n, d), np.float32)
    lib().spex_oracle_spmm_csr_f32(_p(rowptr, _i32p), _p(col, _i32p), _p(val, _f32p), ctypes.c_int32(n),
                                   _p(X, _f32p), _p(Y, _f32p), ctypes.c_int32(d), ctypes.c_int32(n_threads))
    return Y


def spmm_masked(rowptr, col, val, keep, keep_prob, X):
    """__dropout_x (model.py:46-55) + sparse.mm (model.py:91) with an injected keep mask."""
    rowptr, col, val, X = _c(rowptr, np.int32), _c(col, np.int32), _c(val, np.float32), _c(X, np.float32)
    keep = _c(keep, np.uint8).ravel()
    assert keep.size == col.size, f"keep mask of {keep.size} entries for {col.size} stored entries"
    n, d = len(rowptr) - 1, X.shape[1]
    Y = np.empty((n, d), np.float32)
    lib().spex_oracle_spmm_csr_masked_f32(_p(rowptr, _i32p), _p(col, _i32p), _p(val, _f32p), _p(keep, _u8p),
                                          ctypes.c_float(keep_prob), ctypes.c_int32(n), _p(X, _f32p), _p(Y, _f32p),
                                          ctypes.c_int32(d))
    return Y


def dropout_keep_mask(rand_u01, keep_prob):
    """model.py:50-51: (rand + keep_prob).int().bool()."""
    return (np.asarray(rand_u01, np.float32) + np.float32(keep_prob)).astype(np.int32).astype(bool)


def propagate_mean(rowptr, col, val, E0, n_layers, n_threads=1, return_layers=False):
    """LightGCN.computer(), model.py:66-97.  Returns mean(E0..EL) [N,d] (and [E1..EL])."""
    rowptr, col, val, E0 = _c(rowptr, np.int32), _c(col, np.int32), _c(val, np.float32), _c(E0, np.float32)
    n, d = E0.shape
    out = np.empty((n, d), np.float32)
    layers = np.empty((n_layers, n, d), np.float32) if return_layers else None
    lib().spex_oracle_propagate_mean_f32(
        _p(rowptr, _i32p), _p(col, _i32p), _p(val, _f32p), ctypes.c_int32(n), _p(E0, _f32p),
        ctypes.c_int32(n_layers), ctypes.c_int32(d), _p(layers, _f32p) if return_layers else None, _p(out, _f32p),
        ctypes.c_int32(n_threads))
    return (out, list(layers)) if return_layers else out


def propagate_mean_masked(rowptr, col, val, keep, keep_prob, E0, n_layers):
    cur = _c(E0, np.float32)
    acc = cur.copy()
    for _ in range(n_layers):
        cur = spmm_masked(rowptr, col, val, keep, keep_prob, cur)
        acc = acc + cur
    return acc / np.float32(n_layers + 1)


# ----------------------------------------------------------------------------------------------- scoring (a4)
def score_bce(users_tab, items_tab, u_idx, i_idx, labels=None, want_grad=False):
    """LightGCN.forward, model.py:111-121 (flag=1 -> gamma only; flag=0 -> mean BCE-with-logits)."""
    users_tab, items_tab = _c(users_tab, np.float32), _c(items_tab, np.float32)
    u_idx, i_idx = _c(u_idx, np.int64), _c(i_idx, np.int64)
    B, d = len(u_idx), users_tab.shape[1]
    gamma = np.empty(B, np.float32)
    loss = ctypes.c_float(0)
    lab = _c(labels, np.float32) if labels is not None else None
    gu = np.empty_like(users_tab) if want_grad else None
    gi = np.empty_like(items_tab) if want_grad else None
    lib().spex_oracle_score_bce_f32(
        _p(users_tab, _f32p), _p(items_tab, _f32p), _p(u_idx, _i64p), _p(i_idx, _i64p),
        _p(lab, _f32p) if lab is not None else None, ctypes.c_int32(B), ctypes.c_int32(d), _p(gamma, _f32p),
        ctypes.byref(loss), _p(gu, _f32p) if want_grad else None, ctypes.c_int32(users_tab.shape[0]),
        _p(gi, _f32p) if want_grad else None, ctypes.c_int32(items_tab.shape[0]))
    if labels is None:
        return gamma
    if want_grad:
        return gamma, np.float32(loss.value), gu, gi
    return gamma, np.float32(loss.value)


def lightgcn_loss_and_grad(rowptr, col, val, E0, n_rows_user, n_layers, u_idx, i_idx, labels, n_threads=1, t_csr=None):
    """One training forward+backward (main_rec.py:34-35) restated: loss and d loss / d E0 (dense [N,d]).
    Backward of mean-of-layers: G_L = g/(L+1); G_l = g/(L+1) + A^T G_{l+1}  (autograd of model.py:83-95).
    t_csr: the transposed CSR if the caller already has it (a timing loop should not re-sort the matrix every step)."""
    out = propagate_mean(rowptr, col, val, E0, n_layers, n_threads)
    gamma, loss, gu, gi = score_bce(out[:n_rows_user], out[n_rows_user:], u_idx, i_idx, labels, want_grad=True)
    g = np.concatenate([gu, gi]) / np.float32(n_layers + 1)
    t_rowptr, t_col, t_val = t_csr if t_csr is not None else csr_transpose(rowptr, col, val, E0.shape[0])
    G = g.copy()
    for _ in range(n_layers):
        G = g + spmm(t_rowptr, t_col, t_val, G, n_threads)
    return gamma, loss, G


def adam_step(p, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults as used at main_rec.py:23,37.  In place on p, m, v (fp32, contiguous)."""
    for a in (p, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    g = _c(g, np.float32)
    lib().spex_oracle_adam_step_f32(_p(p, _f32p), _p(g, _f32p), _p(m, _f32p), _p(v, _f32p), ctypes.c_int64(p.size),
                                    ctypes.c_int32(t), ctypes.c_float(lr), ctypes.c_float(beta1),
                                    ctypes.c_float(beta2), ctypes.c_float(eps))


def bpr_sgd(U_read, I_read, U_w, I_w, u, ip, in_, lr, reg):
    """North-star extension, PARITY UNPINNED (no reference counterpart): fp64 closed form of the fused BPR-SGD
    step.  Returns (loss, U_new, I_new) in float64."""
    U_read, I_read = _c(U_read, np.float32), _c(I_read, np.float32)
    Uw, Iw = np.array(U_w, np.float64), np.array(I_w, np.float64)
    u, ip, in_ = _c(u, np.int64), _c(ip, np.int64), _c(in_, np.int64)
    loss = ctypes.c_double(0)
    lib().spex_oracle_bpr_sgd_f64(_p(U_read, _f32p), _p(I_read, _f32p), _p(Uw, _f64p), _p(Iw, _f64p), _p(u, _i64p),
                                  _p(ip, _i64p), _p(in_, _i64p), ctypes.c_int64(len(u)),
                                  ctypes.c_int32(U_read.shape[1]), ctypes.c_double(lr), ctypes.c_double(reg),
                                  ctypes.byref(loss))
    return loss.value, Uw, Iw


def bpr_loss(users_tab, items_tab, U0, I0, u, ip, in_):
    """upstream-LightGCN bpr_loss semantics (extension): (mean softplus(neg-pos), 0.5*(|u0|^2+|p0|^2+|n0|^2)/B)."""
    ue, pe, ne = users_tab[u].astype(np.float64), items_tab[ip].astype(np.float64), items_tab[in_].astype(np.float64)
    x = (ue * ne).sum(1) - (ue * pe).sum(1)
    loss = np.mean(np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x))))
    reg = 0.5 * ((U0[u].astype(np.float64) ** 2).sum() + (I0[ip].astype(np.float64) ** 2).sum()
                 + (I0[in_].astype(np.float64) ** 2).sum()) / len(u)
    return loss, reg


# ----------------------------------------------------------------------------------------------- eval (a8)
KS = (10, 20, 50)


def ranklist(test_items, scores, pos_items, k_max=50):
    """ranklist_by_heapq, utility1/batch_test.py:80-90 (+ the dict build at :35-38): heapq.nlargest over a dict
    is a stable descending sort of the keys in insertion order; a repeated item keeps its first position and its
    last score."""
    rating = {}
    for it, s in zip(test_items, scores):
        rating[int(it)] = float(s)
    keys = list(rating.keys())
    order = sorted(range(len(keys)), key=lambda j: -rating[keys[j]])  # stable
    top = [keys[j] for j in order[:k_max]]
    pos = set(int(p) for p in pos_items)
    return [1 if it in pos else 0 for it in top]


def dcg_at_k(r, k):
    """utility1/metrics.py:43-58 (method 1)."""
    r = np.asarray(r, np.float64)[:k]
    return float(np.sum(r / np.log2(np.arange(2, r.size + 2)))) if r.size else 0.0


def ndcg_at_k(r, k):
    """utility1/metrics.py:61-71: the ideal ordering is taken over the returned top list only."""
    best = dcg_at_k(sorted(r, reverse=True), k)
    return dcg_at_k(r, k) / best if best else 0.0


def recall_at_k(r, k, n_pos):
    """utility1/metrics.py:74-80."""
    return float(np.sum(np.asarray(r, np.float64)[:k]) / n_pos) if n_pos else 0.0


def evaluate(score_fn, test_ratings, test_negatives):
    """test(), utility1/batch_test.py:12-40.  score_fn(u, items[int]) -> scores."""
    res = {"recall": np.zeros(len(KS)), "ndcg": np.zeros(len(KS))}
    users = list(test_ratings.keys())
    for u in users:
        pos = test_ratings[u]
        items = list(test_negatives[u]) + list(pos)                                  # :31
        r = ranklist(items, score_fn(u, items), pos)
        res["recall"] += np.array([recall_at_k(r, k, len(pos)) for k in KS]) / len(users)
        res["ndcg"] += np.array([ndcg_at_k(r, k) for k in KS]) / len(users)
    return res


# ----------------------------------------------------------------------------------------------- sampler (a5)
def ng_sample_replay(features_ps, num_item, train_set, num_ng=5):
    """LightTrainData.ng_sample, utility1/dataloader.py:250-265, replaying NumPy's *global* legacy RNG exactly as
    the reference consumes it (one np.random.randint(num_item) per draw, redraw while (u,j) is a train pair).
    Pure-Python: small cases only."""
    out = []
    for x in features_ps:
        u = x[0]
        for _ in range(num_ng):
            j = np.random.randint(num_item)
            while (u, j) in train_set:
                j = np.random.randint(num_item)
            out.append([u, j])
    return out


# ----------------------------------------------------------------------------------------------- NGCF (a9) + gating (a11)
def leaky_relu(x, slope=0.01):
    return np.where(x >= 0, x, x * np.float32(slope)).astype(np.float32)


def ngcf_forward(rowptr, col, val, user_w, item_w, W_gc, b_gc, W_bi, b_bi):
    """Model_Wrapper.forward, NGCF_SPEX/code/main_rec.py:71-86, one layer, eval mode (dropout off):
    side = A ego; ego' = LReLU(side W_gc^T + b_gc) + LReLU((ego*side) W_bi^T + b_bi); out = [ego | ego'/max(|ego'|,1e-12)].
    user_w carries the trailing pad row that :73 slices off."""
    ego = np.concatenate([user_w[:-1], item_w]).astype(np.float32)
    side = spmm(rowptr, col, val, ego)
    s = leaky_relu(side @ W_gc.T.astype(np.float32) + b_gc)
    b = leaky_relu((ego * side) @ W_bi.T.astype(np.float32) + b_bi)
    e1 = s + b
    nrm = np.sqrt((e1.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
    return np.concatenate([ego, e1 / np.maximum(nrm, np.float32(1e-12))], 1)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """philox4x32-10 on uint32 arrays (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): the
    counter-based generator behind libspexhip's dropout masks.  Returns the four output words."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    c0, c1, c2, c3 = (np.asarray(x, np.uint32).astype(np.uint64) for x in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def message_keep_mask(n_rows, d, p_drop, seed, step, layer):
    """The message-dropout mask of spex_ngcf_layer_fwd_f32 / _bwd_f32 (include/spex_hip.h): element e = row * d + col
    keeps iff u_e >= p_drop, u_e = (word[e & 3] of philox4x32-10(counter = (e >> 2, step, layer, 0), key = seed) >> 8)
    * 2^-24.  It stands where the reference draws nn.Dropout's Bernoulli noise (NGCF_SPEX/code/main_rec.py:81)."""
    n = n_rows * d
    assert n % 4 == 0
    w = philox4x32_10(np.arange(n // 4, dtype=np.uint32), np.uint32(step), np.uint32(layer), np.uint32(0),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = (np.stack(w, 1).reshape(-1) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (u >= np.float32(p_drop)).reshape(n_rows, d)


def ngcf_layer_torch(ego, side, W_gc, b_gc, W_bi, b_bi, keep=None, p_drop=0.0):
    """One NGCF layer on torch tensors (any dtype), NGCF_SPEX/code/main_rec.py:77-83, with the dropout noise injected:
    returns (normalised output, e1 after dropout).  Used with autograd as the checker of the fused backward kernel."""
    import torch
    import torch.nn.functional as F
    e1 = F.leaky_relu(F.linear(side, W_gc, b_gc)) + F.leaky_relu(F.linear(ego * side, W_bi, b_bi))
    if keep is not None:
        noise = keep.to(e1.dtype)
        noise = noise / (1 - p_drop)                 # at::dropout: noise.div_(1 - p), then input * noise
        e1 = e1 * noise
    return F.normalize(e1, p=2, dim=1), e1


def expert_gate(raw, prop, att):
    """model_expert_s.py:156-161: att = softmax([raw | prop] @ att_exp, dim=1); raw*att0 + prop*att1."""
    z = np.concatenate([raw, prop], 1).astype(np.float32) @ att.astype(np.float32)
    z = z - z.max(1, keepdims=True)
    e = np.exp(z)
    a = e / e.sum(1, keepdims=True)
    return (raw * a[:, :1] + prop * a[:, 1:2]).astype(np.float32)


# ------------------------------------------------------------------------------------------------ learned edge values
# Diffnet++.  PARITY UNPINNED: the reference for these is TensorFlow (tf.sparse.softmax,
# tf.sparse.sparse_dense_matmul and their gradients), which is not installed in the image, and the reference holds no
# test or golden vector for them.  Restated from Diffnet++_SPEX/code/utility/Model.py and the ops' definitions; the
# gradients are additionally checked against finite differences in fp64 (tests/test_oracle_golden.py).
def _rows_of(rowptr):
    rowptr = np.asarray(rowptr, np.int64)
    return np.repeat(np.arange(len(rowptr) - 1, dtype=np.int64), np.diff(rowptr))


def edge_softmax(rowptr, v, dtype=np.float32):
    """tf.sparse.softmax over the stored entries of each row (Model.py:275-286): exp(v - max_row) / sum_row."""
    v = np.asarray(v, dtype)
    rows = _rows_of(rowptr)
    n = len(rowptr) - 1
    m = np.full(n, -np.inf, dtype)
    np.maximum.at(m, rows, v)
    e = np.exp(v - m[rows]).astype(dtype)
    s = np.zeros(n, dtype)
    np.add.at(s, rows, e)
    return (e / s[rows]).astype(dtype)


def edge_softmax_bwd(rowptr, y, gy, dtype=np.float32):
    y, gy = np.asarray(y, dtype), np.asarray(gy, dtype)
    rows = _rows_of(rowptr)
    s = np.zeros(len(rowptr) - 1, dtype)
    np.add.at(s, rows, y * gy)
    return (y * (gy - s[rows])).astype(dtype)


def sddmm(rowptr, col, A, B, dtype=np.float32):
    """out[e] = <A[row(e)], B[col(e)]>: d/dval of Y = spmm(val, B) contracted with A = dL/dY."""
    rows = _rows_of(rowptr)
    return np.einsum("ed,ed->e", np.asarray(A, dtype)[rows], np.asarray(B, dtype)[np.asarray(col, np.int64)]).astype(dtype)
