"""Adjacency construction (host) and the device graph handle.

Host side builds the normalised user-item adjacency as CSR directly from the interaction pairs — one vectorised
sort instead of the reference's dok fill + lil slicing (3 s + 7 s on Epinion2, LightGCN_SPEX/code/utility1/
dataloader.py:98-100,197-212) — with the same fp32 arithmetic, so the values are bit-identical to what the reference
caches in s_pre_adj_mat.npz.  `SpexGraph` then owns the HBM copy through libspexhip's graph handle.
"""
import ctypes

import numpy as np
import torch

from . import _lib


# ------------------------------------------------------------------------------------------------ host CSR builders
def bipartite_csr(users, items, n_u, n_i, self_loops=False, counts=False):
    """CSR structure of [[0, R], [R^T, 0]] (+ I): rows ascending, columns ascending within a row (= coalesced COO
    order, dataloader.py:222).  A repeated (u, i) pair is one stored entry; with counts=True its multiplicity is
    returned as a fourth array (the reference's LightGCN matrix SUMS repeated pairs: UserItemNet is built by
    csr_matrix((ones, (u, i))), dataloader.py:110 — NGCF's R is a dok assignment, load_data.py:92, and does not)."""
    key, mult = np.unique(np.asarray(users, np.int64) * np.int64(n_i) + np.asarray(items, np.int64), return_counts=True)
    uu, ii = key // n_i, key % n_i
    n = n_u + n_i
    rows = np.concatenate([uu, ii + n_u])
    cols = np.concatenate([ii + n_u, uu])
    mult = np.concatenate([mult, mult])
    if self_loops:
        eye = np.arange(n, dtype=np.int64)
        rows, cols, mult = np.concatenate([rows, eye]), np.concatenate([cols, eye]), np.concatenate([mult, np.ones(n, mult.dtype)])
    order = np.argsort(rows * np.int64(n) + cols, kind="stable")
    rows, cols, mult = rows[order], cols[order], mult[order]
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    out = (rowptr.astype(np.int32), rows.astype(np.int32), cols.astype(np.int32))
    return out + (mult,) if counts else out


def lightgcn_norm_adj(users, items, n_user, m_item):
    """A_hat = D^-1/2 [[0,R],[R^T,0]] D^-1/2 over (n_user + 1 pad row) + m_item nodes; fp32; zero-degree rows stay
    empty; no self loops.  Mirrors dataloader.py:110,197-212: a_rc = number of times the pair occurs in the train file,
    degree = row sum of those counts, val = (d_r^-1/2 * a_rc) * d_c^-1/2 rounded after each product."""
    rowptr, rows, cols, mult = bipartite_csr(users, items, n_user + 1, m_item, counts=True)
    a = mult.astype(np.float32)
    deg = np.bincount(rows, weights=mult, minlength=len(rowptr) - 1).astype(np.float32)
    with np.errstate(divide="ignore"):
        d_inv = np.power(deg, -0.5).astype(np.float32)
    d_inv[np.isinf(d_inv)] = 0.0
    val = ((d_inv[rows] * a) * d_inv[cols]).astype(np.float32)
    return rowptr, cols, val


def ngcf_norm_adj(users, items, n_users, n_items):
    """norm_adj = D^-1 (A + I) (NGCF_SPEX/code/utility/load_data.py:135-144,162), computed in float64 as the
    reference does (sp.eye is float64) and cast to fp32 where the model converts it (NGCF main_rec.py:104)."""
    return ngcf_adjacency(users, items, n_users, n_items, "norm")


def ngcf_adjacency(users, items, n_users, n_items, kind="norm"):
    """The three matrices of Data.create_adj_mat (load_data.py:122-166) as CSR over n_users + n_items nodes:
      plain  A = [[0, R], [R^T, 0]], fp32 ones (:124-130,166);
      norm   D^-1 (A + I): row sums and reciprocals in float64 (sp.eye makes the sum float64, :162), stored fp32;
      mean   D^-1 A: everything in float32 (:163), zero-degree rows empty."""
    if kind == "plain":
        rowptr, rows, cols = bipartite_csr(users, items, n_users, n_items)
        return rowptr, cols, np.ones(len(cols), np.float32)
    if kind == "norm":
        rowptr, rows, cols = bipartite_csr(users, items, n_users, n_items, self_loops=True)
        deg = np.diff(rowptr).astype(np.float64)
        with np.errstate(divide="ignore"):
            d_inv = np.power(deg, -1.0)
        d_inv[np.isinf(d_inv)] = 0.0
        return rowptr, cols, d_inv[rows].astype(np.float32)
    if kind == "mean":
        rowptr, rows, cols = bipartite_csr(users, items, n_users, n_items)
        deg = np.diff(rowptr).astype(np.float32)
        with np.errstate(divide="ignore"):
            d_inv = np.power(deg, np.float32(-1.0)).astype(np.float32)
        d_inv[np.isinf(d_inv)] = 0.0
        return rowptr, cols, d_inv[rows]
    raise ValueError("kind must be plain, norm or mean")


def csr_transpose(rowptr, col, val, n_cols):
    """CSR of A^T plus, per transposed entry, the index of the original entry (edge_id) so that an edge keep-mask
    drawn for A addresses the same edges in A^T."""
    n_rows = len(rowptr) - 1
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr))
    order = np.argsort(np.asarray(col, np.int64) * np.int64(n_rows) + rows, kind="stable")
    t_rowptr = np.zeros(n_cols + 1, np.int64)
    np.cumsum(np.bincount(col, minlength=n_cols), out=t_rowptr[1:])
    return (t_rowptr.astype(np.int32), rows[order].astype(np.int32), np.asarray(val, np.float32)[order],
            order.astype(np.int32))


def row_block(rowptr, col, val, r0, r1, edge_id=None):
    """Rows [r0, r1) of a CSR matrix as a CSR block with global column ids (1-D row partition)."""
    b, e = int(rowptr[r0]), int(rowptr[r1])
    eid = None if edge_id is None else np.ascontiguousarray(edge_id[b:e])
    return (np.ascontiguousarray(rowptr[r0:r1 + 1] - rowptr[r0]).astype(np.int32), np.ascontiguousarray(col[b:e]),
            np.ascontiguousarray(val[b:e]), eid)


# ------------------------------------------------------------------------------------------------ device handle
def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _device_of(device):
    """A torch.device with an explicit index (a bare 'cuda' means the current device)."""
    device = torch.device(device if device is not None else "cuda")
    if device.type != "cuda":
        raise ValueError(f"libspexhip runs on the GPU only (got device {device})")
    return device if device.index is not None else torch.device("cuda", torch.cuda.current_device())


def _launch(device, name, *args):
    """One libspexhip call on `device`: its current stream is appended as the last argument, and when `device` is not the
    process's current device the call runs under a device guard — the kernels are queued on the stream of the device
    that owns the pointers (and the handle's on-demand allocations land there), whatever the caller's current device."""
    if device.index is None or device.index == torch.cuda.current_device():
        _lib.call(name, *args, _stream(device))
    else:
        with torch.cuda.device(device):
            _lib.call(name, *args, _stream(device))


def _bump(*tensors):
    """The kernels write through raw pointers, which torch's version counters do not see: tell autograd (and anything
    keyed on `_version`, e.g. the drop-in model's eval-mode propagation cache) that these tensors were modified."""
    ts = tuple(t for t in tensors if t is not None)
    if ts:
        torch._C._increment_version(ts)      # takes an ITERABLE of tensors (a bare tensor would be iterated row by row)


class SpexGraph:
    """A CSR matrix (or a row block of one) resident in HBM.  Stands where the reference keeps its
    torch.sparse.FloatTensor `Graph` (dataloader.py:221-222; model.py:38)."""

    def __init__(self, rowptr, col, val, n_cols=None, edge_id=None, device=None):
        rowptr = np.ascontiguousarray(rowptr, np.int32)
        col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, np.float32)
        self.n_rows = len(rowptr) - 1
        self.n_cols = int(n_cols if n_cols is not None else self.n_rows)
        self.nnz = int(len(col))
        self.host = (rowptr, col, val)
        if edge_id is not None:
            edge_id = np.ascontiguousarray(edge_id, np.int32)
        self._edge_id_host = edge_id
        self.device = _device_of(device)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            args = (rowptr.ctypes.data_as(ctypes.c_void_p), col.ctypes.data_as(ctypes.c_void_p), val.ctypes.data_as(ctypes.c_void_p),
                    edge_id.ctypes.data_as(ctypes.c_void_p) if edge_id is not None else None, self.n_rows, self.n_cols, self.nnz)
            _lib.call("spex_graph_create", *args, ctypes.byref(handle))
        self._h = handle
        nl, ns = ctypes.c_int32(), ctypes.c_int32()
        _lib.call("spex_graph_info", self._h, None, None, None, ctypes.byref(nl), ctypes.byref(ns))
        self.n_long_rows, self.n_segments = nl.value, ns.value
        self._mask_ref = None
        self.mask_mode = 0

    # torch.sparse-like surface used by callers that only inspect the graph (dataloader.py:223 prints .size())
    def size(self):
        return torch.Size([self.n_rows, self.n_cols])

    shape = property(size)

    def _nnz(self):
        return self.nnz

    def to_torch_sparse(self, device="cpu"):
        rowptr, col, val = self.host
        rows = np.repeat(np.arange(self.n_rows, dtype=np.int64), np.diff(rowptr))
        idx = torch.from_numpy(np.stack([rows, col.astype(np.int64)]))
        return torch.sparse_coo_tensor(idx, torch.from_numpy(val), (self.n_rows, self.n_cols)).coalesce().to(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.detach_timer()
            self.release_workspace()
            with torch.cuda.device(self.device):
                _lib.load().spex_graph_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- profiling hook: hipEvent pairs around the main SpMM kernel, on the stream it is launched on
    def attach_timer(self, capacity, every=1):
        self.detach_timer()
        t = ctypes.c_void_p()
        with torch.cuda.device(self.device):          # the timer's events belong to the graph's device
            _lib.call("spex_timer_create", int(capacity), int(every), ctypes.byref(t))
        _lib.call("spex_timer_attach", self._h, t)
        self._timer, self._timer_cap = t, int(capacity)

    def read_timer(self, reset=True, per_launch=True):
        """Timings recorded since the last reset (synchronises on them).  per_launch=True: milliseconds per SpMM launch
        of every bracket (bracket time / launches in it); False: (bracket ms, launches per bracket)."""
        buf = (ctypes.c_float * self._timer_cap)()
        cnt = (ctypes.c_int32 * self._timer_cap)()
        n = ctypes.c_int32()
        _lib.call("spex_timer_read", self._timer, buf, cnt, self._timer_cap, ctypes.byref(n), 1 if reset else 0)
        ms = np.frombuffer(buf, dtype=np.float32, count=n.value).copy()
        launches = np.frombuffer(cnt, dtype=np.int32, count=n.value).copy()
        if per_launch:
            return ms / np.maximum(launches, 1)
        return ms, launches

    def detach_timer(self):
        if getattr(self, "_timer", None) is not None:
            _lib.call("spex_timer_attach", self._h, None)
            _lib.call("spex_timer_destroy", self._timer)
            self._timer = None

    # -- edge dropout (model.py:46-55)
    def set_edge_mask(self, mode=0, keep=None, keep_prob=1.0, seed=0):
        """mode 0: off; 1: injected device uint8 mask `keep` (indexed by edge id); 2: counter-based sampled mask."""
        if keep is not None:
            assert keep.dtype == torch.uint8 and keep.is_cuda and keep.is_contiguous()
        self._mask_ref = keep  # keep the tensor alive while the handle points at it
        self.mask_mode = int(mode) if float(keep_prob) < 1.0 else 0
        _lib.call("spex_graph_set_edge_mask", self._h, int(mode), _ptr(keep), float(keep_prob), int(seed))

    def _scratch(self, key, shape, device):
        """Per-handle workspace reused across calls (contents are dead once the call's launches are queued: calls on one
        handle are stream-ordered, see spex_hip.h) — no allocation per training step."""
        cache = self.__dict__.setdefault("_scratch_cache", {})
        t = cache.get(key)
        if t is None or t.shape != shape:
            t = cache[key] = torch.empty(shape, dtype=torch.float32, device=self.device)
        return t

    def release_workspace(self):
        """Drop the cached propagate / propagate_bwd workspaces (up to five [N, d] tables per handle)."""
        self.__dict__.pop("_scratch_cache", None)

    # -- kernels
    def _chk(self, t, rows, d, name):
        if t is None:
            return
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError(f"{name}: need a contiguous fp32 tensor on the GPU")
        if t.device != self.device:
            raise ValueError(f"{name}: tensor on {t.device}, graph on {self.device}")
        if t.shape != (rows, d):
            raise ValueError(f"{name}: shape {tuple(t.shape)} != {(rows, d)}")

    def spmm(self, X, Y=None, add_in=None, add_div=1.0, acc_in=None, acc_out=None, acc_div=1.0):
        """y = A X (+ add_in/add_div); Y = y; acc_out = (acc_in + y)/acc_div.  See spex_spmm_f32."""
        d = X.shape[1]
        self._chk(X, self.n_cols, d, "X")
        if Y is None and acc_out is None:
            Y = torch.empty((self.n_rows, d), dtype=torch.float32, device=X.device)
        for t, nm in ((Y, "Y"), (add_in, "add_in"), (acc_in, "acc_in"), (acc_out, "acc_out")):
            self._chk(t, self.n_rows, d, nm)
        if self.n_rows > 0:     # (an empty tensor has a NULL data_ptr; nothing to launch anyway)
            _launch(self.device, "spex_spmm_f32", self._h, _ptr(X), _ptr(Y), _ptr(add_in), float(add_div), _ptr(acc_in),
                    _ptr(acc_out), float(acc_div), d)
            _bump(Y, acc_out)
        return Y if Y is not None else acc_out

    def spmm_rows(self, X, idx_a, idx_b=None, off_a=0, off_b=0, Y=None, acc_in=None, acc_out=None, acc_div=1.0):
        """The product for the listed rows only (idx_a + off_a, idx_b + off_b; device int64), other rows of Y / acc_out
        untouched.  d == 64, no edge dropout.  See spex_spmm_rowlist_f32."""
        d = X.shape[1]
        self._chk(X, self.n_cols, d, "X")
        for t, nm in ((Y, "Y"), (acc_in, "acc_in"), (acc_out, "acc_out")):
            self._chk(t, self.n_rows, d, nm)
        for t in (idx_a, idx_b):
            if t is not None and not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
                raise ValueError("spmm_rows: row lists must be contiguous int64 tensors on the GPU")
        _launch(self.device, "spex_spmm_rowlist_f32", self._h, _ptr(X), _ptr(idx_a), idx_a.numel(), int(off_a), _ptr(idx_b),
                  0 if idx_b is None else idx_b.numel(), int(off_b), _ptr(Y), _ptr(acc_in), _ptr(acc_out), float(acc_div), d)
        _bump(Y, acc_out)
        return Y if Y is not None else acc_out

    def propagate(self, E0, n_layers, mean_out=None, layers_out=None, ws=None):
        """LightGCN.computer() (model.py:66-97) on a whole graph: returns mean(E0..EL)."""
        n, d = E0.shape
        self._chk(E0, self.n_cols, d, "E0")
        if mean_out is None:
            mean_out = torch.empty_like(E0)
        if layers_out is None and ws is None and n_layers > 1:
            ws = self._scratch("fwd", (2, n, d), E0.device)
        _launch(self.device, "spex_propagate_f32", self._h, _ptr(E0), _ptr(mean_out), _ptr(layers_out), _ptr(ws), int(n_layers), d)
        _bump(mean_out, layers_out)
        return mean_out

    def propagate_bwd(self, g_out, n_layers, grad_E0=None, ws=None):
        """Gradient of propagate() w.r.t. E0; `self` must be the handle of A^T."""
        n, d = g_out.shape
        self._chk(g_out, self.n_rows, d, "g_out")
        if grad_E0 is None:
            grad_E0 = torch.empty_like(g_out)
        if ws is None:
            ws = self._scratch("bwd", (3, n, d), g_out.device)
        _launch(self.device, "spex_propagate_bwd_f32", self._h, _ptr(g_out), _ptr(grad_E0), _ptr(ws), int(n_layers), d)
        _bump(grad_E0)
        return grad_E0

    # -- learned edge values (per-edge arrays are indexed by edge id)
    def _chk_edges(self, t, name):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 1):
            raise ValueError(f"{name}: need a contiguous 1-D fp32 tensor on the GPU")
        if t.device != self.device:
            raise ValueError(f"{name}: tensor on {t.device}, graph on {self.device}")
        if t.numel() < self.n_edge_ids:
            raise ValueError(f"{name}: {t.numel()} values for {self.n_edge_ids} edge ids")

    @property
    def n_edge_ids(self):
        """Length a per-edge array must have (largest edge id + 1; nnz for the identity numbering)."""
        if getattr(self, "_n_edge_ids", None) is None:
            self._n_edge_ids = self.nnz if self._edge_id_host is None else (int(self._edge_id_host.max()) + 1 if self.nnz else 0)
        return self._n_edge_ids

    def set_values(self, val):
        """Replace the stored values by val[edge id]; later spmm() launches use them (spex_graph_set_values)."""
        self._chk_edges(val, "val")
        _launch(self.device, "spex_graph_set_values", self._h, _ptr(val), val.numel())

    def sddmm(self, A, B, out=None):
        """out[edge id] = <A[row], B[col]> on the stored pattern: the SpMM's gradient w.r.t. its values."""
        d = A.shape[1]
        self._chk(A, self.n_rows, d, "A")
        self._chk(B, self.n_cols, d, "B")
        if out is None:   # every edge id is written when the ids are a permutation of the entries
            alloc = torch.empty if self.n_edge_ids == self.nnz else torch.zeros
            out = alloc(self.n_edge_ids, dtype=torch.float32, device=A.device)
        self._chk_edges(out, "out")
        if self.nnz:
            _launch(self.device, "spex_sddmm_f32", self._h, _ptr(A), _ptr(B), _ptr(out), out.numel(), d)
            _bump(out)
        return out

    def edge_softmax(self, v, out=None):
        """Row softmax over the stored entries (tf.sparse.softmax on a fixed pattern)."""
        self._chk_edges(v, "v")
        if out is None:
            out = torch.zeros_like(v)
        self._chk_edges(out, "out")
        if self.nnz:
            _launch(self.device, "spex_edge_softmax_f32", self._h, _ptr(v), _ptr(out), v.numel())
            _bump(out)
        return out

    def edge_softmax_bwd(self, y, grad_y, grad_in=None):
        self._chk_edges(y, "y")
        self._chk_edges(grad_y, "grad_y")
        if grad_in is None:
            grad_in = torch.zeros_like(y)
        self._chk_edges(grad_in, "grad_in")
        if self.nnz:
            _launch(self.device, "spex_edge_softmax_bwd_f32", self._h, _ptr(y), _ptr(grad_y), _ptr(grad_in), y.numel())
            _bump(grad_in)
        return grad_in
