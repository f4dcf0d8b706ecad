"""Thin torch-facing wrappers over libspexhip's kernels and the autograd Functions the drop-in models use.

PyTorch here is plumbing (device memory, streams, autograd bookkeeping); every flop below runs in the HIP kernels of
spex_amd/csrc/ through the C ABI.  No function in this module has a CPU path.
"""
import ctypes
import os

import torch

from . import _lib
from .graph import _bump, _launch, _ptr


def _need(t, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name}: need a contiguous {dtype} CUDA tensor (got {t.dtype}, cuda={t.is_cuda}, "
                         f"contiguous={t.is_contiguous()})")


VALIDATE_DEVICE_INDICES = os.environ.get("SPEX_VALIDATE_INDICES") == "1"

# Deterministic accumulation for the autograd Functions below (SPEX_DETERMINISTIC=1, or set_deterministic(True)): the batch's
# per-sample gradient rows are added per table row in ascending slot order (reduce_slots) instead of with float atomics, so the
# table gradient — and with it a whole LightGCN training run through the unchanged driver — repeats bit for bit.  The steppers
# of spex_amd.trainer take the same switch as a constructor argument (flags & SPEX_STEP_DETERMINISTIC of their descriptors).
DETERMINISTIC = os.environ.get("SPEX_DETERMINISTIC", "0") == "1"


def set_deterministic(on=True):
    global DETERMINISTIC
    DETERMINISTIC = bool(on)


def _score_bce_table_grad(light_out, n_u, u_idx, i_idx, labels, need_grad):
    """(loss_sum, dense d loss / d light_out or None) of the mean BCE over a batch — atomics, or under DETERMINISTIC per-sample
    rows added in slot order."""
    B = u_idx.numel()
    grad = torch.zeros_like(light_out) if need_grad else None
    W = light_out.shape[1]
    if need_grad and DETERMINISTIC and W % 64 == 0:
        slots = torch.empty((2 * B, W), dtype=torch.float32, device=light_out.device)
        _, loss_sum = score_bce(light_out[:n_u], light_out[n_u:], u_idx, i_idx, labels, None, None, 1.0 / B, grad_slots=slots)
        if W == 64:
            reduce_slots(u_idx, i_idx, n_u, light_out.shape[0], slots, grad)
        else:       # NGCF's concatenated table [N, 64 (L + 1)]: one 64-column block at a time (a validation mode, not a fast path)
            tmp = torch.empty((light_out.shape[0], 64), dtype=torch.float32, device=light_out.device)
            for c in range(0, W, 64):
                tmp.zero_()
                reduce_slots(u_idx, i_idx, n_u, light_out.shape[0], slots[:, c:c + 64], tmp)
                grad[:, c:c + 64] = tmp
        return loss_sum, grad
    _, loss_sum = score_bce(light_out[:n_u], light_out[n_u:], u_idx, i_idx, labels, grad[:n_u] if need_grad else None,
                            grad[n_u:] if need_grad else None, 1.0 / B)
    return loss_sum, grad


def _idx(t, device, bound=None):
    """Indices as the reference hands them over (int64, possibly on the host: main_rec.py:33-34).  With `bound`, indices
    outside [0, bound) raise IndexError like the reference's table lookup would — checked for free while the indices are
    still on the host; indices that already live on the device are only checked under SPEX_VALIDATE_INDICES=1 (a
    synchronising debug mode; the kernels themselves skip such samples instead of gathering out of bounds)."""
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    if bound is not None and t.numel() and (not t.is_cuda or VALIDATE_DEVICE_INDICES):
        lo, hi = int(t.min()), int(t.max())
        if lo < 0 or hi >= bound:
            raise IndexError(f"index {lo if lo < 0 else hi} is out of range for a table of {bound} rows")
    return t.to(device=device, dtype=torch.int64, non_blocking=True).contiguous()


# ------------------------------------------------------------------------------------------------ raw kernels
def score_bce(users_tab, items_tab, u_idx, i_idx, labels=None, grad_users=None, grad_items=None, grad_scale=0.0,
              loss_sum=None, want_gamma=True, grad_slots=None):
    """gamma (and, with labels, the BCE loss *sum* and optional gradient rows).  spex_score_bce_f32.
    loss_sum: a caller-owned 1-element buffer to ACCUMULATE into (no per-call allocation / fill); want_gamma=False skips
    the score vector (training steps do not read it)."""
    _need(users_tab, "users_tab"); _need(items_tab, "items_tab")
    dev = users_tab.device
    u_idx, i_idx = _idx(u_idx, dev, users_tab.shape[0]), _idx(i_idx, dev, items_tab.shape[0])
    B, d = u_idx.numel(), users_tab.shape[1]
    gamma = torch.empty(B, dtype=torch.float32, device=dev) if (want_gamma or labels is None) else None
    if labels is not None:
        labels = labels.to(device=dev, dtype=torch.float32).contiguous()
        if loss_sum is None:
            loss_sum = torch.zeros(1, dtype=torch.float32, device=dev)
    else:
        loss_sum = None
    _need(grad_users, "grad_users"); _need(grad_items, "grad_items")
    if grad_slots is not None:
        # the gradient also (or only) as per-sample rows [2B, ld]: row b = sample b's user-side row, row B + b its item-side row
        if not (grad_slots.is_cuda and grad_slots.dtype == torch.float32 and grad_slots.stride(1) == 1 and grad_slots.shape[0] >= 2 * B
                and grad_slots.shape[1] >= d):
            raise ValueError("score_bce: grad_slots must be an fp32 [>= 2B, >= d] device tensor with unit column stride")
        _launch(users_tab.device, "spex_score_bce_slots_f32", _ptr(users_tab), _ptr(items_tab), users_tab.stride(0), items_tab.stride(0),
                users_tab.shape[0], items_tab.shape[0], _ptr(u_idx), _ptr(i_idx), _ptr(labels), B, d, _ptr(loss_sum), _ptr(grad_users),
                _ptr(grad_items), float(grad_scale), _ptr(grad_slots), grad_slots.stride(0))
        _bump(grad_users, grad_items, loss_sum, grad_slots)
        return None, loss_sum
    _launch(users_tab.device, "spex_score_bce_f32", _ptr(users_tab), _ptr(items_tab), users_tab.stride(0), items_tab.stride(0),
              users_tab.shape[0], items_tab.shape[0], _ptr(u_idx), _ptr(i_idx), _ptr(labels), B, d, _ptr(gamma),
              _ptr(loss_sum), _ptr(grad_users), _ptr(grad_items), float(grad_scale))
    _bump(grad_users, grad_items, loss_sum)
    return gamma, loss_sum


GROUPED_BPR_MIN_TRIPLES = 1 << 18    # below this the single launch of the atomic form wins (measured crossover ~200 k triples)
_bpr_ws = {}


def _bpr_workspace(T, n_u, n_i, device):
    """Scratch of the grouped BPR form, cached per device and grown on demand (None when the form does not apply)."""
    need = _lib.load().spex_bpr_grouped_workspace_bytes(int(T), int(n_u), int(n_i))
    if need <= 0:
        return None
    ws = _bpr_ws.get(device)
    if ws is None or ws.numel() < need:
        ws = _bpr_ws[device] = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=device)
    return ws


def bpr_sgd_step(U_read, I_read, U_w, I_w, u, i_pos, i_neg, lr, reg=0.0, loss_sum=None, grouped=None):
    """Fused gather + dot + sigmoid + SGD over triples (north-star extension).  Returns the loss *sum* tensor
    (accumulated into `loss_sum` when the caller provides — and zeroes — the buffer).
    grouped: None = pick (the LDS-bucketed form from GROUPED_BPR_MIN_TRIPLES triples up, d == 64, batch-synchronous
    tables), True / False = force (spex_bpr_sgd_step_grouped_f32 / spex_bpr_sgd_step_f32)."""
    for t, n in ((U_read, "U_read"), (I_read, "I_read"), (U_w, "U_w"), (I_w, "I_w")):
        _need(t, n)
    dev = U_read.device
    u, i_pos, i_neg = _idx(u, dev), _idx(i_pos, dev), _idx(i_neg, dev)
    if loss_sum is None:
        loss_sum = torch.zeros(1, dtype=torch.float32, device=dev)
    T = u.numel()
    can_group = (U_read.shape[1] == 64 and U_read.data_ptr() != U_w.data_ptr() and I_read.data_ptr() != I_w.data_ptr())
    ws = _bpr_workspace(T, U_read.shape[0], I_read.shape[0], dev) if (can_group and grouped is not False) else None
    if grouped is None:
        grouped = ws is not None and T >= GROUPED_BPR_MIN_TRIPLES
    if grouped:
        if ws is None:
            raise ValueError("bpr_sgd_step: the grouped form needs d == 64, batch-synchronous tables and <= 8192 * 64 rows")
        _launch(dev, "spex_bpr_sgd_step_grouped_f32", _ptr(U_read), _ptr(I_read), _ptr(U_w), _ptr(I_w), U_read.shape[0],
                I_read.shape[0], _ptr(u), _ptr(i_pos), _ptr(i_neg), T, U_read.shape[1], float(lr), float(reg), _ptr(loss_sum),
                _ptr(ws), ws.numel())
    else:
        _launch(dev, "spex_bpr_sgd_step_f32", _ptr(U_read), _ptr(I_read), _ptr(U_w), _ptr(I_w), U_read.shape[0], I_read.shape[0],
                _ptr(u), _ptr(i_pos), _ptr(i_neg), T, U_read.shape[1], float(lr), float(reg), _ptr(loss_sum))
    _bump(U_w, I_w, loss_sum)
    return loss_sum


def bpr_loss_grad(users_tab, items_tab, u, i_pos, i_neg, grad_users=None, grad_items=None, grad_scale=0.0, grouped=None):
    _need(users_tab, "users_tab"); _need(items_tab, "items_tab")
    dev = users_tab.device
    u, i_pos, i_neg = _idx(u, dev), _idx(i_pos, dev), _idx(i_neg, dev)
    loss_sum = torch.zeros(1, dtype=torch.float32, device=dev)
    T = u.numel()
    ws = None
    if grad_users is not None and users_tab.shape[1] == 64 and grouped is not False and (grouped or T >= GROUPED_BPR_MIN_TRIPLES):
        ws = _bpr_workspace(T, users_tab.shape[0], items_tab.shape[0], dev)
    if ws is not None:
        _launch(dev, "spex_bpr_loss_grouped_f32", _ptr(users_tab), _ptr(items_tab), users_tab.shape[0], items_tab.shape[0],
                _ptr(u), _ptr(i_pos), _ptr(i_neg), T, users_tab.shape[1], _ptr(loss_sum), _ptr(grad_users), _ptr(grad_items),
                float(grad_scale), _ptr(ws), ws.numel())
    else:
        _launch(dev, "spex_bpr_loss_f32", _ptr(users_tab), _ptr(items_tab), users_tab.shape[0], items_tab.shape[0], _ptr(u),
                _ptr(i_pos), _ptr(i_neg), T, users_tab.shape[1], _ptr(loss_sum), _ptr(grad_users),
                _ptr(grad_items), float(grad_scale))
    _bump(grad_users, grad_items)
    return loss_sum


def adam_step(p, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, zero=None):
    """In-place fused Adam over one flat fp32 buffer (main_rec.py:23,37).  t counts from 1.  zero: an optional buffer
    of the same length cleared in the same pass (the next step's gradient accumulation table)."""
    for x, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (zero, "zero")):
        _need(x, n)
    if zero is not None and zero.numel() != p.numel():
        raise ValueError("adam_step: `zero` must have as many elements as the parameters")
    _launch(p.device, "spex_adam_step_f32", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), int(t), float(lr), float(beta1),
              float(beta2), float(eps), _ptr(zero))
    _bump(p, m, v, zero)


def ngcf_layer(ego, side, W_gc, b_gc, W_bi, b_bi, slope=0.01, want_e1=False):
    """[ego | normalize(LReLU(side W_gc^T + b) + LReLU((ego*side) W_bi^T + b))] — NGCF main_rec.py:77-85."""
    for x, n in ((ego, "ego"), (side, "side"), (W_gc, "W_gc"), (b_gc, "b_gc"), (W_bi, "W_bi"), (b_bi, "b_bi")):
        _need(x, n)
    n, d = ego.shape
    out = torch.empty((n, 2 * d), dtype=torch.float32, device=ego.device)
    e1 = torch.empty((n, d), dtype=torch.float32, device=ego.device) if want_e1 else None
    _launch(ego.device, "spex_ngcf_layer_f32", _ptr(ego), _ptr(side), _ptr(W_gc), _ptr(b_gc), _ptr(W_bi), _ptr(b_bi), _ptr(out),
              2 * d, _ptr(e1), n, d, float(slope))
    return (out, e1) if want_e1 else out


def ngcf_layer_fwd(ego, side, W_gc, b_gc, W_bi, b_bi, out, layer, write_ego, e1_out=None, slope=0.01, drop=None,
                   pad_row=-1):
    """One NGCF layer into its slice of the concatenated table `out` ([n, d * (L + 1)]): normalised output to columns
    d*(layer+1) .. d*(layer+2), ego to columns 0..d iff write_ego; e1_out receives the layer's un-normalised output
    (after dropout) = the next layer's input.  drop: None or (p, seed, step) — counter-based message dropout,
    spex_ngcf_layer_fwd_f32."""
    for x, n in ((ego, "ego"), (side, "side"), (W_gc, "W_gc"), (b_gc, "b_gc"), (W_bi, "W_bi"), (b_bi, "b_bi"), (out, "out"),
                 (e1_out, "e1_out")):
        _need(x, n)
    n, d = ego.shape
    p, seed, step = drop if drop is not None else (0.0, 0, 0)
    base = ctypes.c_void_p(out.data_ptr() + 4 * d * layer)
    _launch(ego.device, "spex_ngcf_layer_fwd_f32", _ptr(ego), _ptr(side), _ptr(W_gc), _ptr(b_gc), _ptr(W_bi), _ptr(b_bi), base,
              out.stride(0), 1 if write_ego else 0, _ptr(e1_out), n, d, float(slope), float(p), int(seed), int(step), int(layer),
              int(pad_row))
    _bump(out, e1_out)


def ngcf_layer_fwd_rows(ego, side, W_gc, b_gc, W_bi, b_bi, out, idx_a, idx_b, off_b, write_ego=True, slope=0.01, drop=None, pad_row=-1):
    """The first NGCF layer at a list of rows only (spex_ngcf_layer_fwd_rows_f32): rows idx_a[k] and idx_b[k] + off_b of the dense
    tables `ego` / `side` -> the same rows of `out` ([n, >= 128]: [ego | normalised layer output]); other rows are left untouched."""
    for x, nm in ((ego, "ego"), (side, "side"), (W_gc, "W_gc"), (b_gc, "b_gc"), (W_bi, "W_bi"), (b_bi, "b_bi"), (out, "out")):
        _need(x, nm)
    n, d = ego.shape
    for t in (idx_a, idx_b):
        if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
            raise ValueError("ngcf_layer_fwd_rows: the row lists must be contiguous int64 tensors on the GPU")
    p, seed, step = drop if drop is not None else (0.0, 0, 0)
    _launch(ego.device, "spex_ngcf_layer_fwd_rows_f32", _ptr(ego), _ptr(side), _ptr(W_gc), _ptr(b_gc), _ptr(W_bi), _ptr(b_bi), _ptr(out),
            out.stride(0), 1 if write_ego else 0, n, d, float(slope), float(p), int(seed), int(step), 0, int(pad_row), _ptr(idx_a),
            idx_a.numel(), 0, _ptr(idx_b), idx_b.numel(), int(off_b))
    _bump(out)


def ngcf_layer_bwd(ego, side, W_gc, b_gc, W_bi, b_bi, g_all, layer, g_next, g_side, g_ego, gW_gc, gb_gc, gW_bi, gb_bi,
                   slope=0.01, drop=None, pad_row=-1):
    """Backward of ngcf_layer_fwd for layer `layer` given g_all = d loss / d (concatenated table): reads its slice (and,
    for layer 0, the ego slice as g_direct), writes g_side / g_ego, accumulates the four weight gradients.
    spex_ngcf_layer_bwd_f32."""
    for x, n in ((ego, "ego"), (side, "side"), (W_gc, "W_gc"), (b_gc, "b_gc"), (W_bi, "W_bi"), (b_bi, "b_bi"), (g_all, "g_all"),
                 (g_next, "g_next"), (g_side, "g_side"), (g_ego, "g_ego"), (gW_gc, "gW_gc"), (gb_gc, "gb_gc"), (gW_bi, "gW_bi"),
                 (gb_bi, "gb_bi")):
        _need(x, n)
    n, d = ego.shape
    p, seed, step = drop if drop is not None else (0.0, 0, 0)
    ld = g_all.stride(0)
    g_norm = ctypes.c_void_p(g_all.data_ptr() + 4 * d * (layer + 1))
    g_direct = ctypes.c_void_p(g_all.data_ptr()) if layer == 0 else None
    _launch(ego.device, "spex_ngcf_layer_bwd_f32", _ptr(ego), _ptr(side), _ptr(W_gc), _ptr(b_gc), _ptr(W_bi), _ptr(b_bi), g_norm, ld,
              _ptr(g_next), g_direct, ld, n, d, float(slope), float(p), int(seed), int(step), int(layer), int(pad_row),
              _ptr(g_side), _ptr(g_ego), _ptr(gW_gc), _ptr(gb_gc), _ptr(gW_bi), _ptr(gb_bi))
    _bump(g_side, g_ego, gW_gc, gb_gc, gW_bi, gb_bi)


def ngcf_layer_bwd_rows(ego, side, W_gc, b_gc, W_bi, b_bi, g_slots, layer, g_next, idx_a, idx_b, off_b, g_side_c, g_ego_c, gW_parts,
                        slope=0.01, drop=None, pad_row=-1):
    """ngcf_layer_bwd for the rows of a BATCH only (the last layer's backward in training): slot k is row idx_a[k], slot
    len(idx_a) + k is row idx_b[k] + off_b, each with its own upstream gradient row g_slots[k] ([slots, d (L + 1)] as
    score_bce(grad_slots=...) writes it); compact outputs g_side_c / g_ego_c [slots, d]; the weight gradients leave as
    partial blocks gW_parts [ngcf_bwd_rows_parts(slots), >= 2 (d d + d)].  spex_ngcf_layer_bwd_rows_f32."""
    for x, n in ((ego, "ego"), (side, "side"), (W_gc, "W_gc"), (b_gc, "b_gc"), (W_bi, "W_bi"), (b_bi, "b_bi"), (g_slots, "g_slots"),
                 (g_next, "g_next"), (g_side_c, "g_side_c"), (g_ego_c, "g_ego_c"), (gW_parts, "gW_parts")):
        _need(x, n)
    n, d = ego.shape
    n_a, n_b = idx_a.numel(), idx_b.numel()
    if (g_side_c.shape[0] < n_a + n_b or g_ego_c.shape[0] < n_a + n_b or g_slots.shape[0] < n_a + n_b
            or gW_parts.shape[0] < ngcf_bwd_rows_parts(n_a + n_b)):
        raise ValueError("ngcf_layer_bwd_rows: per-slot arrays / partial blocks are too small for the batch")
    for t in (idx_a, idx_b):
        if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
            raise ValueError("ngcf_layer_bwd_rows: the batch must be contiguous int64 tensors on the GPU")
    p, seed, step = drop if drop is not None else (0.0, 0, 0)
    ld = g_slots.stride(0)
    g_norm = ctypes.c_void_p(g_slots.data_ptr() + 4 * d * (layer + 1))
    g_direct = ctypes.c_void_p(g_slots.data_ptr()) if layer == 0 else None
    _launch(ego.device, "spex_ngcf_layer_bwd_rows_f32", _ptr(ego), _ptr(side), _ptr(W_gc), _ptr(b_gc), _ptr(W_bi), _ptr(b_bi), g_norm,
            ld, _ptr(g_next), g_direct, ld, n, d, float(slope), float(p), int(seed), int(step), int(layer), int(pad_row),
            _ptr(idx_a), n_a, 0, _ptr(idx_b), n_b, int(off_b), _ptr(g_side_c), _ptr(g_ego_c), _ptr(gW_parts), gW_parts.stride(0))
    _bump(g_side_c, g_ego_c, gW_parts)


def ngcf_score_bwd_rows(ego, side, W_gc, b_gc, W_bi, b_bi, all_emb, labels, grad_scale, users, items, n_user_rows, loss_per_sample,
                        g_side_c, g_ego_c, gW_parts, slope=0.01, drop=None, pad_row=-1):
    """score_bce(grad_slots=...) on the concatenated table all_emb [n, 128] followed by ngcf_layer_bwd_rows (layer 0), as ONE
    launch: per-sample losses -> loss_per_sample [B]; g_side_c / g_ego_c [2B, 64]; gW_parts as ngcf_layer_bwd_rows writes them.
    spex_ngcf_score_bwd_rows_f32."""
    for x, nm in ((ego, "ego"), (side, "side"), (W_gc, "W_gc"), (b_gc, "b_gc"), (W_bi, "W_bi"), (b_bi, "b_bi"), (all_emb, "all_emb"),
                  (labels, "labels"), (loss_per_sample, "loss_per_sample"), (g_side_c, "g_side_c"), (g_ego_c, "g_ego_c"), (gW_parts, "gW_parts")):
        _need(x, nm)
    n, d = ego.shape
    B = users.numel()
    if all_emb.shape != (n, 2 * d) or not all_emb.is_contiguous():
        raise ValueError("ngcf_score_bwd_rows: all_emb must be a contiguous [n, 2 d] table")
    if (items.numel() != B or labels.numel() != B or loss_per_sample.numel() < B or g_side_c.shape[0] < 2 * B or g_ego_c.shape[0] < 2 * B
            or gW_parts.shape[0] < ngcf_bwd_rows_parts(2 * B)):
        raise ValueError("ngcf_score_bwd_rows: per-sample / per-slot arrays are too small for the batch")
    for t in (users, items):
        if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
            raise ValueError("ngcf_score_bwd_rows: the batch must be contiguous int64 tensors on the GPU")
    p, seed, step = drop if drop is not None else (0.0, 0, 0)
    _launch(ego.device, "spex_ngcf_score_bwd_rows_f32", _ptr(ego), _ptr(side), _ptr(W_gc), _ptr(b_gc), _ptr(W_bi), _ptr(b_bi),
            _ptr(all_emb), _ptr(labels), float(grad_scale), n, d, float(slope), float(p), int(seed), int(step), 0, int(pad_row),
            _ptr(users), _ptr(items), B, int(n_user_rows), _ptr(loss_per_sample), _ptr(g_side_c), _ptr(g_ego_c), _ptr(gW_parts),
            gW_parts.stride(0))
    _bump(loss_per_sample, g_side_c, g_ego_c, gW_parts)


def ngcf_fwd_score_bwd_rows(ego, side, W_gc, b_gc, W_bi, b_bi, labels, grad_scale, users, items, n_user_rows, loss_per_sample,
                            g_side_c, g_ego_c, gW_parts, slope=0.01, drop=None, pad_row=-1):
    """The layer's forward at the batch's rows, the scoring and the rows backward in ONE launch, without the concatenated table
    (spex_ngcf_fwd_score_bwd_rows_f32): ego / side are the dense tables (side = A ego valid at the batch's rows); outputs as
    ngcf_score_bwd_rows."""
    for x, nm in ((ego, "ego"), (side, "side"), (W_gc, "W_gc"), (b_gc, "b_gc"), (W_bi, "W_bi"), (b_bi, "b_bi"), (labels, "labels"),
                  (loss_per_sample, "loss_per_sample"), (g_side_c, "g_side_c"), (g_ego_c, "g_ego_c"), (gW_parts, "gW_parts")):
        _need(x, nm)
    n, d = ego.shape
    B = users.numel()
    if (items.numel() != B or labels.numel() != B or loss_per_sample.numel() < B or g_side_c.shape[0] < 2 * B or g_ego_c.shape[0] < 2 * B
            or gW_parts.shape[0] < ngcf_bwd_rows_parts(2 * B)):
        raise ValueError("ngcf_fwd_score_bwd_rows: per-sample / per-slot arrays are too small for the batch")
    for t in (users, items):
        if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
            raise ValueError("ngcf_fwd_score_bwd_rows: the batch must be contiguous int64 tensors on the GPU")
    p, seed, step = drop if drop is not None else (0.0, 0, 0)
    _launch(ego.device, "spex_ngcf_fwd_score_bwd_rows_f32", _ptr(ego), _ptr(side), _ptr(W_gc), _ptr(b_gc), _ptr(W_bi), _ptr(b_bi),
            _ptr(labels), float(grad_scale), n, d, float(slope), float(p), int(seed), int(step), 0, int(pad_row), _ptr(users), _ptr(items),
            B, int(n_user_rows), _ptr(loss_per_sample), _ptr(g_side_c), _ptr(g_ego_c), _ptr(gW_parts), gW_parts.stride(0))
    _bump(loss_per_sample, g_side_c, g_ego_c, gW_parts)


def ngcf_bwd_rows_parts(n_slots):
    """Number of partial weight-gradient blocks ngcf_layer_bwd_rows writes for a batch of n_slots slots."""
    return int(_lib.load().spex_ngcf_layer_bwd_rows_parts(int(n_slots)))


def adam_step_sum(p, g_parts, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """Adam over a flat parameter block whose gradient is the sum of the rows of g_parts [n_parts, >= p.numel()]
    (spex_adam_step_sum_f32)."""
    for x, n in ((p, "p"), (g_parts, "g_parts"), (m, "m"), (v, "v")):
        _need(x, n)
    _launch(p.device, "spex_adam_step_sum_f32", _ptr(p), _ptr(g_parts), g_parts.shape[0], g_parts.stride(0), _ptr(m), _ptr(v),
            p.numel(), int(t), float(lr), float(beta1), float(beta2), float(eps))
    _bump(p, m, v)


class UniqueRows:
    """The distinct rows of a batch as a compact device list (spex_unique_rows_i32): `list` int32[capacity], `count`
    int32[1], refreshed by update() — one tiny launch, no sort, no host round trip.  n_rows: height of the table the rows
    index."""

    def __init__(self, n_rows, capacity, device):
        self.n_rows, self.capacity = int(n_rows), int(capacity)
        self.stamp = torch.zeros(self.n_rows, dtype=torch.int32, device=device)
        self.list = torch.zeros(self.capacity, dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)
        self.epoch = 0

    def update(self, idx_a, idx_b=None, off_a=0, off_b=0):
        n_a, n_b = idx_a.numel(), 0 if idx_b is None else idx_b.numel()
        if n_a + n_b > self.capacity:
            raise ValueError(f"UniqueRows: {n_a + n_b} indices for a capacity of {self.capacity}")
        for t in (idx_a, idx_b):
            if t is not None and not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
                raise ValueError("UniqueRows.update: indices must be contiguous int64 tensors on the GPU")
        self.epoch = self.epoch % 0x7FFFFFFF + 1          # never 0; a wrap after 2^31 - 1 steps re-uses stamps that are long stale
        _launch(self.list.device, "spex_unique_rows_i32", _ptr(idx_a), n_a, int(off_a), _ptr(idx_b), n_b, int(off_b), self.n_rows,
                _ptr(self.stamp), self.epoch, _ptr(self.list), _ptr(self.count))
        return self


def spmm_push_rows(graph, rows, src, out, src_indexed, add=None, add_indexed=False, scale=1.0):
    """out += scale * (A^T scatter(src) + scatter(add)) for the rows of a UniqueRows list, A = the matrix of `graph`
    (spex_spmm_push_rows_f32).  src / add: the table itself (indexed=True) or compact [capacity, d] arrays."""
    _need(src, "src"); _need(out, "out"); _need(add, "add")
    d = out.shape[1]
    if out.shape[0] != graph.n_cols or rows.n_rows > graph.n_rows:
        raise ValueError("spmm_push_rows: out must be [graph.n_cols, d] and the row list must index rows of the graph")
    for t, ind, nm in ((src, src_indexed, "src"), (add, add_indexed, "add")):
        if t is not None and (t.shape[1] != d or t.shape[0] < (rows.n_rows if ind else rows.capacity)):
            raise ValueError(f"spmm_push_rows: {nm} has shape {tuple(t.shape)}")
    _launch(out.device, "spex_spmm_push_rows_f32", graph._h, _ptr(rows.list), _ptr(rows.count), rows.capacity, _ptr(src),
            1 if src_indexed else 0, _ptr(add), 1 if add_indexed else 0, float(scale), _ptr(out), d)
    _bump(out)
    return out


def spmm_push_batch(graph, idx_a, idx_b, off_b, src, out, add=None, scale=1.0):
    """out += scale * (A^T scatter(src) + scatter(add)), every slot of the batch contributing its own row (slot k: row
    idx_a[k]; slot len(idx_a) + k: row idx_b[k] + off_b; src / add: per-slot [slots, 64] arrays, e.g. score_bce's
    grad_slots), one launch, d == 64 (spex_spmm_push_batch_f32)."""
    if out.shape != (graph.n_cols, 64) or not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()):
        raise ValueError("spmm_push_batch: out must be a contiguous fp32 [graph.n_cols, 64] device tensor")
    for t in (idx_a, idx_b):
        if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
            raise ValueError("spmm_push_batch: the batch must be contiguous int64 tensors on the GPU")
    slots = idx_a.numel() + idx_b.numel()
    for t, nm in ((src, "src"), (add, "add")):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.stride(1) == 1 and t.shape[0] >= slots and t.shape[1] >= 64):
            raise ValueError(f"spmm_push_batch: {nm} must be an fp32 [>= slots, >= 64] device tensor with unit column stride")
    _launch(out.device, "spex_spmm_push_batch_f32", graph._h, _ptr(idx_a), idx_a.numel(), 0, _ptr(idx_b), idx_b.numel(), int(off_b),
            _ptr(src), src.stride(0), _ptr(add), 0 if add is None else add.stride(0), float(scale), _ptr(out), 64)
    _bump(out)
    return out


def reduce_slots(idx_a, idx_b, off_b, n_rows, slots, out, scale=1.0, accumulate=False):
    """Deterministic accumulation of a batch's per-slot rows into a dense [n_rows, 64] table (spex_reduce_slots_f32): out[r] =
    scale * sum of the slots naming r, in ascending slot order (overwritten, or added with accumulate); slots=None clears the
    named rows.  The atomic-free alternative to score_bce's dense gradient tables / spmm_push_batch's `add` term."""
    if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.shape == (n_rows, 64)):
        raise ValueError(f"reduce_slots: out must be a contiguous fp32 [{n_rows}, 64] device tensor")
    for t in (idx_a, idx_b):
        if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
            raise ValueError("reduce_slots: the batch must be contiguous int64 tensors on the GPU")
    n = idx_a.numel() + idx_b.numel()
    if slots is not None and not (slots.is_cuda and slots.dtype == torch.float32 and slots.stride(1) == 1 and slots.shape[0] >= n
                                  and slots.shape[1] >= 64):
        raise ValueError("reduce_slots: slots must be an fp32 [>= slots, >= 64] device tensor with unit column stride")
    _launch(out.device, "spex_reduce_slots_f32", _ptr(idx_a), idx_a.numel(), 0, _ptr(idx_b), idx_b.numel(), int(off_b), int(n_rows),
            _ptr(slots), 64 if slots is None else slots.stride(0), float(scale), _ptr(out), 1 if accumulate else 0, 64)
    _bump(out)
    return out


def lightgcn_batch_slots(graph, X, acc_in, acc_div, users, items, labels, n_user_rows, grad_scale, grad_slots, loss_sum=None,
                         loss_per_sample=None):
    """spex_lightgcn_batch_slots_f32: the batch kernel's forward + scoring with the two gradient rows of every sample written
    as per-sample slots (grad_slots[b], grad_slots[B + b]) — no push, no float atomics (the deterministic step)."""
    n = graph.n_rows
    for t, nm in ((X, "X"), (acc_in, "acc_in")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == (n, 64)):
            raise ValueError(f"lightgcn_batch_slots: {nm} must be a contiguous fp32 [{n}, 64] device tensor")
    B = users.numel()
    for t, dt, nm in ((users, torch.int64, "users"), (items, torch.int64, "items"), (labels, torch.float32, "labels")):
        if not (t.is_cuda and t.dtype == dt and t.is_contiguous() and t.numel() == B):
            raise ValueError(f"lightgcn_batch_slots: {nm} must be a contiguous device tensor of the batch's length")
    if not (grad_slots.is_cuda and grad_slots.dtype == torch.float32 and grad_slots.is_contiguous() and grad_slots.shape[0] >= 2 * B
            and grad_slots.shape[1] == 64):
        raise ValueError("lightgcn_batch_slots: grad_slots must be a contiguous fp32 [>= 2B, 64] device tensor")
    if loss_sum is None and loss_per_sample is None:
        raise ValueError("lightgcn_batch_slots: needs loss_sum or loss_per_sample")
    _launch(X.device, "spex_lightgcn_batch_slots_f32", graph._h, _ptr(X), _ptr(acc_in), float(acc_div), _ptr(users), _ptr(items),
            _ptr(labels), B, int(n_user_rows), float(grad_scale), _ptr(loss_sum), _ptr(loss_per_sample), _ptr(grad_slots), 64)
    _bump(loss_sum, loss_per_sample, grad_slots)


def lightgcn_batch(graph, X, acc_in, acc_div, users, items, labels, n_user_rows, grad_scale, push_scale, loss_sum, g_out, G,
                   loss_per_sample=None):
    """The batch-sized middle of the exact LightGCN step as one launch (spex_lightgcn_batch_f32): last layer + layer mean at
    the batch's rows, scores + BCE, dense gradient rows added into g_out, and G += push_scale * (g + A^T g) in push form (over
    the rows of `graph` itself: A^T g in push form walks the rows of A).
    loss_sum: fp32 [1] (accumulated) — or loss_per_sample: fp32 [B], every sample's loss stored instead; g_out, G: fp32 [N, 64]
    (accumulated: zero them first)."""
    n = graph.n_rows
    for t, nm in ((X, "X"), (acc_in, "acc_in"), (g_out, "g_out"), (G, "G")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == (n, 64)):
            raise ValueError(f"lightgcn_batch: {nm} must be a contiguous fp32 [{n}, 64] device tensor")
    B = users.numel()
    for t, dt, nm in ((users, torch.int64, "users"), (items, torch.int64, "items"), (labels, torch.float32, "labels")):
        if not (t.is_cuda and t.dtype == dt and t.is_contiguous() and t.numel() == B):
            raise ValueError(f"lightgcn_batch: {nm} must be a contiguous device tensor of the batch's length")
    for t, k, nm in ((loss_sum, 1, "loss_sum"), (loss_per_sample, B, "loss_per_sample")):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() >= k):
            raise ValueError(f"lightgcn_batch: {nm} must be a contiguous fp32 device tensor of >= {k} elements")
    if loss_sum is None and loss_per_sample is None:
        raise ValueError("lightgcn_batch: needs loss_sum or loss_per_sample")
    _launch(X.device, "spex_lightgcn_batch_f32", graph._h, _ptr(X), _ptr(acc_in), float(acc_div), _ptr(users), _ptr(items),
            _ptr(labels), B, int(n_user_rows), float(grad_scale), float(push_scale), _ptr(loss_sum), _ptr(loss_per_sample), _ptr(g_out),
            _ptr(G), 64)
    _bump(loss_sum, loss_per_sample, g_out, G)
    return loss_sum if loss_per_sample is None else loss_per_sample


def gated_batch_fwd(graph, X, acc_in, acc_div, raw, att_u, att_i, users, items, labels, n_user_rows, grad_scale, loss_sum, lo_batch,
                    grad_slots, loss_per_sample=None):
    """Forward half of the dual-task rec branch's batch-sized middle as one launch (spex_gated_batch_fwd_f32): last layer + layer
    mean at the batch's rows (-> lo_batch rows), expert gate, scores + BCE (-> loss_sum, accumulated), per-sample gradient rows
    with respect to the gated rows (-> grad_slots [2B, 64])."""
    n = graph.n_rows
    for t, nm in ((X, "X"), (acc_in, "acc_in"), (raw, "raw"), (lo_batch, "lo_batch")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == (n, 64)):
            raise ValueError(f"gated_batch_fwd: {nm} must be a contiguous fp32 [{n}, 64] device tensor")
    B = users.numel()
    for t, nm in ((att_u, "att_u"), (att_i, "att_i")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == 256):
            raise ValueError(f"gated_batch_fwd: {nm} must be a contiguous fp32 [128, 2] device tensor")
    for t, dt, nm in ((users, torch.int64, "users"), (items, torch.int64, "items"), (labels, torch.float32, "labels")):
        if not (t.is_cuda and t.dtype == dt and t.is_contiguous() and t.numel() == B):
            raise ValueError(f"gated_batch_fwd: {nm} must be a contiguous device tensor of the batch's length")
    if not (grad_slots.is_cuda and grad_slots.dtype == torch.float32 and grad_slots.is_contiguous() and grad_slots.shape[0] >= 2 * B
            and grad_slots.shape[1] == 64):
        raise ValueError("gated_batch_fwd: grad_slots must be a contiguous fp32 [>= 2B, 64] device tensor")
    _launch(X.device, "spex_gated_batch_fwd_f32", graph._h, _ptr(X), _ptr(acc_in), float(acc_div), _ptr(raw), _ptr(att_u), _ptr(att_i),
            _ptr(users), _ptr(items), _ptr(labels), B, int(n_user_rows), float(grad_scale), _ptr(loss_sum), _ptr(loss_per_sample),
            _ptr(lo_batch), _ptr(grad_slots), 64)
    _bump(loss_sum, loss_per_sample, lo_batch, grad_slots)


def gated_batch(graph, X, acc_in, acc_div, raw, att_u, att_i, users, items, labels, n_user_rows, grad_scale, push_scale, loss_sum,
                g_prop, G, g_raw, g_att):
    """The dual-task rec branch's whole batch-sized middle as one launch (spex_gated_batch_f32): gated_batch_fwd's forward, the
    gate's backward at the sample's two rows and the first backward product in push form.  Accumulates (float atomics) into
    g_prop / G / g_raw ([N, 64], three tables), loss_sum and g_att [copies, 2, 128, 2]: sample b adds the two gate matrices'
    gradients into copy b mod copies (g_att.sum(0) is [d att_u | d att_i])."""
    n = graph.n_rows
    for t, nm in ((X, "X"), (acc_in, "acc_in"), (raw, "raw"), (g_prop, "g_prop"), (G, "G"), (g_raw, "g_raw")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == (n, 64)):
            raise ValueError(f"gated_batch: {nm} must be a contiguous fp32 [{n}, 64] device tensor")
    B = users.numel()
    for t, nm in ((att_u, "att_u"), (att_i, "att_i")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == 256):
            raise ValueError(f"gated_batch: {nm} must be a contiguous fp32 [128, 2] device tensor")
    if not (g_att.is_cuda and g_att.dtype == torch.float32 and g_att.is_contiguous() and g_att.dim() == 4 and g_att.shape[1:] == (2, 128, 2)
            and g_att.shape[0] >= 1):
        raise ValueError("gated_batch: g_att must be a contiguous fp32 [copies, 2, 128, 2] device tensor")
    for t, dt, nm in ((users, torch.int64, "users"), (items, torch.int64, "items"), (labels, torch.float32, "labels")):
        if not (t.is_cuda and t.dtype == dt and t.is_contiguous() and t.numel() == B):
            raise ValueError(f"gated_batch: {nm} must be a contiguous device tensor of the batch's length")
    _launch(X.device, "spex_gated_batch_f32", graph._h, _ptr(X), _ptr(acc_in), float(acc_div), _ptr(raw), _ptr(att_u), _ptr(att_i),
            _ptr(users), _ptr(items), _ptr(labels), B, int(n_user_rows), float(grad_scale), float(push_scale), _ptr(loss_sum),
            _ptr(g_prop), _ptr(G), _ptr(g_raw), _ptr(g_att), g_att.shape[0], 64)
    _bump(loss_sum, g_prop, G, g_raw, g_att)


def expert_gate(raw, prop, att_exp):
    """softmax([raw|prop] att_exp) two-way mix — model_expert_s.py:156-161."""
    for x, n in ((raw, "raw"), (prop, "prop"), (att_exp, "att_exp")):
        _need(x, n)
    mixed = torch.empty_like(raw)
    _launch(raw.device, "spex_expert_gate_f32", _ptr(raw), _ptr(prop), _ptr(att_exp), _ptr(mixed), raw.shape[0], raw.shape[1])
    return mixed


def sample_negatives(rowptr, items, pos_user, num_ng, num_item, seed):
    """num_ng uniform non-interacted items per positive, on the GPU (spex_sample_negatives).  rowptr / items: device
    int32 CSR of the user-item matrix R with sorted rows; pos_user: device int64.  Returns int64 [len(pos_user)*num_ng]."""
    for t, n, dt in ((rowptr, "rowptr", torch.int32), (items, "items", torch.int32), (pos_user, "pos_user", torch.int64)):
        if not (t.is_cuda and t.dtype == dt and t.is_contiguous()):
            raise ValueError(f"{n}: need a contiguous {dt} CUDA tensor")
    out = torch.empty(pos_user.numel() * num_ng, dtype=torch.int64, device=pos_user.device)
    _launch(rowptr.device, "spex_sample_negatives", _ptr(rowptr), _ptr(items), rowptr.numel() - 1, _ptr(pos_user), pos_user.numel(),
              int(num_ng), int(num_item), int(seed), _ptr(out))
    return out


# ------------------------------------------------------------------------------------------------ autograd glue
def _flat_tables(user_w, item_w, strict=False):
    """The two embedding tables as one [N, d] buffer.  The drop-in model allocates them back-to-back so this is a
    view; otherwise (foreign parameters, e.g. after load_state_dict(assign=True)) it costs one concatenation, like
    model.py:72 — or, with strict=True (a caller that will WRITE through the result), raises."""
    if (user_w.is_contiguous() and item_w.is_contiguous()
            and user_w.untyped_storage().data_ptr() == item_w.untyped_storage().data_ptr()
            and user_w.data_ptr() + user_w.numel() * 4 == item_w.data_ptr()):
        n = user_w.shape[0] + item_w.shape[0]
        return torch.as_strided(user_w.detach(), (n, user_w.shape[1]), (user_w.shape[1], 1))
    if strict:
        raise RuntimeError("the two embedding tables no longer share one contiguous buffer (parameters were replaced?): "
                           "call the model's _fuse_tables() before training through flat_table()")
    return torch.cat([user_w.detach(), item_w.detach()])


class PropagateMean(torch.autograd.Function):
    """light_out = mean_l(A^l E0) — LightGCN.computer(), model.py:66-97, with its autograd backward.

    graph / graph_t: SpexGraph of A and of A^T (the same object for the symmetric LightGCN adjacency without dropout).
    mask: None or (mode, keep_tensor, keep_prob, seed) applied identically in forward and backward.
    """

    @staticmethod
    def forward(ctx, user_w, item_w, graph, graph_t, n_layers, mask):
        E0 = _flat_tables(user_w, item_w)
        if mask is not None:
            graph.set_edge_mask(*mask)
        try:
            out = graph.propagate(E0, n_layers)
        finally:
            if mask is not None:
                graph.set_edge_mask(0)
        ctx.graph_t, ctx.n_layers, ctx.mask, ctx.n_user_rows = graph_t, n_layers, mask, user_w.shape[0]
        return out

    @staticmethod
    def backward(ctx, g_out):
        g_out = g_out.contiguous()
        gt = ctx.graph_t
        if ctx.mask is not None:
            gt.set_edge_mask(*ctx.mask)
        try:
            gE0 = gt.propagate_bwd(g_out, ctx.n_layers)
        finally:
            if ctx.mask is not None:
                gt.set_edge_mask(0)
        u = ctx.n_user_rows
        return gE0[:u], gE0[u:], None, None, None, None


class PropagateMeanFolds(torch.autograd.Function):
    """PropagateMean with the adjacency held as row blocks (`--A_split`, dataloader.py:167-177, model.py:84-89): one SpMM
    per block and layer, forward on the blocks of A, backward on the blocks of A^T.  A row's sum is the same fmaf chain
    whichever block holds the row, so output and gradient are bit-identical to the unsplit path.
    folds / folds_t: lists of SpexGraph row blocks of A / of A^T (n_cols = N); mask as in PropagateMean (edge ids are
    positions in the unsplit matrix)."""

    @staticmethod
    def _masked(graphs, mask, fn):
        if mask is not None:
            for g in graphs:
                g.set_edge_mask(*mask)
        try:
            return fn()
        finally:
            if mask is not None:
                for g in graphs:
                    g.set_edge_mask(0)

    @staticmethod
    def forward(ctx, user_w, item_w, folds, folds_t, n_layers, mask):
        E0 = _flat_tables(user_w, item_w)
        L = n_layers
        out = torch.empty_like(E0)

        def run():
            cur = E0
            for l in range(L):
                last = l == L - 1
                nxt = None if last else torch.empty_like(E0)
                r0 = 0
                for g in folds:
                    r1 = r0 + g.n_rows
                    g.spmm(cur, Y=None if last else nxt[r0:r1], acc_in=(E0 if l == 0 else out)[r0:r1], acc_out=out[r0:r1],
                           acc_div=float(L + 1) if last else 1.0)
                    r0 = r1
                cur = nxt
            if L == 0:
                out.copy_(E0)
            return out
        PropagateMeanFolds._masked(folds, mask, run)
        ctx.folds_t, ctx.L, ctx.mask, ctx.n_user_rows = folds_t, L, mask, user_w.shape[0]
        return out

    @staticmethod
    def backward(ctx, g_out):
        L, folds_t = ctx.L, ctx.folds_t
        gs = g_out.contiguous() / float(L + 1)

        def run():
            cur = gs
            for _ in range(L):
                nxt = torch.empty_like(gs)
                r0 = 0
                for g in folds_t:
                    r1 = r0 + g.n_rows
                    g.spmm(cur, Y=nxt[r0:r1], add_in=gs[r0:r1], add_div=1.0)
                    r0 = r1
                cur = nxt
            return cur
        gE0 = PropagateMeanFolds._masked(folds_t, ctx.mask, run)
        u = ctx.n_user_rows
        return gE0[:u], gE0[u:], None, None, None, None


class ScoreBCELoss(torch.autograd.Function):
    """mean BCEWithLogits(<users[u], items[i]>, y) over the batch — model.py:115-120 — on the whole [N, d] table
    (users first, items after `n_user_rows`); backward yields the dense, mostly-zero table gradient that the
    propagation backward consumes, exactly like the reference's autograd."""

    @staticmethod
    def forward(ctx, light_out, n_user_rows, u_idx, i_idx, labels):
        B = u_idx.numel()
        loss_sum, grad = _score_bce_table_grad(light_out, n_user_rows, u_idx, i_idx, labels, ctx.needs_input_grad[0])
        ctx.grad = grad
        return (loss_sum / B).reshape(())

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        ctx.grad = None
        return grad * g, None, None, None, None


class LightGCNBCELoss(torch.autograd.Function):
    """PropagateMean followed by ScoreBCELoss as ONE autograd node — the training step of model.py:111-121 behind
    `forward(users, items, labels, flag=0)`: same kernels, one Function.apply and one backward node less per step (the
    step is host-bound under the reference's driver loop)."""

    @staticmethod
    def forward(ctx, user_w, item_w, graph, graph_t, n_layers, mask, u_idx, i_idx, labels):
        E0 = _flat_tables(user_w, item_w)
        if mask is not None:
            graph.set_edge_mask(*mask)
        try:
            light_out = graph.propagate(E0, n_layers)
        finally:
            if mask is not None:
                graph.set_edge_mask(0)
        n_u = user_w.shape[0]
        B = u_idx.numel()
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        loss_sum, grad = _score_bce_table_grad(light_out, n_u, u_idx, i_idx, labels, need)
        ctx.grad, ctx.graph_t, ctx.n_layers, ctx.mask, ctx.n_user_rows = grad, graph_t, n_layers, mask, n_u
        return (loss_sum / B).reshape(())

    @staticmethod
    def backward(ctx, g):
        grad, gt = ctx.grad, ctx.graph_t
        ctx.grad = None
        if ctx.mask is not None:
            gt.set_edge_mask(*ctx.mask)
        try:
            gE0 = gt.propagate_bwd(grad, ctx.n_layers)      # linear in `grad`: the upstream scalar is applied after
        finally:
            if ctx.mask is not None:
                gt.set_edge_mask(0)
        gE0 = gE0 * g
        u = ctx.n_user_rows
        return gE0[:u], gE0[u:], None, None, None, None, None, None, None


class BPRLoss(torch.autograd.Function):
    """mean softplus(<u,i-> - <u,i+>) (north-star extension; upstream LightGCN-PyTorch bpr_loss semantics)."""

    @staticmethod
    def forward(ctx, light_out, n_user_rows, u, i_pos, i_neg):
        users_tab, items_tab = light_out[:n_user_rows], light_out[n_user_rows:]
        T = u.numel()
        grad = torch.zeros_like(light_out) if ctx.needs_input_grad[0] else None
        loss_sum = bpr_loss_grad(users_tab, items_tab, u, i_pos, i_neg,
                                 grad[:n_user_rows] if grad is not None else None,
                                 grad[n_user_rows:] if grad is not None else None, 1.0 / T)
        ctx.grad = grad
        return (loss_sum / T).reshape(())

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        ctx.grad = None
        return grad * g, None, None, None, None


class _message_mask:
    """While open, layer `l`'s kernels take their message-dropout keep decisions from drop[3][l] (uint8 [N, 64], rows in the
    reference's numbering) instead of the counter-based draw: the spex_ngcf_message_mask validation hook, per host thread."""

    def __init__(self, drop, l):
        masks = drop[3] if drop is not None and len(drop) > 3 else None
        self.mask = masks[l] if masks is not None else None

    def __enter__(self):
        if self.mask is not None:
            _lib.call("spex_ngcf_message_mask", ctypes.c_void_p(self.mask.data_ptr()))

    def __exit__(self, *exc):
        if self.mask is not None:
            _lib.call("spex_ngcf_message_mask", None)
        return False


class NGCFPropagate(torch.autograd.Function):
    """The NGCF propagation (Model_Wrapper.forward, NGCF_SPEX/code/main_rec.py:71-86) on one [N, d] table: per layer one
    SpMM (side = A ego) and one fused layer kernel; returns the concatenated table [N, d (L + 1)].  Backward per layer:
    one fused layer-backward kernel (recomputes the layer on the matrix cores, weight gradients included) and one SpMM
    on A^T with the direct part added in its epilogue.  d == 64 for every layer.
    weights: flat tuple (W_gc_0, b_gc_0, W_bi_0, b_bi_0, W_gc_1, ...).  drop: None or (p_per_layer, seed, step[, masks]) — masks:
    None, or per layer a uint8 keep tensor that replaces the counter-based draw (NGCF.dropout_stream == "reference")."""

    @staticmethod
    def forward(ctx, user_w, item_w, graph, graph_t, drop, pad_row, *weights):
        E0 = _flat_tables(user_w, item_w)
        n, d = E0.shape
        L = len(weights) // 4
        out = torch.empty((n, d * (L + 1)), dtype=torch.float32, device=E0.device)
        egos, sides = [E0], []
        for l in range(L):
            W_gc, b_gc, W_bi, b_bi = (w.detach().contiguous() for w in weights[4 * l:4 * l + 4])
            side = graph.spmm(egos[l])
            nxt = torch.empty_like(E0) if l < L - 1 else None
            dl = None if drop is None or drop[0][l] <= 0 else (drop[0][l], drop[1], drop[2])
            with _message_mask(drop, l):
                ngcf_layer_fwd(egos[l], side, W_gc, b_gc, W_bi, b_bi, out, l, l == 0, nxt, drop=dl, pad_row=pad_row)
            sides.append(side)
            if nxt is not None:
                egos.append(nxt)
        if L == 0:
            out.copy_(E0)
        ctx.graph_t, ctx.drop, ctx.pad_row, ctx.n_user_rows, ctx.L = graph_t, drop, pad_row, user_w.shape[0], L
        ctx.save_for_backward(*egos, *sides, *weights)
        return out

    @staticmethod
    def backward(ctx, g_all):
        L = ctx.L
        saved = ctx.saved_tensors
        egos, sides, weights = saved[:max(L, 1)], saved[max(L, 1):max(L, 1) + L], saved[max(L, 1) + L:]
        g_all = g_all.contiguous()
        u = ctx.n_user_rows
        if L == 0:
            return (g_all[:u], g_all[u:], None, None, None, None)
        grads = [None] * (4 * L)
        g_next = None
        for l in range(L - 1, -1, -1):
            W_gc, b_gc, W_bi, b_bi = (w.detach().contiguous() for w in weights[4 * l:4 * l + 4])
            g_side, g_ego = torch.empty_like(egos[l]), torch.empty_like(egos[l])
            gW_gc, gb_gc, gW_bi, gb_bi = (torch.zeros_like(w) for w in (W_gc, b_gc, W_bi, b_bi))
            dl = None if ctx.drop is None or ctx.drop[0][l] <= 0 else (ctx.drop[0][l], ctx.drop[1], ctx.drop[2])
            with _message_mask(ctx.drop, l):
                ngcf_layer_bwd(egos[l], sides[l], W_gc, b_gc, W_bi, b_bi, g_all, l, g_next, g_side, g_ego, gW_gc, gb_gc, gW_bi,
                               gb_bi, drop=dl, pad_row=ctx.pad_row)
            g_next = ctx.graph_t.spmm(g_side, add_in=g_ego, add_div=1.0)      # d loss / d ego_l
            grads[4 * l:4 * l + 4] = [gW_gc, gb_gc, gW_bi, gb_bi]
        return (g_next[:u], g_next[u:], None, None, None, None, *grads)


# ------------------------------------------------------------------------------------------------ learned edge values
class EdgeSoftmax(torch.autograd.Function):
    """Row softmax over the stored entries of `graph` (tf.sparse.softmax on a fixed pattern,
    Diffnet++_SPEX/code/utility/Model.py:275-286).  v and the result are per-edge arrays indexed by edge id."""

    @staticmethod
    def forward(ctx, v, graph):
        y = graph.edge_softmax(v.contiguous())
        ctx.graph = graph
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return ctx.graph.edge_softmax_bwd(y, gy.contiguous()), None


class SpMMLearned(torch.autograd.Function):
    """Y = A(val) X with learned values on a fixed pattern (tf.sparse.sparse_dense_matmul, Model.py:18-83):
    forward = set_values + SpMM; backward: dX = A(val)^T dY on the transposed handle (same values array through its
    edge ids), dval = SDDMM(dY, X).  The values are re-installed before every launch, so several value sets may share
    one pair of handles."""

    @staticmethod
    def forward(ctx, val, X, graph, graph_t):
        val, X = val.contiguous(), X.contiguous()
        graph.set_values(val)
        Y = graph.spmm(X)
        ctx.graph, ctx.graph_t = graph, graph_t
        ctx.save_for_backward(val, X)
        return Y

    @staticmethod
    def backward(ctx, gY):
        val, X = ctx.saved_tensors
        gY = gY.contiguous()
        g_val = g_X = None
        if ctx.needs_input_grad[0]:
            g_val = ctx.graph.sddmm(gY, X)
        if ctx.needs_input_grad[1]:
            ctx.graph_t.set_values(val)
            g_X = ctx.graph_t.spmm(gY)
        return g_val, g_X, None, None


def edge_softmax(v, graph):
    return EdgeSoftmax.apply(v, graph)


def spmm_learned(val, X, graph, graph_t):
    return SpMMLearned.apply(val, X, graph, graph_t)


# ------------------------------------------------------------------------------------------------ trust-path attention
class PathAttention(torch.autograd.Function):
    """All heads of a GraphAttentionLayer (utility2/layers.py:15-71) in one launch, with the matching backward kernel.
    src: [rows, d] table gathered through `seq` ([B, L] int64) — or, with seq None, a dense [B, L, d] tensor.
    a: [n_heads, 2d].  Returns [B, L, n_heads * d]."""

    @staticmethod
    def forward(ctx, src, seq, seq_l, a, positional):
        src, a = src.contiguous(), a.contiguous()
        _need(src, "src"); _need(a, "a")
        dev = src.device
        seq_l = _idx(seq_l, dev)
        if seq is not None:
            seq = _idx(seq, dev)
            B, L = seq.shape
            d = src.shape[1]
            n_rows = src.shape[0]
        else:
            B, L, d = src.shape
            n_rows = B * L
        n_heads = a.shape[0]
        if a.shape[1] != 2 * d or seq_l.numel() != B:
            raise ValueError(f"path_attention: a {tuple(a.shape)} / seq_l {tuple(seq_l.shape)} do not fit B={B} d={d}")
        out = torch.empty((B, L, n_heads * d), dtype=torch.float32, device=dev)
        w0 = torch.empty((B, L, n_heads), dtype=torch.float32, device=dev)
        if B:
            _launch(src.device, "spex_path_attention_f32", _ptr(src), n_rows, _ptr(seq), _ptr(seq_l), _ptr(a), B, L, d, n_heads,
                      1 if positional else 0, _ptr(out), _ptr(w0))
        ctx.save_for_backward(src, seq, seq_l, a, w0)
        ctx.dims = (B, L, d, n_heads, n_rows, 1 if positional else 0)
        return out

    @staticmethod
    def backward(ctx, g):
        src, seq, seq_l, a, w0 = ctx.saved_tensors
        B, L, d, n_heads, n_rows, positional = ctx.dims
        g = g.contiguous()
        g_src = torch.zeros_like(src)
        g_a = torch.zeros_like(a) if ctx.needs_input_grad[3] else None
        if B:
            _launch(src.device, "spex_path_attention_bwd_f32", _ptr(src), n_rows, _ptr(seq), _ptr(seq_l), _ptr(a), B, L, d, n_heads,
                      positional, _ptr(w0), _ptr(g), _ptr(g_src), _ptr(g_a))
        return g_src, None, None, g_a, None


def path_attention(src, seq, seq_l, a, positional):
    return PathAttention.apply(src, seq, seq_l, a, positional)


# ------------------------------------------------------------------------------------------------ trust head (fused)
def trust_param_count(n_heads, d=64):
    return int(_lib.load().spex_trust_param_count(d, n_heads))


def trust_head_supported(d, L, n_heads):
    """The fused trust-head kernels cover hidden size 64, <= 16 path positions, <= 4 input heads."""
    return d == 64 and 1 <= L <= 16 and 1 <= n_heads <= 4


def trust_head_forward(table, params, seq, seq_l, n_heads, hybrid=True):
    """a2 [B, 64]: the readout vector whose product with the user table is the trust logits (model_expert_s.py:128-147 on
    top of :176-187), one launch.  params: the flat parameter block (layout in include/spex_hip.h)."""
    table, params = table.contiguous(), params.contiguous()
    _need(table, "table"); _need(params, "params")
    dev = table.device
    seq, seq_l = _idx(seq, dev), _idx(seq_l, dev)
    B, L = seq.shape
    a2 = torch.empty((B, 64), dtype=torch.float32, device=dev)
    if B:
        _launch(dev, "spex_trust_head_fwd_f32", _ptr(table), table.shape[0], _ptr(params), _ptr(seq), _ptr(seq_l), B, L,
                table.shape[1], n_heads, 1 if hybrid else 0, _ptr(a2))
    return a2


class TrustHeadLoss(torch.autograd.Function):
    """mean cross-entropy of the trust logits against `targets` — the whole trust branch of
    model_expert_s.LightGCN.forward (flag 0), forward and backward, in two launches (spex_trust_head_train_f32).  Like
    ScoreBCELoss the gradients are produced by the forward; backward() scales them by the upstream scalar."""

    @staticmethod
    def forward(ctx, table, params, seq, seq_l, targets, n_heads, hybrid):
        table, params = table.contiguous(), params.contiguous()
        _need(table, "table"); _need(params, "params")
        dev = table.device
        seq, seq_l, targets = _idx(seq, dev), _idx(seq_l, dev), _idx(targets, dev)
        B, L = seq.shape
        n_rows, d = table.shape
        if params.numel() != trust_param_count(n_heads, d):
            raise ValueError(f"trust head: parameter block of {params.numel()} floats, expected {trust_param_count(n_heads, d)}")
        if targets.numel() != B or seq_l.numel() != B:
            raise ValueError("trust head: seq_l / targets do not match the number of paths")
        n_ws = int(_lib.load().spex_trust_workspace_floats(B, L, d, n_heads, n_rows))
        n_users = n_rows - 1
        scratch = torch.empty(B * d + n_ws + B * n_users + B, dtype=torch.float32, device=dev)     # a2 | ws | dscore | loss_b
        a2, ws = scratch[: B * d], scratch[B * d: B * d + n_ws]
        dscore, loss_b = scratch[B * d + n_ws: B * d + n_ws + B * n_users], scratch[B * d + n_ws + B * n_users:]
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        g_table, g_params = torch.zeros_like(table), torch.zeros_like(params)
        if B:
            _launch(dev, "spex_trust_head_train_f32", _ptr(table), n_rows, _ptr(params), _ptr(seq), _ptr(seq_l), _ptr(targets), B, L, d,
                    n_heads, 1 if hybrid else 0, 1.0, None, _ptr(a2), _ptr(dscore), _ptr(loss_b), _ptr(ws), _ptr(loss), 0,
                    _ptr(g_params), _ptr(g_table))
        ctx.save_for_backward(g_table, g_params)
        return loss

    @staticmethod
    def backward(ctx, g):
        g_table, g_params = ctx.saved_tensors
        return g_table * g, g_params * g, None, None, None, None, None


class ExpertGate(torch.autograd.Function):
    """The two-expert gate of the dual-task model (model_expert_s.py:156-161) with its backward kernel:
    mixed = raw * a0 + prop * a1, a = softmax([raw | prop] att_exp)."""

    @staticmethod
    def forward(ctx, raw, prop, att_exp):
        raw, prop, att_exp = raw.contiguous(), prop.contiguous(), att_exp.contiguous()
        mixed = expert_gate(raw, prop, att_exp)
        ctx.save_for_backward(raw, prop, att_exp)
        return mixed

    @staticmethod
    def backward(ctx, g):
        raw, prop, att_exp = ctx.saved_tensors
        g = g.contiguous()
        g_raw, g_prop, g_att = torch.empty_like(raw), torch.empty_like(prop), torch.zeros_like(att_exp)
        if raw.shape[0] and DETERMINISTIC and raw.shape[1] <= 128:
            n_parts = int(_lib.load().spex_expert_gate_bwd_parts(raw.shape[0]))
            parts = torch.empty((n_parts, 4 * raw.shape[1]), dtype=torch.float32, device=raw.device)
            _launch(raw.device, "spex_expert_gate_bwd_det_f32", _ptr(raw), _ptr(prop), _ptr(att_exp), _ptr(g), _ptr(g_raw), _ptr(g_prop),
                    _ptr(g_att), _ptr(parts), raw.shape[0], raw.shape[1])
        elif raw.shape[0]:
            _launch(raw.device, "spex_expert_gate_bwd_f32", _ptr(raw), _ptr(prop), _ptr(att_exp), _ptr(g), _ptr(g_raw), _ptr(g_prop),
                      _ptr(g_att), raw.shape[0], raw.shape[1])
        return g_raw, g_prop, g_att


def expert_gate_rows(raw, prop, att_u, att_i, idx_a, idx_b, n_user_rows):
    """The two-expert gate at a batch's rows only: slot k names row idx_a[k] (k < len(idx_a)) or n_user_rows + idx_b[k - ..];
    rows below n_user_rows use att_u, the others att_i.  Returns the compact [len(idx_a) + len(idx_b), 64] gated rows."""
    _need(raw, "raw"); _need(prop, "prop"); _need(att_u, "att_u"); _need(att_i, "att_i")
    dev = raw.device
    idx_a, idx_b = _idx(idx_a, dev), _idx(idx_b, dev)
    out = torch.empty((idx_a.numel() + idx_b.numel(), raw.shape[1]), dtype=torch.float32, device=dev)
    _launch(dev, "spex_expert_gate_rows_f32", _ptr(raw), _ptr(prop), _ptr(att_u), _ptr(att_i), _ptr(idx_a), idx_a.numel(), 0, _ptr(idx_b),
            idx_b.numel(), n_user_rows, n_user_rows, raw.shape[0], raw.shape[1], _ptr(out))
    return out


def expert_gate_rows_bwd(raw, prop, att_u, att_i, idx_a, idx_b, n_user_rows, grad_slots, grad_prop, grad_raw, grad_att_u, grad_att_i):
    """Backward of expert_gate_rows slot by slot: returns the compact d prop rows; grad_prop / grad_raw ([N, 64]) and the two gate
    matrices' gradients are accumulated (atomics)."""
    dev = raw.device
    idx_a, idx_b = _idx(idx_a, dev), _idx(idx_b, dev)
    _need(grad_slots, "grad_slots")
    out = torch.empty((idx_a.numel() + idx_b.numel(), raw.shape[1]), dtype=torch.float32, device=dev)
    _launch(dev, "spex_expert_gate_rows_bwd_f32", _ptr(raw), _ptr(prop), _ptr(att_u), _ptr(att_i), _ptr(idx_a), idx_a.numel(), 0,
            _ptr(idx_b), idx_b.numel(), n_user_rows, n_user_rows, raw.shape[0], raw.shape[1], _ptr(grad_slots), grad_slots.stride(0),
            _ptr(out), _ptr(grad_prop), _ptr(grad_raw), _ptr(grad_att_u), _ptr(grad_att_i))
    _bump(grad_prop, grad_raw, grad_att_u, grad_att_i)
    return out


def expert_gate_autograd(raw, prop, att_exp):
    return ExpertGate.apply(raw, prop, att_exp)


# ------------------------------------------------------------------------------------------------ Diffnet++ node fusion
class AttnFuse(torch.autograd.Function):
    """Node-level attention fusion of a Diffnet++ layer (Model.py:308-345) as one kernel forward, one backward.
    U: [n, d] or None; X1, X2: [n, d]; p1, p2: per-branch parameter blocks [w1 | b1 | w2 | b2]."""

    @staticmethod
    def forward(ctx, U, X1, X2, p1, p2, c1, c2, base_coef, mix_coef):
        X1, X2, p1, p2 = X1.contiguous(), X2.contiguous(), p1.contiguous(), p2.contiguous()
        U = U.contiguous() if U is not None else None
        for t, nm in ((U, "U"), (X1, "X1"), (X2, "X2"), (p1, "p1"), (p2, "p2")):
            _need(t, nm)
        n, d = X1.shape
        if p1.numel() != (d if U is not None else 0) + d + 3 or p2.numel() != p1.numel():
            raise ValueError("attn_fuse: parameter blocks must hold w1 | b1 | w2 | b2")
        out = torch.empty_like(X1)
        if n:
            _launch(X1.device, "spex_attn_fuse_f32", _ptr(U), _ptr(X1), _ptr(X2), _ptr(p1), _ptr(p2), n, d, float(c1), float(c2),
                      float(base_coef), float(mix_coef), _ptr(out))
        ctx.save_for_backward(U, X1, X2, p1, p2)
        ctx.consts = (float(c1), float(c2), float(base_coef), float(mix_coef))
        return out

    @staticmethod
    def backward(ctx, g):
        U, X1, X2, p1, p2 = ctx.saved_tensors
        g = g.contiguous()
        n, d = X1.shape
        gU = torch.empty_like(U) if U is not None else None
        gX1, gX2 = torch.empty_like(X1), torch.empty_like(X2)
        gp1, gp2 = torch.zeros_like(p1), torch.zeros_like(p2)
        if n:
            _launch(X1.device, "spex_attn_fuse_bwd_f32", _ptr(U), _ptr(X1), _ptr(X2), _ptr(p1), _ptr(p2), n, d, *ctx.consts, _ptr(g),
                      _ptr(gU), _ptr(gX1), _ptr(gX2), _ptr(gp1), _ptr(gp2))
        return gU, gX1, gX2, gp1, gp2, None, None, None, None


def attn_fuse(U, X1, X2, p1, p2, c1, c2, base_coef, mix_coef):
    return AttnFuse.apply(U, X1, X2, p1, p2, c1, c2, base_coef, mix_coef)


# ------------------------------------------------------------------------------------------------ multi-GPU row exchange
def gather_owned_rows(table, pos, lo, out):
    """out[k] = table[pos[k] - lo] where this rank owns the position, 0 elsewhere (spex_gather_owned_rows_f32)."""
    _need(table, "table"); _need(out, "out")
    pos = _idx(pos, out.device)
    if out.shape != (pos.numel(), table.shape[1]):
        raise ValueError("gather_owned_rows: out must be [len(pos), d]")
    _launch(table.device, "spex_gather_owned_rows_f32", _ptr(table), _ptr(pos), pos.numel(), int(lo), table.shape[0], table.shape[1],
              _ptr(out))
    _bump(out)
    return out


def spmm_owned_rows(graph, X, pos, lo, acc_in, acc_div, out_prop, raw=None, out_raw=None, acc2=None, acc3=None):
    """The partitioned steps' last forward layer at a batch's rows (spex_spmm_owned_rows_f32; utility1/model.py:91-97 at the rows of
    :115-116): for the positions this rank owns, out_prop[k] = (acc_in[r] [+ acc2[r] [+ acc3[r]]] + (A X)[r]) / acc_div with
    r = pos[k] - lo and out_raw[k] = raw[r]; ZERO rows elsewhere (the operands of the owner-computes all-reduce).  `graph`: the
    rank's row block."""
    _need(X, "X"); _need(acc_in, "acc_in"); _need(out_prop, "out_prop")
    for t, name in ((raw, "raw"), (out_raw, "out_raw"), (acc2, "acc2"), (acc3, "acc3")):
        if t is not None:
            _need(t, name)
    pos = _idx(pos, out_prop.device)
    d = X.shape[1]
    if out_prop.shape != (pos.numel(), d) or (out_raw is not None and out_raw.shape != out_prop.shape):
        raise ValueError("spmm_owned_rows: out_prop / out_raw must be [len(pos), d]")
    if X.shape[0] != graph.n_cols or any(t is not None and t.shape != (graph.n_rows, d) for t in (acc_in, acc2, acc3, raw)):
        raise ValueError("spmm_owned_rows: X must be [n_cols, d], the row operands [n_rows, d]")
    _launch(X.device, "spex_spmm_owned_rows_f32", graph._h, _ptr(X), _ptr(pos), pos.numel(), int(lo), _ptr(acc_in),
            None if acc2 is None else _ptr(acc2), None if acc3 is None else _ptr(acc3), float(acc_div), None if raw is None else _ptr(raw),
            _ptr(out_prop), None if out_raw is None else _ptr(out_raw), d)
    _bump(out_prop, *([] if out_raw is None else [out_raw]))
    return out_prop


def scatter_add_owned_rows(upd, pos, lo, table, clear=True):
    """table[pos[k] - lo] += upd[k] for the positions this rank owns; upd is cleared afterwards (clear=True)."""
    _need(table, "table"); _need(upd, "upd")
    pos = _idx(pos, upd.device)
    if upd.shape != (pos.numel(), table.shape[1]):
        raise ValueError("scatter_add_owned_rows: upd must be [len(pos), d]")
    _launch(upd.device, "spex_scatter_add_owned_rows_f32", _ptr(upd), _ptr(pos), pos.numel(), int(lo), table.shape[0], table.shape[1],
              _ptr(table), 1 if clear else 0)
    _bump(table, upd)
    return table
