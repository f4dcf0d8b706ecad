/*
 * spex_hip.h — C ABI of libspexhip.so: the MI355X (gfx950) implementation of the SPEX LightGCN / NGCF graph
 * convolution + negative-sampled scoring hot path.
 *
 * The reference (XMUDM/SPEX) is pure Python on stock PyTorch: it has no FFI / plugin layer of its own, so there is no
 * existing binding to match symbol-for-symbol.  Each entry point below therefore names the reference call it
 * replaces (file:line relative to the reference root) — these are the calls a maintainer re-points at this
 * library (INTEGRATION.md shows the ctypes stub and the drop-in Python modules that already do so).
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - every dense pointer is a DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr()); the library never
 *     frees or retains them past the call.  Row-major fp32, leading dimension == d, 16-byte aligned.
 *   - host pointers are marked `h_`.
 *   - every call is asynchronous on `stream` unless stated; no hidden device synchronisation and no allocation after
 *     spex_graph_create (safe to capture in a hipGraph) — with three documented exceptions that allocate once: the
 *     first spex_sddmm_f32 on a handle, the first spex_spmm_f32 / propagate call at d > 64 on a graph with long rows,
 *     and spex_timer_create.
 *   - return 0 on success, a negative spex_status otherwise; spex_last_error() gives the thread-local message.
 */
#ifndef SPEX_HIP_H
#define SPEX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum spex_status {
    SPEX_OK = 0,
    SPEX_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, unsorted CSR, ...) */
    SPEX_ERR_HIP = -2,         /* a HIP runtime call failed                                      */
    SPEX_ERR_COMM = -3,        /* an RCCL call of the spex_comm_* entries failed                   */
    SPEX_ERR_UNSUPPORTED = -4  /* e.g. the NGCF epilogue at d != 64                              */
} spex_status;

typedef struct spex_graph spex_graph_t;

int spex_version(void);                 /* ABI version, currently 5.  5 = 4 minus what round 3's measurements retired
                                         * (spex_graph_create_ex / SPEX_GRAPH_TILE_ROWS and spex_ngcf_spmm_layer_fwd_f32, the `flags`
                                         * argument of spex_graph_pack_digest, the tiled trust head, the one-wave NGCF kernels, the
                                         * narrow d <= 32 SpMM), plus spex_partitioned_dual_task_step_f32 (config 5 row-partitioned
                                         * as one call).  4 = 3 + the deterministic accumulation mode, per-descriptor fork / join
                                         * events, the spex_comm_* collectives, the one-launch middles, SPEX_STEP_PIPELINED. */
const char *spex_last_error(void);      /* thread-local, never NULL */

/* ------------------------------------------------------------------------------------------------ graph handle
 * Replaces: the torch sparse tensor the reference builds once and keeps on the device —
 *   LightGCN_SPEX/code/utility1/dataloader.py:179-185,221-222 (_convert_sp_mat_to_sp_tensor + coalesce + .cuda()),
 *   NGCF_SPEX/code/main_rec.py:47,102-108 (sparse_mx_to_torch_sparse_tensor).
 * Input is host CSR (rows sorted, columns ascending within a row = the coalesced COO order), copied to HBM once.
 * `h_edge_id` (optional, may be NULL = identity) gives, per stored entry, the index used to look the entry up in an
 * edge keep-mask (see spex_graph_set_edge_mask): a transposed copy of a graph passes the permutation back to the
 * original entry order so forward and backward drop the same edges.
 * The handle may hold any row block of a larger matrix (1-D row partition): n_rows local rows, column indices
 * global in [0, n_cols).
 * Synchronous (host -> device copies); builds the long-row segment table used for load balance.
 */
int spex_graph_create(const int32_t *h_rowptr, const int32_t *h_col, const float *h_val, const int32_t *h_edge_id,
                      int32_t n_rows, int32_t n_cols, int64_t nnz, spex_graph_t **out);
int spex_graph_destroy(spex_graph_t *g);
/* HOST ONLY (no device call, works without a GPU): FNV-1a fingerprints of everything spex_graph_create would upload for this
 * matrix — digest[0] the wave-task table, [1..5] the chunk arrays (source offsets, values, end-of-row masks + padding counts, edge
 * ids, output rows), [6] the long-row segment tables, [7] the hub tables and the table sizes.  The
 * host packer runs on SPEX_BUILD_THREADS threads (default min(16, cores)); the layout must not depend on that number. */
int spex_graph_pack_digest(const int32_t *h_rowptr, const int32_t *h_col, const float *h_val, int32_t n_rows, int32_t n_cols,
                           int64_t nnz, uint64_t *digest /* [8] */);
/* HOST ONLY: how the packer laid out the rows beyond 1 024 entries (hubs) — the tables the d == 64 launch folds them by.  out:
 * [n_positions, n_hubs, then per task position of the table's head (chunks, row, task word, 1 = first wave of a group, waves in
 * the group, the group's scratch row, hub), then per hub (its first scratch row, its number of groups)]; *n_out = ints needed
 * (call with out = NULL, cap = 0 to size).  Invariants: every hub starts a 16-task workgroup; its groups are 16 adjacent segments
 * (the last one shorter, the workgroup topped up with ordinary tasks); a hub's scratch rows are consecutive. */
int spex_graph_pack_hub_table(const int32_t *h_rowptr, const int32_t *h_col, const float *h_val, int32_t n_rows, int32_t n_cols,
                              int64_t nnz, int32_t *out, int64_t cap, int64_t *n_out);
/* n_rows, n_cols, nnz, number of long rows, number of long-row segments (any pointer may be NULL) */
int spex_graph_info(const spex_graph_t *g, int32_t *n_rows, int32_t *n_cols, int64_t *nnz, int32_t *n_long_rows,
                    int32_t *n_segments);

/* Edge dropout — replaces LightGCN.__dropout_x, utility1/model.py:46-55: keep entry e iff floor(rand_e + keep_prob),
 * kept values divided by keep_prob.  Two modes:
 *   injected : d_keep = device uint8[mask_len], indexed by edge_id (parity tests inject the reference's mask)
 *   sampled  : d_keep = NULL; the mask is counter-based: keep_e = (u + keep_prob >= 1) with u the top 24 bits of
 *              philox4x32-10(key = seed, counter = edge_id) scaled to [0,1) (torch.rand's grid), recomputed inside
 *              the SpMM, identical in forward / backward and across layers for one seed.
 * keep_prob >= 1 or mode 0 switches dropout off.  Takes effect for subsequent launches on this handle.
 */
int spex_graph_set_edge_mask(spex_graph_t *g, int mode /*0 off, 1 injected, 2 sampled*/, const uint8_t *d_keep,
                             float keep_prob, uint64_t seed);

/* ------------------------------------------------------------------------------------------------ SpMM
 * Replaces: torch.sparse.mm(Graph, all_emb) — utility1/model.py:91 (and :87 for the A_split row blocks),
 * NGCF_SPEX/code/main_rec.py:76; and its autograd backward (A^T g) when called on the transposed handle.
 *
 *   y[r,:]       = sum_e val[e] * X[col[e],:]               (fp32 fmaf chain, ascending column order)
 *   if add_in:     y += add_in[r,:] / add_div               (backward of the layer mean: g/(L+1) + A^T G)
 *   if Y:          Y[r,:] = y
 *   if acc_out:    acc_out[r,:] = (acc_in[r,:] + y) / acc_div   (running layer sum; acc_div = L+1 on the last layer)
 *
 * X: [n_cols, d]; Y, add_in, acc_in, acc_out: [n_rows, d].  acc_in may equal acc_out.  X must not alias Y / acc_out.
 * Any d >= 1 (d == 64 takes the tuned path: one lane per column).  A graph with long rows sums their segments through a
 * scratch buffer owned by the handle (rows beyond 1 024 entries: groups of 16 segments through LDS, one scratch row per group,
 * folded by the row's last group inside the d == 64 launch — in a fixed order: results repeat bit for bit; SPEX_HUB_FOLD=0,
 * edge-dropout launches and other d: a small second launch adds them): calls on ONE stream are ordered by the stream; a call on another stream is made to
 * wait (an event, inserted by the library) for everything queued on the stream that used the scratch last, so one handle
 * may be driven from several streams of one host thread.  Concurrent calls from several HOST threads on one handle are
 * not supported (one handle per thread).
 */
int spex_spmm_f32(const spex_graph_t *g, const float *X, float *Y, const float *add_in, float add_div,
                  const float *acc_in, float *acc_out, float acc_div, int32_t d, void *stream);

/* The same product for a LIST of rows only (d == 64, no edge dropout):
 *   for r in { idx_a[k] + off_a : k < n_a } + { idx_b[k] + off_b : k < n_b }:
 *     y = sum_e val[e] X[col[e],:];   Y[r,:] = y (if Y);   acc_out[r,:] = (acc_in[r,:] + y) / acc_div (if acc_out)
 * Other rows of Y / acc_out are left untouched.  idx_*: device int64 (a batch's users and items as the DataLoader
 * hands them over; off_b = n_user_rows).  Duplicates allowed — every copy stores the same value — PROVIDED acc_out
 * is not acc_in (in place, a second copy could read the row the first has already finished); an index outside
 * [0, n_rows) is skipped.
 * The training step reads the last propagation layer (model.py:91-95) at the batch's rows only (model.py:115-116):
 * this replaces that layer's launch over the whole matrix.  Rows of up to 1024 entries are summed in the order
 * spex_spmm_f32 uses (bit-identical results).
 */
int spex_spmm_rowlist_f32(const spex_graph_t *g, const float *X, const int64_t *idx_a, int32_t n_a, int64_t off_a,
                          const int64_t *idx_b, int32_t n_b, int64_t off_b, float *Y, const float *acc_in, float *acc_out,
                          float acc_div, int32_t d, void *stream);
/* The same rows on a ROW PARTITION (d == 64; the partitioned steps' last forward layer, model.py:91-97 at the rows of model.py:115-116):
 * slot k names position pos[k] of the padded global layout; `g` is the rank's block (n_local rows) and lo the rank's first position.
 * For a slot the rank owns (r = pos[k] - lo in [0, n_local)):
 *   out_prop[k] = (acc_in[r] [+ acc2[r] [+ acc3[r]]] + (A X)[r]) / acc_div  (the row-list kernel's order of sums; acc2 / acc3: further
 *                 layer tables, NULL = absent),   out_raw[k] = raw[r]  (if out_raw)
 * and ZEROS for every other slot: out_prop / out_raw ([n, 64], compact) are the operands of the owner-computes all-reduce
 * (spex_comm_allreduce_sum_f32) that leaves the batch's rows on every rank.  One launch instead of a whole-block product + two
 * spex_gather_owned_rows_f32.  A mask on the handle (edge dropout) is applied entry by entry, as spex_spmm_f32 applies it. */
int spex_spmm_owned_rows_f32(const spex_graph_t *g, const float *X, const int64_t *pos, int32_t n, int64_t lo, const float *acc_in,
                             const float *acc2, const float *acc3, float acc_div, const float *raw, float *out_prop, float *out_raw,
                             int32_t d, void *stream);

/* Whole LightGCN.computer(), utility1/model.py:66-97, for a graph held entirely on this device (n_rows == n_cols):
 *   E^{l+1} = A E^l (l < L);  mean_out = (E^0 + ... + E^L) / (L+1).
 * ws: caller-provided device workspace of 2 * n_rows * d floats (ping-pong layer buffers).
 * layers_out (optional): L * n_rows * d floats receiving E^1..E^L (then ws may be NULL).
 */
int spex_propagate_f32(const spex_graph_t *g, const float *E0, float *mean_out, float *layers_out, float *ws,
                       int32_t L, int32_t d, void *stream);

/* Backward of the above w.r.t. E0 given g_out = d loss / d mean_out  (autograd of model.py:83-95):
 *   G_L = g_out/(L+1);  G_l = g_out/(L+1) + A^T G_{l+1};  grad_E0 = G_0.
 * gt is the handle of A^T (for the symmetric LightGCN adjacency the same handle).  ws: 3 * n_rows * d floats
 * (the scaled gradient g_out/(L+1) plus two ping-pong layer buffers).
 */
int spex_propagate_bwd_f32(const spex_graph_t *gt, const float *g_out, float *grad_E0, float *ws, int32_t L,
                           int32_t d, void *stream);

/* ------------------------------------------------------------------------------------------------ scoring
 * Replaces LightGCN.forward, utility1/model.py:111-121 (gather rows, elementwise product, sum, BCEWithLogitsLoss)
 * and NGCF compute_rec_loss, NGCF_SPEX/code/main_rec.py:89-91,96-100 — plus their autograd backward into the two
 * gathered tables.
 *   gamma[b] = <users[u_idx[b]], items[i_idx[b]]>
 *   labels != NULL: loss_sum += sum_b BCEWithLogits(gamma[b], labels[b])     (caller divides by B; *loss_sum must be
 *                   zeroed by the caller, it is accumulated with atomics)
 *   grad_users/grad_items != NULL: grad_users[u_idx[b],:] += dg_b * items[i_idx[b],:], grad_items likewise, with
 *                   dg_b = (sigmoid(gamma[b]) - labels[b]) * grad_scale   (grad_scale = upstream_grad / B).
 *                   The grad tables are accumulated into (atomics): zero them first.
 * ldu / ldi: row strides (floats) of users / items and of their grad tables (d for LightGCN, 2d.. for NGCF concat).
 * u_idx, i_idx: device int64 (what torch DataLoader hands over, main_rec.py:33-34).
 * n_user_rows / n_item_rows: table heights; a sample whose index falls outside is skipped (gamma = NaN) instead of
 * gathering out of bounds.
 */
int spex_score_bce_f32(const float *users, const float *items, int32_t ldu, int32_t ldi, int64_t n_user_rows,
                       int64_t n_item_rows, const int64_t *u_idx, const int64_t *i_idx, const float *labels, int32_t B,
                       int32_t d, float *gamma, float *loss_sum, float *grad_users, float *grad_items,
                       float grad_scale, void *stream);

/* The same scoring with the gradient ALSO (or only) as per-sample rows: grad_slots [2B, ld_slots], row b = d loss / d
 * users[u_idx[b]] of sample b alone (= dg_b * items[i_idx[b]]), row B + b = d loss / d items[i_idx[b]] of sample b — plain
 * stores, no atomics.  These are the operands of the row-sparse backward (spex_spmm_push_batch_f32,
 * spex_ngcf_layer_bwd_rows_f32).  grad_users / grad_items may be NULL (no table form wanted).
 */
int spex_score_bce_slots_f32(const float *users, const float *items, int32_t ldu, int32_t ldi, int64_t n_user_rows,
                             int64_t n_item_rows, const int64_t *u_idx, const int64_t *i_idx, const float *labels, int32_t B,
                             int32_t d, float *loss_sum, float *grad_users, float *grad_items, float grad_scale,
                             float *grad_slots, int32_t ld_slots, void *stream);

/* North-star extension (no counterpart in the reference): BPR over (u, i+, i-) triples.
 *   x_t = <U_read[u], I_read[i-] - I_read[i+]>;  loss_sum += softplus(x_t)
 * Fused gather + dot + sigmoid + SGD: with s_t = sigmoid(x_t)/T,
 *   U_w[u]  -= lr * (s_t (i- - i+) + reg u / T);  I_w[i+] -= lr * (-s_t u + reg i+ / T);  I_w[i-] -= lr * (s_t u + reg i- / T)
 * Batch-synchronous when the read tables differ from the write tables; passing the same tables gives in-place
 * (hogwild) matrix-factorisation BPR.  Updates use float atomics (duplicates in a batch accumulate).
 */
int spex_bpr_sgd_step_f32(const float *U_read, const float *I_read, float *U_w, float *I_w, int64_t n_user_rows,
                          int64_t n_item_rows, const int64_t *u, const int64_t *i_pos, const int64_t *i_neg, int64_t T,
                          int32_t d, float lr, float reg, float *loss_sum, void *stream);

/* The same triples scored for autograd instead of a fused update (bpr_loss() of the drop-in model):
 *   loss_sum += sum_t softplus(x_t);  grad_users[u] += grad_scale * sigmoid(x_t) (i- - i+);
 *   grad_items[i+] -= grad_scale * sigmoid(x_t) u;  grad_items[i-] += grad_scale * sigmoid(x_t) u.
 * grad tables may be NULL (loss only).  Accumulates with atomics: zero loss_sum / grad tables first.
 */
int spex_bpr_loss_f32(const float *users, const float *items, int64_t n_user_rows, int64_t n_item_rows,
                      const int64_t *u, const int64_t *i_pos, const int64_t *i_neg, int64_t T, int32_t d,
                      float *loss_sum, float *grad_users, float *grad_items, float grad_scale, void *stream);

/* The two BPR entry points above without one float atomic per gathered row, for large batches (d == 64).  The batch's
 * 3 T row updates are binned on the device by destination row (buckets of 64 rows of the joint [users | items] index
 * space, two-phase counting partition), accumulated per bucket in LDS and flushed to the tables once per touched row —
 * the arithmetic and the result (up to fp32 re-association of a row's contributions) are those of spex_bpr_sgd_step_f32 /
 * spex_bpr_loss_f32, which stay the right call below ~16 k triples (one launch, latency-bound).
 * Batch-synchronous only: the read tables must differ from the updated ones.
 * ws: caller-owned device scratch, 256-byte aligned, at least spex_bpr_grouped_workspace_bytes(T, n_user_rows,
 * n_item_rows) bytes; that function returns 0 when the grouped form does not apply (more than 8192 * 64 rows, T >= 2^30):
 * use the atomic form then.  Four launches on `stream`, no host synchronisation, no allocation.
 */
int64_t spex_bpr_grouped_workspace_bytes(int64_t T, int64_t n_user_rows, int64_t n_item_rows);
int spex_bpr_sgd_step_grouped_f32(const float *U_read, const float *I_read, float *U_w, float *I_w, int64_t n_user_rows,
                                  int64_t n_item_rows, const int64_t *u, const int64_t *i_pos, const int64_t *i_neg,
                                  int64_t T, int32_t d, float lr, float reg, float *loss_sum, void *ws, int64_t ws_bytes,
                                  void *stream);
int spex_bpr_loss_grouped_f32(const float *users, const float *items, int64_t n_user_rows, int64_t n_item_rows,
                              const int64_t *u, const int64_t *i_pos, const int64_t *i_neg, int64_t T, int32_t d,
                              float *loss_sum, float *grad_users, float *grad_items, float grad_scale, void *ws,
                              int64_t ws_bytes, void *stream);

/* Owner-computes exchange of a batch's rows on a 1-D row-partitioned table ("replicate the batch on all
 * ranks (owner-computes, no comm)" needs the batch's rows of the propagated table everywhere).  pos: device int64[K]
 * positions in the padded global row layout; this rank owns positions [lo, lo + n_local) = rows of `table`.
 *   gather : out[k,:] = owned(pos[k]) ? table[pos[k]-lo,:] : 0      -> one all-reduce (sum) of `out` completes it
 *   scatter: table[pos[k]-lo,:] += upd[k,:] for the owned k (atomics: positions repeat), then upd[k,:] = 0 if clear_upd
 */
int spex_gather_owned_rows_f32(const float *table, const int64_t *pos, int64_t K, int64_t lo, int64_t n_local, int32_t d,
                               float *out, void *stream);
int spex_scatter_add_owned_rows_f32(float *upd, const int64_t *pos, int64_t K, int64_t lo, int64_t n_local, int32_t d,
                                    float *table, int32_t clear_upd, void *stream);

/* ------------------------------------------------------------------------------------------------ optimiser
 * Replaces torch.optim.Adam(...).step() over the dense embedding tables — LightGCN_SPEX/code/main_rec.py:23,37.
 * One fused pass: m, v, p updated in place (bias-corrected, eps outside the sqrt as torch does), t = step count >= 1.
 * zero_buf (optional, n floats; may be g itself, may not alias p, m or v): cleared in the same pass — the caller's
 * gradient accumulation table for the next step (saves a separate fill launch per step).
 */
int spex_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n, int32_t t, float lr, float beta1,
                       float beta2, float eps, float *zero_buf, void *stream);

/* The same update for a small parameter block whose gradient arrives as n_parts partial blocks (block j at g_parts +
 * j * part_stride): g = the blocks added up in order, then Adam.  Used for NGCF's layer weights, whose gradient leaves
 * spex_ngcf_layer_bwd_rows_f32 as one block per workgroup.
 */
int spex_adam_step_sum_f32(float *p, const float *g_parts, int32_t n_parts, int64_t part_stride, float *m, float *v, int64_t n,
                           int32_t t, float lr, float beta1, float beta2, float eps, void *stream);

/* Validation hook for the NGCF entries' message dropout: while d_keep != NULL (per host thread), every NGCF layer entry called with
 * p_drop > 0 keeps element (row, col) iff d_keep[row' * 64 + col] != 0 (row' = the row in the reference's numbering, see pad_row)
 * instead of the counter-based draw — so that a run can use the REFERENCE's own nn.Dropout noise (main_rec.py:81:
 * `torch.empty(N, 64).bernoulli_(1 - p)` from the global CPU generator, drawn by the host where the reference draws it and uploaded
 * as bytes).  The pointer is read when a launch is queued; NULL restores the counter-based masks. */
int spex_ngcf_message_mask(const uint8_t *d_keep);

/* ------------------------------------------------------------------------------------------------ NGCF layer
 * Replaces NGCF_SPEX/code/main_rec.py:77-83 for one layer, given side = A ego from spex_spmm_f32 (:76):
 *   s    = LeakyReLU(side W_gc^T + b_gc);  b = LeakyReLU((ego * side) W_bi^T + b_bi);  e1 = dropout_p(s + b)
 *   out[r, 0:d] = ego[r,:] (iff write_ego);  out[r, d:2d] = e1 / max(||e1||_2, 1e-12)
 * W_*: [d,d] row-major as nn.Linear stores them (out x in); d == 64.  `out` has row stride ld_out >= 2d: layer l of a
 * deeper model passes out + l*d with write_ego = 0 and lands in its slice of the concatenated table (:85).
 * e1_out (optional, [n,d]) receives e1 after dropout — the next layer's input (:80-81).
 * Message dropout (:81, nn.Dropout(p_drop), training only; p_drop = 0 switches it off) is counter-based so that the
 * backward call reproduces the mask from (seed, step, layer) without storing it: element e = row*64 + col keeps iff
 * u_e >= p_drop, u_e = (word[e & 3] of philox4x32-10(counter = (e >> 2, step, layer, 0), key = seed) >> 8) * 2^-24; kept
 * values are multiplied by 1/(1 - p_drop).  pad_row >= 0: rows above it count one less when numbering elements (a table
 * that keeps the reference's unused pad user row, main_rec.py:67, as an isolated node; -1 = none).
 */
int spex_ngcf_layer_fwd_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc, const float *W_bi,
                            const float *b_bi, float *out, int32_t ld_out, int32_t write_ego, float *e1_out, int32_t n,
                            int32_t d, float slope, float p_drop, uint64_t seed, uint32_t step, uint32_t layer,
                            int32_t pad_row, void *stream);
/* The same layer at a LIST of rows only: slot k names row idx_a[k] + off_a (k < n_a) or idx_b[k - n_a] + off_b.  `ego`, `side`
 * and `out` are the dense tables, read / written at those rows (other rows of `out` are left untouched; a row named twice is
 * stored twice with the same value; an index outside [0, n) is skipped); the dropout mask is indexed by the row.  A one-layer
 * model's training loss reads the layer's output at the batch's rows only (main_rec.py:96-104): with spex_spmm_rowlist_f32 for
 * `side` this replaces the whole-table forward of a training step (2B rows instead of N).  Bit-identical rows.
 */
int spex_ngcf_layer_fwd_rows_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc, const float *W_bi,
                                 const float *b_bi, float *out, int32_t ld_out, int32_t write_ego, int32_t n, int32_t d, float slope,
                                 float p_drop, uint64_t seed, uint32_t step, uint32_t layer, int32_t pad_row, const int64_t *idx_a,
                                 int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b, int64_t off_b, void *stream);
/* The same layer in inference form (dropout off, ego written): kept for ABI-1 callers. */
int spex_ngcf_layer_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc, const float *W_bi,
                        const float *b_bi, float *out, int32_t ld_out, float *e1_out, int32_t n, int32_t d,
                        float slope, void *stream);
/* Autograd backward of the layer (loss.backward(), main_rec.py:127).  The layer is recomputed from (ego, side), nothing
 * else of the forward is kept.  Upstream: g_norm = d loss / d out[:, d:2d] (row stride ld_g), g_next = d loss / d e1
 * ([n,d], from the next layer's call; NULL for the last layer), g_direct = d loss / d out[:, 0:d] (row stride
 * ld_direct; layer 0 only, else NULL).  Writes, for every row,
 *   g_side = d loss / d side   and   g_ego = the part of d loss / d ego that does not pass through side (+ g_direct);
 * the caller completes d loss / d ego = g_ego + A^T g_side with spex_spmm_f32(A^T, g_side, add_in = g_ego).
 * gW_gc, gb_gc, gW_bi, gb_bi ([d,d], [d]) are ACCUMULATED (zero them first) — since ABI 5 without a float atomic: every workgroup
 * leaves its share as one block of a per-(device, stream) scratch the library keeps, and a second small launch adds the blocks to
 * the four arrays in block order (deterministic; plain read-modify-write: two calls that accumulate into the SAME arrays must be
 * ordered, e.g. on one stream).  The scratch is allocated on the first call at a size (not inside a stream capture).
 * Rows whose upstream gradients are all zero (after a 256-sample batch: most) cost a read and two zero rows.
 */
int spex_ngcf_layer_bwd_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc, const float *W_bi,
                            const float *b_bi, const float *g_norm, int32_t ld_g, const float *g_next,
                            const float *g_direct, int32_t ld_direct, int32_t n, int32_t d, float slope, float p_drop,
                            uint64_t seed, uint32_t step, uint32_t layer, int32_t pad_row, float *g_side, float *g_ego,
                            float *gW_gc, float *gb_gc, float *gW_bi, float *gb_bi, void *stream);

/* The same backward for the rows of a BATCH only — the form the LAST layer takes in training: after a B-sample batch
 * only the batch's <= 2B rows carry a gradient behind it (main_rec.py:89-90), so 2B / 16 tiles replace n / 16.
 * The batch is given as it is scored: slot k < n_a is row idx_a[k] + off_a, slot n_a + k is row idx_b[k] + off_b (device
 * int64).  EVERY slot is processed with its own upstream gradient rows — g_norm_c / g_direct_c are COMPACT, row k = slot k,
 * exactly what spex_score_bce_slots_f32 writes (the layer's backward is linear in them: a row named by several slots
 * simply gets several contributions downstream; nothing is deduplicated and no gradient table is involved).
 * g_side_c / g_ego_c: COMPACT outputs [n_a + n_b, d] — the operands of spex_spmm_push_batch_f32, which completes
 * d loss / d ego = scatter(g_ego_c) + A^T scatter(g_side_c).
 * Weight gradients leave as spex_ngcf_layer_bwd_rows_parts(n_a + n_b) partial blocks (one per 16-slot tile), block j at
 * gW_parts + j * part_stride, each laid out [dW_gc d*d | db_gc d | dW_bi d*d | db_bi d] (plain stores): add them up —
 * spex_adam_step_sum_f32 does, inside the optimiser pass.
 */
int32_t spex_ngcf_layer_bwd_rows_parts(int32_t n_slots);
int spex_ngcf_layer_bwd_rows_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc,
                                 const float *W_bi, const float *b_bi, const float *g_norm_c, int32_t ld_g,
                                 const float *g_next, const float *g_direct_c, int32_t ld_direct, int32_t n, int32_t d,
                                 float slope, float p_drop, uint64_t seed, uint32_t step, uint32_t layer, int32_t pad_row,
                                 const int64_t *idx_a, int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b,
                                 int64_t off_b, float *g_side_c, float *g_ego_c, float *gW_parts, int32_t part_stride,
                                 void *stream);

/* ------------------------------------------------------------------------------------------------ row-sparse backward
 * After a B-sample batch d loss / d (propagated table) is non-zero on <= 2B rows (model.py:115-116), so the FIRST product
 * of the backward pass, A^T g, touches only the stored entries of those rows (~14 k of Epinion2's 418 k): it is taken
 * in push form instead of a pull-form SpMM over the whole matrix ("hard parts").
 *
 * spex_unique_rows_i32: the distinct values of { idx_a[k] + off_a } U { idx_b[k] + off_b } that lie in [0, n_rows), as a
 * compact device list (arbitrary order) and its length.  stamp: caller-owned device int32[n_rows], zero-initialised once;
 * epoch: any non-zero value not used on this stamp table before (a step counter).  No sort, no host round trip.
 *
 * spex_spmm_push_rows_f32: for the k-th listed row r = list[k] (k < *count) and every stored entry e of row r of the handle:
 *   out[col[e], :] += scale * val[e] * src_k,   src_k = src[r, :] if src_indexed else src[k, :]
 *   and, if add: out[r, :] += scale * add_k (same indexing rule by add_indexed)
 * i.e. out += scale * (A^T scatter(src) + scatter(add)) for the matrix A held by the handle (scale = 1/(L+1) gives the
 * first step of the layer-mean's backward, G = (g + A^T g)/(L+1), in one launch).  out: [n_cols, d], accumulated with
 * 256-byte float atomics — initialise it first (zero, or the term the product is added to).  max_count bounds the launch
 * (one 16-wave workgroup per list slot).  No edge dropout in this form.
 */
int spex_unique_rows_i32(const int64_t *idx_a, int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b, int64_t off_b,
                         int32_t n_rows, int32_t *stamp, int32_t epoch, int32_t *list, int32_t *count, void *stream);
int spex_spmm_push_rows_f32(const spex_graph_t *g, const int32_t *list, const int32_t *count, int32_t max_count,
                            const float *src, int32_t src_indexed, const float *add, int32_t add_indexed, float scale,
                            float *out, int32_t d, void *stream);
/* The same product driven by the batch itself (d == 64), EVERY slot contributing its own row: for slot k with row
 * r = idx_a[k] + off_a (k < n_a) or idx_b[k - n_a] + off_b:
 *   out[col[e], :] += scale * val[e] * src[k, :]  over the stored entries e of row r;   out[r, :] += scale * add[k, :] (if add)
 * src / add are COMPACT per-slot arrays (row strides ld_src / ld_add) — the per-sample gradient rows
 * spex_score_bce_slots_f32 writes, or spex_ngcf_layer_bwd_rows_f32's outputs; rows named by several slots receive several
 * contributions, so nothing is deduplicated.  ONE launch (each small launch costs ~4 us on the stream: a fill + a
 * unique pass + the list form would cost more than the pull-form SpMM they replace); a slot's row is shared by 4
 * workgroups of 4 waves, so a hub row's atomics spread over several CUs.
 */
int spex_spmm_push_batch_f32(const spex_graph_t *g, const int64_t *idx_a, int32_t n_a, int64_t off_a, const int64_t *idx_b,
                             int32_t n_b, int64_t off_b, const float *src, int32_t ld_src, const float *add, int32_t ld_add,
                             float scale, float *out, int32_t d, void *stream);

/* The batch-sized middle of the exact LightGCN training step (L >= 2, d == 64) as ONE launch (under edge dropout — a mask set on the
 * handle with spex_graph_set_edge_mask — the last layer at the batch's rows and the push apply the handle's keep rule entry by entry,
 * exactly as spex_spmm_f32 does: kept values / keep_prob, dropped entries contribute nothing): what the
 * sequence spex_spmm_rowlist_f32 -> spex_score_bce_slots_f32 -> spex_spmm_push_batch_f32 computes, i.e. for sample b with
 * rows u = users[b], i = items[b] + n_user_rows (utility1/model.py:91-97 at the batch's rows, :111-121, and autograd's first
 * backward product):
 *   light_r = (acc_in[r] + (A X)[r]) / acc_div   for r in {u, i}   — the last layer + layer mean at the two rows (same segment
 *                                                                     order as spex_spmm_rowlist_f32: bit-identical rows)
 *   x = <light_u, light_i>;   loss_b = BCEWithLogits(x, labels[b]);   dg = (sigmoid(x) - labels[b]) * grad_scale
 *   loss_per_sample[b] = loss_b (plain store) if loss_per_sample != NULL, else *loss_sum += loss_b (one atomic per sample)
 *   g_u = dg * light_i,  g_i = dg * light_u
 *   g_out[r] += g_r                                                  — dense d loss / d light_out (atomics: rows repeat)
 *   G[r] += push_scale * g_r;  G[col[e]] += push_scale * val[e] * g_r over the stored entries e of row r of `g` ITSELF:
 *                                                                    G = push_scale (g + A^T g), and A^T g in push form walks
 *                                                                    the rows of A — (A^T g)[c] = sum_r A[r, c] g[r]
 * g_out and G are accumulated into: zero them first (the step's Adam pass does).  Samples with an index out of range are
 * skipped (loss 0).  Up to SPEX_BATCH_PARTS workgroups (default 3) share a sample whose rows are long (their pushes' atomics
 * then come from several CUs).
 */
int spex_lightgcn_batch_f32(const spex_graph_t *g, const float *X, const float *acc_in, float acc_div,
                            const int64_t *users, const int64_t *items, const float *labels, int32_t B, int32_t n_user_rows,
                            float grad_scale, float push_scale, float *loss_sum, float *loss_per_sample, float *g_out, float *G,
                            int32_t d, void *stream);
/* The same launch without the push and without any float atomic (the deterministic step): light_r, x, loss_b, dg as above, then
 *   grad_slots[b] = g_u,  grad_slots[B + b] = g_i      ([2B, d] per-sample rows, plain stores; out-of-range samples: zeros)
 * spex_reduce_slots_f32 then adds the slots per table row in ascending slot order. */
int spex_lightgcn_batch_slots_f32(const spex_graph_t *g, const float *X, const float *acc_in, float acc_div, const int64_t *users,
                                  const int64_t *items, const float *labels, int32_t B, int32_t n_user_rows, float grad_scale,
                                  float *loss_sum, float *loss_per_sample, float *grad_slots, int32_t d, void *stream);

/* Deterministic accumulation of a batch's per-slot rows into a dense [n_rows, 64] table — replaces the float atomics of the row-
 * sparse backward where results must repeat bit for bit (and follows the order of the reference's CPU `index_put_(accumulate)` /
 * index_select backward, LightGCN_SPEX/code/utility1/model.py:115-116 and NGCF_SPEX/code/main_rec.py:89-90 under autograd):
 *   slot k names row r_k = idx_a[k] + off_a (k < n_a) or idx_b[k - n_a] + off_b;
 *   out[r] = scale * (slots[k1] + slots[k2] + ...)  over the slots k1 < k2 < ... with r_k == r, added in ASCENDING slot order
 *   mode 0: out[r] is overwritten (rows no slot names are left alone);  mode 1: out[r] += the sum (plain read-modify-write: the
 *   wave that holds the lowest slot of a row owns it).  slots == NULL: the named rows are set to zero (clears what mode 0 wrote).
 * Rows out of range are skipped.  The duplicate scan is quadratic in the batch (n^2 * 8 bytes of cached reads): a validation-mode
 * kernel for the reference's batches of 256, not a throughput path.  d == 64. */
int spex_reduce_slots_f32(const int64_t *idx_a, int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b, int64_t off_b,
                          int32_t n_rows, const float *slots, int32_t ld_slots, float scale, float *out, int32_t mode, int32_t d,
                          void *stream);

/* Scoring + rows backward of the single-layer NGCF model as ONE launch (what spex_score_bce_slots_f32 followed by
 * spex_ngcf_layer_bwd_rows_f32 computes — NGCF_SPEX/code/main_rec.py:84-100 and autograd through :77-83 at the batch's rows):
 * sample b has rows u = users[b], i = n_user_rows + items[b] of the concatenated table all_emb [n, 2d] (= [ego | normalised layer
 * output], as spex_ngcf_layer_fwd_f32 writes it);  x = <all_emb[u], all_emb[i]>;  loss_per_sample[b] = BCEWithLogits(x, labels[b]);
 * dg = (sigmoid(x) - labels[b]) * grad_scale;  slot b (row u) receives the upstream gradient dg * all_emb[i], slot B + b (row i)
 * dg * all_emb[u] — first d columns: the direct gradient of `ego`, last d: the gradient of the normalised output — and the layer's
 * backward runs on those 2B slots exactly as in spex_ngcf_layer_bwd_rows_f32 (same outputs: g_side_c, g_ego_c [2B, d], gW_parts).
 * A sample with an index out of range contributes nothing (loss 0).  d == 64.
 */
int spex_ngcf_score_bwd_rows_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc, const float *W_bi,
                                 const float *b_bi, const float *all_emb, const float *labels, float grad_scale, int32_t n, int32_t d,
                                 float slope, float p_drop, uint64_t seed, uint32_t step, uint32_t layer, int32_t pad_row,
                                 const int64_t *users, const int64_t *items, int32_t B, int64_t n_user_rows, float *loss_per_sample,
                                 float *g_side_c, float *g_ego_c, float *gW_parts, int32_t part_stride, void *stream);
/* The same WITHOUT the concatenated table: the layer's forward at the batch's rows, the scoring and the rows backward in one launch.
 * A tile of the kernel holds both rows of 8 samples (user rows and item rows of samples 8t .. 8t+7), so a sample's other row is in
 * the tile and x = <ego_u, ego_i> + <out_u, out_i> is formed from the layer output the kernel recomputes anyway (out = the
 * normalised, dropped-out layer output of main_rec.py:77-83 at that row).  Inputs: `ego` and `side` = A ego, dense tables read at the
 * batch's rows only (spex_spmm_rowlist_f32 provides `side` there).  Outputs as spex_ngcf_score_bwd_rows_f32 (slot b: users[b],
 * slot B + b: items[b]; gW_parts: spex_ngcf_layer_bwd_rows_parts(2 B) blocks).  With it a training step of the one-layer model
 * never materialises the layer's output.  d == 64.
 */
int spex_ngcf_fwd_score_bwd_rows_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc, const float *W_bi,
                                     const float *b_bi, const float *labels, float grad_scale, int32_t n, int32_t d, float slope,
                                     float p_drop, uint64_t seed, uint32_t step, uint32_t layer, int32_t pad_row,
                                     const int64_t *users, const int64_t *items, int32_t B, int64_t n_user_rows,
                                     float *loss_per_sample, float *g_side_c, float *g_ego_c, float *gW_parts, int32_t part_stride,
                                     void *stream);

/* The forward half of the dual-task model's batch-sized middle (utility1/model_expert_s.py:95-126 at the batch's rows, :154-168)
 * as ONE launch — what spex_spmm_rowlist_f32 -> spex_expert_gate_rows_f32 -> spex_score_bce_slots_f32 compute: for sample b with
 * rows u = users[b], i = items[b] + n_user_rows:
 *   light_r = (acc_in[r] + (A X)[r]) / acc_div,  lo_batch[r] = light_r                       (r in {u, i}; lo_batch is [N, d])
 *   mixed_r = raw[r] * a0 + light_r * a1,  (a0, a1) = softmax([raw[r] | light_r] att)        (att_u for u, att_i for i)
 *   x = <mixed_u, mixed_i>;  loss_b = BCEWithLogits(x, labels[b]);  dg = (sigmoid(x) - labels[b]) * grad_scale
 *   loss_per_sample[b] = loss_b (plain store) if loss_per_sample != NULL, else *loss_sum += loss_b (one atomic per sample)
 *   grad_slots[b] = dg * mixed_i,  grad_slots[B + b] = dg * mixed_u                          ([2B, d], row stride d)
 * spex_expert_gate_rows_bwd_f32 and the push-form product follow as before.  d == 64; under edge dropout the last layer applies
 * the handle's keep rule (as spex_lightgcn_batch_f32 does).
 */
int spex_gated_batch_fwd_f32(const spex_graph_t *g, const float *X, const float *acc_in, float acc_div, const float *raw,
                             const float *att_u, const float *att_i, const int64_t *users, const int64_t *items, const float *labels,
                             int32_t B, int32_t n_user_rows, float grad_scale, float *loss_sum, float *loss_per_sample,
                             float *lo_batch, float *grad_slots, int32_t d, void *stream);

/* The WHOLE batch-sized middle of the dual-task rec branch as ONE launch: spex_gated_batch_fwd_f32's forward, then
 * spex_expert_gate_rows_bwd_f32 (the backward of model_expert_s.py:156-161 at the sample's two rows; linear in the incoming
 * gradient, so a row named by several samples is handled once per sample) and spex_spmm_push_batch_f32 (the first backward
 * product of the propagation, A^T g in push form over the rows of A) — three launches of a dependent chain in one kernel.
 * With d light_r / d raw_r the gate's two outputs for r in {u, i} (same arithmetic as the separate entries), it ACCUMULATES with
 * float atomics into caller-zeroed buffers:
 *   g_prop[r] += d light_r;  g_raw[r] += d raw_r;  G[r] += push_scale * d light_r;
 *   G[col[e]] += val[e] * push_scale * d light_r   over the stored entries e of row r of A;
 *   g_att[b mod n_att_copies] += the gate matrices' gradients;  *loss_sum += loss_b.
 * g_att: [n_att_copies][2][128, 2] — copy c holds [d att_u | d att_i] of the samples b = c (mod n_att_copies); the gradient is the
 * sum of the copies (all samples adding into ONE copy serialise in L2: 256 adds per word; 64 copies cost nothing).
 * g_prop, G, g_raw: [N, d], three distinct tables.  d == 64; under edge dropout the last layer and the push apply the handle's keep
 * rule.  (The deterministic step keeps the separate,
 * atomic-free entries.)
 */
int spex_gated_batch_f32(const spex_graph_t *g, const float *X, const float *acc_in, float acc_div, const float *raw,
                         const float *att_u, const float *att_i, const int64_t *users, const int64_t *items, const float *labels,
                         int32_t B, int32_t n_user_rows, float grad_scale, float push_scale, float *loss_sum, float *g_prop, float *G,
                         float *g_raw, float *g_att, int32_t n_att_copies, int32_t d, void *stream);

/* Replaces the two-expert gate of the dual-task model, utility1/model_expert_s.py:156-161:
 *   att = softmax([raw | prop] att_exp, dim=1) ([n,2d] x [2d,2]);  mixed = raw * att[:,0] + prop * att[:,1]
 */
int spex_expert_gate_f32(const float *raw, const float *prop, const float *att_exp /*[2d,2]*/, float *mixed, int32_t n,
                         int32_t d, void *stream);
/* Its autograd backward: grad_raw / grad_prop ([n,d]) are written; grad_att ([2d,2]) is ACCUMULATED (zero it first). */
int spex_expert_gate_bwd_f32(const float *raw, const float *prop, const float *att_exp, const float *grad_mixed,
                             float *grad_raw, float *grad_prop, float *grad_att, int32_t n, int32_t d, void *stream);
/* The same with the parameter gradient summed in a fixed order (no float atomics; d <= 128): every workgroup leaves its share in
 * its own block of att_parts — [spex_expert_gate_bwd_parts(n)][4 d] floats of caller-owned scratch — and the blocks are added to
 * grad_att in block order. */
int32_t spex_expert_gate_bwd_parts(int32_t n);
int spex_expert_gate_bwd_det_f32(const float *raw, const float *prop, const float *att_exp, const float *grad_mixed,
                                 float *grad_raw, float *grad_prop, float *grad_att, float *att_parts, int32_t n, int32_t d,
                                 void *stream);

/* The gate at a batch's rows only (the training loss reads the gated tables nowhere else, model_expert_s.py:163-166), d == 64.
 * Slot k names table row r = idx_a[k] + off_a (k < n_a) or idx_b[k - n_a] + off_b (raw / prop: [n_rows, 64] tables holding
 * users then items); rows below n_user_rows are gated with att_u (att_exp1), the others with att_i (att_exp2).
 *   forward : mixed_slots[k, :] = the gated row (compact [n_a + n_b, 64]); an out-of-range row gives zeros.
 *   backward: grad_slots[k, :] = d loss / d mixed_slots[k] (row stride ld_slots) ->
 *             grad_prop_slots[k, :] (compact, what spex_spmm_push_batch_f32 consumes) and, ACCUMULATED with atomics (zero them
 *             first): grad_prop[r, :] += d prop, grad_raw[r, :] += d raw (dense [n_rows, 64]), grad_att_u / grad_att_i ([128, 2]).
 *             A row named by several slots is handled once per slot (the gate's backward is linear in the incoming gradient).
 */
int spex_expert_gate_rows_f32(const float *raw, const float *prop, const float *att_u, const float *att_i, const int64_t *idx_a,
                              int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b, int64_t off_b, int64_t n_user_rows,
                              int64_t n_rows, int32_t d, float *mixed_slots, void *stream);
int spex_expert_gate_rows_bwd_f32(const float *raw, const float *prop, const float *att_u, const float *att_i, const int64_t *idx_a,
                                  int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b, int64_t off_b,
                                  int64_t n_user_rows, int64_t n_rows, int32_t d, const float *grad_slots, int32_t ld_slots,
                                  float *grad_prop_slots, float *grad_prop, float *grad_raw, float *grad_att_u, float *grad_att_i,
                                  void *stream);
/* The same backward without a float atomic (the deterministic step): grad_prop_slots AND grad_raw_slots leave as compact per-slot
 * rows ([n_a + n_b, 64]; spex_reduce_slots_f32 adds them per table row in slot order), and every workgroup writes its share of the
 * two gate gradients to its own block of att_parts — [spex_expert_gate_rows_bwd_parts(n_a + n_b)][512] floats, block p =
 * [d att_u (256) | d att_i (256)] — to be added in block order. */
int32_t spex_expert_gate_rows_bwd_parts(int32_t n_slots);
int spex_expert_gate_rows_bwd_det_f32(const float *raw, const float *prop, const float *att_u, const float *att_i, const int64_t *idx_a,
                                      int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b, int64_t off_b,
                                      int64_t n_user_rows, int64_t n_rows, int32_t d, const float *grad_slots, int32_t ld_slots,
                                      float *grad_prop_slots, float *grad_raw_slots, float *att_parts, void *stream);

/* ------------------------------------------------------------------------------------------------ negative sampler
 * Replaces LightTrainData.ng_sample, LightGCN_SPEX/code/utility1/dataloader.py:250-265 (distribution, not stream):
 * for each of n_pos positives (user d_pos_user[p]) draw num_ng items uniformly from [0, num_item), redrawing while the
 * item is in the user's sorted interaction list d_items[d_rowptr[u] : d_rowptr[u+1]] (CSR of R, device).
 * d_out: int64[n_pos * num_ng], slot p * num_ng + t.  Counter-based (philox4x32-10 keyed by `seed`): reproducible,
 * order-independent; not NumPy's stream (the host sampler of the drop-in Loader replays that one exactly).
 */
int spex_sample_negatives(const int32_t *d_rowptr, const int32_t *d_items, int32_t n_user_rows, const int64_t *d_pos_user,
                          int64_t n_pos, int32_t num_ng, int32_t num_item, uint64_t seed, int64_t *d_out, void *stream);

/* ------------------------------------------------------------------------------------------------ learned edge values
 * the Diffnet++ social / interest diffusion — the same SpMM on user x user, user x item and item x user
 * graphs whose stored values are LEARNED (a per-edge parameter pushed through a row softmax), so the values change
 * every step and need a gradient.  Replaces, on fixed sparsity patterns,
 *   tf.sparse.softmax(SparseTensor(indices, values))        Diffnet++_SPEX/code/utility/Model.py:275-286
 *   tf.sparse.sparse_dense_matmul(att_matrix, embedding)    Model.py:18-83 (forward: spex_spmm_f32 after set_values)
 *   and their gradients w.r.t. the values (tape.gradient, Diffnet++_SPEX/code/main_rec.py:36).
 * Every per-edge array here is indexed by EDGE ID (h_edge_id of spex_graph_create; identity = CSR entry order), so a
 * graph and its transposed copy share one values array.  n_val = length of that array (> the largest edge id).
 */

/* Replace the stored values of the handle by d_val[edge_id] (device, fp32); later SpMM launches on the handle use them. */
int spex_graph_set_values(spex_graph_t *g, const float *d_val, int64_t n_val, void *stream);

/* Sampled dense-dense product on the handle's pattern: d_out[edge_id(e)] = <A[row(e),:], B[col(e),:]> for every stored
 * entry e.  A: [n_rows, d], B: [n_cols, d].  With A = dL/dY and B = X this is dL/dval of Y = spmm(val, X).
 * Entries are not weighted by the stored values.  The first call on a handle allocates and builds a per-entry row
 * index (4 B/entry): make that call outside any stream capture.
 */
int spex_sddmm_f32(spex_graph_t *g, const float *A, const float *B, float *d_out, int64_t n_val, int32_t d, void *stream);

/* Row softmax over the stored entries (empty rows: nothing written):
 *   d_out[id(e)] = exp(d_in[id(e)] - max_row) / sum_row exp(d_in - max_row)
 * and its backward  d_grad_in[id(e)] = d_out[id(e)] * (d_grad_out[id(e)] - sum_row d_out * d_grad_out).
 * d_in may equal d_out; d_grad_out may equal d_grad_in.
 */
int spex_edge_softmax_f32(const spex_graph_t *g, const float *d_in, float *d_out, int64_t n_val, void *stream);
int spex_edge_softmax_bwd_f32(const spex_graph_t *g, const float *d_out_val, const float *d_grad_out, float *d_grad_in,
                              int64_t n_val, void *stream);

/* Node-level attention fusion of a Diffnet++ layer — Diffnet++_SPEX/code/utility/Model.py:308-345 / 352-385 — one row
 * per wave, all of it in one launch:
 *   r_k = [U | X_k] . w1_k   (U == NULL: r_k = X_k . w1_k);   e_k = exp(LeakyReLU_0.2(w2_k tanh(r_k + b1_k) + b2_k)) + c_k
 *   out = base_coef U + mix_coef (e_1 X_1 + e_2 X_2) / (e_1 + e_2)
 * users (:308-321): U = user rows, X_1 = from consumed items (c = 0.7), X_2 = from social neighbours (c = 0.3), 1/2, 1/2;
 * items (:323-343): U = NULL, X_1 = the item rows, X_2 = from customers, c = 1, 1, base 0, mix 1.
 * p_k: device parameter block of branch k = [ w1_k ((U ? d : 0) + d floats), b1_k, w2_k, b2_k ].  d <= 256.
 * Backward: grad_U (iff U) / grad_X1 / grad_X2 are written; grad_p1 / grad_p2 are ACCUMULATED (zero them first).
 */
int spex_attn_fuse_f32(const float *U, const float *X1, const float *X2, const float *p1, const float *p2, int32_t n,
                       int32_t d, float c1, float c2, float base_coef, float mix_coef, float *out, void *stream);
int spex_attn_fuse_bwd_f32(const float *U, const float *X1, const float *X2, const float *p1, const float *p2, int32_t n,
                           int32_t d, float c1, float c2, float base_coef, float mix_coef, const float *grad_out,
                           float *grad_U, float *grad_X1, float *grad_X2, float *grad_p1, float *grad_p2, void *stream);

/* ------------------------------------------------------------------------------------------------ trust-path attention
 * SURVEY.md 8f #1.  Replaces GraphAttentionLayer.forward, LightGCN_SPEX/code/utility2/layers.py:15-71 (Python loops
 * over batch x path position), for all heads of a layer in one launch, and its autograd backward.
 *   path p = x_0 .. x_{l-1} (l = seq_l[p] <= L), position i < l - 1, head h with parameter a_h = [a1_h | a2_h] (2 d floats):
 *     positional != 0 (concat=True,  layers.py:22-31): A = src[seq[p,i]] + (l - i),  Bv = src[seq[p,i+1]] + (l - i - 1)
 *     positional == 0 (concat=False, layers.py:58-63): A = src[.., i],               Bv = src[.., i + 1]
 *     att = softmax([A.a1_h + A.a2_h, A.a1_h + Bv.a2_h]);   out[p, i, h*d : (h+1)*d] = att_0 A + att_1 Bv
 *   positions i >= l - 1 copy the raw source row into every head's slot.
 * src: [n_src_rows, d] device fp32.  seq: device int64 [B, L] of row indices into src, or NULL = dense source
 * (src is [B, L, d], row p * L + i; n_src_rows must equal B * L).  seq_l: device int64 [B].  a: [n_heads, 2 d].
 * out: [B, L, n_heads * d].  w0_out (optional, [B, L, n_heads]): att_0, which the backward reads.  d <= 256.
 * Backward: grad_src ([n_src_rows, d]) and grad_a ([n_heads, 2 d], may be NULL) are ACCUMULATED with atomics — zero
 * them first.  (The a1 halves receive nothing: A.a1 cancels in the softmax, their gradient is exactly zero.)
 */
int spex_path_attention_f32(const float *src, int64_t n_src_rows, const int64_t *seq, const int64_t *seq_l, const float *a,
                            int32_t B, int32_t L, int32_t d, int32_t n_heads, int32_t positional, float *out, float *w0_out,
                            void *stream);
int spex_path_attention_bwd_f32(const float *src, int64_t n_src_rows, const int64_t *seq, const int64_t *seq_l,
                                const float *a, int32_t B, int32_t L, int32_t d, int32_t n_heads, int32_t positional,
                                const float *w0, const float *grad_out, float *grad_src, float *grad_a, void *stream);

/* ------------------------------------------------------------------------------------------------ trust head
 * The whole trust branch of the dual-task model — replaces LightGCN_SPEX/code/utility1/model_expert_s.py:170-192 (forward,
 * flag 0 / 2): the in_att heads and out_att (utility2/layers.py:15-71), `mul_seq @ w` + ELU (:181-183), compute_scores
 * (:128-148: soft-attention readout, linear_transform, max-pool, att_t gate, logits against the user table) and
 * nn.CrossEntropyLoss (:192) with their gradients.  Hidden size must be 64; L <= 16 positions, n_heads <= 4.
 *
 * params / grad_params: ONE flat fp32 block, in this order (d = 64, H = n_heads; spex_trust_param_count gives the total):
 *   attention_0.a .. attention_{H-1}.a [H][2d] | out_att.a [2d] | w [H d, d] | linear_one.weight [d, d] | .bias [d] |
 *   linear_two.weight [d, d] | .bias [d] | linear_three.weight [d] | linear_transform.weight [d, 2d] | .bias [d] | att_t [2d, 2]
 * table: the user table incl. its pad row ([n_rows, 64]); seq: [B, L] int64 user ids padded with the pad row; seq_l: [B].
 * hybrid = !nonhybrid (model_expert_s.py:136-141).
 *
 * spex_trust_head_fwd_f32 (one launch): a2_out [B, 64] = the vector whose product with the user table gives the logits
 *   (:146-147) — the evaluation form (flag 2).
 * spex_trust_head_train_f32 (two launches: the fused path kernel — forward chain, the logits / cross-entropy / d a2 sweep of the
 *   user table shared by up to 8 workgroups per path whose last one folds the shares in a fixed order and runs the backward chain —
 *   and the reductions.  SPEX_TRUST_SPLIT=n caps the workgroups per path — a test hook, read per call): forward, logits = a2 . table[0 : n_rows - 1]^T (`b = table[:-1]`), loss =
 *   mean_b CE(logits_b, targets_b) -> *loss_out (added to it if loss_accumulate; may be NULL), and the whole backward:
 *   grad_params (flat block) is OVERWRITTEN (one thread per weight sums its contributions in a fixed order: deterministic);
 *   grad_table [n_rows, 64] is ACCUMULATED by the launch that owns the rows — the logits' part, then the rows of the paths that
 *   pass through the user in (path, position) order; no atomics: the whole head repeats bit for bit — zero it first or pass
 *   the buffer it is to be added to.  Gradients are scaled by scale * (*scale_dev if
 *   scale_dev else 1) — e.g. the multi-task precision exp(-2 s) of main_auto_expert_s.py:81-82 read on the device.
 *   Scratch (caller-owned): a2 [B, 64], dscore [B, n_rows - 1], loss_b [B], ws [spex_trust_workspace_floats(B, L, 64, H, n_rows)]
 *   (per-path blocks + the user tiles' / the shares' partials and the paths' arrival tickets; the workspace need NOT be zeroed:
 *   a ticket carries the call's tag, whatever else the word holds counts as "nobody arrived yet").  One workspace serves one
 *   call at a time (calls on one stream: fine; the same workspace on two streams at once: not).
 */
int64_t spex_trust_param_count(int32_t d, int32_t n_heads);                                /* -1: unsupported shape */
int64_t spex_trust_workspace_floats(int32_t B, int32_t L, int32_t d, int32_t n_heads,
                                    int64_t n_rows);                                        /* -1: unsupported shape */
int spex_trust_head_fwd_f32(const float *table, int64_t n_rows, const float *params, const int64_t *seq, const int64_t *seq_l,
                            int32_t B, int32_t L, int32_t d, int32_t n_heads, int32_t hybrid, float *a2_out, void *stream);
int spex_trust_head_train_f32(const float *table, int64_t n_rows, const float *params, const int64_t *seq, const int64_t *seq_l,
                              const int64_t *targets, int32_t B, int32_t L, int32_t d, int32_t n_heads, int32_t hybrid,
                              float scale, const float *scale_dev, float *a2, float *dscore, float *loss_b, float *ws,
                              float *loss_out, int32_t loss_accumulate, float *grad_params, float *grad_table, void *stream);

/* ------------------------------------------------------------------------------------------------ one-call training step
 * The exact reference training step — LightGCN_SPEX/code/main_rec.py:32-37: forward (model.py:111-121), BCE,
 * loss.backward(), optimizer.step() — as ONE call that issues the library's own launches back to back:
 *   L-1 x spex_spmm_f32 (running layer sum)
 *   spex_lightgcn_batch_f32 (last layer at the batch's rows + scores + BCE + gradient rows + (g + A^T g)/(L+1) in push form;
 *                            L == 1: spex_spmm_rowlist_f32 -> spex_score_bce_slots_f32 -> one pull-form product instead)
 *   L-1 x spex_spmm_f32 on A^T (g/(L+1) fused) -> spex_adam_step_f32 over the whole table (which clears g_out again).
 * 2 L + 1 launches (seven for L = 3).
 * The descriptor holds the step's device buffers (all caller-owned, N = graph rows, d == 64):
 *   E0, m, v, light_out, lo_batch, g_out, grad_E0: [N, d];  ws_fwd: [2, N, d];  ws_bwd: [3, N, d];
 *   grad_slots: [slot_capacity, d] with slot_capacity >= 2B (the batch's per-sample gradient rows).
 * g_out and the first [N, d] of ws_bwd must be all-zero before the first call (every call leaves them all-zero: the Adam
 * pass clears both).  t is advanced by the call.
 * users / items: device int64[B] (items index the item block: row n_user_rows + items[b]); labels: device fp32[B].
 * *loss_sum (device) accumulates the batch's BCE loss SUM.  L >= 1.  Edge dropout (`--dropout 1 --keepprob p`, the reference's
 * recommended configuration): set the step's mask on BOTH handles with spex_graph_set_edge_mask before the call — graph_t must
 * then be the transposed handle created with the edge-id permutation (a masked adjacency is not symmetric) and L >= 2; every
 * product of the step, the batch kernel's included, drops the same edges.
 * graph is walked by the forward AND by the push-form first backward product (A^T g in push form walks the rows of A); graph_t
 * (A^T) by the pull-form products.  t is advanced only when every launch of the step was queued (a failed call leaves it alone).
 *
 * flags & SPEX_STEP_DETERMINISTIC (all three step descriptors): no float atomics anywhere in the step — per-sample gradient rows
 * are added per table row in ascending slot order (spex_reduce_slots_f32: the order of the reference's CPU index backward), the
 * whole backward propagation runs in pull form (spex_propagate_bwd_f32: one chain per output row in ascending column order, like
 * the reference's sparse addmm), per-sample losses are summed in sample order.  Two runs from the same state then produce
 * bit-identical parameters; the price is one more SpMM-sized launch and two small ones per step (a validation mode: the fast
 * path's atomics only reorder sums, which Adam amplifies along NGCF's scale-invariant direction over thousands of steps).
 */
enum {
    SPEX_STEP_DETERMINISTIC = 1,        /* fixed summation order everywhere (see above) */
    SPEX_STEP_FIXED_TASK_WEIGHTS = 2,   /* dual-task step only: loss = loss1 + loss2 (LightGCN_SPEX/code/main_11.py:69) instead of the
                                         * uncertainty weighting of main_auto_expert_s.py:78-82; task_weights are left untouched */
    SPEX_STEP_PIPELINED = 4             /* dual-task step only, with side_stream: the Adam pass split by owner over the two streams, no
                                         * fork / join on the critical cycle; see spex_dual_task_step_t and spex_dual_task_step_join */
};
/* The north-star step — LightGCN L-layer propagation + the fused BPR gather + dot + sigmoid + SGD kernel over T triples — as one
 * call of L + 1 launches: layer 1 with the running sum fused (sum1 = E^0 + E^1), layers 2 .. L in the plain form (no epilogue
 * operand, one output stream), and the BPR kernel forms the layer mean ((sum1 + E^2) + E^3) / (L + 1) — utility1/model.py:94-95, in
 * the fused epilogues' order — at its triples' rows only, the rows of the propagated table the step reads; the updates go to E^0.
 * Same results as spex_propagate_f32 followed by spex_bpr_sgd_step_f32 reading its output (bit-identical rows; float-atomic
 * updates in both).  E0, sum1: [N, 64]; ws: [2, N, 64]; users index rows [0, n_user_rows), items rows n_user_rows + i.
 * d == 64, 1 <= L <= 3, no edge dropout; *loss_sum accumulates the batch's softplus sum. */
int spex_lightgcn_step_bpr_f32(const spex_graph_t *g, float *E0, float *sum1, float *ws, int32_t n_user_rows, int32_t L, int32_t d,
                               const int64_t *u, const int64_t *i_pos, const int64_t *i_neg, int64_t T, float lr, float reg,
                               float *loss_sum, void *stream);

typedef struct spex_lightgcn_step {
    const spex_graph_t *graph, *graph_t;     /* A and A^T (the same handle for the symmetric LightGCN adjacency) */
    float *E0, *m, *v;
    float *light_out, *ws_fwd, *lo_batch, *g_out, *ws_bwd, *grad_E0, *grad_slots;
    int32_t slot_capacity, n_user_rows, L, d;
    float lr, beta1, beta2, eps;
    int32_t t;
    int32_t flags;                           /* SPEX_STEP_DETERMINISTIC */
} spex_lightgcn_step_t;
int spex_lightgcn_step_bce_f32(spex_lightgcn_step_t *step, const int64_t *users, const int64_t *items, const float *labels,
                               int32_t B, float *loss_sum, void *stream);
/* Train() of main_rec.py:30-37 over a whole pre-shuffled, device-resident epoch as ONE call: batch k = samples [k B, min((k+1) B, n))
 * through spex_lightgcn_step_bce_f32 (at most max_steps batches; < 0: all).  *loss_full accumulates the loss sums of the full batches,
 * *loss_ragged that of a shorter last batch (main_rec.py:36 adds per-batch MEAN losses: loss_full / B + loss_ragged / (n mod B)).
 * keep_prob < 1: edge dropout (model.py:46-55) with the in-kernel sampled mask, a fresh one per step — seed = (drop_seed << 32) | step,
 * steps counted from 1, exactly what spex_graph_set_edge_mask(g, 2, NULL, keep_prob, seed) on both handles before each step gives; the
 * handles are left unmasked.  (The descriptor's graph handles are modified by that: not const here.) */
int spex_lightgcn_epoch_bce_f32(spex_lightgcn_step_t *step, const int64_t *users, const int64_t *items, const float *labels, int64_t n,
                                int32_t B, int64_t max_steps, float keep_prob, uint32_t drop_seed, float *loss_full, float *loss_ragged,
                                void *stream);

/* The single-layer NGCF training step (NGCF_SPEX/code/main_rec.py:122-128 with the default --layer_size [64]) as one call:
 *   spex_spmm_f32 (side = A ego) -> spex_ngcf_layer_fwd_f32 -> spex_ngcf_score_bwd_rows_f32 (scores, BCE, rows backward)
 *   -> spex_spmm_push_batch_f32 -> spex_adam_step_f32 (table; clears its gradient, adds the step's per-sample losses to
 *   loss_sum in a fixed order) -> spex_adam_step_sum_f32 (layer weights).  Six launches.
 * Buffers (caller-owned, N = graph rows incl. an isolated pad row if the table keeps one, d == 64):
 *   E0, mE, vE, side, grad: [N, d] (grad all-zero before the first call; every call leaves it all-zero);  all_emb: unused since ABI 5
 *   (the step no longer forms the concatenated table; the field keeps the layout, NULL is accepted);
 *   W, mW, vW: the layer's weights as one block [W_gc d*d | b_gc d | W_bi d*d | b_bi d] and its Adam moments;
 *   g_slots: [slot_capacity, 2d];  g_side_c, g_ego_c: [slot_capacity, d];  gW_parts: [slot_capacity / 16, 2 (d*d + d)].
 * users index rows [0, n_user_rows), items rows n_user_rows + items[b].  Message dropout: (p_drop, seed, dropout_step, layer 0),
 * dropout_step advanced by the call when p_drop > 0; t advanced by the call.
 */
typedef struct spex_ngcf_step {
    const spex_graph_t *graph;               /* A = D^-1 (A + I) */
    float *E0, *mE, *vE, *W, *mW, *vW;
    float *all_emb, *side, *g_slots, *g_side_c, *g_ego_c, *gW_parts, *grad;
    int32_t slot_capacity, n_user_rows, pad_row;
    float slope, p_drop;
    uint64_t seed;
    int32_t dropout_step, t;
    float lr, beta1, beta2, eps;
    void *side_stream;   /* optional second hipStream_t of the caller (NULL: one stream): the layer weights' Adam pass runs on it
                          * beside the push-form product and the table's Adam pass and is joined at the end of the call.
                          * Measured SLOWER than one stream on the MI355X (69 vs 59 us per step: a cross-stream fork + join
                          * costs ~10 us, the hidden pass takes 5) — leave NULL unless the forked work is long */
    void *ev_fork, *ev_join;   /* the two-stream form's events: zero-initialise; created by the library on first use, one pair per
                                * descriptor; release with spex_step_events_release */
    const spex_graph_t *graph_t;             /* A^T — SPEX_STEP_DETERMINISTIC only (pull-form first backward product), else NULL */
    float *g_side_dense, *g_ego_dense;       /* [N, d] each, all-zero before the first call — SPEX_STEP_DETERMINISTIC only */
    int32_t flags;
} spex_ngcf_step_t;
int spex_step_events_release(void **ev_fork, void **ev_join);   /* destroys the pair a descriptor holds (no step in flight) */
int spex_ngcf_step_bce_f32(spex_ngcf_step_t *step, const int64_t *users, const int64_t *items, const float *labels, int32_t B,
                           float *loss_sum, void *stream);
/* train() of NGCF_SPEX/code/main_rec.py:116-131 over a whole pre-shuffled, device-resident epoch as ONE call: batch k = samples
 * [k B, min((k+1) B, n)) through spex_ngcf_step_bce_f32 (at most max_steps batches; < 0: all); *loss_full / *loss_ragged as in
 * spex_lightgcn_epoch_bce_f32. */
int spex_ngcf_epoch_bce_f32(spex_ngcf_step_t *step, const int64_t *users, const int64_t *items, const float *labels, int64_t n, int32_t B,
                            int64_t max_steps, float *loss_full, float *loss_ragged, void *stream);

/* NGCF with MORE than one layer (`--layer_size [64,64,..]`, NGCF_SPEX/code/ngcf_parser.py:12; the layer loop of main_rec.py:71-93) —
 * the training step as one call of the library's own launches, L >= 2 layers of width 64:
 *   forward   layers 0 .. L-2 over the whole table (spex_spmm_f32 -> spex_ngcf_layer_fwd_f32: their outputs feed the next layer's
 *             product everywhere); the LAST layer at the batch's rows only (spex_spmm_rowlist_f32 -> spex_ngcf_layer_fwd_rows_f32: the
 *             loss reads the concatenated table nowhere else, main_rec.py:96-104)
 *   scoring   spex_score_bce_slots_f32 on the concatenated table [N, 64 (L + 1)]: per-sample gradient rows + the table form the
 *             earlier layers read
 *   backward  last layer on the batch's slots (spex_ngcf_layer_bwd_rows_f32) -> push-form A^T product (spex_spmm_push_batch_f32) ->
 *             for l = L-2 .. 0: spex_ngcf_layer_bwd_f32 (dense, four waves per tile) -> spex_spmm_f32 on A^T with g_ego added
 *   Adam      the table; the last layer's weights from their partial blocks; the other layers' weights (their pass clears gW).
 * Buffers (caller-owned; N = graph rows incl. the isolated pad row if the table keeps one; per = 2 (64 * 64 + 64)):
 *   E0, mE, vE: [N, 64];  W, mW, vW, gW: [L][per] (layer l: [W_gc | b_gc | W_bi | b_bi]; gW all-zero before the first call, every call
 *   leaves it so);  all_emb, g_all: [N, 64 (L + 1)] (g_all all-zero before the first call; every call leaves it so);
 *   sides: [L][N, 64];  egos: [L - 1][N, 64] (the inputs of layers 1 .. L-1);  g_slots: [slot_capacity, 64 (L + 1)];
 *   g_side_c, g_ego_c: [slot_capacity, 64];  gW_parts: [slot_capacity / 16, per];  g_side, g_ego: [N, 64];  g_next: [2][N, 64].
 * p_drop: host [L] (message dropout per layer), mask = (seed, dropout_step, layer); dropout_step advanced when any p_drop > 0;
 * t advanced by the call.  Float atomics in the scoring tables and the push (no deterministic mode for L >= 2). */
typedef struct spex_ngcf_deep_step {
    const spex_graph_t *graph, *graph_t;     /* A = D^-1 (A + I) and its transpose */
    float *E0, *mE, *vE, *W, *mW, *vW, *gW;
    float *all_emb, *g_all, *sides, *egos, *g_slots, *g_side_c, *g_ego_c, *gW_parts, *g_side, *g_ego, *g_next;
    const float *p_drop;
    int32_t L, slot_capacity, n_user_rows, pad_row;
    float slope;
    uint64_t seed;
    int32_t dropout_step, t;
    float lr, beta1, beta2, eps;
} spex_ngcf_deep_step_t;
int spex_ngcf_deep_step_bce_f32(spex_ngcf_deep_step_t *step, const int64_t *users, const int64_t *items, const float *labels, int32_t B,
                                float *loss_sum, void *stream);

/* The dual-task training step of LightGCN_SPEX/code/main_auto_expert_s.py:63-89 (model_expert_s.LightGCN.forward flag 0 +
 * uncertainty-weighted loss + loss.backward() + optimizer.step()) as one call issuing 2 L + 3 launches:
 *   rec branch, row-sparse like spex_lightgcn_step_bce_f32: (L-1) x spex_spmm_f32 + spex_gated_batch_f32 (last layer at the
 *   batch's rows, gate, scores, the gate's backward and the push-form first backward product: one launch; the deterministic
 *   step and L == 1: spex_gated_batch_fwd_f32 -> spex_expert_gate_rows_bwd_[det_]f32 -> push / pull) -> (L-1) x spex_spmm_f32 on
 *   A^T (L == 3: both in the plain form, the Adam pass adds the push target);   trust branch: spex_trust_head_train_f32;
 *   then one Adam pass over the whole parameter arena, which applies the task precisions exp(-2 s_k) to the two branches'
 *   gradients, forms the task weights' own gradients (d/ds0 = -2 p1 loss1 + 2 (n_rec + 1) B, d/ds1 = -2 p2 loss2 + T) and
 *   clears every accumulate-into buffer for the next step.
 * params / m / v: ONE arena (and its two Adam moments), N = graph rows, P = spex_trust_param_count(64, n_heads):
 *   [ table N*64 (users incl. pad row, then items) | trust block P | att_exp1 256 | att_exp2 256 | task_weights 2 ]
 * Work buffers (caller-owned): light, lo_batch, g_prop, g_raw, g_E0: [N, 64]; ws_fwd [2, N, 64]; ws_bwd [3, N, 64];
 *   mixed_slots, grad_slots, g_prop_slots: [slot_capacity, 64] with slot_capacity >= 2B; arange: int64 [slot_capacity] = 0, 1, ..;
 *   g_user [n_user_rows, 64]; g_small [P + 512]; a2 [path_capacity, 64]; trust_ws
 *   [spex_trust_workspace_floats(path_capacity, path_len, 64, n_heads, n_user_rows)]; dscore [path_capacity, n_user_rows - 1];
 *   loss_b [path_capacity]; loss [2], loss_acc [2], precision [2][2].
 * Before the first call: g_prop, g_raw, the first [N, 64] of ws_bwd, g_user, g_small, loss AND grad_slots all-zero (every call
 * leaves them so: on the fast path — no SPEX_STEP_DETERMINISTIC, L >= 2 — the first min(64, slot_capacity / 8) x 512 floats of
 * grad_slots hold the copies of the two gate gradients that spex_gated_batch_f32 accumulates, summed and cleared by the Adam pass;
 * the other paths use the area for per-sample rows and the Adam pass clears it behind them); precision[(t + 1) & 1] = {exp(-2 s0), exp(-2 s1)} for the current task weights (every call writes the next step's
 * slot).  loss_acc accumulates (loss1, loss2) of every call — what Train() sums with .item() per step.  t is advanced.
 * seq: [T, path_len] int64 padded with the pad row's index n_user_rows - 1; T == 0 skips the trust branch (the reference
 * would produce NaN there: CrossEntropyLoss over an empty batch).  L >= 1.  Edge dropout (model_expert_s.py:104-109; the
 * reference's recommended `--dropout 1 --keepprob 0.3`): the step's mask on BOTH handles before the call, graph_t the transposed
 * handle with the edge-id permutation, L >= 2 — as in spex_lightgcn_step_bce_f32.
 * Two streams: the rec branch (2L + 4 launches that fill the chip) and the trust branch (two launches of <= path_capacity
 * workgroups: a latency chain on a few CUs) read the same parameters and write disjoint buffers, so with side_stream set
 * the trust branch is forked onto it behind everything already queued on `stream` and joined again in front of the Adam
 * pass (two events per device, created on first use and kept by the library).  Results are identical to the one-stream
 * order; the caller keeps side_stream alive while steps are in flight.
 * flags & SPEX_STEP_PIPELINED (needs side_stream): the fork and the join above cost ~10 us each on this runtime and sit on the
 * step's critical cycle (Adam -> fork -> trust branch -> join -> Adam).  The pipelined form takes them off it by splitting the
 * Adam pass by OWNER: the user rows, the trust block and the task weights — everything the trust branch reads or writes — are
 * updated by a second Adam launch on side_stream, right behind the trust branch (it waits for the rec branch's gradients, which
 * are ready long before); the item rows and the gate matrices by the launch on `stream`.  The trust branch of step k+1 then
 * follows step k's update on its own stream with no event between them, and `stream` waits for the side update at the START
 * of the next call (its rec branch reads the user rows).  Same arithmetic, same results.  The price is the contract: when the
 * call returns, side_stream is still AHEAD of `stream` — the next pipelined call picks that up itself, anything else that reads
 * or writes the parameters, the moments, loss_acc or the work buffers must be ordered behind spex_dual_task_step_join(step,
 * stream); and seq / seq_l / targets are read on side_stream WITHOUT waiting for `stream` (only the first call after a join
 * forks from it), so they must have been complete before the last join — e.g. staged once per epoch, as Train() allows.
 */
typedef struct spex_dual_task_step {
    const spex_graph_t *graph, *graph_t;
    float *params, *m, *v;
    float *light, *ws_fwd, *lo_batch, *g_prop, *g_raw, *g_E0, *ws_bwd;
    float *mixed_slots, *grad_slots, *g_prop_slots;
    const int64_t *arange;
    float *g_user, *g_small;
    float *a2, *trust_ws, *dscore, *loss_b;
    float *loss, *loss_acc, *precision;
    int32_t slot_capacity, path_capacity, path_len, n_user_rows, L, d, n_heads, hybrid, n_rec;
    float lr, beta1, beta2, eps;
    int32_t t;
    void *side_stream;   /* optional second hipStream_t of the caller (NULL: one stream): the trust branch is issued on it and runs
                          * CONCURRENTLY with the rec branch — the two only meet in the Adam pass (see below) */
    void *ev_fork, *ev_join;   /* as in spex_ngcf_step_t: zero-initialise, release with spex_step_events_release */
    float *g_raw_slots;        /* [slot_capacity, 64]                                   — SPEX_STEP_DETERMINISTIC only */
    float *att_parts;          /* [spex_expert_gate_rows_bwd_parts(slot_capacity)][512] — SPEX_STEP_DETERMINISTIC only */
    float *loss_rows;          /* [slot_capacity / 2] per-sample rec losses             — SPEX_STEP_DETERMINISTIC only */
    int32_t flags;             /* SPEX_STEP_DETERMINISTIC | SPEX_STEP_FIXED_TASK_WEIGHTS | SPEX_STEP_PIPELINED */
    int32_t side_pending;      /* state of the pipelined form (zero-initialise): work of this descriptor is in flight on side_stream */
} spex_dual_task_step_t;
int spex_dual_task_step_f32(spex_dual_task_step_t *step, const int64_t *users, const int64_t *items, const float *labels, int32_t B,
                            const int64_t *seq, const int64_t *seq_l, const int64_t *targets, int32_t T, void *stream);
/* Train() of main_auto_expert_s.py:60-91 over a whole pre-shuffled, device-resident epoch as ONE call: batch k = samples
 * [k B, min((k+1) B, n)) (at most max_steps batches; < 0: all) with the paths [path_off[k], path_off[k+1]) of the epoch's staged path
 * arrays — seq [n_paths, path_len], seq_l, targets on the device, path_off a HOST array of n_batches + 1 ascending offsets (the
 * reference selects each batch's paths with random.sample, :64-71: done up front on the host) — through spex_dual_task_step_f32; the
 * losses accumulate in the descriptor's loss_acc.  keep_prob < 1: the rec branch's sampled edge mask, a fresh one per step, as in
 * spex_lightgcn_epoch_bce_f32.  With SPEX_STEP_PIPELINED, finish with spex_dual_task_step_join. */
int spex_dual_task_epoch_f32(spex_dual_task_step_t *step, const int64_t *users, const int64_t *items, const float *labels, int64_t n,
                             int32_t B, int64_t max_steps, const int64_t *seq, const int64_t *seq_l, const int64_t *targets,
                             const int64_t *path_off, float keep_prob, uint32_t drop_seed, void *stream);
/* Orders everything a pipelined step left on side_stream in front of whatever is queued on `stream` next (a no-op otherwise). */
int spex_dual_task_step_join(spex_dual_task_step_t *step, void *stream);

/* ------------------------------------------------------------------------------------------------ multi-GPU: collectives + partitioned step
 * the graph is 1-D row-partitioned over the GPUs of one node (one process per GPU): rank p owns rows
 * [r_p, r_{p+1}) of A (and of A^T), of every layer's table and of the Adam moments; a layer is an all-gather of the current
 * layer's rows over xGMI followed by the local SpMM.  The reference's analogue is the serial fold loop of --A_split
 * (LightGCN_SPEX/code/utility1/model.py:84-89, dataloader.py:167-177).  Layout: shards are padded to max_rows = the largest
 * shard, the gathered table is [world * max_rows, d] (rank q's rows in slot q), and a rank's block of A has its column indices
 * rewritten once into that layout, so the SpMM reads the gathered buffer as it arrives.
 * The collectives run on RCCL, bound at run time (dlopen: the library loads without RCCL; inside a PyTorch process the
 * librccl torch already holds is the one used).  Errors: SPEX_ERR_COMM with the RCCL message in spex_last_error().
 */
#define SPEX_COMM_ID_BYTES 128
typedef struct spex_comm spex_comm_t;
/* id_out: SPEX_COMM_ID_BYTES bytes (an ncclUniqueId), created on ONE rank and handed to all by the host (any transport). */
int spex_comm_unique_id(void *id_out);
/* Collective over all `world` ranks (ncclCommInitRank), on the calling thread's current device. */
int spex_comm_create(int32_t rank, int32_t world, const void *unique_id, spex_comm_t **out);
int spex_comm_destroy(spex_comm_t *comm);
int spex_comm_info(const spex_comm_t *comm, int32_t *rank, int32_t *world);
/* All-gather of row shards: send = this rank's rows ([<= max_rows, d], padded shard), recv = the gathered table
 * [world * max_rows, d].  rows_per_rank == NULL: one equal-size ncclAllGather of the padded shards (world * max_rows rows on
 * the wire).  rows_per_rank (host, [world]): only the REAL rows move, as one group of point-to-point sends / receives — rank p's
 * rows straight into slot p of every peer's table, world - 1 transfers in flight at once (one per xGMI link of the full mesh,
 * where a ring serialises world - 1 steps over one link); the slots' padding tails are never written nor read.  send may be
 * recv's own slot.  Asynchronous on `stream`. */
int spex_comm_allgather_rows_f32(spex_comm_t *comm, const float *send, float *recv, int64_t max_rows, int32_t d,
                                 const int32_t *rows_per_rank, void *stream);
/* In-place sum over ranks (the owner-computes exchange of a batch's rows: every rank contributes the rows it owns to a zero-filled
 * buffer). */
int spex_comm_allreduce_sum_f32(spex_comm_t *comm, float *buf, int64_t n, void *stream);

/* The row-partitioned propagation / exact training step as one native call each (what spex_amd/dist.py's PartitionedLightGCN.
 * propagate and PartitionedStepper.step_bce issue launch by launch):
 *   spex_partitioned_propagate_f32: L x (exchange of the current rows, spex_spmm_f32 on the rank's block with the running layer
 *     sum fused) -> light_out = the rank's rows of mean(E^0 .. E^L)                    (utility1/model.py:66-97)
 *   spex_partitioned_step_bce_f32: that forward; the batch's 2B propagated rows fetched owner-computes
 *     (spex_gather_owned_rows_f32 + one all-reduce of 2B rows); scores + BCE + gradient rows on the compact rows; the rows this
 *     rank owns added into its gradient block (spex_scatter_add_owned_rows_f32 — or, with SPEX_STEP_DETERMINISTIC, per-sample
 *     rows added in slot order by spex_reduce_slots_f32); backward G_l = g / (L+1) + A^T G_{l+1} as L x (exchange, SpMM on the
 *     block of A^T); Adam on the rank's rows.  2 L exchanges + 1 small all-reduce per step, no host work between launches.
 *   Every exchange is IN PLACE: a layer's SpMM writes its output into the rank's own slot (rows rank * max_rows ..) of the table the
 *   next exchange completes (gathered and gathered1 alternate), so no send buffer and no copy of a layer's rows exist (before: 2 L
 *   staging copies per step).
 *   FAST PATH of spex_partitioned_step_bce_f32 (gathered2 given, L >= 2, not deterministic; the same choice on EVERY rank — the two
 *   schedules differ in their collectives) — spex_lightgcn_step_bce_f32's schedule on the partition: forward layers 1 .. L-1 over the
 *   block (plain form for L <= 3), the LAST layer at the batch's rows on their owners (spex_spmm_owned_rows_f32) -> the all-reduce of
 *   2B rows -> scores, BCE, gradient rows, the owner-computes adds AND the backward's FIRST product in push form in ONE launch on the
 *   replicated rows — no exchange for it: every rank holds all 2B gradient rows and pushes them through graph_push = its own
 *   columns of A -> L - 1 pull-form products on A^T's block, the last one plain (Adam adds its g / (L+1) share; L == 3: both plain,
 *   Adam adds the push target).  2 L - 1 exchanges + 1 all-reduce.  graph_push: handle of the (world * max_rows) x n_local matrix
 *   whose row p holds A[p, c] for the columns c the rank owns (local column indices) = the transpose of the rank's block of A^T;
 *   gathered2 [world * max_rows, 64]: its own slot is the push target, all-zero before the first call (the Adam pass leaves it so).
 * pos: device int64 [2B] — the batch's rows in the padded gathered layout (users, then items): owner(r) * max_rows + r - r_owner.
 * The batch is replicated on every rank; every rank accumulates the same loss sum into its own *loss_sum.
 * Buffers (caller-owned): E0, m, v, light_out, g_local, gs, grad_E0: [n_local, 64] (g_local all-zero before the first call;
 * every call leaves it so); gathered, gathered1: [world * max_rows, 64] each (two tables); rows, grad_rows: [slot_capacity, 64] with
 * slot_capacity >= 2B; arange: device int64 [slot_capacity] = 0, 1, 2, ...
 * graph / graph_t: this rank's row blocks (n_local rows, world * max_rows columns).  d == 64, L >= 1.
 * Edge dropout (utility1/model.py:46-64; since round 4): set the SAME mask on graph and graph_t (spex_graph_set_edge_mask), both created
 * with the entries' GLOBAL edge ids (spex_graph_create's edge_id: the entry's index in A; for the block of A^T the permutation), graph_t a
 * handle of its own — every rank then drops the same edges of A and of A^T.  The fast path stays available when graph_push carries the
 * same mask (and, per entry, the global edge id of the entry of A it came from): its rows-only layer and its push apply the keep rule
 * entry by entry; with an unmasked graph_push a masked step takes the launch-by-launch schedule. */
typedef struct spex_partitioned_step {
    const spex_graph_t *graph, *graph_t;
    spex_comm_t *comm;
    const int32_t *rows_per_rank;      /* host [world], or NULL: see spex_comm_allgather_rows_f32 */
    float *E0, *m, *v;
    float *light_out, *g_local, *gs, *grad_E0, *gathered1, *gathered, *rows, *grad_rows;
    const int64_t *arange;
    int32_t n_local, max_rows, slot_capacity, L, d;
    float lr, beta1, beta2, eps;
    int32_t t, flags;                  /* t is advanced by spex_partitioned_step_bce_f32; flags: SPEX_STEP_DETERMINISTIC */
    const spex_graph_t *graph_push;    /* fast path of spex_partitioned_step_bce_f32 (below); NULL on a rank without rows */
    float *gathered2;                  /* NULL: the launch-by-launch schedule */
} spex_partitioned_step_t;
int spex_partitioned_propagate_f32(spex_partitioned_step_t *step, void *stream);
int spex_partitioned_step_bce_f32(spex_partitioned_step_t *step, const int64_t *pos, const float *labels, int32_t B, float *loss_sum,
                                  void *stream);

/* BASELINE config 5's multi-GPU form — the dual-task step (LightGCN_SPEX/code/main_auto_expert_s.py:53-91 with
 * utility1/model_expert_s.py) on the row partition, as ONE native call:
 *   rec branch   L x (exchange, SpMM on the rank's block, running layer sum fused); the FIRST exchange's table (= E^0 of every rank)
 *                is kept in gathered0.  The batch's 2B rows of E^0 AND of the propagated table are fetched owner-computes (two
 *                spex_gather_owned_rows_f32 into one buffer, ONE all-reduce of 4B rows), and the gate (model_expert_s.py:154-161),
 *                the scores, the BCE loss and the gate's backward run on those compact rows — replicated: every rank computes the
 *                same 2B rows, so the two gate matrices' gradients are COMPLETE on every rank and need no collective (the Python
 *                schedule of round 3 gated all local rows and all-reduced 1 KB of gate gradients instead).  Each rank adds the
 *                gradient rows it owns into its blocks (d loss / d propagated rows -> backward L x (exchange, SpMM on the block of
 *                A^T); d loss / d E^0 rows directly).
 *   trust branch the user block of gathered0 gathered into the contiguous user_table ([n_user_rows, 64], user_pos = the user rows'
 *                positions in the padded layout) and spex_trust_head_train_f32 on it — redundantly on every rank (<= path_capacity
 *                paths), identical results by construction; with side_stream set it runs beside the rec branch (forked behind the
 *                first exchange, joined in front of the Adam pass).
 *   Adam         one pass over the rank's arena [n_local x 64 table rows | trust block P | att_exp1 256 | att_exp2 256 |
 *                task_weights 2]: owner-computes on the table rows (the rank's user rows take the trust head's table gradient at
 *                rows user_lo ..), replicated — identical gradients, identical updates — on the dense parameters; the task
 *                precisions and the task weights' own gradients as in spex_dual_task_step_f32.
 * 2 L exchanges + 1 all-reduce per step, every exchange in place (see spex_partitioned_step_t).  flags: SPEX_STEP_DETERMINISTIC (owned
 * rows added in slot order, no float atomics), SPEX_STEP_FIXED_TASK_WEIGHTS.  d == 64, L >= 1.  Edge dropout on the rec branch
 * (model_expert_s.py:104-109): as for spex_partitioned_step_t — the same mask on both handles (and on graph_push for the fast path).
 * FAST PATH (graph_push and gathered2 given, L >= 2, not deterministic) — spex_dual_task_step_f32's schedule on the partition:
 *   forward layers 1 .. L-1 over the block (plain form for L <= 3), the LAST layer at the batch's rows only, on their owners
 *   (spex_spmm_owned_rows_f32: layer mean + raw rows, zeros elsewhere) -> the one all-reduce of 4B rows -> gate, scores, BCE, the
 *   gate's backward, the owner-computes adds AND the backward's FIRST product in push form, in ONE launch on the replicated compact
 *   rows — that product needs NO exchange: every rank holds all 2B gradient rows and pushes them through graph_push = its own
 *   columns of A -> L - 1 pull-form products on A^T's block.  2 L - 1 exchanges + 1 all-reduce; 2 L - 2 whole-block launches instead
 *   of 2 L and two batch-sized launches instead of nine.  (grad_slots doubles as the gate gradients' scratch copies on this path; the
 *   trust branch's user block is gathered in front of the fork.)
 *   graph_push: handle of the (world * max_rows) x n_local matrix whose row p holds the entries A[p, c] for the columns c the rank
 *   owns (local column indices) — the transpose of the rank's block of A^T; gathered2 [world * max_rows, 64]: its own slot is the
 *   push target, all-zero before the first call (the Adam pass leaves it so).  gathered2 NULL: the launch-by-launch schedule above —
 *   the same choice on EVERY rank (the two schedules differ in their collectives); a rank without rows passes graph_push = NULL.
 * Buffers (caller-owned): params / m / v: the arena above; light, g_prop, g_raw, gs, g_E0: [n_local, 64] (g_prop, g_raw all-zero
 * before the first call; every call leaves them so); gathered, gathered1, gathered0 [world * max_rows, 64] (three tables);
 * rows [2 * slot_capacity, 64]; mixed_slots, grad_slots, g_prop_slots, g_raw_slots [slot_capacity, 64], slot_capacity >= 2B;
 * loss_rows [slot_capacity]; att_parts [spex_expert_gate_rows_bwd_parts(slot_capacity) * 512]; arange int64 [slot_capacity];
 * g_user [n_user_rows, 64] and g_small [P + 512] (both all-zero before the first call; the Adam pass leaves them so); a2, trust_ws, dscore, loss_b, loss [2],
 * loss_acc [2], precision [2][2] as in spex_dual_task_step_t (trust workspace sized for n_user_rows).
 * pos: device int64 [2B], the batch's rows in the padded layout (users, then items).  t is advanced by the call. */
typedef struct spex_partitioned_dual_step {
    const spex_graph_t *graph, *graph_t;
    spex_comm_t *comm;
    const int32_t *rows_per_rank;
    float *params, *m, *v;
    float *light, *g_prop, *g_raw, *gs, *g_E0, *gathered1, *gathered, *gathered0;
    const int64_t *user_pos;
    float *user_table;
    float *rows, *mixed_slots, *grad_slots, *g_prop_slots, *g_raw_slots, *loss_rows, *att_parts;
    const int64_t *arange;
    float *g_user, *g_small;
    float *a2, *trust_ws, *dscore, *loss_b;
    float *loss, *loss_acc, *precision;
    int32_t n_local, max_rows, n_local_users, user_lo;   /* the rank's first n_local_users rows are user rows user_lo .. */
    int32_t slot_capacity, path_capacity, path_len, n_user_rows, L, d, n_heads, hybrid, n_rec;
    float lr, beta1, beta2, eps;
    int32_t t;
    void *side_stream, *ev_fork, *ev_join;   /* optional second stream + the library's event cells (zero-initialise; release with
                                              * spex_step_events_release) */
    int32_t flags;
    const spex_graph_t *graph_push;          /* fast path (see above); NULL: the launch-by-launch schedule */
    float *gathered2;
} spex_partitioned_dual_step_t;
int spex_partitioned_dual_task_step_f32(spex_partitioned_dual_step_t *step, const int64_t *pos, const float *labels, int32_t B,
                                        const int64_t *seq, const int64_t *seq_l, const int64_t *targets, int32_t T, void *stream);

/* ------------------------------------------------------------------------------------------------ profiling hook
 * Not part of any reference interface: lets a caller time the dominant kernel itself, in place, on the stream it is
 * launched on (bench.py's roofline figure).  While a timer is attached to a graph, every (or every n-th) call of
 * spex_spmm_f32 / spex_propagate_f32 / spex_propagate_bwd_f32 on it is bracketed by ONE hipEvent pair around all the
 * main SpMM launches of that call (a 3-layer propagation = one bracket of three back-to-back launches), until
 * `capacity` brackets are used.  spex_timer_read synchronises on the recorded events and returns, per bracket, the
 * elapsed milliseconds and the number of SpMM launches inside it.
 */
typedef struct spex_timer spex_timer_t;
int spex_timer_create(int32_t capacity, int32_t every /* bracket every n-th call */, spex_timer_t **out);
int spex_timer_destroy(spex_timer_t *t);
int spex_timer_attach(spex_graph_t *g, spex_timer_t *t /* NULL detaches */);
int spex_timer_read(spex_timer_t *t, float *h_ms, int32_t *h_launches, int32_t max_count, int32_t *count, int reset);

#ifdef __cplusplus
}
#endif
#endif /* SPEX_HIP_H */
