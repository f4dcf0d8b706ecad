// Whole training steps as ONE library call: the fixed launch sequence of a step issued from native code.
//
// Issued launch by launch from Python through ctypes, the exact training step costs the host ~9 us per launch (argument
// marshalling, stream lookup, tensor version bumps) — 12 launches = 110 us of host time for ~75 us of GPU time: the step
// became host-bound once the row-sparse backward had shortened the kernels.  These entry points take the step's buffers
// in a plain-C descriptor and issue the same launches back to back (~1 us each), so the loop is GPU-bound again.
// Nothing here computes: every launch is one of the library's own entry points.
#include <mutex>

#include "spex_common.h"

using namespace spex;

#define SPEX_TRY(call)           \
    do {                         \
        int rc_ = (call);        \
        if (rc_ != SPEX_OK) return rc_; \
    } while (0)

extern "C" int spex_lightgcn_step_bce_f32(spex_lightgcn_step_t *s, const int64_t *users, const int64_t *items,
                                          const float *labels, int32_t B, float *loss_sum, void *stream)
{
    SPEX_CHECK_ARG(s && s->graph && s->graph_t && s->E0 && s->m && s->v && s->light_out && s->ws_fwd && s->lo_batch && s->g_out
                       && s->ws_bwd && s->grad_E0 && s->grad_slots,
                   "spex_lightgcn_step_bce_f32: NULL field in the step descriptor");
    SPEX_CHECK_ARG(users && items && labels && loss_sum && B >= 1, "spex_lightgcn_step_bce_f32: NULL batch pointer or B < 1");
    const spex_graph *g = s->graph, *gt = s->graph_t;
    const int32_t L = s->L, d = s->d, n_u = s->n_user_rows;
    SPEX_CHECK_ARG(g->n_rows == g->n_cols && gt->n_rows == g->n_rows && gt->n_cols == g->n_rows, "spex_lightgcn_step_bce_f32: square graphs of one size");
    SPEX_CHECK_ARG(L >= 1 && d == 64 && n_u >= 0 && n_u <= g->n_rows, "spex_lightgcn_step_bce_f32: L=%d d=%d n_user_rows=%d (needs L >= 1, d == 64)", L, d, n_u);
    SPEX_CHECK_ARG(s->slot_capacity >= 2 * B, "spex_lightgcn_step_bce_f32: slot capacity %d < 2 B = %d", s->slot_capacity, 2 * B);
    // edge dropout (model.py:46-55; set per step on BOTH handles with spex_graph_set_edge_mask — graph_t must then be the transposed
    // handle carrying the edge-id permutation, a masked adjacency is not symmetric): every product of the step uses the handles'
    // mask — the whole-graph launches through spex_spmm_f32, the batch kernel's last layer and push through the same keep rule.
    SPEX_CHECK_ARG(g->mask_mode == gt->mask_mode && (g->mask_mode == 0 || (g->keep_prob == gt->keep_prob && g->seed == gt->seed && g->keep == gt->keep)),
                   "spex_lightgcn_step_bce_f32: graph and graph_t must carry the same edge-dropout mask");
    SPEX_CHECK_ARG(g->mask_mode == 0 || (gt != g && L >= 2),
                   "spex_lightgcn_step_bce_f32: edge dropout needs L >= 2 and graph_t = the transposed handle (with its edge-id permutation)");
    const size_t sz = (size_t)g->n_rows * d;
    const bool det = (s->flags & SPEX_STEP_DETERMINISTIC) != 0;
    // ---- forward: layers 0 .. L-2 over the whole graph; the last layer is taken at the batch's rows only.  For L <= 3 the whole-graph
    //      launches run in the PLAIN form (no epilogue operand, one output stream: 12.3 vs 14.8 us on Epinion2) into the two halves of
    //      ws_fwd, and the batch kernel forms the layer sum (E^0 + E^1 (+ E^2)) + y at the batch's rows — the only rows the loss
    //      reads — in the fused epilogues' order; deeper models keep the running sum fused into the launches.
    const bool plain = L >= 2 && L <= 3;
    const float *cur = s->E0;
    for (int32_t l = 0; l + 1 < L; ++l) {
        float *nxt = s->ws_fwd + (size_t)(l & 1) * sz;
        if (plain) SPEX_TRY(spex_spmm_f32(g, cur, nxt, nullptr, 1.0f, nullptr, nullptr, 1.0f, d, stream));
        else SPEX_TRY(spex_spmm_f32(g, cur, nxt, nullptr, 1.0f, l == 0 ? s->E0 : s->light_out, s->light_out, 1.0f, d, stream));
        cur = nxt;
    }
    // the tables whose rows the batch kernel adds in front of the last layer's product: E^0 (+ E^1 + E^2), or the running sum
    const float *sum0 = plain || L == 1 ? s->E0 : s->light_out;
    const float *sum1 = plain ? s->ws_fwd : nullptr, *sum2 = plain && L == 3 ? s->ws_fwd + sz : nullptr;
    if (det) {
        // ---- SPEX_STEP_DETERMINISTIC: the same forward (same kernel code for the batch's rows: bit-identical scores), but every sum
        //      that the fast path leaves to float atomics is taken in a fixed order — the sample's two gradient rows leave as
        //      per-sample slots, are added per table row in ascending slot order (what the reference's CPU index backward does,
        //      model.py:115-116), and the whole backward runs in pull form (each output row one chain in ascending column order,
        //      like the reference's sparse addmm).  Per-sample losses: head of lo_batch, summed in order by the Adam pass.
        float *loss_rows = s->lo_batch;
        SPEX_TRY(spex::lightgcn_batch_slots_layers(g, cur, sum0, sum1, sum2, (float)(L + 1), users, items, labels, B, n_u,
                                                   1.0f / (float)B, nullptr, loss_rows, s->grad_slots, d, stream));
        SPEX_TRY(spex_reduce_slots_f32(users, B, 0, items, B, n_u, g->n_rows, s->grad_slots, d, 1.0f, s->g_out, 0, d, stream));
        SPEX_TRY(spex_propagate_bwd_f32(gt, s->g_out, s->grad_E0, s->ws_bwd, L, d, stream));
        SPEX_TRY(spex::adam_step_z2(s->E0, s->grad_E0, s->m, s->v, (int64_t)sz, s->t + 1, s->lr, s->beta1, s->beta2, s->eps, s->g_out,
                                    s->ws_bwd /* keeps the fast path's push target all-zero */, stream, loss_rows, B, loss_sum));
        s->t += 1;
        return SPEX_OK;
    }
    if (L >= 2) {
        // ---- the batch-sized middle as one launch: last layer + layer mean at the batch's rows, scores + BCE, gradient rows, and the
        //      first backward product G_{L-1} = (g + A^T g) / (L+1) in push form — over the rows of A itself: (A^T g)[c] = sum_r
        //      A[r, c] g[r] — (g_out and G are all-zero here: the Adam pass below clears them for the next step; first call: the caller)
        float *G = s->ws_bwd;
        const bool all_plain = L == 3;              // (see below: nothing reads the dense d loss / d light_out then — it is not formed)
        SPEX_TRY(spex::lightgcn_batch_layers(g, cur, sum0, sum1, sum2, (float)(L + 1), users, items, labels, B, n_u, 1.0f / (float)B,
                                             1.0f / (float)(L + 1), nullptr, s->grad_slots /* per-sample losses, summed by the Adam pass */,
                                             all_plain ? nullptr : s->g_out, G, d, stream));
        // ---- L-1 pull-form products G_l = g / (L+1) + A^T G_{l+1}.  The last one (l = 0) runs in the PLAIN form: its g / (L+1) term
        //      is added by the Adam pass, which reads g_out anyway to clear it (one epilogue stream less on a 15 us launch).
        //      L == 3 (the reference's depth): BOTH run plain.  With P = G_2 = (g + A^T g) / 4 the gradient is
        //      A^T (A^T P + g/4) + g/4 = A^T (A^T P) + (A^T g / 4 + g / 4) = A^T (A^T P) + P — and P is the push target, which the
        //      Adam pass touches anyway (to clear it): it adds P instead of g / 4.  No epilogue operand left in the backward.
        const float *c2 = G;
        for (int32_t l = L - 2; l >= 0; --l) {
            float *nxt = l == 0 ? s->grad_E0 : s->ws_bwd + (size_t)(1 + ((L - 2 - l) & 1)) * sz;   // ws_bwd[1], [2], [1] ...: never the source, never G
            if (l == 0 || all_plain) SPEX_TRY(spex_spmm_f32(gt, c2, nxt, nullptr, 1.0f, nullptr, nullptr, 1.0f, d, stream));
            else SPEX_TRY(spex_spmm_f32(gt, c2, nxt, s->g_out, (float)(L + 1), nullptr, nullptr, 1.0f, d, stream));
            c2 = nxt;
        }
        SPEX_TRY(spex::adam_step_z2(s->E0, s->grad_E0, s->m, s->v, (int64_t)sz, s->t + 1, s->lr, s->beta1, s->beta2, s->eps,
                                    all_plain ? nullptr : s->g_out /* untouched (all-zero) in the all-plain step */, s->ws_bwd, stream,
                                    s->grad_slots, B, loss_sum, nullptr, all_plain ? G : s->g_out, all_plain ? 1.0f : (float)(L + 1)));
        s->t += 1;
        return SPEX_OK;
    } else {    // L == 1: the last layer at the batch's rows, scoring, then grad = (g + A^T g) / 2 as one pull-form product
        SPEX_TRY(spex_spmm_rowlist_f32(g, cur, users, B, 0, items, B, n_u, nullptr, s->E0, s->lo_batch, (float)(L + 1), d, stream));
        SPEX_TRY(spex_score_bce_slots_f32(s->lo_batch, s->lo_batch + (size_t)n_u * d, d, d, n_u, g->n_rows - n_u, users, items, labels, B, d,
                                          loss_sum, s->g_out, s->g_out + (size_t)n_u * d, 1.0f / (float)B, s->grad_slots, d, stream));
        SPEX_TRY(spex_propagate_bwd_f32(gt, s->g_out, s->grad_E0, s->ws_bwd, L, d, stream));
    }
    // ---- Adam over the whole table (also clears g_out for the next step); t is advanced only once every launch is queued
    SPEX_TRY(spex::adam_step_z2(s->E0, s->grad_E0, s->m, s->v, (int64_t)sz, s->t + 1, s->lr, s->beta1, s->beta2, s->eps, s->g_out,
                                L >= 2 ? s->ws_bwd : nullptr, stream, L >= 2 ? s->grad_slots : nullptr, B, loss_sum));
    s->t += 1;
    return SPEX_OK;
}

// Train() of main_rec.py:30-37 over a whole (pre-shuffled, device-resident) epoch as ONE call: batch k = samples [k B, min((k+1) B, n))
// through spex_lightgcn_step_bce_f32 — the host issues ~6 launches per step and nothing else (the Python loop around the one-call
// step cost 7 us per 70 us step: slicing three tensors, building the call).  Loss sums of the full batches -> loss_full, of the ragged
// last batch -> loss_ragged (the caller forms main_rec.py:36's sum of per-batch MEAN losses from the two).
// keep_prob < 1: edge dropout with the in-kernel sampled mask (model.py:46-55), a fresh one per step — seed (drop_seed << 32) | step,
// step counted from 1 — set on both handles as spex_graph_set_edge_mask(.., 2, ..) would; the handles are left unmasked.
extern "C" int spex_lightgcn_epoch_bce_f32(spex_lightgcn_step_t *s, const int64_t *users, const int64_t *items, const float *labels,
                                           int64_t n, int32_t B, int64_t max_steps, float keep_prob, uint32_t drop_seed, float *loss_full,
                                           float *loss_ragged, void *stream)
{
    SPEX_CHECK_ARG(s && s->graph && s->graph_t, "spex_lightgcn_epoch_bce_f32: NULL step descriptor or graph");
    SPEX_CHECK_ARG(users && items && labels && loss_full && loss_ragged && n >= 0 && B >= 1, "spex_lightgcn_epoch_bce_f32: NULL pointer, n < 0 or B < 1");
    SPEX_CHECK_ARG(keep_prob > 0.0f && keep_prob <= 1.0f, "spex_lightgcn_epoch_bce_f32: keep_prob %g", keep_prob);
    spex_graph *g = const_cast<spex_graph *>(s->graph), *gt = const_cast<spex_graph *>(s->graph_t);
    const bool drop = keep_prob < 1.0f;
    int rc = SPEX_OK;
    int64_t k = 0;
    for (int64_t b0 = 0; b0 < n && rc == SPEX_OK && (max_steps < 0 || k < max_steps); b0 += B, ++k) {
        const int32_t nb = (int32_t)(n - b0 < B ? n - b0 : B);
        if (drop) {
            const uint64_t seed = ((uint64_t)drop_seed << 32) | (uint64_t)(uint32_t)(k + 1);
            rc = spex_graph_set_edge_mask(g, 2, nullptr, keep_prob, seed);
            if (rc == SPEX_OK && gt != g) rc = spex_graph_set_edge_mask(gt, 2, nullptr, keep_prob, seed);
            if (rc != SPEX_OK) break;
        }
        rc = spex_lightgcn_step_bce_f32(s, users + b0, items + b0, labels + b0, nb, nb == B ? loss_full : loss_ragged, stream);
    }
    if (drop) {
        (void)spex_graph_set_edge_mask(g, 0, nullptr, 1.0f, 0);
        if (gt != g) (void)spex_graph_set_edge_mask(gt, 0, nullptr, 1.0f, 0);
    }
    return rc;
}

// train() of NGCF_SPEX/code/main_rec.py:116-131 over a whole pre-shuffled, device-resident epoch as ONE call: batch k = samples
// [k B, min((k+1) B, n)) through spex_ngcf_step_bce_f32 (the default one-layer model; the step advances its own Adam and dropout
// counters) — the host issues the launches and nothing else.  Loss sums as in spex_lightgcn_epoch_bce_f32.
extern "C" int spex_ngcf_epoch_bce_f32(spex_ngcf_step_t *s, const int64_t *users, const int64_t *items, const float *labels, int64_t n,
                                       int32_t B, int64_t max_steps, float *loss_full, float *loss_ragged, void *stream)
{
    SPEX_CHECK_ARG(s && users && items && labels && loss_full && loss_ragged && n >= 0 && B >= 1,
                   "spex_ngcf_epoch_bce_f32: NULL pointer, n < 0 or B < 1");
    int rc = SPEX_OK;
    int64_t k = 0;
    for (int64_t b0 = 0; b0 < n && rc == SPEX_OK && (max_steps < 0 || k < max_steps); b0 += B, ++k) {
        const int32_t nb = (int32_t)(n - b0 < B ? n - b0 : B);
        rc = spex_ngcf_step_bce_f32(s, users + b0, items + b0, labels + b0, nb, nb == B ? loss_full : loss_ragged, stream);
    }
    return rc;
}

// The north-star step — 3-layer propagation + fused BPR-SGD over a batch of triples — as ONE call of L + 1 launches with the
// layer mean left to the BPR kernel: layer 1 in the running-sum form (sum1 = E^0 + E^1), the later layers PLAIN, and the fused
// gather + dot + sigmoid + SGD kernel forms ((sum1 + E^2) + E^3) / (L + 1) at its triples' rows only — the rows of the propagated
// table the step reads.  Same results as spex_propagate_f32 followed by spex_bpr_sgd_step_f32 on its output (the rows are
// bit-identical; the updates land with float atomics in either form).
extern "C" int spex_lightgcn_step_bpr_f32(const spex_graph_t *g, float *E0, float *sum1, float *ws, int32_t n_user_rows, int32_t L,
                                          int32_t d, const int64_t *u, const int64_t *i_pos, const int64_t *i_neg, int64_t T, float lr,
                                          float reg, float *loss_sum, void *stream)
{
    SPEX_CHECK_ARG(g && E0 && sum1 && ws && u && i_pos && i_neg && T >= 0, "spex_lightgcn_step_bpr_f32: NULL argument or T < 0");
    SPEX_CHECK_ARG(g->n_rows == g->n_cols && n_user_rows >= 0 && n_user_rows <= g->n_rows, "spex_lightgcn_step_bpr_f32: square graph, 0 <= n_user_rows <= N");
    SPEX_CHECK_ARG(g->mask_mode == 0, "spex_lightgcn_step_bpr_f32: edge dropout is not supported in the one-call step");
    if (d != 64 || L < 1 || L > 3) {
        spex::set_error("spex_lightgcn_step_bpr_f32: d == 64 and 1 <= L <= 3 only (d = %d, L = %d): use spex_propagate_f32 + spex_bpr_sgd_step_f32", d, L);
        return SPEX_ERR_UNSUPPORTED;
    }
    const float *tables[3];
    SPEX_TRY(spex::propagate_plain(g, E0, sum1, ws, L, d, stream, tables));
    return spex::bpr_sgd_layers(tables[0], tables[1], tables[2], (float)(L + 1), E0, n_user_rows, (int64_t)g->n_rows - n_user_rows, u, i_pos,
                                i_neg, T, lr, reg, loss_sum, stream);
}

// Fork / join events of the two-stream steps (dual-task, NGCF): one pair PER STEP DESCRIPTOR, created on first use and kept in
// the descriptor's ev_fork / ev_join cells (zero-initialised by the caller, released with spex_step_events_release).  A pair
// shared per device — the first version — could be re-recorded by a second stepper driven from another host thread between
// this stepper's record and its wait, so that a join waited for the wrong stream's work.
static int step_events(void **ev_fork, void **ev_join, hipEvent_t *fork_ev, hipEvent_t *join_ev)
{
    if (!*ev_fork) {
        hipEvent_t e = nullptr;
        SPEX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        *ev_fork = e;
    }
    if (!*ev_join) {
        hipEvent_t e = nullptr;
        SPEX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        *ev_join = e;
    }
    *fork_ev = (hipEvent_t)*ev_fork;
    *join_ev = (hipEvent_t)*ev_join;
    return SPEX_OK;
}

extern "C" int spex_step_events_release(void **ev_fork, void **ev_join)
{
    if (ev_fork && *ev_fork) {
        SPEX_HIP(hipEventDestroy((hipEvent_t)*ev_fork));
        *ev_fork = nullptr;
    }
    if (ev_join && *ev_join) {
        SPEX_HIP(hipEventDestroy((hipEvent_t)*ev_join));
        *ev_join = nullptr;
    }
    return SPEX_OK;
}

extern "C" int spex_ngcf_step_bce_f32(spex_ngcf_step_t *s, const int64_t *users, const int64_t *items, const float *labels,
                                      int32_t B, float *loss_sum, void *stream)
{
    SPEX_CHECK_ARG(s && s->graph && s->E0 && s->mE && s->vE && s->W && s->mW && s->vW && s->side && s->g_slots
                       && s->g_side_c && s->g_ego_c && s->gW_parts && s->grad,
                   "spex_ngcf_step_bce_f32: NULL field in the step descriptor");
    SPEX_CHECK_ARG(users && items && labels && loss_sum && B >= 1, "spex_ngcf_step_bce_f32: NULL batch pointer or B < 1");
    const spex_graph *g = s->graph;
    const int32_t d = 64, n = g->n_rows, n_u = s->n_user_rows;
    SPEX_CHECK_ARG(g->n_rows == g->n_cols && n_u >= 0 && n_u <= n, "spex_ngcf_step_bce_f32: square graph, 0 <= n_user_rows <= N");
    SPEX_CHECK_ARG(s->slot_capacity >= 2 * B, "spex_ngcf_step_bce_f32: slot capacity %d < 2 B = %d", s->slot_capacity, 2 * B);
    SPEX_CHECK_ARG(g->mask_mode == 0, "spex_ngcf_step_bce_f32: edge dropout does not apply to NGCF");
    const float *W_gc = s->W, *b_gc = s->W + d * d, *W_bi = s->W + d * d + d, *b_bi = s->W + 2 * d * d + d;
    const int32_t per = 2 * (d * d + d);
    const uint32_t step = (uint32_t)s->dropout_step;
    // ---- forward, at the batch's rows ONLY: the descriptor is a one-layer model, whose loss reads the concatenated table nowhere
    //      else (main_rec.py:96-104) — side = A ego for the 2B rows (spex_spmm_rowlist_f32: the main kernel's sums, bit for bit, for
    //      rows of <= 1 024 entries) and the layer on 16-slot tiles of those rows, which writes [ego | normalised layer output] into
    //      the dense tables at the rows the scoring reads.  (11.4 + 10.6 us of whole-table launches on Epinion2 became 6 + 4.)
    SPEX_TRY(spex_spmm_rowlist_f32(g, s->E0, users, B, 0, items, B, n_u, s->side, nullptr, nullptr, 1.0f, d, stream));
    // ---- scoring + backward on the batch's 2B slots in one launch (per-sample losses go to the head of g_slots: the table's Adam
    //      pass below adds them to loss_sum in a fixed order), then the push-form A^T product into the (all-zero) table gradient.
    //      Default: the launch also runs the layer's FORWARD at those rows — its tiles hold both rows of 8 samples and score from
    //      the layer output they recompute anyway, so the concatenated table is never formed (spex_ngcf_fwd_score_bwd_rows_f32)
    float *loss_rows = s->g_slots;
    SPEX_TRY(spex_ngcf_fwd_score_bwd_rows_f32(s->E0, s->side, W_gc, b_gc, W_bi, b_bi, labels, 1.0f / (float)B, n, d, s->slope, s->p_drop,
                                              s->seed, step, 0, s->pad_row, users, items, B, n_u, loss_rows, s->g_side_c, s->g_ego_c,
                                              s->gW_parts, per, stream));
    // ---- Adam: the table (its pass clears the gradient again), the layer weights (their pass sums the partial blocks).  The
    //      weights' pass needs only the rows backward: with a second stream it runs beside the push-form product and the table's
    //      pass (a one-workgroup-class launch next to two that fill the chip) and is joined at the end of the step.
    //      t / dropout_step are advanced only after every launch of the step has been queued (a failed call leaves them alone).
    const int32_t t_next = s->t + 1;
    auto weight_adam = [&](void *st) -> int {
        return spex_adam_step_sum_f32(s->W, s->gW_parts, spex_ngcf_layer_bwd_rows_parts(2 * B), per, s->mW, s->vW, per, t_next, s->lr,
                                      s->beta1, s->beta2, s->eps, st);
    };
    const bool det = (s->flags & SPEX_STEP_DETERMINISTIC) != 0;
    if (det)
        SPEX_CHECK_ARG(s->graph_t && s->g_side_dense && s->g_ego_dense && s->graph_t->n_rows == n && s->graph_t->n_cols == n
                           && s->graph_t->mask_mode == 0,
                       "spex_ngcf_step_bce_f32: SPEX_STEP_DETERMINISTIC needs graph_t (A^T) and the two dense [N, 64] gradient buffers");
    const bool two_streams = s->side_stream != nullptr && s->side_stream != stream;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    int rc = SPEX_OK;
    bool forked = false;
    if (two_streams) {
        SPEX_TRY(step_events(&s->ev_fork, &s->ev_join, &fork_ev, &join_ev));
        SPEX_HIP(hipEventRecord(fork_ev, (hipStream_t)stream));
        SPEX_HIP(hipStreamWaitEvent((hipStream_t)s->side_stream, fork_ev, 0));
        forked = true;
        rc = weight_adam(s->side_stream);
        if (hipEventRecord(join_ev, (hipStream_t)s->side_stream) != hipSuccess && rc == SPEX_OK) rc = SPEX_ERR_HIP;
    }
    if (rc == SPEX_OK) {
        if (det) {
            // SPEX_STEP_DETERMINISTIC: the slots' compact rows are added per table row in ascending slot order (dense g_side / g_ego,
            // both all-zero outside the batch's rows), A^T g_side is the pull-form product over the rows of A^T with g_ego added in
            // its epilogue, and the touched rows are cleared again — no float atomics (the weight gradients never had any).
            rc = spex_reduce_slots_f32(users, B, 0, items, B, n_u, n, s->g_side_c, d, 1.0f, s->g_side_dense, 0, d, stream);
            if (rc == SPEX_OK) rc = spex_reduce_slots_f32(users, B, 0, items, B, n_u, n, s->g_ego_c, d, 1.0f, s->g_ego_dense, 0, d, stream);
            if (rc == SPEX_OK) rc = spex_spmm_f32(s->graph_t, s->g_side_dense, s->grad, s->g_ego_dense, 1.0f, nullptr, nullptr, 1.0f, d, stream);
            if (rc == SPEX_OK) rc = spex_reduce_slots_f32(users, B, 0, items, B, n_u, n, nullptr, d, 1.0f, s->g_side_dense, 0, d, stream);
            if (rc == SPEX_OK) rc = spex_reduce_slots_f32(users, B, 0, items, B, n_u, n, nullptr, d, 1.0f, s->g_ego_dense, 0, d, stream);
        } else {
            rc = spex_spmm_push_batch_f32(g, users, B, 0, items, B, n_u, s->g_side_c, d, s->g_ego_c, d, 1.0f, s->grad, d, stream);
        }
    }
    // (one stream: the layer weights' update rides in the table's Adam launch as 33 extra workgroups — a launch of its own cost
    //  the step ~5 us of ramp; parts summed in the same order as spex_adam_step_sum_f32 sums them)
    spex::SmallAdam small;
    if (!two_streams) {
        small.p = s->W; small.m = s->mW; small.v = s->vW; small.parts = s->gW_parts;
        small.n_parts = spex_ngcf_layer_bwd_rows_parts(2 * B); small.stride = per; small.n = per;
    }
    if (rc == SPEX_OK)
        rc = spex::adam_step_z2(s->E0, s->grad, s->mE, s->vE, (int64_t)n * d, t_next, s->lr, s->beta1, s->beta2, s->eps, s->grad, nullptr, stream,
                                loss_rows, B, loss_sum, two_streams ? nullptr : &small);
    if (forked && hipStreamWaitEvent((hipStream_t)stream, join_ev, 0) != hipSuccess && rc == SPEX_OK) rc = SPEX_ERR_HIP;   // joined on every path
    if (rc != SPEX_OK) return rc;
    s->t = t_next;
    if (s->p_drop > 0.0f) s->dropout_step += 1;
    return SPEX_OK;
}

// NGCF with L >= 2 layers: see include/spex_hip.h (spex_ngcf_deep_step_t).  What NGCFStepper.step issued launch by launch from Python
// (~16 launches at ~9 us of host time each) as one native sequence, with the last layer's forward taken at the batch's rows only.
extern "C" int spex_ngcf_deep_step_bce_f32(spex_ngcf_deep_step_t *s, const int64_t *users, const int64_t *items, const float *labels,
                                           int32_t B, float *loss_sum, void *stream)
{
    const char *who = "spex_ngcf_deep_step_bce_f32";
    SPEX_CHECK_ARG(s && s->graph && s->graph_t && s->E0 && s->mE && s->vE && s->W && s->mW && s->vW && s->gW && s->all_emb && s->g_all && s->sides
                       && s->egos && s->g_slots && s->g_side_c && s->g_ego_c && s->gW_parts && s->g_side && s->g_ego && s->g_next && s->p_drop,
                   "%s: NULL field in the step descriptor", who);
    SPEX_CHECK_ARG(users && items && labels && loss_sum && B >= 1, "%s: NULL batch pointer or B < 1", who);
    const spex_graph *g = s->graph, *gt = s->graph_t;
    const int32_t d = 64, n = g->n_rows, n_u = s->n_user_rows, L = s->L, ld = d * (L + 1), per = 2 * (d * d + d);
    SPEX_CHECK_ARG(L >= 2 && L <= 8, "%s: L = %d (two to eight layers; one layer: spex_ngcf_step_bce_f32)", who, L);
    SPEX_CHECK_ARG(g->n_rows == g->n_cols && gt->n_rows == n && gt->n_cols == n && n_u >= 0 && n_u <= n, "%s: square graphs of one size, 0 <= n_user_rows <= N", who);
    SPEX_CHECK_ARG(s->slot_capacity >= 2 * B, "%s: slot capacity %d < 2 B = %d", who, s->slot_capacity, 2 * B);
    SPEX_CHECK_ARG(g->mask_mode == 0 && gt->mask_mode == 0, "%s: edge dropout does not apply to NGCF", who);
    const size_t sz = (size_t)n * d;
    const uint32_t step = (uint32_t)s->dropout_step;
    bool any_drop = false;
    for (int32_t l = 0; l < L; ++l) {
        SPEX_CHECK_ARG(s->p_drop[l] >= 0.0f && s->p_drop[l] < 1.0f, "%s: p_drop[%d] = %f", who, l, (double)s->p_drop[l]);
        any_drop = any_drop || s->p_drop[l] > 0.0f;
    }
    auto Wl = [&](int32_t l, const float *&W_gc, const float *&b_gc, const float *&W_bi, const float *&b_bi) {
        const float *w = s->W + (size_t)l * per;
        W_gc = w; b_gc = w + d * d; W_bi = w + d * d + d; b_bi = w + 2 * d * d + d;
    };
    auto ego_of = [&](int32_t l) -> float * { return l == 0 ? s->E0 : s->egos + (size_t)(l - 1) * sz; };
    const float *W_gc, *b_gc, *W_bi, *b_bi;
    // ---- forward: whole-table layers 0 .. L-2 (each writes [ego |] its normalised output into the concatenated table and its
    //      un-normalised output = the next layer's input), then the last layer at the batch's rows only
    for (int32_t l = 0; l + 1 < L; ++l) {
        float *side = s->sides + (size_t)l * sz;
        SPEX_TRY(spex_spmm_f32(g, ego_of(l), side, nullptr, 1.0f, nullptr, nullptr, 1.0f, d, stream));
        Wl(l, W_gc, b_gc, W_bi, b_bi);
        SPEX_TRY(spex_ngcf_layer_fwd_f32(ego_of(l), side, W_gc, b_gc, W_bi, b_bi, s->all_emb + (size_t)d * l, ld, l == 0 ? 1 : 0, ego_of(l + 1), n, d,
                                         s->slope, s->p_drop[l], s->seed, step, (uint32_t)l, s->pad_row, stream));
    }
    {
        const int32_t l = L - 1;
        float *side = s->sides + (size_t)l * sz;
        SPEX_TRY(spex_spmm_rowlist_f32(g, ego_of(l), users, B, 0, items, B, n_u, side, nullptr, nullptr, 1.0f, d, stream));
        Wl(l, W_gc, b_gc, W_bi, b_bi);
        SPEX_TRY(spex_ngcf_layer_fwd_rows_f32(ego_of(l), side, W_gc, b_gc, W_bi, b_bi, s->all_emb + (size_t)d * l, ld, 0, n, d, s->slope, s->p_drop[l],
                                              s->seed, step, (uint32_t)l, s->pad_row, users, B, 0, items, B, n_u, stream));
    }
    // ---- scoring on the concatenated rows: per-sample gradient rows (the last layer's backward) + the table form (earlier layers)
    SPEX_TRY(spex_score_bce_slots_f32(s->all_emb, s->all_emb + (size_t)n_u * ld, ld, ld, n_u, n - n_u, users, items, labels, B, ld, loss_sum,
                                      s->g_all, s->g_all + (size_t)n_u * ld, 1.0f / (float)B, s->g_slots, ld, stream));
    // ---- backward: the last layer on the batch's slots, A^T g_side in push form, then the dense layers
    float *g_next = s->g_next + (size_t)((L - 1) & 1) * sz;
    {
        const int32_t l = L - 1;
        Wl(l, W_gc, b_gc, W_bi, b_bi);
        SPEX_TRY(spex_ngcf_layer_bwd_rows_f32(ego_of(l), s->sides + (size_t)l * sz, W_gc, b_gc, W_bi, b_bi, s->g_slots + (size_t)d * (l + 1), ld, nullptr,
                                              nullptr, ld, n, d, s->slope, s->p_drop[l], s->seed, step, (uint32_t)l, s->pad_row, users, B, 0, items, B,
                                              n_u, s->g_side_c, s->g_ego_c, s->gW_parts, per, stream));
        SPEX_TRY(spex::zero_f32(g_next, (int64_t)sz, stream));
        SPEX_TRY(spex_spmm_push_batch_f32(g, users, B, 0, items, B, n_u, s->g_side_c, d, s->g_ego_c, d, 1.0f, g_next, d, stream));
    }
    for (int32_t l = L - 2; l >= 0; --l) {
        Wl(l, W_gc, b_gc, W_bi, b_bi);
        float *gw = s->gW + (size_t)l * per;
        SPEX_TRY(spex_ngcf_layer_bwd_f32(ego_of(l), s->sides + (size_t)l * sz, W_gc, b_gc, W_bi, b_bi, s->g_all + (size_t)d * (l + 1), ld, g_next,
                                         l == 0 ? s->g_all : nullptr, ld, n, d, s->slope, s->p_drop[l], s->seed, step, (uint32_t)l, s->pad_row,
                                         s->g_side, s->g_ego, gw, gw + d * d, gw + d * d + d, gw + 2 * d * d + d, stream));
        float *out = s->g_next + (size_t)(l & 1) * sz;
        SPEX_TRY(spex_spmm_f32(gt, s->g_side, out, s->g_ego, 1.0f, nullptr, nullptr, 1.0f, d, stream));
        g_next = out;
    }
    // ---- Adam: the table, the last layer's weights (from their partial blocks), the other layers' (their pass clears gW); the table
    //      form of the scoring gradient is cleared for the next step
    const int32_t t_next = s->t + 1;
    SPEX_TRY(spex::adam_step_z2(s->E0, g_next, s->mE, s->vE, (int64_t)sz, t_next, s->lr, s->beta1, s->beta2, s->eps, nullptr, nullptr, stream));
    const size_t lo = (size_t)(L - 1) * per;
    SPEX_TRY(spex_adam_step_sum_f32(s->W + lo, s->gW_parts, spex_ngcf_layer_bwd_rows_parts(2 * B), per, s->mW + lo, s->vW + lo, per, t_next, s->lr,
                                    s->beta1, s->beta2, s->eps, stream));
    SPEX_TRY(spex::adam_step_z2(s->W, s->gW, s->mW, s->vW, (int64_t)lo, t_next, s->lr, s->beta1, s->beta2, s->eps, s->gW, nullptr, stream));
    SPEX_TRY(spex::zero_f32(s->g_all, (int64_t)n * ld, stream));
    s->t = t_next;
    if (any_drop) s->dropout_step += 1;
    return SPEX_OK;
}

extern "C" int spex_dual_task_step_join(spex_dual_task_step_t *s, void *stream)
{
    SPEX_CHECK_ARG(s, "spex_dual_task_step_join: NULL descriptor");
    if (s->side_pending && s->ev_join) SPEX_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)s->ev_join, 0));
    s->side_pending = 0;
    return SPEX_OK;
}

extern "C" int spex_dual_task_step_f32(spex_dual_task_step_t *s, const int64_t *users, const int64_t *items, const float *labels,
                                       int32_t B, const int64_t *seq, const int64_t *seq_l, const int64_t *targets, int32_t T,
                                       void *stream)
{
    SPEX_CHECK_ARG(s && s->graph && s->graph_t && s->params && s->m && s->v && s->light && s->ws_fwd && s->lo_batch && s->g_prop
                       && s->g_raw && s->g_E0 && s->ws_bwd && s->mixed_slots && s->grad_slots && s->g_prop_slots && s->arange && s->g_user
                       && s->g_small && s->a2 && s->trust_ws && s->dscore && s->loss_b && s->loss && s->loss_acc && s->precision,
                   "spex_dual_task_step_f32: NULL field in the step descriptor");
    SPEX_CHECK_ARG(users && items && labels && B >= 1, "spex_dual_task_step_f32: NULL batch pointer or B < 1");
    SPEX_CHECK_ARG(T >= 0 && T <= s->path_capacity && (T == 0 || (seq && seq_l && targets)),
                   "spex_dual_task_step_f32: T=%d paths (capacity %d) or NULL path pointer", T, s->path_capacity);
    SPEX_CHECK_ARG(s->slot_capacity >= 2 * B, "spex_dual_task_step_f32: slot capacity %d < 2 B = %d", s->slot_capacity, 2 * B);
    const spex_graph *g = s->graph, *gt = s->graph_t;
    const int32_t d = 64, N = g->n_rows, n_u = s->n_user_rows, L = s->L, H = s->n_heads;
    SPEX_CHECK_ARG(g->n_rows == g->n_cols && gt->n_rows == N && gt->n_cols == N, "spex_dual_task_step_f32: square graphs of one size");
    SPEX_CHECK_ARG(s->d == 64 && L >= 1 && n_u >= 2 && n_u <= N, "spex_dual_task_step_f32: d=%d L=%d n_user_rows=%d (needs d == 64)", s->d, L, n_u);
    // edge dropout on the rec branch (model_expert_s.py:104-109; the trust branch reads the raw user table): as in the LightGCN step —
    // the same mask on both handles, graph_t the transposed handle with the edge-id permutation, L >= 2
    SPEX_CHECK_ARG(g->mask_mode == gt->mask_mode && (g->mask_mode == 0 || (g->keep_prob == gt->keep_prob && g->seed == gt->seed && g->keep == gt->keep)),
                   "spex_dual_task_step_f32: graph and graph_t must carry the same edge-dropout mask");
    SPEX_CHECK_ARG(g->mask_mode == 0 || (gt != g && L >= 2),
                   "spex_dual_task_step_f32: edge dropout needs L >= 2 and graph_t = the transposed handle (with its edge-id permutation)");
    const int64_t n_trust = spex_trust_param_count(d, H);
    SPEX_CHECK_ARG(n_trust > 0, "spex_dual_task_step_f32: unsupported number of heads %d", H);
    const bool det = (s->flags & SPEX_STEP_DETERMINISTIC) != 0;
    if (det)
        SPEX_CHECK_ARG(s->g_raw_slots && s->att_parts && s->loss_rows,
                       "spex_dual_task_step_f32: SPEX_STEP_DETERMINISTIC needs g_raw_slots, att_parts and loss_rows");
    const size_t sz = (size_t)N * d, off_u = (size_t)n_u * d;
    float *E0 = s->params, *trust_p = E0 + sz, *att1 = trust_p + n_trust, *att2 = att1 + 4 * d;
    float *g_att1 = s->g_small + n_trust, *g_att2 = g_att1 + 4 * d;
    // ---- trust branch (:170-192) on the raw user table, forward + backward.  It shares nothing with the rec branch but the
    //      (read-only) parameters, so with a second stream it is issued FIRST, behind the previous step's Adam pass, and its
    //      <= path_capacity workgroups run beside the rec branch's launches; the Adam pass waits for both.
    //      (Beside the rec branch the head keeps to ONE workgroup per path: the rec branch's whole-graph launches bound the step and
    //      want every CU; alone on the caller's stream it splits each path's sweep over up to eight — except in the deterministic
    //      step, whose results must not depend on how the streams are arranged: the fold over shares rounds differently.)
    auto trust_branch = [&](void *st) -> int {
        return spex::trust_head_train(E0, n_u, trust_p, seq, seq_l, targets, T, s->path_len, d, H, s->hybrid, 1.0f, nullptr, s->a2, s->dscore,
                                      s->loss_b, s->trust_ws, s->loss + 1, 0, s->g_small, s->g_user, st == stream && !det ? 8 : 1, st);
    };
    // SPEX_STEP_PIPELINED: the trust branch and the update of what it reads (user rows, trust block, task weights) live on
    // side_stream from one step to the next; `stream` and side_stream exchange two events per step, neither on the trust
    // branch's cycle.  side_pending: side_stream holds work `stream` has not been ordered behind yet.
    const bool pipelined = (s->flags & SPEX_STEP_PIPELINED) != 0 && s->side_stream != nullptr && s->side_stream != stream;
    if (s->side_pending && !pipelined) SPEX_TRY(spex_dual_task_step_join(s, stream));      // (the flags changed between two steps)
    const bool two_streams = !pipelined && s->side_stream != nullptr && s->side_stream != stream && T > 0;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    int rc_trust = SPEX_OK;
    bool forked = false;
    if (pipelined) {
        SPEX_TRY(step_events(&s->ev_fork, &s->ev_join, &fork_ev, &join_ev));
        if (!s->side_pending) {            // first step after a join: side_stream starts behind everything queued on `stream`
            SPEX_HIP(hipEventRecord(fork_ev, (hipStream_t)stream));
            SPEX_HIP(hipStreamWaitEvent((hipStream_t)s->side_stream, fork_ev, 0));
        } else {                           // the previous step's side update (user rows!) in front of this step's rec branch
            SPEX_HIP(hipStreamWaitEvent((hipStream_t)stream, join_ev, 0));
        }
        s->side_pending = 1;
        if (T > 0) rc_trust = trust_branch(s->side_stream);
    }
    if (two_streams) {
        SPEX_TRY(step_events(&s->ev_fork, &s->ev_join, &fork_ev, &join_ev));
        SPEX_HIP(hipEventRecord(fork_ev, (hipStream_t)stream));
        SPEX_HIP(hipStreamWaitEvent((hipStream_t)s->side_stream, fork_ev, 0));
        forked = true;
        rc_trust = trust_branch(s->side_stream);
        if (hipEventRecord(join_ev, (hipStream_t)s->side_stream) != hipSuccess && rc_trust == SPEX_OK) rc_trust = SPEX_ERR_HIP;
    }
    // the fused batch kernel's copies of the two gate gradients live in grad_slots (per-sample rows on the other paths): up to 64
    // copies of 512 floats, summed and cleared by the Adam pass — which clears the area after the other paths too, so that a
    // descriptor may change its flags between steps
    const int32_t att_copies_max = s->slot_capacity / 8 < 64 ? s->slot_capacity / 8 : 64;
    int32_t att_copies_used = 0;
    bool fused_middle_used = false;
    bool plain_last = false;          // the rec branch left the g_prop / (L+1) share of its last backward product to the Adam pass
    auto rec_branch = [&]() -> int {
        // ---- rec branch forward (model_expert_s.py:95-126,154-168): layers 1 .. L-1 over the whole graph, the last layer, the gate and
        //      the scores only at the batch's rows
        //      (whole-graph launches in the plain form for L <= 3, the layer sum formed at the batch's rows: see the LightGCN step)
        const bool plain = L >= 2 && L <= 3;
        const float *cur = E0;
        for (int32_t l = 0; l + 1 < L; ++l) {
            float *nxt = s->ws_fwd + (size_t)(l & 1) * sz;
            if (plain) SPEX_TRY(spex_spmm_f32(g, cur, nxt, nullptr, 1.0f, nullptr, nullptr, 1.0f, d, stream));
            else SPEX_TRY(spex_spmm_f32(g, cur, nxt, nullptr, 1.0f, l == 0 ? E0 : s->light, s->light, 1.0f, d, stream));
            cur = nxt;
        }
        if (!det && L >= 2) {
            // (fast path: last layer at the batch's rows + layer mean + gate + scores + the gate's backward + the first backward
            //  product in push form — ONE launch, batch.hip: gated_batch_push_kernel; then the L-1 pull-form launches on A^T, the
            //  last one plain: the Adam pass adds its g_prop / (L+1) share, see the LightGCN step)
            float *G = s->ws_bwd;                     // all-zero here (cleared by the previous step's Adam pass)
            SPEX_TRY(spex::gated_batch_push_layers(g, cur, plain ? E0 : s->light, plain ? s->ws_fwd : nullptr,
                                                   plain && L == 3 ? s->ws_fwd + sz : nullptr, (float)(L + 1), E0, att1, att2, users, items,
                                                   labels, B, n_u, 1.0f / (float)B, 1.0f / (float)(L + 1), s->loss,
                                                   L == 3 ? nullptr : s->g_prop /* L == 3: nothing reads it (all-plain backward) */, G, s->g_raw,
                                                   att_copies_max >= 1 ? s->grad_slots : g_att1, att_copies_max >= 1 ? att_copies_max : 1,
                                                   d, stream));
            att_copies_used = att_copies_max;
            fused_middle_used = true;
            const float *c2 = G;
            for (int32_t l = L - 2; l >= 0; --l) {
                float *nxt = l == 0 ? s->g_E0 : s->ws_bwd + (size_t)(1 + ((L - 2 - l) & 1)) * sz;
                // (L == 3: both products plain, the Adam pass adds the push target P instead of g_prop / 4 — see the LightGCN step)
                if (l == 0 || L == 3) SPEX_TRY(spex_spmm_f32(gt, c2, nxt, nullptr, 1.0f, nullptr, nullptr, 1.0f, d, stream));
                else SPEX_TRY(spex_spmm_f32(gt, c2, nxt, s->g_prop, (float)(L + 1), nullptr, nullptr, 1.0f, d, stream));
                c2 = nxt;
            }
            plain_last = true;
            return SPEX_OK;
        }
        // (L == 1 and the deterministic step: last layer at the batch's rows + layer mean + gate + scores + per-sample gradient
        //  rows in one launch, the gate's backward and the propagation's after it)
        SPEX_TRY(spex::gated_batch_fwd_layers(g, cur, plain || L == 1 ? E0 : s->light, plain ? s->ws_fwd : nullptr,
                                              plain && L == 3 ? s->ws_fwd + sz : nullptr, (float)(L + 1), E0, att1, att2, users, items,
                                              labels, B, n_u, 1.0f / (float)B, s->loss, det ? s->loss_rows : nullptr, s->lo_batch,
                                              s->grad_slots, d, stream));
        if (det) {
            // ---- SPEX_STEP_DETERMINISTIC: every sum the fast path leaves to float atomics is taken in a fixed order — the loss in
            //      sample order, the gate's backward into per-slot rows + per-workgroup blocks of the two gate gradients (added in
            //      block order), the slots per table row in ascending slot order, the propagation's backward in pull form.
            SPEX_TRY(spex::sum_ordered(s->loss_rows, B, 1.0f, s->loss, 1, stream));
            SPEX_TRY(spex_expert_gate_rows_bwd_det_f32(E0, s->lo_batch, att1, att2, users, B, 0, items, B, n_u, n_u, N, d, s->grad_slots, d,
                                                       s->g_prop_slots, s->g_raw_slots, s->att_parts, stream));
            const int32_t n_parts = spex_expert_gate_rows_bwd_parts(2 * B);
            SPEX_TRY(spex::sum_parts(s->att_parts, n_parts, 512, 256, g_att1, 1, stream));
            SPEX_TRY(spex::sum_parts(s->att_parts + 256, n_parts, 512, 256, g_att2, 1, stream));
            SPEX_TRY(spex_reduce_slots_f32(users, B, 0, items, B, n_u, N, s->g_prop_slots, d, 1.0f, s->g_prop, 0, d, stream));
            SPEX_TRY(spex_reduce_slots_f32(users, B, 0, items, B, n_u, N, s->g_raw_slots, d, 1.0f, s->g_raw, 0, d, stream));
            SPEX_TRY(spex_propagate_bwd_f32(gt, s->g_prop, s->g_E0, s->ws_bwd, L, d, stream));
            return SPEX_OK;
        }
        // ---- L == 1: rec branch backward (gradients of the UNWEIGHTED loss1; the precisions are applied in the Adam pass): the gate slot
        //      by slot (dense d loss / d light and d loss / d E0 rows added with atomics), then the one pull-form product
        SPEX_TRY(spex_expert_gate_rows_bwd_f32(E0, s->lo_batch, att1, att2, users, B, 0, items, B, n_u, n_u, N, d, s->grad_slots, d,
                                               s->g_prop_slots, s->g_prop, s->g_raw, g_att1, g_att2, stream));
        SPEX_TRY(spex_propagate_bwd_f32(gt, s->g_prop, s->g_E0, s->ws_bwd, L, d, stream));
        return SPEX_OK;
    };
    int rc = rec_branch();
    if (forked && hipStreamWaitEvent((hipStream_t)stream, join_ev, 0) != hipSuccess && rc == SPEX_OK) rc = SPEX_ERR_HIP;   // joined on every path
    if (rc == SPEX_OK) rc = rc_trust;
    // what the Adam pass adds to the last (plain) backward product: g_prop / (L+1), or — L == 3, both products plain — the push
    // target itself (prop_div < 0)
    const float prop_div = !plain_last ? 0.0f : (L == 3 ? -1.0f : (float)(L + 1));
    const int32_t clear_prop = fused_middle_used && L == 3 ? 0 : 1;      // (the fused middle of the L == 3 step never wrote g_prop)
    if (pipelined) {
        // the rec branch's gradients -> side_stream; Adam part 2 (user rows, trust block, task weights, loss cells) there, part 1
        // (item rows, gate matrices) here.  The join event is recorded on every path so that a later join never waits in vain.
        if (hipEventRecord(fork_ev, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent((hipStream_t)s->side_stream, fork_ev, 0) != hipSuccess)
            rc = rc == SPEX_OK ? SPEX_ERR_HIP : rc;
        for (int32_t part = 2; part >= 1 && rc == SPEX_OK; --part)
            rc = spex::dual_task_adam(s->params, s->m, s->v, s->g_E0, s->g_raw, s->g_user, s->g_small, s->g_prop, L >= 2 ? s->ws_bwd : nullptr,
                                      s->loss, s->loss_acc, s->precision, (int64_t)sz, (int64_t)off_u, n_trust, B, T, s->n_rec, s->t + 1, s->lr,
                                      s->beta1, s->beta2, s->eps, (s->flags & SPEX_STEP_FIXED_TASK_WEIGHTS) != 0,
                                      part == 2 ? s->side_stream : stream, prop_div, s->grad_slots, att_copies_used,
                                      att_copies_max * 512, part, clear_prop);
        if (hipEventRecord(join_ev, (hipStream_t)s->side_stream) != hipSuccess && rc == SPEX_OK) rc = SPEX_ERR_HIP;
        if (rc != SPEX_OK) {
            (void)spex_dual_task_step_join(s, stream);
            return rc;
        }
        s->t += 1;
        return SPEX_OK;
    }
    if (rc == SPEX_OK && T > 0 && !two_streams) rc = trust_branch(stream);                // one-stream order
    if (rc != SPEX_OK) return rc;
    // ---- uncertainty-weighted sum of both losses (main_auto_expert_s.py:78-82; SPEX_STEP_FIXED_TASK_WEIGHTS: loss1 + loss2,
    //      main_11.py:69) + Adam over every parameter (:89); t is advanced once the whole step is queued
    SPEX_TRY(spex::dual_task_adam(s->params, s->m, s->v, s->g_E0, s->g_raw, s->g_user, s->g_small, s->g_prop, L >= 2 ? s->ws_bwd : nullptr,
                                  s->loss, s->loss_acc, s->precision, (int64_t)sz, (int64_t)off_u, n_trust, B, T, s->n_rec, s->t + 1, s->lr,
                                  s->beta1, s->beta2, s->eps, (s->flags & SPEX_STEP_FIXED_TASK_WEIGHTS) != 0, stream,
                                  prop_div, s->grad_slots, att_copies_used, att_copies_max * 512, 0, clear_prop));
    s->t += 1;
    return SPEX_OK;
}

// Train() of main_auto_expert_s.py:60-91 over a whole pre-shuffled, device-resident epoch as ONE call: batch k = samples [k B, (k+1) B)
// with the paths [path_off[k], path_off[k+1]) of the epoch's staged path arrays (seq [*, path_len], seq_l, targets: the reference's
// per-batch selection, made up front by the host) through spex_dual_task_step_f32.  Both losses accumulate in the descriptor's
// loss_acc.  keep_prob < 1: the rec branch's sampled edge mask, a fresh one per step (as spex_lightgcn_epoch_bce_f32).
extern "C" int spex_dual_task_epoch_f32(spex_dual_task_step_t *s, const int64_t *users, const int64_t *items, const float *labels, int64_t n,
                                        int32_t B, int64_t max_steps, const int64_t *seq, const int64_t *seq_l, const int64_t *targets,
                                        const int64_t *path_off, float keep_prob, uint32_t drop_seed, void *stream)
{
    SPEX_CHECK_ARG(s && s->graph && s->graph_t && users && items && labels && path_off && n >= 0 && B >= 1,
                   "spex_dual_task_epoch_f32: NULL pointer, n < 0 or B < 1");
    SPEX_CHECK_ARG(keep_prob > 0.0f && keep_prob <= 1.0f, "spex_dual_task_epoch_f32: keep_prob %g", keep_prob);
    spex_graph *g = const_cast<spex_graph *>(s->graph), *gt = const_cast<spex_graph *>(s->graph_t);
    const bool drop = keep_prob < 1.0f;
    int rc = SPEX_OK;
    int64_t k = 0;
    for (int64_t b0 = 0; b0 < n && rc == SPEX_OK && (max_steps < 0 || k < max_steps); b0 += B, ++k) {
        const int32_t nb = (int32_t)(n - b0 < B ? n - b0 : B);
        const int64_t p0 = path_off[k], T = path_off[k + 1] - p0;
        if (T < 0 || (T > 0 && !(seq && seq_l && targets))) {
            spex::set_error("spex_dual_task_epoch_f32: batch %lld has %lld paths (offsets must ascend; path arrays must be given)", (long long)k,
                            (long long)T);
            rc = SPEX_ERR_INVALID;
            break;
        }
        if (drop) {
            const uint64_t seed = ((uint64_t)drop_seed << 32) | (uint64_t)(uint32_t)(k + 1);
            rc = spex_graph_set_edge_mask(g, 2, nullptr, keep_prob, seed);
            if (rc == SPEX_OK && gt != g) rc = spex_graph_set_edge_mask(gt, 2, nullptr, keep_prob, seed);
            if (rc != SPEX_OK) break;
        }
        rc = spex_dual_task_step_f32(s, users + b0, items + b0, labels + b0, nb, T ? seq + p0 * s->path_len : nullptr, T ? seq_l + p0 : nullptr,
                                     T ? targets + p0 : nullptr, (int32_t)T, stream);
    }
    if (drop) {
        (void)spex_graph_set_edge_mask(g, 0, nullptr, 1.0f, 0);
        if (gt != g) (void)spex_graph_set_edge_mask(gt, 0, nullptr, 1.0f, 0);
    }
    return rc;
}
