/*
 * spex_oracle.c — CPU restatement of the SPEX LightGCN/NGCF hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  Nothing under spex_amd/ links or calls it; the product path has no CPU
 * fallback and raises when libspexhip.so is missing.
 *
 * Parity pin: every function here is checked against golden vectors minted from the reference itself
 * (oracle/gen_golden.py -> the .npz files in tests/golden; tests/test_oracle_golden.py).
 *
 * Third-party arithmetic: the sparse product the reference calls, torch.sparse.mm (LightGCN_SPEX/code/utility1/
 * model.py:91, NGCF_SPEX/code/main_rec.py:76), is ATen's CPU sparse addmm, not a file under /root/reference.  The
 * reference pins torch 1.5.1 (README.md:44, no lock file); goldens were minted with torch 2.10.0.  Its published
 * algorithm: result rows zero-initialised, then for every stored entry of the coalesced (row-major, ascending
 * column) operand `out[row,:] += val * dense[col,:]` as one axpy per entry — restated below as a sequential fmaf
 * chain per output element, ascending column order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Y[n_rows,d] = A[n_rows,*] (CSR) * X[*,d].   model.py:91 / NGCF main_rec.py:76 */
void spex_oracle_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val, int32_t n_rows,
                              const float *X, float *Y, int32_t d, int32_t n_threads)
{
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (int32_t r = 0; r < n_rows; ++r) {
        float *y = Y + (size_t)r * d;
        for (int32_t k = 0; k < d; ++k) y[k] = 0.0f;
        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            const float a = val[e];
            const float *x = X + (size_t)col[e] * d;
            for (int32_t k = 0; k < d; ++k) y[k] = fmaf(a, x[k], y[k]);
        }
    }
}

/* Same product with a per-entry keep mask and 1/keep rescale: model.py:46-55 (__dropout_x) followed by model.py:91.
 * Kept entries stay in their original (ascending-column) order, dropped entries simply vanish. */
void spex_oracle_spmm_csr_masked_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                     const uint8_t *keep, float inv_keep_divisor, int32_t n_rows, const float *X,
                                     float *Y, int32_t d)
{
    for (int32_t r = 0; r < n_rows; ++r) {
        float *y = Y + (size_t)r * d;
        for (int32_t k = 0; k < d; ++k) y[k] = 0.0f;
        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            if (!keep[e]) continue;
            const float a = val[e] / inv_keep_divisor; /* values[random_index] / keep_prob, model.py:53 */
            const float *x = X + (size_t)col[e] * d;
            for (int32_t k = 0; k < d; ++k) y[k] = fmaf(a, x[k], y[k]);
        }
    }
}

/* LightGCN.computer(), model.py:66-97: E^{l+1} = A E^l for l < L; out = mean(E^0..E^L) (stack + mean over dim 1:
 * ((E0+E1)+E2)+... then one division by L+1).  layers_out (may be NULL) receives E^1..E^L back to back. */
void spex_oracle_propagate_mean_f32(const int32_t *rowptr, const int32_t *col, const float *val, int32_t n,
                                    const float *E0, int32_t L, int32_t d, float *layers_out, float *mean_out,
                                    int32_t n_threads)
{
    const size_t sz = (size_t)n * d;
    float *cur = (float *)malloc(sz * sizeof(float));
    float *nxt = (float *)malloc(sz * sizeof(float));
    memcpy(cur, E0, sz * sizeof(float));
    memcpy(mean_out, E0, sz * sizeof(float));
    for (int32_t l = 0; l < L; ++l) {
        spex_oracle_spmm_csr_f32(rowptr, col, val, n, cur, nxt, d, n_threads);
        if (layers_out) memcpy(layers_out + (size_t)l * sz, nxt, sz * sizeof(float));
        for (size_t i = 0; i < sz; ++i) mean_out[i] += nxt[i];
        float *t = cur; cur = nxt; nxt = t;
    }
    const float cnt = (float)(L + 1);
    for (size_t i = 0; i < sz; ++i) mean_out[i] = mean_out[i] / cnt;
    free(cur);
    free(nxt);
}

/* LightGCN.forward, model.py:111-121: gamma_b = <users[u_b], items[i_b]>; loss = mean BCEWithLogits(gamma, y).
 * Also the gradient of the loss w.r.t. the two propagated tables (dense, zero except the batch rows), which is what
 * autograd hands back to the propagation (main_rec.py:35).  gamma is accumulated in double and rounded once: the
 * reference's torch.sum uses a vectorised tree whose order is an ATen detail; tests allow 1e-6 relative on it. */
void spex_oracle_score_bce_f32(const float *users, const float *items, const int64_t *u_idx, const int64_t *i_idx,
                               const float *labels, int32_t B, int32_t d, float *gamma, float *loss,
                               float *grad_users, int32_t n_users_rows, float *grad_items, int32_t n_items_rows)
{
    double acc_loss = 0.0;
    if (grad_users) memset(grad_users, 0, (size_t)n_users_rows * d * sizeof(float));
    if (grad_items) memset(grad_items, 0, (size_t)n_items_rows * d * sizeof(float));
    for (int32_t b = 0; b < B; ++b) {
        const float *u = users + (size_t)u_idx[b] * d;
        const float *it = items + (size_t)i_idx[b] * d;
        double g = 0.0;
        for (int32_t k = 0; k < d; ++k) g += (double)(u[k] * it[k]);
        const float x = (float)g;
        gamma[b] = x;
        if (labels) {
            const float y = labels[b];
            /* max(x,0) - x*y + log1p(exp(-|x|))  (ATen binary_cross_entropy_with_logits) */
            acc_loss += (double)((x > 0 ? x : 0.0f) - x * y + log1pf(expf(-fabsf(x))));
            if (grad_users && grad_items) {
                const float s = 1.0f / (1.0f + expf(-x));
                const float dg = (s - y) / (float)B;
                float *gu = grad_users + (size_t)u_idx[b] * d;
                float *gi = grad_items + (size_t)i_idx[b] * d;
                for (int32_t k = 0; k < d; ++k) {
                    gu[k] += dg * it[k];
                    gi[k] += dg * u[k];
                }
            }
        }
    }
    if (loss) *loss = (float)(acc_loss / (double)B);
}

/* torch.optim.Adam single-tensor step (main_rec.py:23,37; defaults betas 0.9/0.999, eps 1e-8, no weight decay):
 *   m.lerp_(g, 1-b1); v = v*b2 + (1-b2) g*g; denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m/denom */
void spex_oracle_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n, int32_t t, float lr,
                               float beta1, float beta2, float eps)
{
    const double bc1 = 1.0 - pow((double)beta1, (double)t);
    const double bc2 = 1.0 - pow((double)beta2, (double)t);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float w1 = 1.0f - beta1, w2 = 1.0f - beta2;
    for (int64_t i = 0; i < n; ++i) {
        m[i] = m[i] + w1 * (g[i] - m[i]);
        v[i] = v[i] * beta2 + w2 * g[i] * g[i];
        const float denom = sqrtf(v[i]) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (m[i] / denom);
    }
}

/* North-star extension (no counterpart in the reference; PARITY UNPINNED — closed form only):
 * BPR over triples, loss = mean softplus(<u,i-> - <u,i+>), batch-synchronous SGD on the three gathered rows with
 * L2 term reg*row (upstream LightGCN-PyTorch semantics).  Reads from (U_read, I_read), updates (U_w, I_w). */
void spex_oracle_bpr_sgd_f64(const float *U_read, const float *I_read, double *U_w, double *I_w, const int64_t *u,
                             const int64_t *ip, const int64_t *in, int64_t T, int32_t d, double lr, double reg,
                             double *loss)
{
    double acc = 0.0;
    for (int64_t t = 0; t < T; ++t) {
        const float *pu = U_read + (size_t)u[t] * d, *pp = I_read + (size_t)ip[t] * d, *pn = I_read + (size_t)in[t] * d;
        double x = 0.0;
        for (int32_t k = 0; k < d; ++k) x += (double)pu[k] * ((double)pn[k] - (double)pp[k]);
        acc += (x > 0 ? x : 0.0) + log1p(exp(-fabs(x)));
        const double s = 1.0 / (1.0 + exp(-x)) / (double)T; /* d loss / d x */
        for (int32_t k = 0; k < d; ++k) {
            U_w[(size_t)u[t] * d + k] -= lr * (s * ((double)pn[k] - (double)pp[k]) + reg * (double)pu[k] / (double)T);
            I_w[(size_t)ip[t] * d + k] -= lr * (-s * (double)pu[k] + reg * (double)pp[k] / (double)T);
            I_w[(size_t)in[t] * d + k] -= lr * (s * (double)pu[k] + reg * (double)pn[k] / (double)T);
        }
    }
    if (loss) *loss = acc / (double)T;
}
