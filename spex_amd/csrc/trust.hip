// The trust head of the dual-task model as three launches (SURVEY.md 8f "next" #1 — the caller on the other side of the
// shared user table): LightGCN_SPEX/code/utility1/model_expert_s.py:170-192 (`forward`, flag 0/2) with `compute_scores`
// (:128-148) and the two GraphAttentionLayer stages (utility2/layers.py:15-71), followed by nn.CrossEntropyLoss (:192).
//
// Per path x_0 .. x_{l-1} (user ids; padded to L with the table's pad row), hidden size 64 == one wavefront, lane == column:
//   e_i = E[x_i]
//   in_att (H heads, positional, layers.py:22-31), i < l-1:  A = e_i + (l-i), Bv = e_{i+1} + (l-i-1),
//            (w0, w1) = softmax([A.a1 + A.a2, A.a1 + Bv.a2]),  m_ih = w0 A + w1 Bv;   m_{l-1,h} = e_{l-1}
//   r_i = [m_i0 | .. | m_i,H-1] @ w,   o_i = ELU(r_i)                                        (model_expert_s.py:181-183)
//   out_att (one head, dense, layers.py:58-63), i < l-1:  h_i = v0 o_i + v1 o_{i+1} (same softmax on o);  h_{l-1} = o_{l-1}
//   ht = h_{l-1};  q1 = W1 ht + b1;  s_i = sigmoid(q1 + W2 h_i + b2);  alpha_i = w3 . s_i;  a = sum_{i<l} alpha_i h_i   (:130-135)
//   p_a = Wt [a | ht] + bt  (hybrid; else p_a = a);  pm = max_i (e_i * mask_i)  (a padded position contributes 0)   (:136-142)
//   (g0, g1) = softmax([p_a | pm] @ att_t);  a2 = g0 p_a + g1 pm;  scores = a2 . E[:-1]^T;  loss = mean_b CE(scores_b, target_b)
// Positions >= l only ever reach masked terms, so they are not evaluated.  In torch this is ~60 small launches forward
// and ~120 backward per step for <= 15 paths (1.8 ms of a 2.7 ms dual-task step, almost all of it launch latency); here:
//   trust_head_fwd   one wave per path: everything up to a2 (matrices read straight from L2, vectors staged in LDS)
//   trust_ce         one workgroup per path: logits against the whole user table, log-sum-exp, loss, d scores (kept in a
//                    [B, n_users] buffer) and d a2;  then one wave per user row: d E[u] += sum_b d scores[b,u] a2[b]
//   trust_head_bwd   one wave per path: the chain above backwards; parameter gradients and table rows added with atomics
// All parameters live in ONE flat block (layout below) so that a step's gradients are one buffer and one Adam launch.
#include <math.h>

#include "spex_common.h"

using namespace spex;

namespace {

constexpr int kD = 64;        // hidden size: one lane per column
constexpr int kMaxL = 16;     // longest padded path
constexpr int kMaxH = 4;      // input attention heads
constexpr int kCeThreads = 1024;

struct Layout {
    int in_att, out_att, w, W1, b1, W2, b2, w3, Wt, bt, att_t, total;
};

__host__ __device__ inline Layout layout(int H)
{
    Layout o;
    int p = 0;
    o.in_att = p;  p += H * 2 * kD;      // attention_h.a, [H][2d]
    o.out_att = p; p += 2 * kD;          // out_att.a
    o.w = p;       p += H * kD * kD;     // w [H d, d]
    o.W1 = p;      p += kD * kD;         // linear_one.weight [d, d] (out, in)
    o.b1 = p;      p += kD;
    o.W2 = p;      p += kD * kD;         // linear_two.weight
    o.b2 = p;      p += kD;
    o.w3 = p;      p += kD;              // linear_three.weight [1, d]
    o.Wt = p;      p += 2 * kD * kD;     // linear_transform.weight [d, 2d]
    o.bt = p;      p += kD;
    o.att_t = p;   p += 4 * kD;          // att_t [2d, 2]
    o.total = p;
    return o;
}

// per-path workspace written by the forward for the backward (floats)
struct WsLayout {
    int w0, v0, alpha, r, h, s, a, pa, pm, arg, g0, stride;
};

__host__ __device__ inline WsLayout ws_layout(int L, int H)
{
    WsLayout o;
    int p = 0;
    o.w0 = p;    p += L * H;
    o.v0 = p;    p += L;
    o.alpha = p; p += L;
    o.g0 = p;    p += 1;
    p = (p + 63) / 64 * 64;
    o.r = p;     p += L * kD;
    o.h = p;     p += L * kD;
    o.s = p;     p += L * kD;
    o.a = p;     p += kD;
    o.pa = p;    p += kD;
    o.pm = p;    p += kD;
    o.arg = p;   p += kD;
    o.stride = p;
    return o;
}

struct TrustArgs {
    const float *table;      // [n_rows, 64] user table incl. the pad row
    int64_t n_rows;
    const float *P;          // flat parameter block
    const int64_t *seq;      // [B, L]
    const int64_t *seq_l;    // [B]
    int B, L, H, hybrid;
};

__device__ __forceinline__ void load_row(const float *row, float4 (&wr)[16])
{
#pragma unroll
    for (int j = 0; j < 16; ++j) wr[j] = reinterpret_cast<const float4 *>(row)[j];
}

__device__ __forceinline__ float dot_row(const float4 (&wr)[16], const float *x)   // x: 64 floats in LDS (broadcast reads)
{
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 xv = reinterpret_cast<const float4 *>(x)[j];
        acc = fmaf(wr[j].x, xv.x, acc);
        acc = fmaf(wr[j].y, xv.y, acc);
        acc = fmaf(wr[j].z, xv.z, acc);
        acc = fmaf(wr[j].w, xv.w, acc);
    }
    return acc;
}

__device__ __forceinline__ void softmax2(float x0, float x1, float &p0, float &p1)
{
    const float m = fmaxf(x0, x1), e0 = expf(x0 - m), e1 = expf(x1 - m);
    p0 = e0 / (e0 + e1);
    p1 = e1 / (e0 + e1);
}

__device__ __forceinline__ int path_len(const TrustArgs &p, int b)
{
    const int l = (int)p.seq_l[b];
    return l < 1 ? 1 : (l > p.L ? p.L : l);
}

__device__ __forceinline__ float table_at(const TrustArgs &p, int64_t x, int lane)
{
    return (x >= 0 && x < p.n_rows) ? p.table[(size_t)x * kD + lane] : 0.0f;   // never gather out of bounds
}

// ---------------------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(kWave) void trust_head_fwd_kernel(const TrustArgs p, float *__restrict__ a2_out,
                                                               float *__restrict__ ws)
{
    extern __shared__ float4 s_raw[];
    float *s = reinterpret_cast<float *>(s_raw);
    const int lane = threadIdx.x, b = blockIdx.x, L = p.L, H = p.H;
    const Layout lo = layout(H);
    const WsLayout wl = ws_layout(L, H);
    float *e = s, *M = e + L * kD, *o = M + L * H * kD, *h = o + L * kD, *vec = h + L * kD;
    float *W = ws ? ws + (size_t)b * wl.stride : nullptr;
    const int l = path_len(p, b);
    const float *P = p.P;

    for (int i = 0; i < l; ++i) e[i * kD + lane] = table_at(p, p.seq[(size_t)b * L + i], lane);
    // input attention heads
    for (int i = 0; i < l; ++i) {
        const float ei = e[i * kD + lane];
        if (i < l - 1) {
            const float A = ei + (float)(l - i), Bv = e[(i + 1) * kD + lane] + (float)(l - i - 1);
            for (int hh = 0; hh < H; ++hh) {
                const float a1 = P[lo.in_att + hh * 2 * kD + lane], a2 = P[lo.in_att + hh * 2 * kD + kD + lane];
                const float s1 = wave_sum_f32(A * a1), s2 = wave_sum_f32(A * a2), s3 = wave_sum_f32(Bv * a2);
                float w0, w1;
                softmax2(s1 + s2, s1 + s3, w0, w1);
                M[(i * H + hh) * kD + lane] = w0 * A + w1 * Bv;
                if (W && lane == 0) W[wl.w0 + i * H + hh] = w0;
            }
        } else {
            for (int hh = 0; hh < H; ++hh) M[(i * H + hh) * kD + lane] = ei;
        }
    }
    __syncthreads();
    // r_i = M_i @ w (w row k is coalesced across lanes; M_i[k] is an LDS broadcast), o_i = ELU(r_i)
    {
        float acc[kMaxL];
#pragma unroll
        for (int i = 0; i < kMaxL; ++i) acc[i] = 0.0f;
        for (int k = 0; k < H * kD; k += 4) {
            const float w0 = P[lo.w + (k + 0) * kD + lane], w1 = P[lo.w + (k + 1) * kD + lane];
            const float w2 = P[lo.w + (k + 2) * kD + lane], w3 = P[lo.w + (k + 3) * kD + lane];
#pragma unroll
            for (int i = 0; i < kMaxL; ++i)
                if (i < l) {
                    const float4 m = *reinterpret_cast<const float4 *>(&M[i * H * kD + k]);
                    acc[i] = fmaf(m.x, w0, acc[i]);
                    acc[i] = fmaf(m.y, w1, acc[i]);
                    acc[i] = fmaf(m.z, w2, acc[i]);
                    acc[i] = fmaf(m.w, w3, acc[i]);
                }
        }
#pragma unroll
        for (int i = 0; i < kMaxL; ++i)
            if (i < l) {
                const float r = acc[i];
                if (W) W[wl.r + i * kD + lane] = r;
                o[i * kD + lane] = r > 0.0f ? r : expm1f(r);
            }
    }
    // output attention layer
    {
        const float c1 = P[lo.out_att + lane], c2 = P[lo.out_att + kD + lane];
        for (int i = 0; i < l; ++i) {
            const float oi = o[i * kD + lane];
            float hv = oi;
            if (i < l - 1) {
                const float on = o[(i + 1) * kD + lane];
                const float s1 = wave_sum_f32(oi * c1), s2 = wave_sum_f32(oi * c2), s3 = wave_sum_f32(on * c2);
                float v0, v1;
                softmax2(s1 + s2, s1 + s3, v0, v1);
                hv = v0 * oi + v1 * on;
                if (W && lane == 0) W[wl.v0 + i] = v0;
            }
            h[i * kD + lane] = hv;
            if (W) W[wl.h + i * kD + lane] = hv;
        }
    }
    __syncthreads();
    // soft-attention readout
    float4 wr[16];
    const float *ht = h + (l - 1) * kD;
    load_row(P + lo.W1 + lane * kD, wr);
    const float q1 = P[lo.b1 + lane] + dot_row(wr, ht);
    load_row(P + lo.W2 + lane * kD, wr);
    const float b2 = P[lo.b2 + lane], w3 = P[lo.w3 + lane];
    float av = 0.0f;
    for (int i = 0; i < l; ++i) {
        const float q2 = b2 + dot_row(wr, h + i * kD);
        const float sg = 1.0f / (1.0f + expf(-(q1 + q2)));
        const float alpha = wave_sum_f32(w3 * sg);
        if (W) {
            W[wl.s + i * kD + lane] = sg;
            if (lane == 0) W[wl.alpha + i] = alpha;
        }
        av = fmaf(alpha, h[i * kD + lane], av);
    }
    vec[lane] = av;
    __syncthreads();
    float pa = av;
    if (p.hybrid) {
        load_row(P + lo.Wt + lane * 2 * kD, wr);
        pa = P[lo.bt + lane] + dot_row(wr, vec);
        load_row(P + lo.Wt + lane * 2 * kD + kD, wr);
        pa += dot_row(wr, ht);
    }
    // max-pool over the path's own rows (a padded position contributes 0)
    float pm = -INFINITY;
    int arg = -1;
    for (int i = 0; i < l; ++i) {
        const float v = e[i * kD + lane];
        if (v > pm) { pm = v; arg = i; }
    }
    if (l < L && !(pm > 0.0f)) { pm = 0.0f; arg = -1; }
    const float t0 = wave_sum_f32(pa * P[lo.att_t + lane * 2] + pm * P[lo.att_t + (kD + lane) * 2]);
    const float t1 = wave_sum_f32(pa * P[lo.att_t + lane * 2 + 1] + pm * P[lo.att_t + (kD + lane) * 2 + 1]);
    float g0, g1;
    softmax2(t0, t1, g0, g1);
    a2_out[(size_t)b * kD + lane] = pa * g0 + pm * g1;
    if (W) {
        W[wl.a + lane] = av;
        W[wl.pa + lane] = pa;
        W[wl.pm + lane] = pm;
        W[wl.arg + lane] = __int_as_float(arg);
        if (lane == 0) W[wl.g0] = g0;
    }
}

// ---------------------------------------------------------------------------------------------------- logits + CE
// One workgroup per path.  thread t owns users t, t + 1024, ..: score = a2 . E[u] (its own 256-byte row, 16 float4);
// block max / sum-exp; loss_b = lse - score[target];  d score = (softmax - onehot) * scale / B, left in dscore[b, :];
// then wave w sums d score[b,u] E[u] over users w, w + 16, .. (lane == column) -> d a2[b].
__device__ __forceinline__ float block_reduce(float v, float *red, bool is_max)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float r = red[0];
    for (int k = 1; k < nw; ++k) r = is_max ? fmaxf(r, red[k]) : r + red[k];
    return r;
}

__global__ __launch_bounds__(kCeThreads) void trust_ce_kernel(const float *__restrict__ table, int n_users,
                                                             const float *__restrict__ a2, const int64_t *__restrict__ targets,
                                                             int B, float scale, const float *__restrict__ scale_dev,
                                                             float *__restrict__ dscore, float *__restrict__ loss_b,
                                                             float *__restrict__ grad_a2)
{
    __shared__ float4 s_a2[16];
    __shared__ float red[kCeThreads / 64];
    __shared__ float s_acc[kCeThreads / 64][kD];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < 16) s_a2[t] = reinterpret_cast<const float4 *>(a2 + (size_t)b * kD)[t];
    __syncthreads();
    float *ds = dscore + (size_t)b * n_users;
    float mx = -INFINITY;
    for (int u = t; u < n_users; u += kCeThreads) {
        const float4 *row = reinterpret_cast<const float4 *>(table + (size_t)u * kD);
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 r = row[j], x = s_a2[j];
            acc = fmaf(r.x, x.x, acc);
            acc = fmaf(r.y, x.y, acc);
            acc = fmaf(r.z, x.z, acc);
            acc = fmaf(r.w, x.w, acc);
        }
        ds[u] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = block_reduce(mx, red, true);
    float se = 0.0f;
    for (int u = t; u < n_users; u += kCeThreads) se += expf(ds[u] - mx);
    se = block_reduce(se, red, false);
    const float lse = mx + logf(se);
    const int64_t tg = targets[b];
    const bool tg_ok = tg >= 0 && tg < n_users;
    if (t == 0) loss_b[b] = tg_ok ? lse - ds[tg] : 0.0f;     // (thread 0 wrote nothing others read: ds[tg] is global memory)
    __syncthreads();
    const float k = scale * (scale_dev ? *scale_dev : 1.0f) / (float)B;
    for (int u = t; u < n_users; u += kCeThreads) ds[u] = tg_ok ? (expf(ds[u] - lse) - (u == tg ? 1.0f : 0.0f)) * k : 0.0f;
    __syncthreads();
    const int lane = t & 63, wv = t >> 6;
    float acc = 0.0f;
    for (int u = wv; u < n_users; u += kCeThreads / 64) acc = fmaf(ds[u], table[(size_t)u * kD + lane], acc);
    s_acc[wv][lane] = acc;
    __syncthreads();
    if (t < kD) {
        float g = 0.0f;
        for (int w = 0; w < kCeThreads / 64; ++w) g += s_acc[w][t];
        grad_a2[(size_t)b * kD + t] = g;
    }
}

// d E[u, :] += sum_b d score[b, u] a2[b, :]  (one wave per user row, no atomics: this launch owns the rows);  block 0 also
// reduces the per-path losses in path order.
__global__ __launch_bounds__(256) void trust_table_grad_kernel(const float *__restrict__ dscore, const float *__restrict__ a2,
                                                              int n_users, int B, const float *__restrict__ loss_b,
                                                              float *__restrict__ grad_table, float *__restrict__ loss_out,
                                                              int loss_accumulate)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (int u = wave; u < n_users; u += n_waves) {
        float acc = 0.0f;
        for (int b = 0; b < B; ++b) acc = fmaf(dscore[(size_t)b * n_users + u], a2[(size_t)b * kD + lane], acc);
        grad_table[(size_t)u * kD + lane] += acc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out) {
        float sum = 0.0f;
        for (int b = 0; b < B; ++b) sum += loss_b[b];
        sum /= (float)B;
        *loss_out = loss_accumulate ? *loss_out + sum : sum;
    }
}

// --------------------------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(kWave) void trust_head_bwd_kernel(const TrustArgs p, const float *__restrict__ ws,
                                                               const float *__restrict__ grad_a2, float *grad_P,
                                                               float *grad_table)
{
    extern __shared__ float4 s_raw[];
    float *s = reinterpret_cast<float *>(s_raw);
    const int lane = threadIdx.x, b = blockIdx.x, L = p.L, H = p.H;
    const Layout lo = layout(H);
    const WsLayout wl = ws_layout(L, H);
    float *e = s, *M = e + L * kD, *dM = M + L * H * kD, *o = dM + L * H * kD, *h = o + L * kD, *dh = h + L * kD;
    float *dO = dh + L * kD, *du = dO + L * kD, *vec = du + L * kD;      // vec: [4][64]
    const float *W = ws + (size_t)b * wl.stride;
    const int l = path_len(p, b);
    const float *P = p.P;
    float *G = grad_P;

    // restore the forward's state: rows, attention mixes (from the saved softmax weights), o = ELU(r), h
    for (int i = 0; i < l; ++i) {
        e[i * kD + lane] = table_at(p, p.seq[(size_t)b * L + i], lane);
        const float r = W[wl.r + i * kD + lane];
        o[i * kD + lane] = r > 0.0f ? r : expm1f(r);
        h[i * kD + lane] = W[wl.h + i * kD + lane];
        dO[i * kD + lane] = 0.0f;
    }
    for (int i = 0; i < l; ++i) {
        const float ei = e[i * kD + lane];
        if (i < l - 1) {
            const float A = ei + (float)(l - i), Bv = e[(i + 1) * kD + lane] + (float)(l - i - 1);
            for (int hh = 0; hh < H; ++hh) {
                const float w0 = W[wl.w0 + i * H + hh];
                M[(i * H + hh) * kD + lane] = w0 * A + (1.0f - w0) * Bv;
            }
        } else {
            for (int hh = 0; hh < H; ++hh) M[(i * H + hh) * kD + lane] = ei;
        }
    }
    const float av = W[wl.a + lane], pa = W[wl.pa + lane], pm = W[wl.pm + lane], g0 = W[wl.g0], g1 = 1.0f - g0;
    const int arg = __float_as_int(W[wl.arg + lane]);
    const float ht = h[(l - 1) * kD + lane];
    const float da2 = grad_a2[(size_t)b * kD + lane];

    // gate between the pooled vector and the max-pool
    const float dg0 = wave_sum_f32(da2 * pa), dg1 = wave_sum_f32(da2 * pm);
    const float dt0 = g0 * g1 * (dg0 - dg1);
    const float dpa = g0 * da2 + (P[lo.att_t + lane * 2] - P[lo.att_t + lane * 2 + 1]) * dt0;
    const float dpm = g1 * da2 + (P[lo.att_t + (kD + lane) * 2] - P[lo.att_t + (kD + lane) * 2 + 1]) * dt0;
    atomicAdd(G + lo.att_t + lane * 2, pa * dt0);
    atomicAdd(G + lo.att_t + lane * 2 + 1, -pa * dt0);
    atomicAdd(G + lo.att_t + (kD + lane) * 2, pm * dt0);
    atomicAdd(G + lo.att_t + (kD + lane) * 2 + 1, -pm * dt0);
    if (arg >= 0) {
        const int64_t x = p.seq[(size_t)b * L + arg];
        if (x >= 0 && x < p.n_rows) atomicAdd(grad_table + (size_t)x * kD + lane, dpm);
    }
    // p_a = Wt [a | ht] + bt
    float da = dpa, dht = 0.0f;
    if (p.hybrid) {
        atomicAdd(G + lo.bt + lane, dpa);
        vec[lane] = dpa;
        __syncthreads();
        da = 0.0f;
        for (int c = 0; c < kD; ++c) {
            const float g = vec[c];
            da = fmaf(P[lo.Wt + c * 2 * kD + lane], g, da);
            dht = fmaf(P[lo.Wt + c * 2 * kD + kD + lane], g, dht);
            atomicAdd(G + lo.Wt + c * 2 * kD + lane, g * av);
            atomicAdd(G + lo.Wt + c * 2 * kD + kD + lane, g * ht);
        }
    }
    // a = sum alpha_i h_i,  alpha_i = w3 . s_i,  s_i = sigmoid(q1 + q2_i)
    {
        const float w3 = P[lo.w3 + lane];
        float dw3 = 0.0f, dq1 = 0.0f;
        for (int i = 0; i < l; ++i) {
            const float hi = h[i * kD + lane], si = W[wl.s + i * kD + lane];
            const float dalpha = wave_sum_f32(da * hi);
            dh[i * kD + lane] = W[wl.alpha + i] * da;
            dw3 = fmaf(dalpha, si, dw3);
            const float d = dalpha * w3 * si * (1.0f - si);
            du[i * kD + lane] = d;
            dq1 += d;
        }
        atomicAdd(G + lo.w3 + lane, dw3);
        atomicAdd(G + lo.b1 + lane, dq1);
        atomicAdd(G + lo.b2 + lane, dq1);
        vec[kD + lane] = dq1;
    }
    __syncthreads();
    // linear_one: d W1[c][k] += dq1[c] ht[k];  d ht[k] += sum_c W1[c][k] dq1[c]          (lane == k: coalesced rows)
    for (int c = 0; c < kD; ++c) {
        const float g = vec[kD + c];
        dht = fmaf(P[lo.W1 + c * kD + lane], g, dht);
        atomicAdd(G + lo.W1 + c * kD + lane, g * ht);
    }
    // linear_two: d W2[c][k] += sum_i du_i[c] h_i[k];  d h_i[k] += sum_c W2[c][k] du_i[c]
    {
        float acc[kMaxL], hreg[kMaxL];
#pragma unroll
        for (int i = 0; i < kMaxL; ++i) {
            acc[i] = 0.0f;
            hreg[i] = i < l ? h[i * kD + lane] : 0.0f;
        }
        for (int c = 0; c < kD; ++c) {
            const float wv = P[lo.W2 + c * kD + lane];
            float g = 0.0f;
#pragma unroll
            for (int i = 0; i < kMaxL; ++i)
                if (i < l) {
                    const float d = du[i * kD + c];
                    g = fmaf(d, hreg[i], g);
                    acc[i] = fmaf(wv, d, acc[i]);
                }
            atomicAdd(G + lo.W2 + c * kD + lane, g);
        }
#pragma unroll
        for (int i = 0; i < kMaxL; ++i)
            if (i < l) dh[i * kD + lane] += acc[i];
    }
    dh[(l - 1) * kD + lane] += dht;
    // output attention layer (the first half of its parameter cancels in the softmax: gradient exactly 0)
    {
        const float c2 = P[lo.out_att + kD + lane];
        float dc2 = 0.0f;
        for (int i = 0; i < l; ++i) {
            const float g = dh[i * kD + lane];
            if (i < l - 1) {
                const float delta = o[i * kD + lane] - o[(i + 1) * kD + lane], v0 = W[wl.v0 + i];
                const float dz = wave_sum_f32(g * delta) * v0 * (1.0f - v0);
                dO[i * kD + lane] += v0 * g + dz * c2;
                dO[(i + 1) * kD + lane] += (1.0f - v0) * g - dz * c2;
                dc2 = fmaf(dz, delta, dc2);
            } else {
                dO[i * kD + lane] += g;
            }
        }
        atomicAdd(G + lo.out_att + kD + lane, dc2);
    }
    // ELU, then r = M w:  d w[k][c] += sum_i M_i[k] dr_i[c]  (lane == c);  d M_i[k] = sum_c w[k][c] dr_i[c]  (lane == k mod 64)
    float *dr = du;     // du is dead
    {
        float drreg[kMaxL];
#pragma unroll
        for (int i = 0; i < kMaxL; ++i) {
            drreg[i] = 0.0f;
            if (i < l) {
                const float r = W[wl.r + i * kD + lane];
                drreg[i] = dO[i * kD + lane] * (r > 0.0f ? 1.0f : expf(r));
                dr[i * kD + lane] = drreg[i];
            }
        }
        __syncthreads();
        for (int k = 0; k < H * kD; ++k) {
            float g = 0.0f;
#pragma unroll
            for (int i = 0; i < kMaxL; ++i)
                if (i < l) g = fmaf(M[i * H * kD + k], drreg[i], g);
            atomicAdd(G + lo.w + k * kD + lane, g);
        }
        float4 wr[16];
        for (int hh = 0; hh < H; ++hh) {
            load_row(P + lo.w + (hh * kD + lane) * kD, wr);
            for (int i = 0; i < l; ++i) dM[(i * H + hh) * kD + lane] = dot_row(wr, dr + i * kD);
        }
    }
    // input attention heads
    {
        float dE[kMaxL], da2h[kMaxH];
#pragma unroll
        for (int i = 0; i < kMaxL; ++i) dE[i] = 0.0f;
#pragma unroll
        for (int hh = 0; hh < kMaxH; ++hh) da2h[hh] = 0.0f;
#pragma unroll
        for (int i = 0; i < kMaxL; ++i) {
            if (i < l - 1) {
                const float delta = e[i * kD + lane] - e[(i + 1) * kD + lane] + 1.0f;
#pragma unroll
                for (int hh = 0; hh < kMaxH; ++hh)
                    if (hh < H) {
                        const float g = dM[(i * H + hh) * kD + lane], w0 = W[wl.w0 + i * H + hh];
                        const float a2 = P[lo.in_att + hh * 2 * kD + kD + lane];
                        const float dz = wave_sum_f32(g * delta) * w0 * (1.0f - w0);
                        dE[i] += w0 * g + dz * a2;
                        if (i + 1 < kMaxL) dE[i + 1] += (1.0f - w0) * g - dz * a2;
                        da2h[hh] = fmaf(dz, delta, da2h[hh]);
                    }
            } else if (i == l - 1) {
#pragma unroll
                for (int hh = 0; hh < kMaxH; ++hh)
                    if (hh < H) dE[i] += dM[(i * H + hh) * kD + lane];
            }
        }
#pragma unroll
        for (int i = 0; i < kMaxL; ++i)
            if (i < l) {
                const int64_t x = p.seq[(size_t)b * L + i];
                if (x >= 0 && x < p.n_rows) atomicAdd(grad_table + (size_t)x * kD + lane, dE[i]);
            }
#pragma unroll
        for (int hh = 0; hh < kMaxH; ++hh)
            if (hh < H) atomicAdd(G + lo.in_att + hh * 2 * kD + kD + lane, da2h[hh]);
    }
}

int check_head(const char *fn, const float *table, int64_t n_rows, const float *params, const int64_t *seq, const int64_t *seq_l,
               int32_t B, int32_t L, int32_t d, int32_t H)
{
    SPEX_CHECK_ARG(table && params && seq && seq_l, "%s: NULL pointer", fn);
    SPEX_CHECK_ARG(B >= 0 && n_rows >= 1, "%s: B=%d n_rows=%lld", fn, B, (long long)n_rows);
    if (d != kD || L < 1 || L > kMaxL || H < 1 || H > kMaxH) {
        spex::set_error("%s: needs hidden size 64, 1..%d path positions, 1..%d heads (got d=%d L=%d heads=%d)", fn, kMaxL, kMaxH, d, L, H);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG((((uintptr_t)table | (uintptr_t)params) & 15) == 0, "%s: table and params must be 16-byte aligned", fn);
    return SPEX_OK;
}

}  // namespace

extern "C" int64_t spex_trust_param_count(int32_t d, int32_t n_heads)
{
    return (d == kD && n_heads >= 1 && n_heads <= kMaxH) ? layout(n_heads).total : -1;
}

extern "C" int64_t spex_trust_workspace_floats(int32_t B, int32_t L, int32_t d, int32_t n_heads)
{
    if (d != kD || L < 1 || L > kMaxL || n_heads < 1 || n_heads > kMaxH || B < 0) return -1;
    return (int64_t)B * ws_layout(L, n_heads).stride;
}

extern "C" int spex_trust_head_fwd_f32(const float *table, int64_t n_rows, const float *params, const int64_t *seq,
                                       const int64_t *seq_l, int32_t B, int32_t L, int32_t d, int32_t n_heads, int32_t hybrid,
                                       float *a2_out, float *ws, void *stream)
{
    if (int rc = check_head("spex_trust_head_fwd_f32", table, n_rows, params, seq, seq_l, B, L, d, n_heads)) return rc;
    SPEX_CHECK_ARG(a2_out, "spex_trust_head_fwd_f32: NULL output");
    if (B == 0) return SPEX_OK;
    const TrustArgs p{table, n_rows, params, seq, seq_l, B, L, n_heads, hybrid};
    const size_t lds = ((size_t)L * kD * 3 + (size_t)L * n_heads * kD + 4 * kD) * sizeof(float);
    hipLaunchKernelGGL(trust_head_fwd_kernel, dim3((unsigned)B), dim3(kWave), lds, (hipStream_t)stream, p, a2_out, ws);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_trust_ce_f32(const float *table, int32_t n_users, const float *a2, const int64_t *targets, int32_t B, int32_t d,
                                 float scale, const float *scale_dev, float *dscore, float *loss_b, float *loss_out,
                                 int32_t loss_accumulate, float *grad_a2, float *grad_table, void *stream)
{
    SPEX_CHECK_ARG(table && a2 && targets && dscore && loss_b && grad_a2 && grad_table, "spex_trust_ce_f32: NULL pointer");
    SPEX_CHECK_ARG(B >= 0 && n_users >= 1, "spex_trust_ce_f32: B=%d n_users=%d", B, n_users);
    if (d != kD) {
        spex::set_error("spex_trust_ce_f32: needs hidden size 64 (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG((((uintptr_t)table | (uintptr_t)a2) & 15) == 0, "spex_trust_ce_f32: table and a2 must be 16-byte aligned");
    if (B == 0) return SPEX_OK;
    hipLaunchKernelGGL(trust_ce_kernel, dim3((unsigned)B), dim3(kCeThreads), 0, (hipStream_t)stream, table, n_users, a2, targets, B,
                       scale, scale_dev, dscore, loss_b, grad_a2);
    SPEX_HIP(hipGetLastError());
    int blocks = (n_users + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(trust_table_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dscore, a2, n_users, B,
                       loss_b, grad_table, loss_out, loss_accumulate);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_trust_head_bwd_f32(const float *table, int64_t n_rows, const float *params, const int64_t *seq,
                                       const int64_t *seq_l, int32_t B, int32_t L, int32_t d, int32_t n_heads, int32_t hybrid,
                                       const float *ws, const float *grad_a2, float *grad_params, float *grad_table, void *stream)
{
    if (int rc = check_head("spex_trust_head_bwd_f32", table, n_rows, params, seq, seq_l, B, L, d, n_heads)) return rc;
    SPEX_CHECK_ARG(ws && grad_a2 && grad_params && grad_table, "spex_trust_head_bwd_f32: NULL pointer");
    if (B == 0) return SPEX_OK;
    const TrustArgs p{table, n_rows, params, seq, seq_l, B, L, n_heads, hybrid};
    const size_t lds = ((size_t)L * kD * 6 + (size_t)L * n_heads * kD * 2 + 4 * kD) * sizeof(float);
    hipLaunchKernelGGL(trust_head_bwd_kernel, dim3((unsigned)B), dim3(kWave), lds, (hipStream_t)stream, p, ws, grad_a2, grad_params,
                       grad_table);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
