// The trust head of the dual-task model as TWO launches per training step (SURVEY.md 8f "next" #1 — the caller on the
// other side of the shared user table): LightGCN_SPEX/code/utility1/model_expert_s.py:170-192 (`forward`, flag 0/2) with
// `compute_scores` (:128-148) and the two GraphAttentionLayer stages (utility2/layers.py:15-71), followed by
// nn.CrossEntropyLoss (:192), forward and backward.
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
// and ~120 backward per step for <= 15 paths (1.8 ms of a 2.7 ms dual-task step, almost all of it launch latency).  Here, in
// training, two launches:
//   trust_path_split_kernel / trust_path_train_kernel   the chains and the sweep of the user table in one launch: one 16-wave workgroup
//                         per path (train), or S workgroups per path sharing the sweep (split), whose LAST arriver folds the shares
//                         in a fixed order and runs the backward chain; operands of every weight gradient and the path's own row
//                         gradients -> workspace
//   trust_reduce_kernel   the path rows that name a user added to the table in (path, position) order (one wave per user row,
//                         no atomics), the logits' share of the table gradient, every weight gradient as a small [rows, 64]^T
//                         [rows, 64] product over the workspaces (one thread per weight), the bias / vector gradients, the mean
//                         loss.  Nothing in the head is atomic: it repeats bit for bit.
// Evaluation (flag 2) keeps the single forward launch (trust_path_kernel<false>).
// All parameters live in ONE flat block (layout below) so that a step's gradients are one buffer and one Adam launch.
#include <math.h>
#include <atomic>

#include <type_traits>

#include "spex_common.h"

using namespace spex;

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));       // packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32)

constexpr int kD = 64;        // hidden size: one lane per column
constexpr int kMaxL = 16;     // longest padded path
constexpr int kMaxH = 4;      // input attention heads
constexpr int kMaxSplit = 8;         // workgroups per path in the split fused form
                                     // (= the default; SPEX_TRUST_SPLIT overrides; 0 / 1 = the one-workgroup form.  16 was tried: the
                                     //  wider fold costs 1.4 us at 3 185 users and gains 3 % at 26 000+)
constexpr int kCUs = 256;            // MI355X compute units: shares x paths stay within one dispatch round
constexpr int kSweepBatchRows = 512; // rows per sweep batch of a 16-wave workgroup: (16 * 64 / 16 lanes per row) * 8 in flight
constexpr int kPathWaves = 16;      // sizes the partial-sum rows of the LDS layout (the chain kernels use the first four)

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

// Per-path workspace: the operands of the weight gradients, written by the path kernel, read by the reduce kernel.
//   vectors (64 floats each): dq1, ht, a, dpa, pa*dt0, pm*dt0, dw3, dc2, da2h[H];  rows (per position i < l): h_i, du_i, dr_i, M_i
struct WsLayout {
    int dq1, ht, a, dpa, vpa, vpm, dw3, dc2, da2h, Hm, DU, DR, M, DE, St, stride;
};

__host__ __device__ inline WsLayout ws_layout(int L, int H)
{
    WsLayout o;
    int p = 0;
    o.dq1 = p; p += kD;  o.ht = p;  p += kD;  o.a = p;   p += kD;  o.dpa = p; p += kD;
    o.vpa = p; p += kD;  o.vpm = p; p += kD;  o.dw3 = p; p += kD;  o.dc2 = p; p += kD;
    o.da2h = p; p += H * kD;
    o.Hm = p;  p += L * kD;
    o.DU = p;  p += L * kD;
    o.DR = p;  p += L * kD;
    o.M = p;   p += L * H * kD;
    o.DE = p;  p += L * kD;              // d loss / d E[x_i] of the path's own rows (added to the table by the reduce kernel)
    o.St = p;  p += (L * (4 + H) + 8 + 6) * kD;   // the forward chain's state: LDS image (e, M, o, h, vec, sigmoids) + 6 register rows
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

struct TrainArgs {
    const int64_t *targets;  // [B]
    int n_users;             // logits are taken against table[0 : n_users]
    float scale;
    const float *scale_dev;
    float *a2;               // [B, 64]
    float *dscore;           // [B, n_users]
    float *loss_b;           // [B]
    float *ws;               // [B, ws stride]
    float *grad_table;
    // the split form (trust_path_split_kernel): S workgroups per path, each sweeping a share of the user table
    int S;                            // workgroups per path
    float *part_da2;                  // [S][B][64]  the shares' sum exp * row, each on its own maximum
    float *part_ms;                   // [S][B][2]   the shares' (max, sum-exp)
    unsigned long long *tickets;      // [B]         (call tag << 32) | arrivals: whatever an earlier call (or nobody) left there counts as 0
    uint32_t tag;                     // this call's tag (unique per process)
};

// LDS of the path kernel (floats).  s (the readout's sigmoids) shares dM's space: s is dead before dM is written.
struct LdsLayout {
    int e, M, dM, o, h, dh, dO, du, vec, sacc, red, tk, total;
};

__host__ __device__ inline LdsLayout lds_layout(int L, int H, bool train)
{
    LdsLayout o;
    int p = 0;
    o.e = p;   p += L * kD;
    o.M = p;   p += L * H * kD;
    o.o = p;   p += L * kD;
    o.h = p;   p += L * kD;
    o.vec = p; p += 8 * kD;              // [0]: a / dpa, [1]: dq1, [2]: a2, [3]: d a2, [4]: w0 (L*H <= 64), [5]: v0 | alpha
    o.dM = p;  p += train ? L * H * kD : 0;
    o.dh = p;  p += train ? L * kD : 0;
    o.dO = p;  p += train ? L * kD : 0;
    o.du = p;  p += train ? L * kD : 0;
    o.sacc = p; p += train ? (kPathWaves + 1) * kD : 0;      // one row per wave + the target's table row
    o.red = p;  p += train ? 2 * kPathWaves : 0;
    o.tk = p;   p += train ? 4 : 0;                          // split form: the path's ticket word as read at kernel start
    o.total = p;
    return o;
}

#ifdef SPEX_STAMPS   // debug build only (tools/trust_stamps.py): phase stamps of workgroup 0 / thread 0, wall_clock64 = 100 MHz
__device__ unsigned long long g_tstamps[32];
#define TSTAMP(k)                                                            \
    do {                                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_tstamps[k] = wall_clock64(); \
    } while (0)
#else
#define TSTAMP(k)
#endif

// Orders one wave's LDS traffic (lane A writes, lane B reads): the LDS pipe serves a wave's instructions in order, so a
// compiler-level fence + scheduling barrier is all that is needed — no s_barrier, the other waves are not involved.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void load_row(const float *__restrict__ row, float4 (&wr)[16])
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

struct FwdState {     // what the forward leaves in registers for the backward chain (wave 0)
    float av, pa, pm, g0, ht;
    int arg;
};

// ------------------------------------------------------------------------------------------------ forward chain (one wave)
// forward, part 1 (wave 0): the path's rows and the input attention heads -> e, M (and the heads' softmax weights)
__device__ __forceinline__ void forward_head(const TrustArgs &p, const LdsLayout &ll, float *s, int b, int l, int lane)
{
    const int L = p.L, H = p.H;
    const Layout lo = layout(H);
    const float *__restrict__ P = p.P;
    float *e = s + ll.e, *M = s + ll.M, *vec = s + ll.vec;

    // path rows: all gathers in flight at once (row index -1 / out of range -> zeros, never an out-of-bounds gather)
    {
        float row[kMaxL];
#pragma unroll
        for (int i = 0; i < kMaxL; ++i) {
            row[i] = 0.0f;
            if (i < l) {
                const int64_t x = p.seq[(size_t)b * L + i];
                if (x >= 0 && x < p.n_rows) row[i] = p.table[(size_t)x * kD + lane];
            }
        }
#pragma unroll
        for (int i = 0; i < kMaxL; ++i)
            if (i < l) e[i * kD + lane] = row[i];
    }
    // input attention heads
    {
        float a1[kMaxH], a2[kMaxH];
#pragma unroll
        for (int hh = 0; hh < kMaxH; ++hh) {
            a1[hh] = hh < H ? P[lo.in_att + hh * 2 * kD + lane] : 0.0f;
            a2[hh] = hh < H ? P[lo.in_att + hh * 2 * kD + kD + lane] : 0.0f;
        }
        for (int i = 0; i < l; ++i) {
            const float ei = e[i * kD + lane];
            if (i < l - 1) {
                const float A = ei + (float)(l - i), Bv = e[(i + 1) * kD + lane] + (float)(l - i - 1);
#pragma unroll
                for (int hh = 0; hh < kMaxH; ++hh)
                    if (hh < H) {
                        const float s1 = wave_sum_f32(A * a1[hh]), s2 = wave_sum_f32(A * a2[hh]), s3 = wave_sum_f32(Bv * a2[hh]);
                        float w0, w1;
                        softmax2(s1 + s2, s1 + s3, w0, w1);
                        M[(i * H + hh) * kD + lane] = w0 * A + w1 * Bv;
                        if (lane == 0) vec[4 * kD + i * H + hh] = w0;
                    }
            } else {
                for (int hh = 0; hh < H; ++hh) M[(i * H + hh) * kD + lane] = ei;
            }
        }
    }
}

// forward, part 2 (kSplit waves, each a quarter of w's rows): acc_i = sum_{k in [k_beg, k_end)} M_i[k] w[k][lane]
// (w row k is coalesced across lanes, 16 independent loads per batch; M_i[k] is an LDS broadcast)
__device__ __forceinline__ void forward_mw(const TrustArgs &p, const LdsLayout &ll, const float *s, int l, int lane, int k_beg, int k_end,
                                           float (&acc)[kMaxL])
{
    const int H = p.H;
    const Layout lo = layout(H);
    const float *__restrict__ P = p.P;
    const float *M = s + ll.M;
    {
#pragma unroll
        for (int i = 0; i < kMaxL; ++i) acc[i] = 0.0f;
        for (int k0 = k_beg; k0 < k_end; k0 += 16) {
            float wk[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) wk[j] = P[lo.w + (k0 + j) * kD + lane];
#pragma unroll
            for (int i = 0; i < kMaxL; ++i)
                if (i < l) {
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const float4 m = *reinterpret_cast<const float4 *>(&M[i * H * kD + k0 + j4 * 4]);
                        acc[i] = fmaf(m.x, wk[j4 * 4 + 0], acc[i]);
                        acc[i] = fmaf(m.y, wk[j4 * 4 + 1], acc[i]);
                        acc[i] = fmaf(m.z, wk[j4 * 4 + 2], acc[i]);
                        acc[i] = fmaf(m.w, wk[j4 * 4 + 3], acc[i]);
                    }
                }
        }
    }
}

// forward, part 3 (wave 0): ELU, the output attention layer, the readout, the gate -> a2
__device__ __forceinline__ FwdState forward_tail(const TrustArgs &p, const LdsLayout &ll, float *s, int b, int l, int lane,
                                                 const float (&acc)[kMaxL], float *__restrict__ a2_out)
{
    const int L = p.L, H = p.H;
    const Layout lo = layout(H);
    const float *__restrict__ P = p.P;
    float *e = s + ll.e, *o = s + ll.o, *h = s + ll.h, *vec = s + ll.vec;
    float *sg_all = s + ll.dM;                 // the sigmoids (training form only)
    const bool train = ll.total > ll.dM;         // the training layout carries the backward's arrays after dM
#pragma unroll
    for (int i = 0; i < kMaxL; ++i)
        if (i < l) o[i * kD + lane] = acc[i] > 0.0f ? acc[i] : expm1f(acc[i]);
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
                if (lane == 0) vec[5 * kD + i] = v0;
            }
            h[i * kD + lane] = hv;
        }
    }
    wave_sync();
    TSTAMP(16);
    // soft-attention readout
    FwdState st;
    float4 wr[16], wr2[16];
    const float *ht = h + (l - 1) * kD;
    load_row(P + lo.W1 + lane * kD, wr);
    load_row(P + lo.W2 + lane * kD, wr2);
    const float b1 = P[lo.b1 + lane], b2 = P[lo.b2 + lane], w3 = P[lo.w3 + lane];
    const float q1 = b1 + dot_row(wr, ht);
    float av = 0.0f;
    for (int i = 0; i < l; ++i) {
        const float q2 = b2 + dot_row(wr2, h + i * kD);
        const float sg = 1.0f / (1.0f + expf(-(q1 + q2)));
        const float alpha = wave_sum_f32(w3 * sg);
        if (train) {
            sg_all[i * kD + lane] = sg;
            if (lane == 0) vec[5 * kD + kMaxL + i] = alpha;
        }
        av = fmaf(alpha, h[i * kD + lane], av);
    }
    vec[lane] = av;
    wave_sync();
    TSTAMP(17);
    float pa = av;
    if (p.hybrid) {
        load_row(P + lo.Wt + lane * 2 * kD, wr);
        load_row(P + lo.Wt + lane * 2 * kD + kD, wr2);
        pa = P[lo.bt + lane] + dot_row(wr, vec);
        pa += dot_row(wr2, ht);
    }
    TSTAMP(18);
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
    const float a2v = pa * g0 + pm * g1;
    a2_out[(size_t)b * kD + lane] = a2v;
    vec[2 * kD + lane] = a2v;
    st.av = av; st.pa = pa; st.pm = pm; st.g0 = g0; st.arg = arg; st.ht = ht[lane];
    return st;
}

// ------------------------------------------------------------------------------------ training forward (the whole workgroup)
// The same chain as forward_head / forward_mw / forward_tail, spread over the workgroup's waves wherever its steps are
// independent — wave 0 alone ran it as ~8 dependent L2 round trips plus ~100 dependent cross-lane sums while 3 (tiled form) or
// 15 (fused form) waves waited at the next barrier:
//   F0  waves 0 .. H-1: the path's rows (every wave gathers them itself: the loads hit the same lines) and ONE input head each
//   F1  waves 0 .. 3:   `M @ w`, a quarter of w's rows each, all of a quarter's loads in flight at once
//   F2  wave 0:         partial sums in wave order, ELU, the output attention layer (a serial recurrence over the positions)
//   F3  waves 1 ..:     the readout's matrix-vector products as independent tasks — W2 h_i per position, W1 ht, Wt[:, d:] ht —
//                       each wave holding its matrix row in registers
//   F4  wave 0:         (Wt[:, :d] requested first) sigmoids, alphas, the pooled vector, p_a, the max-pool gate -> a2
// Every value is formed by the same operations in the same order as in the one-wave chain (bit-identical forward).
// NW: waves in the workgroup (>= 4).  Training LDS layout only (q rows live in `dh`, q1 / the ht half of p_a in `sacc`).
__device__ __forceinline__ void forward_mw_all(const TrustArgs &p, const LdsLayout &ll, const float *s, int l, int lane, int k_beg, int n_k,
                                               float (&acc)[kMaxL])
{
    const int H = p.H;
    const Layout lo = layout(H);
    const float *__restrict__ P = p.P;
    const float *M = s + ll.M;
    float wk[4][16];                                       // n_k <= 64 rows of w, requested together
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c * 16 < n_k) {
#pragma unroll
            for (int j = 0; j < 16; ++j) wk[c][j] = P[lo.w + (k_beg + c * 16 + j) * kD + lane];
        }
#pragma unroll
    for (int i = 0; i < kMaxL; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c * 16 < n_k) {
#pragma unroll
            for (int i = 0; i < kMaxL; ++i)
                if (i < l) {
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const float4 m = *reinterpret_cast<const float4 *>(&M[i * H * kD + k_beg + c * 16 + j4 * 4]);
                        acc[i] = fmaf(m.x, wk[c][j4 * 4 + 0], acc[i]);
                        acc[i] = fmaf(m.y, wk[c][j4 * 4 + 1], acc[i]);
                        acc[i] = fmaf(m.z, wk[c][j4 * 4 + 2], acc[i]);
                        acc[i] = fmaf(m.w, wk[c][j4 * 4 + 3], acc[i]);
                    }
                }
        }
}

template <int NW>
__device__ __forceinline__ FwdState forward_train(const TrustArgs &p, const LdsLayout &ll, float *s, int b, int l, int lane, int wave,
                                                  float *__restrict__ a2_out)
{
    static_assert(NW >= 4, "the training chain splits its work over at least four waves");
    const int L = p.L, H = p.H;
    const Layout lo = layout(H);
    const float *__restrict__ P = p.P;
    float *e = s + ll.e, *M = s + ll.M, *o = s + ll.o, *h = s + ll.h, *vec = s + ll.vec;
    float *sg_all = s + ll.dM, *part = s + ll.dM, *q = s + ll.dh, *sacc = s + ll.sacc;
    FwdState st{};
    TSTAMP(0);
    // ---- F0: rows + one input head per wave
    if (wave < H) {
        const int hh = wave;
        const float a1 = P[lo.in_att + hh * 2 * kD + lane], a2 = P[lo.in_att + hh * 2 * kD + kD + lane];
        {
            float row[kMaxL];
#pragma unroll
            for (int i = 0; i < kMaxL; ++i) {
                row[i] = 0.0f;
                if (i < l) {
                    const int64_t x = p.seq[(size_t)b * L + i];
                    if (x >= 0 && x < p.n_rows) row[i] = p.table[(size_t)x * kD + lane];
                }
            }
#pragma unroll
            for (int i = 0; i < kMaxL; ++i)
                if (i < l) e[i * kD + lane] = row[i];             // (every head's wave stores the same values)
        }
        for (int i = 0; i < l; ++i) {
            const float ei = e[i * kD + lane];
            if (i < l - 1) {
                const float A = ei + (float)(l - i), Bv = e[(i + 1) * kD + lane] + (float)(l - i - 1);
                const float s1 = wave_sum_f32(A * a1), s2 = wave_sum_f32(A * a2), s3 = wave_sum_f32(Bv * a2);
                float w0, w1;
                softmax2(s1 + s2, s1 + s3, w0, w1);
                M[(i * H + hh) * kD + lane] = w0 * A + w1 * Bv;
                if (lane == 0) vec[4 * kD + i * H + hh] = w0;
            } else {
                M[(i * H + hh) * kD + lane] = ei;
            }
        }
    }
    const float c1 = P[lo.out_att + lane], c2 = P[lo.out_att + kD + lane];      // (wave 0 needs them in F2: requested early)
    __syncthreads();
    TSTAMP(1);
    // ---- F1: M @ w on four waves
    float acc[kMaxL];
    if (wave < 4) {
        const int per = H * kD / 4;                      // 16 H rows of w per wave
        forward_mw_all(p, ll, s, l, lane, wave * per, per, acc);
        if (wave > 0)
            for (int i = 0; i < l; ++i) part[((wave - 1) * L + i) * kD + lane] = acc[i];
    }
    __syncthreads();
    TSTAMP(2);
    // ---- F2: wave 0 — partials in wave order, ELU, output attention
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < kMaxL; ++i)
            if (i < l)
                for (int w = 0; w + 1 < 4; ++w) acc[i] += part[(w * L + i) * kD + lane];
#pragma unroll
        for (int i = 0; i < kMaxL; ++i)
            if (i < l) o[i * kD + lane] = acc[i] > 0.0f ? acc[i] : expm1f(acc[i]);
        for (int i = 0; i < l; ++i) {
            const float oi = o[i * kD + lane];
            float hv = oi;
            if (i < l - 1) {
                const float on = o[(i + 1) * kD + lane];
                const float s1 = wave_sum_f32(oi * c1), s2 = wave_sum_f32(oi * c2), s3 = wave_sum_f32(on * c2);
                float v0, v1;
                softmax2(s1 + s2, s1 + s3, v0, v1);
                hv = v0 * oi + v1 * on;
                if (lane == 0) vec[5 * kD + i] = v0;
            }
            h[i * kD + lane] = hv;
        }
    }
    __syncthreads();
    TSTAMP(16);
    // ---- F3: the readout's matrix-vector products as tasks on waves 1 .. NW-1 (task k < l: q2_k = b2 + W2 h_k; k == l: q1 = b1 + W1 ht;
    //      k == l + 1: the ht half of p_a)
    const float *ht = h + (l - 1) * kD;
    if (wave != 0) {                              // (three separate blocks, one row of 64 registers live at a time)
        if (wave - 1 < l) {
            float4 wr[16];
            load_row(P + lo.W2 + lane * kD, wr);
            const float b2 = P[lo.b2 + lane];
            for (int k = wave - 1; k < l; k += NW - 1) q[k * kD + lane] = b2 + dot_row(wr, h + k * kD);
        }
        if (wave == 1 + l % (NW - 1)) {
            float4 wr[16];
            load_row(P + lo.W1 + lane * kD, wr);
            sacc[lane] = P[lo.b1 + lane] + dot_row(wr, ht);
        }
        if (p.hybrid && wave == 1 + (l + 1) % (NW - 1)) {
            float4 wr[16];
            load_row(P + lo.Wt + lane * 2 * kD + kD, wr);
            sacc[kD + lane] = dot_row(wr, ht);
        }
    }
    __syncthreads();
    TSTAMP(17);
    // ---- F4: wave 0 — sigmoids, alphas, pooled vector, p_a, max-pool gate
    if (wave == 0) {
        float4 wt[16];                                   // Wt[:, :d], requested here: the sigmoid loop below covers its round trip
        if (p.hybrid) load_row(P + lo.Wt + lane * 2 * kD, wt);
        const float w3 = P[lo.w3 + lane];
        const float q1 = sacc[lane];
        float av = 0.0f;
        for (int i = 0; i < l; ++i) {
            const float q2 = q[i * kD + lane];
            const float sg = 1.0f / (1.0f + expf(-(q1 + q2)));
            const float alpha = wave_sum_f32(w3 * sg);
            sg_all[i * kD + lane] = sg;
            if (lane == 0) vec[5 * kD + kMaxL + i] = alpha;
            av = fmaf(alpha, h[i * kD + lane], av);
        }
        vec[lane] = av;
        wave_sync();
        float pa = av;
        if (p.hybrid) {
            pa = P[lo.bt + lane] + dot_row(wt, vec);
            pa += sacc[kD + lane];
        }
        TSTAMP(18);
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
        const float a2v = pa * g0 + pm * g1;
        a2_out[(size_t)b * kD + lane] = a2v;
        vec[2 * kD + lane] = a2v;
        st.av = av; st.pa = pa; st.pm = pm; st.g0 = g0; st.arg = arg; st.ht = ht[lane];
    }
    TSTAMP(3);
    return st;
}

// --------------------------------------------------------------------------------------- logits + CE (the whole workgroup)
__device__ __forceinline__ float block_reduce(float v, float *red, bool is_max)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float r = red[0];
    for (int k = 1; k < kPathWaves; ++k) r = is_max ? fmaxf(r, red[k]) : r + red[k];
    return r;
}

// scores against table[0 : n_users] (16 lanes per 256-byte row), log-sum-exp, loss_b, d scores -> tr.dscore[b, :], d a2 -> vec[3].
// ONE sweep over the table: d a2 = k (sum_u softmax_u E[u] - E[target]) is accumulated together with the scores, flash-attention
// style — every 16-lane row group keeps a running (max, sum-exp, sum exp * row) and rescales it when its max moves; the 64 groups
// are combined in a fixed order at the end.  (The first version swept the table twice — scores, then sum_u d score[u] E[u] —
// 2 x 815 KB per path through one CU's L1: 28 of the kernel's 70 us.)
//
// SPLIT (trust_path_split_kernel): workgroup `sp` of the path's tr.S sweeps only its share of the batches and publishes
// (max, sum-exp, sum exp * row) — and its raw scores — with agent-scope stores; the workgroups of a path take a ticket, and the
// LAST one to arrive folds the shares in share order (a fixed order, whoever is last), writes loss / d scores / d a2 and goes on
// to the backward chain (returns true); the others are done (false).  No workgroup ever waits for another.
template <bool SPLIT>
__device__ __forceinline__ bool logits_ce(const TrustArgs &p, const TrainArgs &tr, const LdsLayout &ll, float *s, int b, int sp)
{
    const int t = threadIdx.x, n_users = tr.n_users;
    float *vec = s + ll.vec, *red = s + ll.red, *sacc = s + ll.sacc;
    float *__restrict__ ds = tr.dscore + (size_t)b * n_users;
    const float *__restrict__ table = p.table;
    const int sub = t & 15, grp = t >> 4;                        // 64 row groups of 16 lanes
    const float4 x = reinterpret_cast<const float4 *>(vec + 2 * kD)[sub];
    const int64_t tg = tr.targets[b];
    const bool tg_ok = tg >= 0 && tg < n_users;
    float4 e_tg = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (tg_ok) e_tg = reinterpret_cast<const float4 *>(table + (size_t)tg * kD)[sub];
    // The sweep is VALU-bound, not memory-bound (16 lanes per row: every instruction serves four rows; 815 KB through one CU is ~6 us of
    // transfer, the first version's ~56 instructions per row-quad were 17 us): full batches run without a single predicate (no
    // bounds checks, one score store per batch — lane `sub == j` of a row group keeps row j's score), the exponentials are bare
    // v_exp_f32 on a base-2 argument, dot and accumulation are packed FMAs, and the next batch's loads are issued before the
    // current one is consumed.  Only the last, partial batch pays for clamped addresses and masks.
    constexpr int kRows = (kPathWaves * kWave) / 16, kFly = 8;           // row groups per batch row; rows per batch and lane
    constexpr float kLog2e = 1.44269504088896340736f;
    float m_g = -INFINITY, s_g = 0.0f;
    v2f acc_lo = {0.0f, 0.0f}, acc_hi = {0.0f, 0.0f};
    const v2f x_lo = {x.x, x.y}, x_hi = {x.z, x.w};
    const char *__restrict__ tb = reinterpret_cast<const char *>(table);
    const uint32_t off0 = (uint32_t)grp * (kD * 4) + (uint32_t)sub * 16;
    auto load_full = [&](int u0, float4 (&dst)[kFly]) {
#pragma unroll
        for (int j = 0; j < kFly; ++j)
            dst[j] = *reinterpret_cast<const float4 *>(tb + (off0 + (uint32_t)(u0 + j * kRows) * (kD * 4)));
    };
    auto consume = [&](int u0, const float4 (&r)[kFly], auto tail) {
        constexpr bool TAIL = decltype(tail)::value;
        float sc[kFly], keep = 0.0f, bm = -INFINITY;
#pragma unroll
        for (int j = 0; j < kFly; ++j) {
            const v2f lo = {r[j].x, r[j].y}, hi = {r[j].z, r[j].w};
            const v2f pr = __builtin_elementwise_fma(hi, x_hi, lo * x_lo);
            float v = row16_sum_f32(pr.x + pr.y);
            if (TAIL && u0 + j * kRows + grp >= n_users) v = -INFINITY;
            sc[j] = v;
            keep = sub == j ? v : keep;
            bm = fmaxf(bm, v);
        }
        if (sub < kFly) {
            const int u = u0 + sub * kRows + grp;
            if (!TAIL || u < n_users) {
                if constexpr (SPLIT) __hip_atomic_store(ds + u, keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else ds[u] = keep;
            }
        }
        if (!TAIL || bm > -INFINITY) {                            // (uniform over the 16 lanes of a group)
            const float m_new = fmaxf(m_g, bm), f = __builtin_amdgcn_exp2f((m_g - m_new) * kLog2e);   // first batch: 2^-inf = 0
            s_g *= f;
            acc_lo *= f;
            acc_hi *= f;
            const float mb = m_new * kLog2e;
#pragma unroll
            for (int j = 0; j < kFly; ++j) {
                const float w = __builtin_amdgcn_exp2f(fmaf(sc[j], kLog2e, -mb));      // rows beyond n_users: 2^-inf = 0
                s_g += w;
                const v2f lo = {r[j].x, r[j].y}, hi = {r[j].z, r[j].w}, w2 = {w, w};
                acc_lo = __builtin_elementwise_fma(w2, lo, acc_lo);
                acc_hi = __builtin_elementwise_fma(w2, hi, acc_hi);
            }
            m_g = m_new;
        }
    };
    const int n_full = n_users / (kRows * kFly);
    int k_beg = 0, k_end = n_full;
    bool with_tail = n_full * kRows * kFly < n_users;
    if constexpr (SPLIT) {                                        // this share's batches (the partial one is the last batch)
        const int nb = n_full + (with_tail ? 1 : 0);
        k_beg = nb * sp / tr.S;
        k_end = nb * (sp + 1) / tr.S;
        with_tail = with_tail && k_end == nb;
        k_end = k_end < n_full ? k_end : n_full;
    }
    {
        float4 r[kFly], rn[kFly];
        if (k_beg < k_end) load_full(k_beg * kRows * kFly, r);
        for (int k = k_beg; k < k_end; ++k) {
            if (k + 1 < k_end) load_full((k + 1) * kRows * kFly, rn);
            consume(k * kRows * kFly, r, std::false_type{});
#pragma unroll
            for (int j = 0; j < kFly; ++j) r[j] = rn[j];      // (a manual ping-pong of the two buffers spills and is slower: 13.3 vs 11.3 us)
        }
        const int u0 = n_full * kRows * kFly;
        if (with_tail) {
#pragma unroll
            for (int j = 0; j < kFly; ++j) {
                const int u = u0 + j * kRows + grp, uc = u < n_users ? u : n_users - 1;
                r[j] = reinterpret_cast<const float4 *>(table + (size_t)uc * kD)[sub];
            }
            consume(u0, r, std::true_type{});
        }
    }
    float4 acc4 = make_float4(acc_lo.x, acc_lo.y, acc_hi.x, acc_hi.y);
    TSTAMP(4);
    const float mx = block_reduce(m_g, red, true);               // (its barriers also publish ds within the workgroup)
    const float f_g = m_g > -INFINITY ? expf(m_g - mx) : 0.0f;    // a group without rows contributes nothing
    const float se = block_reduce(sub == 0 ? s_g * f_g : 0.0f, red + kPathWaves, false);
    if constexpr (SPLIT) {
        const int wv = t >> 6, B = p.B;
        acc4.x *= f_g; acc4.y *= f_g; acc4.z *= f_g; acc4.w *= f_g;
        acc4.x += __shfl_xor(acc4.x, 16); acc4.y += __shfl_xor(acc4.y, 16); acc4.z += __shfl_xor(acc4.z, 16); acc4.w += __shfl_xor(acc4.w, 16);
        acc4.x += __shfl_xor(acc4.x, 32); acc4.y += __shfl_xor(acc4.y, 32); acc4.z += __shfl_xor(acc4.z, 32); acc4.w += __shfl_xor(acc4.w, 32);
        if ((t & 63) < 16) reinterpret_cast<float4 *>(sacc + wv * kD)[sub] = acc4;
        __syncthreads();
        // ---- publish this share: agent-scope stores go through to memory (the path's workgroups may sit on different XCDs, whose
        //      L2s do not see each other's lines), and every thread has its own stores acknowledged (explicit vmcnt(0) below) before
        //      the barrier in front of the ticket.  No agent-scope FENCE anywhere: on gfx950 that is a writeback + invalidate of the
        //      XCD's whole L2.
        if (t < kD) {
            float g = 0.0f;
            for (int w = 0; w < kPathWaves; ++w) g += sacc[w * kD + t];
            __hip_atomic_store(tr.part_da2 + ((size_t)sp * B + b) * kD + t, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t == 0) {
            __hip_atomic_store(tr.part_ms + ((size_t)sp * B + b) * 2, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tr.part_ms + ((size_t)sp * B + b) * 2 + 1, se, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // EVERY thread waits here for the acknowledgement of ALL its outstanding stores — the raw scores of the sweep above and the
        // share just written — with an explicit s_waitcnt vmcnt(0).  (A workgroup-scope release fence emits NO vmcnt wait on gfx950
        // outside tgsplit mode; an agent-scope release would add a buffer_wbl2 of the XCD's L2.  The stores are sc1 write-through, so
        // "acknowledged" = visible to every XCD.)  The barrier then orders all threads' acknowledgements before thread 0's ticket.
        // tests/test_isa_folds.py asserts the wait in the compiled ISA.
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0); expcnt / lgkmcnt unconstrained
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (compiler ordering + LDS; not what acknowledges the stores)
        __syncthreads();
        if (t == 0) {
            // the ticket word: (this call's tag << 32) | arrivals.  Whatever else is there — an earlier call's word, or nothing the
            // library ever wrote — counts as "nobody yet": the first share installs the tag by compare-exchange, the others add 1.
            // (`cur` was read at kernel start — trust_path_split_kernel — to keep that round trip off this path: a stale value only
            //  makes the exchange fail, and then the tag is in place)
            unsigned long long cur = *reinterpret_cast<const unsigned long long *>(s + ll.tk);
            const unsigned long long first = ((unsigned long long)tr.tag << 32) | 1ull;
            unsigned arrived = 1;
            if ((unsigned)(cur >> 32) == tr.tag
                || !__hip_atomic_compare_exchange_strong(tr.tickets + b, &cur, first, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                // (a failed exchange: the word changed under us — only shares of THIS call write it, so the tag is in place now)
                arrived = (unsigned)__hip_atomic_fetch_add(tr.tickets + b, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            red[0] = arrived == (unsigned)tr.S ? 1.0f : 0.0f;
            // (the last share puts the word back to 0: a launch REPLAYED from a captured HIP graph carries the same tag again)
            if (arrived == (unsigned)tr.S) __hip_atomic_store(tr.tickets + b, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (red[0] == 0.0f) return false;
        // ---- the last share to arrive: all S shares are in memory.  Everything the fold needs is requested at once (one round trip
        //      through memory, not one per share), then folded in share order.
        float mq[kMaxSplit], sq[kMaxSplit], gq[kMaxSplit];
#pragma unroll
        for (int q = 0; q < kMaxSplit; ++q) {
            const bool on = q < tr.S;
            const size_t cell = (size_t)(on ? q : 0) * B + b;
            mq[q] = __hip_atomic_load(tr.part_ms + cell * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sq[q] = __hip_atomic_load(tr.part_ms + cell * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gq[q] = __hip_atomic_load(tr.part_da2 + cell * kD + (t & (kD - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!on) mq[q] = -INFINITY;
        }
        constexpr int kThreads = kPathWaves * kWave;
        float raw[4];                                                  // the first round of raw scores rides along
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = t + j * kThreads < n_users ? __hip_atomic_load(ds + t + j * kThreads, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
        const float e_tg_t = tg_ok ? table[(size_t)tg * kD + (t & (kD - 1))] : 0.0f;
        float big = -INFINITY;
#pragma unroll
        for (int q = 0; q < kMaxSplit; ++q) big = fmaxf(big, mq[q]);
        float se_all = 0.0f, g = 0.0f;
#pragma unroll
        for (int q = 0; q < kMaxSplit; ++q) {
            const float f = mq[q] > -INFINITY ? expf(mq[q] - big) : 0.0f;      // (a share without rows, or beyond S, contributes nothing)
            se_all += f > 0.0f ? sq[q] * f : 0.0f;
            g += f > 0.0f ? gq[q] * f : 0.0f;
        }
        const float lse = big + logf(se_all);
        const float k = tr.scale * (tr.scale_dev ? *tr.scale_dev : 1.0f) / (float)p.B;
        if (t < kD) vec[3 * kD + t] = tg_ok ? k * (g / se_all - e_tg_t) : 0.0f;
        if (t == 0 && !tg_ok) tr.loss_b[b] = 0.0f;                     // (else: written by the thread that holds the target's raw score)
        for (int u0 = t;;) {                                           // d scores, four agent-scope loads in flight per thread
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = u0 + j * kThreads;
                if (u < n_users) ds[u] = tg_ok ? (expf(raw[j] - lse) - (u == tg ? 1.0f : 0.0f)) * k : 0.0f;
                if (tg_ok && u == tg) tr.loss_b[b] = lse - raw[j];
            }
            u0 += 4 * kThreads;
            if (u0 >= n_users) break;                                  // (uniform enough: whole waves leave together except the last)
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[j] = u0 + j * kThreads < n_users ? __hip_atomic_load(ds + u0 + j * kThreads, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
        }
        return true;
    }
    const float lse = mx + logf(se);
    TSTAMP(5);
    if (t == 0) tr.loss_b[b] = tg_ok ? lse - ds[tg] : 0.0f;
    __syncthreads();
    TSTAMP(6);
    const float k = tr.scale * (tr.scale_dev ? *tr.scale_dev : 1.0f) / (float)p.B;
    for (int u = t; u < n_users; u += (kPathWaves * kWave)) ds[u] = tg_ok ? (expf(ds[u] - lse) - (u == tg ? 1.0f : 0.0f)) * k : 0.0f;
    TSTAMP(7);
    // d a2: the groups' accumulators on the common maximum, the four row groups of a wave folded with two cross-lane adds, the
    // waves in LDS in wave order
    const int wv = t >> 6;
    acc4.x *= f_g; acc4.y *= f_g; acc4.z *= f_g; acc4.w *= f_g;
    acc4.x += __shfl_xor(acc4.x, 16); acc4.y += __shfl_xor(acc4.y, 16); acc4.z += __shfl_xor(acc4.z, 16); acc4.w += __shfl_xor(acc4.w, 16);
    acc4.x += __shfl_xor(acc4.x, 32); acc4.y += __shfl_xor(acc4.y, 32); acc4.z += __shfl_xor(acc4.z, 32); acc4.w += __shfl_xor(acc4.w, 32);
    if ((t & 63) < 16) reinterpret_cast<float4 *>(sacc + wv * kD)[sub] = acc4;
    if (t < 16) reinterpret_cast<float4 *>(sacc + kPathWaves * kD)[sub] = e_tg;      // (the row behind sacc: see lds_layout)
    __syncthreads();
    if (t < kD) {
        float g = 0.0f;
        for (int w = 0; w < kPathWaves; ++w) g += sacc[w * kD + t];
        vec[3 * kD + t] = tg_ok ? k * (g / se - sacc[kPathWaves * kD + t]) : 0.0f;
    }
    return true;
}

// ------------------------------------------------------------------------------------ training backward (the whole workgroup)
// The chain backwards, its independent pieces on separate waves (the one-wave version spent 10 of its 23 us on three dependent
// `load a row of w, l dot products` rounds and 5 on four dependent batches of W1 / W2 rows):
//   P0  wave 0:          the gate between pooled vector and max-pool -> d p_a, d pm
//   P1  waves 1, 2:      Wt^T d p_a — the `a` half and the `ht` half, each wave with all 64 rows in flight
//   P2  wave 0:          the readout backwards (alphas, sigmoids) -> du_i, d h_i (first term), dq1
//   P3  waves 1 ..:      W2^T du_i per position (each wave holds W2 in registers);  wave 0: W1^T dq1
//   P4  wave 0:          the output attention layer and ELU backwards -> dr_i;   the other waves copy M into the workspace
//   P5  waves 0 .. H-1:  ONE head each — d M_ih = w_h dr_i (the wave's rows of w in registers), then the input head's own backward
//                        -> that head's share of d e_i (left in the head's dM rows) and its d a2
//   P6  wave 0:          d e_i = the heads' shares in head order (+ the max-pool's d pm) -> workspace
// Weight-gradient operands go to the per-path workspace as before; nothing is atomic, every sum has a fixed order.
template <int NW>
__device__ __forceinline__ void backward_train(const TrustArgs &p, const TrainArgs &tr, const LdsLayout &ll, float *s, int b, int l,
                                               int lane, int wave, const FwdState &st)
{
    static_assert(NW >= 4, "the training chain splits its work over at least four waves");
    const int L = p.L, H = p.H;
    const Layout lo = layout(H);
    const WsLayout wl = ws_layout(L, H);
    const float *__restrict__ P = p.P;
    float *__restrict__ W = tr.ws + (size_t)b * wl.stride;
    float *e = s + ll.e, *M = s + ll.M, *dM = s + ll.dM, *o = s + ll.o, *h = s + ll.h, *dh = s + ll.dh, *dO = s + ll.dO;
    float *du = s + ll.du, *vec = s + ll.vec, *sacc = s + ll.sacc;
    const float *sg_all = s + ll.dM;
    float dpa = 0.0f, dpm = 0.0f;
    TSTAMP(8);
    // ---- P0
    if (wave == 0) {
        const float av = st.av, pa = st.pa, pm = st.pm, g0 = st.g0, g1 = 1.0f - st.g0, ht = st.ht;
        const float da2 = vec[3 * kD + lane];
        const float dg0 = wave_sum_f32(da2 * pa), dg1 = wave_sum_f32(da2 * pm);
        const float dt0 = g0 * g1 * (dg0 - dg1);
        dpa = g0 * da2 + (P[lo.att_t + lane * 2] - P[lo.att_t + lane * 2 + 1]) * dt0;
        dpm = g1 * da2 + (P[lo.att_t + (kD + lane) * 2] - P[lo.att_t + (kD + lane) * 2 + 1]) * dt0;
        W[wl.vpa + lane] = pa * dt0;
        W[wl.vpm + lane] = pm * dt0;
        W[wl.dpa + lane] = p.hybrid ? dpa : 0.0f;
        W[wl.a + lane] = av;
        W[wl.ht + lane] = ht;
        vec[lane] = dpa;
    }
    __syncthreads();
    TSTAMP(9);
    // ---- P1: p_a = Wt [a | ht] + bt:  d a[k] = sum_c Wt[c][k] dpa[c] (wave 1),  d ht[k] = sum_c Wt[c][64 + k] dpa[c] (wave 2)
    if (p.hybrid && (wave == 1 || wave == 2)) {
        const int half = wave == 1 ? 0 : kD;
        float wv[4][16];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 16; ++j) wv[c][j] = P[lo.Wt + (c * 16 + j) * 2 * kD + half + lane];
        float g = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 16; ++j) g = fmaf(wv[c][j], vec[c * 16 + j], g);
        sacc[half + lane] = g;
    }
    __syncthreads();
    TSTAMP(10);
    // ---- P2: a = sum alpha_i h_i,  alpha_i = w3 . s_i,  s_i = sigmoid(q1 + q2_i)
    if (wave == 0) {
        const float da = p.hybrid ? sacc[lane] : dpa;
        const float w3 = P[lo.w3 + lane];
        float dw3 = 0.0f, dq1 = 0.0f;
        for (int i = 0; i < l; ++i) {
            const float hi = h[i * kD + lane], si = sg_all[i * kD + lane];
            const float dalpha = wave_sum_f32(da * hi);
            dh[i * kD + lane] = vec[5 * kD + kMaxL + i] * da;
            dw3 = fmaf(dalpha, si, dw3);
            const float d = dalpha * w3 * si * (1.0f - si);
            du[i * kD + lane] = d;
            W[wl.DU + i * kD + lane] = d;
            W[wl.Hm + i * kD + lane] = hi;
            dq1 += d;
        }
        W[wl.dw3 + lane] = dw3;
        W[wl.dq1 + lane] = dq1;
        vec[kD + lane] = dq1;
    }
    __syncthreads();
    TSTAMP(11);
    // ---- P3: linear_two transposed per position, d h_i[k] += sum_c W2[c][k] du_i[c] (waves 1 ..), linear_one transposed (wave 0)
    float dht1 = 0.0f;
    {
        const float *__restrict__ Wm = P + (wave == 0 ? lo.W1 : lo.W2);
        const bool busy = wave == 0 || wave - 1 < l;
        if (busy) {
            float wv[4][16];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 16; ++j) wv[c][j] = Wm[(c * 16 + j) * kD + lane];
            if (wave == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int j = 0; j < 16; ++j) dht1 = fmaf(wv[c][j], vec[kD + c * 16 + j], dht1);
            } else {
                for (int i = wave - 1; i < l; i += NW - 1) {
                    float g = 0.0f;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int j4 = 0; j4 < 4; ++j4) {
                            const float4 d = *reinterpret_cast<const float4 *>(&du[i * kD + c * 16 + j4 * 4]);
                            g = fmaf(wv[c][j4 * 4 + 0], d.x, g);
                            g = fmaf(wv[c][j4 * 4 + 1], d.y, g);
                            g = fmaf(wv[c][j4 * 4 + 2], d.z, g);
                            g = fmaf(wv[c][j4 * 4 + 3], d.w, g);
                        }
                    dh[i * kD + lane] += g;
                }
            }
        }
    }
    __syncthreads();
    TSTAMP(12);
    // ---- P4: wave 0 — output attention layer (the first half of its parameter cancels in the softmax: gradient exactly 0), then
    //      ELU (from o alone: o > 0 <=> r > 0, else exp(r) = o + 1);  the other waves — M and the padded positions' zero rows
    float *dr = du;     // du is dead
    if (wave == 0) {
        dh[(l - 1) * kD + lane] += dht1 + (p.hybrid ? sacc[kD + lane] : 0.0f);
        const float c2 = P[lo.out_att + kD + lane];
        float dc2 = 0.0f;
        for (int i = 0; i < l; ++i) dO[i * kD + lane] = 0.0f;
        for (int i = 0; i < l; ++i) {
            const float g = dh[i * kD + lane];
            if (i < l - 1) {
                const float delta = o[i * kD + lane] - o[(i + 1) * kD + lane], v0 = vec[5 * kD + i];
                const float dz = wave_sum_f32(g * delta) * v0 * (1.0f - v0);
                dO[i * kD + lane] += v0 * g + dz * c2;
                dO[(i + 1) * kD + lane] += (1.0f - v0) * g - dz * c2;
                dc2 = fmaf(dz, delta, dc2);
            } else {
                dO[i * kD + lane] += g;
            }
        }
        W[wl.dc2 + lane] = dc2;
        for (int i = 0; i < l; ++i) {
            const float ov = o[i * kD + lane];
            const float d = dO[i * kD + lane] * (ov > 0.0f ? 1.0f : ov + 1.0f);
            dr[i * kD + lane] = d;
            W[wl.DR + i * kD + lane] = d;
        }
    } else {
        for (int r = wave - 1; r < L * H; r += NW - 1) {            // M rows; rows of padded positions: zeros, so the reduce kernel
            const int i = r / H;                                      // walks B x L rows flat
            W[wl.M + r * kD + lane] = i < l ? M[r * kD + lane] : 0.0f;
        }
        for (int i = l + wave - 1; i < L; i += NW - 1) {
            W[wl.DU + i * kD + lane] = 0.0f;
            W[wl.Hm + i * kD + lane] = 0.0f;
            W[wl.DR + i * kD + lane] = 0.0f;
        }
    }
    __syncthreads();
    TSTAMP(13);
    // ---- P5: one head per wave: d M_ih[k] = sum_c w[h d + k][c] dr_i[c] (lane == k), then the head's positional attention backwards
    if (wave < H) {
        const int hh = wave;
        const float a2 = P[lo.in_att + hh * 2 * kD + kD + lane];
        float4 wr[16];
        load_row(P + lo.w + (hh * kD + lane) * kD, wr);
        float g[kMaxL], dE[kMaxL];
#pragma unroll
        for (int i = 0; i < kMaxL; ++i) {
            dE[i] = 0.0f;
            g[i] = i < l ? dot_row(wr, dr + i * kD) : 0.0f;
        }
        float da2h = 0.0f;
#pragma unroll
        for (int i = 0; i < kMaxL; ++i) {
            if (i < l - 1) {
                const float delta = e[i * kD + lane] - e[(i + 1) * kD + lane] + 1.0f;
                const float w0 = vec[4 * kD + i * H + hh];
                const float dz = wave_sum_f32(g[i] * delta) * w0 * (1.0f - w0);
                dE[i] += w0 * g[i] + dz * a2;
                if (i + 1 < kMaxL) dE[i + 1] += (1.0f - w0) * g[i] - dz * a2;
                da2h = fmaf(dz, delta, da2h);
            } else if (i == l - 1) {
                dE[i] += g[i];
            }
        }
#pragma unroll
        for (int i = 0; i < kMaxL; ++i)
            if (i < l) dM[(i * H + hh) * kD + lane] = dE[i];          // this head's share of d e_i
        W[wl.da2h + hh * kD + lane] = da2h;
    }
    __syncthreads();
    TSTAMP(14);
    // ---- P6: the path's own table rows: plain stores into the workspace — the reduce kernel owns the table rows and adds them in
    //      (path, position) order.  d pm goes to the row the max-pool picked for this column.
    if (wave == 0) {
        for (int i = 0; i < L; ++i) {
            float d = 0.0f;
            if (i < l) {
                for (int hh = 0; hh < H; ++hh) d += dM[(i * H + hh) * kD + lane];
                if (st.arg == i) d += dpm;
            }
            W[wl.DE + i * kD + lane] = d;
        }
    }
    TSTAMP(15);
}

// The FUSED training form, one workgroup per path (round 2's kernel, rebuilt in round 3): one 16-wave workgroup does everything —
// forward chain (its independent pieces on separate waves), logits + CE on all 16 waves, backward chain.  Used when the split form
// below cannot be (many paths: S x paths beyond one dispatch round; a table of a single sweep batch).
__global__ __launch_bounds__(kPathWaves *kWave) void trust_path_train_kernel(const TrustArgs p, const TrainArgs tr,
                                                                            float *__restrict__ a2_out)
{
    extern __shared__ float4 s_raw[];
    float *s = reinterpret_cast<float *>(s_raw);
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const LdsLayout ll = lds_layout(p.L, p.H, true);
    const int l = path_len(p, b);
    const FwdState st = forward_train<kPathWaves>(p, ll, s, b, l, lane, wave, a2_out);
    __syncthreads();
    logits_ce<false>(p, tr, ll, s, b, 0);
    __syncthreads();
    backward_train<kPathWaves>(p, tr, ll, s, b, l, lane, wave, st);
}

// The SPLIT fused form: tr.S workgroups per path.  Every one of them runs the (cheap, deterministic) forward chain — the same
// instructions on the same inputs: the same bits, so the duplicate writes of a2 / the workspace state are benign — then sweeps its
// share of the user table; the path's last workgroup to arrive folds the shares and runs the backward chain (logits_ce<true>).
// The sweep is bounded by what ONE CU can pull (n_users x 256 B through a 64 B / clk L1: ~5 us at 3 185 users, 11 us achieved) —
// this is the form that puts a path's sweep on S CUs without a launch boundary.  Grid: S x pad8(B) workgroups, share-major, so
// that the shares of a path have the same workgroup id mod 8 (the same XCD, if the dispatcher deals workgroups round-robin —
// a locality hint only: correctness rests on the agent-scope accesses).
__global__ __launch_bounds__(kPathWaves *kWave) void trust_path_split_kernel(const TrustArgs p, const TrainArgs tr, float *__restrict__ a2_out,
                                                                            int b_pad)
{
    extern __shared__ float4 s_raw[];
    float *s = reinterpret_cast<float *>(s_raw);
    const int sp = blockIdx.x / b_pad, b = blockIdx.x - sp * b_pad;
    if (b >= p.B) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const LdsLayout ll = lds_layout(p.L, p.H, true);
    const int l = path_len(p, b);
    if (threadIdx.x == 0)
        *reinterpret_cast<unsigned long long *>(s + ll.tk) = __hip_atomic_load(tr.tickets + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const FwdState st = forward_train<kPathWaves>(p, ll, s, b, l, lane, wave, a2_out);
    __syncthreads();
    if (!logits_ce<true>(p, tr, ll, s, b, sp)) return;
    __syncthreads();
    backward_train<kPathWaves>(p, tr, ll, s, b, l, lane, wave, st);
}

// Evaluation form (flag 2): one wave per path, forward chain only.
template <bool TRAIN>
__global__ __launch_bounds__(kWave) void trust_path_kernel(const TrustArgs p, const TrainArgs tr, float *__restrict__ a2_out)
{
    static_assert(!TRAIN, "training runs the five-launch form");
    extern __shared__ float4 s_raw[];
    float *s = reinterpret_cast<float *>(s_raw);
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const LdsLayout ll = lds_layout(p.L, p.H, false);
    const int l = path_len(p, b);
    forward_head(p, ll, s, b, l, lane);
    wave_sync();
    float acc[kMaxL];
    forward_mw(p, ll, s, l, lane, 0, p.H * kD, acc);
    forward_tail(p, ll, s, b, l, lane, acc, a2_out);
}

// ------------------------------------------------------------------------------------------------------------ reductions
struct ReduceArgs {
    const float *dscore, *a2, *loss_b, *ws;
    const int64_t *seq, *seq_l;
    float *grad_table, *grad_P, *loss_out;
    int n_users, B, L, H, hybrid, loss_accumulate;
    int users_logits;                  // 1: add sum_b d score[b, u] a2[b] here (fused form); 0: trust_ce_kernel already did (tiled form)
    int blocks_users, blocks_mat;      // block ranges: [0, blocks_users) user rows, then the weight tiles, then one vector block
};

__device__ __forceinline__ int clamp_len(int64_t l, int L)
{
    return l < 1 ? 1 : (l > L ? L : (int)l);
}

__global__ __launch_bounds__(256) void trust_reduce_kernel(const ReduceArgs a)
{
    const Layout lo = layout(a.H);
    const WsLayout wl = ws_layout(a.L, a.H);
    const int t = threadIdx.x;
    int blk = blockIdx.x;
    if (blk < a.blocks_users) {
        // d E[u, :] += sum_b d score[b, u] a2[b, :]  (one wave per user row; this launch owns the rows)
        const int lane = t & 63;
        const int n_pos = a.B * a.L;
        for (int u = blk * 4 + (t >> 6); u <= a.n_users; u += a.blocks_users * 4) {     // (row n_users = the pad row: paths only)
            float acc = 0.0f;          // (tiled form: the logits' share of the row was added by trust_ce_kernel, which owned it in ITS launch)
            if (a.users_logits && u < a.n_users)
                for (int b0 = 0; b0 < a.B; b0 += 8) {
                    float dv[8], xv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        dv[j] = b0 + j < a.B ? a.dscore[(size_t)(b0 + j) * a.n_users + u] : 0.0f;
                        xv[j] = b0 + j < a.B ? a.a2[(size_t)(b0 + j) * kD + lane] : 0.0f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc = fmaf(dv[j], xv[j], acc);
                }
            // the rows of the paths that pass through u (model_expert_s.py:175: embedding_user(inputs) under autograd), in
            // (path, position) order: 64 positions per load, matches walked in ascending order
            bool any = false;
            for (int base = 0; base < n_pos; base += 64) {
                const int idx = base + lane;
                bool match = false;
                if (idx < n_pos) {
                    const int b = idx / a.L, i = idx - b * a.L;
                    match = i < clamp_len(a.seq_l[b], a.L) && a.seq[idx] == (int64_t)u;
                }
                unsigned long long m = __ballot(match);
                while (m) {
                    const int idx2 = base + (int)__builtin_ctzll(m);
                    m &= m - 1;
                    const int b = idx2 / a.L, i = idx2 - b * a.L;
                    acc += a.ws[(size_t)b * wl.stride + wl.DE + i * kD + lane];
                    any = true;
                }
            }
            if (any || (a.users_logits && u < a.n_users)) a.grad_table[(size_t)u * kD + lane] += acc;
        }
        return;
    }
    blk -= a.blocks_users;
    if (blk < a.blocks_mat) {
        // weight gradients: out[x][y] = sum over rows of A[row][x] * Bm[row][y]; a block covers 4 x-values x 64 y-values
        //   W1 [c][k]   = sum_b       dq1_b[c] ht_b[k]            16 blocks
        //   W2 [c][k]   = sum_{b,i}   du_bi[c] h_bi[k]            16 blocks
        //   Wt [c][k']  = sum_b       dpa_b[c] [a_b | ht_b][k']   32 blocks (one 64-wide half of k' each)
        //   w  [k][c]   = sum_{b,i}   M_bi[k]  dr_bi[c]           16 H blocks
        const int y = t & 63, xs = t >> 6;
        int x, offA, offB, out, ldA = 0;
        bool per_pos;
        if (blk < 16) {
            x = blk * 4 + xs; offA = wl.dq1 + x; offB = wl.ht + y; out = lo.W1 + x * kD + y; per_pos = false;
        } else if (blk < 32) {
            x = (blk - 16) * 4 + xs; offA = wl.DU + x; offB = wl.Hm + y; out = lo.W2 + x * kD + y; per_pos = true; ldA = kD;
        } else if (blk < 64) {
            const int half = (blk - 32) & 1;
            x = ((blk - 32) >> 1) * 4 + xs; offA = wl.dpa + x; offB = (half ? wl.ht : wl.a) + y;
            out = lo.Wt + x * 2 * kD + half * kD + y; per_pos = false;
        } else {
            x = (blk - 64) * 4 + xs; offA = wl.M + x; offB = wl.DR + y; out = lo.w + x * kD + y; per_pos = true; ldA = a.H * kD;
        }
        float acc = 0.0f;
        const int rows = per_pos ? a.B * a.L : a.B, per = per_pos ? a.L : 1;       // padded positions hold zeros
        // 32 rows (64 loads) in flight per thread: the per-position products run over B L rows (90 for 15 paths of 6), and every
        // batch is a dependent round trip — six of them at 16 rows per batch were most of this kernel's 12 us
        constexpr int kRowsInFlight = 32;
        for (int r0 = 0; r0 < rows; r0 += kRowsInFlight) {
            float av[kRowsInFlight], bv[kRowsInFlight];
#pragma unroll
            for (int j = 0; j < kRowsInFlight; ++j) {
                const int r = r0 + j, b = r / per, i = r - b * per;
                const float *W = a.ws + (size_t)b * wl.stride;
                av[j] = r < rows ? W[offA + i * ldA] : 0.0f;
                bv[j] = r < rows ? W[offB + i * kD] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < kRowsInFlight; ++j) acc = fmaf(av[j], bv[j], acc);
        }
        a.grad_P[out] = acc;
        return;
    }
    // vector gradients (sums over paths of per-path vectors) and the loss
    for (int k = t; k < (7 + a.H) * kD; k += 256) {
        const int which = k / kD, c = k - which * kD;
        int off;
        switch (which) {
            case 0: off = wl.dq1; break;       // b1 and b2
            case 1: off = wl.dpa; break;       // bt
            case 2: off = wl.dw3; break;
            case 3: off = wl.dc2; break;
            case 4: off = wl.vpa; break;
            case 5: off = wl.vpm; break;
            case 6: off = -1; break;           // (zero halves, below)
            default: off = wl.da2h + (which - 7) * kD; break;
        }
        float sum = 0.0f;
        if (off >= 0)
            for (int b0 = 0; b0 < a.B; b0 += 16) {          // 16 independent loads in flight, added in path order
                float x[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = b0 + j < a.B ? a.ws[(size_t)(b0 + j) * wl.stride + off + c] : 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) sum += x[j];
            }
        switch (which) {
            case 0: a.grad_P[lo.b1 + c] = sum; a.grad_P[lo.b2 + c] = sum; break;
            case 1: a.grad_P[lo.bt + c] = sum; break;
            case 2: a.grad_P[lo.w3 + c] = sum; break;
            case 3: a.grad_P[lo.out_att + kD + c] = sum; break;
            case 4: a.grad_P[lo.att_t + c * 2] = sum; a.grad_P[lo.att_t + c * 2 + 1] = -sum; break;
            case 5: a.grad_P[lo.att_t + (kD + c) * 2] = sum; a.grad_P[lo.att_t + (kD + c) * 2 + 1] = -sum; break;
            case 6:                                                  // the a1 halves cancel in the softmax: exactly 0
                a.grad_P[lo.out_att + c] = 0.0f;
                for (int hh = 0; hh < a.H; ++hh) a.grad_P[lo.in_att + hh * 2 * kD + c] = 0.0f;
                break;
            default: a.grad_P[lo.in_att + (which - 7) * 2 * kD + kD + c] = sum; break;
        }
    }
    if (t == 0 && a.loss_out) {
        float sum = 0.0f;
        for (int b = 0; b < a.B; ++b) sum += a.loss_b[b];
        sum /= (float)a.B;
        *a.loss_out = a.loss_accumulate ? *a.loss_out + sum : sum;
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

constexpr int kTileUsers = 32;       // the workspace's shares area is sized in units of 32 users (spex_trust_workspace_floats: ABI)
static int n_user_tiles(int64_t n_rows) { return (int)((n_rows - 1 + kTileUsers - 1) / kTileUsers); }

// per-path blocks | per-tile partials of d a2 [n_tiles][B][64] | per-tile (max, sum-exp) [n_tiles][B][2]
extern "C" int64_t spex_trust_workspace_floats(int32_t B, int32_t L, int32_t d, int32_t n_heads, int64_t n_rows)
{
    if (d != kD || L < 1 || L > kMaxL || n_heads < 1 || n_heads > kMaxH || B < 0 || n_rows < 2) return -1;
    return (int64_t)B * ws_layout(L, n_heads).stride + (int64_t)n_user_tiles(n_rows) * B * (kD + 2);
}

extern "C" int spex_trust_head_fwd_f32(const float *table, int64_t n_rows, const float *params, const int64_t *seq,
                                       const int64_t *seq_l, int32_t B, int32_t L, int32_t d, int32_t n_heads, int32_t hybrid,
                                       float *a2_out, void *stream)
{
    if (int rc = check_head("spex_trust_head_fwd_f32", table, n_rows, params, seq, seq_l, B, L, d, n_heads)) return rc;
    SPEX_CHECK_ARG(a2_out, "spex_trust_head_fwd_f32: NULL output");
    if (B == 0) return SPEX_OK;
    const TrustArgs p{table, n_rows, params, seq, seq_l, B, L, n_heads, hybrid};
    const TrainArgs tr{};
    const size_t lds = (size_t)lds_layout(L, n_heads, false).total * sizeof(float);
    hipLaunchKernelGGL(trust_path_kernel<false>, dim3((unsigned)B), dim3(kWave), lds, (hipStream_t)stream, p, tr, a2_out);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_trust_head_train_f32(const float *table, int64_t n_rows, const float *params, const int64_t *seq,
                                         const int64_t *seq_l, const int64_t *targets, int32_t B, int32_t L, int32_t d,
                                         int32_t n_heads, int32_t hybrid, float scale, const float *scale_dev, float *a2,
                                         float *dscore, float *loss_b, float *ws, float *loss_out, int32_t loss_accumulate,
                                         float *grad_params, float *grad_table, void *stream)
{
    return spex::trust_head_train(table, n_rows, params, seq, seq_l, targets, B, L, d, n_heads, hybrid, scale, scale_dev, a2, dscore, loss_b, ws,
                                  loss_out, loss_accumulate, grad_params, grad_table, kMaxSplit, stream);
}

// split_cap: the most workgroups per path the fused kernel may use.  The public entry passes kMaxSplit (the head has the chip to
// itself); the two-stream dual-task step passes 1 — there the head runs BESIDE the rec branch's whole-graph launches, which bound the
// step, and every CU the head takes is taken from them (Epinion2 step 93.5 us with one workgroup per path, 95.7 with eight; the
// Weibo shape 139.0 / 147.8).
int spex::trust_head_train(const float *table, int64_t n_rows, const float *params, const int64_t *seq, const int64_t *seq_l,
                           const int64_t *targets, int32_t B, int32_t L, int32_t d, int32_t n_heads, int32_t hybrid, float scale,
                           const float *scale_dev, float *a2, float *dscore, float *loss_b, float *ws, float *loss_out,
                           int32_t loss_accumulate, float *grad_params, float *grad_table, int32_t split_cap, void *stream)
{
    if (int rc = check_head("spex_trust_head_train_f32", table, n_rows, params, seq, seq_l, B, L, d, n_heads)) return rc;
    SPEX_CHECK_ARG(targets && a2 && dscore && loss_b && ws && grad_params && grad_table, "spex_trust_head_train_f32: NULL pointer");
    SPEX_CHECK_ARG(n_rows >= 2, "spex_trust_head_train_f32: the table needs at least one user row besides the pad row");
    SPEX_CHECK_ARG((((uintptr_t)a2 | (uintptr_t)ws) & 15) == 0, "spex_trust_head_train_f32: a2 and ws must be 16-byte aligned");
    if (B == 0) return SPEX_OK;
    const int n_users = (int)(n_rows - 1);                           // logits against table[:-1]
    const TrustArgs p{table, n_rows, params, seq, seq_l, B, L, n_heads, hybrid};
    TrainArgs tr{targets, n_users, scale, scale_dev, a2, dscore, loss_b, ws, grad_table, 1, nullptr, nullptr, nullptr, 0u};
    const size_t lds = (size_t)lds_layout(L, n_heads, true).total * sizeof(float);
    SPEX_CHECK_ARG(lds <= 64 * 1024, "spex_trust_head_train_f32: LDS %zu bytes", lds);
    const int n_tiles = n_user_tiles(n_rows);
    float *part_da2 = ws + (size_t)B * ws_layout(L, n_heads).stride;
    float *part_ms = part_da2 + (size_t)n_tiles * B * kD;
    hipStream_t st = (hipStream_t)stream;
    // The fused kernel sweeps the user table once PER PATH (n_users x 256 B per path — through one CU in its one-workgroup form,
    // through S CUs in the split form).  (Rounds 2-3 also carried a five-launch form with the logits tiled over the user table — read
    // once for all paths on ~n_users / 32 workgroups; since the split form exists it won at no measured size — 3 185 users x 15 paths
    // 48.5 vs 59.2 us, 26 000 x 15: 74 vs 93, 100 000 x 15: 182 vs 687 — and was removed in round 4.)
    {
        // the split form: S workgroups per path (the shares live behind the per-path blocks of the workspace: [S][B][64] and [S][B][2]
        // fit in the [n_tiles][B][...] area spex_trust_workspace_floats sizes, one more tile's cells holding the B tickets)
        // How many: as many as there are sweep batches, up to kMaxSplit, while S x paths fits one dispatch round of the 256 CUs
        // (every workgroup repeats the forward chain — free on an idle CU, a second round when there is none — and the fold costs
        // ~5 us of publish / ticket / fetch round trips through memory whatever S is).  Measured (tools/trust_forms_time.py, us per
        // call incl. the reduce launch; S = 1 / 3 / 4 / 6 / 8): 3 185 users x 15 paths 51.5 / 49.7 / 48.2 / 47.8 / 46.4; 6 812 x 15:
        // 62.0 / 55.5 / 53.6 / 51.7 / 50.4; 6 812 x 30: - / 65.1 / 63.6 / 62.0 / 60.0; 6 812 x 45: 81.9 / - / 73.8 (S = 5) / - / 89.9
        // (360 workgroups: a second round); 26 000 x 15: 128.6 / - / 81.3 (S = 5) / - / 75.7; 3 185 x 192: 156 / 176 (S = 2) / 193 / - / 232.
        const int b_pad8 = (B + 7) & ~7;
        const char *split_env = getenv("SPEX_TRUST_SPLIT");
        int S = split_env && split_env[0] ? atoi(split_env) : (split_cap < kMaxSplit ? split_cap : kMaxSplit);
        const int n_batches = (n_users + kSweepBatchRows - 1) / kSweepBatchRows;
        if (!(split_env && split_env[0])) S = S < kCUs / b_pad8 ? S : kCUs / b_pad8;
        S = S < n_batches ? S : n_batches;
        S = S < kMaxSplit ? S : kMaxSplit;
        if (S >= 2 && n_tiles >= S + 1) {
            static std::atomic<uint32_t> call_tag{0x2f000000u};
            tr.S = S;
            tr.part_da2 = part_da2;
            tr.part_ms = part_ms;
            tr.tickets = reinterpret_cast<unsigned long long *>(part_ms + (size_t)2 * S * B);
            tr.tag = call_tag.fetch_add(1u) + 1u;
            const int b_pad = b_pad8;
            hipLaunchKernelGGL(trust_path_split_kernel, dim3((unsigned)(S * b_pad)), dim3(kPathWaves * kWave), lds, st, p, tr, a2, b_pad);
        } else {
            hipLaunchKernelGGL(trust_path_train_kernel, dim3((unsigned)B), dim3(kPathWaves * kWave), lds, st, p, tr, a2);
        }
    }
    SPEX_HIP(hipGetLastError());
    ReduceArgs r{dscore, a2, loss_b, ws, seq, seq_l, grad_table, grad_params, loss_out, n_users, B, L, n_heads, hybrid, loss_accumulate,
                 1, 0, 0};
    r.blocks_users = (n_users + 3) / 4 < 1024 ? (n_users + 3) / 4 : 1024;
    r.blocks_mat = 64 + 16 * n_heads;
    hipLaunchKernelGGL(trust_reduce_kernel, dim3((unsigned)(r.blocks_users + r.blocks_mat + 1)), dim3(256), 0, (hipStream_t)stream, r);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

#ifdef SPEX_STAMPS
extern "C" int spex_debug_trust_stamps(unsigned long long *out)
{
    SPEX_HIP(hipDeviceSynchronize());
    SPEX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tstamps), sizeof(unsigned long long) * 32));
    return SPEX_OK;
}
#endif
