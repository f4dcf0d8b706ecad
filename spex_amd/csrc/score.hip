// Negative-sampled scoring kernels: gather + dot + BCE-with-logits (reference semantics) and BPR (north-star
// extension), forward and backward fused, one wavefront per sample / triple.
//
// Replaces LightGCN.forward, LightGCN_SPEX/code/utility1/model.py:111-121, and NGCF compute_rec_loss,
// NGCF_SPEX/code/main_rec.py:89-100, together with their autograd backward into the gathered tables.
//
// A sample touches two (BPR: three) 256-byte rows and writes as many gradient rows: ~1 KB (1.5 KB) of traffic for 128
// (192) flops — HBM/L2-bound gather/scatter.  Lane == embedding column: each row access is one coalesced 256-byte
// wave load, the dot product is a 6-step cross-lane butterfly, the gradient rows are added with hardware float
// atomics shaped as one contiguous 256-byte wave instruction per row (the full-rate shape on MI355X; duplicates of a
// user within a batch accumulate correctly).
#include "spex_common.h"

using namespace spex;

namespace {

// 1024-thread workgroups: the loss is reduced per workgroup and added with ONE atomic to a single address; those
// same-address atomics serialise in the L2 (~5 ns each), so 4-wave workgroups made a 16 k-sample launch 2.5x slower
// (35 vs 14 us) than the same launch without a loss.
constexpr int kScoreWaves = 16;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Per-block loss reduction: lane 0 of each wave holds a partial; one atomic per block.
__device__ __forceinline__ void block_loss_add(float wave_partial, float *loss_sum)
{
    __shared__ float s_part[kScoreWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (lane == 0) s_part[wave] = wave_partial;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < kScoreWaves; ++w) t += s_part[w];
        atomicAdd(loss_sum, t);
    }
}

__global__ __launch_bounds__(kWave *kScoreWaves) void score_bce_kernel(
    const float *__restrict__ users, const float *__restrict__ items, int ldu, int ldi, const int64_t *__restrict__ u_idx,
    const int64_t *__restrict__ i_idx, const float *__restrict__ labels, int B, int d, int64_t n_user_rows,
    int64_t n_item_rows, float *__restrict__ gamma, float *loss_sum, float *grad_users, float *grad_items,
    float grad_scale)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_global = blockIdx.x * kScoreWaves + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * kScoreWaves;
    float lsum = 0.0f;
    for (int b = wave_global; b < B; b += n_waves) {
        const int64_t u = u_idx[b], it = i_idx[b];
        if (u < 0 || u >= n_user_rows || it < 0 || it >= n_item_rows) {  // never gather out of bounds
            if (lane == 0 && gamma) gamma[b] = __int_as_float(0x7fc00000);
            continue;
        }
        const float *pu = users + (size_t)u * ldu, *pi = items + (size_t)it * ldi;
        float s = 0.0f;
        for (int c = lane; c < d; c += kWave) s = fmaf(pu[c], pi[c], s);
        const float x = wave_sum(s);
        if (lane == 0 && gamma) gamma[b] = x;
        if (labels) {
            const float y = labels[b];
            lsum += fmaxf(x, 0.0f) - x * y + log1pf(expf(-fabsf(x)));
            if (grad_users) {
                const float dg = (sigmoid_f(x) - y) * grad_scale;
                float *gu = grad_users + (size_t)u * ldu, *gi = grad_items + (size_t)it * ldi;
                for (int c = lane; c < d; c += kWave) {
                    atomicAdd(gu + c, dg * pi[c]);
                    atomicAdd(gi + c, dg * pu[c]);
                }
            }
        }
    }
    if (labels && loss_sum) block_loss_add(lsum, loss_sum);
}

// a_coef: multiplies sigmoid(x) (SGD: -lr/T, autograd: grad_scale); b_coef: multiplies the read row (SGD: -lr*reg/T).
// A wave owns `per_wave` CONSECUTIVE triples and keeps the running update of the current user row and of the current
// positive-item row in registers, flushing one with a single 256-byte atomic instruction only when the index changes.
// Triples arrive in any order — nothing is assumed — but the sampler's natural order (num_ng negatives per positive,
// positives sorted by user: dataloader.py:250-265) then costs ~1.3 atomic row updates per triple instead of 3, and the
// kernel is bound by the float-atomic rate (~1.3 TB/s chip-wide), not by HBM.  per_wave == 1 for small batches (latency).
// Measured at T ~ 1 M on Epinion2's tables: 1.5-1.6 G triples/s in random order, 2.25 G in sampler order (hot rows then
// serialise in the L2: a user's ~330 consecutive triples are flushed by ~20 waves at about the same time); fetching a
// run's rows ahead of use (48 loads in flight, 98 VGPRs) changed neither figure.
__global__ __launch_bounds__(kWave *kScoreWaves) void bpr_kernel(
    const float *__restrict__ U_read, const float *__restrict__ I_read, float *U_w, float *I_w,
    const int64_t *__restrict__ u_idx, const int64_t *__restrict__ p_idx, const int64_t *__restrict__ n_idx, int64_t T,
    int d, int64_t n_user_rows, int64_t n_item_rows, float a_coef, float b_coef, float *loss_sum, int per_wave)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave_global = (int64_t)blockIdx.x * kScoreWaves + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * kScoreWaves;
    float lsum = 0.0f;
    if (d == kWave && U_w) {   // the tuned form: one column per lane, rows held in registers across triples
        for (int64_t t0 = wave_global * per_wave; t0 < T; t0 += n_waves * per_wave) {
            const int64_t t1 = (t0 + per_wave < T) ? t0 + per_wave : T;
            int64_t cu = -1, cp = -1;
            float uu = 0.0f, vp = 0.0f, acc_u = 0.0f, acc_p = 0.0f;
            for (int64_t t = t0; t < t1; ++t) {
                const int64_t u = u_idx[t], ip = p_idx[t], in = n_idx[t];
                if (u < 0 || u >= n_user_rows || ip < 0 || ip >= n_item_rows || in < 0 || in >= n_item_rows) continue;
                if (u != cu) {
                    if (cu >= 0) atomicAdd(U_w + (size_t)cu * kWave + lane, acc_u);
                    cu = u;
                    uu = U_read[(size_t)u * kWave + lane];
                    acc_u = 0.0f;
                }
                if (ip != cp) {
                    if (cp >= 0) atomicAdd(I_w + (size_t)cp * kWave + lane, acc_p);
                    cp = ip;
                    vp = I_read[(size_t)ip * kWave + lane];
                    acc_p = 0.0f;
                }
                const float vn = I_read[(size_t)in * kWave + lane];
                const float x = wave_sum(uu * vn) - wave_sum(uu * vp);   // neg_score - pos_score
                lsum += softplus_f(x);
                const float a = a_coef * sigmoid_f(x);
                acc_u += a * (vn - vp) + b_coef * uu;
                acc_p += -a * uu + b_coef * vp;
                atomicAdd(I_w + (size_t)in * kWave + lane, a * uu + b_coef * vn);
            }
            if (cu >= 0) atomicAdd(U_w + (size_t)cu * kWave + lane, acc_u);
            if (cp >= 0) atomicAdd(I_w + (size_t)cp * kWave + lane, acc_p);
        }
    } else {
        for (int64_t t = wave_global; t < T; t += n_waves) {
            const int64_t u = u_idx[t], ip = p_idx[t], in = n_idx[t];
            if (u < 0 || u >= n_user_rows || ip < 0 || ip >= n_item_rows || in < 0 || in >= n_item_rows) continue;
            const float *pu = U_read + (size_t)u * d, *pp = I_read + (size_t)ip * d, *pn = I_read + (size_t)in * d;
            float sp = 0.0f, sn = 0.0f;
            for (int c = lane; c < d; c += kWave) {
                const float uu = pu[c];
                sp = fmaf(uu, pp[c], sp);
                sn = fmaf(uu, pn[c], sn);
            }
            const float x = wave_sum(sn) - wave_sum(sp);  // neg_score - pos_score
            lsum += softplus_f(x);
            if (U_w) {
                const float a = a_coef * sigmoid_f(x);
                for (int c = lane; c < d; c += kWave) {
                    const float uu = pu[c], vp = pp[c], vn = pn[c];
                    atomicAdd(U_w + (size_t)u * d + c, a * (vn - vp) + b_coef * uu);
                    atomicAdd(I_w + (size_t)ip * d + c, -a * uu + b_coef * vp);
                    atomicAdd(I_w + (size_t)in * d + c, a * uu + b_coef * vn);
                }
            }
        }
    }
    if (loss_sum) block_loss_add(lsum, loss_sum);
}

// consecutive triples per wave: 1 until the batch fills the chip a few times over, then up to 16
inline int bpr_per_wave(int64_t T)
{
    int64_t k = T / (256 * 64);
    return (int)(k < 1 ? 1 : (k > 16 ? 16 : k));
}

// Owner-computes exchange of a batch's rows on a row-partitioned table (spex_amd/dist.py): every rank contributes the
// rows it owns to a zero-filled buffer (summed by one small all-reduce), and after scoring adds the update rows it
// owns back into its shard.  One wave per listed position; `pos` are positions in the padded global layout, the rank
// owns [lo, lo + n_local).
__global__ __launch_bounds__(kWave *kScoreWaves) void gather_owned_rows_kernel(const float *__restrict__ table,
                                                                              const int64_t *__restrict__ pos, int64_t K,
                                                                              int64_t lo, int64_t n_local, int d,
                                                                              float *__restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    for (int64_t k = (int64_t)blockIdx.x * kScoreWaves + (threadIdx.x >> 6); k < K; k += (int64_t)gridDim.x * kScoreWaves) {
        const int64_t p = pos[k] - lo;
        const bool own = p >= 0 && p < n_local;
        for (int c = lane; c < d; c += kWave) out[(size_t)k * d + c] = own ? table[(size_t)p * d + c] : 0.0f;
    }
}

__global__ __launch_bounds__(kWave *kScoreWaves) void scatter_add_owned_rows_kernel(float *upd, const int64_t *__restrict__ pos,
                                                                                   int64_t K, int64_t lo, int64_t n_local,
                                                                                   int d, float *table, int clear)
{
    const int lane = threadIdx.x & (kWave - 1);
    for (int64_t k = (int64_t)blockIdx.x * kScoreWaves + (threadIdx.x >> 6); k < K; k += (int64_t)gridDim.x * kScoreWaves) {
        const int64_t p = pos[k] - lo;
        const bool own = p >= 0 && p < n_local;
        for (int c = lane; c < d; c += kWave) {
            if (own) atomicAdd(table + (size_t)p * d + c, upd[(size_t)k * d + c]);   // positions repeat within a batch
            if (clear) upd[(size_t)k * d + c] = 0.0f;
        }
    }
}

inline unsigned grid_for(int64_t waves_wanted)
{
    int64_t blocks = (waves_wanted + kScoreWaves - 1) / kScoreWaves;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;  // 8 blocks per CU, grid-stride beyond
    return (unsigned)blocks;
}

}  // namespace

extern "C" int spex_score_bce_f32(const float *users, const float *items, int32_t ldu, int32_t ldi,
                                  int64_t n_user_rows, int64_t n_item_rows, const int64_t *u_idx, const int64_t *i_idx,
                                  const float *labels, int32_t B, int32_t d, float *gamma, float *loss_sum,
                                  float *grad_users, float *grad_items, float grad_scale, void *stream)
{
    SPEX_CHECK_ARG(users && items && u_idx && i_idx, "spex_score_bce_f32: NULL table or index pointer");
    SPEX_CHECK_ARG(B >= 0 && d >= 1 && ldu >= d && ldi >= d, "spex_score_bce_f32: B=%d d=%d ldu=%d ldi=%d", B, d, ldu, ldi);
    SPEX_CHECK_ARG(n_user_rows >= 0 && n_item_rows >= 0, "spex_score_bce_f32: negative table size");
    SPEX_CHECK_ARG((grad_users == nullptr) == (grad_items == nullptr), "spex_score_bce_f32: give both grad tables or neither");
    SPEX_CHECK_ARG(!grad_users || labels, "spex_score_bce_f32: gradients need labels");
    SPEX_CHECK_ARG(gamma || labels, "spex_score_bce_f32: nothing to compute");
    if (B == 0) return SPEX_OK;
    hipLaunchKernelGGL(score_bce_kernel, dim3(grid_for(B)), dim3(kWave * kScoreWaves), 0, (hipStream_t)stream, users,
                       items, ldu, ldi, u_idx, i_idx, labels, B, d, n_user_rows, n_item_rows, gamma, loss_sum, grad_users,
                       grad_items, grad_scale);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_bpr_sgd_step_f32(const float *U_read, const float *I_read, float *U_w, float *I_w,
                                     int64_t n_user_rows, int64_t n_item_rows, const int64_t *u, const int64_t *i_pos,
                                     const int64_t *i_neg, int64_t T, int32_t d, float lr, float reg, float *loss_sum,
                                     void *stream)
{
    SPEX_CHECK_ARG(U_read && I_read && U_w && I_w && u && i_pos && i_neg, "spex_bpr_sgd_step_f32: NULL pointer");
    SPEX_CHECK_ARG(T >= 0 && d >= 1, "spex_bpr_sgd_step_f32: T=%lld d=%d", (long long)T, d);
    if (T == 0) return SPEX_OK;
    const int per_wave = bpr_per_wave(T);
    hipLaunchKernelGGL(bpr_kernel, dim3(grid_for((T + per_wave - 1) / per_wave)), dim3(kWave * kScoreWaves), 0,
                       (hipStream_t)stream, U_read, I_read, U_w, I_w, u, i_pos, i_neg, T, d, n_user_rows, n_item_rows,
                       -lr / (float)T, -lr * reg / (float)T, loss_sum, per_wave);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_bpr_loss_f32(const float *users, const float *items, int64_t n_user_rows, int64_t n_item_rows,
                                 const int64_t *u, const int64_t *i_pos, const int64_t *i_neg, int64_t T, int32_t d,
                                 float *loss_sum, float *grad_users, float *grad_items, float grad_scale, void *stream)
{
    SPEX_CHECK_ARG(users && items && u && i_pos && i_neg, "spex_bpr_loss_f32: NULL pointer");
    SPEX_CHECK_ARG(T >= 0 && d >= 1, "spex_bpr_loss_f32: T=%lld d=%d", (long long)T, d);
    SPEX_CHECK_ARG((grad_users == nullptr) == (grad_items == nullptr), "spex_bpr_loss_f32: give both grad tables or neither");
    if (T == 0) return SPEX_OK;
    const int per_wave = (grad_users && d == kWave) ? bpr_per_wave(T) : 1;
    hipLaunchKernelGGL(bpr_kernel, dim3(grid_for((T + per_wave - 1) / per_wave)), dim3(kWave * kScoreWaves), 0,
                       (hipStream_t)stream, users, items, grad_users, grad_items, u, i_pos, i_neg, T, d, n_user_rows,
                       n_item_rows, grad_scale, 0.0f, loss_sum, per_wave);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_gather_owned_rows_f32(const float *table, const int64_t *pos, int64_t K, int64_t lo, int64_t n_local,
                                          int32_t d, float *out, void *stream)
{
    SPEX_CHECK_ARG((table || n_local == 0) && pos && out, "spex_gather_owned_rows_f32: NULL pointer");
    SPEX_CHECK_ARG(K >= 0 && n_local >= 0 && d >= 1, "spex_gather_owned_rows_f32: K=%lld n_local=%lld d=%d", (long long)K,
                   (long long)n_local, d);
    if (K == 0) return SPEX_OK;
    hipLaunchKernelGGL(gather_owned_rows_kernel, dim3(grid_for(K)), dim3(kWave * kScoreWaves), 0, (hipStream_t)stream, table, pos,
                       K, lo, n_local, d, out);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_scatter_add_owned_rows_f32(float *upd, const int64_t *pos, int64_t K, int64_t lo, int64_t n_local,
                                               int32_t d, float *table, int32_t clear_upd, void *stream)
{
    SPEX_CHECK_ARG(upd && pos && (table || n_local == 0), "spex_scatter_add_owned_rows_f32: NULL pointer");
    SPEX_CHECK_ARG(K >= 0 && n_local >= 0 && d >= 1, "spex_scatter_add_owned_rows_f32: K=%lld n_local=%lld d=%d", (long long)K,
                   (long long)n_local, d);
    if (K == 0) return SPEX_OK;
    hipLaunchKernelGGL(scatter_add_owned_rows_kernel, dim3(grid_for(K)), dim3(kWave * kScoreWaves), 0, (hipStream_t)stream, upd,
                       pos, K, lo, n_local, d, table, clear_upd);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
