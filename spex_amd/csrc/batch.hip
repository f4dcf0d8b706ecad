// The batch-sized middle of the exact LightGCN training step as ONE launch.
//
// Between the dense forward layers and the dense backward layers the step only touches the batch's rows
// (LightGCN_SPEX/code/utility1/model.py:91-97,111-121 forward, autograd backward): the last layer at the 2B rows of the
// batch, the layer mean there, B dot products + BCE, the 2B gradient rows, and the first backward product A^T g in push
// form.  Issued as three launches (spex_spmm_rowlist_f32 -> spex_score_bce_slots_f32 -> spex_spmm_push_batch_f32) each
// costs a ramp plus its own chain of dependent round trips: 6.1 + 5.2 + 9 us for ~14 k gathers, 256 dot products and
// ~14 k row atomics on Epinion2.  spex_lightgcn_batch_f32 does the three in one kernel:
//
//   workgroup (sample b, part p), 16 waves: waves 0-7 own the sample's user row, waves 8-15 its item row
//     1. last layer at both rows — the row's 64-entry segments dealt to the half's waves exactly as the row-list kernel
//        deals them to 16 (virtual wave v = w and w + 8 on wave w, separate accumulators), segment sums through LDS in
//        segment order: bit-identical to spex_spmm_rowlist_f32 / the main kernel for rows of <= 1024 entries;
//     2. light = (running sum + y) / (L + 1) for both rows -> LDS; every wave forms x = <light_u, light_i>, the sample's loss
//        share and its own row's gradient g = (sigmoid(x) - label) / B * (the other row);
//     3. push: out[col[e]] += val[e] * g / (L + 1) over the stored entries of the row in A^T, 16-entry runs dealt over
//        (part, wave) like spex_spmm_push_batch_f32 deals them, plus out[row] += g / (L + 1) and the dense
//        d loss / d light row (added by part 0 only).
//   `parts` workgroups share a sample so that a hub row's ~1 000 row atomics spread over several CUs; each part repeats
//   the (cheap, L2-resident) forward of its sample instead of reading it back from another workgroup.
#include "spex_common.h"

using namespace spex;

namespace {

__device__ __forceinline__ float lane_bcast(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int kHalfWaves = kWgWaves / 2;     // waves per row

__global__ __launch_bounds__(kWave *kWgWaves) void lightgcn_batch_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const int32_t *__restrict__ t_rowptr, const int32_t *__restrict__ t_col, const float *__restrict__ t_val, int n_rows,
    int n_user_rows, const float *__restrict__ X, const float *__restrict__ acc_in, float acc_div,
    const int64_t *__restrict__ users, const int64_t *__restrict__ items, const float *__restrict__ labels, int parts,
    float grad_scale, float push_scale, float *loss_sum, float *g_out, float *G)
{
    __shared__ float s_part[2][kWgWaves][kWave];   // [row half][virtual wave = segment mod 16][column]
    __shared__ float s_light[2][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = wave / kHalfWaves, w8 = wave % kHalfWaves;
    const int b = blockIdx.x / parts, part = blockIdx.x % parts;
    const int64_t u64 = users[b], i64 = items[b];
    if (u64 < 0 || u64 >= n_user_rows || i64 < 0 || i64 + n_user_rows >= n_rows) return;   // workgroup-uniform: never gather out of range
    const int r = half ? (int)i64 + n_user_rows : (int)u64;
    // everything whose address is known now is requested now: both row ranges, the running sum, the label
    const int beg = rowptr[r], deg = rowptr[r + 1] - beg;
    const int t_beg = t_rowptr[r], t_end = t_rowptr[r + 1];
    const float run = acc_in[(size_t)r * kWave + lane];
    const float y_lab = labels[b];
    const int nseg = (deg + kTaskEntries - 1) / kTaskEntries;
    const float *__restrict__ Xl = X + lane;
    // ---- 1. last layer at this row: segment sums
#pragma unroll
    for (int vi = 0; vi < 2; ++vi) {
        const int v = w8 + vi * kHalfWaves;                       // virtual wave of the row-list kernel
        float acc = 0.0f;
        for (int sgi = v; sgi < nseg; sgi += kWgWaves) {
            const int e0 = beg + sgi * kTaskEntries;
            const int cnt = (deg - sgi * kTaskEntries < kTaskEntries) ? deg - sgi * kTaskEntries : kTaskEntries;
            int my_col = 0;
            float my_val = 0.0f;
            if (lane < cnt) {
                my_col = col[e0 + lane];
                my_val = val[e0 + lane];
            }
            const int last_col = __builtin_amdgcn_readlane(my_col, (cnt - 1) & 63);
            if (lane >= cnt) my_col = last_col;                    // padding: value 0 on a row already being fetched
            for (int c = 0; c * kChunk < cnt; ++c) {
                float x[kChunk];
#pragma unroll
                for (int k = 0; k < kChunk; ++k)
                    x[k] = Xl[(size_t)(uint32_t)__builtin_amdgcn_readlane(my_col, c * kChunk + k) * kWave];
#pragma unroll
                for (int k = 0; k < kChunk; ++k) acc = fmaf(lane_bcast(my_val, c * kChunk + k), x[k], acc);
            }
        }
        if (v < nseg) s_part[half][v][lane] = acc;
    }
    __syncthreads();
    // ---- 2. layer mean at both rows, the score, this row's gradient
    if (w8 == 0) {
        float y = nseg > 0 ? s_part[half][0][lane] : 0.0f;
        const int lim = nseg < kWgWaves ? nseg : kWgWaves;
        for (int w = 1; w < lim; ++w) y = y + s_part[half][w][lane];          // segment order
        float s = run + y;
        if (acc_div != 1.0f) s = s / acc_div;
        s_light[half][lane] = s;
    }
    __syncthreads();
    const float lu = s_light[0][lane], li = s_light[1][lane];
    const float x = wave_sum_f32(fmaf(lu, li, 0.0f));
    const float dg = (sigmoid_f(x) - y_lab) * grad_scale;
    const float g = dg * (half ? lu : li);
    if (part == 0 && w8 == 0) {
        if (half == 0 && lane == 0) atomicAdd(loss_sum, fmaxf(x, 0.0f) - x * y_lab + log1pf(expf(-fabsf(x))));
        atomicAdd(g_out + (size_t)r * kWave + lane, g);                      // dense d loss / d light_out (rows may repeat in a batch)
        atomicAdd(G + (size_t)r * kWave + lane, push_scale * g);             // the `g` of (g + A^T g) / (L + 1)
    }
    // ---- 3. push over the row's entries in A^T: runs of 16 dealt over (part, wave of the half); every run's (col, val) pair
    //         is loaded and lane 0 of each read before the first atomic, the entry loop is a real loop (rows.hip explains why)
    const float gs = push_scale * g;
    const int w_all = part * kHalfWaves + w8;
    const int stride = parts * kHalfWaves * 16;
    constexpr int kPre = 2;
    float *out_l = G + lane;
    for (int base0 = t_beg + w_all * 16; base0 < t_end; base0 += kPre * stride) {
        int my_col[kPre], c0[kPre];
        float my_val[kPre], v0[kPre];
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            const int base = base0 + p * stride;
            my_col[p] = 0;
            my_val[p] = 0.0f;
            if (base + lane < t_end && lane < 16) {
                my_col[p] = t_col[base + lane];
                my_val[p] = t_val[base + lane];
            }
        }
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            c0[p] = __builtin_amdgcn_readlane(my_col[p], 0);
            v0[p] = lane_bcast(my_val[p], 0);
        }
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            const int base = base0 + p * stride;
            const int cnt = t_end - base < 16 ? t_end - base : 16;           // (<= 0 past the row's end)
            if (cnt > 0) atomicAdd(out_l + (size_t)c0[p] * kWave, v0[p] * gs);
#pragma unroll 1
            for (int j = 1; j < cnt; ++j) {
                const int c = __builtin_amdgcn_readlane(my_col[p], j);
                const float v = lane_bcast(my_val[p], j);
                atomicAdd(out_l + (size_t)c * kWave, v * gs);
            }
        }
    }
}

}  // namespace

extern "C" int spex_lightgcn_batch_f32(const spex_graph_t *g, const spex_graph_t *gt, const float *X, const float *acc_in, float acc_div,
                                       const int64_t *users, const int64_t *items, const float *labels, int32_t B,
                                       int32_t n_user_rows, float grad_scale, float push_scale, float *loss_sum, float *g_out, float *G,
                                       int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g && gt && X && acc_in && users && items && labels && loss_sum && g_out && G, "spex_lightgcn_batch_f32: NULL argument");
    SPEX_CHECK_ARG(B >= 0 && n_user_rows >= 0 && n_user_rows <= g->n_rows, "spex_lightgcn_batch_f32: B=%d n_user_rows=%d", B, n_user_rows);
    SPEX_CHECK_ARG(g->n_rows == g->n_cols && gt->n_rows == g->n_rows && gt->n_cols == g->n_rows, "spex_lightgcn_batch_f32: square graphs of one size");
    SPEX_CHECK_ARG(g->mask_mode == 0 && gt->mask_mode == 0, "spex_lightgcn_batch_f32: edge dropout is not supported here");
    if (d != kWave) {
        spex::set_error("spex_lightgcn_batch_f32: d == 64 only (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    if (B == 0 || g->n_rows == 0) return SPEX_OK;
    static const int parts = []() {
        const char *e = getenv("SPEX_BATCH_PARTS");
        const int p = e ? atoi(e) : 4;
        return p < 1 ? 1 : (p > 16 ? 16 : p);
    }();
    hipLaunchKernelGGL(lightgcn_batch_kernel, dim3((unsigned)B * parts), dim3(kWave * kWgWaves), 0, (hipStream_t)stream, g->rowptr, g->col,
                       g->val, gt->rowptr, gt->col, gt->val, g->n_rows, n_user_rows, X, acc_in, acc_div, users, items, labels, parts,
                       grad_scale, push_scale, loss_sum, g_out, G);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
