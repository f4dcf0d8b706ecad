// NGCF layer epilogue and the dual-task expert gate — the small dense pieces that sit on top of the SpMM.
//
// spex_ngcf_layer_f32 replaces NGCF_SPEX/code/main_rec.py:77-83 (one layer, dropout off):
//   s = LeakyReLU(side W_gc^T + b_gc);  b = LeakyReLU((ego * side) W_bi^T + b_bi);  e1 = s + b
//   out = [ego | e1 / max(||e1||_2, 1e-12)]
// spex_expert_gate_f32 replaces LightGCN_SPEX/code/utility1/model_expert_s.py:156-161.
//
// Both are bandwidth/latency-bound per row (2 x 256 B in, 512 B out; 8 kflop per row, 128 MFLOP on Epinion2).  The
// NGCF layer's two [N,64]x[64,64] products run on the matrix cores (f32 MFMA, see the kernel); the gate is a pair of
// 128-long dot products per row and stays on the vector unit.
#include <map>
#include <mutex>
#include <utility>

#include "spex_common.h"

using namespace spex;

namespace {

__device__ __forceinline__ float wave_sum(float v)
{
    return wave_sum_f32(v);
}

__device__ __forceinline__ float bcast(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// The two [N,64]x[64,64] products are the one GEMM-shaped piece of this path, so they run on the matrix cores:
// v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate — a k-ordered fmaf chain, bit-identical to the scalar loop it
// replaces; there is no reduced-precision fp32 path on gfx950 and none is wanted here).  One wave owns a tile of 16
// rows: the 16x64 `side` tile and the 16x64 (ego * side) tile go through LDS (row stride 66 floats: the A-operand read
// pattern row*66 + 4s + h then hits 32 distinct banks per half-wave), the two 64x64 weight matrices are staged once
// per workgroup with the same stride for the B operands, and the four 16x16 output blocks of each product accumulate
// in 2 x 4 x 4 registers.  Bias, LeakyReLU, the add, the row L2-norm (16-lane butterfly) and the concat write follow
// in registers.  8 kflop per row is tiny — the kernel stays bandwidth/latency-bound — but the MFMA form needs 128
// matrix instructions per tile where the vector form needed 2 x 64 x (v_readlane + v_fmac) per ROW.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Message dropout (NGCF_SPEX/code/main_rec.py:81, nn.Dropout(p) on sum + bi): counter-based, so the forward kernel and
// the backward kernel (which recomputes the layer) see the same mask without storing it.  Element e = row * 64 + col of
// the layer's [N, 64] output keeps iff u_e >= p with u_e = (word[e & 3] of philox4x32-10(counter = (e >> 2, step,
// layer, 0), key = seed) >> 8) * 2^-24; kept values are multiplied by 1 / (1 - p) (at::dropout: x * (mask / (1 - p))).
// `pad_row`: rows above it count one less (the model keeps the reference's unused pad user row, main_rec.py:67, as an
// isolated node inside the table; the mask is indexed in the reference's row numbering, which skips it).
struct MsgDrop {
    float p, scale;
    uint32_t k0, k1, step, layer;
    int pad_row;
    const uint8_t *mask;     // validation hook (spex_ngcf_message_mask): keep bytes [rows in the reference's numbering][64], or NULL
};

__device__ __forceinline__ bool msg_keep(const MsgDrop &dr, int row, int col)
{
    const uint32_t e = (uint32_t)(row - (row > dr.pad_row ? 1 : 0)) * 64u + (uint32_t)col;
    if (dr.mask) return dr.mask[e] != 0;
    uint32_t c0 = e >> 2, c1 = dr.step, c2 = dr.layer, c3 = 0u, k0 = dr.k0, k1 = dr.k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const uint32_t sel = e & 3u;
    const uint32_t w = sel == 0u ? c0 : (sel == 1u ? c1 : (sel == 2u ? c2 : c3));
    return (float)(w >> 8) * 5.9604644775390625e-8f >= dr.p;
}

// The same decisions for the four rows a lane holds in the accumulator layout (rows row[0..3], column col = 16 b + i16), with ONE
// Philox evaluation per lane instead of four.  A draw is keyed by (row, col / 4) and yields the four words of columns 4m .. 4m + 3:
// the four lanes of a quad (i16 = 4m .. 4m + 3: same rows, same column group) used to compute the identical ten rounds each, once per
// row.  Here quad lane s draws for row[s] and the quad exchanges the words with DPP broadcasts: lane s takes word s (its own column)
// of the draw made by quad lane q for row[q].  Same mask, bit for bit; the ten rounds are ~40 quarter-rate integer multiplies —
// ~1.3 us per tile and wave saved.  (row[q] < 0 — an absent slot — yields an unspecified value: callers test the row themselves.)
__device__ __forceinline__ void msg_keep4(const MsgDrop &dr, const int (&row)[4], int col, bool (&keep)[4])
{
#ifdef SPEX_NO_QUAD_PHILOX
    const bool every_lane_draws = true;
#else
    const bool every_lane_draws = false;
#endif
    if (dr.mask || every_lane_draws) {
#pragma unroll
        for (int q = 0; q < 4; ++q) keep[q] = row[q] >= 0 && msg_keep(dr, row[q], col);
        return;
    }
    const int s = (int)(threadIdx.x & 3u);                        // == col & 3 (col = 16 b + i16, i16 = lane & 15)
    const int my_row = s == 0 ? row[0] : (s == 1 ? row[1] : (s == 2 ? row[2] : row[3]));
    const uint32_t e = (uint32_t)(my_row - (my_row > dr.pad_row ? 1 : 0)) * 64u + (uint32_t)col;
    uint32_t c0 = e >> 2, c1 = dr.step, c2 = dr.layer, c3 = 0u, k0 = dr.k0, k1 = dr.k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#define SPEX_QUAD_BCAST(v, q) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(v), (q) * 0x55, 0xF, 0xF, true))
#define SPEX_KEEP_FROM(q)                                                                                                       \
    {                                                                                                                           \
        const uint32_t w0 = SPEX_QUAD_BCAST(c0, q), w1 = SPEX_QUAD_BCAST(c1, q), w2 = SPEX_QUAD_BCAST(c2, q), w3 = SPEX_QUAD_BCAST(c3, q); \
        const uint32_t w = s == 0 ? w0 : (s == 1 ? w1 : (s == 2 ? w2 : w3));                                                    \
        keep[q] = (float)(w >> 8) * 5.9604644775390625e-8f >= dr.p;                                                             \
    }
    SPEX_KEEP_FROM(0) SPEX_KEEP_FROM(1) SPEX_KEEP_FROM(2) SPEX_KEEP_FROM(3)
#undef SPEX_KEEP_FROM
#undef SPEX_QUAD_BCAST
}


// The same layer with FOUR waves per 16-row tile (16-wave workgroups = 4 tiles; the weights are staged once per workgroup):
// wave b of a tile owns output columns 16b .. 16b+15 — 2 x 16 MFMAs instead of 2 x 64, a quarter of the Philox draws — and
// the row's sum of squares over all 64 columns goes through LDS (4 partials per row, one workgroup barrier).  Operands are
// dealt k = 16h + j (16 consecutive floats per lane: ds_read_b128), as in the backward kernels.  The one-wave-per-tile form
// above is a single wave's chain per tile: 13.3 us on Epinion2's 975 tiles (with message dropout), this form: see DESIGN.md.
constexpr int kFwdStride = 68;
constexpr int kFwdTiles = 4;                                      // tiles per workgroup

// ROWS (spex_ngcf_layer_fwd_rows_f32): the same layer at a LIST of rows — tile slot k names row idx_a[k] + off_a (k < n_a) or
// idx_b[k - n_a] + off_b; `ego`, `side` and `out` stay the dense tables and are read / written at those rows only (a row named
// twice is computed twice and stored twice with the same value), the dropout mask is indexed by the row, not by the slot.  A
// one-layer model's training loss reads the layer's output at the batch's rows only (main_rec.py:96-104), so the step computes
// it there (2B <= 512 rows instead of N = 15 592 on Epinion2).
struct RowList {
    const int64_t *idx_a, *idx_b;
    int n_a, n_b;
    int64_t off_a, off_b;
};

template <bool ROWS>
__global__ __launch_bounds__(kWave * 4 * kFwdTiles) void ngcf_layer_fwd4_kernel(
    const float *__restrict__ ego, const float *__restrict__ side, const float *__restrict__ W_gc,
    const float *__restrict__ b_gc, const float *__restrict__ W_bi, const float *__restrict__ b_bi,
    float *__restrict__ out, int ld_out, int write_ego, float *__restrict__ e1_out, int n, float slope, const MsgDrop drop,
    const RowList rl)
{
    __shared__ float s_w[2][64 * kFwdStride];                       // W_gc, W_bi as [out j][in k]
    __shared__ float s_t[kFwdTiles][2][16 * kFwdStride];            // per tile: side, ego * side
    __shared__ float s_sq[kFwdTiles][4][16];                        // per tile and column block: partial sums of squares per row
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int tl = wave >> 2, b = wave & 3;
    const int i16 = lane & 15, h = lane >> 4;
    const int r0 = (blockIdx.x * kFwdTiles + tl) << 4;              // (tiles past the end: every row fails r < n)
    float *t_side = s_t[tl][0], *t_prod = s_t[tl][1];
    // the tile's 16 rows: slot k of the tile -> table row (lane k < 16 holds it), -1 = none
    int slot_row = -1;
    if (lane < 16) {
        const int k = r0 + lane;
        if (ROWS) {
            if (k < rl.n_a + rl.n_b) {
                const long long r = batch_row(rl.idx_a, rl.n_a, rl.off_a, rl.idx_b, rl.off_b, k);
                slot_row = (r >= 0 && r < n) ? (int)r : -1;
            }
        } else {
            slot_row = k < n ? k : -1;
        }
    }
    int row_q[4];                                                   // rows of this lane's accumulator entries (row 4h + q of the tile)
#pragma unroll
    for (int q = 0; q < 4; ++q) row_q[q] = __shfl(slot_row, 4 * h + q, kWave);
    // this wave's four rows of the tile (lane == column), requested before the weights
    float e_reg[4], s_reg[4];
    int row_i[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = __builtin_amdgcn_readlane(slot_row, 4 * b + i);
        row_i[i] = r;
        e_reg[i] = s_reg[i] = 0.0f;
        if (r >= 0) {
            e_reg[i] = ego[(size_t)r * 64 + lane];
            s_reg[i] = side[(size_t)r * 64 + lane];
        }
    }
    const float bias_g = b_gc[16 * b + i16], bias_b = b_bi[16 * b + i16];
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {       // 1024 float4 per matrix, coalesced
        const int r = i >> 4, c4 = (i & 15) * 4;
        *reinterpret_cast<float4 *>(&s_w[0][r * kFwdStride + c4]) = *reinterpret_cast<const float4 *>(W_gc + r * 64 + c4);
        *reinterpret_cast<float4 *>(&s_w[1][r * kFwdStride + c4]) = *reinterpret_cast<const float4 *>(W_bi + r * 64 + c4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row_i[i];
        if (write_ego && r >= 0) out[(size_t)r * ld_out + lane] = e_reg[i];     // `ego` passes through to the output's first half
        t_side[(4 * b + i) * kFwdStride + lane] = s_reg[i];
        t_prod[(4 * b + i) * kFwdStride + lane] = e_reg[i] * s_reg[i];
    }
    __syncthreads();
    f32x4 acc_g = (f32x4){0.f, 0.f, 0.f, 0.f}, acc_b = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        float4 ag[4], ab[4], wg[4], wb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ag[j] = *reinterpret_cast<const float4 *>(&t_side[i16 * kFwdStride + 16 * h + 4 * j]);
            ab[j] = *reinterpret_cast<const float4 *>(&t_prod[i16 * kFwdStride + 16 * h + 4 * j]);
            wg[j] = *reinterpret_cast<const float4 *>(&s_w[0][(16 * b + i16) * kFwdStride + 16 * h + 4 * j]);
            wb[j] = *reinterpret_cast<const float4 *>(&s_w[1][(16 * b + i16) * kFwdStride + 16 * h + 4 * j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].x, wg[j].x, acc_g, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].x, wb[j].x, acc_b, 0, 0, 0);
            acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].y, wg[j].y, acc_g, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].y, wb[j].y, acc_b, 0, 0, 0);
            acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].z, wg[j].z, acc_g, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].z, wb[j].z, acc_b, 0, 0, 0);
            acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].w, wg[j].w, acc_g, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].w, wb[j].w, acc_b, 0, 0, 0);
        }
    }
    // C layout: row 4h + q, column 16b + i16
    float e1[4];
    bool keep4[4] = {true, true, true, true};
    if (drop.p > 0.0f) msg_keep4(drop, row_q, 16 * b + i16, keep4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x = acc_g[q] + bias_g, y = acc_b[q] + bias_b;
        x = x >= 0.0f ? x : x * slope;
        y = y >= 0.0f ? y : y * slope;
        float v = x + y;
        if (drop.p > 0.0f) v = (row_q[q] >= 0 && keep4[q]) ? v * drop.scale : 0.0f;
        e1[q] = v;
        const float sq = row16_sum_f32(v * v);
        if (i16 == 0) s_sq[tl][b][4 * h + q] = sq;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = 4 * h + q, r = row_q[q];
        if (r >= 0) {
            const float sq = ((s_sq[tl][0][rr] + s_sq[tl][1][rr]) + s_sq[tl][2][rr]) + s_sq[tl][3][rr];
            const float den = fmaxf(sqrtf(sq), 1e-12f);
            out[(size_t)r * ld_out + 64 + 16 * b + i16] = e1[q] / den;
            if (e1_out) e1_out[(size_t)r * 64 + 16 * b + i16] = e1[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of one NGCF layer (autograd of NGCF_SPEX/code/main_rec.py:77-83 given side = A ego from the SpMM).
// Per row, with s = side W_gc^T + b_gc, t = (ego * side) W_bi^T + b_bi, e1 = LReLU(s) + LReLU(t), e1d = dropout(e1),
// out = e1d / max(|e1d|, 1e-12) and upstream gradients g_norm (w.r.t. out) and g_next (w.r.t. e1d, from the next layer):
//   g_e1d = (g_norm - out <g_norm, out>) / |e1d| + g_next;   g_e1 = g_e1d * keep / (1 - p)
//   G_s = g_e1 * LReLU'(s);  G_t = g_e1 * LReLU'(t)
//   dW_gc += G_s^T side;  db_gc += sum G_s;  dW_bi += G_t^T (ego * side);  db_bi += sum G_t
//   g_side = G_s W_gc + (G_t W_bi) * ego;      g_ego = (G_t W_bi) * side (+ g_direct)      [then g_ego += A^T g_side: SpMM]
// Nothing of the forward is stored besides `side`: the layer is recomputed per 16-row tile on the matrix cores (two
// [16,64]x[64,64] products), the two input-gradient products are two more, and the weight gradients are a fifth and
// sixth MFMA product whose contraction runs over the tile's ROWS: with the rows taken in the order 4h + s (lane group
// h, step s) the A operand is G in the accumulator layout it already has and the B operand is the side / product tile
// read from LDS in that same layout — no transposes.  One WORKGROUP of four waves per tile, every product split by output block.
// Two forms:
//   dense (ngcf_layer_bwd_dense4_kernel)  tiles of 16 consecutive rows; a tile whose upstream gradient is all zero writes zeros
//                         and skips the arithmetic; weight gradients as one partial block per workgroup, added in block order;
//   rows  (ngcf_layer_bwd_rows4_kernel)   tiles of 16 entries of a device row list (the <= 512 distinct rows of a batch — the only
//                         rows with a gradient behind the LAST layer): 32 tiles instead of 975, compact [K, 64] outputs for the
//                         push-form SpMM, weight gradients as one partial block per tile (plain stores, summed in order).
constexpr int kBwdWaves = 4;
constexpr int kBwdStride = 68;     // LDS row stride of the backward's tiles: rows 16-byte aligned for ds_read_b128
constexpr int kPartFloats = 2 * (64 * 64 + 64);    // one block of weight-gradient partials: [dW_gc | db_gc | dW_bi | db_bi]

// Dense form (every row of the table; multi-layer models, whose earlier layers see a dense upstream gradient): one WORKGROUP per
// 16-row tile, the four waves splitting every product by output block exactly as in the rows form below (ngcf_layer_bwd_rows4_kernel:
// same operand dealing, same arithmetic per element) — the first version walked its tiles with ONE wave each, a dependent chain of
// ~400 MFMAs per tile in 252 VGPRs (60.9 us for Epinion2's 975 tiles).  A workgroup is persistent over a strided set of tiles, which
// lets it keep
//   * both weight matrices in REGISTERS instead of LDS: wave b only ever needs rows 16b .. 16b+15 of W (recompute) and columns
//     16b .. 16b+15 of W (input gradients), 16 floats per lane each — 64 VGPRs, loaded once; the 70 KB of LDS the rows form spends on
//     two copies of the weights is gone, so several workgroups share a CU;
//   * its share of the weight gradients (wave b: rows 16b .. 16b+15 of dW_gc, dW_bi) in MFMA accumulators across its tiles, added to
//     its own block of partials once at the end — plain stores; a second small launch adds the blocks in order (float atomics from
//     256+ workgroups onto the same 8 320 addresses cost 10 us per 256 workgroups and would cap the grid at one workgroup per CU).
// A tile whose upstream gradient is all zero (most tiles behind a 256-sample batch) writes zeros / g_direct and skips the arithmetic.
__global__ __launch_bounds__(kWave *kBwdWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) void ngcf_layer_bwd_dense4_kernel(
    const float *__restrict__ ego, const float *__restrict__ side, const float *__restrict__ W_gc,
    const float *__restrict__ b_gc, const float *__restrict__ W_bi, const float *__restrict__ b_bi,
    const float *__restrict__ g_norm, int ld_g, const float *__restrict__ g_next, const float *__restrict__ g_direct,
    int ld_direct, int n, float slope, const MsgDrop drop, float *__restrict__ g_side, float *__restrict__ g_ego,
    float *__restrict__ parts)
{
    __shared__ float s_tile[5][16 * kBwdStride];                // side, ego, ego * side, G_s, G_t
    __shared__ float s_red[2][kBwdWaves][16];                   // per-wave partial row sums: squares, <g_norm, out>
    const int lane = threadIdx.x & (kWave - 1), b = threadIdx.x >> 6;      // wave == output block
    const int i16 = lane & 15, h = lane >> 4;
    const int n_tiles = (n + 15) >> 4;
    float *t_side = s_tile[0], *t_ego = s_tile[1], *t_prod = s_tile[2], *t_gs = s_tile[3], *t_gt = s_tile[4];
    // ---- the weights this wave needs, once: W[o = 16b + i16][k = 16h + 4j ..] (B operand of the recomputation) and
    //      W[o = 16h + 4j ..][c = 16b + i16] (B operand of the input gradients: the transposed matrix, read with a stride)
    float4 wg[4], wb[4], wgT[4], wbT[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wg[j] = *reinterpret_cast<const float4 *>(W_gc + (16 * b + i16) * 64 + 16 * h + 4 * j);
        wb[j] = *reinterpret_cast<const float4 *>(W_bi + (16 * b + i16) * 64 + 16 * h + 4 * j);
        const int o = 16 * h + 4 * j, c = 16 * b + i16;
        wgT[j] = make_float4(W_gc[(o + 0) * 64 + c], W_gc[(o + 1) * 64 + c], W_gc[(o + 2) * 64 + c], W_gc[(o + 3) * 64 + c]);
        wbT[j] = make_float4(W_bi[(o + 0) * 64 + c], W_bi[(o + 1) * 64 + c], W_bi[(o + 2) * 64 + c], W_bi[(o + 3) * 64 + c]);
    }
    const float bias_g = b_gc[16 * b + i16], bias_b = b_bi[16 * b + i16];
    f32x4 dWg[4], dWb[4];                                        // rows 16b + 4h + q, column block bn
#pragma unroll
    for (int bn = 0; bn < 4; ++bn) dWg[bn] = dWb[bn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbg = 0.0f, dbb = 0.0f;
    bool did_work = false;
    // A tile's inputs: the upstream gradients of this wave's 16 columns in the accumulator layout (row 4h + q, column 16b + i16) and the
    // rows 4b .. 4b+3 of ego / side (lane == column).  The NEXT tile's are requested before the current one is computed: a tile is a
    // chain of global round trips and barriers, and a CU holds two such workgroups at most.
    float gn[4], gx[4], gd[4], e_reg[4], s_reg[4];
    auto load_tile = [&](int tl) {
        const int r0 = tl << 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + 4 * h + q;
            gn[q] = gx[q] = gd[q] = 0.0f;
            if (tl < n_tiles && r < n) {
                gn[q] = g_norm[(size_t)r * ld_g + 16 * b + i16];
                if (g_next) gx[q] = g_next[(size_t)r * 64 + 16 * b + i16];
                if (g_direct) gd[q] = g_direct[(size_t)r * ld_direct + 16 * b + i16];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 4 * b + i;
            e_reg[i] = s_reg[i] = 0.0f;
            if (tl < n_tiles && r < n) {
                e_reg[i] = ego[(size_t)r * 64 + lane];
                s_reg[i] = side[(size_t)r * 64 + lane];
            }
        }
    };
    load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int r0 = tile << 4;
        // (looked at only now: testing the values where they are loaded would wait for the prefetch on the spot)
        int any = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) any |= (gn[q] != 0.0f) | (gx[q] != 0.0f);
        // (the barrier also separates the previous tile's last LDS reads from this tile's staging)
        if (!__syncthreads_or(any)) {                            // no gradient reaches this tile
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + 4 * h + q;
                if (r < n) {
                    g_side[(size_t)r * 64 + 16 * b + i16] = 0.0f;
                    g_ego[(size_t)r * 64 + 16 * b + i16] = gd[q];
                }
            }
            load_tile(tile + gridDim.x);
            continue;
        }
        did_work = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t_side[(4 * b + i) * kBwdStride + lane] = s_reg[i];
            t_ego[(4 * b + i) * kBwdStride + lane] = e_reg[i];
            t_prod[(4 * b + i) * kBwdStride + lane] = e_reg[i] * s_reg[i];
        }
        const float gn_c[4] = {gn[0], gn[1], gn[2], gn[3]}, gx_c[4] = {gx[0], gx[1], gx[2], gx[3]}, gd_c[4] = {gd[0], gd[1], gd[2], gd[3]};
        load_tile(tile + gridDim.x);                                                      // in flight during this tile's arithmetic
        __syncthreads();                                                                  // (1) tile staged
        // ---- recompute this block's columns of s and t (contraction index dealt k = 16h + j, see the rows form)
        f32x4 acc_g = (f32x4){0.f, 0.f, 0.f, 0.f}, acc_b = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            float4 ag[4], ab[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ag[j] = *reinterpret_cast<const float4 *>(&t_side[i16 * kBwdStride + 16 * h + 4 * j]);
                ab[j] = *reinterpret_cast<const float4 *>(&t_prod[i16 * kBwdStride + 16 * h + 4 * j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].x, wg[j].x, acc_g, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].x, wb[j].x, acc_b, 0, 0, 0);
                acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].y, wg[j].y, acc_g, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].y, wb[j].y, acc_b, 0, 0, 0);
                acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].z, wg[j].z, acc_g, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].z, wb[j].z, acc_b, 0, 0, 0);
                acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].w, wg[j].w, acc_g, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].w, wb[j].w, acc_b, 0, 0, 0);
            }
        }
        // ---- elementwise chain on this block's columns; row sums over all 64 columns through LDS
        float e1d[4], kscale[4];
        bool keep4[4] = {true, true, true, true};
        if (drop.p > 0.0f) {
            const int rows4[4] = {r0 + 4 * h, r0 + 4 * h + 1, r0 + 4 * h + 2, r0 + 4 * h + 3};
            msg_keep4(drop, rows4, 16 * b + i16, keep4);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + 4 * h + q;
            const float x = acc_g[q] + bias_g, y = acc_b[q] + bias_b;
            float e = (x >= 0.0f ? x : x * slope) + (y >= 0.0f ? y : y * slope);
            float ks = 1.0f;
            if (drop.p > 0.0f) {
                ks = (r < n && keep4[q]) ? drop.scale : 0.0f;
                e = ks != 0.0f ? e * ks : 0.0f;
            }
            kscale[q] = ks;
            e1d[q] = e;
            const float v = row16_sum_f32(e * e);
            if (i16 == 0) s_red[0][b][4 * h + q] = v;
        }
        __syncthreads();                                                                  // (2) squares
        float den[4], nrm[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = 4 * h + q;
            nrm[q] = sqrtf(((s_red[0][0][rr] + s_red[0][1][rr]) + s_red[0][2][rr]) + s_red[0][3][rr]);
            den[q] = fmaxf(nrm[q], 1e-12f);
            const float v = row16_sum_f32(gn_c[q] * (e1d[q] / den[q]));
            if (i16 == 0) s_red[1][b][rr] = v;
        }
        __syncthreads();                                                                  // (3) <g_norm, out>
        float Gs[4], Gt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = 4 * h + q;
            float dot = ((s_red[1][0][rr] + s_red[1][1][rr]) + s_red[1][2][rr]) + s_red[1][3][rr];
            dot = (nrm[q] >= 1e-12f) ? dot : 0.0f;               // below eps the clamp holds the denominator constant
            const float o = e1d[q] / den[q];
            const float ge1d = (gn_c[q] - o * dot) / den[q] + gx_c[q];
            const float ge1 = r0 + rr < n ? ge1d * kscale[q] : 0.0f;
            const float x = acc_g[q] + bias_g, y = acc_b[q] + bias_b;
            Gs[q] = ge1 * (x > 0.0f ? 1.0f : slope);
            Gt[q] = ge1 * (y > 0.0f ? 1.0f : slope);
            t_gs[rr * kBwdStride + 16 * b + i16] = Gs[q];
            t_gt[rr * kBwdStride + 16 * b + i16] = Gt[q];
        }
        // ---- weight gradients, rows 16b .. 16b+15, accumulated over this workgroup's tiles
        dbg += (Gs[0] + Gs[1]) + (Gs[2] + Gs[3]);
        dbb += (Gt[0] + Gt[1]) + (Gt[2] + Gt[3]);
#pragma unroll
        for (int bn = 0; bn < 4; ++bn) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float sc = t_side[(4 * h + st) * kBwdStride + 16 * bn + i16];
                const float pc = t_prod[(4 * h + st) * kBwdStride + 16 * bn + i16];
                dWg[bn] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs[st], sc, dWg[bn], 0, 0, 0);
                dWb[bn] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gt[st], pc, dWb[bn], 0, 0, 0);
            }
        }
        __syncthreads();                                                                  // (4) G tiles complete
        // ---- input gradients, columns 16b .. 16b+15: ts = G_s W_gc, tb = G_t W_bi (contraction over the 64 outputs o = 16h + j)
        f32x4 ts = (f32x4){0.f, 0.f, 0.f, 0.f}, tb = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            float4 as[4], at[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                as[j] = *reinterpret_cast<const float4 *>(&t_gs[i16 * kBwdStride + 16 * h + 4 * j]);
                at[j] = *reinterpret_cast<const float4 *>(&t_gt[i16 * kBwdStride + 16 * h + 4 * j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ts = __builtin_amdgcn_mfma_f32_16x16x4f32(as[j].x, wgT[j].x, ts, 0, 0, 0);
                tb = __builtin_amdgcn_mfma_f32_16x16x4f32(at[j].x, wbT[j].x, tb, 0, 0, 0);
                ts = __builtin_amdgcn_mfma_f32_16x16x4f32(as[j].y, wgT[j].y, ts, 0, 0, 0);
                tb = __builtin_amdgcn_mfma_f32_16x16x4f32(at[j].y, wbT[j].y, tb, 0, 0, 0);
                ts = __builtin_amdgcn_mfma_f32_16x16x4f32(as[j].z, wgT[j].z, ts, 0, 0, 0);
                tb = __builtin_amdgcn_mfma_f32_16x16x4f32(at[j].z, wbT[j].z, tb, 0, 0, 0);
                ts = __builtin_amdgcn_mfma_f32_16x16x4f32(as[j].w, wgT[j].w, ts, 0, 0, 0);
                tb = __builtin_amdgcn_mfma_f32_16x16x4f32(at[j].w, wbT[j].w, tb, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = 4 * h + q, r = r0 + rr, c = 16 * b + i16;
            if (r < n) {
                g_side[(size_t)r * 64 + c] = ts[q] + tb[q] * t_ego[rr * kBwdStride + c];
                g_ego[(size_t)r * 64 + c] = tb[q] * t_side[rr * kBwdStride + c] + gd_c[q];
            }
        }
    }
    // ---- this workgroup's share of the weight gradients: its own block of `parts` ([dW_gc 64x64 | db_gc 64 | dW_bi 64x64 | db_bi 64],
    //      plain stores — zeros from a workgroup none of whose tiles carried a gradient); ngcf_add_parts_kernel adds the blocks in order
    (void)did_work;
    float *dst = parts + (size_t)blockIdx.x * kPartFloats;
#pragma unroll
    for (int bn = 0; bn < 4; ++bn) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a_ = (16 * b + 4 * h + q) * 64 + 16 * bn + i16;
            dst[a_] = dWg[bn][q];
            dst[64 * 64 + 64 + a_] = dWb[bn][q];
        }
    }
    dbg += __shfl_xor(dbg, 16, kWave); dbg += __shfl_xor(dbg, 32, kWave);
    dbb += __shfl_xor(dbb, 16, kWave); dbb += __shfl_xor(dbb, 32, kWave);
    if (h == 0) {
        dst[64 * 64 + 16 * b + i16] = dbg;
        dst[2 * 64 * 64 + 64 + 16 * b + i16] = dbb;
    }
}

// gW += sum over the workgroups' blocks in a FIXED order (no atomics, the same bits every run): a workgroup owns 32 consecutive
// weights, thread (k, g) adds blocks g, g + 8, g + 16, ... (eight loads in flight), the eight partial sums meet in LDS and are added in
// group order.  (One thread per weight walking all 512 blocks was a 64-round-trip chain: 24 us — as long as the backward itself.)
__global__ __launch_bounds__(256) void ngcf_add_parts_kernel(const float *__restrict__ parts, int n_parts, float *gW_gc, float *gb_gc,
                                                             float *gW_bi, float *gb_bi)
{
    __shared__ float s_sum[8][32];
    const int kk = threadIdx.x & 31, g = threadIdx.x >> 5, k = blockIdx.x * 32 + kk;     // kPartFloats is a multiple of 32
    float acc = 0.0f;
    int w = g;
    for (; w + 56 < n_parts; w += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = parts[(size_t)(w + 8 * u) * kPartFloats + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; w < n_parts; w += 8) acc += parts[(size_t)w * kPartFloats + k];
    s_sum[g][kk] = acc;
    __syncthreads();
    if (g == 0) {
        float t = s_sum[0][kk];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += s_sum[u][kk];
        float *dst = k < 4096 ? gW_gc + k : (k < 4160 ? gb_gc + (k - 4096) : (k < 8256 ? gW_bi + (k - 4160) : gb_bi + (k - 8256)));
        *dst += t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Rows form, one WORKGROUP per 16-slot tile: the four waves split every product by output block.  The one-wave-per-tile form
// above is a single wave's dependent chain of ~400 MFMAs and their LDS operand reads per tile (26 us for the 32 tiles of a
// 256-sample batch, on 32 of the chip's 1 024 SIMDs); here wave b owns
//   recompute     columns 16b .. 16b+15 of s and t               (2 x 16 MFMAs, contraction over the 64 inputs)
//   elementwise   the same 16 columns; the two row sums over all 64 columns go through LDS (4 partials per row)
//   dW            rows 16b .. 16b+15 of dW_gc, dW_bi             (2 x 16 MFMAs, contraction over the tile's 16 rows)
//   dX            columns 16b .. 16b+15 of G_s W_gc, G_t W_bi    (2 x 16 MFMAs, contraction over the 64 outputs)
// with four workgroup barriers (tile + weights staged; sum of squares; <g, out>; G tiles).  Same operand dealing as above
// (k = 16h + j, transposed weight copy), same arithmetic per element; the 64-column row sums are associated as 4 x 16.
// SCORE = true folds the scoring step in (single-layer model: the rows backward directly follows it): the upstream gradients
// are not read from per-sample rows but formed here — slot s belongs to sample k = s mod B (n_a == n_b == B), its row `own`
// and the sample's other row `oth` of the concatenated table all_emb [N, 2 x 64] give x = <own, oth> (128 columns),
// dg = (sigmoid(x) - label) * grad_scale, and the slot's gradient row is dg * oth: its first 64 columns are g_direct, its last
// 64 g_norm.  A sample's loss is stored by its user-side slot (loss_rows[k], plain store).  One launch and the gradient rows'
// round trip through memory less than spex_score_bce_slots_f32 + this kernel.
struct ScoreArgs {
    const float *all_emb;     // [N, 128]
    const float *labels;      // [B]
    float *loss_rows;         // [B]
    float grad_scale;
};

// SCORE == 2 (spex_ngcf_fwd_score_bwd_rows_f32, the one-call step): the tile holds BOTH rows of 8 samples — tile row j < 8 is the
// user row of sample 8 tile + j (slot 8 tile + j), row 8 + j its item row (slot B + 8 tile + j) — so the other row of every sample is
// in the same tile and the scores can be formed from the layer output this kernel recomputes anyway: no concatenated table is read,
// the layer's forward at the batch's rows is not a launch of its own.  After the row norms the normalised outputs go to LDS, wave b
// scores samples 2b and 2b + 1 (x = <ego_u, ego_i> + <out_u, out_i>, the order of the 128-column dot), dg comes back through LDS and
// the upstream gradient rows are dg x the partner's [ego | out].  Same arithmetic per element as SCORE == 1 on a table holding the
// same rows.
template <int SCORE>
__global__ __launch_bounds__(kWave *kBwdWaves) void ngcf_layer_bwd_rows4_kernel(
    const float *__restrict__ ego, const float *__restrict__ side, const float *__restrict__ W_gc,
    const float *__restrict__ b_gc, const float *__restrict__ W_bi, const float *__restrict__ b_bi,
    const float *__restrict__ g_norm, int ld_g, const float *__restrict__ g_next, const float *__restrict__ g_direct,
    int ld_direct, int n, float slope, const MsgDrop drop, const int64_t *__restrict__ idx_a, int n_a, int64_t off_a,
    const int64_t *__restrict__ idx_b, int n_b, int64_t off_b, float *__restrict__ g_side, float *__restrict__ g_ego,
    float *__restrict__ partials, int part_stride, const ScoreArgs sc)
{
    __shared__ float s_dg[16];
    __shared__ float s_w[2][64 * kBwdStride];                   // W_gc, W_bi as [out o][in k]
    __shared__ float s_wT[2][64 * kBwdStride];                  // transposed, [in k][out o]
    __shared__ float s_tile[5][16 * kBwdStride];                // side, ego, ego * side, G_s, G_t
    __shared__ float s_red[2][kBwdWaves][16];                   // per-wave partial row sums: squares, <g_norm, out>
    const int lane = threadIdx.x & (kWave - 1), b = threadIdx.x >> 6;      // wave == output block
    const int i16 = lane & 15, h = lane >> 4;
    const int n_items = n_a + n_b, tile = blockIdx.x, r0 = tile << 4;
    float *t_side = s_tile[0], *t_ego = s_tile[1], *t_prod = s_tile[2], *t_gs = s_tile[3], *t_gt = s_tile[4];
    // tile row rr -> slot: consecutive slots, or (SCORE == 2) both rows of samples 8 tile .. 8 tile + 7
    auto slot_of = [&](int rr) { return SCORE == 2 ? ((rr & 8) ? n_a : 0) + 8 * tile + (rr & 7) : r0 + rr; };
    auto slot_ok = [&](int rr) { return SCORE == 2 ? 8 * tile + (rr & 7) < n_a : r0 + rr < n_items; };
    // ---- the tile's rows (lane i < 16: slot_of(i)), requested first; wave b stages rows 4b .. 4b+3 (lane == column)
    int slot_row = -1, oth_row = -1;
    if (lane < 16 && slot_ok(lane)) {
        const long long r = batch_row(idx_a, n_a, off_a, idx_b, off_b, slot_of(lane));
        slot_row = (r >= 0 && r < n) ? (int)r : -1;
        if (SCORE == 1) {       // the sample's other row: slot s < B pairs with slot s + B
            const int s_ = r0 + lane;
            const long long o = batch_row(idx_a, n_a, off_a, idx_b, off_b, s_ < n_a ? s_ + n_a : s_ - n_a);
            oth_row = (o >= 0 && o < n) ? (int)o : -1;                         // a sample with a bad index gets dg = 0: its valid
        }                                                                       // slot is still processed and receives zero rows
    }
    int row_q[4], oth_q[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        row_q[q] = __shfl(slot_row, 4 * h + q, kWave);
        oth_q[q] = SCORE == 1 ? __shfl(oth_row, 4 * h + q, kWave) : -1;
    }
    float e_reg[4], s_reg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = __builtin_amdgcn_readlane(slot_row, 4 * b + i);
        e_reg[i] = s_reg[i] = 0.0f;
        if (r >= 0) {
            e_reg[i] = ego[(size_t)r * 64 + lane];
            s_reg[i] = side[(size_t)r * 64 + lane];
        }
    }
    // upstream gradients of this wave's 16 columns, accumulator layout (row 4h + q, column 16b + i16)
    float gn[4], gx[4], gd[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        gn[q] = gx[q] = gd[q] = 0.0f;
        if (row_q[q] >= 0) {
            if (SCORE == 2) {   // formed after the row norms (below)
            } else if (SCORE == 1) {   // the other row's columns now, times dg after the barrier
                const int o = oth_q[q];
                if (o >= 0) {
                    gn[q] = sc.all_emb[(size_t)o * 128 + 64 + 16 * b + i16];
                    gd[q] = sc.all_emb[(size_t)o * 128 + 16 * b + i16];
                }
            } else {
                gn[q] = g_norm[(size_t)(r0 + 4 * h + q) * ld_g + 16 * b + i16];
                if (g_direct) gd[q] = g_direct[(size_t)(r0 + 4 * h + q) * ld_direct + 16 * b + i16];
            }
            if (g_next) gx[q] = g_next[(size_t)row_q[q] * 64 + 16 * b + i16];
        }
    }
    if (SCORE == 1) {           // wave b scores slots 4b .. 4b+3: x over the 128 columns, lane == column (two halves)
        float own0[4], own1[4], ot0[4], ot1[4], lab[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = __builtin_amdgcn_readlane(slot_row, 4 * b + i), o = __builtin_amdgcn_readlane(oth_row, 4 * b + i);
            own0[i] = own1[i] = ot0[i] = ot1[i] = lab[i] = 0.0f;
            if (r >= 0 && o >= 0) {
                own0[i] = sc.all_emb[(size_t)r * 128 + lane];
                own1[i] = sc.all_emb[(size_t)r * 128 + 64 + lane];
                ot0[i] = sc.all_emb[(size_t)o * 128 + lane];
                ot1[i] = sc.all_emb[(size_t)o * 128 + 64 + lane];
                const int s_ = r0 + 4 * b + i;
                lab[i] = sc.labels[s_ < n_a ? s_ : s_ - n_a];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = __builtin_amdgcn_readlane(slot_row, 4 * b + i), o = __builtin_amdgcn_readlane(oth_row, 4 * b + i);
            const int s_ = r0 + 4 * b + i;
            // the user-side row leads the product, as in the scoring kernel (users[c] * items[c], columns in order)
            const bool user_side = s_ < n_a;
            const float p0 = user_side ? own0[i] : ot0[i], q0 = user_side ? ot0[i] : own0[i];
            const float p1 = user_side ? own1[i] : ot1[i], q1 = user_side ? ot1[i] : own1[i];
            const float x = wave_sum(fmaf(p1, q1, fmaf(p0, q0, 0.0f)));
            float dg = 0.0f;
            if (r >= 0 && o >= 0) {
                dg = (1.0f / (1.0f + expf(-x)) - lab[i]) * sc.grad_scale;
                if (user_side && lane == 0) sc.loss_rows[s_] = fmaxf(x, 0.0f) - x * lab[i] + log1pf(expf(-fabsf(x)));
            } else if (user_side && lane == 0) {
                sc.loss_rows[s_] = 0.0f;
            }
            if (lane == 0) s_dg[4 * b + i] = dg;
        }
    }
    const float bias_g = b_gc[16 * b + i16], bias_b = b_bi[16 * b + i16];
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {
        const int r = i >> 4, c4 = (i & 15) * 4;
        const float4 a = *reinterpret_cast<const float4 *>(W_gc + r * 64 + c4);
        const float4 w = *reinterpret_cast<const float4 *>(W_bi + r * 64 + c4);
        *reinterpret_cast<float4 *>(&s_w[0][r * kBwdStride + c4]) = a;
        *reinterpret_cast<float4 *>(&s_w[1][r * kBwdStride + c4]) = w;
        s_wT[0][(c4 + 0) * kBwdStride + r] = a.x; s_wT[0][(c4 + 1) * kBwdStride + r] = a.y;
        s_wT[0][(c4 + 2) * kBwdStride + r] = a.z; s_wT[0][(c4 + 3) * kBwdStride + r] = a.w;
        s_wT[1][(c4 + 0) * kBwdStride + r] = w.x; s_wT[1][(c4 + 1) * kBwdStride + r] = w.y;
        s_wT[1][(c4 + 2) * kBwdStride + r] = w.z; s_wT[1][(c4 + 3) * kBwdStride + r] = w.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t_side[(4 * b + i) * kBwdStride + lane] = s_reg[i];
        t_ego[(4 * b + i) * kBwdStride + lane] = e_reg[i];
        t_prod[(4 * b + i) * kBwdStride + lane] = e_reg[i] * s_reg[i];
    }
    __syncthreads();                                                                  // (1) tile + weights staged (+ dg)
    if (SCORE == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float dg = s_dg[4 * h + q];
            gn[q] = dg * gn[q];
            gd[q] = dg * gd[q];
        }
    }
    // ---- recompute this block's columns of s and t
    f32x4 acc_g = (f32x4){0.f, 0.f, 0.f, 0.f}, acc_b = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        float4 ag[4], ab[4], wg[4], wb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ag[j] = *reinterpret_cast<const float4 *>(&t_side[i16 * kBwdStride + 16 * h + 4 * j]);
            ab[j] = *reinterpret_cast<const float4 *>(&t_prod[i16 * kBwdStride + 16 * h + 4 * j]);
            wg[j] = *reinterpret_cast<const float4 *>(&s_w[0][(16 * b + i16) * kBwdStride + 16 * h + 4 * j]);
            wb[j] = *reinterpret_cast<const float4 *>(&s_w[1][(16 * b + i16) * kBwdStride + 16 * h + 4 * j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].x, wg[j].x, acc_g, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].x, wb[j].x, acc_b, 0, 0, 0);
            acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].y, wg[j].y, acc_g, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].y, wb[j].y, acc_b, 0, 0, 0);
            acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].z, wg[j].z, acc_g, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].z, wb[j].z, acc_b, 0, 0, 0);
            acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[j].w, wg[j].w, acc_g, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[j].w, wb[j].w, acc_b, 0, 0, 0);
        }
    }
    // ---- elementwise chain on this block's columns; row sums over all 64 columns through LDS
    float e1d[4], kscale[4];
    bool keep4[4] = {true, true, true, true};
    if (drop.p > 0.0f) msg_keep4(drop, row_q, 16 * b + i16, keep4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x = acc_g[q] + bias_g, y = acc_b[q] + bias_b;
        float e = (x >= 0.0f ? x : x * slope) + (y >= 0.0f ? y : y * slope);
        float ks = 1.0f;
        if (drop.p > 0.0f) {
            ks = (row_q[q] >= 0 && keep4[q]) ? drop.scale : 0.0f;
            e = ks != 0.0f ? e * ks : 0.0f;
        }
        kscale[q] = ks;
        e1d[q] = e;
        const float v = row16_sum_f32(e * e);
        if (i16 == 0) s_red[0][b][4 * h + q] = v;
    }
    __syncthreads();                                                                  // (2) squares
    float den[4], nrm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = 4 * h + q;
        nrm[q] = sqrtf(((s_red[0][0][rr] + s_red[0][1][rr]) + s_red[0][2][rr]) + s_red[0][3][rr]);
        den[q] = fmaxf(nrm[q], 1e-12f);
    }
    if (SCORE == 2) {
        // the normalised layer output of the tile's 16 rows -> LDS (the G_s tile's space: written only after barrier (3))
        float *t_o = t_gs;
#pragma unroll
        for (int q = 0; q < 4; ++q) t_o[(4 * h + q) * kBwdStride + 16 * b + i16] = row_q[q] >= 0 ? e1d[q] / den[q] : 0.0f;
        __syncthreads();                                                              // (2b) outputs visible
        // wave b scores samples 2b, 2b + 1 of the tile: lane == column, the user-side row leads the product, ego columns first
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int sj = 2 * b + j, k = 8 * tile + sj;                               // tile-local sample, sample of the batch
            const int ru = __builtin_amdgcn_readlane(slot_row, sj), ri = __builtin_amdgcn_readlane(slot_row, 8 + sj);
            const float eu = t_ego[sj * kBwdStride + lane], ei = t_ego[(8 + sj) * kBwdStride + lane];
            const float ou = t_o[sj * kBwdStride + lane], oi = t_o[(8 + sj) * kBwdStride + lane];
            const float x = wave_sum(fmaf(ou, oi, fmaf(eu, ei, 0.0f)));
            float dg = 0.0f;
            if (k < n_a) {
                const float lab = sc.labels[k];
                if (ru >= 0 && ri >= 0) {
                    dg = (1.0f / (1.0f + expf(-x)) - lab) * sc.grad_scale;
                    if (lane == 0) sc.loss_rows[k] = fmaxf(x, 0.0f) - x * lab + log1pf(expf(-fabsf(x)));
                } else if (lane == 0) {
                    sc.loss_rows[k] = 0.0f;
                }
            }
            if (lane == 0) s_dg[sj] = s_dg[8 + sj] = dg;
        }
        __syncthreads();                                                              // (2c) dg
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = 4 * h + q, pr = rr ^ 8;                                     // the sample's other row in this tile
            const float dg = s_dg[rr];
            gn[q] = row_q[q] >= 0 ? dg * t_o[pr * kBwdStride + 16 * b + i16] : 0.0f;
            gd[q] = row_q[q] >= 0 ? dg * t_ego[pr * kBwdStride + 16 * b + i16] : 0.0f;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = 4 * h + q;
        const float v = row16_sum_f32(gn[q] * (e1d[q] / den[q]));
        if (i16 == 0) s_red[1][b][rr] = v;
    }
    __syncthreads();                                                                  // (3) <g_norm, out>
    float Gs[4], Gt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = 4 * h + q;
        float dot = ((s_red[1][0][rr] + s_red[1][1][rr]) + s_red[1][2][rr]) + s_red[1][3][rr];
        dot = (nrm[q] >= 1e-12f) ? dot : 0.0f;                   // below eps the clamp holds the denominator constant
        const float o = e1d[q] / den[q];
        const float ge1d = (gn[q] - o * dot) / den[q] + gx[q];
        const float ge1 = row_q[q] >= 0 ? ge1d * kscale[q] : 0.0f;
        const float x = acc_g[q] + bias_g, y = acc_b[q] + bias_b;
        Gs[q] = ge1 * (x > 0.0f ? 1.0f : slope);
        Gt[q] = ge1 * (y > 0.0f ? 1.0f : slope);
        t_gs[rr * kBwdStride + 16 * b + i16] = Gs[q];
        t_gt[rr * kBwdStride + 16 * b + i16] = Gt[q];
    }
    // ---- weight gradients, rows 16b .. 16b+15: dW[o][k] = sum_r G[r][o] X[r][k], row 4h + st at step st of lane group h
    float *dst = partials + (size_t)tile * part_stride;
    {
        float cs = (Gs[0] + Gs[1]) + (Gs[2] + Gs[3]), ct = (Gt[0] + Gt[1]) + (Gt[2] + Gt[3]);
        cs += __shfl_xor(cs, 16, kWave); cs += __shfl_xor(cs, 32, kWave);
        ct += __shfl_xor(ct, 16, kWave); ct += __shfl_xor(ct, 32, kWave);
        if (h == 0) {
            dst[64 * 64 + 16 * b + i16] = cs;
            dst[2 * 64 * 64 + 64 + 16 * b + i16] = ct;
        }
    }
#pragma unroll
    for (int bn = 0; bn < 4; ++bn) {
        f32x4 dWg = (f32x4){0.f, 0.f, 0.f, 0.f}, dWb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float sc = t_side[(4 * h + st) * kBwdStride + 16 * bn + i16];
            const float pc = t_prod[(4 * h + st) * kBwdStride + 16 * bn + i16];
            dWg = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs[st], sc, dWg, 0, 0, 0);
            dWb = __builtin_amdgcn_mfma_f32_16x16x4f32(Gt[st], pc, dWb, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a_ = (16 * b + 4 * h + q) * 64 + 16 * bn + i16;
            dst[a_] = dWg[q];
            dst[64 * 64 + 64 + a_] = dWb[q];
        }
    }
    __syncthreads();                                                                  // (4) G tiles complete
    // ---- input gradients, columns 16b .. 16b+15: ts = G_s W_gc, tb = G_t W_bi (contraction over the 64 outputs o = 16h + j)
    f32x4 ts = (f32x4){0.f, 0.f, 0.f, 0.f}, tb = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        float4 as[4], at[4], wg[4], wb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            as[j] = *reinterpret_cast<const float4 *>(&t_gs[i16 * kBwdStride + 16 * h + 4 * j]);
            at[j] = *reinterpret_cast<const float4 *>(&t_gt[i16 * kBwdStride + 16 * h + 4 * j]);
            wg[j] = *reinterpret_cast<const float4 *>(&s_wT[0][(16 * b + i16) * kBwdStride + 16 * h + 4 * j]);
            wb[j] = *reinterpret_cast<const float4 *>(&s_wT[1][(16 * b + i16) * kBwdStride + 16 * h + 4 * j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ts = __builtin_amdgcn_mfma_f32_16x16x4f32(as[j].x, wg[j].x, ts, 0, 0, 0);
            tb = __builtin_amdgcn_mfma_f32_16x16x4f32(at[j].x, wb[j].x, tb, 0, 0, 0);
            ts = __builtin_amdgcn_mfma_f32_16x16x4f32(as[j].y, wg[j].y, ts, 0, 0, 0);
            tb = __builtin_amdgcn_mfma_f32_16x16x4f32(at[j].y, wb[j].y, tb, 0, 0, 0);
            ts = __builtin_amdgcn_mfma_f32_16x16x4f32(as[j].z, wg[j].z, ts, 0, 0, 0);
            tb = __builtin_amdgcn_mfma_f32_16x16x4f32(at[j].z, wb[j].z, tb, 0, 0, 0);
            ts = __builtin_amdgcn_mfma_f32_16x16x4f32(as[j].w, wg[j].w, ts, 0, 0, 0);
            tb = __builtin_amdgcn_mfma_f32_16x16x4f32(at[j].w, wb[j].w, tb, 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (row_q[q] >= 0) {
            const int rr = 4 * h + q, c = 16 * b + i16;
            const size_t o = (size_t)slot_of(rr) * 64 + c;                    // compact slot
            g_side[o] = ts[q] + tb[q] * t_ego[rr * kBwdStride + c];
            g_ego[o] = tb[q] * t_side[rr * kBwdStride + c] + gd[q];
        }
    }
}


__global__ __launch_bounds__(kWave *kWavesPerBlock) void expert_gate_kernel(const float *__restrict__ raw,
                                                                           const float *__restrict__ prop,
                                                                           const float *__restrict__ att,
                                                                           float *__restrict__ mixed, int n, int d)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_global = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * kWavesPerBlock;
    for (int r = wave_global; r < n; r += n_waves) {
        float z0 = 0.0f, z1 = 0.0f;
        for (int c = lane; c < d; c += kWave) {
            const float a = raw[(size_t)r * d + c], b = prop[(size_t)r * d + c];
            z0 = fmaf(a, att[2 * c], z0);
            z1 = fmaf(a, att[2 * c + 1], z1);
            z0 = fmaf(b, att[2 * (d + c)], z0);
            z1 = fmaf(b, att[2 * (d + c) + 1], z1);
        }
        z0 = wave_sum(z0);
        z1 = wave_sum(z1);
        const float mx = fmaxf(z0, z1);
        const float e0 = expf(z0 - mx), e1 = expf(z1 - mx);
        const float a0 = e0 / (e0 + e1), a1 = e1 / (e0 + e1);
        for (int c = lane; c < d; c += kWave)
            mixed[(size_t)r * d + c] = raw[(size_t)r * d + c] * a0 + prop[(size_t)r * d + c] * a1;
    }
}

// Backward of the gate.  Per row (one wave): with z = [raw | prop] W, a = softmax(z), mixed = a0 raw + a1 prop and
// g = dL/dmixed:  da_k = g . src_k;  dz_k = a_k (da_k - (a0 da0 + a1 da1));
//   d raw = a0 g + dz0 W[c,0] + dz1 W[c,1];  d prop = a1 g + dz0 W[d+c,0] + dz1 W[d+c,1];  dW[c,k] += raw[c] dz_k, dW[d+c,k] += prop[c] dz_k.
// The parameter gradient ([2d, 2]) is reduced per workgroup in LDS, then added with atomics (zero it first).
constexpr int kGateWaves = 16;

__global__ __launch_bounds__(kWave *kGateWaves) void expert_gate_bwd_kernel(const float *__restrict__ raw,
                                                                           const float *__restrict__ prop,
                                                                           const float *__restrict__ att,
                                                                           const float *__restrict__ g,
                                                                           float *__restrict__ d_raw, float *__restrict__ d_prop,
                                                                           float *d_att, int n, int d, int per_wave_slots,
                                                                           float *__restrict__ att_parts)
{
    extern __shared__ float s_datt[];   // [2d, 2] (+ one such block per wave when per_wave_slots)
    const int lane = threadIdx.x & (kWave - 1);
    for (int k = threadIdx.x; k < 4 * d; k += blockDim.x) s_datt[k] = 0.0f;
    __syncthreads();
    const int wave_global = blockIdx.x * kGateWaves + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * kGateWaves;
    // a lane owns its columns of the parameter gradient: accumulated in registers over the wave's rows (d <= 256) and added
    // to the workgroup's LDS copy once per wave — per-row LDS float atomics (~200 cycles per wave instruction) made this
    // kernel 4x slower than the data it streams
    constexpr int kCols = 4;
    const bool in_regs = d <= kCols * kWave;
    float acc[kCols][4] = {};
    for (int r = wave_global; r < n; r += n_waves) {
        float z0 = 0.0f, z1 = 0.0f, da0 = 0.0f, da1 = 0.0f;
        for (int c = lane; c < d; c += kWave) {
            const float a = raw[(size_t)r * d + c], b = prop[(size_t)r * d + c], gg = g[(size_t)r * d + c];
            z0 = fmaf(a, att[2 * c], z0);
            z1 = fmaf(a, att[2 * c + 1], z1);
            z0 = fmaf(b, att[2 * (d + c)], z0);
            z1 = fmaf(b, att[2 * (d + c) + 1], z1);
            da0 = fmaf(gg, a, da0);
            da1 = fmaf(gg, b, da1);
        }
        z0 = wave_sum(z0); z1 = wave_sum(z1); da0 = wave_sum(da0); da1 = wave_sum(da1);
        const float mx = fmaxf(z0, z1);
        const float e0 = expf(z0 - mx), e1 = expf(z1 - mx);
        const float a0 = e0 / (e0 + e1), a1 = e1 / (e0 + e1);
        const float dot = a0 * da0 + a1 * da1;
        const float dz0 = a0 * (da0 - dot), dz1 = a1 * (da1 - dot);
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < kCols; ++k) {
                const int c = lane + k * kWave;
                if (c < d) {
                    const float a = raw[(size_t)r * d + c], b = prop[(size_t)r * d + c], gg = g[(size_t)r * d + c];
                    d_raw[(size_t)r * d + c] = a0 * gg + dz0 * att[2 * c] + dz1 * att[2 * c + 1];
                    d_prop[(size_t)r * d + c] = a1 * gg + dz0 * att[2 * (d + c)] + dz1 * att[2 * (d + c) + 1];
                    acc[k][0] = fmaf(a, dz0, acc[k][0]);
                    acc[k][1] = fmaf(a, dz1, acc[k][1]);
                    acc[k][2] = fmaf(b, dz0, acc[k][2]);
                    acc[k][3] = fmaf(b, dz1, acc[k][3]);
                }
            }
        } else {
            for (int c = lane; c < d; c += kWave) {
                const float a = raw[(size_t)r * d + c], b = prop[(size_t)r * d + c], gg = g[(size_t)r * d + c];
                d_raw[(size_t)r * d + c] = a0 * gg + dz0 * att[2 * c] + dz1 * att[2 * c + 1];
                d_prop[(size_t)r * d + c] = a1 * gg + dz0 * att[2 * (d + c)] + dz1 * att[2 * (d + c) + 1];
                atomicAdd(&s_datt[2 * c], a * dz0);
                atomicAdd(&s_datt[2 * c + 1], a * dz1);
                atomicAdd(&s_datt[2 * (d + c)], b * dz0);
                atomicAdd(&s_datt[2 * (d + c) + 1], b * dz1);
            }
        }
    }
    if (in_regs && per_wave_slots) {
        // every wave leaves its sums in its own LDS slot (slots follow the shared copy), added up after the barrier: LDS float
        // atomics from 16 waves onto the same words serialise in the LDS unit (~200 cycles per wave instruction)
        float *mine = s_datt + (size_t)(1 + (threadIdx.x >> 6)) * 4 * d;
#pragma unroll
        for (int k = 0; k < kCols; ++k) {
            const int c = lane + k * kWave;
            if (c < d) {
                mine[2 * c] = acc[k][0];
                mine[2 * c + 1] = acc[k][1];
                mine[2 * (d + c)] = acc[k][2];
                mine[2 * (d + c) + 1] = acc[k][3];
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < 4 * d; k += blockDim.x) {
            float sum = 0.0f;
            for (int v = 0; v < kGateWaves; ++v) sum += s_datt[(size_t)(1 + v) * 4 * d + k];
            s_datt[k] = sum;
        }
    } else if (in_regs) {
#pragma unroll
        for (int k = 0; k < kCols; ++k) {
            const int c = lane + k * kWave;
            if (c < d) {
                atomicAdd(&s_datt[2 * c], acc[k][0]);
                atomicAdd(&s_datt[2 * c + 1], acc[k][1]);
                atomicAdd(&s_datt[2 * (d + c)], acc[k][2]);
                atomicAdd(&s_datt[2 * (d + c) + 1], acc[k][3]);
            }
        }
    }
    __syncthreads();
    // att_parts (deterministic form): the workgroup's share leaves as its own block, added in block order by the caller
    for (int k = threadIdx.x; k < 4 * d; k += blockDim.x) {
        if (att_parts) att_parts[(size_t)blockIdx.x * 4 * d + k] = s_datt[k];
        else atomicAdd(d_att + k, s_datt[k]);
    }
}

// The gate at a batch's rows only — the training loss reads the gated tables nowhere else (model_expert_s.py:163-166), so
// the dual-task step gates <= 2B rows instead of all N.  Slot k names row r = idx_a[k] + off_a (k < n_a) or
// idx_b[k - n_a] + off_b; rows below n_user_rows use att_u, the others att_i.  d == 64 (lane == column); same arithmetic
// order as expert_gate_kernel.
__global__ __launch_bounds__(kWave *kWavesPerBlock) void expert_gate_rows_kernel(
    const float *__restrict__ raw, const float *__restrict__ prop, const float *__restrict__ att_u, const float *__restrict__ att_i,
    const int64_t *__restrict__ idx_a, int n_a, int64_t off_a, const int64_t *__restrict__ idx_b, int n_b, int64_t off_b,
    int64_t n_user_rows, int64_t n_rows, float *__restrict__ mixed_c)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int slot = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (slot >= n_a + n_b) return;
    const long long r = batch_row(idx_a, n_a, off_a, idx_b, off_b, slot);
    if (r < 0 || r >= n_rows) {                      // never gather out of bounds; the scoring kernel skips nothing here
        mixed_c[(size_t)slot * 64 + lane] = 0.0f;
        return;
    }
    const float *att = r < n_user_rows ? att_u : att_i;
    const float a = raw[(size_t)r * 64 + lane], b = prop[(size_t)r * 64 + lane];
    float z0 = 0.0f, z1 = 0.0f;
    z0 = fmaf(a, att[2 * lane], z0);
    z1 = fmaf(a, att[2 * lane + 1], z1);
    z0 = fmaf(b, att[2 * (64 + lane)], z0);
    z1 = fmaf(b, att[2 * (64 + lane) + 1], z1);
    z0 = wave_sum(z0);
    z1 = wave_sum(z1);
    const float mx = fmaxf(z0, z1);
    const float e0 = expf(z0 - mx), e1 = expf(z1 - mx);
    const float a0 = e0 / (e0 + e1), a1 = e1 / (e0 + e1);
    mixed_c[(size_t)slot * 64 + lane] = a * a0 + b * a1;
}

// Its backward, slot by slot (the gate's Jacobian is linear in the incoming gradient, so a row named by several slots is
// simply handled once per slot): g_slots[k] = d loss / d mixed_c[k] ->
//   d_prop_c[k] (compact, the push-form A^T product reads it), and with atomics g_prop[r] += d prop, g_raw[r] += d raw (dense
//   tables, zeroed by the caller), g_att_u / g_att_i += the two gate matrices' gradients (register accumulators per wave).
// DET (spex_expert_gate_rows_bwd_det_f32, the deterministic step): no atomics at all — d raw leaves as a compact per-slot row
// too (d_raw_c; both are summed per table row in slot order by spex_reduce_slots_f32) and the workgroup's share of the two gate
// gradients goes to its own block of att_parts ([gridDim.x][512], summed in block order afterwards).
template <bool DET>
__global__ __launch_bounds__(kWave *kGateWaves) void expert_gate_rows_bwd_kernel(
    const float *__restrict__ raw, const float *__restrict__ prop, const float *__restrict__ att_u, const float *__restrict__ att_i,
    const int64_t *__restrict__ idx_a, int n_a, int64_t off_a, const int64_t *__restrict__ idx_b, int n_b, int64_t off_b,
    int64_t n_user_rows, int64_t n_rows, const float *__restrict__ g_slots, int ld_g, float *__restrict__ d_prop_c, float *g_prop,
    float *g_raw, float *g_att_u, float *g_att_i, float *__restrict__ d_raw_c, float *__restrict__ att_parts)
{
    __shared__ float s_att[kGateWaves][2][256];    // one slot per wave, summed after the barrier (LDS float atomics from 16 waves
                                                   // onto the same 512 words serialise in the LDS unit: 12 us for 128 instructions)
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    float acc[2][4] = {};
    const int n = n_a + n_b;
    for (int slot = blockIdx.x * kGateWaves + (threadIdx.x >> 6); slot < n; slot += gridDim.x * kGateWaves) {
        const long long r = batch_row(idx_a, n_a, off_a, idx_b, off_b, slot);
        if (r < 0 || r >= n_rows) {
            d_prop_c[(size_t)slot * 64 + lane] = 0.0f;
            if (DET) d_raw_c[(size_t)slot * 64 + lane] = 0.0f;
            continue;
        }
        const int which = r < n_user_rows ? 0 : 1;
        const float *att = which ? att_i : att_u;
        const float a = raw[(size_t)r * 64 + lane], b = prop[(size_t)r * 64 + lane], gg = g_slots[(size_t)slot * ld_g + lane];
        const float w00 = att[2 * lane], w01 = att[2 * lane + 1], w10 = att[2 * (64 + lane)], w11 = att[2 * (64 + lane) + 1];
        float z0 = fmaf(b, w10, fmaf(a, w00, 0.0f)), z1 = fmaf(b, w11, fmaf(a, w01, 0.0f));
        z0 = wave_sum(z0); z1 = wave_sum(z1);
        const float da0 = wave_sum(gg * a), da1 = wave_sum(gg * b);
        const float mx = fmaxf(z0, z1);
        const float e0 = expf(z0 - mx), e1 = expf(z1 - mx);
        const float a0 = e0 / (e0 + e1), a1 = e1 / (e0 + e1);
        const float dot = a0 * da0 + a1 * da1;
        const float dz0 = a0 * (da0 - dot), dz1 = a1 * (da1 - dot);
        const float d_raw = a0 * gg + dz0 * w00 + dz1 * w01, d_prop = a1 * gg + dz0 * w10 + dz1 * w11;
        d_prop_c[(size_t)slot * 64 + lane] = d_prop;
        if (DET) {
            d_raw_c[(size_t)slot * 64 + lane] = d_raw;
        } else {
            atomicAdd(g_prop + (size_t)r * 64 + lane, d_prop);
            atomicAdd(g_raw + (size_t)r * 64 + lane, d_raw);
        }
        acc[which][0] = fmaf(a, dz0, acc[which][0]);
        acc[which][1] = fmaf(a, dz1, acc[which][1]);
        acc[which][2] = fmaf(b, dz0, acc[which][2]);
        acc[which][3] = fmaf(b, dz1, acc[which][3]);
    }
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        s_att[wave][w][2 * lane] = acc[w][0];
        s_att[wave][w][2 * lane + 1] = acc[w][1];
        s_att[wave][w][2 * (64 + lane)] = acc[w][2];
        s_att[wave][w][2 * (64 + lane) + 1] = acc[w][3];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 512; k += blockDim.x) {
        const int w = k >> 8, j = k & 255;
        float sum = 0.0f;
        for (int v = 0; v < kGateWaves; ++v) sum += s_att[v][w][j];
        if (DET) att_parts[(size_t)blockIdx.x * 512 + k] = sum;
        else if (sum != 0.0f) atomicAdd((w ? g_att_i : g_att_u) + j, sum);
    }
}

inline unsigned grid_for_rows(int n)
{
    int64_t blocks = ((int64_t)n + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 4) blocks = 256 * 4;
    return (unsigned)blocks;
}

}  // namespace

extern "C" int32_t spex_ngcf_layer_bwd_rows_parts(int32_t n_slots) { return (n_slots + 15) / 16; }   // one block per 16-slot tile

// Validation hook: while a mask is set (per host thread), every NGCF layer entry takes its message-dropout keep decisions from it
// instead of the counter-based draw — e.g. the reference's own nn.Dropout noise, `torch.empty(N, 64).bernoulli_(1 - p)` drawn on
// the CPU from the global generator where main_rec.py:81 draws it, uploaded as bytes.  The pointer is read when a launch is queued.
static thread_local const uint8_t *t_msg_mask = nullptr;
extern "C" int spex_ngcf_message_mask(const uint8_t *d_keep)
{
    t_msg_mask = d_keep;
    return SPEX_OK;
}

// Scratch of the dense layer backward (the workgroups' weight-gradient blocks), one buffer per (device, stream): launches on one stream
// are ordered, two streams never share a buffer.  Grown on demand (hipMalloc: the first call at a size must not sit inside a stream
// capture), kept for the life of the process.
static int stream_scratch(hipStream_t stream, size_t bytes, float **out)
{
    struct Buf { float *p = nullptr; size_t cap = 0; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Buf> bufs;
    int dev = 0;
    SPEX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    Buf &b = bufs[std::make_pair(dev, stream)];
    if (b.cap < bytes) {
        if (b.p) {
            SPEX_HIP(hipStreamSynchronize(stream));
            SPEX_HIP(hipFree(b.p));
            b.p = nullptr; b.cap = 0;
        }
        SPEX_HIP(hipMalloc((void **)&b.p, bytes));
        b.cap = bytes;
    }
    *out = b.p;
    return SPEX_OK;
}

static MsgDrop make_drop(float p_drop, uint64_t seed, uint32_t step, uint32_t layer, int32_t pad_row)
{
    MsgDrop d;
    d.mask = p_drop > 0.0f ? t_msg_mask : nullptr;
    d.p = p_drop > 0.0f ? p_drop : 0.0f;
    d.scale = 1.0f / (float)(1.0 - (double)d.p);          // at::dropout divides its Bernoulli noise by (1 - p) once
    d.k0 = (uint32_t)seed;
    d.k1 = (uint32_t)(seed >> 32);
    d.step = step;
    d.layer = layer;
    d.pad_row = pad_row < 0 ? 0x7fffffff : pad_row;
    return d;
}

extern "C" int spex_ngcf_layer_fwd_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc,
                                       const float *W_bi, const float *b_bi, float *out, int32_t ld_out, int32_t write_ego,
                                       float *e1_out, int32_t n, int32_t d, float slope, float p_drop, uint64_t seed,
                                       uint32_t step, uint32_t layer, int32_t pad_row, void *stream)
{
    SPEX_CHECK_ARG(ego && side && W_gc && b_gc && W_bi && b_bi && out, "spex_ngcf_layer_fwd_f32: NULL pointer");
    SPEX_CHECK_ARG(n >= 0 && ld_out >= 2 * d, "spex_ngcf_layer_fwd_f32: n=%d ld_out=%d", n, ld_out);
    SPEX_CHECK_ARG(p_drop >= 0.0f && p_drop < 1.0f, "spex_ngcf_layer_fwd_f32: p_drop=%f", (double)p_drop);
    if (d != 64) {
        spex::set_error("spex_ngcf_layer_fwd_f32: only d == 64 is implemented (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG((((uintptr_t)W_gc | (uintptr_t)W_bi) & 15) == 0, "spex_ngcf_layer_fwd_f32: weights must be 16-byte aligned");
    if (n == 0) return SPEX_OK;
    const int n_tiles = (n + 15) / 16;
    // four waves per tile, four tiles per workgroup
    hipLaunchKernelGGL(ngcf_layer_fwd4_kernel<false>, dim3((unsigned)((n_tiles + kFwdTiles - 1) / kFwdTiles)), dim3(kWave * 4 * kFwdTiles), 0,
                       (hipStream_t)stream, ego, side, W_gc, b_gc, W_bi, b_bi, out, ld_out, write_ego, e1_out, n, slope,
                       make_drop(p_drop, seed, step, layer, pad_row), RowList{});
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_ngcf_layer_fwd_rows_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc,
                                            const float *W_bi, const float *b_bi, float *out, int32_t ld_out, int32_t write_ego,
                                            int32_t n, int32_t d, float slope, float p_drop, uint64_t seed, uint32_t step,
                                            uint32_t layer, int32_t pad_row, const int64_t *idx_a, int32_t n_a, int64_t off_a,
                                            const int64_t *idx_b, int32_t n_b, int64_t off_b, void *stream)
{
    SPEX_CHECK_ARG(ego && side && W_gc && b_gc && W_bi && b_bi && out, "spex_ngcf_layer_fwd_rows_f32: NULL pointer");
    SPEX_CHECK_ARG(n >= 0 && ld_out >= 2 * d, "spex_ngcf_layer_fwd_rows_f32: n=%d ld_out=%d", n, ld_out);
    SPEX_CHECK_ARG(n_a >= 0 && n_b >= 0 && (n_a == 0 || idx_a) && (n_b == 0 || idx_b), "spex_ngcf_layer_fwd_rows_f32: row lists");
    SPEX_CHECK_ARG(p_drop >= 0.0f && p_drop < 1.0f, "spex_ngcf_layer_fwd_rows_f32: p_drop=%f", (double)p_drop);
    if (d != 64) {
        spex::set_error("spex_ngcf_layer_fwd_rows_f32: only d == 64 is implemented (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG((((uintptr_t)W_gc | (uintptr_t)W_bi) & 15) == 0, "spex_ngcf_layer_fwd_rows_f32: weights must be 16-byte aligned");
    const int n_slots = n_a + n_b;
    if (n == 0 || n_slots == 0) return SPEX_OK;
    const int n_tiles = (n_slots + 15) / 16;
    hipLaunchKernelGGL(ngcf_layer_fwd4_kernel<true>, dim3((unsigned)((n_tiles + kFwdTiles - 1) / kFwdTiles)), dim3(kWave * 4 * kFwdTiles), 0,
                       (hipStream_t)stream, ego, side, W_gc, b_gc, W_bi, b_bi, out, ld_out, write_ego, nullptr, n, slope,
                       make_drop(p_drop, seed, step, layer, pad_row), RowList{idx_a, idx_b, n_a, n_b, off_a, off_b});
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_ngcf_layer_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc,
                                   const float *W_bi, const float *b_bi, float *out, int32_t ld_out, float *e1_out,
                                   int32_t n, int32_t d, float slope, void *stream)
{
    return spex_ngcf_layer_fwd_f32(ego, side, W_gc, b_gc, W_bi, b_bi, out, ld_out, 1, e1_out, n, d, slope, 0.0f, 0, 0, 0, -1,
                                   stream);
}

extern "C" int spex_ngcf_layer_bwd_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc,
                                       const float *W_bi, const float *b_bi, const float *g_norm, int32_t ld_g,
                                       const float *g_next, const float *g_direct, int32_t ld_direct, int32_t n, int32_t d,
                                       float slope, float p_drop, uint64_t seed, uint32_t step, uint32_t layer,
                                       int32_t pad_row, float *g_side, float *g_ego, float *gW_gc, float *gb_gc,
                                       float *gW_bi, float *gb_bi, void *stream)
{
    SPEX_CHECK_ARG(ego && side && W_gc && b_gc && W_bi && b_bi && g_norm && g_side && g_ego && gW_gc && gb_gc && gW_bi && gb_bi,
                   "spex_ngcf_layer_bwd_f32: NULL pointer");
    SPEX_CHECK_ARG(n >= 0 && ld_g >= d && (!g_direct || ld_direct >= d), "spex_ngcf_layer_bwd_f32: n=%d ld_g=%d ld_direct=%d", n,
                   ld_g, ld_direct);
    SPEX_CHECK_ARG(p_drop >= 0.0f && p_drop < 1.0f, "spex_ngcf_layer_bwd_f32: p_drop=%f", (double)p_drop);
    SPEX_CHECK_ARG(g_side != g_ego && g_side != side && g_ego != ego, "spex_ngcf_layer_bwd_f32: outputs must not alias inputs");
    if (d != 64) {
        spex::set_error("spex_ngcf_layer_bwd_f32: only d == 64 is implemented (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG((((uintptr_t)W_gc | (uintptr_t)W_bi) & 15) == 0, "spex_ngcf_layer_bwd_f32: weights must be 16-byte aligned");
    if (n == 0) return SPEX_OK;
    const int n_tiles = (n + 15) / 16;
    // persistent workgroups (two fit a CU): each walks its tiles with a stride and leaves ONE block of weight-gradient partials in the
    // stream's scratch; the second launch adds the blocks to the caller's gradients in block order (deterministic, no atomics)
    const int blocks = n_tiles < 512 ? n_tiles : 512;      // (measured on Epinion2's 975 tiles: 256 workgroups 27.9 us, 512: 24.3, 975: slower)
    float *parts = nullptr;
    if (int rc = stream_scratch((hipStream_t)stream, (size_t)blocks * kPartFloats * sizeof(float), &parts)) return rc;
    hipLaunchKernelGGL(ngcf_layer_bwd_dense4_kernel, dim3((unsigned)blocks), dim3(kWave * kBwdWaves), 0, (hipStream_t)stream, ego, side, W_gc,
                       b_gc, W_bi, b_bi, g_norm, ld_g, g_next, g_direct, ld_direct, n, slope, make_drop(p_drop, seed, step, layer, pad_row),
                       g_side, g_ego, parts);
    hipLaunchKernelGGL(ngcf_add_parts_kernel, dim3(kPartFloats / 32), dim3(256), 0, (hipStream_t)stream, parts, blocks, gW_gc, gb_gc,
                       gW_bi, gb_bi);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_ngcf_layer_bwd_rows_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc,
                                            const float *W_bi, const float *b_bi, const float *g_norm_c, int32_t ld_g,
                                            const float *g_next, const float *g_direct_c, int32_t ld_direct, int32_t n, int32_t d,
                                            float slope, float p_drop, uint64_t seed, uint32_t step, uint32_t layer,
                                            int32_t pad_row, const int64_t *idx_a, int32_t n_a, int64_t off_a,
                                            const int64_t *idx_b, int32_t n_b, int64_t off_b, float *g_side_c, float *g_ego_c,
                                            float *gW_parts, int32_t part_stride, void *stream)
{
    SPEX_CHECK_ARG(ego && side && W_gc && b_gc && W_bi && b_bi && g_norm_c && g_side_c && g_ego_c && gW_parts,
                   "spex_ngcf_layer_bwd_rows_f32: NULL pointer");
    SPEX_CHECK_ARG(n_a >= 0 && n_b >= 0 && (n_a == 0 || idx_a) && (n_b == 0 || idx_b), "spex_ngcf_layer_bwd_rows_f32: bad index lists");
    SPEX_CHECK_ARG(n >= 0 && ld_g >= d && (!g_direct_c || ld_direct >= d) && part_stride >= 2 * (d * d + d),
                   "spex_ngcf_layer_bwd_rows_f32: n=%d ld_g=%d ld_direct=%d part_stride=%d", n, ld_g, ld_direct, part_stride);
    SPEX_CHECK_ARG(p_drop >= 0.0f && p_drop < 1.0f, "spex_ngcf_layer_bwd_rows_f32: p_drop=%f", (double)p_drop);
    if (d != 64) {
        spex::set_error("spex_ngcf_layer_bwd_rows_f32: only d == 64 is implemented (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG((((uintptr_t)W_gc | (uintptr_t)W_bi) & 15) == 0, "spex_ngcf_layer_bwd_rows_f32: weights must be 16-byte aligned");
    if (n == 0 || n_a + n_b == 0) return SPEX_OK;
    const int tiles = spex_ngcf_layer_bwd_rows_parts(n_a + n_b);
    hipLaunchKernelGGL((ngcf_layer_bwd_rows4_kernel<0>), dim3((unsigned)tiles), dim3(kWave * kBwdWaves), 0, (hipStream_t)stream,
                       ego, side, W_gc, b_gc, W_bi, b_bi, g_norm_c, ld_g, g_next, g_direct_c, ld_direct, n, slope,
                       make_drop(p_drop, seed, step, layer, pad_row), idx_a, n_a, off_a, idx_b, n_b, off_b, g_side_c, g_ego_c,
                       gW_parts, part_stride, ScoreArgs{nullptr, nullptr, nullptr, 0.0f});
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_ngcf_score_bwd_rows_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc, const float *W_bi,
                                            const float *b_bi, const float *all_emb, const float *labels, float grad_scale, int32_t n,
                                            int32_t d, float slope, float p_drop, uint64_t seed, uint32_t step, uint32_t layer,
                                            int32_t pad_row, const int64_t *users, const int64_t *items, int32_t B,
                                            int64_t n_user_rows, float *loss_per_sample, float *g_side_c, float *g_ego_c,
                                            float *gW_parts, int32_t part_stride, void *stream)
{
    SPEX_CHECK_ARG(ego && side && W_gc && b_gc && W_bi && b_bi && all_emb && labels && users && items && loss_per_sample && g_side_c
                       && g_ego_c && gW_parts,
                   "spex_ngcf_score_bwd_rows_f32: NULL pointer");
    SPEX_CHECK_ARG(n >= 0 && B >= 0 && n_user_rows >= 0 && n_user_rows <= n && part_stride >= 2 * (d * d + d),
                   "spex_ngcf_score_bwd_rows_f32: n=%d B=%d n_user_rows=%lld part_stride=%d", n, B, (long long)n_user_rows, part_stride);
    SPEX_CHECK_ARG(p_drop >= 0.0f && p_drop < 1.0f, "spex_ngcf_score_bwd_rows_f32: p_drop=%f", (double)p_drop);
    if (d != 64) {
        spex::set_error("spex_ngcf_score_bwd_rows_f32: only d == 64 is implemented (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG((((uintptr_t)W_gc | (uintptr_t)W_bi) & 15) == 0, "spex_ngcf_score_bwd_rows_f32: weights must be 16-byte aligned");
    if (n == 0 || B == 0) return SPEX_OK;
    const int tiles = spex_ngcf_layer_bwd_rows_parts(2 * B);
    hipLaunchKernelGGL((ngcf_layer_bwd_rows4_kernel<1>), dim3((unsigned)tiles), dim3(kWave * kBwdWaves), 0, (hipStream_t)stream, ego,
                       side, W_gc, b_gc, W_bi, b_bi, nullptr, 0, nullptr, nullptr, 0, n, slope,
                       make_drop(p_drop, seed, step, layer, pad_row), users, B, 0, items, B, n_user_rows, g_side_c, g_ego_c, gW_parts,
                       part_stride, ScoreArgs{all_emb, labels, loss_per_sample, grad_scale});
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_ngcf_fwd_score_bwd_rows_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc,
                                               const float *W_bi, const float *b_bi, const float *labels, float grad_scale, int32_t n,
                                               int32_t d, float slope, float p_drop, uint64_t seed, uint32_t step, uint32_t layer,
                                               int32_t pad_row, const int64_t *users, const int64_t *items, int32_t B,
                                               int64_t n_user_rows, float *loss_per_sample, float *g_side_c, float *g_ego_c,
                                               float *gW_parts, int32_t part_stride, void *stream)
{
    SPEX_CHECK_ARG(ego && side && W_gc && b_gc && W_bi && b_bi && labels && users && items && loss_per_sample && g_side_c && g_ego_c
                       && gW_parts,
                   "spex_ngcf_fwd_score_bwd_rows_f32: NULL pointer");
    SPEX_CHECK_ARG(n >= 0 && B >= 0 && n_user_rows >= 0 && n_user_rows <= n && part_stride >= 2 * (d * d + d),
                   "spex_ngcf_fwd_score_bwd_rows_f32: n=%d B=%d n_user_rows=%lld part_stride=%d", n, B, (long long)n_user_rows, part_stride);
    SPEX_CHECK_ARG(p_drop >= 0.0f && p_drop < 1.0f, "spex_ngcf_fwd_score_bwd_rows_f32: p_drop=%f", (double)p_drop);
    if (d != 64) {
        spex::set_error("spex_ngcf_fwd_score_bwd_rows_f32: only d == 64 is implemented (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG((((uintptr_t)W_gc | (uintptr_t)W_bi) & 15) == 0, "spex_ngcf_fwd_score_bwd_rows_f32: weights must be 16-byte aligned");
    if (n == 0 || B == 0) return SPEX_OK;
    const int tiles = (B + 7) / 8;                       // == spex_ngcf_layer_bwd_rows_parts(2 B): one block of weight partials per tile
    hipLaunchKernelGGL((ngcf_layer_bwd_rows4_kernel<2>), dim3((unsigned)tiles), dim3(kWave * kBwdWaves), 0, (hipStream_t)stream, ego,
                       side, W_gc, b_gc, W_bi, b_bi, nullptr, 0, nullptr, nullptr, 0, n, slope,
                       make_drop(p_drop, seed, step, layer, pad_row), users, B, 0, items, B, n_user_rows, g_side_c, g_ego_c, gW_parts,
                       part_stride, ScoreArgs{nullptr, labels, loss_per_sample, grad_scale});
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_expert_gate_f32(const float *raw, const float *prop, const float *att_exp, float *mixed, int32_t n,
                                    int32_t d, void *stream)
{
    SPEX_CHECK_ARG(raw && prop && att_exp && mixed, "spex_expert_gate_f32: NULL pointer");
    SPEX_CHECK_ARG(n >= 0 && d >= 1, "spex_expert_gate_f32: n=%d d=%d", n, d);
    if (n == 0) return SPEX_OK;
    hipLaunchKernelGGL(expert_gate_kernel, dim3(grid_for_rows(n)), dim3(kWave * kWavesPerBlock), 0, (hipStream_t)stream, raw,
                       prop, att_exp, mixed, n, d);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_expert_gate_bwd_f32(const float *raw, const float *prop, const float *att_exp, const float *grad_mixed,
                                        float *grad_raw, float *grad_prop, float *grad_att, int32_t n, int32_t d, void *stream)
{
    SPEX_CHECK_ARG(raw && prop && att_exp && grad_mixed && grad_raw && grad_prop && grad_att, "spex_expert_gate_bwd_f32: NULL pointer");
    SPEX_CHECK_ARG(n >= 0 && d >= 1 && (size_t)d * 16 <= 64 * 1024, "spex_expert_gate_bwd_f32: n=%d d=%d", n, d);
    if (n == 0) return SPEX_OK;
    int64_t blocks = ((int64_t)n + kGateWaves - 1) / kGateWaves;
    if (blocks > 256) blocks = 256;                         // one workgroup per CU: 256 x 4d parameter-gradient atomics
    const int slots = d <= 128 ? 1 : 0;     // (1 + 16) x 4d floats of LDS: 34 KB at d = 128
    hipLaunchKernelGGL(expert_gate_bwd_kernel, dim3((unsigned)blocks), dim3(kWave * kGateWaves),
                       (size_t)d * 4 * sizeof(float) * (slots ? 1 + kGateWaves : 1), (hipStream_t)stream, raw, prop, att_exp, grad_mixed,
                       grad_raw, grad_prop, grad_att, n, d, slots, nullptr);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int32_t spex_expert_gate_bwd_parts(int32_t n)
{
    const int64_t blocks = ((int64_t)n + kGateWaves - 1) / kGateWaves;
    return (int32_t)(blocks > 256 ? 256 : (blocks < 1 ? 1 : blocks));
}

extern "C" int spex_expert_gate_bwd_det_f32(const float *raw, const float *prop, const float *att_exp, const float *grad_mixed,
                                            float *grad_raw, float *grad_prop, float *grad_att, float *att_parts, int32_t n, int32_t d,
                                            void *stream)
{
    SPEX_CHECK_ARG(raw && prop && att_exp && grad_mixed && grad_raw && grad_prop && grad_att && att_parts,
                   "spex_expert_gate_bwd_det_f32: NULL pointer");
    SPEX_CHECK_ARG(n >= 0 && d >= 1 && d <= 128, "spex_expert_gate_bwd_det_f32: n=%d d=%d (d <= 128: the per-wave LDS slots)", n, d);
    if (n == 0) return SPEX_OK;
    const int32_t blocks = spex_expert_gate_bwd_parts(n);
    hipLaunchKernelGGL(expert_gate_bwd_kernel, dim3((unsigned)blocks), dim3(kWave * kGateWaves),
                       (size_t)d * 4 * sizeof(float) * (1 + kGateWaves), (hipStream_t)stream, raw, prop, att_exp, grad_mixed, grad_raw, grad_prop,
                       nullptr, n, d, 1, att_parts);
    SPEX_HIP(hipGetLastError());
    return spex::sum_parts(att_parts, blocks, (int64_t)4 * d, 4 * d, grad_att, 1, stream);
}

extern "C" int spex_expert_gate_rows_f32(const float *raw, const float *prop, const float *att_u, const float *att_i,
                                         const int64_t *idx_a, int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b,
                                         int64_t off_b, int64_t n_user_rows, int64_t n_rows, int32_t d, float *mixed_c, void *stream)
{
    SPEX_CHECK_ARG(raw && prop && att_u && att_i && mixed_c && (idx_a || n_a == 0) && (idx_b || n_b == 0),
                   "spex_expert_gate_rows_f32: NULL pointer");
    SPEX_CHECK_ARG(n_a >= 0 && n_b >= 0 && n_rows >= 0, "spex_expert_gate_rows_f32: negative size");
    if (d != 64) {
        spex::set_error("spex_expert_gate_rows_f32: d = %d (the row form needs d == 64)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    const int n = n_a + n_b;
    if (n == 0) return SPEX_OK;
    hipLaunchKernelGGL(expert_gate_rows_kernel, dim3((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kWave * kWavesPerBlock), 0,
                       (hipStream_t)stream, raw, prop, att_u, att_i, idx_a, n_a, off_a, idx_b, n_b, off_b, n_user_rows, n_rows, mixed_c);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_expert_gate_rows_bwd_f32(const float *raw, const float *prop, const float *att_u, const float *att_i,
                                             const int64_t *idx_a, int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b,
                                             int64_t off_b, int64_t n_user_rows, int64_t n_rows, int32_t d, const float *grad_slots,
                                             int32_t ld_slots, float *grad_prop_slots, float *grad_prop, float *grad_raw,
                                             float *grad_att_u, float *grad_att_i, void *stream)
{
    SPEX_CHECK_ARG(raw && prop && att_u && att_i && grad_slots && grad_prop_slots && grad_prop && grad_raw && grad_att_u && grad_att_i
                       && (idx_a || n_a == 0) && (idx_b || n_b == 0),
                   "spex_expert_gate_rows_bwd_f32: NULL pointer");
    SPEX_CHECK_ARG(n_a >= 0 && n_b >= 0 && n_rows >= 0 && ld_slots >= 64, "spex_expert_gate_rows_bwd_f32: bad size");
    if (d != 64) {
        spex::set_error("spex_expert_gate_rows_bwd_f32: d = %d (the row form needs d == 64)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    const int n = n_a + n_b;
    if (n == 0) return SPEX_OK;
    int blocks = (n + kGateWaves - 1) / kGateWaves;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(expert_gate_rows_bwd_kernel<false>, dim3((unsigned)blocks), dim3(kWave * kGateWaves), 0, (hipStream_t)stream, raw, prop,
                       att_u, att_i, idx_a, n_a, off_a, idx_b, n_b, off_b, n_user_rows, n_rows, grad_slots, ld_slots, grad_prop_slots,
                       grad_prop, grad_raw, grad_att_u, grad_att_i, nullptr, nullptr);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int32_t spex_expert_gate_rows_bwd_parts(int32_t n_slots)
{
    const int blocks = (n_slots + kGateWaves - 1) / kGateWaves;
    return blocks > 64 ? 64 : (blocks < 1 ? 1 : blocks);
}

extern "C" int spex_expert_gate_rows_bwd_det_f32(const float *raw, const float *prop, const float *att_u, const float *att_i,
                                                 const int64_t *idx_a, int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b,
                                                 int64_t off_b, int64_t n_user_rows, int64_t n_rows, int32_t d, const float *grad_slots,
                                                 int32_t ld_slots, float *grad_prop_slots, float *grad_raw_slots, float *att_parts,
                                                 void *stream)
{
    SPEX_CHECK_ARG(raw && prop && att_u && att_i && grad_slots && grad_prop_slots && grad_raw_slots && att_parts
                       && (idx_a || n_a == 0) && (idx_b || n_b == 0),
                   "spex_expert_gate_rows_bwd_det_f32: NULL pointer");
    SPEX_CHECK_ARG(n_a >= 0 && n_b >= 0 && n_rows >= 0 && ld_slots >= 64, "spex_expert_gate_rows_bwd_det_f32: bad size");
    if (d != 64) {
        spex::set_error("spex_expert_gate_rows_bwd_det_f32: d = %d (the row form needs d == 64)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    const int n = n_a + n_b;
    if (n == 0) return SPEX_OK;
    hipLaunchKernelGGL(expert_gate_rows_bwd_kernel<true>, dim3((unsigned)spex_expert_gate_rows_bwd_parts(n)), dim3(kWave * kGateWaves), 0,
                       (hipStream_t)stream, raw, prop, att_u, att_i, idx_a, n_a, off_a, idx_b, n_b, off_b, n_user_rows, n_rows, grad_slots,
                       ld_slots, grad_prop_slots, nullptr, nullptr, nullptr, nullptr, grad_raw_slots, att_parts);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
