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
#include "spex_common.h"

using namespace spex;

namespace {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
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
constexpr int kLdsStride = 66;

__global__ __launch_bounds__(kWave *kWavesPerBlock) void ngcf_layer_kernel(
    const float *__restrict__ ego, const float *__restrict__ side, const float *__restrict__ W_gc,
    const float *__restrict__ b_gc, const float *__restrict__ W_bi, const float *__restrict__ b_bi,
    float *__restrict__ out, int ld_out, float *__restrict__ e1_out, int n, float slope)
{
    __shared__ float s_w[2][64 * kLdsStride];                       // W_gc, W_bi as [out j][in k]
    __shared__ float s_t[kWavesPerBlock][2][16 * kLdsStride];       // per wave: side tile, (ego*side) tile
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int n_tiles = (n + 15) >> 4;
    int tile = blockIdx.x * kWavesPerBlock + wave;
    // The wave's first tile is requested BEFORE the weights are staged (32 independent row loads in flight while the
    // workgroup fills s_w and waits at the barrier): the kernel is a chain of dependent memory round trips — with the
    // tile fetched after the barrier, in four batches of four rows, it took 15.4 us on Epinion2.
    float e_reg[16], s_reg[16];
    auto fetch_tile = [&](int tl) {
        const int r0 = tl << 4;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + i;
            e_reg[i] = s_reg[i] = 0.0f;
            if (tl < n_tiles && r < n) {
                e_reg[i] = ego[(size_t)r * 64 + lane];
                s_reg[i] = side[(size_t)r * 64 + lane];
            }
        }
    };
    fetch_tile(tile);
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {       // 1024 float4 per matrix, coalesced
        const int r = i >> 4, c4 = (i & 15) * 4;
        const float4 a = *reinterpret_cast<const float4 *>(W_gc + r * 64 + c4);
        const float4 b = *reinterpret_cast<const float4 *>(W_bi + r * 64 + c4);
        float *pa = &s_w[0][r * kLdsStride + c4], *pb = &s_w[1][r * kLdsStride + c4];
        pa[0] = a.x; pa[1] = a.y; pa[2] = a.z; pa[3] = a.w;
        pb[0] = b.x; pb[1] = b.y; pb[2] = b.z; pb[3] = b.w;
    }
    __syncthreads();
    const int i16 = lane & 15, h = lane >> 4;                       // MFMA operand coordinates of this lane
    float bias_g[4], bias_b[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        bias_g[b] = b_gc[16 * b + i16];
        bias_b[b] = b_bi[16 * b + i16];
    }
    float *t_side = s_t[wave][0], *t_prod = s_t[wave][1];
    for (; tile < n_tiles; tile += gridDim.x * kWavesPerBlock) {
        const int r0 = tile << 4;
        // stage the tile (lane == column: coalesced 256-byte rows) and pass `ego` through to the output's first half
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + i;
            if (r < n) out[(size_t)r * ld_out + lane] = e_reg[i];
            t_side[i * kLdsStride + lane] = s_reg[i];
            t_prod[i * kLdsStride + lane] = e_reg[i] * s_reg[i];
        }
        fetch_tile(tile + gridDim.x * kWavesPerBlock);              // the next tile of this wave, if any, overlaps the MFMAs
        // (each wave only reads back its own tile: no workgroup barrier, the LDS ops of one wave are ordered)
        f32x4 acc_g[4], acc_b[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            acc_g[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc_b[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int k = 4 * s + h;
            const float a_g = t_side[i16 * kLdsStride + k];
            const float a_b = t_prod[i16 * kLdsStride + k];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float w_g = s_w[0][(16 * b + i16) * kLdsStride + k];
                const float w_b = s_w[1][(16 * b + i16) * kLdsStride + k];
                acc_g[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_g, w_g, acc_g[b], 0, 0, 0);
                acc_b[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_b, w_b, acc_b[b], 0, 0, 0);
            }
        }
        // C layout: col = 16 b + i16, row = 4 h + reg
        float e1[4][4], sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x = acc_g[b][q] + bias_g[b], y = acc_b[b][q] + bias_b[b];
                x = x >= 0.0f ? x : x * slope;
                y = y >= 0.0f ? y : y * slope;
                e1[b][q] = x + y;
            }
        }
        // row sums of squares in column order (0..63), as the scalar kernel's butterfly produced them up to rounding:
        // first the lane's own 4 column blocks, then the 16 lanes of the row group
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = 0.0f;
#pragma unroll
            for (int b = 0; b < 4; ++b) v = fmaf(e1[b][q], e1[b][q], v);
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
            sq[q] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + 4 * h + q;
            if (r < n) {
                const float den = fmaxf(sqrtf(sq[q]), 1e-12f);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    out[(size_t)r * ld_out + 64 + 16 * b + i16] = e1[b][q] / den;
                    if (e1_out) e1_out[(size_t)r * 64 + 16 * b + i16] = e1[b][q];
                }
            }
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
                                                                           float *d_att, int n, int d)
{
    extern __shared__ float s_datt[];   // [2d, 2]
    const int lane = threadIdx.x & (kWave - 1);
    for (int k = threadIdx.x; k < 4 * d; k += blockDim.x) s_datt[k] = 0.0f;
    __syncthreads();
    const int wave_global = blockIdx.x * kGateWaves + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * kGateWaves;
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
    __syncthreads();
    for (int k = threadIdx.x; k < 4 * d; k += blockDim.x) atomicAdd(d_att + k, s_datt[k]);
}

inline unsigned grid_for_rows(int n)
{
    int64_t blocks = ((int64_t)n + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 4) blocks = 256 * 4;
    return (unsigned)blocks;
}

}  // namespace

extern "C" int spex_ngcf_layer_f32(const float *ego, const float *side, const float *W_gc, const float *b_gc,
                                   const float *W_bi, const float *b_bi, float *out, int32_t ld_out, float *e1_out,
                                   int32_t n, int32_t d, float slope, void *stream)
{
    SPEX_CHECK_ARG(ego && side && W_gc && b_gc && W_bi && b_bi && out, "spex_ngcf_layer_f32: NULL pointer");
    SPEX_CHECK_ARG(n >= 0 && ld_out >= 2 * d, "spex_ngcf_layer_f32: n=%d ld_out=%d", n, ld_out);
    if (d != 64) {
        spex::set_error("spex_ngcf_layer_f32: only d == 64 is implemented (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG((((uintptr_t)W_gc | (uintptr_t)W_bi) & 15) == 0, "spex_ngcf_layer_f32: weights must be 16-byte aligned");
    if (n == 0) return SPEX_OK;
    const int n_tiles = (n + 15) / 16;  // one wave per 16-row tile
    hipLaunchKernelGGL(ngcf_layer_kernel, dim3(grid_for_rows(n_tiles)), dim3(kWave * kWavesPerBlock), 0, (hipStream_t)stream,
                       ego, side, W_gc, b_gc, W_bi, b_bi, out, ld_out, e1_out, n, slope);
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
    hipLaunchKernelGGL(expert_gate_bwd_kernel, dim3((unsigned)blocks), dim3(kWave * kGateWaves), (size_t)d * 4 * sizeof(float),
                       (hipStream_t)stream, raw, prop, att_exp, grad_mixed, grad_raw, grad_prop, grad_att, n, d);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
