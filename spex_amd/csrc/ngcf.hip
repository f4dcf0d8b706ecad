// NGCF layer epilogue and the dual-task expert gate — the small dense pieces that sit on top of the SpMM.
//
// spex_ngcf_layer_f32 replaces NGCF_SPEX/code/main_rec.py:77-83 (one layer, dropout off):
//   s = LeakyReLU(side W_gc^T + b_gc);  b = LeakyReLU((ego * side) W_bi^T + b_bi);  e1 = s + b
//   out = [ego | e1 / max(||e1||_2, 1e-12)]
// spex_expert_gate_f32 replaces LightGCN_SPEX/code/utility1/model_expert_s.py:156-161.
//
// Both are bandwidth-bound per row (2 x 256 B in, 512 B out; the two 64x64 weight matrices are 32 KB and live in
// registers: lane j keeps row j of each, and the row's 64 inputs are broadcast lane-by-lane with v_readlane).  The
// [N,64]x[64,64] products are 8 kflop per row — 128 MFLOP on Epinion2 — far below what would justify staging tiles
// for MFMA; one wave per row keeps the whole layer a single streaming pass fused behind the SpMM output.
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

__global__ __launch_bounds__(kWave *kWavesPerBlock) void ngcf_layer_kernel(
    const float *__restrict__ ego, const float *__restrict__ side, const float *__restrict__ W_gc,
    const float *__restrict__ b_gc, const float *__restrict__ W_bi, const float *__restrict__ b_bi,
    float *__restrict__ out, int ld_out, float *__restrict__ e1_out, int n, float slope)
{
    const int j = threadIdx.x & (kWave - 1);
    const int wave_global = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * kWavesPerBlock;
    // Lane j keeps row j of both weight matrices in registers.  Reading those rows straight from global memory is a
    // 256-byte-strided access (64 cache lines per wave instruction, 32 KB per wave): stage the two matrices through
    // LDS once per workgroup instead — coalesced float4 loads in, padded rows (stride 65) out, conflict-free.
    __shared__ float s_w[2][64 * 65];
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {       // 1024 float4 per matrix
        const int r = i >> 4, c4 = (i & 15) * 4;
        const float4 a = *reinterpret_cast<const float4 *>(W_gc + r * 64 + c4);
        const float4 b = *reinterpret_cast<const float4 *>(W_bi + r * 64 + c4);
        float *pa = &s_w[0][r * 65 + c4], *pb = &s_w[1][r * 65 + c4];
        pa[0] = a.x; pa[1] = a.y; pa[2] = a.z; pa[3] = a.w;
        pb[0] = b.x; pb[1] = b.y; pb[2] = b.z; pb[3] = b.w;
    }
    __syncthreads();
    float wg[64], wb[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        wg[k] = s_w[0][j * 65 + k];
        wb[k] = s_w[1][j * 65 + k];
    }
    const float bg = b_gc[j], bb = b_bi[j];
    for (int r = wave_global; r < n; r += n_waves) {
        const float e = ego[(size_t)r * 64 + j], s = side[(size_t)r * 64 + j];
        const float t = e * s;
        float a1 = 0.0f, a2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            a1 = fmaf(bcast(s, k), wg[k], a1);
            a2 = fmaf(bcast(t, k), wb[k], a2);
        }
        a1 += bg;
        a2 += bb;
        a1 = a1 >= 0.0f ? a1 : a1 * slope;
        a2 = a2 >= 0.0f ? a2 : a2 * slope;
        const float e1 = a1 + a2;
        const float nrm = sqrtf(wave_sum(e1 * e1));
        out[(size_t)r * ld_out + j] = e;
        out[(size_t)r * ld_out + 64 + j] = e1 / fmaxf(nrm, 1e-12f);
        if (e1_out) e1_out[(size_t)r * 64 + j] = e1;
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
    hipLaunchKernelGGL(ngcf_layer_kernel, dim3(grid_for_rows(n)), dim3(kWave * kWavesPerBlock), 0, (hipStream_t)stream,
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
