// Node-level attention fusion of the Diffnet++ diffusion layers (the literal "social trust graph"
// propagation of the reference): after the three SpMMs of a layer, every user mixes what came from its consumed items
// and from its social neighbours, every item mixes itself with what came from its customers —
// Diffnet++_SPEX/code/utility/Model.py:308-345 (first layer) and :352-385 (second layer):
//
//   users:  e_k = exp(LReLU_0.2(w2_k tanh([U | X_k] . w1_k + b1_k) + b2_k)) + c_k      k = items (c = 0.7), social (c = 0.3)
//           out = 1/2 U + 1/2 (e_1 X_1 + e_2 X_2) / (e_1 + e_2)
//   items:  e_k = exp(LReLU_0.2(w2_k tanh(X_k . w1_k + b1_k) + b2_k)) + 1              X_1 = the item itself, X_2 = customers
//           out = (e_1 X_1 + e_2 X_2) / (e_1 + e_2)
//
// In TensorFlow (and in a tensor-op restatement) that is ~40 small launches per layer and direction; here one wave per
// row does it in one launch (lane == column; two to four dot products, a handful of scalars), and the backward kernel
// recomputes the scalars, writes the three row gradients and reduces the 2 x (|w1| + 3) parameter gradients per
// workgroup in LDS before adding them with atomics.  Rows are independent: bandwidth / latency-bound streaming.
#include "spex_common.h"

using namespace spex;

namespace {

constexpr int kFuseWaves = 16;   // 1024-thread workgroups (few parameter-gradient atomics per address)
constexpr int kMaxCols = 4;      // d <= 256

__device__ __forceinline__ float wave_sum(float v)
{
    return wave_sum_f32(v);
}

// parameter block of one branch: w1[(U ? d : 0) + d], b1, w2, b2
struct FuseArgs {
    const float *U;            // [n, d] or NULL
    const float *X1, *X2;      // [n, d]
    const float *p1, *p2;      // branch parameter blocks
    int n, d;
    float c1, c2, base_coef, mix_coef;
};

struct BranchScalars {
    float t, q, e;             // tanh output, pre-LeakyReLU value, exp(l) + c
};

__device__ __forceinline__ BranchScalars branch_forward(float r, const float *p, int n_w1, float c)
{
    BranchScalars s;
    s.t = tanhf(r + p[n_w1]);
    s.q = p[n_w1 + 1] * s.t + p[n_w1 + 2];
    const float l = s.q > 0.0f ? s.q : 0.2f * s.q;
    s.e = expf(l) + c;
    return s;
}

__global__ __launch_bounds__(kWave *kFuseWaves) void attn_fuse_kernel(const FuseArgs a, float *__restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int n_w1 = (a.U ? a.d : 0) + a.d, off_x = a.U ? a.d : 0;
    for (int r = blockIdx.x * kFuseWaves + (threadIdx.x >> 6); r < a.n; r += gridDim.x * kFuseWaves) {
        float u[kMaxCols], x1[kMaxCols], x2[kMaxCols];
        float r1 = 0.0f, r2 = 0.0f;
#pragma unroll
        for (int k = 0; k < kMaxCols; ++k) {
            const int c = lane + k * kWave;
            u[k] = x1[k] = x2[k] = 0.0f;
            if (c < a.d) {
                x1[k] = a.X1[(size_t)r * a.d + c];
                x2[k] = a.X2[(size_t)r * a.d + c];
                r1 = fmaf(x1[k], a.p1[off_x + c], r1);
                r2 = fmaf(x2[k], a.p2[off_x + c], r2);
                if (a.U) {
                    u[k] = a.U[(size_t)r * a.d + c];
                    r1 = fmaf(u[k], a.p1[c], r1);
                    r2 = fmaf(u[k], a.p2[c], r2);
                }
            }
        }
        const BranchScalars s1 = branch_forward(wave_sum(r1), a.p1, n_w1, a.c1);
        const BranchScalars s2 = branch_forward(wave_sum(r2), a.p2, n_w1, a.c2);
        const float tot = s1.e + s2.e, a1 = s1.e / tot, a2 = s2.e / tot;
#pragma unroll
        for (int k = 0; k < kMaxCols; ++k) {
            const int c = lane + k * kWave;
            if (c < a.d) out[(size_t)r * a.d + c] = a.base_coef * u[k] + a.mix_coef * (a1 * x1[k] + a2 * x2[k]);
        }
    }
}

__global__ __launch_bounds__(kWave *kFuseWaves) void attn_fuse_bwd_kernel(const FuseArgs a, const float *__restrict__ g,
                                                                         float *__restrict__ gU, float *__restrict__ gX1,
                                                                         float *__restrict__ gX2, float *gp1, float *gp2)
{
    extern __shared__ float s_gp[];   // [2][n_w1 + 3]
    const int lane = threadIdx.x & (kWave - 1);
    const int n_w1 = (a.U ? a.d : 0) + a.d, off_x = a.U ? a.d : 0, np = n_w1 + 3;
    for (int k = threadIdx.x; k < 2 * np; k += blockDim.x) s_gp[k] = 0.0f;
    __syncthreads();
    // a lane owns its columns of the parameter gradient: accumulated in registers over the wave's rows, added to the
    // workgroup's LDS copy once per wave (per-row LDS atomics made this kernel 3x slower than the forward)
    float gw1u[kMaxCols] = {}, gw1x[kMaxCols] = {}, gw2u[kMaxCols] = {}, gw2x[kMaxCols] = {};
    float gs1[3] = {}, gs2[3] = {};
    for (int r = blockIdx.x * kFuseWaves + (threadIdx.x >> 6); r < a.n; r += gridDim.x * kFuseWaves) {
        float u[kMaxCols], x1[kMaxCols], x2[kMaxCols], gg[kMaxCols];
        float r1 = 0.0f, r2 = 0.0f, d1 = 0.0f, d2 = 0.0f;
#pragma unroll
        for (int k = 0; k < kMaxCols; ++k) {
            const int c = lane + k * kWave;
            u[k] = x1[k] = x2[k] = gg[k] = 0.0f;
            if (c < a.d) {
                x1[k] = a.X1[(size_t)r * a.d + c];
                x2[k] = a.X2[(size_t)r * a.d + c];
                gg[k] = g[(size_t)r * a.d + c];
                r1 = fmaf(x1[k], a.p1[off_x + c], r1);
                r2 = fmaf(x2[k], a.p2[off_x + c], r2);
                if (a.U) {
                    u[k] = a.U[(size_t)r * a.d + c];
                    r1 = fmaf(u[k], a.p1[c], r1);
                    r2 = fmaf(u[k], a.p2[c], r2);
                }
                d1 = fmaf(gg[k], x1[k], d1);
                d2 = fmaf(gg[k], x2[k], d2);
            }
        }
        const BranchScalars s1 = branch_forward(wave_sum(r1), a.p1, n_w1, a.c1);
        const BranchScalars s2 = branch_forward(wave_sum(r2), a.p2, n_w1, a.c2);
        const float tot = s1.e + s2.e, a1 = s1.e / tot, a2 = s2.e / tot;
        const float da1 = a.mix_coef * wave_sum(d1), da2 = a.mix_coef * wave_sum(d2);
        const float S = a1 * da1 + a2 * da2;
        // through a_k = e_k / (e_1 + e_2), e_k = exp(l_k) + c_k, l_k = LReLU(q_k), q_k = w2 t_k + b2, t_k = tanh(r_k)
        const float dq1 = (da1 - S) / tot * (s1.e - a.c1) * (s1.q > 0.0f ? 1.0f : 0.2f);
        const float dq2 = (da2 - S) / tot * (s2.e - a.c2) * (s2.q > 0.0f ? 1.0f : 0.2f);
        const float dr1 = dq1 * a.p1[n_w1 + 1] * (1.0f - s1.t * s1.t);
        const float dr2 = dq2 * a.p2[n_w1 + 1] * (1.0f - s2.t * s2.t);
#pragma unroll
        for (int k = 0; k < kMaxCols; ++k) {
            const int c = lane + k * kWave;
            if (c < a.d) {
                gX1[(size_t)r * a.d + c] = a.mix_coef * a1 * gg[k] + dr1 * a.p1[off_x + c];
                gX2[(size_t)r * a.d + c] = a.mix_coef * a2 * gg[k] + dr2 * a.p2[off_x + c];
                gw1x[k] = fmaf(dr1, x1[k], gw1x[k]);
                gw2x[k] = fmaf(dr2, x2[k], gw2x[k]);
                if (a.U) {
                    gU[(size_t)r * a.d + c] = a.base_coef * gg[k] + dr1 * a.p1[c] + dr2 * a.p2[c];
                    gw1u[k] = fmaf(dr1, u[k], gw1u[k]);
                    gw2u[k] = fmaf(dr2, u[k], gw2u[k]);
                }
            }
        }
        gs1[0] += dr1; gs1[1] += dq1 * s1.t; gs1[2] += dq1;       // b1, w2, b2
        gs2[0] += dr2; gs2[1] += dq2 * s2.t; gs2[2] += dq2;
    }
#pragma unroll
    for (int k = 0; k < kMaxCols; ++k) {
        const int c = lane + k * kWave;
        if (c < a.d) {
            atomicAdd(&s_gp[off_x + c], gw1x[k]);
            atomicAdd(&s_gp[np + off_x + c], gw2x[k]);
            if (a.U) {
                atomicAdd(&s_gp[c], gw1u[k]);
                atomicAdd(&s_gp[np + c], gw2u[k]);
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            atomicAdd(&s_gp[n_w1 + j], gs1[j]);
            atomicAdd(&s_gp[np + n_w1 + j], gs2[j]);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < np; k += blockDim.x) {
        atomicAdd(gp1 + k, s_gp[k]);
        atomicAdd(gp2 + k, s_gp[np + k]);
    }
}

int check(const char *fn, const float *X1, const float *X2, const float *p1, const float *p2, int32_t n, int32_t d)
{
    SPEX_CHECK_ARG(X1 && X2 && p1 && p2, "%s: NULL pointer", fn);
    SPEX_CHECK_ARG(n >= 0, "%s: n=%d", fn, n);
    if (d < 1 || d > kMaxCols * kWave) {
        spex::set_error("%s: d = %d not in [1, %d]", fn, d, kMaxCols * kWave);
        return SPEX_ERR_UNSUPPORTED;
    }
    return SPEX_OK;
}

inline unsigned fuse_grid(int n)
{
    int64_t blocks = ((int64_t)n + kFuseWaves - 1) / kFuseWaves;
    if (blocks > 256) blocks = 256;     // one workgroup per CU, grid-stride beyond (a wave then owns a few rows)
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" int spex_attn_fuse_f32(const float *U, const float *X1, const float *X2, const float *p1, const float *p2,
                                  int32_t n, int32_t d, float c1, float c2, float base_coef, float mix_coef, float *out,
                                  void *stream)
{
    if (int rc = check("spex_attn_fuse_f32", X1, X2, p1, p2, n, d)) return rc;
    SPEX_CHECK_ARG(out, "spex_attn_fuse_f32: NULL out");
    if (n == 0) return SPEX_OK;
    const FuseArgs a{U, X1, X2, p1, p2, n, d, c1, c2, base_coef, mix_coef};
    hipLaunchKernelGGL(attn_fuse_kernel, dim3(fuse_grid(n)), dim3(kWave * kFuseWaves), 0, (hipStream_t)stream, a, out);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_attn_fuse_bwd_f32(const float *U, const float *X1, const float *X2, const float *p1, const float *p2,
                                      int32_t n, int32_t d, float c1, float c2, float base_coef, float mix_coef,
                                      const float *grad_out, float *grad_U, float *grad_X1, float *grad_X2, float *grad_p1,
                                      float *grad_p2, void *stream)
{
    if (int rc = check("spex_attn_fuse_bwd_f32", X1, X2, p1, p2, n, d)) return rc;
    SPEX_CHECK_ARG(grad_out && grad_X1 && grad_X2 && grad_p1 && grad_p2 && ((U == nullptr) == (grad_U == nullptr)),
                   "spex_attn_fuse_bwd_f32: NULL pointer (grad_U goes with U)");
    if (n == 0) return SPEX_OK;
    const FuseArgs a{U, X1, X2, p1, p2, n, d, c1, c2, base_coef, mix_coef};
    const size_t lds = 2 * ((size_t)(U ? d : 0) + d + 3) * sizeof(float);
    hipLaunchKernelGGL(attn_fuse_bwd_kernel, dim3(fuse_grid(n)), dim3(kWave * kFuseWaves), lds, (hipStream_t)stream, a, grad_out,
                       grad_U, grad_X1, grad_X2, grad_p1, grad_p2);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
