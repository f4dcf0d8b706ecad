// Path attention of the trust head: the reference's GraphAttentionLayer,
// LightGCN_SPEX/code/utility2/layers.py:15-71, which walks batch x path positions in Python, building a 2 x 2H matrix
// per position and calling mm / softmax on it (with .item() syncs and host->device copies, :20-23).
//
// Closed form for a path x_0 .. x_{l-1} of batch row p, position i < l - 1 (n_heads parameter vectors a_h = [a1_h | a2_h]):
//     positional (concat=True, :22-31):  A = src[x_i] + (l - i),   Bv = src[x_{i+1}] + (l - i - 1)
//     dense      (concat=False, :58-63): A = src[p, i],            Bv = src[p, i + 1]
//     att = softmax([A.a1 + A.a2,  A.a1 + Bv.a2]);   out_h[p, i] = att_0 A + att_1 Bv
// and out_h[p, i] = the raw source row for i >= l - 1.  All heads of a position are produced by one wave (lane ==
// column): two row loads, three dot products per head, n_heads x d outputs — tiny and latency-bound; the point is ONE
// launch instead of ~40 small tensor ops per head.  The backward is the matching kernel: A.a1 cancels in the softmax
// (the gradient of a1 is exactly 0), with z = (A - Bv).a2, w0 = sigmoid(z):
//     dz = (g.(A - Bv)) w0 (1 - w0);   dA = w0 g + dz a2;   dBv = (1 - w0) g - dz a2;   da2 += dz (A - Bv)
// Row gradients are added with atomics (a row is A of one position and Bv of the previous one, and users repeat across
// paths); the a2 gradient is reduced per workgroup in LDS first.
#include "spex_common.h"

using namespace spex;

namespace {

constexpr int kPathWaves = 16;   // 1024-thread workgroups: few same-address atomics for the parameter gradient
constexpr int kMaxCols = 4;      // d <= 256: columns per lane held in registers

__device__ __forceinline__ float wave_sum(float v)
{
    return wave_sum_f32(v);
}

struct PathArgs {
    const float *src;        // [n_src_rows, d]
    const int64_t *seq;      // [B, L] row indices into src, or NULL (dense: row p * L + i)
    const int64_t *seq_l;    // [B]
    const float *a;          // [n_heads, 2 d]
    int64_t n_src_rows;
    int B, L, d, n_heads, positional;
};

__device__ __forceinline__ bool position_rows(const PathArgs &p, int pos, int64_t &row_a, int64_t &row_b, int &len, int &i)
{
    const int b = pos / p.L;
    i = pos - b * p.L;
    len = (int)p.seq_l[b];
    const bool valid = i < len - 1;
    if (p.seq) {
        row_a = p.seq[pos];
        row_b = valid ? p.seq[pos + 1] : row_a;
    } else {
        row_a = pos;
        row_b = valid ? pos + 1 : pos;
    }
    return valid;
}

__global__ __launch_bounds__(kWave *kPathWaves) void path_attention_kernel(const PathArgs p, float *__restrict__ out,
                                                                          float *__restrict__ w0_out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int pos = blockIdx.x * kPathWaves + (threadIdx.x >> 6);
    if (pos >= p.B * p.L) return;
    int64_t ra, rb;
    int len, i;
    const bool valid = position_rows(p, pos, ra, rb, len, i);
    const bool in_range = ra >= 0 && ra < p.n_src_rows && rb >= 0 && rb < p.n_src_rows;   // never gather out of bounds
    const float off_a = p.positional ? (float)(len - i) : 0.0f, off_b = p.positional ? (float)(len - i - 1) : 0.0f;
    float A[kMaxCols], Bv[kMaxCols], raw[kMaxCols];
#pragma unroll
    for (int k = 0; k < kMaxCols; ++k) {
        const int c = lane + k * kWave;
        raw[k] = (c < p.d && in_range) ? p.src[(size_t)ra * p.d + c] : 0.0f;
        A[k] = raw[k] + off_a;
        Bv[k] = ((c < p.d && in_range && valid) ? p.src[(size_t)rb * p.d + c] : 0.0f) + off_b;
    }
    float *o = out + (size_t)pos * p.n_heads * p.d;
    for (int h = 0; h < p.n_heads; ++h) {
        const float *a1 = p.a + (size_t)h * 2 * p.d, *a2 = a1 + p.d;
        float w0 = 1.0f, w1 = 0.0f;
        if (valid) {
            float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
            for (int k = 0; k < kMaxCols; ++k) {
                const int c = lane + k * kWave;
                if (c < p.d) {
                    s1 = fmaf(A[k], a1[c], s1);
                    s2 = fmaf(A[k], a2[c], s2);
                    s3 = fmaf(Bv[k], a2[c], s3);
                }
            }
            s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
            const float att0 = s1 + s2, att1 = s1 + s3, m = fmaxf(att0, att1);
            const float e0 = expf(att0 - m), e1 = expf(att1 - m);
            w0 = e0 / (e0 + e1);
            w1 = e1 / (e0 + e1);
        }
#pragma unroll
        for (int k = 0; k < kMaxCols; ++k) {
            const int c = lane + k * kWave;
            if (c < p.d) o[(size_t)h * p.d + c] = valid ? w0 * A[k] + w1 * Bv[k] : raw[k];
        }
        if (w0_out && lane == 0) w0_out[(size_t)pos * p.n_heads + h] = w0;
    }
}

__global__ __launch_bounds__(kWave *kPathWaves) void path_attention_bwd_kernel(const PathArgs p,
                                                                              const float *__restrict__ w0_in,
                                                                              const float *__restrict__ grad_out,
                                                                              float *grad_src, float *grad_a)
{
    extern __shared__ float s_ga[];   // [n_heads, d] gradient of the a2 halves, reduced per workgroup
    const int lane = threadIdx.x & (kWave - 1);
    const int pos = blockIdx.x * kPathWaves + (threadIdx.x >> 6);
    for (int k = threadIdx.x; k < p.n_heads * p.d; k += blockDim.x) s_ga[k] = 0.0f;
    __syncthreads();
    if (pos < p.B * p.L) {
        int64_t ra, rb;
        int len, i;
        const bool valid = position_rows(p, pos, ra, rb, len, i);
        const bool in_range = ra >= 0 && ra < p.n_src_rows && rb >= 0 && rb < p.n_src_rows;
        if (in_range) {
            const float off = (p.positional && valid) ? 1.0f : 0.0f;   // (l - i) - (l - i - 1)
            float delta[kMaxCols], dA[kMaxCols], dB[kMaxCols];
#pragma unroll
            for (int k = 0; k < kMaxCols; ++k) {
                const int c = lane + k * kWave;
                dA[k] = dB[k] = 0.0f;
                delta[k] = (c < p.d && valid) ? p.src[(size_t)ra * p.d + c] - p.src[(size_t)rb * p.d + c] + off : 0.0f;
            }
            const float *g = grad_out + (size_t)pos * p.n_heads * p.d;
            for (int h = 0; h < p.n_heads; ++h) {
                const float *a2 = p.a + (size_t)h * 2 * p.d + p.d;
                float gk[kMaxCols];
                float dw = 0.0f;
#pragma unroll
                for (int k = 0; k < kMaxCols; ++k) {
                    const int c = lane + k * kWave;
                    gk[k] = c < p.d ? g[(size_t)h * p.d + c] : 0.0f;
                    dw = fmaf(gk[k], delta[k], dw);
                }
                if (valid) {
                    const float w0 = w0_in[(size_t)pos * p.n_heads + h];
                    const float dz = wave_sum(dw) * w0 * (1.0f - w0);
#pragma unroll
                    for (int k = 0; k < kMaxCols; ++k) {
                        const int c = lane + k * kWave;
                        if (c < p.d) {
                            dA[k] += w0 * gk[k] + dz * a2[c];
                            dB[k] += (1.0f - w0) * gk[k] - dz * a2[c];
                            atomicAdd(&s_ga[h * p.d + c], dz * delta[k]);
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < kMaxCols; ++k) dA[k] += gk[k];   // out = raw row
                }
            }
#pragma unroll
            for (int k = 0; k < kMaxCols; ++k) {
                const int c = lane + k * kWave;
                if (c < p.d) {
                    atomicAdd(grad_src + (size_t)ra * p.d + c, dA[k]);
                    if (valid) atomicAdd(grad_src + (size_t)rb * p.d + c, dB[k]);
                }
            }
        }
    }
    __syncthreads();
    if (grad_a)
        for (int k = threadIdx.x; k < p.n_heads * p.d; k += blockDim.x) {
            const int h = k / p.d, c = k - h * p.d;
            const float v = s_ga[k];
            if (v != 0.0f) atomicAdd(grad_a + (size_t)h * 2 * p.d + p.d + c, v);
        }
}

int check_args(const char *fn, const float *src, int64_t n_src_rows, const int64_t *seq, const int64_t *seq_l, const float *a,
               int32_t B, int32_t L, int32_t d, int32_t n_heads)
{
    SPEX_CHECK_ARG(src && seq_l && a, "%s: NULL pointer", fn);
    SPEX_CHECK_ARG(B >= 0 && L >= 1 && n_heads >= 1, "%s: B=%d L=%d n_heads=%d", fn, B, L, n_heads);
    if (d < 1 || d > kMaxCols * kWave) {
        spex::set_error("%s: d = %d not in [1, %d]", fn, d, kMaxCols * kWave);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG(seq || n_src_rows == (int64_t)B * L, "%s: dense source needs n_src_rows == B * L", fn);
    return SPEX_OK;
}

}  // namespace

extern "C" int spex_path_attention_f32(const float *src, int64_t n_src_rows, const int64_t *seq, const int64_t *seq_l,
                                       const float *a, int32_t B, int32_t L, int32_t d, int32_t n_heads, int32_t positional,
                                       float *out, float *w0_out, void *stream)
{
    if (int rc = check_args("spex_path_attention_f32", src, n_src_rows, seq, seq_l, a, B, L, d, n_heads)) return rc;
    SPEX_CHECK_ARG(out, "spex_path_attention_f32: NULL out");
    if (B == 0) return SPEX_OK;
    const PathArgs p{src, seq, seq_l, a, n_src_rows, B, L, d, n_heads, positional};
    const int64_t blocks = ((int64_t)B * L + kPathWaves - 1) / kPathWaves;
    hipLaunchKernelGGL(path_attention_kernel, dim3((unsigned)blocks), dim3(kWave * kPathWaves), 0, (hipStream_t)stream, p, out,
                       w0_out);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_path_attention_bwd_f32(const float *src, int64_t n_src_rows, const int64_t *seq, const int64_t *seq_l,
                                           const float *a, int32_t B, int32_t L, int32_t d, int32_t n_heads,
                                           int32_t positional, const float *w0, const float *grad_out, float *grad_src,
                                           float *grad_a, void *stream)
{
    if (int rc = check_args("spex_path_attention_bwd_f32", src, n_src_rows, seq, seq_l, a, B, L, d, n_heads)) return rc;
    SPEX_CHECK_ARG(w0 && grad_out && grad_src, "spex_path_attention_bwd_f32: NULL pointer");
    SPEX_CHECK_ARG((size_t)n_heads * d * sizeof(float) <= 64 * 1024, "spex_path_attention_bwd_f32: n_heads * d too large");
    if (B == 0) return SPEX_OK;
    const PathArgs p{src, seq, seq_l, a, n_src_rows, B, L, d, n_heads, positional};
    const int64_t blocks = ((int64_t)B * L + kPathWaves - 1) / kPathWaves;
    hipLaunchKernelGGL(path_attention_bwd_kernel, dim3((unsigned)blocks), dim3(kWave * kPathWaves),
                       (size_t)n_heads * d * sizeof(float), (hipStream_t)stream, p, w0, grad_out, grad_src, grad_a);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
