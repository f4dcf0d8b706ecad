// Learned edge values on a fixed sparsity pattern (the Diffnet++ diffusion layers): refresh the
// handle's stored values from a per-edge array, the row softmax that produces those values, and the sampled
// dense-dense product that is the SpMM's gradient with respect to them.
//
//   tf.sparse.softmax(...)                     Diffnet++_SPEX/code/utility/Model.py:275-286  -> spex_edge_softmax_f32 (+ _bwd)
//   tf.sparse.sparse_dense_matmul(att, emb)    Model.py:18-83          -> spex_graph_set_values + spex_spmm_f32
//   d loss / d values (tape.gradient)          main_rec.py:36          -> spex_sddmm_f32
//
// All three are bandwidth-bound: the softmax streams 8-12 B per entry; the SDDMM gathers one 4d-byte row of B per entry
// exactly like the SpMM (algorithmic bytes nnz (4d + 12) + reads of A, which stay in cache because consecutive
// entries share their row).  Per-edge arrays are addressed by edge id so that a graph and its transposed copy (built
// with the permutation as h_edge_id) read and write the same array.
#include <mutex>

#include "spex_common.h"

using namespace spex;

namespace {

constexpr int kGroup = 16;                 // lanes per entry (SDDMM) / per row (softmax): a quarter wave

__device__ __forceinline__ float group_sum(float v)   // butterfly inside a 16-lane group: every lane ends with the sum
{
#pragma unroll
    for (int off = kGroup / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, kGroup);
    return v;
}

__device__ __forceinline__ float group_max(float v)
{
#pragma unroll
    for (int off = kGroup / 2; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kGroup));
    return v;
}

// ---- values: CSR copy (generic SpMM path) and chunked copy (d == 64 path); padding entries keep value 0
__global__ void set_csr_values_kernel(float *__restrict__ val, const int32_t *__restrict__ edge_id,
                                      const float *__restrict__ src, int64_t nnz)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x)
        val[e] = src[edge_id ? edge_id[e] : e];
}

__global__ void set_chunk_values_kernel(float *__restrict__ chunk_val, const uint32_t *__restrict__ chunk_eid,
                                        const uint32_t *__restrict__ chunk_mask, const float *__restrict__ src,
                                        int64_t n_entries)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_entries; k += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t n_pad = (chunk_mask[k / kChunk] >> 16) & 31u;
        const bool pad = (uint32_t)(k % kChunk) >= (uint32_t)kChunk - n_pad;
        chunk_val[k] = pad ? 0.0f : src[chunk_eid[k]];
    }
}

__global__ void row_of_kernel(int32_t *__restrict__ row_of, const int32_t *__restrict__ rowptr, int32_t n_rows)
{
    const int lane = threadIdx.x & (kGroup - 1);
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    const int64_t n_groups = (int64_t)gridDim.x * blockDim.x / kGroup;
    for (int64_t r = group; r < n_rows; r += n_groups)
        for (int32_t e = rowptr[r] + lane; e < rowptr[r + 1]; e += kGroup) row_of[e] = (int32_t)r;
}

// ---- SDDMM.  A wave owns 64 consecutive stored entries: lane l loads the (row, col, edge id) of entry l with one
// coalesced load each; 16-lane group q then walks entries 16 q .. 16 q + 15, every lane holding VEC consecutive
// columns per 16 VEC-column slab (d == 64, VEC == 4: one float4 of the A row and one of the B row per lane = the
// quarter-wave gather shape, four independent 256-byte rows per load instruction, unrolled four entries deep).  The
// group butterfly leaves the dot product in all 16 lanes; lane (entry mod 16) keeps it, so the 64 results leave the
// wave with one coalesced store (in edge-id order).
template <int VEC>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void sddmm_kernel(const int32_t *__restrict__ row_of,
                                                                     const int32_t *__restrict__ col,
                                                                     const int32_t *__restrict__ edge_id,
                                                                     const float *__restrict__ A,
                                                                     const float *__restrict__ B, float *__restrict__ out,
                                                                     int64_t nnz, int d)
{
    const int lane = threadIdx.x & (kWave - 1), sub = lane & (kGroup - 1), grp = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t base = wave * kWave;
    if (base >= nnz) return;
    const int64_t mine = base + lane;
    int my_row = 0, my_col = 0;
    if (mine < nnz) {
        my_row = __builtin_nontemporal_load(row_of + mine);
        my_col = __builtin_nontemporal_load(col + mine);
    }
    float res = 0.0f;
#pragma unroll 4
    for (int it = 0; it < kGroup; ++it) {
        const int src = grp * kGroup + it;                         // entry base + src, same for the whole group
        const int r = __shfl(my_row, src), c = __shfl(my_col, src);
        const float *a = A + (size_t)r * d, *b = B + (size_t)c * d;
        float p = 0.0f;
        if (VEC == 4) {
            for (int k = 4 * sub; k < d; k += 4 * kGroup) {
                const float4 x = *reinterpret_cast<const float4 *>(a + k), y = *reinterpret_cast<const float4 *>(b + k);
                p = fmaf(x.x, y.x, p);
                p = fmaf(x.y, y.y, p);
                p = fmaf(x.z, y.z, p);
                p = fmaf(x.w, y.w, p);
            }
        } else {
            for (int k = sub; k < d; k += kGroup) p = fmaf(a[k], b[k], p);
        }
        p = group_sum(p);
        if (sub == it) res = p;
    }
    if (mine < nnz) out[edge_id ? edge_id[mine] : mine] = res;   // (entries past nnz computed <A[0], B[0]> and are dropped)
}

// ---- row softmax over stored entries: a 16-lane group per row, three passes over the row's values (the second and
// third hit L1/L2)
__global__ __launch_bounds__(kWave *kWavesPerBlock) void edge_softmax_kernel(const int32_t *__restrict__ rowptr,
                                                                            const int32_t *__restrict__ edge_id,
                                                                            const float *in, float *out, int32_t n_rows)
{
    const int sub = threadIdx.x & (kGroup - 1);
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    const int64_t n_groups = (int64_t)gridDim.x * blockDim.x / kGroup;
    for (int64_t r = group; r < n_rows; r += n_groups) {
        const int32_t b = rowptr[r], e = rowptr[r + 1];
        float m = -INFINITY;
        for (int32_t k = b + sub; k < e; k += kGroup) m = fmaxf(m, in[edge_id ? edge_id[k] : k]);
        m = group_max(m);
        float s = 0.0f;
        for (int32_t k = b + sub; k < e; k += kGroup) s += expf(in[edge_id ? edge_id[k] : k] - m);
        s = group_sum(s);
        for (int32_t k = b + sub; k < e; k += kGroup) {
            const int32_t id = edge_id ? edge_id[k] : k;
            out[id] = expf(in[id] - m) / s;
        }
    }
}

__global__ __launch_bounds__(kWave *kWavesPerBlock) void edge_softmax_bwd_kernel(const int32_t *__restrict__ rowptr,
                                                                                const int32_t *__restrict__ edge_id,
                                                                                const float *__restrict__ y, const float *gy,
                                                                                float *gx, int32_t n_rows)
{
    const int sub = threadIdx.x & (kGroup - 1);
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    const int64_t n_groups = (int64_t)gridDim.x * blockDim.x / kGroup;
    for (int64_t r = group; r < n_rows; r += n_groups) {
        const int32_t b = rowptr[r], e = rowptr[r + 1];
        float s = 0.0f;
        for (int32_t k = b + sub; k < e; k += kGroup) {
            const int32_t id = edge_id ? edge_id[k] : k;
            s = fmaf(y[id], gy[id], s);
        }
        s = group_sum(s);
        for (int32_t k = b + sub; k < e; k += kGroup) {
            const int32_t id = edge_id ? edge_id[k] : k;
            gx[id] = y[id] * (gy[id] - s);
        }
    }
}

inline unsigned stream_grid(int64_t n, int per_block)
{
    int64_t blocks = (n + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 32) blocks = 256 * 32;   // grid-stride beyond 32 workgroups per CU
    return (unsigned)blocks;
}

std::mutex g_row_of_mutex;

}  // namespace

extern "C" int spex_graph_set_values(spex_graph_t *g, const float *d_val, int64_t n_val, void *stream)
{
    SPEX_CHECK_ARG(g && (d_val || g->nnz == 0), "spex_graph_set_values: NULL handle / values");
    SPEX_CHECK_ARG(n_val > g->max_edge_id, "spex_graph_set_values: %lld values but the largest edge id is %lld",
                   (long long)n_val, (long long)g->max_edge_id);
    if (g->nnz == 0) return SPEX_OK;
    const int threads = kWave * kWavesPerBlock;
    hipLaunchKernelGGL(set_csr_values_kernel, dim3(stream_grid(g->nnz, threads)), dim3(threads), 0, (hipStream_t)stream, g->val,
                       g->edge_id, d_val, g->nnz);
    const int64_t n_entries = g->n_chunks * kChunk;
    if (n_entries > 0)
        hipLaunchKernelGGL(set_chunk_values_kernel, dim3(stream_grid(n_entries, threads)), dim3(threads), 0, (hipStream_t)stream,
                           g->chunk_val, g->chunk_eid, g->chunk_mask, d_val, n_entries);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_sddmm_f32(spex_graph_t *g, const float *A, const float *B, float *d_out, int64_t n_val, int32_t d,
                              void *stream)
{
    SPEX_CHECK_ARG(g && d >= 1, "spex_sddmm_f32: NULL handle or d=%d", d);
    SPEX_CHECK_ARG(n_val > g->max_edge_id, "spex_sddmm_f32: %lld outputs but the largest edge id is %lld", (long long)n_val,
                   (long long)g->max_edge_id);
    if (g->nnz == 0) return SPEX_OK;
    SPEX_CHECK_ARG(A && B && d_out, "spex_sddmm_f32: NULL pointer");
    {
        std::lock_guard<std::mutex> lock(g_row_of_mutex);
        if (!g->row_of) {
            int32_t *p = nullptr;
            SPEX_HIP(hipMalloc((void **)&p, (size_t)g->nnz * sizeof(int32_t)));
            const int threads = kWave * kWavesPerBlock;
            // on the caller's stream: the first SDDMM (and any other stream's, ordered by the caller) follows it
            hipLaunchKernelGGL(row_of_kernel, dim3(stream_grid((int64_t)g->n_rows * kGroup, threads)), dim3(threads), 0,
                               (hipStream_t)stream, p, g->rowptr, g->n_rows);
            SPEX_HIP(hipGetLastError());
            g->row_of = p;
        }
    }
    const int64_t n_waves = (g->nnz + kWave - 1) / kWave;
    const int64_t blocks = (n_waves + kWavesPerBlock - 1) / kWavesPerBlock;
    SPEX_CHECK_ARG(blocks <= 0x7FFFFFFF, "spex_sddmm_f32: grid too large");
    const bool vec = (d % 4 == 0) && ((((uintptr_t)A | (uintptr_t)B) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(sddmm_kernel<4>, dim3((unsigned)blocks), dim3(kWave * kWavesPerBlock), 0, (hipStream_t)stream, g->row_of,
                           g->col, g->edge_id, A, B, d_out, g->nnz, d);
    else
        hipLaunchKernelGGL(sddmm_kernel<1>, dim3((unsigned)blocks), dim3(kWave * kWavesPerBlock), 0, (hipStream_t)stream, g->row_of,
                           g->col, g->edge_id, A, B, d_out, g->nnz, d);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_edge_softmax_f32(const spex_graph_t *g, const float *d_in, float *d_out, int64_t n_val, void *stream)
{
    SPEX_CHECK_ARG(g, "spex_edge_softmax_f32: NULL handle");
    SPEX_CHECK_ARG(n_val > g->max_edge_id, "spex_edge_softmax_f32: %lld values but the largest edge id is %lld", (long long)n_val,
                   (long long)g->max_edge_id);
    if (g->nnz == 0) return SPEX_OK;
    SPEX_CHECK_ARG(d_in && d_out, "spex_edge_softmax_f32: NULL pointer");
    const int threads = kWave * kWavesPerBlock;
    hipLaunchKernelGGL(edge_softmax_kernel, dim3(stream_grid((int64_t)g->n_rows * kGroup, threads)), dim3(threads), 0,
                       (hipStream_t)stream, g->rowptr, g->edge_id, d_in, d_out, g->n_rows);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_edge_softmax_bwd_f32(const spex_graph_t *g, const float *d_out_val, const float *d_grad_out,
                                         float *d_grad_in, int64_t n_val, void *stream)
{
    SPEX_CHECK_ARG(g, "spex_edge_softmax_bwd_f32: NULL handle");
    SPEX_CHECK_ARG(n_val > g->max_edge_id, "spex_edge_softmax_bwd_f32: %lld values but the largest edge id is %lld",
                   (long long)n_val, (long long)g->max_edge_id);
    if (g->nnz == 0) return SPEX_OK;
    SPEX_CHECK_ARG(d_out_val && d_grad_out && d_grad_in, "spex_edge_softmax_bwd_f32: NULL pointer");
    const int threads = kWave * kWavesPerBlock;
    hipLaunchKernelGGL(edge_softmax_bwd_kernel, dim3(stream_grid((int64_t)g->n_rows * kGroup, threads)), dim3(threads), 0,
                       (hipStream_t)stream, g->rowptr, g->edge_id, d_out_val, d_grad_out, d_grad_in, g->n_rows);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
