// Learned edge values on a fixed sparsity pattern (the Diffnet++ diffusion layers): refresh the
// handle's stored values from a per-edge array, the row softmax that produces those values, and the sampled
// dense-dense product that is the SpMM's gradient with respect to them.
//
//   tf.sparse.softmax(...)                     Diffnet++_SPEX/code/utility/Model.py:275-286  -> spex_edge_softmax_f32 (+ _bwd)
//   tf.sparse.sparse_dense_matmul(att, emb)    Model.py:18-83          -> spex_graph_set_values + spex_spmm_f32
//   d loss / d values (tape.gradient)          main_rec.py:36          -> spex_sddmm_f32
//
// The SDDMM gathers one 4d-byte row of B per entry exactly like the SpMM and runs at the same gather ceiling (its A
// rows are cache hits: consecutive entries share their row; algorithmic bytes nnz (4d + 12) + n_rows 4d).  The softmax
// streams 8-12 B per entry but is issue-bound (measured, see its kernel).  Per-edge arrays are addressed by edge id so
// that a graph and its transposed copy (built with the permutation as h_edge_id) read and write the same array.
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

// ---- values: CSR copy (generic SpMM path) and chunked copy (d = 64 / 128 / 256 path); padding entries keep value 0
// one launch refreshes both copies
__global__ void set_values_kernel(float *__restrict__ val, const int32_t *__restrict__ edge_id, int64_t nnz,
                                  float *__restrict__ chunk_val, const uint32_t *__restrict__ chunk_eid,
                                  const uint8_t *__restrict__ chunk_pad, int64_t n_entries, const float *__restrict__ src)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t e = first; e < nnz; e += stride) val[e] = src[edge_id ? edge_id[e] : e];
    for (int64_t k = first; k < n_entries; k += stride) {
        const uint32_t n_pad = chunk_pad[k / kChunk];
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

// ---- row softmax over stored entries.  A 256-thread workgroup owns the rows that START inside its tile of
// kSoftmaxTile consecutive stored entries (spex_graph::tile_row, built with the handle).  It first pulls the tile's
// values and row pointers into LDS with coalesced loads, all in flight at once, then a 16-lane group per row reduces
// from LDS (rows of up to kWgRowMax = 1024 entries: at most 64 LDS-only trips), and the results leave with coalesced
// stores.  Hub rows (> 1024 entries, the handle's hub list) take a workgroup each, three passes (the later ones hit
// L2), reductions through LDS.
constexpr int kTileRowCap = 1024;          // row pointers of a tile staged in LDS (rows past that are read from memory)

template <bool BWD>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void edge_softmax_tile_kernel(const int32_t *__restrict__ rowptr,
                                                                                 const int32_t *__restrict__ tile_row,
                                                                                 const int32_t *__restrict__ edge_id,
                                                                                 const float *in, const float *gy,
                                                                                 float *out, int tile)
{
    __shared__ float s_x[kSoftmaxTile + kWgRowMax];
    __shared__ float s_g[BWD ? kSoftmaxTile + kWgRowMax : 1];
    __shared__ uint8_t s_own[kSoftmaxTile + kWgRowMax];             // entry belongs to a row this workgroup reduced
    __shared__ int32_t s_rp[kTileRowCap + 1];
    const int32_t r_lo = tile_row[blockIdx.x], r_hi = tile_row[blockIdx.x + 1];
    if (r_lo >= r_hi) return;                                      // no row starts in this tile (inside a long row)
    const int32_t e_lo = rowptr[r_lo];
    int32_t e_hi = rowptr[r_hi];
    const int64_t cap = ((int64_t)blockIdx.x + 1) * tile + kWgRowMax;   // a non-hub last row ends before this
    if ((int64_t)e_hi > cap) e_hi = (int32_t)cap;
    // everything the workgroup needs from memory is requested here, coalesced and all in flight at once; the per-row
    // phase below touches LDS only.  Measured (Epinion2 x 269, 113 M entries): 0.55-0.6 ms = 1.5-1.7 TB/s of the 8 B per
    // entry it moves — issue-bound, not memory-bound: a 16-lane group per row means 4 rows (~108 entries) per wave
    // trip at ~150 instructions (row bounds, three short loops, two butterflies, exp, divide); variants that read
    // straight from memory or held the row in registers ran at the same speed.  It is ~7 % of a Diffnet++ step next
    // to the SpMM / SDDMM launches (4 ms each on that graph).
    for (int32_t k = e_lo + (int32_t)threadIdx.x; k < e_hi; k += blockDim.x) {
        const int32_t id = edge_id ? edge_id[k] : k;
        s_x[k - e_lo] = in[id];
        if (BWD) s_g[k - e_lo] = gy[id];
        s_own[k - e_lo] = 0;
    }
    const int32_t n_staged = (r_hi - r_lo < kTileRowCap) ? r_hi - r_lo : kTileRowCap;
    for (int32_t i = threadIdx.x; i <= n_staged; i += blockDim.x) s_rp[i] = rowptr[r_lo + i];
    __syncthreads();
    const int sub = threadIdx.x & (kGroup - 1), grp = threadIdx.x / kGroup;
    for (int32_t i = grp; i < r_hi - r_lo; i += (kWave * kWavesPerBlock) / kGroup) {
        const int32_t b = i < n_staged ? s_rp[i] : rowptr[r_lo + i];
        const int32_t len = (i < n_staged ? s_rp[i + 1] : rowptr[r_lo + i + 1]) - b;
        if (len > kWgRowMax) continue;                             // group-uniform; the hub kernel owns this row
        float *x = s_x + (b - e_lo);
        uint8_t *own = s_own + (b - e_lo);
        if (!BWD) {
            float m = -INFINITY;
            for (int k = sub; k < len; k += kGroup) m = fmaxf(m, x[k]);
            m = group_max(m);
            float s = 0.0f;
            for (int k = sub; k < len; k += kGroup) {
                const float ex = expf(x[k] - m);
                x[k] = ex;
                s += ex;
            }
            s = group_sum(s);
            for (int k = sub; k < len; k += kGroup) {
                x[k] = x[k] / s;
                own[k] = 1;
            }
        } else {                                                   // in = y, gy = dL/dy
            const float *w = s_g + (b - e_lo);
            float s = 0.0f;
            for (int k = sub; k < len; k += kGroup) s = fmaf(x[k], w[k], s);
            s = group_sum(s);
            for (int k = sub; k < len; k += kGroup) {
                x[k] = x[k] * (w[k] - s);
                own[k] = 1;
            }
        }
    }
    __syncthreads();
    for (int32_t k = e_lo + (int32_t)threadIdx.x; k < e_hi; k += blockDim.x)   // coalesced write-back
        if (s_own[k - e_lo]) out[edge_id ? edge_id[k] : k] = s_x[k - e_lo];
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float *s_red)
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(v, off, kWave);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();                                               // s_red may still be read from the previous reduction
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    v = s_red[0];
#pragma unroll
    for (int i = 1; i < kWavesPerBlock; ++i) v = is_max ? fmaxf(v, s_red[i]) : v + s_red[i];
    return v;
}

template <bool BWD>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void edge_softmax_hub_kernel(const int32_t *__restrict__ rowptr,
                                                                                 const int32_t *__restrict__ hub_row,
                                                                                 const int32_t *__restrict__ edge_id,
                                                                                 const float *in, const float *gy,
                                                                                 float *out)
{
    __shared__ float s_red[kWavesPerBlock];
    const int32_t r = hub_row[blockIdx.x];
    const int32_t b = rowptr[r], e = rowptr[r + 1];
    if (!BWD) {
        float m = -INFINITY;
        for (int32_t k = b + threadIdx.x; k < e; k += blockDim.x) m = fmaxf(m, in[edge_id ? edge_id[k] : k]);
        m = block_reduce(m, true, s_red);
        float s = 0.0f;
        for (int32_t k = b + threadIdx.x; k < e; k += blockDim.x) s += expf(in[edge_id ? edge_id[k] : k] - m);
        s = block_reduce(s, false, s_red);
        for (int32_t k = b + threadIdx.x; k < e; k += blockDim.x) {
            const int32_t id = edge_id ? edge_id[k] : k;
            out[id] = expf(in[id] - m) / s;
        }
    } else {
        float s = 0.0f;
        for (int32_t k = b + threadIdx.x; k < e; k += blockDim.x) {
            const int32_t id = edge_id ? edge_id[k] : k;
            s = fmaf(in[id], gy[id], s);
        }
        s = block_reduce(s, false, s_red);
        for (int32_t k = b + threadIdx.x; k < e; k += blockDim.x) {
            const int32_t id = edge_id ? edge_id[k] : k;
            out[id] = in[id] * (gy[id] - s);
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
    const int64_t n_entries = g->n_chunks * kChunk;
    hipLaunchKernelGGL(set_values_kernel, dim3(stream_grid(n_entries > g->nnz ? n_entries : g->nnz, threads)), dim3(threads), 0,
                       (hipStream_t)stream, g->val, g->edge_id, g->nnz, g->chunk_val, g->chunk_eid, g->chunk_pad, n_entries, d_val);
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

template <bool BWD>
static int launch_edge_softmax_impl(const spex_graph *g, const float *in, const float *gy, float *out, hipStream_t stream)
{
    const int threads = kWave * kWavesPerBlock;
    hipLaunchKernelGGL(edge_softmax_tile_kernel<BWD>, dim3((unsigned)g->n_tiles), dim3(threads), 0, stream, g->rowptr, g->tile_row,
                       g->edge_id, in, gy, out, g->tile);
    if (g->n_hub > 0)
        hipLaunchKernelGGL(edge_softmax_hub_kernel<BWD>, dim3((unsigned)g->n_hub), dim3(threads), 0, stream, g->rowptr,
                           g->hub_row, g->edge_id, in, gy, out);
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
    return launch_edge_softmax_impl<false>(g, d_in, nullptr, d_out, (hipStream_t)stream);
}

extern "C" int spex_edge_softmax_bwd_f32(const spex_graph_t *g, const float *d_out_val, const float *d_grad_out,
                                         float *d_grad_in, int64_t n_val, void *stream)
{
    SPEX_CHECK_ARG(g, "spex_edge_softmax_bwd_f32: NULL handle");
    SPEX_CHECK_ARG(n_val > g->max_edge_id, "spex_edge_softmax_bwd_f32: %lld values but the largest edge id is %lld",
                   (long long)n_val, (long long)g->max_edge_id);
    if (g->nnz == 0) return SPEX_OK;
    SPEX_CHECK_ARG(d_out_val && d_grad_out && d_grad_in, "spex_edge_softmax_bwd_f32: NULL pointer");
    return launch_edge_softmax_impl<true>(g, d_out_val, d_grad_out, d_grad_in, (hipStream_t)stream);
}
