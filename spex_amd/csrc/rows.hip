// Row-sparse pieces of the training step's backward pass.
//
// After a 256-sample batch the gradient with respect to the propagated table is non-zero on at most 512 rows (the
// batch's users and items, LightGCN_SPEX/code/utility1/model.py:115-116; NGCF_SPEX/code/main_rec.py:89-90).  The first
// product of the backward pass, A^T g, therefore touches ~14 k of Epinion2's 418 k stored entries: it is evaluated in
// PUSH form — for every non-zero row r of g, out[c] += A[r, c] * g[r] over the stored entries of row r — instead of the
// pull-form SpMM over the whole matrix (15 us).  SURVEY.md 7 "hard parts" names this.
//
//   spex_unique_rows_i32      the distinct rows of a batch (two index lists with offsets) as a compact device list;
//                             deduplication by an epoch-stamp table, no sort, no host round trip
//   spex_spmm_push_rows_f32   out[col[e], :] += scale * val[e] * src_k  for every stored entry e of every listed row (k-th
//                             listed row r: src_k = src[r] or src[k]), optionally out[r, :] += scale * add_k — 256-byte float atomics
// Sums arrive in arbitrary order (atomics): results agree with the pull form to fp32 re-association.
#include "spex_common.h"

using namespace spex;

namespace {

__global__ __launch_bounds__(256) void unique_rows_kernel(const int64_t *__restrict__ idx_a, int n_a, int64_t off_a,
                                                          const int64_t *__restrict__ idx_b, int n_b, int64_t off_b,
                                                          int32_t n_rows, int32_t *stamp, int32_t epoch, int32_t *list,
                                                          int32_t *count)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_a + n_b) return;
    const int64_t r = k < n_a ? idx_a[k] + off_a : idx_b[k - n_a] + off_b;
    if (r < 0 || r >= n_rows) return;
    if (atomicExch(stamp + r, epoch) != epoch) list[atomicAdd(count, 1)] = (int32_t)r;   // first visitor of the row this epoch
}

// One 16-wave workgroup per listed row: wave w takes the row's entries w, w + 16, ... (a hub row of 1 000 entries is 64
// atomics per wave, not 1 000 in one).
__global__ __launch_bounds__(kWave *kWgWaves) void spmm_push_rows_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const int32_t *__restrict__ list, const int32_t *__restrict__ count, int n_rows, const float *__restrict__ src,
    int src_indexed, const float *__restrict__ add, int add_indexed, float scale, float *out, int d)
{
    const int k = blockIdx.x;
    if (k >= *count) return;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int r = list[k];
    if (r < 0 || r >= n_rows) return;                      // never index the matrix out of range
    const int beg = rowptr[r], end = rowptr[r + 1];
    for (int c0 = 0; c0 < d; c0 += kWave) {
        const int c = c0 + lane;
        const bool col_ok = c < d;                         // (every lane stays in the loop: lanes 0..15 carry the run's metadata)
        const float g = col_ok ? scale * src[(size_t)(src_indexed ? r : k) * d + c] : 0.0f;
        for (int base = beg + wave * 16; base < end; base += kWgWaves * 16) {   // 16-entry runs: one coalesced (col, val) load each
            const int cnt = end - base < 16 ? end - base : 16;
            int my_col = 0;
            float my_val = 0.0f;
            if (lane < cnt) {
                my_col = col[base + lane];
                my_val = val[base + lane];
            }
            // lane 0 is read before the first atomic and the entry loop is a real loop: see spmm_push_batch_kernel
            const int cc0 = __builtin_amdgcn_readlane(my_col, 0);
            const float v0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_val), 0));
            if (col_ok) atomicAdd(out + (size_t)cc0 * d + c, v0 * g);
#pragma unroll 1
            for (int j = 1; j < cnt; ++j) {
                const int cc = __builtin_amdgcn_readlane(my_col, j);
                const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_val), j));
                if (col_ok) atomicAdd(out + (size_t)cc * d + c, v * g);
            }
        }
        if (add && wave == 0 && col_ok) atomicAdd(out + (size_t)r * d + c, scale * add[(size_t)(add_indexed ? r : k) * d + c]);
    }
}

// The same product driven by the batch itself, EVERY slot contributing: slot k (row idx_a[k] + off_a, then idx_b[k] + off_b)
// pushes its own source row src[k] — the slot's own gradient row as the scoring kernel leaves it — so rows named by several
// slots simply receive several contributions and nothing has to be deduplicated (a first version tested "is this the first
// slot naming the row" with a scan of the batch in every wave: 128 MB of index reads per launch, 16 us).  kPushParts
// workgroups of 4 waves share a slot's row, entry e going to (part, wave) = (e mod 4, (e div 4) mod 4): the ~50 ns a CU needs
// per 256-byte float atomic would otherwise put a 1 000-entry hub row's atomics on one CU.
constexpr int kPushParts = 4;
constexpr int kPushWaves = 4;

__global__ __launch_bounds__(kWave *kPushWaves) void spmm_push_batch_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val, int n_rows,
    const int64_t *__restrict__ idx_a, int n_a, int64_t off_a, const int64_t *__restrict__ idx_b, int n_b, int64_t off_b,
    const float *__restrict__ src, int ld_src, const float *__restrict__ add, int ld_add, float scale, float *out)
{
    const int k = blockIdx.x / kPushParts, part = blockIdx.x % kPushParts;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const long long r = batch_row(idx_a, n_a, off_a, idx_b, off_b, k);
    if (r < 0 || r >= n_rows) return;
    const int beg = rowptr[r], end = rowptr[r + 1];
    const float g = scale * src[(size_t)k * ld_src + lane];
    // 16-entry runs dealt round-robin to the slot's 16 waves: ONE coalesced load of the run's (col, val) pairs — lane j holds
    // entry j — then the atomics are issued back to back (they return nothing, so nothing waits).  The first version walked
    // the row entry by entry, each atomic behind its own col/val load: a 500-entry row took ~30 dependent round trips per wave
    // and set the launch time (11-22 us).
    // 16-entry runs dealt round-robin to the slot's 16 waves: ONE coalesced load of a run's (col, val) pairs — lane j holds
    // entry j — and v_readlane hands them to the atomics.  Two things matter for the atomics to leave back to back:
    //   * every run of the wave (4 at a time: rows of up to 1 024 entries in one go) is loaded, and lane 0 of each is read,
    //     BEFORE the first atomic — gfx9 has one vmcnt counter for loads and atomics, and the compiler's wait insertion
    //     re-waits conservatively at every control-flow join while a load is pending: with the read inside the guarded,
    //     unrolled loop it put `s_waitcnt vmcnt(0)` in front of EVERY atomic (each then waited out its predecessor's
    //     ~350 ns round trip: 13.5 us for batches whose longest row had <= 512 entries, 19 us up to 768, 25 us up to 1 024);
    //   * the entry loop is a real loop (dynamic trip count), not 16 guarded copies.
    const int w16 = part * kPushWaves + wave;
    constexpr int kStride = kPushParts * kPushWaves * 16, kPre = 4;
    float *out_l = out + lane;
    for (int base0 = beg + w16 * 16; base0 < end; base0 += kPre * kStride) {
        int my_col[kPre], c0[kPre];
        float my_val[kPre], v0[kPre];
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            const int base = base0 + p * kStride;
            my_col[p] = 0;
            my_val[p] = 0.0f;
            if (base + lane < end && lane < 16) {
                my_col[p] = col[base + lane];
                my_val[p] = val[base + lane];
            }
        }
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            c0[p] = __builtin_amdgcn_readlane(my_col[p], 0);
            v0[p] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_val[p]), 0));
        }
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            const int base = base0 + p * kStride;
            const int cnt = end - base < 16 ? end - base : 16;           // (<= 0 past the row's end)
            if (cnt > 0) atomicAdd(out_l + (size_t)c0[p] * kWave, v0[p] * g);
#pragma unroll 1
            for (int j = 1; j < cnt; ++j) {
                const int c = __builtin_amdgcn_readlane(my_col[p], j);
                const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_val[p]), j));
                atomicAdd(out_l + (size_t)c * kWave, v * g);
            }
        }
    }
    if (add && part == 0 && wave == 0) atomicAdd(out + (size_t)r * kWave + lane, scale * add[(size_t)k * ld_add + lane]);
}

// Deterministic accumulation of a batch's per-slot rows into a dense table — the atomic-free alternative to the float
// atomics above (SPEX_STEP_DETERMINISTIC).  out[r] = scale * (slots[k1] + slots[k2] + ...) over the slots k1 < k2 < ... that
// name row r, in ASCENDING slot order — the order in which the reference's CPU index backward (index_put_ with accumulate,
// LightGCN_SPEX/code/utility1/model.py:115-116 under autograd) adds them, so results repeat bit for bit from run to run.
// One wave per slot: it scans the batch's row list (64 slots per load), leaves if a lower-numbered slot names the same row,
// otherwise adds the later duplicates in order and writes the row with a plain store (mode 0) or a plain read-modify-write
// (mode 1: the wave owns the row in this launch).  The scan is quadratic in the batch (n^2 * 8 B of L2 reads: 2 MB for the
// reference's 2 x 256 slots), which is what a validation mode can afford; slots == NULL clears the named rows instead.
constexpr int kReduceWaves = 4;

__global__ __launch_bounds__(kWave *kReduceWaves) void reduce_slots_kernel(
    const int64_t *__restrict__ idx_a, int n_a, int64_t off_a, const int64_t *__restrict__ idx_b, int n_b, int64_t off_b,
    int n_rows, const float *__restrict__ slots, int ld_slots, float scale, float *out, int mode)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int k = blockIdx.x * kReduceWaves + (threadIdx.x >> 6);
    const int n = n_a + n_b;
    if (k >= n) return;
    const long long r = batch_row(idx_a, n_a, off_a, idx_b, off_b, k);
    if (r < 0 || r >= n_rows) return;
    if (!slots) {                                              // clear form: every slot zeroes its row (idempotent)
        out[(size_t)r * kWave + lane] = 0.0f;
        return;
    }
    for (int base = 0; base < k; base += kWave) {              // a lower-numbered slot with this row owns it
        const int kk = base + lane;
        const bool match = kk < k && batch_row(idx_a, n_a, off_a, idx_b, off_b, kk) == r;
        if (__ballot(match)) return;
    }
    float acc = slots[(size_t)k * ld_slots + lane];
    for (int base = k & ~(kWave - 1); base < n; base += kWave) {
        const int kk = base + lane;
        const bool match = kk > k && kk < n && batch_row(idx_a, n_a, off_a, idx_b, off_b, kk) == r;
        unsigned long long m = __ballot(match);
        while (m) {                                            // ascending slot order
            const int j = (int)__builtin_ctzll(m);
            m &= m - 1;
            acc = acc + slots[(size_t)(base + j) * ld_slots + lane];
        }
    }
    if (scale != 1.0f) acc = acc * scale;
    float *o = out + (size_t)r * kWave + lane;
    *o = mode == 1 ? *o + acc : acc;
}

// out[0] (+)= x[0] + x[1] + ... + x[n - 1], added in index order by one thread-free wave pattern: lane j sums x[j], x[j + 64], ...
// and the 64 partials are combined by the fixed DPP tree — the same order adam_kernel uses for a step's per-sample losses.
__global__ __launch_bounds__(kWave) void sum_ordered_kernel(const float *__restrict__ x, int n, float scale, float *out, int accumulate)
{
    float t = 0.0f;
    for (int i = threadIdx.x; i < n; i += kWave) t += x[i];
    t = wave_sum_f32(t);
    if (threadIdx.x == 0) *out = accumulate ? *out + t * scale : t * scale;
}

// Sum of n_parts partial blocks of n floats, in part order (deterministic): out[i] (+)= sum_k parts[k][i].
__global__ __launch_bounds__(256) void sum_parts_kernel(const float *__restrict__ parts, int n_parts, int64_t stride, int n, float *out,
                                                        int accumulate)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g = 0.0f;
    for (int k = 0; k < n_parts; ++k) g += parts[(size_t)k * stride + i];
    out[i] = accumulate ? out[i] + g : g;
}

}  // namespace

extern "C" int spex_reduce_slots_f32(const int64_t *idx_a, int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b, int64_t off_b,
                                     int32_t n_rows, const float *slots, int32_t ld_slots, float scale, float *out, int32_t mode,
                                     int32_t d, void *stream)
{
    SPEX_CHECK_ARG(n_a >= 0 && n_b >= 0 && (n_a == 0 || idx_a) && (n_b == 0 || idx_b), "spex_reduce_slots_f32: bad index lists");
    SPEX_CHECK_ARG(out && n_rows >= 0 && (mode == 0 || mode == 1) && (!slots || ld_slots >= d), "spex_reduce_slots_f32: out=%p mode=%d ld=%d",
                   (void *)out, mode, ld_slots);
    if (d != kWave) {
        spex::set_error("spex_reduce_slots_f32: d == 64 only (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    const int n = n_a + n_b;
    if (n == 0 || n_rows == 0) return SPEX_OK;
    hipLaunchKernelGGL(reduce_slots_kernel, dim3((unsigned)((n + kReduceWaves - 1) / kReduceWaves)), dim3(kWave * kReduceWaves), 0,
                       (hipStream_t)stream, idx_a, n_a, off_a, idx_b, n_b, off_b, n_rows, slots, ld_slots, scale, out, mode);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

int spex::sum_ordered(const float *x, int32_t n, float scale, float *out, int accumulate, void *stream)
{
    hipLaunchKernelGGL(sum_ordered_kernel, dim3(1), dim3(kWave), 0, (hipStream_t)stream, x, n, scale, out, accumulate);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

int spex::sum_parts(const float *parts, int32_t n_parts, int64_t stride, int32_t n, float *out, int accumulate, void *stream)
{
    if (n <= 0) return SPEX_OK;
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, parts, n_parts, stride, n, out,
                       accumulate);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_unique_rows_i32(const int64_t *idx_a, int32_t n_a, int64_t off_a, const int64_t *idx_b, int32_t n_b,
                                    int64_t off_b, int32_t n_rows, int32_t *stamp, int32_t epoch, int32_t *list, int32_t *count,
                                    void *stream)
{
    SPEX_CHECK_ARG(n_a >= 0 && n_b >= 0 && (n_a == 0 || idx_a) && (n_b == 0 || idx_b), "spex_unique_rows_i32: bad index lists");
    SPEX_CHECK_ARG(stamp && list && count && n_rows >= 0 && epoch != 0, "spex_unique_rows_i32: NULL buffer or epoch 0");
    SPEX_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), (hipStream_t)stream));
    if (n_a + n_b == 0) return SPEX_OK;
    hipLaunchKernelGGL(unique_rows_kernel, dim3((unsigned)((n_a + n_b + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx_a, n_a,
                       off_a, idx_b, n_b, off_b, n_rows, stamp, epoch, list, count);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_spmm_push_rows_f32(const spex_graph_t *g, const int32_t *list, const int32_t *count, int32_t max_count,
                                       const float *src, int32_t src_indexed, const float *add, int32_t add_indexed, float scale,
                                       float *out, int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g && list && count && src && out, "spex_spmm_push_rows_f32: NULL argument");
    SPEX_CHECK_ARG(max_count >= 0 && d >= 1, "spex_spmm_push_rows_f32: max_count=%d d=%d", max_count, d);
    SPEX_CHECK_ARG(g->mask_mode == 0, "spex_spmm_push_rows_f32: edge dropout is not supported in push form");
    if (max_count == 0 || g->n_rows == 0) return SPEX_OK;
    hipLaunchKernelGGL(spmm_push_rows_kernel, dim3((unsigned)max_count), dim3(kWave * kWgWaves), 0, (hipStream_t)stream, g->rowptr,
                       g->col, g->val, list, count, g->n_rows, src, src_indexed, add, add_indexed, scale, out, d);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_spmm_push_batch_f32(const spex_graph_t *g, const int64_t *idx_a, int32_t n_a, int64_t off_a, const int64_t *idx_b,
                                        int32_t n_b, int64_t off_b, const float *src, int32_t ld_src, const float *add, int32_t ld_add,
                                        float scale, float *out, int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g && src && out, "spex_spmm_push_batch_f32: NULL argument");
    SPEX_CHECK_ARG(n_a >= 0 && n_b >= 0 && (n_a == 0 || idx_a) && (n_b == 0 || idx_b), "spex_spmm_push_batch_f32: bad index lists");
    SPEX_CHECK_ARG(g->mask_mode == 0, "spex_spmm_push_batch_f32: edge dropout is not supported in push form");
    if (d != kWave) {
        spex::set_error("spex_spmm_push_batch_f32: d == 64 only (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG(ld_src >= d && (!add || ld_add >= d), "spex_spmm_push_batch_f32: ld_src=%d ld_add=%d", ld_src, ld_add);
    if (n_a + n_b == 0 || g->n_rows == 0) return SPEX_OK;
    hipLaunchKernelGGL(spmm_push_batch_kernel, dim3((unsigned)(n_a + n_b) * kPushParts), dim3(kWave * kPushWaves), 0, (hipStream_t)stream,
                       g->rowptr, g->col, g->val, g->n_rows, idx_a, n_a, off_a, idx_b, n_b, off_b, src, ld_src, add, ld_add, scale, out);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

// x[0 : n] = 0 as a kernel launch (hipMemsetAsync costs the host ~10 us a call on this runtime; a launch ~1-2).
namespace {
__global__ __launch_bounds__(256) void zero_kernel(float *__restrict__ x, int64_t n4, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<float4 *>(x)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) x[i] = 0.0f;
}
}  // namespace

int spex::zero_f32(float *x, int64_t n, void *stream)
{
    if (n <= 0) return SPEX_OK;
    if ((((uintptr_t)x) & 15) != 0) {
        SPEX_HIP(hipMemsetAsync(x, 0, (size_t)n * sizeof(float), (hipStream_t)stream));
        return SPEX_OK;
    }
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n4, n);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
