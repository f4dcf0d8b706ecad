// Negative sampler on the GPU — SURVEY.md 8f "next" #2.
//
// Replaces LightTrainData.ng_sample, LightGCN_SPEX/code/utility1/dataloader.py:250-265: for every positive (u, i) draw
// num_ng items j uniformly from [0, num_item), redrawing while (u, j) is a training interaction.  The reference walks
// 1 M draws in a Python loop against a dok matrix (~27 s per Epinion2 epoch).  Here one thread owns one slot: a
// counter-based draw (philox4x32-10 keyed by the seed, counter = (slot, attempt)) mapped to [0, num_item) by a
// multiply-high, and a binary search in the user's sorted item list (CSR of R).  The distribution is the reference's
// (uniform over the user's non-interacted items); the stream is not NumPy's Mersenne Twister, so for the same seed the
// individual negatives differ — the drop-in Loader keeps its exact-replay host sampler as the default and offers this
// one as `ng_sample(device=...)`.
#include "spex_common.h"

namespace {

__device__ __forceinline__ uint32_t philox2(uint32_t c0_, uint32_t c1_, uint32_t k0, uint32_t k1)
{
    uint32_t c0 = c0_, c1 = c1_, c2 = 0u, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0;
}

__global__ __launch_bounds__(256) void sample_negatives_kernel(const int32_t *__restrict__ rowptr,
                                                               const int32_t *__restrict__ items,
                                                               const int64_t *__restrict__ pos_user, int64_t n_slots,
                                                               int num_ng, int num_item, int n_user_rows, uint32_t seed_lo,
                                                               uint32_t seed_hi, int64_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; slot < n_slots; slot += stride) {
        const int64_t u = pos_user[slot / num_ng];
        int beg = 0, end = 0;
        if (u >= 0 && u < n_user_rows) {
            beg = rowptr[u];
            end = rowptr[u + 1];
        }
        int j = 0;
        bool done = false;
        // Rejection sampling, bounded: a user who has interacted with most of the catalogue would redraw for a long
        // time (for ever, in the reference, if with all of it).  After 16 rejections fall back to drawing the k-th
        // admissible item directly (same uniform law, one more binary search).
        for (uint32_t attempt = 0; attempt < 16u; ++attempt) {
            const uint32_t x = philox2((uint32_t)slot, ((uint32_t)(slot >> 32) << 8) | attempt, seed_lo, seed_hi);
            j = (int)(((uint64_t)x * (uint64_t)(uint32_t)num_item) >> 32);
            int lo = beg, hi = end;  // binary search for j in items[beg, end)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (items[mid] < j) lo = mid + 1;
                else hi = mid;
            }
            if (!(lo < end && items[lo] == j)) {
                done = true;
                break;
            }
        }
        if (!done) {
            const int deg = end - beg, admissible = num_item - deg;
            if (admissible > 0) {
                const uint32_t x = philox2((uint32_t)slot, ((uint32_t)(slot >> 32) << 8) | 255u, seed_lo, seed_hi);
                const int k = (int)(((uint64_t)x * (uint64_t)(uint32_t)admissible) >> 32);   // k-th admissible item
                int lo = 0, hi = deg;  // first position m with items[beg+m] - m > k  (admissible items below it)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (items[beg + mid] - mid > k) hi = mid;
                    else lo = mid + 1;
                }
                j = k + lo;
            } else {
                j = 0;  // the user has every item: no valid negative exists (the reference would not terminate)
            }
        }
        out[slot] = j;
    }
}

}  // namespace

extern "C" int spex_sample_negatives(const int32_t *d_rowptr, const int32_t *d_items, int32_t n_user_rows,
                                     const int64_t *d_pos_user, int64_t n_pos, int32_t num_ng, int32_t num_item,
                                     uint64_t seed, int64_t *d_out, void *stream)
{
    SPEX_CHECK_ARG(d_rowptr && d_pos_user && d_out, "spex_sample_negatives: NULL pointer");
    SPEX_CHECK_ARG(n_pos >= 0 && num_ng >= 1 && num_item >= 1 && n_user_rows >= 0, "spex_sample_negatives: bad sizes");
    const int64_t n_slots = n_pos * num_ng;
    if (n_slots == 0) return SPEX_OK;
    int64_t blocks = (n_slots + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(sample_negatives_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_rowptr, d_items,
                       d_pos_user, n_slots, num_ng, num_item, n_user_rows, (uint32_t)seed, (uint32_t)(seed >> 32), d_out);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
