// Negative-sampled scoring kernels: gather + dot + BCE-with-logits (reference semantics) and BPR (north-star
// extension), forward and backward fused, one wavefront per sample / triple.
//
// Replaces LightGCN.forward, LightGCN_SPEX/code/utility1/model.py:111-121, and NGCF compute_rec_loss,
// NGCF_SPEX/code/main_rec.py:89-100, together with their autograd backward into the gathered tables.
//
// A sample touches two (BPR: three) 256-byte rows and writes as many gradient rows: ~1 KB (1.5 KB) of traffic for 128
// (192) flops — HBM/L2-bound gather/scatter.  Lane == embedding column: each row access is one coalesced 256-byte
// wave load, the dot product is a 6-step cross-lane butterfly, the gradient rows are added with hardware float
// atomics shaped as one contiguous 256-byte wave instruction per row (the full-rate shape on MI355X; duplicates of a
// user within a batch accumulate correctly).
#include "spex_common.h"

using namespace spex;

namespace {

// 1024-thread workgroups: the loss is reduced per workgroup and added with ONE atomic to a single address; those
// same-address atomics serialise in the L2 (~5 ns each), so 4-wave workgroups made a 16 k-sample launch 2.5x slower
// (35 vs 14 us) than the same launch without a loss.
constexpr int kScoreWaves = 16;

__device__ __forceinline__ float wave_sum(float v)
{
    return wave_sum_f32(v);
}

__device__ __forceinline__ float lane_bcast_f(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Per-block loss reduction: lane 0 of each wave holds a partial; one atomic per block.
__device__ __forceinline__ void block_loss_add(float wave_partial, float *loss_sum)
{
    __shared__ float s_part[kScoreWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (lane == 0) s_part[wave] = wave_partial;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < kScoreWaves; ++w) t += s_part[w];
        atomicAdd(loss_sum, t);
    }
}

__global__ __launch_bounds__(kWave *kScoreWaves) void score_bce_kernel(
    const float *__restrict__ users, const float *__restrict__ items, int ldu, int ldi, const int64_t *__restrict__ u_idx,
    const int64_t *__restrict__ i_idx, const float *__restrict__ labels, int B, int d, int64_t n_user_rows,
    int64_t n_item_rows, float *__restrict__ gamma, float *loss_sum, float *grad_users, float *grad_items,
    float grad_scale, float *__restrict__ grad_slots, int ld_slots, float *__restrict__ loss_rows)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_global = blockIdx.x * kScoreWaves + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * kScoreWaves;
    float lsum = 0.0f;
    for (int b = wave_global; b < B; b += n_waves) {
        const int64_t u = u_idx[b], it = i_idx[b];
        if (u < 0 || u >= n_user_rows || it < 0 || it >= n_item_rows) {  // never gather out of bounds
            if (lane == 0 && gamma) gamma[b] = __int_as_float(0x7fc00000);
            if (lane == 0 && loss_rows) loss_rows[b] = 0.0f;
            if (grad_slots)
                for (int c = lane; c < d; c += kWave) grad_slots[(size_t)b * ld_slots + c] = grad_slots[(size_t)(B + b) * ld_slots + c] = 0.0f;
            continue;
        }
        const float *pu = users + (size_t)u * ldu, *pi = items + (size_t)it * ldi;
        float s = 0.0f;
        for (int c = lane; c < d; c += kWave) s = fmaf(pu[c], pi[c], s);
        const float x = wave_sum(s);
        if (lane == 0 && gamma) gamma[b] = x;
        if (labels) {
            const float y = labels[b];
            const float bce = fmaxf(x, 0.0f) - x * y + log1pf(expf(-fabsf(x)));
            if (loss_rows) {                                  // per-sample losses, summed in sample order by the caller
                if (lane == 0) loss_rows[b] = bce;
            } else {
                lsum += bce;
            }
            if (grad_slots) {     // the sample's two gradient rows, compact: slot b = user side, slot B + b = item side
                const float dg = (sigmoid_f(x) - y) * grad_scale;
                for (int c = lane; c < d; c += kWave) {
                    grad_slots[(size_t)b * ld_slots + c] = dg * pi[c];
                    grad_slots[(size_t)(B + b) * ld_slots + c] = dg * pu[c];
                }
            }
            if (grad_users) {
                const float dg = (sigmoid_f(x) - y) * grad_scale;
                float *gu = grad_users + (size_t)u * ldu, *gi = grad_items + (size_t)it * ldi;
                for (int c = lane; c < d; c += kWave) {
                    atomicAdd(gu + c, dg * pi[c]);
                    atomicAdd(gi + c, dg * pu[c]);
                }
            }
        }
    }
    if (labels && loss_sum && !loss_rows) block_loss_add(lsum, loss_sum);
}

// a_coef: multiplies sigmoid(x) (SGD: -lr/T, autograd: grad_scale); b_coef: multiplies the read row (SGD: -lr*reg/T).
// A wave owns `per_wave` CONSECUTIVE triples and keeps the running update of the current user row and of the current
// positive-item row in registers, flushing one with a single 256-byte atomic instruction only when the index changes.
// Triples arrive in any order — nothing is assumed — but the sampler's natural order (num_ng negatives per positive,
// positives sorted by user: dataloader.py:250-265) then costs ~1.3 atomic row updates per triple instead of 3, and the
// kernel is bound by the float-atomic rate (~1.3 TB/s chip-wide), not by HBM.  per_wave == 1 for small batches (latency).
// Measured at T ~ 1 M on Epinion2's tables: 1.5-1.6 G triples/s in random order, 2.25 G in sampler order (hot rows then
// serialise in the L2: a user's ~330 consecutive triples are flushed by ~20 waves at about the same time); fetching a
// run's rows ahead of use (48 loads in flight, 98 VGPRs) changed neither figure.
// The same update with the propagated rows formed ON THE FLY from up to three tables — row r = ((t0[r] + t1[r]) + t2[r]) / div,
// the layer mean of utility1/model.py:94-95 in the order the SpMM's fused epilogues form it — so that the propagation in front
// of it can run without its layer-mean epilogue (spex::propagate_plain): the mean is needed at the batch's <= 3 T rows only.
// Tables hold users first, items from row n_user_rows on; table_w (= E^0) receives the updates and is none of t0 / t1 / t2, so
// the step stays batch-synchronous.  d == 64, one triple per wave (the step's batches are small: latency-bound by design).
__global__ __launch_bounds__(kWave *kScoreWaves) void bpr_layers_kernel(
    const float *__restrict__ t0, const float *__restrict__ t1, const float *__restrict__ t2, float div, float *table_w,
    const int64_t *__restrict__ u_idx, const int64_t *__restrict__ p_idx, const int64_t *__restrict__ n_idx, int64_t T,
    int64_t n_user_rows, int64_t n_item_rows, float a_coef, float b_coef, float *loss_sum)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave_global = (int64_t)blockIdx.x * kScoreWaves + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * kScoreWaves;
    float lsum = 0.0f;
    for (int64_t t = wave_global; t < T; t += n_waves) {
        const int64_t u = u_idx[t], ip = p_idx[t], in = n_idx[t];
        if (u < 0 || u >= n_user_rows || ip < 0 || ip >= n_item_rows || in < 0 || in >= n_item_rows) continue;
        const size_t ou = (size_t)u * kWave + lane, op = (size_t)(n_user_rows + ip) * kWave + lane, on = (size_t)(n_user_rows + in) * kWave + lane;
        float uu = t0[ou], vp = t0[op], vn = t0[on];              // every load of the triple in flight before the first add
        float u1 = 0.0f, p1 = 0.0f, n1 = 0.0f, u2 = 0.0f, p2 = 0.0f, n2 = 0.0f;
        if (t1) { u1 = t1[ou]; p1 = t1[op]; n1 = t1[on]; }
        if (t2) { u2 = t2[ou]; p2 = t2[op]; n2 = t2[on]; }
        if (t1) { uu = uu + u1; vp = vp + p1; vn = vn + n1; }
        if (t2) { uu = uu + u2; vp = vp + p2; vn = vn + n2; }
        if (div != 1.0f) { uu = uu / div; vp = vp / div; vn = vn / div; }
        const float x = wave_sum(uu * vn) - wave_sum(uu * vp);   // neg_score - pos_score
        lsum += softplus_f(x);
        const float a = a_coef * sigmoid_f(x);
        atomicAdd(table_w + ou, a * (vn - vp) + b_coef * uu);
        atomicAdd(table_w + op, -a * uu + b_coef * vp);
        atomicAdd(table_w + on, a * uu + b_coef * vn);
    }
    if (loss_sum) block_loss_add(lsum, loss_sum);
}

__global__ __launch_bounds__(kWave *kScoreWaves) void bpr_kernel(
    const float *__restrict__ U_read, const float *__restrict__ I_read, float *U_w, float *I_w,
    const int64_t *__restrict__ u_idx, const int64_t *__restrict__ p_idx, const int64_t *__restrict__ n_idx, int64_t T,
    int d, int64_t n_user_rows, int64_t n_item_rows, float a_coef, float b_coef, float *loss_sum, int per_wave)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave_global = (int64_t)blockIdx.x * kScoreWaves + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * kScoreWaves;
    float lsum = 0.0f;
    if (d == kWave && U_w) {   // the tuned form: one column per lane, rows held in registers across triples
        for (int64_t t0 = wave_global * per_wave; t0 < T; t0 += n_waves * per_wave) {
            const int64_t t1 = (t0 + per_wave < T) ? t0 + per_wave : T;
            int64_t cu = -1, cp = -1;
            float uu = 0.0f, vp = 0.0f, acc_u = 0.0f, acc_p = 0.0f;
            for (int64_t t = t0; t < t1; ++t) {
                const int64_t u = u_idx[t], ip = p_idx[t], in = n_idx[t];
                if (u < 0 || u >= n_user_rows || ip < 0 || ip >= n_item_rows || in < 0 || in >= n_item_rows) continue;
                if (u != cu) {
                    if (cu >= 0) atomicAdd(U_w + (size_t)cu * kWave + lane, acc_u);
                    cu = u;
                    uu = U_read[(size_t)u * kWave + lane];
                    acc_u = 0.0f;
                }
                if (ip != cp) {
                    if (cp >= 0) atomicAdd(I_w + (size_t)cp * kWave + lane, acc_p);
                    cp = ip;
                    vp = I_read[(size_t)ip * kWave + lane];
                    acc_p = 0.0f;
                }
                const float vn = I_read[(size_t)in * kWave + lane];
                const float x = wave_sum(uu * vn) - wave_sum(uu * vp);   // neg_score - pos_score
                lsum += softplus_f(x);
                const float a = a_coef * sigmoid_f(x);
                acc_u += a * (vn - vp) + b_coef * uu;
                acc_p += -a * uu + b_coef * vp;
                atomicAdd(I_w + (size_t)in * kWave + lane, a * uu + b_coef * vn);
            }
            if (cu >= 0) atomicAdd(U_w + (size_t)cu * kWave + lane, acc_u);
            if (cp >= 0) atomicAdd(I_w + (size_t)cp * kWave + lane, acc_p);
        }
    } else {
        for (int64_t t = wave_global; t < T; t += n_waves) {
            const int64_t u = u_idx[t], ip = p_idx[t], in = n_idx[t];
            if (u < 0 || u >= n_user_rows || ip < 0 || ip >= n_item_rows || in < 0 || in >= n_item_rows) continue;
            const float *pu = U_read + (size_t)u * d, *pp = I_read + (size_t)ip * d, *pn = I_read + (size_t)in * d;
            float sp = 0.0f, sn = 0.0f;
            for (int c = lane; c < d; c += kWave) {
                const float uu = pu[c];
                sp = fmaf(uu, pp[c], sp);
                sn = fmaf(uu, pn[c], sn);
            }
            const float x = wave_sum(sn) - wave_sum(sp);  // neg_score - pos_score
            lsum += softplus_f(x);
            if (U_w) {
                const float a = a_coef * sigmoid_f(x);
                for (int c = lane; c < d; c += kWave) {
                    const float uu = pu[c], vp = pp[c], vn = pn[c];
                    atomicAdd(U_w + (size_t)u * d + c, a * (vn - vp) + b_coef * uu);
                    atomicAdd(I_w + (size_t)ip * d + c, -a * uu + b_coef * vp);
                    atomicAdd(I_w + (size_t)in * d + c, a * uu + b_coef * vn);
                }
            }
        }
    }
    if (loss_sum) block_loss_add(lsum, loss_sum);
}

// ---------------------------------------------------------------------------------------------------------------
// Grouped BPR update: the same arithmetic without a float atomic per gathered row.
//
// The atomic form above adds three 256-byte rows per triple with hardware float atomics and is bound by their chip-wide
// rate (~1.3 TB/s of added bytes: 0.32 of the HBM roofline at T = 2^20).  Here the batch's 3 T (triple, role) entries
// are sorted by DESTINATION row first — a two-phase counting partition into buckets of 64 rows (user rows and item rows
// in separate buckets), then a 64-bin counting sort inside the workgroup that owns a bucket slice — so that 64
// consecutive entries belong to one or two rows: a wave sums a run of equal rows in a register and adds it to the table
// with ONE 256-byte float atomic per run (~2 per 64 entries: ~1.5 % of the atomic form's atomic traffic).
// Sorting by destination also removes the separate scoring pass: for the entries whose destination is the USER row the
// wave holds U[u] for the whole run, gathers I[i-] and I[i+] (which it needs for the update anyway), and has the triple's
// score <U[u], I[i-] - I[i+]> for one DPP reduction; it stores coef_t = a sigmoid(x_t) for the two item-destination entries
// of the triple, which then cost one gather (U[u]) each.  Gathers per triple: 4 (the atomic form: 3, plus 3 atomic rows).
//
//   k_count    thread per triple: per-workgroup LDS histogram of the three destination buckets -> global counts
//   k_scan     one workgroup: exclusive scans of the counts (entry offsets) and of ceil(count / slice) (first workgroup
//              of each bucket); clears the cursors
//   k_scatter  thread per entry: per-workgroup LDS histogram again, ONE global cursor add per bucket and workgroup
//              reserves a range; entries (triple * 4 + role, row within the bucket) land bucket by bucket
//   k_users    workgroup per (user bucket, slice of <= 4096 entries): row sort in LDS, scores + loss + coef_t + user rows
//   k_items    the same for the item buckets, reading coef_t
// What was tried first and measured (T = 2^20, Epinion2's tables): accumulating a bucket in LDS with ds_add_f32 — one such
// wave-instruction costs ~200 cycles, 3 M of them 1.06 ms of a 1.34 ms kernel; and 64-lane dot products through
// __shfl_xor — six ds_bpermute_b32 each on the CU's single LDS pipe, 184 us for 2 M dot products with the gathers removed.
// Sums inside a run are accumulated in arrival order: like the atomic form, results agree with the closed form to fp32
// re-association (<= 1e-5 relative), not bit for bit.
constexpr int kBucketRows = 64;
constexpr int kSliceEntries = 2048;
constexpr int kMaxBuckets = 8192;      // LDS histogram of k_count / k_scatter: 32 KB
constexpr int kBinThreads = 1024;

struct BprGroupedArgs {
    const float *U_read, *I_read;
    float *U_w, *I_w;
    const int64_t *u, *ip, *in;
    int64_t T, n_user_rows, n_item_rows;
    float a_coef, b_coef;
    float *loss_sum;
    float *coef;          // [T]
    uint32_t *ent_key;    // [3T] triple * 4 + role (0: user row, 1: positive item row, 2: negative item row)
    uint32_t *ent_row;    // [3T] destination row within its bucket
    int32_t *count;       // [n_buckets]
    int32_t *offset;      // [n_buckets + 1] entry offsets
    int32_t *wg_first;    // [n_buckets + 1] first update workgroup of each bucket
    int32_t *cursor;      // [n_buckets]
    int32_t n_user_buckets, n_buckets;   // user buckets come first
};

__device__ __forceinline__ bool bpr_valid(const BprGroupedArgs &a, int64_t u, int64_t ip, int64_t in)
{
    return u >= 0 && u < a.n_user_rows && ip >= 0 && ip < a.n_item_rows && in >= 0 && in < a.n_item_rows;
}

__global__ __launch_bounds__(kBinThreads) void bpr_count_kernel(const BprGroupedArgs a)
{
    __shared__ int s_hist[kMaxBuckets];
    for (int b = threadIdx.x; b < a.n_buckets; b += blockDim.x) s_hist[b] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < a.T; t += stride) {
        const int64_t u = a.u[t], ip = a.ip[t], in = a.in[t];
        if (!bpr_valid(a, u, ip, in)) continue;
        atomicAdd(&s_hist[(int)(u / kBucketRows)], 1);
        atomicAdd(&s_hist[a.n_user_buckets + (int)(ip / kBucketRows)], 1);
        atomicAdd(&s_hist[a.n_user_buckets + (int)(in / kBucketRows)], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < a.n_buckets; b += blockDim.x)
        if (s_hist[b]) atomicAdd(a.count + b, s_hist[b]);
}

__global__ __launch_bounds__(1024) void bpr_bucket_scan_kernel(const BprGroupedArgs a)
{
    __shared__ int s_ent[1024], s_wg[1024];
    // each thread owns a contiguous run of buckets; two scans at once
    const int per = (a.n_buckets + 1023) / 1024;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, a.n_buckets);
    int e = 0, w = 0;
    for (int b = b0; b < b1; ++b) {
        const int c = a.count[b];
        e += c;
        w += (c + kSliceEntries - 1) / kSliceEntries;
    }
    s_ent[threadIdx.x] = e;
    s_wg[threadIdx.x] = w;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {       // Hillis-Steele inclusive scan over the 1024 partial sums
        int ve = 0, vw = 0;
        if ((int)threadIdx.x >= off) {
            ve = s_ent[threadIdx.x - off];
            vw = s_wg[threadIdx.x - off];
        }
        __syncthreads();
        s_ent[threadIdx.x] += ve;
        s_wg[threadIdx.x] += vw;
        __syncthreads();
    }
    e = s_ent[threadIdx.x] - e;                      // exclusive prefix of this thread's run
    w = s_wg[threadIdx.x] - w;
    for (int b = b0; b < b1; ++b) {
        const int c = a.count[b];
        a.offset[b] = e;
        a.wg_first[b] = w;
        a.cursor[b] = 0;
        e += c;
        w += (c + kSliceEntries - 1) / kSliceEntries;
    }
    if (threadIdx.x == 1023) {
        a.offset[a.n_buckets] = s_ent[1023];
        a.wg_first[a.n_buckets] = s_wg[1023];
    }
}

__global__ __launch_bounds__(kBinThreads) void bpr_bin_scatter_kernel(const BprGroupedArgs a)
{
    __shared__ int s_hist[kMaxBuckets];     // counts, then this workgroup's next slot per bucket
    for (int b = threadIdx.x; b < a.n_buckets; b += blockDim.x) s_hist[b] = 0;
    __syncthreads();
    // a workgroup owns a contiguous range of triples; a thread handles a triple's three entries
    const int64_t per_wg = (a.T + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * per_wg, t1 = min(t0 + per_wg, a.T);
    for (int64_t t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
        const int64_t u = a.u[t], ip = a.ip[t], in = a.in[t];
        if (!bpr_valid(a, u, ip, in)) continue;
        atomicAdd(&s_hist[(int)(u / kBucketRows)], 1);
        atomicAdd(&s_hist[a.n_user_buckets + (int)(ip / kBucketRows)], 1);
        atomicAdd(&s_hist[a.n_user_buckets + (int)(in / kBucketRows)], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < a.n_buckets; b += blockDim.x) {
        const int c = s_hist[b];
        s_hist[b] = c ? a.offset[b] + atomicAdd(a.cursor + b, c) : 0;   // reserve [base, base + c) in bucket b
    }
    __syncthreads();
    for (int64_t t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
        const int64_t u = a.u[t], ip = a.ip[t], in = a.in[t];
        if (!bpr_valid(a, u, ip, in)) continue;
        const uint32_t key = (uint32_t)t * 4u;
        int slot = atomicAdd(&s_hist[(int)(u / kBucketRows)], 1);
        a.ent_key[slot] = key;
        a.ent_row[slot] = (uint32_t)(u % kBucketRows);
        slot = atomicAdd(&s_hist[a.n_user_buckets + (int)(ip / kBucketRows)], 1);
        a.ent_key[slot] = key + 1u;
        a.ent_row[slot] = (uint32_t)(ip % kBucketRows);
        slot = atomicAdd(&s_hist[a.n_user_buckets + (int)(in / kBucketRows)], 1);
        a.ent_key[slot] = key + 2u;
        a.ent_row[slot] = (uint32_t)(in % kBucketRows);
    }
}

// USERS: the bucket holds user rows (entries of role 0); else item rows (roles 1 and 2).
template <bool USERS>
__global__ __launch_bounds__(kWave *kScoreWaves) void bpr_bucket_update_kernel(const BprGroupedArgs a)
{
    __shared__ uint32_t s_key[kSliceEntries], s_row[kSliceEntries];
    __shared__ int s_bin[kBucketRows], s_cur[kBucketRows];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int b_lo = USERS ? 0 : a.n_user_buckets, b_hi = USERS ? a.n_user_buckets : a.n_buckets;
    const int wg = blockIdx.x + a.wg_first[b_lo];
    float lsum = 0.0f;
    if (wg < a.wg_first[b_hi]) {                          // (workgroup-uniform; the loss reduction below needs every wave)
        int lo = b_lo, hi = b_hi;                         // which bucket: the last b with wg_first[b] <= wg
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.wg_first[mid] <= wg) lo = mid; else hi = mid;
        }
        const int bucket = lo;
        const int slice = wg - a.wg_first[bucket];
        const int beg = a.offset[bucket] + slice * kSliceEntries;
        const int end = min(beg + kSliceEntries, a.offset[bucket + 1]);
        const int n_ent = end - beg;
        const int64_t row0 = (int64_t)(bucket - b_lo) * kBucketRows;
        // ---- counting sort of the slice by destination row (64 bins), in LDS
        if (threadIdx.x < kBucketRows) s_bin[threadIdx.x] = 0;
        __syncthreads();
        constexpr int kPerThread = kSliceEntries / (kWave * kScoreWaves);      // 4
        uint32_t my_key[kPerThread], my_row[kPerThread];
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const int e = threadIdx.x + j * kWave * kScoreWaves;
            my_row[j] = 0xFFFFFFFFu;
            if (e < n_ent) {
                my_key[j] = a.ent_key[beg + e];
                my_row[j] = a.ent_row[beg + e];
                atomicAdd(&s_bin[my_row[j]], 1);
            }
        }
        __syncthreads();
        if (wave == 0) {                                  // exclusive scan of the 64 bins by one wave
            const int c = s_bin[lane];
            int incl = c;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const int v = __shfl_up(incl, off, kWave);
                if (lane >= off) incl += v;
            }
            s_cur[lane] = incl - c;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            if (my_row[j] != 0xFFFFFFFFu) {
                const int pos = atomicAdd(&s_cur[my_row[j]], 1);
                s_key[pos] = my_key[j];
                s_row[pos] = my_row[j];
            }
        }
        __syncthreads();
        // ---- a wave takes 64 consecutive sorted entries at a time: lane k resolves entry k, the wave walks them with
        //      v_readlane, a batch of gathers in flight at a time
        float *__restrict__ dst_tab = USERS ? a.U_w : a.I_w;
        const float *__restrict__ Ul = a.U_read + lane, *__restrict__ Il = a.I_read + lane;
        for (int base = wave * kWave; base < n_ent; base += kScoreWaves * kWave) {
            const int n = min(kWave, n_ent - base);
            int m_t = 0, m_r = 0, m_a = 0, m_b = 0;       // triple, destination row in the bucket, source rows
            float m_c = 0.0f;
            if (lane < n) {
                const uint32_t key = s_key[base + lane];
                m_t = (int)(key >> 2);
                m_r = (int)s_row[base + lane];
                if (USERS) {
                    m_a = (int)a.in[m_t];                 // v = c (I[i-] - I[i+]),  c = a sigmoid(<U[u], I[i-] - I[i+]>)
                    m_b = (int)a.ip[m_t];
                } else {
                    const float c = a.coef[m_t];          // v = -c U[u] (positive item) / +c U[u] (negative item)
                    m_c = (key & 3u) == 1u ? -c : c;
                    m_a = (int)a.u[m_t];
                }
            }
            // walk the chunk run by run (a run = consecutive entries with the same destination row)
            for (int i = 0; i < n;) {
                const int cur = __builtin_amdgcn_readlane(m_r, i);
                const unsigned long long other = __ballot(lane > i && lane < n && m_r != cur);
                const int run_end = other ? (int)__builtin_ctzll(other) : n;
                float own = 0.0f;                           // the destination's own row: scores (users) / regularisation
                if (USERS) own = Ul[(size_t)(row0 + cur) * kWave];
                else if (a.b_coef != 0.0f) own = Il[(size_t)(row0 + cur) * kWave];
                float acc = a.b_coef != 0.0f ? a.b_coef * own * (float)(run_end - i) : 0.0f;
                constexpr int kBatch = USERS ? 8 : 16;
                for (; i < run_end; i += kBatch) {
                    const int nb = min(kBatch, run_end - i);
                    float xa[kBatch];
#pragma unroll
                    for (int j = 0; j < kBatch; ++j) {
                        const int k = j < nb ? i + j : i + nb - 1;  // a ragged batch re-reads its last entry (unused)
                        const uint32_t ra = (uint32_t)__builtin_amdgcn_readlane(m_a, k);
                        if (USERS) {
                            xa[j] = Il[(size_t)ra * kWave] - Il[(size_t)(uint32_t)__builtin_amdgcn_readlane(m_b, k) * kWave];
                        } else {
                            xa[j] = Ul[(size_t)ra * kWave];
                        }
                    }
                    if (USERS) {
                        // scores of the batch, entry j's into lane j; then sigmoid / softplus for all of them at once
                        float x_mine = 0.0f;
#pragma unroll
                        for (int j = 0; j < kBatch; ++j) {
                            const float x = wave_sum(own * xa[j]);           // <U[u], I[i-] - I[i+]> = neg score - pos score
                            if (lane == j) x_mine = x;
                        }
                        float c_mine = 0.0f;
                        const int t_mine = __shfl(m_t, (i + lane) & (kWave - 1), kWave);   // (every lane active: a bpermute
                        if (lane < nb) {                                                    //  reads 0 from inactive lanes)
                            c_mine = a.a_coef * sigmoid_f(x_mine);
                            lsum += softplus_f(x_mine);
                            a.coef[t_mine] = c_mine;
                        }
#pragma unroll
                        for (int j = 0; j < kBatch; ++j) acc += lane_bcast_f(c_mine, j) * xa[j];   // (lanes >= nb hold 0)
                    } else {
#pragma unroll
                        for (int j = 0; j < kBatch; ++j)
                            if (j < nb) acc += lane_bcast_f(m_c, i + j) * xa[j];
                    }
                }
                i = run_end;
                atomicAdd(dst_tab + (size_t)(row0 + cur) * kWave + lane, acc);
            }
        }
    }
    if (USERS && a.loss_sum) {
        lsum = wave_sum(lsum);
        block_loss_add(lsum, a.loss_sum);
    }
}

// consecutive triples per wave: 1 until the batch fills the chip a few times over, then up to 16
inline int bpr_per_wave(int64_t T)
{
    int64_t k = T / (256 * 64);
    return (int)(k < 1 ? 1 : (k > 16 ? 16 : k));
}

// Owner-computes exchange of a batch's rows on a row-partitioned table (spex_amd/dist.py): every rank contributes the
// rows it owns to a zero-filled buffer (summed by one small all-reduce), and after scoring adds the update rows it
// owns back into its shard.  One wave per listed position; `pos` are positions in the padded global layout, the rank
// owns [lo, lo + n_local).
__global__ __launch_bounds__(kWave *kScoreWaves) void gather_owned_rows_kernel(const float *__restrict__ table,
                                                                              const int64_t *__restrict__ pos, int64_t K,
                                                                              int64_t lo, int64_t n_local, int d,
                                                                              float *__restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    for (int64_t k = (int64_t)blockIdx.x * kScoreWaves + (threadIdx.x >> 6); k < K; k += (int64_t)gridDim.x * kScoreWaves) {
        const int64_t p = pos[k] - lo;
        const bool own = p >= 0 && p < n_local;
        for (int c = lane; c < d; c += kWave) out[(size_t)k * d + c] = own ? table[(size_t)p * d + c] : 0.0f;
    }
}

__global__ __launch_bounds__(kWave *kScoreWaves) void scatter_add_owned_rows_kernel(float *upd, const int64_t *__restrict__ pos,
                                                                                   int64_t K, int64_t lo, int64_t n_local,
                                                                                   int d, float *table, int clear)
{
    const int lane = threadIdx.x & (kWave - 1);
    for (int64_t k = (int64_t)blockIdx.x * kScoreWaves + (threadIdx.x >> 6); k < K; k += (int64_t)gridDim.x * kScoreWaves) {
        const int64_t p = pos[k] - lo;
        const bool own = p >= 0 && p < n_local;
        for (int c = lane; c < d; c += kWave) {
            if (own) atomicAdd(table + (size_t)p * d + c, upd[(size_t)k * d + c]);   // positions repeat within a batch
            if (clear) upd[(size_t)k * d + c] = 0.0f;
        }
    }
}

inline unsigned grid_for(int64_t waves_wanted)
{
    int64_t blocks = (waves_wanted + kScoreWaves - 1) / kScoreWaves;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;  // 8 blocks per CU, grid-stride beyond
    return (unsigned)blocks;
}

}  // namespace

static int launch_score_bce(const float *users, const float *items, int32_t ldu, int32_t ldi, int64_t n_user_rows,
                            int64_t n_item_rows, const int64_t *u_idx, const int64_t *i_idx, const float *labels, int32_t B,
                            int32_t d, float *gamma, float *loss_sum, float *grad_users, float *grad_items, float grad_scale,
                            float *grad_slots, int32_t ld_slots, void *stream, const char *who, float *loss_rows = nullptr)
{
    SPEX_CHECK_ARG(users && items && u_idx && i_idx, "%s: NULL table or index pointer", who);
    SPEX_CHECK_ARG(B >= 0 && d >= 1 && ldu >= d && ldi >= d, "%s: B=%d d=%d ldu=%d ldi=%d", who, B, d, ldu, ldi);
    SPEX_CHECK_ARG(n_user_rows >= 0 && n_item_rows >= 0, "%s: negative table size", who);
    SPEX_CHECK_ARG((grad_users == nullptr) == (grad_items == nullptr), "%s: give both grad tables or neither", who);
    SPEX_CHECK_ARG((!grad_users && !grad_slots) || labels, "%s: gradients need labels", who);
    SPEX_CHECK_ARG(!grad_slots || ld_slots >= d, "%s: ld_slots=%d < d", who, ld_slots);
    SPEX_CHECK_ARG(gamma || labels, "%s: nothing to compute", who);
    if (B == 0) return SPEX_OK;
    hipLaunchKernelGGL(score_bce_kernel, dim3(grid_for(B)), dim3(kWave * kScoreWaves), 0, (hipStream_t)stream, users,
                       items, ldu, ldi, u_idx, i_idx, labels, B, d, n_user_rows, n_item_rows, gamma, loss_sum, grad_users,
                       grad_items, grad_scale, grad_slots, ld_slots, loss_rows);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

// Internal form for the deterministic steps: per-sample gradient rows AND per-sample losses (plain stores, no atomics at all).
int spex::score_bce_slots_rows(const float *users, const float *items, int32_t ldu, int32_t ldi, int64_t n_user_rows,
                               int64_t n_item_rows, const int64_t *u_idx, const int64_t *i_idx, const float *labels, int32_t B, int32_t d,
                               float *loss_rows, float grad_scale, float *grad_slots, int32_t ld_slots, void *stream)
{
    SPEX_CHECK_ARG(loss_rows && grad_slots && labels, "score_bce_slots_rows: NULL pointer");
    return launch_score_bce(users, items, ldu, ldi, n_user_rows, n_item_rows, u_idx, i_idx, labels, B, d, nullptr, nullptr, nullptr, nullptr,
                            grad_scale, grad_slots, ld_slots, stream, "score_bce_slots_rows", loss_rows);
}

extern "C" int spex_score_bce_f32(const float *users, const float *items, int32_t ldu, int32_t ldi,
                                  int64_t n_user_rows, int64_t n_item_rows, const int64_t *u_idx, const int64_t *i_idx,
                                  const float *labels, int32_t B, int32_t d, float *gamma, float *loss_sum,
                                  float *grad_users, float *grad_items, float grad_scale, void *stream)
{
    return launch_score_bce(users, items, ldu, ldi, n_user_rows, n_item_rows, u_idx, i_idx, labels, B, d, gamma, loss_sum, grad_users,
                            grad_items, grad_scale, nullptr, 0, stream, "spex_score_bce_f32");
}

extern "C" int spex_score_bce_slots_f32(const float *users, const float *items, int32_t ldu, int32_t ldi,
                                        int64_t n_user_rows, int64_t n_item_rows, const int64_t *u_idx, const int64_t *i_idx,
                                        const float *labels, int32_t B, int32_t d, float *loss_sum, float *grad_users,
                                        float *grad_items, float grad_scale, float *grad_slots, int32_t ld_slots, void *stream)
{
    SPEX_CHECK_ARG(grad_slots && labels, "spex_score_bce_slots_f32: needs grad_slots and labels");
    return launch_score_bce(users, items, ldu, ldi, n_user_rows, n_item_rows, u_idx, i_idx, labels, B, d, nullptr, loss_sum, grad_users,
                            grad_items, grad_scale, grad_slots, ld_slots, stream, "spex_score_bce_slots_f32");
}

extern "C" int spex_bpr_sgd_step_f32(const float *U_read, const float *I_read, float *U_w, float *I_w,
                                     int64_t n_user_rows, int64_t n_item_rows, const int64_t *u, const int64_t *i_pos,
                                     const int64_t *i_neg, int64_t T, int32_t d, float lr, float reg, float *loss_sum,
                                     void *stream)
{
    SPEX_CHECK_ARG(U_read && I_read && U_w && I_w && u && i_pos && i_neg, "spex_bpr_sgd_step_f32: NULL pointer");
    SPEX_CHECK_ARG(T >= 0 && d >= 1, "spex_bpr_sgd_step_f32: T=%lld d=%d", (long long)T, d);
    if (T == 0) return SPEX_OK;
    const int per_wave = bpr_per_wave(T);
    hipLaunchKernelGGL(bpr_kernel, dim3(grid_for((T + per_wave - 1) / per_wave)), dim3(kWave * kScoreWaves), 0,
                       (hipStream_t)stream, U_read, I_read, U_w, I_w, u, i_pos, i_neg, T, d, n_user_rows, n_item_rows,
                       -lr / (float)T, -lr * reg / (float)T, loss_sum, per_wave);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

int spex::bpr_sgd_layers(const float *t0, const float *t1, const float *t2, float div, float *table_w, int64_t n_user_rows,
                         int64_t n_item_rows, const int64_t *u, const int64_t *i_pos, const int64_t *i_neg, int64_t T, float lr, float reg,
                         float *loss_sum, void *stream)
{
    SPEX_CHECK_ARG(t0 && table_w && u && i_pos && i_neg && T >= 0 && div != 0.0f, "bpr_sgd_layers: bad argument");
    SPEX_CHECK_ARG(table_w != t0 && table_w != t1 && table_w != t2, "bpr_sgd_layers: the updated table must not be a read table");
    if (T == 0) return SPEX_OK;
    hipLaunchKernelGGL(bpr_layers_kernel, dim3(grid_for(T)), dim3(kWave * kScoreWaves), 0, (hipStream_t)stream, t0, t1, t2, div, table_w, u,
                       i_pos, i_neg, T, n_user_rows, n_item_rows, -lr / (float)T, -lr * reg / (float)T, loss_sum);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int64_t spex_bpr_grouped_workspace_bytes(int64_t T, int64_t n_user_rows, int64_t n_item_rows)
{
    if (T < 0 || n_user_rows < 0 || n_item_rows < 0) return -1;
    const int64_t n_buckets = (n_user_rows + kBucketRows - 1) / kBucketRows + (n_item_rows + kBucketRows - 1) / kBucketRows;
    if (n_buckets > kMaxBuckets || T >= ((int64_t)1 << 30)) return 0;      // not supported: use the atomic form
    // coef [T] | ent_key [3T] | ent_row [3T] | count, cursor [n_buckets] | offset, wg_first [n_buckets + 1], 256-B aligned parts
    auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
    return up(4 * T) + 2 * up(12 * T) + 2 * up(4 * n_buckets) + 2 * up(4 * (n_buckets + 1));
}

// Shared by the fused-SGD and the gradient form (same contract as bpr_kernel: a_coef multiplies sigmoid(x), b_coef the row).
static int launch_bpr_grouped(const float *U_read, const float *I_read, float *U_w, float *I_w, int64_t n_user_rows,
                              int64_t n_item_rows, const int64_t *u, const int64_t *i_pos, const int64_t *i_neg, int64_t T,
                              float a_coef, float b_coef, float *loss_sum, void *ws, int64_t ws_bytes, hipStream_t stream)
{
    const int64_t need = spex_bpr_grouped_workspace_bytes(T, n_user_rows, n_item_rows);
    if (need <= 0) {
        spex::set_error("grouped BPR: %lld rows / %lld triples are outside the grouped form's range (use the atomic form)",
                        (long long)(n_user_rows + n_item_rows), (long long)T);
        return SPEX_ERR_UNSUPPORTED;
    }
    SPEX_CHECK_ARG(ws && ws_bytes >= need && (((uintptr_t)ws) & 255) == 0, "grouped BPR: workspace of %lld bytes (256-B aligned) needed, got %lld",
                   (long long)need, (long long)ws_bytes);
    auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
    BprGroupedArgs a;
    a.U_read = U_read; a.I_read = I_read; a.U_w = U_w; a.I_w = I_w; a.u = u; a.ip = i_pos; a.in = i_neg;
    a.T = T; a.n_user_rows = n_user_rows; a.n_item_rows = n_item_rows; a.a_coef = a_coef; a.b_coef = b_coef; a.loss_sum = loss_sum;
    a.n_user_buckets = (int32_t)((n_user_rows + kBucketRows - 1) / kBucketRows);
    a.n_buckets = a.n_user_buckets + (int32_t)((n_item_rows + kBucketRows - 1) / kBucketRows);
    char *p = (char *)ws;
    a.coef = (float *)p; p += up(4 * T);
    a.ent_key = (uint32_t *)p; p += up(12 * T);
    a.ent_row = (uint32_t *)p; p += up(12 * T);
    a.count = (int32_t *)p; p += up(4 * (int64_t)a.n_buckets);
    a.cursor = (int32_t *)p; p += up(4 * (int64_t)a.n_buckets);
    a.offset = (int32_t *)p; p += up(4 * ((int64_t)a.n_buckets + 1));
    a.wg_first = (int32_t *)p;
    SPEX_HIP(hipMemsetAsync(a.count, 0, 4 * (size_t)a.n_buckets, stream));
    int64_t bin_blocks = (T + (int64_t)kBinThreads * 4 - 1) / ((int64_t)kBinThreads * 4);
    if (bin_blocks > 1024) bin_blocks = 1024;
    if (bin_blocks < 1) bin_blocks = 1;
    hipLaunchKernelGGL(bpr_count_kernel, dim3((unsigned)bin_blocks), dim3(kBinThreads), 0, stream, a);
    hipLaunchKernelGGL(bpr_bucket_scan_kernel, dim3(1), dim3(1024), 0, stream, a);
    hipLaunchKernelGGL(bpr_bin_scatter_kernel, dim3((unsigned)bin_blocks), dim3(kBinThreads), 0, stream, a);
    // upper bounds of sum_b ceil(count_b / slice): every bucket wastes at most one partial slice
    const int64_t wg_users = a.n_user_buckets + (T + kSliceEntries - 1) / kSliceEntries;
    const int64_t wg_items = (a.n_buckets - a.n_user_buckets) + (2 * T + kSliceEntries - 1) / kSliceEntries;
    hipLaunchKernelGGL((bpr_bucket_update_kernel<true>), dim3((unsigned)wg_users), dim3(kWave * kScoreWaves), 0, stream, a);
    hipLaunchKernelGGL((bpr_bucket_update_kernel<false>), dim3((unsigned)wg_items), dim3(kWave * kScoreWaves), 0, stream, a);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_bpr_sgd_step_grouped_f32(const float *U_read, const float *I_read, float *U_w, float *I_w,
                                             int64_t n_user_rows, int64_t n_item_rows, const int64_t *u, const int64_t *i_pos,
                                             const int64_t *i_neg, int64_t T, int32_t d, float lr, float reg, float *loss_sum,
                                             void *ws, int64_t ws_bytes, void *stream)
{
    SPEX_CHECK_ARG(U_read && I_read && U_w && I_w && u && i_pos && i_neg, "spex_bpr_sgd_step_grouped_f32: NULL pointer");
    SPEX_CHECK_ARG(T >= 0, "spex_bpr_sgd_step_grouped_f32: T=%lld", (long long)T);
    SPEX_CHECK_ARG(U_read != U_w && I_read != I_w, "spex_bpr_sgd_step_grouped_f32: batch-synchronous form only (read tables must differ from the updated tables)");
    if (d != kWave) {
        spex::set_error("spex_bpr_sgd_step_grouped_f32: d == 64 only (got %d): use spex_bpr_sgd_step_f32", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    if (T == 0) return SPEX_OK;
    return launch_bpr_grouped(U_read, I_read, U_w, I_w, n_user_rows, n_item_rows, u, i_pos, i_neg, T, -lr / (float)T,
                              -lr * reg / (float)T, loss_sum, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int spex_bpr_loss_grouped_f32(const float *users, const float *items, int64_t n_user_rows, int64_t n_item_rows,
                                         const int64_t *u, const int64_t *i_pos, const int64_t *i_neg, int64_t T, int32_t d,
                                         float *loss_sum, float *grad_users, float *grad_items, float grad_scale, void *ws,
                                         int64_t ws_bytes, void *stream)
{
    SPEX_CHECK_ARG(users && items && u && i_pos && i_neg && grad_users && grad_items, "spex_bpr_loss_grouped_f32: NULL pointer");
    SPEX_CHECK_ARG(T >= 0, "spex_bpr_loss_grouped_f32: T=%lld", (long long)T);
    if (d != kWave) {
        spex::set_error("spex_bpr_loss_grouped_f32: d == 64 only (got %d): use spex_bpr_loss_f32", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    if (T == 0) return SPEX_OK;
    return launch_bpr_grouped(users, items, grad_users, grad_items, n_user_rows, n_item_rows, u, i_pos, i_neg, T, grad_scale, 0.0f,
                              loss_sum, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int spex_bpr_loss_f32(const float *users, const float *items, int64_t n_user_rows, int64_t n_item_rows,
                                 const int64_t *u, const int64_t *i_pos, const int64_t *i_neg, int64_t T, int32_t d,
                                 float *loss_sum, float *grad_users, float *grad_items, float grad_scale, void *stream)
{
    SPEX_CHECK_ARG(users && items && u && i_pos && i_neg, "spex_bpr_loss_f32: NULL pointer");
    SPEX_CHECK_ARG(T >= 0 && d >= 1, "spex_bpr_loss_f32: T=%lld d=%d", (long long)T, d);
    SPEX_CHECK_ARG((grad_users == nullptr) == (grad_items == nullptr), "spex_bpr_loss_f32: give both grad tables or neither");
    if (T == 0) return SPEX_OK;
    const int per_wave = (grad_users && d == kWave) ? bpr_per_wave(T) : 1;
    hipLaunchKernelGGL(bpr_kernel, dim3(grid_for((T + per_wave - 1) / per_wave)), dim3(kWave * kScoreWaves), 0,
                       (hipStream_t)stream, users, items, grad_users, grad_items, u, i_pos, i_neg, T, d, n_user_rows,
                       n_item_rows, grad_scale, 0.0f, loss_sum, per_wave);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_gather_owned_rows_f32(const float *table, const int64_t *pos, int64_t K, int64_t lo, int64_t n_local,
                                          int32_t d, float *out, void *stream)
{
    SPEX_CHECK_ARG((table || n_local == 0) && pos && out, "spex_gather_owned_rows_f32: NULL pointer");
    SPEX_CHECK_ARG(K >= 0 && n_local >= 0 && d >= 1, "spex_gather_owned_rows_f32: K=%lld n_local=%lld d=%d", (long long)K,
                   (long long)n_local, d);
    if (K == 0) return SPEX_OK;
    hipLaunchKernelGGL(gather_owned_rows_kernel, dim3(grid_for(K)), dim3(kWave * kScoreWaves), 0, (hipStream_t)stream, table, pos,
                       K, lo, n_local, d, out);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_scatter_add_owned_rows_f32(float *upd, const int64_t *pos, int64_t K, int64_t lo, int64_t n_local,
                                               int32_t d, float *table, int32_t clear_upd, void *stream)
{
    SPEX_CHECK_ARG(upd && pos && (table || n_local == 0), "spex_scatter_add_owned_rows_f32: NULL pointer");
    SPEX_CHECK_ARG(K >= 0 && n_local >= 0 && d >= 1, "spex_scatter_add_owned_rows_f32: K=%lld n_local=%lld d=%d", (long long)K,
                   (long long)n_local, d);
    if (K == 0) return SPEX_OK;
    hipLaunchKernelGGL(scatter_add_owned_rows_kernel, dim3(grid_for(K)), dim3(kWave * kScoreWaves), 0, (hipStream_t)stream, upd,
                       pos, K, lo, n_local, d, table, clear_upd);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
