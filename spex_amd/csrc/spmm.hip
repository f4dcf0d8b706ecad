// CSR x dense fp32 SpMM for CDNA4 (gfx950) — the LightGCN / NGCF propagation kernel.
//
// Replaces torch.sparse.mm(Graph, all_emb): LightGCN_SPEX/code/utility1/model.py:91, NGCF_SPEX/code/main_rec.py:76.
//
// Shape of the work: N rows, ~27 stored entries per row on Epinion2 (median 13, max 1020), each entry gathers one
// 256-byte embedding row (d = 64 fp32).  0.5 flop per byte: a gather/reduce bound by HBM (or, for the cache-resident
// named datasets, by L2 / Infinity Cache) — no MFMA, no GEMM reshaping.
//
// Mapping: one 64-lane wavefront per row (or per 128-entry segment of a long row); lane == embedding column, so one
// gathered row is exactly one fully coalesced 256-byte wave load (two 128-byte lines).  The wave first loads up to 64
// (col, val) pairs with one coalesced load each (lane k holds entry k), then walks them with v_readlane: the column
// index becomes a scalar, the row address a scalar base + lane offset, and 8 independent row gathers are issued
// back-to-back before the first fmaf consumes them.  The accumulation is a single fmaf chain in ascending column
// order, i.e. bit-for-bit the order the reference's CPU kernel uses.
//
// Long rows (> 128 entries) would serialise one wave for tens of microseconds, so graph creation cuts them into
// 128-entry segments; each segment is an ordinary wave task that writes a 256-byte partial row, and a tiny second
// launch adds a row's partials in segment order (deterministic, no atomics) and applies the epilogue.
//
// Epilogue (fused, saves one full pass over the embedding matrix per layer each):
//   y += add_in / add_div            backward of the layer mean (g/(L+1) + A^T G)
//   Y = y                            next layer's input (skipped on the last layer)
//   acc_out = (acc_in + y) / acc_div running sum of layers; acc_div = L+1 on the last layer gives the mean
//
// Edge dropout (model.py:46-55) is applied while the (col, val) pairs are loaded: dropped entries are removed from
// the wave's ballot mask, so they vanish from the sum exactly as if the sparse matrix had been rebuilt without them.
#include "spex_common.h"

using namespace spex;

namespace {

struct SpmmParams {
    const int32_t *rowptr, *col;
    const float *val;
    const int32_t *edge_id;
    const int32_t *seg_beg, *seg_end, *long_row, *long_seg0;
    int32_t n_rows, n_seg, n_long;
    const int4 *task;
    const int32_t *entry_row;
    int32_t n_tasks;
    const float *X;
    float *Y;
    const float *add_in;
    float add_div;
    const float *acc_in;
    float *acc_out;
    float acc_div;
    float *partial;
    int32_t d;
    int mask_mode;
    const uint8_t *keep;
    float keep_prob;
    uint32_t seed_lo, seed_hi;
};

constexpr int kUnroll = 8;

// philox4x32-10, counter = (edge_id, 0, 0, 0), key = seed.  Returns the first output word.
__device__ __forceinline__ uint32_t philox_first(uint32_t ctr0, uint32_t k0, uint32_t k1)
{
    uint32_t c0 = ctr0, c1 = 0u, c2 = 0u, c3 = 0u;
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

__device__ __forceinline__ bool edge_kept(const SpmmParams &p, int eid)
{
    if (p.mask_mode == 1) return p.keep[eid] != 0;
    // floor(rand + keep_prob) with rand a 24-bit uniform in [0,1), as torch.rand produces (model.py:50-51)
    const float u = (float)(philox_first((uint32_t)eid, p.seed_lo, p.seed_hi) >> 8) * 5.9604644775390625e-8f;
    return (u + p.keep_prob) >= 1.0f;
}

__device__ __forceinline__ float lane_bcast(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// Accumulate entries [beg, end) of one row into acc for the column this lane owns.
//   Xl      X + (column owned by this lane)
//   ldx     row stride of X in floats
//   col_ok  false for lanes beyond d in the generic-d tile
template <bool MASKED>
__device__ __forceinline__ float accumulate_range(const SpmmParams &p, int beg, int end, int lane,
                                                  const float *__restrict__ Xl, int ldx, bool col_ok, float acc)
{
    for (int base = beg; base < end; base += kWave) {
        const int n = (end - base < kWave) ? end - base : kWave;  // wave-uniform
        int my_col = 0;
        float my_val = 0.0f;
        bool my_keep = false;
        if (lane < n) {
            my_col = p.col[base + lane];
            my_val = p.val[base + lane];
            if (MASKED) {
                const int eid = p.edge_id ? p.edge_id[base + lane] : base + lane;
                my_keep = edge_kept(p, eid);
                my_val = my_val / p.keep_prob;  // values[random_index] / keep_prob, model.py:53
            }
        }
        if (!MASKED) {
            int i = 0;
            for (; i + kUnroll <= n; i += kUnroll) {
                float x[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const int c = __builtin_amdgcn_readlane(my_col, i + u);
                    x[u] = col_ok ? Xl[(size_t)c * ldx] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) acc = fmaf(lane_bcast(my_val, i + u), x[u], acc);
            }
            if (i < n) {  // ragged tail: same batch with wave-uniform predicates, loads still issue back-to-back
                float x[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    x[u] = 0.0f;
                    if (i + u < n) {
                        const int c = __builtin_amdgcn_readlane(my_col, i + u);
                        x[u] = col_ok ? Xl[(size_t)c * ldx] : 0.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u)
                    if (i + u < n) acc = fmaf(lane_bcast(my_val, i + u), x[u], acc);
            }
        } else {
            unsigned long long m = __ballot(my_keep);  // wave-uniform set of surviving entries
            while (m) {
                int idx[kUnroll];
                int cnt = 0;
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    idx[u] = 0;
                    if (m) {
                        idx[u] = __builtin_ctzll(m);
                        m &= m - 1;
                        cnt = u + 1;
                    }
                }
                float x[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    x[u] = 0.0f;
                    if (u < cnt) {
                        const int c = __builtin_amdgcn_readlane(my_col, idx[u]);
                        x[u] = col_ok ? Xl[(size_t)c * ldx] : 0.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u)
                    if (u < cnt) acc = fmaf(lane_bcast(my_val, idx[u]), x[u], acc);
            }
        }
    }
    return acc;
}

__device__ __forceinline__ void finish_row(const SpmmParams &p, int r, int c, float y)
{
    const size_t o = (size_t)r * p.d + c;
    if (p.add_in) y = y + p.add_in[o] / p.add_div;
    if (p.Y) p.Y[o] = y;
    if (p.acc_out) p.acc_out[o] = (p.acc_in[o] + y) / p.acc_div;
}

// Blocks are dealt round-robin over the 8 XCDs; remap so that each XCD walks a contiguous range of row blocks
// (neighbouring rows share the cache lines where one row's (col,val) run ends and the next begins).  Speed only.
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nblk)
{
    return ((nblk & 7) == 0) ? (bid & 7) * (nblk >> 3) + (bid >> 3) : bid;
}

// One wave per task.  Tasks [0, n_seg) are long-row segments (heaviest work first), tasks [n_seg, n_seg + n_rows)
// are rows; rows longer than kLongRow are left to their segments.
template <bool MASKED>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_rows_kernel(const SpmmParams p)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int task = xcd_contiguous_block(blockIdx.x, gridDim.x) * kWavesPerBlock + wave;
    const int d = p.d;

    if (task < p.n_seg) {
        const int beg = p.seg_beg[task], end = p.seg_end[task];
        for (int c0 = 0; c0 < d; c0 += kWave) {
            const bool ok = c0 + lane < d;
            const float y = accumulate_range<MASKED>(p, beg, end, lane, p.X + c0 + lane, d, ok, 0.0f);
            if (ok) p.partial[(size_t)task * d + c0 + lane] = y;
        }
        return;
    }
    const int r = task - p.n_seg;
    if (r >= p.n_rows) return;
    const int beg = p.rowptr[r], end = p.rowptr[r + 1];
    if (end - beg > kLongRow) return;
    for (int c0 = 0; c0 < d; c0 += kWave) {
        const bool ok = c0 + lane < d;
        const float y = accumulate_range<MASKED>(p, beg, end, lane, p.X + c0 + lane, d, ok, 0.0f);
        if (ok) finish_row(p, r, c0 + lane, y);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// d == 64 kernel.  A 256-byte embedding row is 16 lanes x float4, so one wave-wide `global_load_dwordx4` gathers FOUR
// rows (1 KiB per instruction, the widest access the memory pipeline has; 4-byte-per-lane gathers measured 3x fewer
// bytes per CU-clock on this part).  The wave is therefore split into four 16-lane groups, each walking its own TASK:
// a contiguous run of stored entries that covers whole short rows (<= 16 entries in total), one row of 17..128
// entries, or one 128-entry segment of a long row.  Per 16-entry chunk the group's lanes load (col, val, row) with one
// coalesced load each and hand them round with ds_bpermute; four gathers per group are kept in flight (16 rows per
// wave) and consumed in order — one fmaf chain per output element, ascending column, the reference's summation
// order.  A row's epilogue operand (running layer sum, or g/(L+1)) is fetched in the same batch as the row's last
// gather, so finishing a row never waits on a dependent load.  Divergence is only ever at group granularity.
constexpr int kQ = 16;      // lanes per group == entries per metadata chunk
constexpr int kUnrollQ = 4; // gathers in flight per group

__device__ __forceinline__ int bperm_i(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ float bperm_f(int src_lane, float v)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

template <bool MASKED, int EPI>  // EPI: 0 none, 1 acc (forward), 2 add (backward), 3 both (generic, loads late)
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_q4_kernel(const SpmmParams p)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int q = lane & (kQ - 1);
    const int gbase = lane & ~(kQ - 1);
    const int wave = (int)(threadIdx.x >> 6);
    const int task_id = (xcd_contiguous_block(blockIdx.x, gridDim.x) * kWavesPerBlock + wave) * 4 + (lane >> 4);
    int4 t = make_int4(0, 0, -1, -1);
    if (task_id < p.n_tasks) t = p.task[task_id];
    const int n = t.y - t.x;
    const bool partial = t.z >= 0;
    int nmax = n;  // wave-uniform trip count = longest of the four tasks
    nmax = max(nmax, __shfl_xor(nmax, 16, kWave));
    nmax = max(nmax, __shfl_xor(nmax, 32, kWave));
    nmax = __builtin_amdgcn_readfirstlane(nmax);

    const float *__restrict__ Xq = p.X + q * 4;
    const float *__restrict__ epi_ptr = ((EPI == 1) ? p.acc_in : p.add_in);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int cb = 0; cb < nmax; cb += kQ) {
        // this group's next 16 entries, one per lane
        const int e = t.x + cb + q;
        const bool valid = cb + q < n;
        int my_cl = 0, my_row = 0;
        float my_val = 0.0f;
        if (valid) {
            my_cl = p.col[e];
            my_val = p.val[e];
            my_row = p.entry_row[e];
            bool last;
            if (partial) last = (e + 1 == t.y);
            else last = (e + 1 < t.y) ? (p.entry_row[e + 1] != my_row) : true;
            if (last) my_cl |= (int)0x80000000u;
            if (MASKED) {
                const int eid = p.edge_id ? p.edge_id[e] : e;
                if (!edge_kept(p, eid)) my_cl |= 0x40000000;
                my_val = my_val / p.keep_prob;
            }
        }
        const int cnt = n - cb;  // entries of this group still to do in this chunk (may be <= 0)
        for (int j0 = 0; j0 < kQ && cb + j0 < nmax; j0 += kUnrollQ) {
            float4 x[kUnrollQ], ep[kUnrollQ];
            float v[kUnrollQ];
            int cl[kUnrollQ], rr[kUnrollQ];
#pragma unroll
            for (int u = 0; u < kUnrollQ; ++u) {
                const int j = j0 + u;
                cl[u] = bperm_i(gbase + j, my_cl);
                v[u] = bperm_f(gbase + j, my_val);
                rr[u] = bperm_i(gbase + j, my_row);
                x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                ep[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < cnt) {
                    if (!MASKED || !(cl[u] & 0x40000000)) {
                        const int c = cl[u] & 0x3fffffff;
                        x[u] = *reinterpret_cast<const float4 *>(Xq + (size_t)c * 64);
                    }
                    if ((EPI == 1 || EPI == 2) && cl[u] < 0 && !partial)
                        ep[u] = *reinterpret_cast<const float4 *>(epi_ptr + (size_t)rr[u] * 64 + q * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < kUnrollQ; ++u) {
                const int j = j0 + u;
                if (j < cnt) {
                    if (!MASKED || !(cl[u] & 0x40000000)) {
                        acc.x = fmaf(v[u], x[u].x, acc.x);
                        acc.y = fmaf(v[u], x[u].y, acc.y);
                        acc.z = fmaf(v[u], x[u].z, acc.z);
                        acc.w = fmaf(v[u], x[u].w, acc.w);
                    }
                    if (cl[u] < 0) {  // last entry of a row (or of the segment): emit
                        if (partial) {
                            *reinterpret_cast<float4 *>(p.partial + (size_t)t.z * 64 + q * 4) = acc;
                        } else {
                            const size_t o = (size_t)rr[u] * 64 + q * 4;
                            if (EPI == 0) {
                                *reinterpret_cast<float4 *>(p.Y + o) = acc;
                            } else if (EPI == 1) {
                                if (p.Y) *reinterpret_cast<float4 *>(p.Y + o) = acc;
                                float4 s = make_float4(ep[u].x + acc.x, ep[u].y + acc.y, ep[u].z + acc.z, ep[u].w + acc.w);
                                if (p.acc_div != 1.0f) {
                                    s.x = s.x / p.acc_div; s.y = s.y / p.acc_div; s.z = s.z / p.acc_div; s.w = s.w / p.acc_div;
                                }
                                *reinterpret_cast<float4 *>(p.acc_out + o) = s;
                            } else if (EPI == 2) {
                                float4 a = ep[u];
                                if (p.add_div != 1.0f) {
                                    a.x = a.x / p.add_div; a.y = a.y / p.add_div; a.z = a.z / p.add_div; a.w = a.w / p.add_div;
                                }
                                *reinterpret_cast<float4 *>(p.Y + o) = make_float4(acc.x + a.x, acc.y + a.y, acc.z + a.z, acc.w + a.w);
                            } else {
                                finish_row(p, rr[u], q * 4 + 0, acc.x);
                                finish_row(p, rr[u], q * 4 + 1, acc.y);
                                finish_row(p, rr[u], q * 4 + 2, acc.z);
                                finish_row(p, rr[u], q * 4 + 3, acc.w);
                            }
                        }
                        acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
    }
    if (n == 0 && t.w >= 0) {  // a row without stored entries: y = 0, epilogue still applies
        finish_row(p, t.w, q * 4 + 0, 0.0f);
        finish_row(p, t.w, q * 4 + 1, 0.0f);
        finish_row(p, t.w, q * 4 + 2, 0.0f);
        finish_row(p, t.w, q * 4 + 3, 0.0f);
    }
}

// One wave per long row: add its segment partials in order, then the shared epilogue.
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_long_fixup_kernel(const SpmmParams p)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= p.n_long) return;
    const int r = p.long_row[i];
    const int s0 = p.long_seg0[i], s1 = p.long_seg0[i + 1];
    for (int c = lane; c < p.d; c += kWave) {
        float y = 0.0f;
        int s = s0;
        for (; s + 4 <= s1; s += 4) {
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = p.partial[(size_t)(s + u) * p.d + c];
#pragma unroll
            for (int u = 0; u < 4; ++u) y = y + t[u];
        }
        for (; s < s1; ++s) y = y + p.partial[(size_t)s * p.d + c];
        finish_row(p, r, c, y);
    }
}

// out = in / div, float4-wide grid-stride (the mean's backward share g/(L+1), model.py:95).
__global__ __launch_bounds__(256) void div_kernel(const float *__restrict__ in, float *__restrict__ out, float div,
                                                  int64_t n4, int64_t rem)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4 *>(in)[i];
        v.x = v.x / div; v.y = v.y / div; v.z = v.z / div; v.w = v.w / div;
        reinterpret_cast<float4 *>(out)[i] = v;
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < rem) out[n4 * 4 + threadIdx.x] = in[n4 * 4 + threadIdx.x] / div;
}

int launch_spmm(const spex_graph *g, const float *X, float *Y, const float *add_in, float add_div, const float *acc_in,
                float *acc_out, float acc_div, int32_t d, hipStream_t stream)
{
    if (g->n_rows == 0) return SPEX_OK;
    SpmmParams p;
    p.rowptr = g->rowptr; p.col = g->col; p.val = g->val; p.edge_id = g->edge_id;
    p.seg_beg = g->seg_beg; p.seg_end = g->seg_end; p.long_row = g->long_row; p.long_seg0 = g->long_seg0;
    p.n_rows = g->n_rows; p.n_seg = g->n_seg; p.n_long = g->n_long;
    p.X = X; p.Y = Y; p.add_in = add_in; p.add_div = add_div; p.acc_in = acc_in; p.acc_out = acc_out; p.acc_div = acc_div;
    p.partial = g->partial; p.d = d;
    p.mask_mode = g->mask_mode; p.keep = g->keep; p.keep_prob = g->keep_prob;
    p.seed_lo = (uint32_t)g->seed; p.seed_hi = (uint32_t)(g->seed >> 32);

    p.task = g->task; p.entry_row = g->entry_row; p.n_tasks = g->n_tasks;
    const bool masked = g->mask_mode != 0, d64 = d == 64;
    const int64_t tasks = d64 ? (int64_t)g->n_tasks / 4 : (int64_t)g->n_seg + g->n_rows;  // waves
    int64_t blocks = (tasks + kWavesPerBlock - 1) / kWavesPerBlock;
    blocks = (blocks + 7) / 8 * 8;  // multiple of the XCD count so the remap is a bijection
    const dim3 grid((unsigned)blocks), block(kWave * kWavesPerBlock);
    spex_timer *tm = g->timer;
    const bool timed = tm && tm->used < (int32_t)tm->start.size();
    if (timed) SPEX_HIP(hipEventRecord(tm->start[tm->used], stream));
    if (d64) {
        const int epi = (acc_out ? 1 : 0) | (add_in ? 2 : 0);
#define SPEX_LAUNCH_TASKS(M, E) hipLaunchKernelGGL((spmm_q4_kernel<M, E>), grid, block, 0, stream, p)
        if (masked) {
            if (epi == 0) SPEX_LAUNCH_TASKS(true, 0);
            else if (epi == 1) SPEX_LAUNCH_TASKS(true, 1);
            else if (epi == 2) SPEX_LAUNCH_TASKS(true, 2);
            else SPEX_LAUNCH_TASKS(true, 3);
        } else {
            if (epi == 0) SPEX_LAUNCH_TASKS(false, 0);
            else if (epi == 1) SPEX_LAUNCH_TASKS(false, 1);
            else if (epi == 2) SPEX_LAUNCH_TASKS(false, 2);
            else SPEX_LAUNCH_TASKS(false, 3);
        }
#undef SPEX_LAUNCH_TASKS
    } else if (masked) {
        hipLaunchKernelGGL((spmm_rows_kernel<true>), grid, block, 0, stream, p);
    } else {
        hipLaunchKernelGGL((spmm_rows_kernel<false>), grid, block, 0, stream, p);
    }
    if (timed) {
        SPEX_HIP(hipEventRecord(tm->stop[tm->used], stream));
        tm->used++;
    }
    if (g->n_long > 0) {
        const dim3 fgrid((unsigned)((g->n_long + kWavesPerBlock - 1) / kWavesPerBlock));
        hipLaunchKernelGGL(spmm_long_fixup_kernel, fgrid, block, 0, stream, p);
    }
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

int ensure_partial(spex_graph *g, int32_t d)
{
    const int64_t need = (int64_t)g->n_seg * d;
    if (need <= g->partial_cap) return SPEX_OK;
    // Only reached for d > 64 on a graph with long rows, on the first call at that d (allocates: do it once outside
    // any stream capture).
    SPEX_HIP(hipDeviceSynchronize());
    if (g->partial) SPEX_HIP(hipFree(g->partial));
    g->partial = nullptr;
    SPEX_HIP(hipMalloc((void **)&g->partial, (size_t)need * sizeof(float)));
    g->partial_cap = need;
    return SPEX_OK;
}

}  // namespace

extern "C" int spex_spmm_f32(const spex_graph_t *g, const float *X, float *Y, const float *add_in, float add_div,
                             const float *acc_in, float *acc_out, float acc_div, int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g && X, "spex_spmm_f32: NULL graph or X");
    SPEX_CHECK_ARG(d >= 1, "spex_spmm_f32: d = %d", d);
    SPEX_CHECK_ARG(Y || acc_out, "spex_spmm_f32: neither Y nor acc_out given");
    SPEX_CHECK_ARG(!acc_out || acc_in, "spex_spmm_f32: acc_out needs acc_in");
    SPEX_CHECK_ARG(X != Y && X != acc_out, "spex_spmm_f32: X must not alias an output");
    SPEX_CHECK_ARG(!add_in || add_div != 0.0f, "spex_spmm_f32: add_div == 0");
    SPEX_CHECK_ARG(!acc_out || acc_div != 0.0f, "spex_spmm_f32: acc_div == 0");
    int rc = ensure_partial(const_cast<spex_graph *>(g), d);
    if (rc) return rc;
    return launch_spmm(g, X, Y, add_in, add_div, acc_in, acc_out, acc_div, d, (hipStream_t)stream);
}

extern "C" int spex_propagate_f32(const spex_graph_t *g, const float *E0, float *mean_out, float *layers_out, float *ws,
                                  int32_t L, int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g && E0 && mean_out, "spex_propagate_f32: NULL argument");
    SPEX_CHECK_ARG(g->n_rows == g->n_cols, "spex_propagate_f32: needs a square graph (%d x %d); use spex_spmm_f32 per layer for a row block", g->n_rows, g->n_cols);
    SPEX_CHECK_ARG(L >= 0 && d >= 1, "spex_propagate_f32: L = %d, d = %d", L, d);
    SPEX_CHECK_ARG(layers_out || ws || L <= 1, "spex_propagate_f32: needs ws or layers_out");
    SPEX_CHECK_ARG(E0 != mean_out, "spex_propagate_f32: mean_out must not alias E0");
    int rc = ensure_partial(const_cast<spex_graph *>(g), d);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t sz = (size_t)g->n_rows * d;
    if (L == 0) {
        SPEX_HIP(hipMemcpyAsync(mean_out, E0, sz * sizeof(float), hipMemcpyDeviceToDevice, s));
        return SPEX_OK;
    }
    const float *cur = E0;
    for (int32_t l = 0; l < L; ++l) {
        const bool last = l == L - 1;
        float *nxt = layers_out ? layers_out + (size_t)l * sz : (last ? nullptr : ws + (size_t)(l & 1) * sz);
        rc = launch_spmm(g, cur, nxt, nullptr, 1.0f, l == 0 ? E0 : mean_out, mean_out, last ? (float)(L + 1) : 1.0f, d, s);
        if (rc) return rc;
        cur = nxt;
    }
    return SPEX_OK;
}

extern "C" int spex_propagate_bwd_f32(const spex_graph_t *gt, const float *g_out, float *grad_E0, float *ws, int32_t L,
                                      int32_t d, void *stream)
{
    SPEX_CHECK_ARG(gt && g_out && grad_E0 && ws, "spex_propagate_bwd_f32: NULL argument");
    SPEX_CHECK_ARG(gt->n_rows == gt->n_cols, "spex_propagate_bwd_f32: needs a square graph");
    SPEX_CHECK_ARG(L >= 0 && d >= 1, "spex_propagate_bwd_f32: L = %d, d = %d", L, d);
    SPEX_CHECK_ARG(g_out != grad_E0, "spex_propagate_bwd_f32: grad_E0 must not alias g_out");
    int rc = ensure_partial(const_cast<spex_graph *>(gt), d);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t sz = (size_t)gt->n_rows * d;
    if (sz == 0) return SPEX_OK;
    // gs = g_out / (L+1) is every layer's share of the mean's gradient and the first gather source (= G_L).
    float *gs = (L == 0) ? grad_E0 : ws;
    {
        const int64_t n4 = (int64_t)(sz / 4), rem = (int64_t)(sz % 4);
        const int64_t blocks = (n4 + 255) / 256 < 2048 ? ((n4 + 255) / 256 > 0 ? (n4 + 255) / 256 : 1) : 2048;
        hipLaunchKernelGGL(div_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g_out, gs, (float)(L + 1), n4, rem);
    }
    const float *cur = gs;
    for (int32_t l = L - 1; l >= 0; --l) {
        float *nxt = (l == 0) ? grad_E0 : ws + (size_t)(1 + (l & 1)) * sz;
        rc = launch_spmm(gt, cur, nxt, gs, 1.0f, nullptr, nullptr, 1.0f, d, s);
        if (rc) return rc;
        cur = nxt;
    }
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
