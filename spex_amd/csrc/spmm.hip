// CSR x dense fp32 SpMM for CDNA4 (gfx950) — the LightGCN / NGCF propagation kernel.
//
// Replaces torch.sparse.mm(Graph, all_emb): LightGCN_SPEX/code/utility1/model.py:91, NGCF_SPEX/code/main_rec.py:76.
//
// Shape of the work: N rows, ~27 stored entries per row on Epinion2 (median 13, max 1020), each entry gathers one
// 256-byte embedding row (d = 64 fp32).  0.5 flop per byte: a gather/reduce bound by HBM (or, for the cache-resident
// named datasets, by L2 / Infinity Cache) — no MFMA, no GEMM reshaping.
//
// Two kernels share the lane mapping (lane == embedding column: one gathered row is one coalesced 256-byte wave load,
// two 128-byte lines) and the arithmetic (one fmaf chain per output element in ascending column order — bit for bit
// the order of the reference's CPU kernel):
//
//   spmm_chunk_kernel  d == 64, the tuned path (described at its definition): 16-wave workgroups, one wave per
//                      <= 64-entry task read from the chunked task table, long rows summed through LDS.
//   spmm_rows_kernel   any d: one wave per row (or per kSegLen-entry segment of a row with more than kLongRow entries), walking
//                      the plain CSR in column tiles of 64; segment sums go through global scratch and
//                      spmm_long_fixup_kernel adds them in segment order.  Correct everywhere, tuned nowhere.
//
// Fused epilogue (each saves a full pass over the embedding matrix per layer):
//   y += add_in / add_div            backward of the layer mean (g/(L+1) + A^T G)
//   Y = y                            next layer's input (skipped on the last layer)
//   acc_out = (acc_in + y) / acc_div running sum of layers; acc_div = L+1 on the last layer gives the mean
//
// Edge dropout (model.py:46-55) is decided per stored entry while its (col, val) pair is loaded — injected mask byte
// or Philox draw keyed by the entry's edge id — so forward, backward and all layers of a step drop the same edges.
#include <mutex>

#include "spex_common.h"

using namespace spex;

namespace {

struct SpmmParams {
    const int32_t *rowptr, *col;
    const float *val;
    const int32_t *edge_id;
    const int32_t *seg_beg, *seg_end, *long_row, *long_seg0;
    int32_t n_rows, n_seg, n_long;

    const float *X;
    float *Y;
    const float *add_in;
    float add_div;
    float out_div;  // y = (y + add_in / add_div) / out_div
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
    if (p.out_div != 1.0f) y = y / p.out_div;
    if (p.Y) p.Y[o] = y;
    if (p.acc_out) p.acc_out[o] = (p.acc_in[o] + y) / p.acc_div;
}

// Blocks are dealt round-robin over the 8 XCDs; remap so that each XCD walks a contiguous range of row blocks
// (neighbouring rows share the cache lines where one row's (col,val) run ends and the next begins).  Speed only.
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nblk)
{
    return ((nblk & 7) == 0) ? (bid & 7) * (nblk >> 3) + (bid >> 3) : bid;
}

// Generic-d kernel: one wave per task.  Tasks [0, n_seg) are long-row segments (heaviest work first), tasks
// [n_seg, n_seg + n_rows) are rows; rows longer than kLongRow are left to their segments.
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
// d == 64 kernel: 16-wave workgroups, one wave per TASK (a run of <= 4 sixteen-entry chunks, see spex_common.h),
// lane == embedding column.
//
// What was measured on MI355X on the way here (tools/gather_bench.hip, profiles/r01*):
//   * random 256-byte row gathers run at the same rate — 6.1 TB/s from HBM, 15-24 TB/s from L2 / Infinity Cache —
//     whether a row is fetched as 64 x 4 B or 16 x 16 B lanes, with 8 or 64 rows in flight per wave: the memory system,
//     not the load shape, sets the ceiling;
//   * what separates an SpMM from that ceiling is everything it issues AROUND the gathers.  The first version (one
//     wave per row, rows walked with per-entry row-boundary tests and 64-bit scalar address arithmetic) spent 15 scalar
//     + 6 vector instructions per gathered row and saturated the CU's scalar unit at a third of the gather rate;
//   * feeding the per-entry metadata through scalar loads (s_load_dwordx16) starves on scalar-cache misses as soon
//     as the matrix streams from HBM (29 ms vs 13 ms per launch on a 2^23-node graph); buffer loads with a scalar
//     offset are ~25 % slower than global loads when the source table is cache-resident.
// Hence this shape:
//   * per 64 entries the wave loads source-row index, value (and, for bin-packed tasks, output row) with ONE coalesced
//     vector load each — lane k holds entry k — and hands them to the scalar side with v_readlane;
//   * per 16-entry chunk: 16 independent `global_load_dword` (scalar row base + lane offset) issued back to back,
//     then 16 x { v_fmac; s_bitcmp on the chunk's end-of-row mask; branch }.  A row's epilogue operand (running layer
//     sum, or g/(L+1)) is fetched in the same batch as its last gather, so emitting a row never waits on a
//     dependent load;
//   * tasks are padded to whole chunks with value-0 entries on the task's last real source row (an L1 hit);
//   * rows of 65..1024 entries are cut into 64-entry segments that all sit in one workgroup; their sums meet in LDS
//     and are added in segment order by the row's first wave: deterministic, no atomics, no second launch.  Rows beyond
//     1024 entries (hubs) go through global scratch: their segments lead the task table in groups of 16 adjacent waves, a
//     group is summed through LDS the same way and leaves ONE partial row, and the hub's last group to arrive (a ticket in
//     memory, no waiting) adds the groups in order and runs the epilogue — inside this launch (HubFold; the wide kernels
//     and SPEX_HUB_FOLD=0 leave one partial row per segment to spmm_long_fixup_kernel instead).
// The accumulation is one fmaf chain per output element in ascending column order (bit-exact vs the reference's CPU
// kernel) for every row that fits a task; segmented rows re-associate <= 16 partial sums.

// Edge dropout (MASKED): lane k decides for its own entry (injected mask byte or Philox draw keyed by the entry's edge
// id, so forward / backward / every layer of a step agree); a dropped entry keeps its slot with value 0 and the
// source row of the super-chunk's first kept entry (already being fetched), so it costs no extra traffic — and its
// gathered value is REPLACED by 0 before the fmaf (a wave-uniform select per entry), so that it contributes exactly
// nothing, as in the reference, where the entry is gone from the matrix: multiplying instead would turn a stand-in row
// holding Inf / NaN (a diverged table) into 0 * Inf = NaN on a row that never referenced it.
// In-kernel fold of the rows beyond kWgRowMax entries (tag != 0; spex_common.h: hub_grp / hub_fold / hub_ticket).  tag == 0: the
// hub segments leave one partial row each and spmm_long_fixup_kernel adds them (the wide kernels, and SPEX_HUB_FOLD=0).
#ifndef SPEX_EPI_NT
#define SPEX_EPI_NT 1
#endif
#if SPEX_EPI_NT
#define SPEX_EPI_LOAD(p) __builtin_nontemporal_load(p)
#else
#define SPEX_EPI_LOAD(p) (*(p))
#endif

struct HubFold {
    const int4 *grp;
    const int2 *fold;
    unsigned long long *ticket;
    uint32_t tag;
};

struct DropArgs {
    const uint32_t *chunk_eid;
    const uint8_t *keep;
    int mode;
    float keep_prob;
    uint32_t seed_lo, seed_hi;
};

// EPI 0: Y = y;  1: Y = y (optional), acc_out = (acc_in + y) / epi_div;  2: Y = (y + add_in / epi_div) / out_div.
// ROWIDS: a task's rows are listed per entry (bin-packed tasks) instead of being adjacent from task.z.
// FOLD: the hub fold is compiled in.  The unmasked instantiations always carry it (it costs their chunk loop nothing); the
// edge-dropout ones sit ~20 scalar registers higher (the Philox key schedule lives in SGPRs) and with the fold's operands live across
// the loop they schedule worse (<0, MASKED> 13.1 -> 15.7 us on Epinion2, a graph without a single hub), so a masked launch takes the
// FOLD instantiation only on a graph that HAS hubs — where it replaces a ~5 us dependent fix-up launch per product.
template <int EPI, bool MASKED, bool ROWIDS, bool FOLD>
__global__ __launch_bounds__(kWave *kWgWaves) void spmm_chunk_kernel(
    const float *__restrict__ X, const uint32_t *__restrict__ chunk_off, const float *__restrict__ chunk_val,
    const uint32_t *__restrict__ chunk_mask, const int32_t *__restrict__ chunk_row, const int4 *__restrict__ task,
    int n_tasks, float *__restrict__ Y,
    const float *epi_in, float epi_div, float out_div, float *acc_out,   // may alias each other (running sum in place): no __restrict__
    float *__restrict__ partial,
    const DropArgs drop, const int xcd_contiguous, const HubFold hf_args)
{
    __shared__ float s_part[kWgWaves][kWave];  // segment sums of the rows this workgroup combines
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Workgroups are sorted heaviest-first and dealt round-robin over the 8 XCDs by the dispatcher, which gives every
    // XCD the same mix of weights: right for a graph that streams from HBM (+14 % on Epinion2x538 over handing XCD 0
    // all the heavy workgroups).  A graph whose whole source table sits in cache instead prefers each XCD to walk a
    // contiguous range of workgroups (user rows and item rows of the bipartite graph then gather from different
    // halves of the table in different L2s: 76 % vs 71 % L2 hits on Epinion2).  Placement is a speed matter only.
    const int wg = xcd_contiguous ? xcd_contiguous_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tid = wg * kWgWaves + wave;
    if (tid >= n_tasks) return;  // whole workgroups only (n_tasks is a multiple of kWgWaves)
    const int4 t = task[tid];
    const int kind = t.w & 3;
    const float *__restrict__ Xl = X + lane;
    const float *El = (EPI ? epi_in : X) + lane;
    int row = t.z;
    float acc = 0.0f;

    // v / div for a wave-uniform divisor, as REAL branches: written `if (div != 1.0f) v = v / div;` the compiler computes the quotient
    // always and selects (an fp32 division is ~11 vector instructions; two per emitted row in the backward form — 0.4 M of the launch's
    // 3 M vector instructions on Epinion2, spent on divisors that are 1.0 in most launches).  A power-of-two divisor (L + 1 = 4: the
    // reference's depth) is one exact multiply — x / 2^k == x * 2^-k bit for bit.  (The empty asm keeps each arm from being speculated.)
    const float epi_inv = 1.0f / epi_div, out_inv = 1.0f / out_div;      // (once per wave; exact for a power of two)
    auto scaled = [](float v, float div, float inv) {
        if (div != 1.0f) {
            if ((__float_as_uint(div) & 0x007FFFFFu) == 0u) {
                v = v * inv;
                asm volatile("" : "+v"(v));
            } else {
                v = v / div;
                asm volatile("" : "+v"(v));
            }
        }
        return v;
    };
    auto emit = [&](int r, float y, float e) {
        const size_t o = (size_t)r * 64 + lane;
        // outputs are streamed (non-temporal): they should not evict the gather source from the XCD's L2
        if (EPI == 0) {
            __builtin_nontemporal_store(y, Y + o);
        } else if (EPI == 1) {
            const float s = scaled(e + y, epi_div, epi_inv);
            if (Y) __builtin_nontemporal_store(y, Y + o);     // both stores after the last use of a loaded value
            __builtin_nontemporal_store(s, acc_out + o);
        } else {
            const float s = scaled(y + scaled(e, epi_div, epi_inv), out_div, out_inv);
            __builtin_nontemporal_store(s, Y + o);
        }
    };

    // Metadata of up to 4 chunks (64 entries) per vector load — lane k holds entry k — handed to the scalar side with
    // v_readlane.  (Feeding it through s_load instead starves on scalar-cache misses once the matrix streams from
    // HBM: 29 ms vs 13 ms per launch on a 2^23-node graph.)
    // A LIST of rows without stored entries (kind 3, bin-packed tables only; <= 64 rows, lane k holds the k-th row id): y = 0, the
    // epilogue still applies.  Done on the VECTOR side, four rows per instruction (16 lanes x float4 per row), sixteen rows' operands
    // in flight: taken one row at a time through the scalar emit path (load, wait, store) the list was a chain of up to 64 dependent
    // round trips — 32 us on the Weibo shape with its 2 441 empty rows, whose epilogue launches took twice as long as its plain ones —
    // and batching THAT path held sixteen row ids in SGPRs across the chunk loop (+1.5 us on Epinion2's <1> launch).
    if (ROWIDS && kind == 3) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int count = t.w >> 4;
        const int my_r = lane < count ? chunk_row[(size_t)t.x * kChunk + lane] : -1;
        const int sub = lane & 15, grp = lane >> 4;
        const f4 zero = (f4)(0.0f);
        for (int k0 = 0; k0 < count; k0 += 16) {
            int r[4];
            f4 e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[j] = __shfl(my_r, (k0 + 4 * j + grp) & 63, kWave);
                if (k0 + 4 * j + grp >= count) r[j] = -1;
                e[j] = zero;
                if (EPI != 0 && r[j] >= 0) e[j] = *reinterpret_cast<const f4 *>(epi_in + (size_t)r[j] * 64 + sub * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (r[j] < 0) continue;
                const size_t o = (size_t)r[j] * 64 + sub * 4;
                if (EPI == 0) {
                    __builtin_nontemporal_store(zero, reinterpret_cast<f4 *>(Y + o));
                } else if (EPI == 1) {
                    f4 sv = e[j] + zero;
                    if (epi_div != 1.0f) sv = sv / epi_div;
                    if (Y) __builtin_nontemporal_store(zero, reinterpret_cast<f4 *>(Y + o));
                    __builtin_nontemporal_store(sv, reinterpret_cast<f4 *>(acc_out + o));
                } else {
                    f4 ev = e[j];
                    if (epi_div != 1.0f) ev = ev / epi_div;
                    f4 sv = zero + ev;
                    if (out_div != 1.0f) sv = sv / out_div;
                    __builtin_nontemporal_store(sv, reinterpret_cast<f4 *>(Y + o));
                }
            }
        }
    }
    const int n_ch = kind == 3 ? 0 : t.y;              // (kind 3: handled above)
    for (int sc = 0; sc < n_ch; sc += 4) {
        const int nc = (n_ch - sc < 4) ? n_ch - sc : 4;  // chunks in this super-chunk (wave-uniform)
        uint32_t my_off = 0u, my_mask = 0u, own_off = 0u;
        unsigned long long drop_bits = 0ull;                 // MASKED: entries of this super-chunk that are dropped (wave-uniform)
        int my_row = 0;
        float my_val = 0.0f;
        if (lane < nc * kChunk) {
            const size_t e = (size_t)(t.x + sc) * kChunk + lane;
            my_off = __builtin_nontemporal_load(chunk_off + e);   // metadata is read once per launch
            my_val = __builtin_nontemporal_load(chunk_val + e);
            if (ROWIDS) my_row = __builtin_nontemporal_load(chunk_row + e);
            if (MASKED) {
                const uint32_t eid = __builtin_nontemporal_load(drop.chunk_eid + e);
                bool kept;
                if (drop.mode == 1) {
                    kept = drop.keep[eid] != 0;
                } else {
                    const float u01 = (float)(philox_first(eid, drop.seed_lo, drop.seed_hi) >> 8) * 5.9604644775390625e-8f;
                    kept = (u01 + drop.keep_prob) >= 1.0f;
                }
                my_val = kept ? my_val / drop.keep_prob : 0.0f;   // values[random_index] / keep_prob, model.py:53
                own_off = my_off;
                if (!kept) my_off = 0xFFFFFFFFu;
            }
        }
        if (MASKED) {
            // a dropped entry re-reads the source row of the first KEPT entry of this super-chunk (fetched anyway, so
            // it adds +0 and no traffic); if every entry was dropped each keeps its own row (value 0)
            const bool dropped = my_off == 0xFFFFFFFFu;      // (lanes past the super-chunk hold 0: neither kept nor dropped)
            drop_bits = __ballot(dropped);
            const unsigned long long kept_lanes = __ballot(!dropped && lane < nc * kChunk);
            const int src = kept_lanes ? (int)__builtin_ctzll(kept_lanes) : 0;
            const uint32_t stand_in = (uint32_t)__builtin_amdgcn_readlane((int)my_off, src);   // wave-uniform
            if (dropped) my_off = kept_lanes ? stand_in : own_off;
        }
        if (lane < nc) my_mask = __builtin_nontemporal_load(chunk_mask + t.x + sc + lane);
        float ep[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; ++u) ep[u] = 0.0f;
        for (int c = 0; c < nc; ++c) {
            const uint32_t mask = (uint32_t)__builtin_amdgcn_readlane((int)my_mask, c);
            const uint32_t dbits = MASKED ? (uint32_t)(drop_bits >> (c * kChunk)) & 0xFFFFu : 0u;
            float x[kChunk];
#pragma unroll
            for (int u = 0; u < kChunk; ++u)
                x[u] = Xl[(size_t)(uint32_t)__builtin_amdgcn_readlane((int)my_off, c * kChunk + u) * 64];
            // (ep[] lives ACROSS the chunks — declared in front of the loop — and is written only where a row ends: an entry that ends
            //  no row keeps whatever the register held, which nothing reads.  Defined per chunk, every such entry cost a v_mov to
            //  zero it, and a chunk without any row end sixteen of them plus the register shuffles of the join: ~0.6 M of the
            //  1.0 M vector instructions the epilogue forms issue beyond the plain form's 2.1 M on Epinion2)
            if (EPI != 0 && mask != 0u) {
                int r = row;
#pragma unroll
                for (int u = 0; u < kChunk; ++u) {
                    if (mask & (1u << u)) {
                        const int rr = ROWIDS ? __builtin_amdgcn_readlane(my_row, c * kChunk + u) : r;
                        ep[u] = SPEX_EPI_LOAD(El + (size_t)rr * 64);
                        ++r;
                    }
                }
            }
            // A chunk that ends rows waits for ALL of its loads here, once.  gfx9 counts loads and stores in one vmcnt and the
            // compiler re-derives its waits at every control-flow join: without this, the wait for a row's epilogue operand
            // (loaded under the mask's branches) became `s_waitcnt vmcnt(0)` placed AFTER the previous row's stores were
            // issued, i.e. every emitted row waited out its predecessor's store round trip.
            if (EPI != 0) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0); expcnt / lgkmcnt unconstrained
            if (EPI == 0 && mask == 0u) {   // no row ends in this chunk (about every second chunk on Epinion2): sixteen fmafs, no per-entry
                                           // test (plain form only: in the epilogue forms the second copy of the loop costs the 65th VGPR)
#pragma unroll
                for (int u = 0; u < kChunk; ++u) {
                    const float xv = (MASKED && (dbits & (1u << u))) ? 0.0f : x[u];
                    acc = fmaf(lane_bcast(my_val, c * kChunk + u), xv, acc);
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < kChunk; ++u) {
                const float xv = (MASKED && (dbits & (1u << u))) ? 0.0f : x[u];     // a dropped entry contributes nothing
                acc = fmaf(lane_bcast(my_val, c * kChunk + u), xv, acc);
                if (mask & (1u << u)) {  // only packs of whole rows carry mask bits
                    emit(ROWIDS ? __builtin_amdgcn_readlane(my_row, c * kChunk + u) : row, acc, ep[u]);
                    acc = 0.0f;
                    ++row;
                }
            }
        }
    }
    if (kind == 0 && t.y == 0 && t.z >= 0) {  // a row without stored entries: y = 0, the epilogue still applies
        float e = 0.0f;
        if (EPI != 0) e = El[(size_t)t.z * 64];
        emit(t.z, 0.0f, e);
    }
    const bool hub = FOLD && kind == 2 && hf_args.tag != 0u;
    if (kind == 2 && !hub) partial[(size_t)(t.w >> 4) * 64 + lane] = acc;  // hub segment: summed by the fix-up launch
    if (t.w & 4) {  // this workgroup combines the segments of rows with 65..1024 entries (and of hub groups) through LDS
        const bool leader = kind == 1 && (t.w & 8);
        const int slot = (t.w >> 4) & 15, nseg = (t.w >> 8) & 31;
        int4 hg = make_int4(0, 0, 0, 0);
        if (hub) hg = hf_args.grp[tid];
        const bool hub_leader = hub && hg.x != 0;
        float e = 0.0f;
        if ((leader || hub_leader) && EPI != 0) e = El[(size_t)t.z * 64];
        if (kind == 1 && !leader) s_part[slot][lane] = acc;
        if (hub && !hub_leader) s_part[wave][lane] = acc;
        __syncthreads();
        if (leader) {
            float y = acc;
            for (int sgi = 1; sgi < nseg; ++sgi) y = y + s_part[slot + sgi][lane];  // segment order: deterministic
            emit(t.z, y, e);
        }
        if (hub_leader) {
            // A hub's segments are adjacent in the table: this wave and the hg.y - 1 behind it hold consecutive segments of row t.z.
            // Their sum is ONE partial row of the hub; the hub's groups meet through memory: agent-scope store (write-through: the
            // groups may run on different XCDs), the wave's own stores acknowledged (explicit vmcnt(0)), a ticket on the hub (an arrival
            // counter, see below) and the LAST group to arrive adds the partial rows in group
            // order (whoever is last: the same order, the same bits) and runs the row's epilogue.  No fence (an agent-scope fence on
            // gfx950 writes back / invalidates the XCD's whole L2), no waiting, no second launch.
            float y = acc;
            for (int i = 1; i < hg.y; ++i) y = y + s_part[wave + i][lane];
            const int2 hf = hf_args.fold[hg.w];
            bool last = true;
            if (hf.y > 1) {
                __hip_atomic_store(partial + (size_t)hg.z * 64 + lane, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the wave's partial-row store is ACKNOWLEDGED (sc1 write-through: visible to every XCD) before lane 0 touches the
                // ticket: an explicit s_waitcnt vmcnt(0) — a workgroup-scope release fence emits no vmcnt wait on gfx950.  One wave =
                // one instruction stream, so the wait covers all 64 lanes' stores.  tests/test_isa_folds.py asserts it in the ISA.
                __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0); expcnt / lgkmcnt unconstrained
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (compiler ordering only)
                // The ticket is a COUNTER that is never reset: the handle zeroes it at creation and every launch that folds adds
                // exactly hf.y arrivals to it (launches on one handle are ordered: they share its scratch), so the group whose
                // increment lands on a multiple of hf.y is the launch's last — ONE read-modify-write round trip through memory on the
                // hub's tail.  (Round 3 carried a per-launch tag in the word: a load, a compare-exchange and often an add — three
                // dependent agent-scope round trips, ~2 us of the hub rows' 5.)  A replay from a captured graph adds hf.y again.
                unsigned long long arrived = 0ull;
                if (lane == 0) arrived = __hip_atomic_fetch_add(hf_args.ticket + hg.w, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(arrived % (unsigned long long)hf.y));
                last = lo == 0u;
                if (last) {
                    y = 0.0f;
                    const float *pr = partial + (size_t)hf.x * 64 + lane;
                    int q = 0;
                    for (; q + 8 <= hf.y; q += 8) {          // eight partial rows in flight (the kernel lives in 64 VGPRs)
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(pr + (size_t)(q + u) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int u = 0; u < 8; ++u) y = y + v[u];
                    }
                    for (; q < hf.y; ++q) y = y + __hip_atomic_load(pr + (size_t)q * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (last) emit(t.z, y, e);
        }
    }
}

// The same kernel for wider embeddings.
// V: consecutive embedding columns per lane — d = 64 V (128, 256; V = 1 is the d == 64 kernel above, kept as its own
// source because its inner-loop schedule is worth ~10 % there and the generic form schedules differently): a gathered
// row is one coalesced 256 V-byte wave load (dwordx2 / dwordx4 per lane).  The batch of gathers in flight shrinks with V (16 / 8 / 4 rows = 4 KB per
// wave either way) so the kernel stays inside 128 VGPRs at 16 waves per workgroup.
template <int V>
struct VecOf {
    typedef float T __attribute__((ext_vector_type(V)));
};

template <int EPI, bool MASKED, bool ROWIDS, int V>
__global__ __launch_bounds__(kWave *kWgWaves) void spmm_chunk_wide_kernel(
    const float *__restrict__ X, const uint32_t *__restrict__ chunk_off, const float *__restrict__ chunk_val,
    const uint32_t *__restrict__ chunk_mask, const int32_t *__restrict__ chunk_row, const int4 *__restrict__ task,
    int n_tasks, float *__restrict__ Y,
    const float *epi_in, float epi_div, float out_div, float *acc_out,   // may alias each other (running sum in place): no __restrict__
    float *__restrict__ partial,
    const DropArgs drop, const int xcd_contiguous)
{
    typedef typename VecOf<V>::T vec;
    constexpr int kRow = kWave * V;          // floats per embedding row
    constexpr int kBatch = kChunk / V;       // gathers issued back to back
    __shared__ float s_part[kWgWaves][kRow];  // segment sums of the rows this workgroup combines
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Workgroups are sorted heaviest-first and dealt round-robin over the 8 XCDs by the dispatcher, which gives every
    // XCD the same mix of weights: right for a graph that streams from HBM (+14 % on Epinion2x538 over handing XCD 0
    // all the heavy workgroups).  A graph whose whole source table sits in cache instead prefers each XCD to walk a
    // contiguous range of workgroups (user rows and item rows of the bipartite graph then gather from different
    // halves of the table in different L2s: 76 % vs 71 % L2 hits on Epinion2).  Placement is a speed matter only.
    const int wg = xcd_contiguous ? xcd_contiguous_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tid = wg * kWgWaves + wave;
    if (tid >= n_tasks) return;  // whole workgroups only (n_tasks is a multiple of kWgWaves)
    const int4 t = task[tid];
    const int kind = t.w & 3;
    const float *__restrict__ Xl = X + lane * V;
    const float *El = (EPI ? epi_in : X) + lane * V;
    int row = t.z;
    vec acc = (vec)(0.0f);

    auto emit = [&](int r, vec y, vec e) {
        const size_t o = (size_t)r * kRow + lane * V;
        // outputs are streamed (non-temporal): they should not evict the gather source from the XCD's L2
        if (EPI == 0) {
            __builtin_nontemporal_store(y, reinterpret_cast<vec *>(Y + o));
        } else if (EPI == 1) {
            if (Y) __builtin_nontemporal_store(y, reinterpret_cast<vec *>(Y + o));
            vec s = e + y;
            if (epi_div != 1.0f) s = s / epi_div;
            __builtin_nontemporal_store(s, reinterpret_cast<vec *>(acc_out + o));
        } else {
            if (epi_div != 1.0f) e = e / epi_div;
            vec s = y + e;
            if (out_div != 1.0f) s = s / out_div;
            __builtin_nontemporal_store(s, reinterpret_cast<vec *>(Y + o));
        }
    };

    // Metadata of up to 4 chunks (64 entries) per vector load — lane k holds entry k — handed to the scalar side with
    // v_readlane.  (Feeding it through s_load instead starves on scalar-cache misses once the matrix streams from
    // HBM: 29 ms vs 13 ms per launch on a 2^23-node graph.)
    const int n_ch = kind == 3 ? 0 : t.y;              // (kind 3: a list of rows without stored entries, handled after the loop)
    for (int sc = 0; sc < n_ch; sc += 4) {
        const int nc = (n_ch - sc < 4) ? n_ch - sc : 4;  // chunks in this super-chunk (wave-uniform)
        uint32_t my_off = 0u, my_mask = 0u, own_off = 0u;
        unsigned long long drop_bits = 0ull;                 // MASKED: entries of this super-chunk that are dropped (wave-uniform)
        int my_row = 0;
        float my_val = 0.0f;
        if (lane < nc * kChunk) {
            const size_t e = (size_t)(t.x + sc) * kChunk + lane;
            my_off = __builtin_nontemporal_load(chunk_off + e);   // metadata is read once per launch
            my_val = __builtin_nontemporal_load(chunk_val + e);
            if (ROWIDS) my_row = __builtin_nontemporal_load(chunk_row + e);
            if (MASKED) {
                const uint32_t eid = __builtin_nontemporal_load(drop.chunk_eid + e);
                bool kept;
                if (drop.mode == 1) {
                    kept = drop.keep[eid] != 0;
                } else {
                    const float u01 = (float)(philox_first(eid, drop.seed_lo, drop.seed_hi) >> 8) * 5.9604644775390625e-8f;
                    kept = (u01 + drop.keep_prob) >= 1.0f;
                }
                my_val = kept ? my_val / drop.keep_prob : 0.0f;   // values[random_index] / keep_prob, model.py:53
                own_off = my_off;
                if (!kept) my_off = 0xFFFFFFFFu;
            }
        }
        if (MASKED) {
            // a dropped entry re-reads the source row of the first KEPT entry of this super-chunk (fetched anyway, so
            // it adds +0 and no traffic); if every entry was dropped each keeps its own row (value 0)
            const bool dropped = my_off == 0xFFFFFFFFu;      // (lanes past the super-chunk hold 0: neither kept nor dropped)
            drop_bits = __ballot(dropped);
            const unsigned long long kept_lanes = __ballot(!dropped && lane < nc * kChunk);
            const int src = kept_lanes ? (int)__builtin_ctzll(kept_lanes) : 0;
            const uint32_t stand_in = (uint32_t)__builtin_amdgcn_readlane((int)my_off, src);   // wave-uniform
            if (dropped) my_off = kept_lanes ? stand_in : own_off;
        }
        if (lane < nc) my_mask = __builtin_nontemporal_load(chunk_mask + t.x + sc + lane);
        for (int c = 0; c < nc; ++c) {
            const uint32_t mask = (uint32_t)__builtin_amdgcn_readlane((int)my_mask, c);
            const uint32_t dbits = MASKED ? (uint32_t)(drop_bits >> (c * kChunk)) & 0xFFFFu : 0u;
#pragma unroll
            for (int u0 = 0; u0 < kChunk; u0 += kBatch) {
                vec x[kBatch], ep[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u)
                    x[u] = *reinterpret_cast<const vec *>(
                        Xl + (size_t)(uint32_t)__builtin_amdgcn_readlane((int)my_off, c * kChunk + u0 + u) * kRow);
                const uint32_t bmask = (mask >> u0) & ((1u << kBatch) - 1u);
                if (EPI != 0 && bmask != 0u) {
                    int r = row;
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) {
                        ep[u] = (vec)(0.0f);
                        if (bmask & (1u << u)) {
                            const int rr = ROWIDS ? __builtin_amdgcn_readlane(my_row, c * kChunk + u0 + u) : r;
                            ep[u] = __builtin_nontemporal_load(reinterpret_cast<const vec *>(El + (size_t)rr * kRow));
                            ++r;
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) ep[u] = (vec)(0.0f);
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const float a = lane_bcast(my_val, c * kChunk + u0 + u);
                    const bool gone = MASKED && (dbits & (1u << (u0 + u)));          // a dropped entry contributes nothing
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] = fmaf(a, gone ? 0.0f : x[u][j], acc[j]);
                    if (bmask & (1u << u)) {  // only packs of whole rows carry mask bits
                        emit(ROWIDS ? __builtin_amdgcn_readlane(my_row, c * kChunk + u0 + u) : row, acc, ep[u]);
                        acc = (vec)(0.0f);
                        ++row;
                    }
                }
            }
        }
    }
    if (kind == 0 && t.y == 0 && t.z >= 0) {  // a row without stored entries: y = 0, the epilogue still applies
        vec e = (vec)(0.0f);
        if (EPI != 0) e = *reinterpret_cast<const vec *>(El + (size_t)t.z * kRow);
        emit(t.z, (vec)(0.0f), e);
    }
    if (ROWIDS && kind == 3) {                // a list of such rows (see the d == 64 kernel)
        const int count = t.w >> 4;
        const int my_r = lane < count ? chunk_row[(size_t)t.x * kChunk + lane] : 0;
        for (int k0 = 0; k0 < count; k0 += kBatch) {    // the rows' epilogue operands in flight together (see the d == 64 kernel)
            vec e[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                e[u] = (vec)(0.0f);
                if (EPI != 0 && k0 + u < count) e[u] = *reinterpret_cast<const vec *>(El + (size_t)__builtin_amdgcn_readlane(my_r, (k0 + u) & 63) * kRow);
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u)
                if (k0 + u < count) emit(__builtin_amdgcn_readlane(my_r, (k0 + u) & 63), (vec)(0.0f), e[u]);
        }
    }
    if (kind == 2) {  // hub segment: summed by the fix-up launch
#pragma unroll
        for (int j = 0; j < V; ++j) partial[(size_t)(t.w >> 4) * kRow + lane * V + j] = acc[j];
    }
    if (t.w & 4) {  // this workgroup combines the segments of rows with 65..1024 entries through LDS
        const bool leader = kind == 1 && (t.w & 8);
        const int slot = (t.w >> 4) & 15, nseg = (t.w >> 8) & 31;
        vec e = (vec)(0.0f);
        if (leader && EPI != 0) e = *reinterpret_cast<const vec *>(El + (size_t)t.z * kRow);
        if (kind == 1 && !leader) {
#pragma unroll
            for (int j = 0; j < V; ++j) s_part[slot][lane + j * kWave] = acc[j];   // column-major per lane: conflict-free
        }
        __syncthreads();
        if (leader) {
            vec y = acc;
            for (int sgi = 1; sgi < nseg; ++sgi) {  // segment order: deterministic
#pragma unroll
                for (int j = 0; j < V; ++j) y[j] = y[j] + s_part[slot + sgi][lane + j * kWave];
            }
            emit(t.z, y, e);
        }
    }
}

// Row-list SpMM (d == 64): y[r] = sum_e val[e] X[col[e]] for the listed rows only, with the running-sum epilogue.
// The exact training step reads the last forward layer at the batch's <= 2 B rows only; computing just those replaces
// a 15 us launch over the whole matrix by a ~6 us one.  One 16-wave workgroup per listed row: wave w takes the row's
// 64-entry segment w (w + 16, ... for rows beyond 1024 entries) straight from the CSR arrays — lane k loads entry k —
// and runs the same 16-gathers-then-16-fmaf batches; segment sums meet in LDS and are added in segment order: the
// main kernel's summation order for every row of up to 1024 entries, i.e. bit-identical results there.
// The list is two index arrays with an offset each (batch users, batch items + n_user_rows), duplicates allowed
// (the same value is stored twice).
__global__ __launch_bounds__(kWave *kWgWaves) void spmm_rowlist_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ X, const int64_t *__restrict__ idx_a, int n_a, int64_t off_a,
    const int64_t *__restrict__ idx_b, int64_t off_b, int32_t n_rows, float *__restrict__ Y,
    const float *acc_in, float *acc_out, float acc_div)   // documented NOT to alias for duplicates, but kept unqualified
{
    __shared__ float s_part[kWgWaves][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t r64 = (int)blockIdx.x < n_a ? idx_a[blockIdx.x] + off_a : idx_b[blockIdx.x - n_a] + off_b;
    if (r64 < 0 || r64 >= n_rows) return;                         // workgroup-uniform: never index out of range
    const int r = (int)r64;
    const int beg = rowptr[r], deg = rowptr[r + 1] - beg;
    const int nseg = (deg + kTaskEntries - 1) / kTaskEntries;
    const float *__restrict__ Xl = X + lane;
    float acc = 0.0f;
    for (int sgi = wave; sgi < nseg; sgi += kWgWaves) {
        const int e0 = beg + sgi * kTaskEntries;
        const int cnt = (deg - sgi * kTaskEntries < kTaskEntries) ? deg - sgi * kTaskEntries : kTaskEntries;
        int my_col = 0;
        float my_val = 0.0f;
        if (lane < cnt) {
            my_col = col[e0 + lane];
            my_val = val[e0 + lane];
        }
        // entries past the segment's end: value 0 on the segment's last real source row (a line already being fetched)
        const int last_col = __builtin_amdgcn_readlane(my_col, (cnt - 1) & 63);
        if (lane >= cnt) my_col = last_col;
        for (int c = 0; c * kChunk < cnt; ++c) {
            float x[kChunk];
#pragma unroll
            for (int u = 0; u < kChunk; ++u)
                x[u] = Xl[(size_t)(uint32_t)__builtin_amdgcn_readlane(my_col, c * kChunk + u) * 64];
#pragma unroll
            for (int u = 0; u < kChunk; ++u) acc = fmaf(lane_bcast(my_val, c * kChunk + u), x[u], acc);
        }
    }
    if (nseg > 1) {
        if (wave != 0 && wave < nseg) s_part[wave][lane] = acc;
        __syncthreads();
    }
    if (wave == 0) {
        float y = acc;
        const int lim = nseg < kWgWaves ? nseg : kWgWaves;
        for (int w = 1; w < lim; ++w) y = y + s_part[w][lane];    // segment order
        const size_t o = (size_t)r * 64 + lane;
        if (Y) Y[o] = y;
        if (acc_out) {
            float s = acc_in[o] + y;
            if (acc_div != 1.0f) s = s / acc_div;
            acc_out[o] = s;
        }
    }
}

// One wave per long row: add its segment partials in order, then the shared epilogue.
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_long_fixup_kernel(const SpmmParams p,
                                                                               const int32_t *__restrict__ rows,
                                                                               const int32_t *__restrict__ seg_lo,
                                                                               const int32_t *__restrict__ seg_hi,
                                                                               int seg_stride, int n_list)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n_list) return;
    const int r = rows[i];
    const int s0 = seg_lo[(size_t)i * seg_stride], s1 = seg_hi[(size_t)i * seg_stride];
    for (int c = lane; c < p.d; c += kWave) {
        float y = 0.0f;
        int s = s0;
        for (; s + 32 <= s1; s += 32) {          // 32 partial rows in flight: a 6 812-entry row has 107 segments — a chain of 27
            float t[32];                         // round trips with four in flight was most of this launch's 6 us
#pragma unroll
            for (int u = 0; u < 32; ++u) t[u] = p.partial[(size_t)(s + u) * p.d + c];
#pragma unroll
            for (int u = 0; u < 32; ++u) y = y + t[u];
        }
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

// Pick the instantiation of the chunk kernel (epilogue form is a template argument of the caller; V = d / 64).
template <int EPI, int V>
void launch_chunk_v(bool masked, bool row_ids, dim3 grid, dim3 block, hipStream_t stream, const float *X, const spex_graph *g,
                    float *Y, const float *epi_in, float epi_div, float out_div, float *acc_out, const DropArgs &da,
                    int xcd_contig, const HubFold &hub)
{
#define SPEX_GO(M, R)                                                                                                  \
    do {                                                                                                               \
        if (V == 1 && (!M || hub.tag != 0u))                                                                           \
            hipLaunchKernelGGL((spmm_chunk_kernel<EPI, M, R, true>), grid, block, 0, stream, X, g->chunk_off,          \
                               g->chunk_val, g->chunk_mask, g->chunk_row, g->task, g->n_tasks, Y, epi_in, epi_div,     \
                               out_div, acc_out, g->partial, da, xcd_contig, hub);                                     \
        else if (V == 1)                                                                                               \
            hipLaunchKernelGGL((spmm_chunk_kernel<EPI, M, R, !M>), grid, block, 0, stream, X, g->chunk_off,            \
                               g->chunk_val, g->chunk_mask, g->chunk_row, g->task, g->n_tasks, Y, epi_in, epi_div,     \
                               out_div, acc_out, g->partial, da, xcd_contig, hub);                                     \
        else                                                                                                           \
            hipLaunchKernelGGL((spmm_chunk_wide_kernel<EPI, M, R, (V == 1 ? 2 : V)>), grid, block, 0, stream, X,       \
                               g->chunk_off, g->chunk_val, g->chunk_mask, g->chunk_row, g->task, g->n_tasks, Y, epi_in, \
                               epi_div, out_div, acc_out, g->partial, da, xcd_contig);                                 \
    } while (0)
    if (masked) {
        if (row_ids) {
            SPEX_GO(true, true);
        } else {
            SPEX_GO(true, false);
        }
    } else {
        if (row_ids) {
            SPEX_GO(false, true);
        } else {
            SPEX_GO(false, false);
        }
    }
#undef SPEX_GO
}

template <int EPI>
void launch_chunk(int d, bool masked, bool row_ids, dim3 grid, dim3 block, hipStream_t stream, const float *X,
                  const spex_graph *g, float *Y, const float *epi_in, float epi_div, float out_div, float *acc_out,
                  const DropArgs &da, int xcd_contig, const HubFold &hub)
{
    if (d == 64) launch_chunk_v<EPI, 1>(masked, row_ids, grid, block, stream, X, g, Y, epi_in, epi_div, out_div, acc_out, da, xcd_contig, hub);
    else if (d == 128) launch_chunk_v<EPI, 2>(masked, row_ids, grid, block, stream, X, g, Y, epi_in, epi_div, out_div, acc_out, da, xcd_contig, hub);
    else launch_chunk_v<EPI, 4>(masked, row_ids, grid, block, stream, X, g, Y, epi_in, epi_div, out_div, acc_out, da, xcd_contig, hub);
}

// Profiling hook: one hipEvent pair around ALL the SpMM launches of an API call (a 3-layer propagation is one bracket
// of three back-to-back launches), so the ~3 us an event pair costs on the stream is shared by the launches it
// brackets instead of being charged to each.
struct TimerBracket {
    spex_timer *tm = nullptr;
    hipStream_t stream;
    TimerBracket(const spex_graph *g, hipStream_t s) : stream(s)
    {
        spex_timer *t = g->timer;
        if (t && !t->open && t->used < (int32_t)t->start.size() && (t->seen++ % t->every) == 0) {
            if (hipEventRecord(t->start[t->used], stream) == hipSuccess) {
                tm = t;
                tm->open = true;
                tm->launches[tm->used] = 0;
            }
        }
    }
    ~TimerBracket()
    {
        if (tm) {
            (void)hipEventRecord(tm->stop[tm->used], stream);
            tm->open = false;
            tm->used++;
        }
    }
};

int launch_spmm(const spex_graph *g, const float *X, float *Y, const float *add_in, float add_div, const float *acc_in,
                float *acc_out, float acc_div, int32_t d, hipStream_t stream, float out_div = 1.0f)
{
    if (g->n_rows == 0) return SPEX_OK;
    SpmmParams p;
    p.rowptr = g->rowptr; p.col = g->col; p.val = g->val; p.edge_id = g->edge_id;
    p.seg_beg = g->seg_beg; p.seg_end = g->seg_end; p.long_row = g->long_row; p.long_seg0 = g->long_seg0;
    p.n_rows = g->n_rows; p.n_seg = g->n_seg; p.n_long = g->n_long;
    p.X = X; p.Y = Y; p.add_in = add_in; p.add_div = add_div; p.out_div = out_div; p.acc_in = acc_in; p.acc_out = acc_out; p.acc_div = acc_div;
    p.partial = g->partial; p.d = d;
    p.mask_mode = g->mask_mode; p.keep = g->keep; p.keep_prob = g->keep_prob;
    p.seed_lo = (uint32_t)g->seed; p.seed_hi = (uint32_t)(g->seed >> 32);

    HubFold hub{nullptr, nullptr, nullptr, 0u};
    const bool masked = g->mask_mode != 0;
    // fast path: d in {64, 128, 256}, chunked table present, and not both epilogues at once
    const bool fast = (d == 64 || d == 128 || d == 256) && g->task != nullptr && !(acc_out && add_in);
    const int per_block = fast ? kWgWaves : kWavesPerBlock;
    const int64_t tasks = fast ? (int64_t)g->n_tasks : (int64_t)g->n_seg + g->n_rows;  // waves
    int64_t blocks = (tasks + per_block - 1) / per_block;
    blocks = (blocks + 7) / 8 * 8;  // multiple of the XCD count so the remap is a bijection
    const dim3 grid((unsigned)blocks), block(kWave * per_block), block4(kWave * kWavesPerBlock);
    spex_timer *tm = g->timer;
    if (tm && tm->open) tm->launches[tm->used]++;
    std::unique_lock<std::mutex> scratch_lock;   // (held until the launches below are queued)
    if (fast ? g->n_hub > 0 : g->n_long > 0) {   // this launch writes the handle's scratch: order it behind its last user
        spex_graph *gm = const_cast<spex_graph *>(g);
        scratch_lock = std::unique_lock<std::mutex>(gm->scratch_mu);
        if (gm->scratch_used && gm->scratch_stream != stream) {
            if (!gm->scratch_ev) SPEX_HIP(hipEventCreateWithFlags(&gm->scratch_ev, hipEventDisableTiming));
            SPEX_HIP(hipEventRecord(gm->scratch_ev, gm->scratch_stream));
            SPEX_HIP(hipStreamWaitEvent(stream, gm->scratch_ev, 0));
        }
        gm->scratch_stream = stream;
        gm->scratch_used = true;
    }
    if (fast) {
        // (rows are moved as float4 where a wave handles several rows per instruction: every table 16-byte aligned — any row of a
        //  contiguous fp32 [*, 64 V] tensor is)
        SPEX_CHECK_ARG(((((uintptr_t)X) | ((uintptr_t)Y) | ((uintptr_t)add_in) | ((uintptr_t)acc_in) | ((uintptr_t)acc_out)) & 15) == 0,
                       "spex_spmm_f32: X / Y / add_in / acc_in / acc_out must be 16-byte aligned");
        const int xcd_contig = ((int64_t)g->n_cols * d * 4 <= (int64_t)16 << 20) ? 1 : 0;  // source table <= 16 MiB
        DropArgs da;
        da.chunk_eid = g->chunk_eid; da.keep = g->keep; da.mode = g->mask_mode; da.keep_prob = g->keep_prob;
        da.seed_lo = (uint32_t)g->seed; da.seed_hi = (uint32_t)(g->seed >> 32);
        // rows beyond kWgRowMax entries: folded inside the d == 64 launch (SPEX_HUB_FOLD=0: by the fix-up launch, as for the wide kernels)
        const char *fold_sw = g->n_hub > 0 ? getenv("SPEX_HUB_FOLD") : nullptr;      // (read per launch: the tests run both forms in one process)
        const bool fold_env = !(fold_sw && fold_sw[0] == '0');
        hub.grp = g->hub_grp; hub.fold = g->hub_fold; hub.ticket = g->hub_ticket;
        hub.tag = (d == 64 && fold_env && g->n_hub > 0 && g->hub_grp && g->hub_ticket) ? 1u : 0u;      // (on / off)
        if (acc_out) launch_chunk<1>(d, masked, g->row_ids, grid, block, stream, X, g, Y, acc_in, acc_div, 1.0f, acc_out, da, xcd_contig, hub);
        else if (add_in) launch_chunk<2>(d, masked, g->row_ids, grid, block, stream, X, g, Y, add_in, add_div, out_div, nullptr, da, xcd_contig, hub);
        else launch_chunk<0>(d, masked, g->row_ids, grid, block, stream, X, g, Y, nullptr, 1.0f, 1.0f, nullptr, da, xcd_contig, hub);
    } else if (masked) {
        hipLaunchKernelGGL((spmm_rows_kernel<true>), grid, block, 0, stream, p);
    } else {
        hipLaunchKernelGGL((spmm_rows_kernel<false>), grid, block, 0, stream, p);
    }
    if (fast) {
        if (g->n_hub > 0 && hub.tag == 0u) {  // only rows longer than kWgRowMax go through global scratch on the fast path
            const dim3 fgrid((unsigned)((g->n_hub + kWavesPerBlock - 1) / kWavesPerBlock));
            hipLaunchKernelGGL(spmm_long_fixup_kernel, fgrid, block4, 0, stream, p, g->hub_row, g->hub_seg0, g->hub_seg0 + 1, 2,
                               g->n_hub);
        }
    } else if (g->n_long > 0) {
        const dim3 fgrid((unsigned)((g->n_long + kWavesPerBlock - 1) / kWavesPerBlock));
        hipLaunchKernelGGL(spmm_long_fixup_kernel, fgrid, block4, 0, stream, p, g->long_row, g->long_seg0, g->long_seg0 + 1, 1,
                           g->n_long);
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
    TimerBracket bracket(g, (hipStream_t)stream);
    return launch_spmm(g, X, Y, add_in, add_div, acc_in, acc_out, acc_div, d, (hipStream_t)stream);
}

extern "C" int spex_spmm_rowlist_f32(const spex_graph_t *g, const float *X, const int64_t *idx_a, int32_t n_a, int64_t off_a,
                                     const int64_t *idx_b, int32_t n_b, int64_t off_b, float *Y, const float *acc_in,
                                     float *acc_out, float acc_div, int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g && X, "spex_spmm_rowlist_f32: NULL graph or X");
    SPEX_CHECK_ARG(n_a >= 0 && n_b >= 0 && (n_a == 0 || idx_a) && (n_b == 0 || idx_b), "spex_spmm_rowlist_f32: bad row lists");
    SPEX_CHECK_ARG(Y || acc_out, "spex_spmm_rowlist_f32: neither Y nor acc_out given");
    SPEX_CHECK_ARG(!acc_out || (acc_in && acc_div != 0.0f), "spex_spmm_rowlist_f32: acc_out needs acc_in and acc_div != 0");
    SPEX_CHECK_ARG(X != Y && X != acc_out, "spex_spmm_rowlist_f32: X must not alias an output");
    if (d != 64 || g->mask_mode != 0) {
        spex::set_error("spex_spmm_rowlist_f32: d == 64 without edge dropout only (d = %d, mask mode %d): use spex_spmm_f32", d,
                        g->mask_mode);
        return SPEX_ERR_UNSUPPORTED;
    }
    if (n_a + n_b == 0 || g->n_rows == 0) return SPEX_OK;
    hipLaunchKernelGGL(spmm_rowlist_kernel, dim3((unsigned)(n_a + n_b)), dim3(kWave * kWgWaves), 0, (hipStream_t)stream, g->rowptr,
                       g->col, g->val, X, idx_a, n_a, off_a, idx_b, off_b, g->n_rows, Y, acc_in, acc_out, acc_div);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
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
    TimerBracket bracket(g, s);
    for (int32_t l = 0; l < L; ++l) {
        const bool last = l == L - 1;
        float *nxt = layers_out ? layers_out + (size_t)l * sz : (last ? nullptr : ws + (size_t)(l & 1) * sz);
        rc = launch_spmm(g, cur, nxt, nullptr, 1.0f, l == 0 ? E0 : mean_out, mean_out, last ? (float)(L + 1) : 1.0f, d, s);
        if (rc) return rc;
        cur = nxt;
    }
    return SPEX_OK;
}

// The propagation with the layer mean LEFT TO THE CONSUMER (the fused BPR step reads the propagated table at its triples' rows
// only): layer 1 in the running-sum form — sum1 = E^0 + E^1, kept apart so that the consumer may update E^0 while it reads —,
// the later layers in the PLAIN form (no epilogue operand, one output stream: 12.3 vs 14.8 us per launch on Epinion2) into the
// two halves of ws, alternating.  tables[0..2]: what the consumer adds, in this order, before dividing by L + 1 — (sum1, E^2,
// E^3) for L = 3, i.e. ((E^0 + E^1) + E^2) + E^3, the order of the fused epilogues.  1 <= L <= 3.  One timer bracket (the
// profiling hook) spans the L launches like spex_propagate_f32's.
int spex::propagate_plain(const spex_graph_t *g, const float *E0, float *sum1, float *ws, int32_t L, int32_t d, void *stream,
                          const float **tables)
{
    SPEX_CHECK_ARG(g && E0 && sum1 && ws && tables, "propagate_plain: NULL argument");
    SPEX_CHECK_ARG(g->n_rows == g->n_cols && L >= 1 && L <= 3, "propagate_plain: square graph, 1 <= L <= 3 (L = %d)", L);
    int rc = ensure_partial(const_cast<spex_graph *>(g), d);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t sz = (size_t)g->n_rows * d;
    TimerBracket bracket(g, s);
    tables[0] = sum1; tables[1] = nullptr; tables[2] = nullptr;
    float *e1 = ws, *e2 = ws + sz;
    rc = launch_spmm(g, E0, L > 1 ? e1 : nullptr, nullptr, 1.0f, E0, sum1, 1.0f, d, s);                     // E^1, sum1 = E^0 + E^1
    if (rc || L == 1) return rc;
    rc = launch_spmm(g, e1, e2, nullptr, 1.0f, nullptr, nullptr, 1.0f, d, s);                                 // E^2
    tables[1] = e2;
    if (rc || L == 2) return rc;
    rc = launch_spmm(g, e2, e1, nullptr, 1.0f, nullptr, nullptr, 1.0f, d, s);                                 // E^3 (E^1 is dead: in sum1)
    tables[2] = e1;
    return rc;
}

// out = in / div over n floats (the mean's backward share), for the other translation units.
int spex::scale_div(const float *in, float *out, float div, int64_t n, void *stream)
{
    if (n <= 0) return SPEX_OK;
    SPEX_CHECK_ARG(in && out && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0, "scale_div: NULL or unaligned pointer");
    const int64_t n4 = n / 4, rem = n % 4;
    const int64_t blocks = (n4 + 255) / 256 < 2048 ? ((n4 + 255) / 256 > 0 ? (n4 + 255) / 256 : 1) : 2048;
    hipLaunchKernelGGL(div_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, div, n4, rem);
    SPEX_HIP(hipGetLastError());
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
    TimerBracket bracket(gt, s);
    if (L >= 1 && ((L + 1) & L) == 0) {
        // L+1 is a power of two: scaling by it commutes with every rounding below, so carry H_l = (L+1) G_l instead
        // (H_L = g, H_l = g + A^T H_{l+1}) and divide once in the last launch's epilogue — bit-identical to the
        // general path, one elementwise pass and one launch fewer.
        const float *cur = g_out;
        for (int32_t l = L - 1; l >= 0; --l) {
            float *nxt = (l == 0) ? grad_E0 : ws + (size_t)(1 + (l & 1)) * sz;
            rc = launch_spmm(gt, cur, nxt, g_out, 1.0f, nullptr, nullptr, 1.0f, d, s, l == 0 ? (float)(L + 1) : 1.0f);
            if (rc) return rc;
            cur = nxt;
        }
        SPEX_HIP(hipGetLastError());
        return SPEX_OK;
    }
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
