// Internal helpers shared by the libspexhip.so translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/spex_hip.h"

namespace spex {

void set_error(const char *fmt, ...);
struct SmallAdam {          // a second, small parameter block updated by the same Adam launch; gradient = sum of n_parts partial blocks
    float *p = nullptr, *m = nullptr, *v = nullptr;
    const float *parts = nullptr;
    int32_t n_parts = 0;
    int64_t stride = 0;
    int32_t n = 0;
};
int adam_step_z2(float *p, const float *g, float *m, float *v, int64_t n, int32_t t, float lr, float beta1, float beta2, float eps,
                 float *zero_buf, float *zero_buf2, void *stream, const float *loss_rows = nullptr, int32_t n_loss = 0,
                 float *loss_sum = nullptr, const SmallAdam *small = nullptr, const float *add_g = nullptr, float add_div = 1.0f);
                                               // add_g: gradient = g + add_g / add_div (add_g may be zero_buf: read, then cleared);   // optim.hip: spex_adam_step_f32 with a second buffer to clear and the step's
                                               // per-sample losses to fold into an accumulator

int dual_task_adam(float *p, float *m, float *v, const float *g_E0, float *g_raw, float *g_user, float *g_small, float *g_prop,
                   float *push_zero, float *loss, float *loss_acc, float *prec, int64_t n_table, int64_t n_user, int64_t n_trust,
                   int32_t B, int32_t T, int32_t n_rec, int32_t t, float lr, float beta1, float beta2, float eps, int fixed_weights,
                   void *stream, float prop_div = 0.0f, float *att_copies = nullptr, int32_t n_att_copies = 0,
                   int32_t att_clear = 0, int32_t part = 0, int32_t clear_prop = 1, float *zero_a = nullptr, int64_t n_zero_a = 0,
                   float *zero_b = nullptr, int64_t n_zero_b = 0);           // optim.hip: Adam over the dual-task parameter arena (spex_dual_task_step_f32)

// trust.hip: spex_trust_head_train_f32 with a cap on the fused kernel's workgroups per path (the public entry: 8; beside other work: 1)
int trust_head_train(const float *table, int64_t n_rows, const float *params, const int64_t *seq, const int64_t *seq_l,
                     const int64_t *targets, int32_t B, int32_t L, int32_t d, int32_t n_heads, int32_t hybrid, float scale,
                     const float *scale_dev, float *a2, float *dscore, float *loss_b, float *ws, float *loss_out, int32_t loss_accumulate,
                     float *grad_params, float *grad_table, int32_t split_cap, void *stream);

// batch.hip: the batch kernels with the layer sum formed at the batch's rows from up to three tables (acc_in + acc2 + acc3, in that
// order; NULL = absent) — what lets the one-call steps run their forward layers in the plain form
int lightgcn_batch_layers(const spex_graph_t *g, const float *X, const float *acc_in, const float *acc2, const float *acc3, float acc_div,
                          const int64_t *users, const int64_t *items, const float *labels, int32_t B, int32_t n_user_rows, float grad_scale,
                          float push_scale, float *loss_sum, float *loss_per_sample, float *g_out, float *G, int32_t d, void *stream);
int lightgcn_batch_slots_layers(const spex_graph_t *g, const float *X, const float *acc_in, const float *acc2, const float *acc3,
                                float acc_div, const int64_t *users, const int64_t *items, const float *labels, int32_t B,
                                int32_t n_user_rows, float grad_scale, float *loss_sum, float *loss_per_sample, float *grad_slots, int32_t d,
                                void *stream);
int gated_batch_fwd_layers(const spex_graph_t *g, const float *X, const float *acc_in, const float *acc2, const float *acc3, float acc_div,
                           const float *raw, const float *att_u, const float *att_i, const int64_t *users, const int64_t *items,
                           const float *labels, int32_t B, int32_t n_user_rows, float grad_scale, float *loss_sum, float *loss_per_sample,
                           float *lo_batch, float *grad_slots, int32_t d, void *stream);
// (the rec branch's whole batch-sized middle — forward, gate, scores, the gate's backward, push — in one launch: batch.hip)
int gated_batch_push_layers(const spex_graph_t *g, const float *X, const float *acc_in, const float *acc2, const float *acc3, float acc_div,
                            const float *raw, const float *att_u, const float *att_i, const int64_t *users, const int64_t *items,
                            const float *labels, int32_t B, int32_t n_user_rows, float grad_scale, float push_scale, float *loss_sum,
                            float *g_prop, float *G, float *g_raw, float *g_att, int32_t n_att_copies, int32_t d, void *stream);
// (the batch-sized middle of the partitioned one-call steps on the batch's compact replicated rows — [gate,] scores, [the gate's
//  backward,] owner-computes adds, and the first backward product pushed through the rank's own columns of A: batch.hip;
//  rows_raw == NULL: the LightGCN form; push == NULL: a rank without rows)
int rows_train_push(const spex_graph_t *push, const float *rows_raw, const float *rows_prop, const float *att_u, const float *att_i,
                    const int64_t *pos, int64_t lo, int32_t n_local, const float *labels, int32_t B, float grad_scale, float push_scale,
                    float *loss_sum, float *g_prop, float *P, float *g_raw, float *g_att, int32_t n_att_copies, void *stream);
int score_bce_slots_rows(const float *users, const float *items, int32_t ldu, int32_t ldi, int64_t n_user_rows, int64_t n_item_rows,
                         const int64_t *u_idx, const int64_t *i_idx, const float *labels, int32_t B, int32_t d, float *loss_rows,
                         float grad_scale, float *grad_slots, int32_t ld_slots, void *stream);     // score.hip: per-sample rows + losses
int propagate_plain(const spex_graph_t *g, const float *E0, float *sum1, float *ws, int32_t L, int32_t d, void *stream,
                    const float **tables);    // spmm.hip: L launches, the layer mean left to the consumer (tables[0..2], L + 1)
int bpr_sgd_layers(const float *t0, const float *t1, const float *t2, float div, float *table_w, int64_t n_user_rows, int64_t n_item_rows,
                   const int64_t *u, const int64_t *i_pos, const int64_t *i_neg, int64_t T, float lr, float reg, float *loss_sum,
                   void *stream);             // score.hip: the fused BPR-SGD kernel reading rows as ((t0 + t1) + t2) / div
int scale_div(const float *in, float *out, float div, int64_t n, void *stream);                          // spmm.hip: out = in / div
int zero_f32(float *x, int64_t n, void *stream);                                                       // rows.hip: x[0 : n] = 0 (a kernel launch)
int sum_ordered(const float *x, int32_t n, float scale, float *out, int accumulate, void *stream);     // rows.hip: fixed-order sum
int sum_parts(const float *parts, int32_t n_parts, int64_t stride, int32_t n, float *out, int accumulate,
              void *stream);                                                                        // rows.hip: partial blocks, in order

#define SPEX_CHECK_ARG(cond, ...)             \
    do {                                      \
        if (!(cond)) {                        \
            ::spex::set_error(__VA_ARGS__);   \
            return SPEX_ERR_INVALID;          \
        }                                     \
    } while (0)

#define SPEX_HIP(call)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            ::spex::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SPEX_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kWavesPerBlock = 4;  // 256-thread workgroups
constexpr int kLongRow = 128;      // generic kernel / hub path: rows with more stored entries are cut into segments
constexpr int kSegLen = 64;        // entries per such segment (partial rows go through global scratch + the fix-up launch); 64 = one
                                   // wave task of the d == 64 kernel: a hub segment of 128 made its wave run twice as long as every other
                                   // wave of a one-round launch, i.e. set the launch time (Weibo-shaped graph: 19.8 us per product)
constexpr int kTaskEntries = 64;   // entries per wave task of the d == 64 kernel (4 chunks: 4 memory round trips)
constexpr int kWgWaves = 16;       // the d == 64 kernel runs 1024-thread workgroups = 16 wave tasks
constexpr int kWgRowMax = kWgWaves * kTaskEntries;  // longest row whose segments are combined inside one workgroup
constexpr int kOpenTasks = 4;      // first-fit packing of short rows keeps this many tasks open
constexpr int kChunk = 16;         // entries per chunk = gathers a wave keeps in flight
constexpr int kTileRows = 64;      // rows a workgroup completes in tile mode (4 MFMA tiles of 16)
constexpr int kSoftmaxTile = 2048; // stored entries per workgroup of the row-softmax kernels (512 on small graphs)

}  // namespace spex

// Cross-lane sums on the vector ALU's DPP path.  `__shfl_xor` compiles to ds_bpermute_b32 — an LDS-pipe instruction, one
// pipe per CU shared by four SIMDs — and a 64-lane butterfly is six of them per value: the scoring kernels' dot
// products were bound by that pipe, not by their gathers (2 M dot products of a 2^20-triple BPR batch: 184 us with the
// gathers removed).  These use DPP row shifts / broadcasts instead (no LDS traffic).
namespace spex {
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float v)   // source lane per CTRL; lanes without a valid source read 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true));
}

// Sum over the 64 lanes, returned to every lane (the value is wave-uniform: it comes back through an SGPR).
__device__ __forceinline__ float wave_sum_f32(float v)
{
    v += dpp_f32<0xb1>(v);          // quad_perm [1,0,3,2]
    v += dpp_f32<0x4e>(v);          // quad_perm [2,3,0,1]   every lane: its quad's sum
    v += dpp_f32<0x114>(v);         // row_shr:4
    v += dpp_f32<0x118>(v);         // row_shr:8             lane 15 of each row: the row's sum
    v += dpp_f32<0x142, 0xa>(v);    // row_bcast:15 into rows 1 and 3
    v += dpp_f32<0x143, 0xc>(v);    // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Sum over each row of 16 lanes, returned to every lane of the row (rotations within the row).
__device__ __forceinline__ float row16_sum_f32(float v)
{
    v += dpp_f32<0x128>(v);         // row_ror:8
    v += dpp_f32<0x124>(v);         // row_ror:4
    v += dpp_f32<0x122>(v);         // row_ror:2
    v += dpp_f32<0x121>(v);         // row_ror:1
    return v;
}
}  // namespace spex

namespace spex {
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

// Edge dropout (model.py:46-55) as the kernels see it: mode 1 = a device keep mask indexed by edge id, mode 2 = the counter-based
// draw keyed by the edge id; a kept value is divided by keep_prob, a dropped entry contributes nothing.
struct EdgeDrop {
    const uint8_t *keep;
    const int32_t *edge_id;      // entry -> edge id (NULL: the entry index itself)
    int mode;
    float keep_prob;
    uint32_t seed_lo, seed_hi;
};
__device__ __forceinline__ bool edge_kept(const EdgeDrop &dr, int entry)
{
    const uint32_t eid = (uint32_t)(dr.edge_id ? dr.edge_id[entry] : entry);
    if (dr.mode == 1) return dr.keep[eid] != 0;
    const float u01 = (float)(philox_first(eid, dr.seed_lo, dr.seed_hi) >> 8) * 5.9604644775390625e-8f;
    return (u01 + dr.keep_prob) >= 1.0f;
}
}  // namespace spex

// A batch as two device index lists with offsets (users; items + n_user_rows): slot k's row.
namespace spex {
__device__ __forceinline__ long long batch_row(const int64_t *idx_a, int n_a, int64_t off_a, const int64_t *idx_b, int64_t off_b,
                                               int k)
{
    return k < n_a ? idx_a[k] + off_a : idx_b[k - n_a] + off_b;
}

}  // namespace spex

struct spex_timer {
    std::vector<hipEvent_t> start, stop;
    int32_t used = 0;
    std::vector<int32_t> launches;  // main-kernel launches inside each bracket
    int32_t every = 1;   // bracket every `every`-th API call (spmm / propagate / propagate_bwd)
    int64_t seen = 0;    // API calls seen since the last reset
    bool open = false;   // a bracket is open (launch_spmm counts its launches instead of bracketing them itself)
};

// The opaque handle.  All pointers are device memory owned by the handle.
struct spex_graph {
    int32_t n_rows = 0, n_cols = 0;
    int64_t nnz = 0;
    int32_t *rowptr = nullptr;   // [n_rows+1]
    int32_t *col = nullptr;      // [nnz]
    float *val = nullptr;        // [nnz]
    int32_t *edge_id = nullptr;  // [nnz] or null (identity)
    int64_t max_edge_id = -1;    // largest edge id (nnz - 1 for identity)
    int32_t *row_of = nullptr;   // [nnz] row of each stored entry, built by the first spex_sddmm_f32
    int32_t tile = 0;            // stored entries per row-softmax workgroup: kSoftmaxTile, or 512 when that fills < 2 k workgroups
    int32_t n_tiles = 0;         // ceil(nnz / tile)
    int32_t *tile_row = nullptr; // [n_tiles+1] first row starting at or after entry t * tile
    // long rows (> kLongRow entries): segment table + per-segment partial rows + per-row fix-up table
    int32_t n_long = 0, n_seg = 0;
    int32_t *seg_beg = nullptr;   // [n_seg] first entry of the segment
    int32_t *seg_end = nullptr;   // [n_seg]
    int32_t *long_row = nullptr;  // [n_long] row index
    int32_t *long_seg0 = nullptr; // [n_long+1] first segment of each long row
    float *partial = nullptr;     // [n_seg * d_cap] scratch, grown on demand
    int64_t partial_cap = 0;      // floats
    // The scratch is the one piece of a handle that launches WRITE.  Calls on one stream are ordered by the stream; a call
    // on another stream first waits (event) for everything queued on the stream that used the scratch last — so two streams
    // may share a handle, and so may two HOST threads: a launch that writes the scratch holds scratch_mu from the ordering
    // decision until its kernels are queued (otherwise the other thread's event could be recorded in front of them).
    std::mutex scratch_mu;
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;
    hipEvent_t scratch_ev = nullptr;
    // Chunked copy of the matrix for the d == 64 kernel (built when n_cols * 256 B fits a 32-bit buffer offset).
    // A chunk is 16 stored entries: byte offsets of their source rows (col * 256), their values, and a 16-bit mask
    // marking entries that end an output row.  A wave TASK is a run of <= 4 chunks (64 entries): whole consecutive
    // short rows, or one 64-entry segment of a row with 65..1024 entries — all segments of such a row sit in ONE
    // 16-wave workgroup and are summed through LDS in segment order — or (rows > 1024 entries only) a 128-entry
    // segment whose partial row goes through global scratch and the fix-up launch.  Tasks are padded to whole chunks
    // with value-0 entries on the task's last real source row (a line already being fetched); a chunk's padding sits at
    // its end (count in chunk_pad).
    //   task.x = first chunk, .y = number of chunks, .z = (first) row or -1, .w = kind | flags (see graph.hip)
    int32_t n_tasks = 0;
    int4 *task = nullptr;          // [n_tasks], heaviest first
    int64_t n_chunks = 0;
    uint32_t *chunk_off = nullptr; // [n_chunks * 16]
    float *chunk_val = nullptr;    // [n_chunks * 16]
    uint32_t *chunk_mask = nullptr; // [n_chunks] end-of-row flags (16 bits)
    uint8_t *chunk_pad = nullptr;   // [n_chunks] number of padding entries at the end of the chunk
    uint32_t *chunk_eid = nullptr;  // [n_chunks * 16] edge id of each entry (keep-mask index); read only under dropout
    bool row_ids = false;           // tasks pack non-adjacent rows (cache-resident graphs): the kernel reads chunk_row
    int32_t *chunk_row = nullptr;   // [n_chunks * 16] output row of each entry, only when row_ids
    // Tile mode (spex_graph_create_ex, SPEX_GRAPH_TILE_ROWS): every workgroup of the task table completes at most kTileRows rows,
    // task.w bits 16-23 = the workgroup-local slot of the task's first completed row — what the fused NGCF layer kernel needs
    // to hold a workgroup's rows in LDS.  0 = ordinary table.
    int32_t tile_rows = 0;
    int32_t n_wgs = 0;              // workgroups of the task table (tile mode)
    int32_t *wg_rows = nullptr;     // [n_wgs] rows each workgroup completes (tile mode)
    int32_t n_hub = 0;              // rows longer than kWgRowMax (global-scratch path of the d == 64 kernel)
    int32_t *hub_row = nullptr;     // [n_hub]
    int32_t *hub_seg0 = nullptr;    // [2 * n_hub] (first, one-past-last) segment of each hub in the kLongRow segment table
    // In-kernel fold of the hubs (d == 64 chunk kernel): the hub segments lead the task table, so 16 consecutive segments of a
    // hub share a workgroup — a GROUP, summed through LDS by its first wave, which publishes one partial row and takes a ticket
    // on the hub; the hub's last group to arrive folds the groups' partials in group order and runs the epilogue.
    int32_t n_hub_tasks = 0;        // hub segment tasks = task[0 : n_hub_tasks]
    int4 *hub_grp = nullptr;        // [n_hub_tasks] x: 1 = the group's first wave, y: waves in the group, z: the group's partial row, w: hub
    int2 *hub_fold = nullptr;       // [n_hub] x: the hub's first partial row, y: its number of groups
    unsigned long long *hub_ticket = nullptr;   // [n_hub] arrival counter: every folding launch adds the hub's group count (zeroed at creation, never reset)
    // edge dropout
    int mask_mode = 0;
    const uint8_t *keep = nullptr;
    float keep_prob = 1.0f;
    uint64_t seed = 0;
    spex_timer *timer = nullptr;  // profiling hook (not owned)
};
