// The batch-sized middle of the exact LightGCN training step as ONE launch.
//
// Between the dense forward layers and the dense backward layers the step only touches the batch's rows
// (LightGCN_SPEX/code/utility1/model.py:91-97,111-121 forward, autograd backward): the last layer at the 2B rows of the
// batch, the layer mean there, B dot products + BCE, the 2B gradient rows, and the first backward product A^T g in push
// form.  Issued as three launches (spex_spmm_rowlist_f32 -> spex_score_bce_slots_f32 -> spex_spmm_push_batch_f32) each
// costs a ramp plus its own chain of dependent round trips: 6.1 + 5.2 + 9 us for ~14 k gathers, 256 dot products and
// ~14 k row atomics on Epinion2.  spex_lightgcn_batch_f32 does the three in one kernel:
//
//   workgroup (sample b, part p), 16 waves
//     1. last layer at both rows of the sample — each row's 64-entry segments go to virtual waves v = segment mod 16 exactly
//        as the row-list kernel deals them (one accumulator chain per virtual wave), the two rows' virtual waves numbered
//        jointly and dealt to the 16 waves; a segment's 64 gathers are all in flight at once; segment sums meet in LDS and
//        are added in segment order: bit-identical to spex_spmm_rowlist_f32 / the main kernel for rows of <= 1024 entries;
//     2. light = (running sum + y) / (L + 1) for both rows -> LDS; every wave forms x = <light_u, light_i>, the sample's loss
//        and both gradient rows g_u = dg * light_i, g_i = dg * light_u, dg = (sigmoid(x) - label) / B;
//     3. push: out[col[e]] += val[e] * g / (L + 1) over the stored entries of both rows OF A (A^T g in push form walks the rows
//        of A: (A^T g)[c] = sum_r A[r, c] g[r]), runs of 16 entries numbered jointly and dealt over (part, wave) — their
//        (col, val) pairs were requested before step 1 —, plus out[row] += g / (L + 1) and the dense d loss / d light rows
//        (part 0 only).
//   PUSH == false (spex_lightgcn_batch_slots_f32, the deterministic step): steps 1 and 2 only; the sample's two gradient rows
//   leave with plain stores as grad_slots[b] (user side) and grad_slots[B + b] (item side) — no float atomics anywhere.
//   Up to kBatchParts = 3 workgroups share a sample when its rows are long (more than kBatchRunsPerPart = 16 runs per part) — each repeats the cheap, L2-resident forward, the parts of a shorter sample
//   leave at once.  The longest sample's push sets the launch time, and the reference's training batches (a random observed pair
//   or one of its five same-user negatives: users and positive items arrive in proportion to their degree) carry rows of ~1 000
//   entries all the time.  Measured on Epinion2, B = 256, us per step on uniform / training-shaped batches: 1 part 81.1 / 86.5,
//   2 parts (16 runs each) 83.6 / 84.7, 3 parts 81.1 / 83.2, 4 parts 89.5 / 91.4 (1 024 workgroups of 1 024 threads).
#include "spex_common.h"

using namespace spex;

constexpr int kBatchParts = 3, kBatchRunsPerPart = 16;      // measured: see the table above

namespace {

__device__ __forceinline__ float lane_bcast(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

#ifdef SPEX_STAMPS   // debug build only (tools/batch_stamps.py): phase stamps of workgroup 0 / wave 0, wall_clock64 = 100 MHz
__device__ unsigned long long g_stamps[16];
#define STAMP(k)                                                           \
    do {                                                                   \
        if (blockIdx.x == 8 && threadIdx.x == 0) g_stamps[k] = wall_clock64(); \
    } while (0)
#else
#define STAMP(k)
#endif

constexpr int kPre = 2;                      // push runs a wave loads ahead (2 x 16 waves x 16 entries = 512 entries per part)

// One 64-entry segment [e0, e0 + cnt) of a row, all of its gathers in flight at once (one workgroup per CU here: the wave
// can afford 64 result registers, and the kernel is a latency chain — four dependent gather round trips became one).
// The fmaf chain runs in entry order exactly like the row-list kernel's.
__device__ __forceinline__ float segment_sum(const int32_t *__restrict__ col, const float *__restrict__ val, const float *__restrict__ Xl,
                                             int e0, int cnt, int lane, float acc)
{
    int my_col = 0;
    float my_val = 0.0f;
    if (lane < cnt) {
        my_col = col[e0 + lane];
        my_val = val[e0 + lane];
    }
    const int last_col = __builtin_amdgcn_readlane(my_col, (cnt - 1) & 63);
    if (lane >= cnt) my_col = last_col;                    // padding: value 0 on a row already being fetched
    float x[4][kChunk];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c * kChunk < cnt) {
#pragma unroll
            for (int k = 0; k < kChunk; ++k)
                x[c][k] = Xl[(size_t)(uint32_t)__builtin_amdgcn_readlane(my_col, c * kChunk + k) * kWave];
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c * kChunk < cnt) {
#pragma unroll
            for (int k = 0; k < kChunk; ++k) acc = fmaf(lane_bcast(my_val, c * kChunk + k), x[c][k], acc);
        }
    }
    return acc;
}

// The same segment under edge dropout (model.py:46-55): a kept entry's value is divided by keep_prob, a dropped entry contributes
// fmaf(0, 0, acc) = acc — exactly what spmm_chunk_kernel<.., MASKED> makes of it (dropped entries keep their place in the chain, so
// the row's sum is the masked main kernel's, bit for bit).
__device__ __forceinline__ float segment_sum_masked(const int32_t *__restrict__ col, const float *__restrict__ val,
                                                    const float *__restrict__ Xl, int e0, int cnt, int lane, float acc,
                                                    const spex::EdgeDrop &dr)
{
    int my_col = 0;
    float my_val = 0.0f;
    bool dropped = false;
    if (lane < cnt) {
        my_col = col[e0 + lane];
        const bool kept = spex::edge_kept(dr, e0 + lane);
        my_val = kept ? val[e0 + lane] / dr.keep_prob : 0.0f;
        dropped = !kept;
    }
    const unsigned long long dbits = __ballot(dropped);
    const int last_col = __builtin_amdgcn_readlane(my_col, (cnt - 1) & 63);
    if (lane >= cnt) my_col = last_col;
    float x[4][kChunk];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c * kChunk < cnt) {
#pragma unroll
            for (int k = 0; k < kChunk; ++k)
                x[c][k] = Xl[(size_t)(uint32_t)__builtin_amdgcn_readlane(my_col, c * kChunk + k) * kWave];
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c * kChunk < cnt) {
#pragma unroll
            for (int k = 0; k < kChunk; ++k) {
                const float xv = (dbits >> (c * kChunk + k)) & 1ull ? 0.0f : x[c][k];     // a dropped entry contributes nothing
                acc = fmaf(lane_bcast(my_val, c * kChunk + k), xv, acc);
            }
        }
    }
    return acc;
}

template <bool PUSH>
__global__ __launch_bounds__(kWave *kWgWaves) void lightgcn_batch_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val, int n_rows,
    int n_user_rows, const float *__restrict__ X, const float *__restrict__ acc_in, float acc_div,
    const int64_t *__restrict__ users, const int64_t *__restrict__ items, const float *__restrict__ labels, int parts,
    float grad_scale, float push_scale, float *loss_sum, float *__restrict__ loss_rows, float *g_out, float *G, int runs_per_part,
    float *__restrict__ grad_slots, int B, const float *__restrict__ acc2, const float *__restrict__ acc3, const spex::EdgeDrop drop)
{
    // (the push walks the same rows of the same matrix as the forward: t_* are aliases kept for readability)
    const int32_t *__restrict__ t_rowptr = rowptr, *__restrict__ t_col = col;
    const float *__restrict__ t_val = val;
    __shared__ float s_part[2][kWgWaves][kWave];   // [row: user, item][virtual wave of the row-list kernel][column]
    __shared__ float s_light[2][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x / parts, part = blockIdx.x % parts;
    STAMP(0);
    const int64_t u64 = users[b], i64 = items[b];
    if (u64 < 0 || u64 >= n_user_rows || i64 < 0 || i64 + n_user_rows >= n_rows) {   // workgroup-uniform: never gather out of range
        if (loss_rows && part == 0 && threadIdx.x == 0) loss_rows[b] = 0.0f;
        if (!PUSH && wave < 2) grad_slots[(size_t)(wave * B + b) * kWave + lane] = 0.0f;
        return;
    }
    const int row[2] = {(int)u64, (int)i64 + n_user_rows};
    // everything whose address is known now is requested now: both rows' ranges in both matrices, the label, the running sums
    const int beg[2] = {rowptr[row[0]], rowptr[row[1]]};
    const int deg[2] = {rowptr[row[0] + 1] - beg[0], rowptr[row[1] + 1] - beg[1]};
    const int t_beg[2] = {t_rowptr[row[0]], t_rowptr[row[1]]};
    const int t_len[2] = {t_rowptr[row[0] + 1] - t_beg[0], t_rowptr[row[1] + 1] - t_beg[1]};
    const float y_lab = labels[b];
    // acc_in is the running layer sum — or, when the forward layers ran in the plain form (the one-call step: no epilogue operand,
    // no second output stream per launch), E^0 with acc2 / acc3 the later layers' tables: the sum is formed HERE, at the batch's
    // rows only, in the same order ((E^0 + E^1) + E^2) + y as the fused epilogues form it — bit-identical
    float run = 0.0f;
    if (wave < 2) {
        const size_t o = (size_t)row[wave] * kWave + lane;
        run = acc_in[o];
        float r2 = 0.0f, r3 = 0.0f;
        if (acc2) r2 = acc2[o];
        if (acc3) r3 = acc3[o];
        if (acc2) run = run + r2;
        if (acc3) run = run + r3;
    }
    // the push's runs of 16 entries, both rows' runs numbered jointly and dealt over (part, wave); the first kPre of this wave
    // are loaded here, ahead of the forward, so that their round trip is off the chain
    const int n_run0 = (t_len[0] + 15) >> 4, n_runs = n_run0 + ((t_len[1] + 15) >> 4);
    // A wave issues one 256-byte row atomic per ~150 ns, so a sample with 600 stored entries pushes for ~6 us (38 runs over 16
    // waves) while the median sample is done in 2.4.  Only such samples are shared: part p of a sample stays if the sample has more
    // than runs_per_part * p runs — the others leave here, before the forward — and the active parts split the runs.
    const int want = (n_runs + runs_per_part - 1) / runs_per_part;
    const int act = !PUSH ? 1 : (want < parts ? (want < 1 ? 1 : want) : parts);
    if (part >= act) return;
    const int q_step = act * kWgWaves;
    int q = part * kWgWaves + wave;
    int p_col[kPre], p_cnt[kPre], p_side[kPre];
    float p_val[kPre];
    auto load_runs = [&](int q0) {
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            const int qq = q0 + p * q_step;
            p_col[p] = 0;
            p_val[p] = 0.0f;
            p_cnt[p] = 0;
            p_side[p] = 0;
            if (qq < n_runs) {
                const int side = qq >= n_run0, rr = side ? qq - n_run0 : qq;
                const int base = t_beg[side] + rr * 16, left = t_len[side] - rr * 16;
                p_side[p] = side;
                p_cnt[p] = left < 16 ? left : 16;
                if (lane < p_cnt[p]) {
                    p_col[p] = t_col[base + lane];
                    p_val[p] = t_val[base + lane];
                    if (drop.mode != 0) p_val[p] = spex::edge_kept(drop, base + lane) ? p_val[p] / drop.keep_prob : 0.0f;   // the forward's mask
                }
            }
        }
    };
    if (PUSH) load_runs(q);
    STAMP(1);
    // ---- 1. last layer at both rows.  Row k's segments go to its virtual waves v = segment mod 16 as in the row-list kernel
    //         (one accumulator chain per virtual wave); the two rows' virtual waves are numbered jointly and dealt to the 16 waves.
    const int nseg[2] = {(deg[0] + kTaskEntries - 1) / kTaskEntries, (deg[1] + kTaskEntries - 1) / kTaskEntries};
    const int nv0 = nseg[0] < kWgWaves ? nseg[0] : kWgWaves, nv = nv0 + (nseg[1] < kWgWaves ? nseg[1] : kWgWaves);
    const float *__restrict__ Xl = X + lane;
    for (int j = wave; j < nv; j += kWgWaves) {
        const int side = j >= nv0, v = side ? j - nv0 : j;
        float acc = 0.0f;
        for (int sgi = v; sgi < nseg[side]; sgi += kWgWaves) {
            const int left = deg[side] - sgi * kTaskEntries;
            if (drop.mode != 0)
                acc = segment_sum_masked(col, val, Xl, beg[side] + sgi * kTaskEntries, left < kTaskEntries ? left : kTaskEntries, lane, acc, drop);
            else
                acc = segment_sum(col, val, Xl, beg[side] + sgi * kTaskEntries, left < kTaskEntries ? left : kTaskEntries, lane, acc);
        }
        s_part[side][v][lane] = acc;
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    // ---- 2. layer mean at both rows (waves 0 and 1), the score, both gradient rows
    if (wave < 2) {
        const int lim = nseg[wave] < kWgWaves ? nseg[wave] : kWgWaves;
        float y = lim > 0 ? s_part[wave][0][lane] : 0.0f;
        for (int w = 1; w < lim; ++w) y = y + s_part[wave][w][lane];          // segment order
        float s = run + y;
        if (acc_div != 1.0f) s = s / acc_div;
        s_light[wave][lane] = s;
    }
    __syncthreads();
    STAMP(4);
    const float lu = s_light[0][lane], li = s_light[1][lane];
    const float x = wave_sum_f32(fmaf(lu, li, 0.0f));
    const float dg = (sigmoid_f(x) - y_lab) * grad_scale;
    const float g2[2] = {dg * li, dg * lu};                                  // d loss / d light at the user row, at the item row
    if (part == 0 && wave < 2) {
        if (wave == 0 && lane == 0) {
            const float bce = fmaxf(x, 0.0f) - x * y_lab + log1pf(expf(-fabsf(x)));
            if (loss_rows) loss_rows[b] = bce;
            else atomicAdd(loss_sum, bce);
        }
        if (!PUSH) {
            grad_slots[(size_t)(wave * B + b) * kWave + lane] = g2[wave];     // per-sample rows; summed per table row in slot order later
        } else {
            if (g_out) atomicAdd(g_out + (size_t)row[wave] * kWave + lane, g2[wave]);   // dense d loss / d light_out (rows may repeat in a
                                                                                         // batch); NULL: nobody reads it (L == 3 step)
            atomicAdd(G + (size_t)row[wave] * kWave + lane, push_scale * g2[wave]);   // the `g` of (g + A^T g) / (L + 1)
        }
    }
    if (!PUSH) return;
    // ---- 3. push over both rows' entries in A^T.  Lane 0 of every loaded run is read before the first atomic and the entry
    //         loop is a real loop (rows.hip explains why: one vmcnt for loads and atomics).
    float *out_l = G + lane;
    STAMP(5);
    for (;;) {
        int c0[kPre];
        float v0[kPre];
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            c0[p] = __builtin_amdgcn_readlane(p_col[p], 0);
            v0[p] = lane_bcast(p_val[p], 0);
        }
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            const float gs = push_scale * (p_side[p] ? g2[1] : g2[0]);
            if (p_cnt[p] > 0) atomicAdd(out_l + (size_t)c0[p] * kWave, v0[p] * gs);
#pragma unroll 1
            for (int j = 1; j < p_cnt[p]; ++j) {
                const int c = __builtin_amdgcn_readlane(p_col[p], j);
                const float v = lane_bcast(p_val[p], j);
                atomicAdd(out_l + (size_t)c * kWave, v * gs);
            }
        }
        q += kPre * q_step;
        if (q >= n_runs) break;
        load_runs(q);
    }
    STAMP(6);
}


// The dual-task model's rec branch has the expert gate between the layer mean and the score (utility1/model_expert_s.py:154-168),
// so its batch-sized middle cannot include the push (the gate's backward comes first).  Its FORWARD half is one launch of the
// same shape: last layer at the sample's two rows (step 1 above) -> layer mean (written to lo_batch[row]: the gate's backward
// reads it) -> waves 0 / 1 gate the user / item row (softmax([raw | light] att) two-way mix) -> score, loss, and the two
// per-sample gradient rows with respect to the GATED rows (grad_slots[b], grad_slots[B + b]) — what spex_spmm_rowlist_f32 ->
// spex_expert_gate_rows_f32 -> spex_score_bce_slots_f32 computed in three launches.
__global__ __launch_bounds__(kWave *kWgWaves) void gated_batch_fwd_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val, int n_rows, int n_user_rows,
    const float *__restrict__ X, const float *__restrict__ acc_in, float acc_div, const float *__restrict__ raw,
    const float *__restrict__ att_u, const float *__restrict__ att_i, const int64_t *__restrict__ users,
    const int64_t *__restrict__ items, const float *__restrict__ labels, int B, float grad_scale, float *loss_sum,
    float *__restrict__ lo_batch, float *__restrict__ grad_slots, float *__restrict__ loss_rows, const float *__restrict__ acc2,
    const float *__restrict__ acc3, const spex::EdgeDrop drop)
{
    __shared__ float s_part[2][kWgWaves][kWave];
    __shared__ float s_mixed[2][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x;
    const int64_t u64 = users[b], i64 = items[b];
    const float y_lab = labels[b];
    if (u64 < 0 || u64 >= n_user_rows || i64 < 0 || i64 + n_user_rows >= n_rows) {   // workgroup-uniform: never gather out of range
        if (wave < 2) grad_slots[(size_t)(wave * B + b) * kWave + lane] = 0.0f;       // gated rows of 0: x = 0, no gradient
        if (threadIdx.x == 0) {                                                        // (what the three launches did: BCE(0, y))
            if (loss_rows) loss_rows[b] = 0.693147180559945309f;
            else atomicAdd(loss_sum, 0.693147180559945309f);
        }
        return;
    }
    const int row[2] = {(int)u64, (int)i64 + n_user_rows};
    const int beg[2] = {rowptr[row[0]], rowptr[row[1]]};
    const int deg[2] = {rowptr[row[0] + 1] - beg[0], rowptr[row[1] + 1] - beg[1]};
    float run = 0.0f, a_raw = 0.0f, w00 = 0.0f, w01 = 0.0f, w10 = 0.0f, w11 = 0.0f;
    if (wave < 2) {                       // the gate's operands, requested with everything else
        const float *att = wave ? att_i : att_u;
        run = acc_in[(size_t)row[wave] * kWave + lane];
        float r2 = 0.0f, r3 = 0.0f;                          // (plain-form forward layers: see lightgcn_batch_kernel)
        if (acc2) r2 = acc2[(size_t)row[wave] * kWave + lane];
        if (acc3) r3 = acc3[(size_t)row[wave] * kWave + lane];
        if (acc2) run = run + r2;
        if (acc3) run = run + r3;
        a_raw = raw[(size_t)row[wave] * kWave + lane];
        w00 = att[2 * lane]; w01 = att[2 * lane + 1];
        w10 = att[2 * (kWave + lane)]; w11 = att[2 * (kWave + lane) + 1];
    }
    const int nseg[2] = {(deg[0] + kTaskEntries - 1) / kTaskEntries, (deg[1] + kTaskEntries - 1) / kTaskEntries};
    const int nv0 = nseg[0] < kWgWaves ? nseg[0] : kWgWaves, nv = nv0 + (nseg[1] < kWgWaves ? nseg[1] : kWgWaves);
    const float *__restrict__ Xl = X + lane;
    for (int j = wave; j < nv; j += kWgWaves) {
        const int side = j >= nv0, v = side ? j - nv0 : j;
        float acc = 0.0f;
        for (int sgi = v; sgi < nseg[side]; sgi += kWgWaves) {
            const int left = deg[side] - sgi * kTaskEntries;
            if (drop.mode != 0)             // edge dropout: the handle's keep rule, as in lightgcn_batch_kernel
                acc = segment_sum_masked(col, val, Xl, beg[side] + sgi * kTaskEntries, left < kTaskEntries ? left : kTaskEntries, lane, acc, drop);
            else
                acc = segment_sum(col, val, Xl, beg[side] + sgi * kTaskEntries, left < kTaskEntries ? left : kTaskEntries, lane, acc);
        }
        s_part[side][v][lane] = acc;
    }
    __syncthreads();
    if (wave < 2) {
        const int lim = nseg[wave] < kWgWaves ? nseg[wave] : kWgWaves;
        float y = lim > 0 ? s_part[wave][0][lane] : 0.0f;
        for (int w = 1; w < lim; ++w) y = y + s_part[wave][w][lane];          // segment order
        float s = run + y;
        if (acc_div != 1.0f) s = s / acc_div;
        lo_batch[(size_t)row[wave] * kWave + lane] = s;                        // (a row named twice is written twice with the same value)
        // the gate, as expert_gate_rows_kernel computes it
        float z0 = fmaf(s, w10, fmaf(a_raw, w00, 0.0f)), z1 = fmaf(s, w11, fmaf(a_raw, w01, 0.0f));
        z0 = wave_sum_f32(z0);
        z1 = wave_sum_f32(z1);
        const float mx = fmaxf(z0, z1);
        const float e0 = expf(z0 - mx), e1 = expf(z1 - mx);
        const float a0 = e0 / (e0 + e1), a1 = e1 / (e0 + e1);
        s_mixed[wave][lane] = a_raw * a0 + s * a1;
    }
    __syncthreads();
    if (wave < 2) {
        const float mu = s_mixed[0][lane], mi = s_mixed[1][lane];
        const float x = wave_sum_f32(fmaf(mu, mi, 0.0f));
        const float dg = (sigmoid_f(x) - y_lab) * grad_scale;
        grad_slots[(size_t)(wave * B + b) * kWave + lane] = dg * (wave ? mu : mi);
        if (wave == 0 && lane == 0) {
            const float bce = fmaxf(x, 0.0f) - x * y_lab + log1pf(expf(-fabsf(x)));
            if (loss_rows) loss_rows[b] = bce;            // deterministic step: summed in sample order afterwards
            else atomicAdd(loss_sum, bce);
        }
    }
}


// The WHOLE batch-sized middle of the dual-task rec branch as one launch (spex_gated_batch_f32; the fast, atomic path of
// spex_dual_task_step_f32): gated_batch_fwd_kernel's forward, then — the gate's Jacobian is linear in the incoming gradient and
// local to the sample's two rows — the gate's backward in the same two waves (expert_gate_rows_bwd_kernel's arithmetic, operand
// for operand: d raw, d light, the two gate matrices' gradients), and the first backward product in push form over the rows of A
// exactly as lightgcn_batch_kernel<true> runs it (runs of 16 entries dealt over (part, wave), loaded ahead of the forward).
// Three launches (10.1 + 5.0 + 10.2 us on Epinion2, B = 256) and two launch boundaries of a dependent chain become one.
//   g_prop[r]  += d light            (dense d loss / d light: the epilogue operand of the later backward launches, Adam's share)
//   G[r]       += push_scale * d light,   G[col[e]] += val[e] * push_scale * d light over the stored entries of row r of A
//   g_raw[r]   += d raw              (the gate's direct path into E^0)
//   g_att[b mod n_att_copies] += the gate matrices' gradients ([att_u | att_i], [128, 2] each; 4 values per lane, one atomic each)
__global__ __launch_bounds__(kWave *kWgWaves) void gated_batch_push_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val, int n_rows, int n_user_rows,
    const float *__restrict__ X, const float *__restrict__ acc_in, float acc_div, const float *__restrict__ raw,
    const float *__restrict__ att_u, const float *__restrict__ att_i, const int64_t *__restrict__ users,
    const int64_t *__restrict__ items, const float *__restrict__ labels, int parts, int runs_per_part, float grad_scale,
    float push_scale, float *loss_sum, float *g_prop, float *G, float *g_raw, float *g_att, int n_att_copies,
    const float *__restrict__ acc2, const float *__restrict__ acc3, const spex::EdgeDrop drop)
{
    __shared__ float s_part[2][kWgWaves][kWave];
    __shared__ float s_mixed[2][kWave];
    __shared__ float s_dprop[2][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x / parts, part = blockIdx.x % parts;
    const int64_t u64 = users[b], i64 = items[b];
    const float y_lab = labels[b];
    if (u64 < 0 || u64 >= n_user_rows || i64 < 0 || i64 + n_user_rows >= n_rows) {   // workgroup-uniform: never gather out of range
        if (part == 0 && threadIdx.x == 0) atomicAdd(loss_sum, 0.693147180559945309f);   // gated rows of 0: BCE(0, y), no gradient
        return;
    }
    const int row[2] = {(int)u64, (int)i64 + n_user_rows};
    const int beg[2] = {rowptr[row[0]], rowptr[row[1]]};
    const int deg[2] = {rowptr[row[0] + 1] - beg[0], rowptr[row[1] + 1] - beg[1]};
    float run = 0.0f, a_raw = 0.0f, w00 = 0.0f, w01 = 0.0f, w10 = 0.0f, w11 = 0.0f;
    if (wave < 2) {                       // the gate's operands, requested with everything else
        const float *att = wave ? att_i : att_u;
        const size_t o = (size_t)row[wave] * kWave + lane;
        run = acc_in[o];
        float r2 = 0.0f, r3 = 0.0f;                          // (plain-form forward layers: see lightgcn_batch_kernel)
        if (acc2) r2 = acc2[o];
        if (acc3) r3 = acc3[o];
        if (acc2) run = run + r2;
        if (acc3) run = run + r3;
        a_raw = raw[o];
        w00 = att[2 * lane]; w01 = att[2 * lane + 1];
        w10 = att[2 * (kWave + lane)]; w11 = att[2 * (kWave + lane) + 1];
    }
    // the push's runs (the forward's rows again: A^T g in push form walks the rows of A), shared by up to `parts` workgroups
    const int n_run0 = (deg[0] + 15) >> 4, n_runs = n_run0 + ((deg[1] + 15) >> 4);
    const int want = (n_runs + runs_per_part - 1) / runs_per_part;
    const int act = want < parts ? (want < 1 ? 1 : want) : parts;
    if (part >= act) return;
    const int q_step = act * kWgWaves;
    int q = part * kWgWaves + wave;
    int p_col[kPre], p_cnt[kPre], p_side[kPre];
    float p_val[kPre];
    auto load_runs = [&](int q0) {
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            const int qq = q0 + p * q_step;
            p_col[p] = 0;
            p_val[p] = 0.0f;
            p_cnt[p] = 0;
            p_side[p] = 0;
            if (qq < n_runs) {
                const int side = qq >= n_run0, rr = side ? qq - n_run0 : qq;
                const int base = beg[side] + rr * 16, left = deg[side] - rr * 16;
                p_side[p] = side;
                p_cnt[p] = left < 16 ? left : 16;
                if (lane < p_cnt[p]) {
                    p_col[p] = col[base + lane];
                    p_val[p] = val[base + lane];
                    if (drop.mode != 0) p_val[p] = spex::edge_kept(drop, base + lane) ? p_val[p] / drop.keep_prob : 0.0f;   // the forward's mask
                }
            }
        }
    };
    load_runs(q);
    // ---- 1. last layer at both rows (gated_batch_fwd_kernel's / the row-list kernel's order of sums)
    const int nseg[2] = {(deg[0] + kTaskEntries - 1) / kTaskEntries, (deg[1] + kTaskEntries - 1) / kTaskEntries};
    const int nv0 = nseg[0] < kWgWaves ? nseg[0] : kWgWaves, nv = nv0 + (nseg[1] < kWgWaves ? nseg[1] : kWgWaves);
    const float *__restrict__ Xl = X + lane;
    for (int j = wave; j < nv; j += kWgWaves) {
        const int side = j >= nv0, v = side ? j - nv0 : j;
        float acc = 0.0f;
        for (int sgi = v; sgi < nseg[side]; sgi += kWgWaves) {
            const int left = deg[side] - sgi * kTaskEntries;
            if (drop.mode != 0)             // edge dropout: the handle's keep rule, as in lightgcn_batch_kernel
                acc = segment_sum_masked(col, val, Xl, beg[side] + sgi * kTaskEntries, left < kTaskEntries ? left : kTaskEntries, lane, acc, drop);
            else
                acc = segment_sum(col, val, Xl, beg[side] + sgi * kTaskEntries, left < kTaskEntries ? left : kTaskEntries, lane, acc);
        }
        s_part[side][v][lane] = acc;
    }
    __syncthreads();
    // ---- 2. layer mean + gate (waves 0 / 1: the user / the item row)
    float s = 0.0f, a0 = 0.0f, a1 = 0.0f;
    if (wave < 2) {
        const int lim = nseg[wave] < kWgWaves ? nseg[wave] : kWgWaves;
        float y = lim > 0 ? s_part[wave][0][lane] : 0.0f;
        for (int w = 1; w < lim; ++w) y = y + s_part[wave][w][lane];          // segment order
        s = run + y;
        if (acc_div != 1.0f) s = s / acc_div;
        float z0 = fmaf(s, w10, fmaf(a_raw, w00, 0.0f)), z1 = fmaf(s, w11, fmaf(a_raw, w01, 0.0f));
        z0 = wave_sum_f32(z0);
        z1 = wave_sum_f32(z1);
        const float mx = fmaxf(z0, z1);
        const float e0 = expf(z0 - mx), e1 = expf(z1 - mx);
        a0 = e0 / (e0 + e1); a1 = e1 / (e0 + e1);
        s_mixed[wave][lane] = a_raw * a0 + s * a1;
    }
    __syncthreads();
    // ---- 3. score, loss, d loss / d gated rows, and the gate's backward for this sample's two rows
    if (wave < 2) {
        const float mu = s_mixed[0][lane], mi = s_mixed[1][lane];
        const float x = wave_sum_f32(fmaf(mu, mi, 0.0f));
        const float dg = (sigmoid_f(x) - y_lab) * grad_scale;
        const float gg = dg * (wave ? mu : mi);
        const float da0 = wave_sum_f32(gg * a_raw), da1 = wave_sum_f32(gg * s);
        const float dot = a0 * da0 + a1 * da1;
        const float dz0 = a0 * (da0 - dot), dz1 = a1 * (da1 - dot);
        const float d_raw = a0 * gg + dz0 * w00 + dz1 * w01, d_prop = a1 * gg + dz0 * w10 + dz1 * w11;
        s_dprop[wave][lane] = d_prop;
        if (part == 0) {
            const size_t o = (size_t)row[wave] * kWave + lane;
            if (g_prop) atomicAdd(g_prop + o, d_prop);                    // (NULL: nobody reads it — the L == 3 step)
            atomicAdd(G + o, push_scale * d_prop);
            atomicAdd(g_raw + o, d_raw);
            // (every sample adds to the same 512 words: 256 x 64 lane-atomics per cache line serialise in L2 — 13 us of a 25 us
            //  launch when all went to ONE copy — so sample b adds into copy b mod n_att_copies and the caller sums the copies)
            float *ga = g_att + (size_t)(b % n_att_copies) * 512 + wave * 256;
            atomicAdd(ga + 2 * lane, fmaf(a_raw, dz0, 0.0f));
            atomicAdd(ga + 2 * lane + 1, fmaf(a_raw, dz1, 0.0f));
            atomicAdd(ga + 2 * (kWave + lane), fmaf(s, dz0, 0.0f));
            atomicAdd(ga + 2 * (kWave + lane) + 1, fmaf(s, dz1, 0.0f));
            if (wave == 0 && lane == 0) atomicAdd(loss_sum, fmaxf(x, 0.0f) - x * y_lab + log1pf(expf(-fabsf(x))));
        }
    }
    __syncthreads();
    // ---- 4. push over both rows' entries (lightgcn_batch_kernel's loop)
    const float g2[2] = {push_scale * s_dprop[0][lane], push_scale * s_dprop[1][lane]};
    float *out_l = G + lane;
    for (;;) {
        int c0[kPre];
        float v0[kPre];
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            c0[p] = __builtin_amdgcn_readlane(p_col[p], 0);
            v0[p] = lane_bcast(p_val[p], 0);
        }
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            const float gs = p_side[p] ? g2[1] : g2[0];
            if (p_cnt[p] > 0) atomicAdd(out_l + (size_t)c0[p] * kWave, v0[p] * gs);
#pragma unroll 1
            for (int j = 1; j < p_cnt[p]; ++j) {
                const int c = __builtin_amdgcn_readlane(p_col[p], j);
                const float v = lane_bcast(p_val[p], j);
                atomicAdd(out_l + (size_t)c * kWave, v * gs);
            }
        }
        q += kPre * q_step;
        if (q >= n_runs) break;
        load_runs(q);
    }
}



// The last forward layer of a ROW-PARTITIONED step at the batch's rows (d == 64; comm.hip).  Slot k names position pos[k] of the
// padded global layout; on the rank that owns it (lo <= pos[k] < lo + n_rows) the row-list kernel's sum for the rank's local row
// r = pos[k] - lo — wave w takes segments w, w + 16, ... as one chain, the waves' sums are added in wave order: the same bits for
// every row of up to 1 024 entries — gives the layer mean (acc_in [+ acc2 + acc3] + A X)[r] / acc_div, stored COMPACT at out_prop[k]
// beside the raw row out_raw[k] = raw[r]; every other slot is ZERO-filled on this rank: the two buffers are the operands of the
// owner-computes all-reduce that hands every rank the batch's rows.  Replaces a whole-block launch + two gather launches.  Under
// edge dropout (a mask on the handle) the entries are kept / dropped by the handle's rule, exactly as spmm_chunk_kernel<.., MASKED>.
__global__ __launch_bounds__(kWave *kWgWaves) void owned_rows_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ X, const int64_t *__restrict__ pos, int64_t lo, int32_t n_rows, const float *__restrict__ acc_in,
    const float *__restrict__ acc2, const float *__restrict__ acc3, float acc_div, const float *__restrict__ raw,
    float *__restrict__ out_prop, float *__restrict__ out_raw, const spex::EdgeDrop drop)
{
    __shared__ float s_part[kWgWaves][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t ko = (size_t)blockIdx.x * kWave + lane;
    const int64_t r64 = pos[blockIdx.x] - lo;
    if (r64 < 0 || r64 >= n_rows) {                               // workgroup-uniform: another rank's row (or no row at all)
        if (wave == 0) out_prop[ko] = 0.0f;
        if (wave == 1 && out_raw) out_raw[ko] = 0.0f;
        return;
    }
    const int r = (int)r64;
    const size_t o = (size_t)r * kWave + lane;
    float run = 0.0f, r2 = 0.0f, r3 = 0.0f;
    if (wave == 0) {                                               // the epilogue's operands, requested with the row's entries
        run = acc_in[o];
        if (acc2) r2 = acc2[o];
        if (acc3) r3 = acc3[o];
    }
    if (wave == 1 && out_raw) out_raw[ko] = raw[o];
    const int beg = rowptr[r], deg = rowptr[r + 1] - beg;
    const int nseg = (deg + kTaskEntries - 1) / kTaskEntries;
    const float *__restrict__ Xl = X + lane;
    float acc = 0.0f;
    for (int sgi = wave; sgi < nseg; sgi += kWgWaves) {
        const int left = deg - sgi * kTaskEntries;
        if (drop.mode != 0)
            acc = segment_sum_masked(col, val, Xl, beg + sgi * kTaskEntries, left < kTaskEntries ? left : kTaskEntries, lane, acc, drop);
        else
            acc = segment_sum(col, val, Xl, beg + sgi * kTaskEntries, left < kTaskEntries ? left : kTaskEntries, lane, acc);
    }
    if (nseg > 1) {
        if (wave != 0 && wave < nseg) s_part[wave][lane] = acc;
        __syncthreads();
    }
    if (wave == 0) {
        float y = acc;
        const int lim = nseg < kWgWaves ? nseg : kWgWaves;
        for (int w = 1; w < lim; ++w) y = y + s_part[w][lane];    // wave order
        if (acc2) run = run + r2;                                  // (the layer tables in layer order, as the one-GPU batch kernels add them)
        if (acc3) run = run + r3;
        float s = run + y;
        if (acc_div != 1.0f) s = s / acc_div;
        out_prop[ko] = s;
    }
}

// The batch-sized middle of the one-call steps on a ROW PARTITION (comm.hip: the fast paths of spex_partitioned_step_bce_f32 and
// spex_partitioned_dual_task_step_f32).  The batch's rows arrive COMPACT and complete on every rank (rows_prop / rows_raw [2B, 64]:
// slot b = sample b's user row, slot B + b its item row — the owner-computes all-reduce behind spex_spmm_owned_rows_f32), so
// everything between the forward and the first backward product is one launch, computed REDUNDANTLY by every rank (which is why
// the loss and the two gate matrices' gradients are complete everywhere without a collective):
//   GATED  gate, score, loss and the gate's backward — gated_batch_push_kernel's steps 2-3, operand for operand (waves 0 / 1)
//   plain  score, loss, the two gradient rows — lightgcn_batch_kernel's step 2
//   part 0 of a sample, for a slot whose row this rank owns (lo <= pos[slot] < lo + n_local; r = pos[slot] - lo):
//          P[r] += push_scale * d_prop  (the g term of (g + A^T g) / (L+1)),  g_prop[r] += d_prop (if g_prop),  g_raw[r] += d_raw
//   then the first backward product in push form WITHOUT an exchange: every rank pushes BOTH gradient rows of EVERY sample through
//   its own columns of A — the matrix (rowptr, col, val) is the (world * max_rows) x n_local transpose of the rank's block of A^T,
//   row p = the entries A[p, c] for the rank's columns c: P[col[e]] += push_scale * val[e] * d_prop over the entries of rows
//   pos[b], pos[B + b] (runs of 16 entries dealt over (part, wave) and loaded ahead, as in lightgcn_batch_kernel<true>; under edge
//   dropout the structure carries the entries' global edge ids and the forward's mask: kept values / keep_prob, dropped ones nothing).
// rowptr == NULL: a rank without rows (nothing to push, nothing owned) — it still forms the loss and the gate gradients.
template <bool GATED>
__global__ __launch_bounds__(kWave *kWgWaves) void rows_train_push_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val, int n_push_rows,
    const float *__restrict__ rows_raw, const float *__restrict__ rows_prop, const float *__restrict__ att_u,
    const float *__restrict__ att_i, const int64_t *__restrict__ pos, int64_t lo, int n_local, const float *__restrict__ labels, int B,
    int parts, int runs_per_part, float grad_scale, float push_scale, float *loss_sum, float *g_prop, float *P, float *g_raw,
    float *g_att, int n_att_copies, const spex::EdgeDrop drop)
{
    __shared__ float s_mixed[2][kWave];
    __shared__ float s_dprop[2][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x / parts, part = blockIdx.x % parts;
    const int64_t p64[2] = {pos[b], pos[(size_t)B + b]};
    const float y_lab = labels[b];
    int beg[2] = {0, 0}, deg[2] = {0, 0};
#pragma unroll
    for (int w = 0; w < 2; ++w)
        if (rowptr && p64[w] >= 0 && p64[w] < n_push_rows) {
            beg[w] = rowptr[p64[w]];
            deg[w] = rowptr[p64[w] + 1] - beg[w];
        }
    float a_raw = 0.0f, s = 0.0f, w00 = 0.0f, w01 = 0.0f, w10 = 0.0f, w11 = 0.0f;
    if (wave < 2) {                       // the operands, requested with everything else
        const size_t o = ((size_t)wave * B + b) * kWave + lane;
        s = rows_prop[o];
        if (GATED) {
            const float *att = wave ? att_i : att_u;
            a_raw = rows_raw[o];
            w00 = att[2 * lane]; w01 = att[2 * lane + 1];
            w10 = att[2 * (kWave + lane)]; w11 = att[2 * (kWave + lane) + 1];
        }
    }
    const int n_run0 = (deg[0] + 15) >> 4, n_runs = n_run0 + ((deg[1] + 15) >> 4);
    const int want = (n_runs + runs_per_part - 1) / runs_per_part;
    const int act = want < parts ? (want < 1 ? 1 : want) : parts;
    if (part >= act) return;
    const int q_step = act * kWgWaves;
    int q = part * kWgWaves + wave;
    int p_col[kPre], p_cnt[kPre], p_side[kPre];
    float p_val[kPre];
    auto load_runs = [&](int q0) {
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            const int qq = q0 + p * q_step;
            p_col[p] = 0;
            p_val[p] = 0.0f;
            p_cnt[p] = 0;
            p_side[p] = 0;
            if (qq < n_runs) {
                const int side = qq >= n_run0, rr = side ? qq - n_run0 : qq;
                const int base = beg[side] + rr * 16, left = deg[side] - rr * 16;
                p_side[p] = side;
                p_cnt[p] = left < 16 ? left : 16;
                if (lane < p_cnt[p]) {
                    p_col[p] = col[base + lane];
                    p_val[p] = val[base + lane];
                    if (drop.mode != 0) p_val[p] = spex::edge_kept(drop, base + lane) ? p_val[p] / drop.keep_prob : 0.0f;   // the forward's mask
                }
            }
        }
    };
    load_runs(q);
    // ---- gate (GATED) -> the rows that are scored
    float a0 = 0.0f, a1 = 0.0f;
    if (wave < 2) {
        float mixed = s;
        if (GATED) {
            float z0 = fmaf(s, w10, fmaf(a_raw, w00, 0.0f)), z1 = fmaf(s, w11, fmaf(a_raw, w01, 0.0f));
            z0 = wave_sum_f32(z0);
            z1 = wave_sum_f32(z1);
            const float mx = fmaxf(z0, z1);
            const float e0 = expf(z0 - mx), e1 = expf(z1 - mx);
            a0 = e0 / (e0 + e1); a1 = e1 / (e0 + e1);
            mixed = a_raw * a0 + s * a1;
        }
        s_mixed[wave][lane] = mixed;
    }
    __syncthreads();
    // ---- score, loss, d loss / d scored rows (and through the gate), the owned rows' shares
    if (wave < 2) {
        const float mu = s_mixed[0][lane], mi = s_mixed[1][lane];
        const float x = wave_sum_f32(fmaf(mu, mi, 0.0f));
        const float dg = (sigmoid_f(x) - y_lab) * grad_scale;
        const float gg = dg * (wave ? mu : mi);
        float d_raw = 0.0f, d_prop = gg, dz0 = 0.0f, dz1 = 0.0f;
        if (GATED) {
            const float da0 = wave_sum_f32(gg * a_raw), da1 = wave_sum_f32(gg * s);
            const float dot = a0 * da0 + a1 * da1;
            dz0 = a0 * (da0 - dot); dz1 = a1 * (da1 - dot);
            d_raw = a0 * gg + dz0 * w00 + dz1 * w01;
            d_prop = a1 * gg + dz0 * w10 + dz1 * w11;
        }
        s_dprop[wave][lane] = d_prop;
        if (part == 0) {
            const int64_t r64 = p64[wave] - lo;
            if (r64 >= 0 && r64 < n_local) {                              // wave-uniform: the rows this rank owns
                const size_t o = (size_t)r64 * kWave + lane;
                if (g_prop) atomicAdd(g_prop + o, d_prop);
                atomicAdd(P + o, push_scale * d_prop);
                if (GATED) atomicAdd(g_raw + o, d_raw);
            }
            if (GATED) {
                float *ga = g_att + (size_t)(b % n_att_copies) * 512 + wave * 256;   // (copies: see gated_batch_push_kernel)
                atomicAdd(ga + 2 * lane, fmaf(a_raw, dz0, 0.0f));
                atomicAdd(ga + 2 * lane + 1, fmaf(a_raw, dz1, 0.0f));
                atomicAdd(ga + 2 * (kWave + lane), fmaf(s, dz0, 0.0f));
                atomicAdd(ga + 2 * (kWave + lane) + 1, fmaf(s, dz1, 0.0f));
            }
            if (wave == 0 && lane == 0) atomicAdd(loss_sum, fmaxf(x, 0.0f) - x * y_lab + log1pf(expf(-fabsf(x))));
        }
    }
    if (n_runs == 0) return;                                               // (workgroup-uniform)
    __syncthreads();
    // ---- push over both rows' entries in the rank's columns (lightgcn_batch_kernel's loop)
    const float g2[2] = {push_scale * s_dprop[0][lane], push_scale * s_dprop[1][lane]};
    float *out_l = P + lane;
    for (;;) {
        int c0[kPre];
        float v0[kPre];
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            c0[p] = __builtin_amdgcn_readlane(p_col[p], 0);
            v0[p] = lane_bcast(p_val[p], 0);
        }
#pragma unroll
        for (int p = 0; p < kPre; ++p) {
            const float gs = p_side[p] ? g2[1] : g2[0];
            if (p_cnt[p] > 0) atomicAdd(out_l + (size_t)c0[p] * kWave, v0[p] * gs);
#pragma unroll 1
            for (int j = 1; j < p_cnt[p]; ++j) {
                const int c = __builtin_amdgcn_readlane(p_col[p], j);
                const float v = lane_bcast(p_val[p], j);
                atomicAdd(out_l + (size_t)c * kWave, v * gs);
            }
        }
        q += kPre * q_step;
        if (q >= n_runs) break;
        load_runs(q);
    }
}

}  // namespace

static spex::EdgeDrop edge_drop_of(const spex_graph_t *g)
{
    return spex::EdgeDrop{g->keep, g->edge_id, g->mask_mode, g->keep_prob, (uint32_t)g->seed, (uint32_t)(g->seed >> 32)};
}

extern "C" int spex_gated_batch_fwd_f32(const spex_graph_t *g, const float *X, const float *acc_in, float acc_div, const float *raw,
                                       const float *att_u, const float *att_i, const int64_t *users, const int64_t *items,
                                       const float *labels, int32_t B, int32_t n_user_rows, float grad_scale, float *loss_sum,
                                       float *loss_per_sample, float *lo_batch, float *grad_slots, int32_t d, void *stream)
{
    return spex::gated_batch_fwd_layers(g, X, acc_in, nullptr, nullptr, acc_div, raw, att_u, att_i, users, items, labels, B, n_user_rows,
                                        grad_scale, loss_sum, loss_per_sample, lo_batch, grad_slots, d, stream);
}

int spex::gated_batch_fwd_layers(const spex_graph_t *g, const float *X, const float *acc_in, const float *acc2, const float *acc3,
                                 float acc_div, const float *raw, const float *att_u, const float *att_i, const int64_t *users,
                                 const int64_t *items, const float *labels, int32_t B, int32_t n_user_rows, float grad_scale,
                                 float *loss_sum, float *loss_per_sample, float *lo_batch, float *grad_slots, int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g && X && acc_in && raw && att_u && att_i && users && items && labels && (loss_sum || loss_per_sample) && lo_batch
                       && grad_slots,
                   "spex_gated_batch_fwd_f32: NULL argument");
    SPEX_CHECK_ARG(B >= 0 && n_user_rows >= 0 && n_user_rows <= g->n_rows && g->n_rows == g->n_cols,
                   "spex_gated_batch_fwd_f32: B=%d n_user_rows=%d on a %d x %d graph", B, n_user_rows, g->n_rows, g->n_cols);
    if (d != kWave) {
        spex::set_error("spex_gated_batch_fwd_f32: d == 64 only (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    if (B == 0 || g->n_rows == 0) return SPEX_OK;
    hipLaunchKernelGGL(gated_batch_fwd_kernel, dim3((unsigned)B), dim3(kWave * kWgWaves), 0, (hipStream_t)stream, g->rowptr, g->col, g->val,
                       g->n_rows, n_user_rows, X, acc_in, acc_div, raw, att_u, att_i, users, items, labels, B, grad_scale, loss_sum, lo_batch,
                       grad_slots, loss_per_sample, acc2, acc3, edge_drop_of(g));
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_gated_batch_f32(const spex_graph_t *g, const float *X, const float *acc_in, float acc_div, const float *raw,
                                   const float *att_u, const float *att_i, const int64_t *users, const int64_t *items,
                                   const float *labels, int32_t B, int32_t n_user_rows, float grad_scale, float push_scale,
                                   float *loss_sum, float *g_prop, float *G, float *g_raw, float *g_att, int32_t n_att_copies,
                                   int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g_prop, "spex_gated_batch_f32: NULL g_prop");
    return spex::gated_batch_push_layers(g, X, acc_in, nullptr, nullptr, acc_div, raw, att_u, att_i, users, items, labels, B, n_user_rows,
                                         grad_scale, push_scale, loss_sum, g_prop, G, g_raw, g_att, n_att_copies, d, stream);
}

int spex::gated_batch_push_layers(const spex_graph_t *g, const float *X, const float *acc_in, const float *acc2, const float *acc3,
                                  float acc_div, const float *raw, const float *att_u, const float *att_i, const int64_t *users,
                                  const int64_t *items, const float *labels, int32_t B, int32_t n_user_rows, float grad_scale,
                                  float push_scale, float *loss_sum, float *g_prop, float *G, float *g_raw, float *g_att,
                                  int32_t n_att_copies, int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g && X && acc_in && raw && att_u && att_i && users && items && labels && loss_sum && G && g_raw && g_att,
                   "spex_gated_batch_f32: NULL argument");
    SPEX_CHECK_ARG(n_att_copies >= 1, "spex_gated_batch_f32: n_att_copies=%d", n_att_copies);
    SPEX_CHECK_ARG(B >= 0 && n_user_rows >= 0 && n_user_rows <= g->n_rows && g->n_rows == g->n_cols,
                   "spex_gated_batch_f32: B=%d n_user_rows=%d on a %d x %d graph", B, n_user_rows, g->n_rows, g->n_cols);
    SPEX_CHECK_ARG(G != g_prop && G != g_raw && g_prop != g_raw, "spex_gated_batch_f32: g_prop, G and g_raw are three tables");
    if (d != kWave) {
        spex::set_error("spex_gated_batch_f32: d == 64 only (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    if (B == 0 || g->n_rows == 0) return SPEX_OK;
    constexpr int runs_per_part = kBatchRunsPerPart, parts = kBatchParts;
    hipLaunchKernelGGL(gated_batch_push_kernel, dim3((unsigned)B * parts), dim3(kWave * kWgWaves), 0, (hipStream_t)stream, g->rowptr, g->col,
                       g->val, g->n_rows, n_user_rows, X, acc_in, acc_div, raw, att_u, att_i, users, items, labels, parts, runs_per_part,
                       grad_scale, push_scale, loss_sum, g_prop, G, g_raw, g_att, n_att_copies, acc2, acc3, edge_drop_of(g));
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

#ifdef SPEX_STAMPS
extern "C" int spex_debug_batch_stamps(unsigned long long *out)
{
    SPEX_HIP(hipDeviceSynchronize());
    SPEX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16));
    return SPEX_OK;
}
#endif

extern "C" int spex_lightgcn_batch_slots_f32(const spex_graph_t *g, const float *X, const float *acc_in, float acc_div,
                                             const int64_t *users, const int64_t *items, const float *labels, int32_t B,
                                             int32_t n_user_rows, float grad_scale, float *loss_sum, float *loss_per_sample,
                                             float *grad_slots, int32_t d, void *stream)
{
    return spex::lightgcn_batch_slots_layers(g, X, acc_in, nullptr, nullptr, acc_div, users, items, labels, B, n_user_rows, grad_scale,
                                             loss_sum, loss_per_sample, grad_slots, d, stream);
}

int spex::lightgcn_batch_slots_layers(const spex_graph_t *g, const float *X, const float *acc_in, const float *acc2, const float *acc3,
                                      float acc_div, const int64_t *users, const int64_t *items, const float *labels, int32_t B,
                                      int32_t n_user_rows, float grad_scale, float *loss_sum, float *loss_per_sample, float *grad_slots,
                                      int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g && X && acc_in && users && items && labels && (loss_sum || loss_per_sample) && grad_slots,
                   "spex_lightgcn_batch_slots_f32: NULL argument");
    SPEX_CHECK_ARG(B >= 0 && n_user_rows >= 0 && n_user_rows <= g->n_rows, "spex_lightgcn_batch_slots_f32: B=%d n_user_rows=%d", B, n_user_rows);
    SPEX_CHECK_ARG(g->n_rows == g->n_cols, "spex_lightgcn_batch_slots_f32: square graph");
    if (d != kWave) {
        spex::set_error("spex_lightgcn_batch_slots_f32: d == 64 only (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    if (B == 0 || g->n_rows == 0) return SPEX_OK;
    hipLaunchKernelGGL(lightgcn_batch_kernel<false>, dim3((unsigned)B), dim3(kWave * kWgWaves), 0, (hipStream_t)stream, g->rowptr, g->col,
                       g->val, g->n_rows, n_user_rows, X, acc_in, acc_div, users, items, labels, 1, grad_scale, 0.0f, loss_sum,
                       loss_per_sample, nullptr, nullptr, 1, grad_slots, B, acc2, acc3, edge_drop_of(g));
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_lightgcn_batch_f32(const spex_graph_t *g, const float *X, const float *acc_in, float acc_div,
                                       const int64_t *users, const int64_t *items, const float *labels, int32_t B,
                                       int32_t n_user_rows, float grad_scale, float push_scale, float *loss_sum, float *loss_per_sample,
                                       float *g_out, float *G, int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g_out, "spex_lightgcn_batch_f32: NULL g_out");
    return spex::lightgcn_batch_layers(g, X, acc_in, nullptr, nullptr, acc_div, users, items, labels, B, n_user_rows, grad_scale, push_scale,
                                       loss_sum, loss_per_sample, g_out, G, d, stream);
}

int spex::lightgcn_batch_layers(const spex_graph_t *g, const float *X, const float *acc_in, const float *acc2, const float *acc3,
                                float acc_div, const int64_t *users, const int64_t *items, const float *labels, int32_t B,
                                int32_t n_user_rows, float grad_scale, float push_scale, float *loss_sum, float *loss_per_sample, float *g_out,
                                float *G, int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g && X && acc_in && users && items && labels && (loss_sum || loss_per_sample) && G,
                   "spex_lightgcn_batch_f32: NULL argument");
    SPEX_CHECK_ARG(B >= 0 && n_user_rows >= 0 && n_user_rows <= g->n_rows, "spex_lightgcn_batch_f32: B=%d n_user_rows=%d", B, n_user_rows);
    SPEX_CHECK_ARG(g->n_rows == g->n_cols, "spex_lightgcn_batch_f32: square graph");
    if (d != kWave) {
        spex::set_error("spex_lightgcn_batch_f32: d == 64 only (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    if (B == 0 || g->n_rows == 0) return SPEX_OK;
    constexpr int runs_per_part = kBatchRunsPerPart, parts = kBatchParts;
    hipLaunchKernelGGL(lightgcn_batch_kernel<true>, dim3((unsigned)B * parts), dim3(kWave * kWgWaves), 0, (hipStream_t)stream, g->rowptr,
                       g->col, g->val, g->n_rows, n_user_rows, X, acc_in, acc_div, users, items, labels, parts, grad_scale, push_scale,
                       loss_sum, loss_per_sample, g_out, G, runs_per_part, nullptr, B, acc2, acc3, edge_drop_of(g));
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

int spex::rows_train_push(const spex_graph_t *push, const float *rows_raw, const float *rows_prop, const float *att_u, const float *att_i,
                          const int64_t *pos, int64_t lo, int32_t n_local, const float *labels, int32_t B, float grad_scale, float push_scale,
                          float *loss_sum, float *g_prop, float *P, float *g_raw, float *g_att, int32_t n_att_copies, void *stream)
{
    const bool gated = rows_raw != nullptr;
    SPEX_CHECK_ARG(rows_prop && pos && labels && loss_sum && P, "rows_train_push: NULL argument");
    SPEX_CHECK_ARG(!gated || (att_u && att_i && g_raw && g_att && n_att_copies >= 1), "rows_train_push: the gated form needs att_u, att_i, g_raw, g_att");
    SPEX_CHECK_ARG(B >= 0 && n_local >= 0, "rows_train_push: B=%d n_local=%d", B, n_local);
    SPEX_CHECK_ARG(!push || push->n_cols == n_local, "rows_train_push: the push structure has %d columns for %d local rows",
                   push ? push->n_cols : 0, n_local);
    const spex::EdgeDrop drop = push ? edge_drop_of(push) : spex::EdgeDrop{nullptr, nullptr, 0, 1.0f, 0u, 0u};
    SPEX_CHECK_ARG(P != g_prop && P != g_raw && (!g_prop || g_prop != g_raw), "rows_train_push: g_prop, P and g_raw are three tables");
    if (B == 0) return SPEX_OK;
    constexpr int parts = kBatchParts, runs_per_part = kBatchRunsPerPart;
    const int32_t *rp = push && push->n_rows > 0 ? push->rowptr : nullptr;
    if (gated)
        hipLaunchKernelGGL(rows_train_push_kernel<true>, dim3((unsigned)B * parts), dim3(kWave * kWgWaves), 0, (hipStream_t)stream, rp,
                           push ? push->col : nullptr, push ? push->val : nullptr, push ? push->n_rows : 0, rows_raw, rows_prop, att_u, att_i, pos, lo,
                           n_local, labels, B, parts, runs_per_part, grad_scale, push_scale, loss_sum, g_prop, P, g_raw, g_att, n_att_copies, drop);
    else
        hipLaunchKernelGGL(rows_train_push_kernel<false>, dim3((unsigned)B * parts), dim3(kWave * kWgWaves), 0, (hipStream_t)stream, rp,
                           push ? push->col : nullptr, push ? push->val : nullptr, push ? push->n_rows : 0, nullptr, rows_prop, nullptr, nullptr, pos, lo,
                           n_local, labels, B, parts, runs_per_part, grad_scale, push_scale, loss_sum, g_prop, P, nullptr, nullptr, 1, drop);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_spmm_owned_rows_f32(const spex_graph_t *g, const float *X, const int64_t *pos, int32_t n, int64_t lo,
                                        const float *acc_in, const float *acc2, const float *acc3, float acc_div, const float *raw,
                                        float *out_prop, float *out_raw, int32_t d, void *stream)
{
    SPEX_CHECK_ARG(g && X && out_prop && acc_in && n >= 0 && (n == 0 || pos), "spex_spmm_owned_rows_f32: NULL argument");
    SPEX_CHECK_ARG(acc_div != 0.0f && (!out_raw || raw) && (!acc3 || acc2), "spex_spmm_owned_rows_f32: acc_div == 0, out_raw without raw, or acc3 without acc2");
    if (d != kWave) {
        spex::set_error("spex_spmm_owned_rows_f32: d == 64 only (got %d)", d);
        return SPEX_ERR_UNSUPPORTED;
    }
    if (n == 0) return SPEX_OK;
    hipLaunchKernelGGL(owned_rows_kernel, dim3((unsigned)n), dim3(kWave * kWgWaves), 0, (hipStream_t)stream, g->rowptr, g->col, g->val, X, pos, lo,
                       g->n_rows, acc_in, acc2, acc3, acc_div, raw, out_prop, out_raw, edge_drop_of(g));
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
