// Fused Adam over a dense fp32 table — replaces torch.optim.Adam(...).step(), LightGCN_SPEX/code/main_rec.py:23,37.
//
// torch runs this as several elementwise passes (lerp, mul/addcmul, sqrt/div/add, addcdiv); here it is one pass:
// 16 B read per parameter (p, g, m, v) and 12 B written (p, m, v) = 28 B/param, pure HBM streaming, float4 per lane.
// Optionally the same pass clears one more buffer of the same length (the dense d loss / d light_out table the next
// step's scoring kernel accumulates into): 4 B/param more instead of a separate fill launch.
// Arithmetic order follows torch's single-tensor Adam so that the result matches it to rounding:
//   m += (g - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g * g
//   denom = sqrt(v) / sqrt(1 - beta2^t) + eps;  p -= (lr / (1 - beta1^t)) * (m / denom)
#include <math.h>

#include "spex_common.h"

namespace {

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, float w1, float beta2, float w2,
                                      float bc2_sqrt, float eps, float step_size)
{
    m = m + w1 * (g - m);
    v = v * beta2 + w2 * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

// (g and zero_buf carry no __restrict__: the caller may pass the gradient buffer itself to be cleared — every element is
// read before the same thread clears it)
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *g,
                                                   float *__restrict__ m, float *__restrict__ v, int64_t n4,
                                                   int64_t rem, float w1, float beta2, float w2, float bc2_sqrt,
                                                   float eps, float step_size, float *zero_buf, float *zero_buf2,
                                                   const float *__restrict__ loss_rows, int n_loss, float *loss_sum,
                                                   const spex::SmallAdam small, const int main_blocks, const float *add_g,
                                                   const float add_div)
{
    // A second, small parameter block rides in the same launch (the NGCF step's layer weights: 8 320 parameters whose gradient
    // arrives as partial blocks — a launch of their own cost the step ~5 us of ramp for 33 workgroups): the blocks behind the
    // table's take one parameter per thread, sum its partial gradients in part order (deterministic) and update it.
    if ((int)blockIdx.x >= main_blocks) {
        const int i = ((int)blockIdx.x - main_blocks) * (int)blockDim.x + (int)threadIdx.x;
        if (i < small.n) {
            float gs = 0.0f;
            int k = 0;
            for (; k + 8 <= small.n_parts; k += 8) {             // eight independent loads in flight, added in part order
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = small.parts[(size_t)(k + j) * small.stride + i];
#pragma unroll
                for (int j = 0; j < 8; ++j) gs += x[j];
            }
            for (; k < small.n_parts; ++k) gs += small.parts[(size_t)k * small.stride + i];
            float P = small.p[i], M = small.m[i], V = small.v[i];
            adam1(P, gs, M, V, w1, beta2, w2, bc2_sqrt, eps, step_size);
            small.p[i] = P; small.m[i] = M; small.v[i] = V;
        }
        return;
    }
    // the step's per-sample losses (written by the batch kernel with plain stores) are summed HERE, in a fixed order, and added
    // to the epoch's accumulator as ONE addend per step: one atomic per sample onto the accumulator (1.25 M adds of ~0.7 onto a
    // sum that reaches 8.5e5 in an Epinion2 epoch) lost 6e-5 of the epoch's loss to fp32 rounding
    if (loss_rows && (int)blockIdx.x == main_blocks - 1 && threadIdx.x < spex::kWave) {
        float t = 0.0f;
        for (int i = threadIdx.x; i < n_loss; i += spex::kWave) t += loss_rows[i];
        t = spex::wave_sum_f32(t);
        if (threadIdx.x == 0) *loss_sum += t;
    }
    const int64_t stride = (int64_t)main_blocks * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 P = reinterpret_cast<float4 *>(p)[i];
        float4 G = reinterpret_cast<const float4 *>(g)[i];
        if (add_g) {      // gradient = g + add_g / add_div: the mean's own share g/(L+1) of the LAST backward product, added here (this
                          // pass reads add_g anyway, to clear it) so that that product runs in the plain form
            const float4 A = reinterpret_cast<const float4 *>(add_g)[i];
            G.x = G.x + A.x / add_div; G.y = G.y + A.y / add_div; G.z = G.z + A.z / add_div; G.w = G.w + A.w / add_div;
        }
        float4 M = reinterpret_cast<float4 *>(m)[i];
        float4 V = reinterpret_cast<float4 *>(v)[i];
        adam1(P.x, G.x, M.x, V.x, w1, beta2, w2, bc2_sqrt, eps, step_size);
        adam1(P.y, G.y, M.y, V.y, w1, beta2, w2, bc2_sqrt, eps, step_size);
        adam1(P.z, G.z, M.z, V.z, w1, beta2, w2, bc2_sqrt, eps, step_size);
        adam1(P.w, G.w, M.w, V.w, w1, beta2, w2, bc2_sqrt, eps, step_size);
        reinterpret_cast<float4 *>(p)[i] = P;
        reinterpret_cast<float4 *>(m)[i] = M;
        reinterpret_cast<float4 *>(v)[i] = V;
        if (zero_buf) reinterpret_cast<float4 *>(zero_buf)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (zero_buf2) reinterpret_cast<float4 *>(zero_buf2)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < rem) {
        const int64_t i = n4 * 4 + threadIdx.x;
        float P = p[i], M = m[i], V = v[i];
        adam1(P, add_g ? g[i] + add_g[i] / add_div : g[i], M, V, w1, beta2, w2, bc2_sqrt, eps, step_size);
        p[i] = P; m[i] = M; v[i] = V;
        if (zero_buf) zero_buf[i] = 0.0f;
        if (zero_buf2) zero_buf2[i] = 0.0f;
    }
}

}  // namespace

// Internal form with a second buffer to clear (the one-call training step keeps its push target all-zero this way).
int spex::adam_step_z2(float *p, const float *g, float *m, float *v, int64_t n, int32_t t, float lr, float beta1, float beta2,
                       float eps, float *zero_buf, float *zero_buf2, void *stream, const float *loss_rows, int32_t n_loss, float *loss_sum,
                       const spex::SmallAdam *small_in, const float *add_g, float add_div)
{
    SPEX_CHECK_ARG(!add_g || ((((uintptr_t)add_g) & 15) == 0 && add_div != 0.0f), "spex_adam_step_f32: add_g unaligned or add_div == 0");
    spex::SmallAdam small{};
    if (small_in) small = *small_in;
    SPEX_CHECK_ARG(small.n == 0 || (small.p && small.m && small.v && small.parts && small.n_parts >= 1 && small.stride >= small.n),
                   "spex_adam_step_f32: bad second parameter block");
    SPEX_CHECK_ARG(!loss_rows || (loss_sum && n_loss >= 0), "spex_adam_step_f32: loss rows without an accumulator");
    SPEX_CHECK_ARG(p && g && m && v, "spex_adam_step_f32: NULL pointer");
    SPEX_CHECK_ARG(n >= 0 && t >= 1, "spex_adam_step_f32: n=%lld t=%d (t counts from 1)", (long long)n, t);
    SPEX_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)zero_buf | (uintptr_t)zero_buf2) & 15) == 0,
                   "spex_adam_step_f32: pointers must be 16-byte aligned");
    SPEX_CHECK_ARG(zero_buf != p && zero_buf != m && zero_buf != v, "spex_adam_step_f32: zero_buf aliases p, m or v");
    SPEX_CHECK_ARG(!zero_buf2 || (zero_buf2 != p && zero_buf2 != m && zero_buf2 != v), "spex_adam_step_f32: zero_buf aliases p, m or v");
    if (n == 0) return SPEX_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)t);
    const double bc2 = 1.0 - pow((double)beta2, (double)t);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const int64_t n4 = n / 4, rem = n % 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    const int64_t small_blocks = (small.n + 255) / 256;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks + small_blocks)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, rem,
                       1.0f - beta1, beta2, 1.0f - beta2, bc2_sqrt, eps, step_size, zero_buf, zero_buf2, loss_rows, (int)n_loss, loss_sum,
                       small, (int)blocks, add_g, add_div);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

extern "C" int spex_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n, int32_t t, float lr,
                                  float beta1, float beta2, float eps, float *zero_buf, void *stream)
{
    return spex::adam_step_z2(p, g, m, v, n, t, lr, beta1, beta2, eps, zero_buf, nullptr, stream);
}

// Adam over a small parameter block whose gradient arrives as n_parts partial blocks (one per workgroup of the kernel that
// produced it): g = sum of the parts, taken in part order (deterministic), then the usual update.  One thread per parameter.
namespace {
__global__ __launch_bounds__(256) void adam_sum_kernel(float *__restrict__ p, const float *__restrict__ parts, int n_parts,
                                                       int64_t part_stride, float *__restrict__ m, float *__restrict__ v, int64_t n,
                                                       float w1, float beta2, float w2, float bc2_sqrt, float eps, float step_size)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g = 0.0f;
    int k = 0;
    for (; k + 8 <= n_parts; k += 8) {                   // eight independent loads in flight, added in part order
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = parts[(size_t)(k + j) * part_stride + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) g += x[j];
    }
    for (; k < n_parts; ++k) g += parts[(size_t)k * part_stride + i];
    float P = p[i], M = m[i], V = v[i];
    adam1(P, g, M, V, w1, beta2, w2, bc2_sqrt, eps, step_size);
    p[i] = P; m[i] = M; v[i] = V;
}
}  // namespace

extern "C" int spex_adam_step_sum_f32(float *p, const float *g_parts, int32_t n_parts, int64_t part_stride, float *m, float *v,
                                      int64_t n, int32_t t, float lr, float beta1, float beta2, float eps, void *stream)
{
    SPEX_CHECK_ARG(p && g_parts && m && v && n >= 0 && n_parts >= 1 && part_stride >= n && t >= 1,
                   "spex_adam_step_sum_f32: bad argument (n=%lld n_parts=%d stride=%lld t=%d)", (long long)n, n_parts, (long long)part_stride, t);
    if (n == 0) return SPEX_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    hipLaunchKernelGGL(adam_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g_parts, n_parts,
                       part_stride, m, v, n, 1.0f - beta1, beta2, 1.0f - beta2, (float)sqrt(bc2), eps, (float)(lr / bc1));
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Adam over the dual-task model's parameter arena (steps.hip, spex_dual_task_step_f32).  The step's loss is
//   exp(-2 s0) loss1 + exp(-2 s1) loss2 + 2 (n_rec + 1) B s0 + T s1        (main_auto_expert_s.py:78-82, s = task_weights)
// and both branches' gradients arrive UNSCALED: the precisions are applied here, where the gradient is read anyway —
//   table:        p1 (g_E0 + g_raw) + p2 g_user (user rows)      trust block: p2 g_small      gate matrices: p1 g_small
//   task weights: d/ds0 = -2 p1 loss1 + 2 (n_rec + 1) B,  d/ds1 = -2 p2 loss2 + T
// p1, p2 are read from prec[t & 1] (a snapshot: the thread that updates the task weights writes exp(-2 s_new) into the
// other slot for the next step, so no thread reads a weight another one is updating).  The same pass clears every
// accumulate-into buffer of the next step (g_raw, g_prop, the push target, g_user, g_small, the step's loss cells).
namespace {
struct DualAdamArgs {
    float *p, *m, *v;
    const float *g_E0, *g_raw;
    float *g_raw_w, *g_user, *g_small, *g_prop, *push_zero;     // cleared as they are read (push_zero may be NULL)
    float *loss, *loss_acc, *prec;
    int64_t n_table, n_user, n_trust, n_total;     // floats: table, user rows of it, trust block, whole arena (excl. padding)
    int32_t B, T, n_rec, slot, fixed;
    float w1, beta2, w2, bc2_sqrt, eps, step_size;
    float *att_copies;   // the fused batch kernel's copies of the two gate gradients ([n_att_copies][512], sample b -> copy b mod n):
    int n_att_copies;    //   summed into the gate parameters' gradient and cleared here; n_att_copies == 0: nothing to sum, but the
    int att_clear;       //   first att_clear floats are cleared (the other paths use the area for per-sample rows)
    float prop_div;      // > 0: g_E0 is the PLAIN last backward product and its g_prop / prop_div share is added here (g_prop is read
                         // before this pass clears it); 0: g_E0 already holds it; < 0: both backward products ran plain (L == 3) and
                         // the push target (push_zero) is added instead
    int clear_prop;      // 0: g_prop was not written this step (the fused middle of the L == 3 step forms no dense d loss / d light)
    int part;            // 0: the whole arena; the pipelined step splits the pass in two launches on two streams — 1: the item rows and
                         // the gate matrices (gradients of the rec branch alone), 2: the user rows, the trust block, the task weights
    int role;            // number of leading blocks that only sum the gate-gradient copies (0 or 2)
    float *zero_a, *zero_b;      // two further ranges cleared by this pass (the partitioned step: the trust head's table-gradient rows of
    int64_t n_zero_a, n_zero_b;  //   the OTHER ranks' users — g_user in front of and behind the rank's own rows); multiples of 4 floats
};

__global__ __launch_bounds__(256) void dual_task_adam_kernel(const DualAdamArgs a)
{
    // fixed: loss = loss1 + loss2 (main_11.py:69) — both precisions 1; the task weights get no gradient (torch's Adam skips a
    // parameter whose .grad is None: value and moments stay as they are)
    const float p1 = a.fixed ? 1.0f : a.prec[a.slot * 2], p2 = a.fixed ? 1.0f : a.prec[a.slot * 2 + 1];
    const int64_t n_gate_end = a.n_table + a.n_trust + 512;
    // ---- role blocks (the first a.role blocks): the 512 gate parameters when their gradient arrives as copies — each thread sums
    //      its column of the copies (independent, coalesced loads, 16 in flight), is the column's only reader and clears it.  The
    //      role sits on its own blocks, which do nothing else: as an iteration of the main loop it was a 4 us tail of the launch.
    if ((int)blockIdx.x < a.role) {
        const int jj = blockIdx.x * 256 + threadIdx.x;                    // 0 .. 511
        const int64_t j = a.n_trust + jj, i = a.n_table + j;
        float gs = a.g_small[j];
        a.g_small[j] = 0.0f;
        float *c = a.att_copies + jj;
        for (int c0 = 0; c0 < a.n_att_copies; c0 += 16) {
            float part[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) part[k] = c0 + k < a.n_att_copies ? c[(size_t)(c0 + k) * 512] : 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                gs += part[k];
                if (c0 + k < a.n_att_copies) c[(size_t)(c0 + k) * 512] = 0.0f;
            }
        }
        float P = a.p[i], M = a.m[i], V = a.v[i];
        adam1(P, p1 * gs, M, V, a.w1, a.beta2, a.w2, a.bc2_sqrt, a.eps, a.step_size);
        a.p[i] = P; a.m[i] = M; a.v[i] = V;
        return;
    }
    // ---- the table rows of this part, FOUR parameters per thread and step (float4: the scalar form moved 4 bytes per lane and load
    //      — 11.7 us for the 1 M-parameter Epinion2 arena where the LightGCN step's float4 pass takes 8):
    //      part 0: the whole table;  part 1 (what the rec branch alone owns): the item rows;  part 2: the user rows.
    //      (n_table and n_user are multiples of 64: whole rows)
    const int64_t stride = (int64_t)(gridDim.x - a.role) * blockDim.x;
    const int64_t tid = (int64_t)(blockIdx.x - a.role) * blockDim.x + threadIdx.x;
    const int64_t q_lo = (a.part == 1 ? a.n_user : 0) / 4, q_hi = (a.part == 2 ? a.n_user : a.n_table) / 4, q_user = a.n_user / 4;
    for (int64_t q = q_lo + tid; q < q_hi; q += stride) {
        float4 ge = reinterpret_cast<const float4 *>(a.g_E0)[q];
        if (a.prop_div > 0.0f) {
            const float4 gp = reinterpret_cast<const float4 *>(a.g_prop)[q];
            ge.x = ge.x + gp.x / a.prop_div; ge.y = ge.y + gp.y / a.prop_div; ge.z = ge.z + gp.z / a.prop_div; ge.w = ge.w + gp.w / a.prop_div;
        } else if (a.prop_div < 0.0f) {                                  // (read before this thread clears it below)
            const float4 pz = reinterpret_cast<const float4 *>(a.push_zero)[q];
            ge.x = ge.x + pz.x; ge.y = ge.y + pz.y; ge.z = ge.z + pz.z; ge.w = ge.w + pz.w;
        }
        const float4 gr = reinterpret_cast<const float4 *>(a.g_raw)[q];
        float4 g = make_float4(p1 * (ge.x + gr.x), p1 * (ge.y + gr.y), p1 * (ge.z + gr.z), p1 * (ge.w + gr.w));
        const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (q < q_user) {
            const float4 gu = reinterpret_cast<const float4 *>(a.g_user)[q];
            g.x = fmaf(p2, gu.x, g.x); g.y = fmaf(p2, gu.y, g.y); g.z = fmaf(p2, gu.z, g.z); g.w = fmaf(p2, gu.w, g.w);
            reinterpret_cast<float4 *>(a.g_user)[q] = zero4;
        }
        reinterpret_cast<float4 *>(a.g_raw_w)[q] = zero4;
        if (a.clear_prop) reinterpret_cast<float4 *>(a.g_prop)[q] = zero4;
        if (a.push_zero) reinterpret_cast<float4 *>(a.push_zero)[q] = zero4;         // (every part clears its own rows of the push target)
        float4 P = reinterpret_cast<float4 *>(a.p)[q], M = reinterpret_cast<float4 *>(a.m)[q], V = reinterpret_cast<float4 *>(a.v)[q];
        adam1(P.x, g.x, M.x, V.x, a.w1, a.beta2, a.w2, a.bc2_sqrt, a.eps, a.step_size);
        adam1(P.y, g.y, M.y, V.y, a.w1, a.beta2, a.w2, a.bc2_sqrt, a.eps, a.step_size);
        adam1(P.z, g.z, M.z, V.z, a.w1, a.beta2, a.w2, a.bc2_sqrt, a.eps, a.step_size);
        adam1(P.w, g.w, M.w, V.w, a.w1, a.beta2, a.w2, a.bc2_sqrt, a.eps, a.step_size);
        reinterpret_cast<float4 *>(a.p)[q] = P;
        reinterpret_cast<float4 *>(a.m)[q] = M;
        reinterpret_cast<float4 *>(a.v)[q] = V;
    }
    // ---- the small dense parameters behind the table, one per thread:  part 0: trust block [+ the gate parameters unless a role has
    //      them];  part 1: the gate parameters [ditto];  part 2: the trust block
    const int64_t gate_k = a.role ? 0 : 512;
    const int64_t j_lo = a.part == 1 ? a.n_trust : 0, j_hi = a.part == 2 ? a.n_trust : a.n_trust + gate_k;
    for (int64_t j = j_lo + tid; j < j_hi; j += stride) {
        const int64_t i = a.n_table + j;
        const float g = (j < a.n_trust ? p2 : p1) * a.g_small[j];
        a.g_small[j] = 0.0f;
        float P = a.p[i], M = a.m[i], V = a.v[i];
        adam1(P, g, M, V, a.w1, a.beta2, a.w2, a.bc2_sqrt, a.eps, a.step_size);
        a.p[i] = P; a.m[i] = M; a.v[i] = V;
    }
    if (a.part != 2 && a.n_att_copies == 0)
        for (int64_t k = tid; k < a.att_clear; k += stride) a.att_copies[k] = 0.0f;
    if (a.part != 1) {
        const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int64_t k = tid; k < a.n_zero_a / 4; k += stride) reinterpret_cast<float4 *>(a.zero_a)[k] = zero4;
        for (int64_t k = tid; k < a.n_zero_b / 4; k += stride) reinterpret_cast<float4 *>(a.zero_b)[k] = zero4;
    }
    if (a.part != 1 && (int)blockIdx.x == a.role && threadIdx.x == 0) {
        const float loss1 = a.loss[0] / (float)a.B, loss2 = a.loss[1];
        const float g[2] = {-2.0f * p1 * loss1 + 2.0f * (float)(a.n_rec + 1) * (float)a.B, -2.0f * p2 * loss2 + (float)a.T};
        for (int k = 0; k < 2 && !a.fixed; ++k) {
            const int64_t i = n_gate_end + k;
            float P = a.p[i], M = a.m[i], V = a.v[i];
            adam1(P, g[k], M, V, a.w1, a.beta2, a.w2, a.bc2_sqrt, a.eps, a.step_size);
            a.p[i] = P; a.m[i] = M; a.v[i] = V;
            a.prec[(1 - a.slot) * 2 + k] = expf(-2.0f * P);
        }
        a.loss_acc[0] += loss1;
        a.loss_acc[1] += loss2;
        a.loss[0] = 0.0f;
        a.loss[1] = 0.0f;
    }
}
}  // namespace

int spex::dual_task_adam(float *p, float *m, float *v, const float *g_E0, float *g_raw, float *g_user, float *g_small,
                         float *g_prop, float *push_zero, float *loss, float *loss_acc, float *prec, int64_t n_table, int64_t n_user,
                         int64_t n_trust,
                         int32_t B, int32_t T, int32_t n_rec, int32_t t, float lr, float beta1, float beta2, float eps, int fixed_weights,
                         void *stream, float prop_div, float *att_copies, int32_t n_att_copies, int32_t att_clear, int32_t part,
                         int32_t clear_prop, float *zero_a, int64_t n_zero_a, float *zero_b, int64_t n_zero_b)
{
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    if (!att_copies) n_att_copies = 0, att_clear = 0;
    const int role = n_att_copies > 0 && part != 2 ? 2 : 0;
    const DualAdamArgs a{p, m, v, g_E0, g_raw, g_raw, g_user, g_small, g_prop, push_zero, loss, loss_acc, prec, n_table, n_user, n_trust,
                         n_table + n_trust + 512 + 2, B, T, n_rec, t & 1, fixed_weights, 1.0f - beta1, beta2, 1.0f - beta2, (float)sqrt(bc2), eps,
                         (float)((double)lr / bc1), att_copies, n_att_copies, att_clear, prop_div, clear_prop, part, role,
                         zero_a, zero_b, zero_a ? n_zero_a : 0, zero_b ? n_zero_b : 0};
    SPEX_CHECK_ARG(((n_zero_a | n_zero_b) & 3) == 0 && n_zero_a >= 0 && n_zero_b >= 0 && ((((uintptr_t)zero_a) | ((uintptr_t)zero_b)) & 15) == 0,
                   "dual_task_adam: the extra ranges to clear must be whole float4s of 16-byte aligned buffers");
    SPEX_CHECK_ARG((n_table & 3) == 0 && (n_user & 3) == 0 && ((((uintptr_t)p) | ((uintptr_t)m) | ((uintptr_t)v) | ((uintptr_t)g_E0) | ((uintptr_t)g_raw) | ((uintptr_t)g_user)
                                                                  | ((uintptr_t)g_prop) | ((uintptr_t)push_zero)) & 15) == 0,
                   "dual_task_adam: the table blocks must be whole rows of 16-byte aligned buffers");
    const int64_t n_k = (part == 0 ? n_table : (part == 1 ? n_table - n_user : n_user)) / 4;      // float4 steps over the part's table rows
    int64_t blocks = (n_k + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(dual_task_adam_kernel, dim3((unsigned)(blocks + role)), dim3(256), 0, (hipStream_t)stream, a);
    SPEX_HIP(hipGetLastError());
    return SPEX_OK;
}
