// Collectives behind the C ABI (spex_comm_init / spex_allgather_rows; 8e: 1-D row partition with an RCCL
// all-gather of the layer's rows before each SpMM) and the row-partitioned training step as ONE native call.
//
// The reference's only analogue is the serial fold loop of --A_split (LightGCN_SPEX/code/utility1/model.py:84-89,
// dataloader.py:167-177): the same row blocks, one after the other on one device.  Here rank p owns a block of rows of A (and
// of A^T), of every layer's table and of the Adam moments; a layer is  all-gather of the current rows over xGMI  ->  local
// SpMM on the block.  Rounds 1-2 issued every collective from Python through torch.distributed; a binding in another language
// had no multi-GPU path at all and each step paid 3-4 Python-issued collectives against ~50 us of compute.
//
// RCCL is bound at RUN time (dlopen / dlsym), not linked: libspexhip.so keeps loading on a machine without RCCL (the
// single-GPU entry points need none), and inside a PyTorch process the copy of librccl torch has already loaded is the one
// that is used (RTLD_NOLOAD first) instead of a second copy with its own state.
//
//   spex_comm_unique_id      the 128-byte id rank 0 creates and the host hands to every rank (any transport: MPI, a file, a
//                            torch.distributed broadcast)
//   spex_comm_create         ncclCommInitRank on the calling thread's current device
//   spex_comm_allgather_rows_f32   equal padded shards (ncclAllGather), or — with the ranks' real row counts — a grouped
//                            send / recv of the REAL rows only, every rank writing straight into its slot of every peer's
//                            table: world - 1 concurrent point-to-point transfers, one per xGMI link of the full mesh, where
//                            a ring moves the same bytes through one link at a time
//   spex_comm_allreduce_sum_f32    in place (the owner-computes row exchange; loss / flag cells)
//   spex_partitioned_propagate_f32 / spex_partitioned_step_bce_f32   L x (exchange + SpMM) forward, the batch's rows fetched
//                            owner-computes, scoring, L x (exchange + SpMM) backward on A^T's blocks, Adam on the rank's rows —
//                            every exchange in place; fast path: the one-GPU step's schedule (rows-only last layer, push-form
//                            first backward product without an exchange: 2 L - 1 exchanges + one all-reduce)
//   spex_partitioned_dual_task_step_f32   the dual-task step (rec branch + gate + trust head) on the partition, same two schedules
#include <dlfcn.h>
#include <string.h>
#include <mutex>

#include <rccl/rccl.h>

#include "spex_common.h"

using namespace spex;

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        // SPEX_RCCL_LIB: the library to bind instead (a site's own build of RCCL; tests/stubs/rccl_record_stub.c — a recording
        // stand-in that lets the exchange's peer / count / offset arithmetic be checked for any world size without a wire)
        const char *forced = getenv("SPEX_RCCL_LIB");
        if (forced && forced[0]) {
            r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!r.lib) return;                                      // a named library that does not load is an error, not a fallback
        }
        for (const char *n : names)                                  // a copy the process already holds (PyTorch's) first
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) return;
#define SPEX_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name))
        SPEX_SYM(GetUniqueId, "ncclGetUniqueId");
        SPEX_SYM(CommInitRank, "ncclCommInitRank");
        SPEX_SYM(CommDestroy, "ncclCommDestroy");
        SPEX_SYM(AllGather, "ncclAllGather");
        SPEX_SYM(AllReduce, "ncclAllReduce");
        SPEX_SYM(Send, "ncclSend");
        SPEX_SYM(Recv, "ncclRecv");
        SPEX_SYM(GroupStart, "ncclGroupStart");
        SPEX_SYM(GroupEnd, "ncclGroupEnd");
        SPEX_SYM(GetErrorString, "ncclGetErrorString");
#undef SPEX_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce && r.Send && r.Recv && r.GroupStart
               && r.GroupEnd && r.GetErrorString;
    });
    return r;
}

#define SPEX_NCCL(call)                                                                                   \
    do {                                                                                                  \
        ncclResult_t r_ = (call);                                                                         \
        if (r_ != ncclSuccess) {                                                                          \
            ::spex::set_error("%s failed: %s (%s:%d)", #call, rccl().GetErrorString(r_), __FILE__, __LINE__); \
            return SPEX_ERR_COMM;                                                                         \
        }                                                                                                 \
    } while (0)

int need_rccl(const char *who)
{
    if (!rccl().ok) {
        spex::set_error("%s: librccl.so could not be loaded (%s)", who, rccl().lib ? "a symbol is missing" : dlerror());
        return SPEX_ERR_COMM;
    }
    return SPEX_OK;
}

}  // namespace

// Device-to-device copy of n floats on `stream`: the library's float4 kernel (x / 1.0f == x bit for bit) where the pointers allow it —
// a kernel launch costs the host ~1-2 us to issue, hipMemcpyAsync ~10 us on this runtime, and a partitioned step makes a dozen of them.
static int copy_f32(float *dst, const float *src, size_t n, void *stream)
{
    if (n == 0 || dst == src) return SPEX_OK;
    if (((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0) return spex::scale_div(src, dst, 1.0f, (int64_t)n, stream);
    SPEX_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SPEX_OK;
}

struct spex_comm {
    ncclComm_t comm = nullptr;
    int32_t rank = 0, world = 1;
    // world == 1: the exchange is a local copy and no RCCL collective is issued — unless SPEX_COMM_NO_SHORTCUT=1 was set when the
    // communicator was created (tests: RCCL accepts a one-rank communicator, so ncclAllGather / ncclAllReduce / an empty send-recv
    // group then really execute on a one-GPU box)
    bool shortcut = true;
};

extern "C" int spex_comm_unique_id(void *id_out)
{
    SPEX_CHECK_ARG(id_out, "spex_comm_unique_id: NULL output");
    if (int rc = need_rccl("spex_comm_unique_id")) return rc;
    ncclUniqueId id;
    SPEX_NCCL(rccl().GetUniqueId(&id));
    static_assert(sizeof(id) == SPEX_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return SPEX_OK;
}

extern "C" int spex_comm_create(int32_t rank, int32_t world, const void *unique_id, spex_comm_t **out)
{
    SPEX_CHECK_ARG(out && unique_id && world >= 1 && rank >= 0 && rank < world, "spex_comm_create: rank %d of %d", rank, world);
    if (int rc = need_rccl("spex_comm_create")) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    spex_comm *c = new spex_comm;
    c->rank = rank;
    c->world = world;
    const char *ns = getenv("SPEX_COMM_NO_SHORTCUT");
    c->shortcut = !(ns && ns[0] == '1');
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        spex::set_error("ncclCommInitRank failed: %s", rccl().GetErrorString(r));
        delete c;
        return SPEX_ERR_COMM;
    }
    *out = c;
    return SPEX_OK;
}

extern "C" int spex_comm_destroy(spex_comm_t *c)
{
    if (!c) return SPEX_OK;
    if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
    delete c;
    return SPEX_OK;
}

extern "C" int spex_comm_info(const spex_comm_t *c, int32_t *rank, int32_t *world)
{
    SPEX_CHECK_ARG(c, "spex_comm_info: NULL communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return SPEX_OK;
}

extern "C" int spex_comm_allgather_rows_f32(spex_comm_t *c, const float *send, float *recv, int64_t max_rows, int32_t d,
                                            const int32_t *rows_per_rank, void *stream)
{
    SPEX_CHECK_ARG(c && send && recv && max_rows >= 0 && d >= 1, "spex_comm_allgather_rows_f32: bad argument");
    if (max_rows == 0) return SPEX_OK;
    const size_t slot = (size_t)max_rows * d;
    if (c->world == 1 && c->shortcut) {
        const int64_t rows = rows_per_rank ? rows_per_rank[0] : max_rows;
        SPEX_CHECK_ARG(rows >= 0 && rows <= max_rows, "spex_comm_allgather_rows_f32: %lld rows in a slot of %lld", (long long)rows, (long long)max_rows);
        return copy_f32(recv, send, (size_t)rows * d, stream);
    }
    if (!rows_per_rank) {                                            // equal padded shards: the plain collective
        SPEX_NCCL(rccl().AllGather(send, recv, slot, ncclFloat32, c->comm, (hipStream_t)stream));
        return SPEX_OK;
    }
    // real rows only, point to point: rank p's rows land in slot p of every rank's table (the slots keep the padded stride;
    // their tails are never read).  One group = world - 1 sends + world - 1 receives in flight at once, one per link.
    for (int q = 0; q < c->world; ++q)
        SPEX_CHECK_ARG(rows_per_rank[q] >= 0 && rows_per_rank[q] <= max_rows, "spex_comm_allgather_rows_f32: rank %d has %d rows in a slot of %lld", q,
                       rows_per_rank[q], (long long)max_rows);
    const size_t mine = (size_t)rows_per_rank[c->rank] * d;
    if (int rc = copy_f32(recv + c->rank * slot, send, mine, stream)) return rc;
    SPEX_NCCL(rccl().GroupStart());
    ncclResult_t bad = ncclSuccess;
    const char *what = "";
    for (int q = 0; q < c->world && bad == ncclSuccess; ++q) {
        if (q == c->rank) continue;
        if (mine && (bad = rccl().Send(send, mine, ncclFloat32, q, c->comm, (hipStream_t)stream)) != ncclSuccess) { what = "ncclSend"; break; }
        const size_t theirs = (size_t)rows_per_rank[q] * d;
        if (theirs && (bad = rccl().Recv(recv + q * slot, theirs, ncclFloat32, q, c->comm, (hipStream_t)stream)) != ncclSuccess) what = "ncclRecv";
    }
    if (bad != ncclSuccess) {
        // close the group whatever happened: a group left open on this thread would swallow every later collective (they would be
        // queued into it and never launch — a hang instead of an error)
        (void)rccl().GroupEnd();
        spex::set_error("%s failed inside the all-gather's group: %s", what, rccl().GetErrorString(bad));
        return SPEX_ERR_COMM;
    }
    SPEX_NCCL(rccl().GroupEnd());
    return SPEX_OK;
}

extern "C" int spex_comm_allreduce_sum_f32(spex_comm_t *c, float *buf, int64_t n, void *stream)
{
    SPEX_CHECK_ARG(c && (buf || n == 0) && n >= 0, "spex_comm_allreduce_sum_f32: bad argument");
    if (n == 0 || (c->world == 1 && c->shortcut)) return SPEX_OK;
    SPEX_NCCL(rccl().AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
    return SPEX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The row-partitioned propagation and training step (spex_amd/dist.py: PartitionedLightGCN.propagate / PartitionedStepper)
// issued from native code.  Every launch is one of the library's own entry points.
#define SPEX_TRY(call)                  \
    do {                                \
        int rc_ = (call);               \
        if (rc_ != SPEX_OK) return rc_; \
    } while (0)

static int check_step(const spex_partitioned_step_t *s, const char *who)
{
    SPEX_CHECK_ARG(s && s->graph && s->graph_t && s->comm && s->E0 && s->light_out && s->gathered && s->gathered1,
                   "%s: NULL field in the step descriptor", who);
    SPEX_CHECK_ARG(s->gathered != s->gathered1, "%s: gathered and gathered1 are two tables", who);
    SPEX_CHECK_ARG(s->d == 64 && s->L >= 1 && s->n_local >= 0 && s->n_local <= s->max_rows, "%s: d=%d L=%d n_local=%d max_rows=%d", who, s->d,
                   s->L, s->n_local, s->max_rows);
    SPEX_CHECK_ARG(s->graph->n_rows == s->n_local && s->graph_t->n_rows == s->n_local
                       && s->graph->n_cols == s->comm->world * s->max_rows && s->graph_t->n_cols == s->graph->n_cols,
                   "%s: the row blocks must be n_local x (world * max_rows) in the padded layout", who);
    // edge dropout (utility1/model.py:46-64): the same mask on both handles — the blocks carry the entries' GLOBAL edge ids, so every
    // rank drops the same edges of A and of A^T —, graph_t a handle of its own (a masked operator is not symmetric); the steps then take
    // the launch-by-launch schedule, whose products are all whole-block launches
    SPEX_CHECK_ARG(s->graph->mask_mode == s->graph_t->mask_mode
                       && (s->graph->mask_mode == 0
                           || (s->graph_t != s->graph && s->graph->keep_prob == s->graph_t->keep_prob && s->graph->seed == s->graph_t->seed
                               && s->graph->keep == s->graph_t->keep)),
                   "%s: edge dropout needs the same mask on graph and graph_t, graph_t a separate handle carrying the edge-id permutation", who);
    return SPEX_OK;
}

// One exchange IN PLACE: the rank's rows lie in their own slot of `table` (rows rank * max_rows ..) — the SpMM that produced them wrote
// them there — or are copied there from `local` first; the peers' rows arrive in the other slots.  No staging buffer and, for a layer's
// output, no copy: the collective sends from the slot it also receives around (ncclAllGather's in-place form; the point-to-point form
// sends the real rows from where they lie).  Round 3 wrote every layer into a send buffer and copied it into the slot: 2 L + 1 copies
// of the rank's rows per step.
static int exchange_in_place(spex_comm_t *comm, const int32_t *rows_per_rank, const float *local, int64_t n_local, int64_t max_rows, int32_t d,
                             float *table, void *stream)
{
    float *own = table + (size_t)comm->rank * max_rows * d;
    if (local && local != own)
        if (int rc = copy_f32(own, local, (size_t)n_local * d, stream)) return rc;
    return spex_comm_allgather_rows_f32(comm, own, table, max_rows, d, rows_per_rank, stream);
}

// edge dropout and the fast path: its rows-only layer reads `graph` and its push reads `graph_push` entry by entry with the handles' keep
// rule — the push structure must then carry the same mask as the blocks (and the entries' global edge ids); otherwise a masked step
// takes the launch-by-launch schedule, whose products are whole-block launches
static inline bool push_mask_matches(const spex_graph_t *g, const spex_graph_t *push, int64_t n_loc)
{
    if (g->mask_mode == 0) return push == nullptr || push->mask_mode == 0;
    if (n_loc == 0) return true;                                     // (a rank without rows pushes nothing)
    return push != nullptr && push->mask_mode == g->mask_mode && push->keep_prob == g->keep_prob && push->seed == g->seed && push->keep == g->keep;
}

static inline float *own_slot(const spex_comm_t *comm, float *table, int64_t max_rows, int32_t d)
{
    return table + (size_t)comm->rank * max_rows * d;
}

extern "C" int spex_partitioned_propagate_f32(spex_partitioned_step_t *s, void *stream)
{
    SPEX_TRY(check_step(s, "spex_partitioned_propagate_f32"));
    if (s->n_local == 0 && s->comm->world == 1) return SPEX_OK;
    float *T[2] = {s->gathered, s->gathered1};
    for (int32_t l = 0; l < s->L; ++l) {
        // layer l reads T[l & 1] (its own slot: E^0 copied in for l == 0, the previous layer's output otherwise) and writes its output
        // into the own slot of the other table
        SPEX_TRY(exchange_in_place(s->comm, s->rows_per_rank, l == 0 ? s->E0 : nullptr, s->n_local, s->max_rows, s->d, T[l & 1], stream));
        const bool last = l == s->L - 1;
        float *nxt = last ? nullptr : own_slot(s->comm, T[(l + 1) & 1], s->max_rows, s->d);
        if (s->n_local)
            SPEX_TRY(spex_spmm_f32(s->graph, T[l & 1], nxt, nullptr, 1.0f, l == 0 ? s->E0 : s->light_out, s->light_out,
                                   last ? (float)(s->L + 1) : 1.0f, s->d, stream));
    }
    return SPEX_OK;
}

extern "C" int spex_partitioned_step_bce_f32(spex_partitioned_step_t *s, const int64_t *pos, const float *labels, int32_t B,
                                             float *loss_sum, void *stream)
{
    SPEX_TRY(check_step(s, "spex_partitioned_step_bce_f32"));
    SPEX_CHECK_ARG(s->m && s->v && s->g_local && s->gs && s->grad_E0 && s->rows && s->grad_rows && s->arange,
                   "spex_partitioned_step_bce_f32: NULL field in the step descriptor");
    SPEX_CHECK_ARG(pos && labels && loss_sum && B >= 1 && s->slot_capacity >= 2 * B, "spex_partitioned_step_bce_f32: batch of %d for a slot capacity of %d",
                   B, s->slot_capacity);
    const int32_t d = s->d, L = s->L;
    const int64_t lo = (int64_t)s->comm->rank * s->max_rows, n_loc = s->n_local;
    const bool det = (s->flags & SPEX_STEP_DETERMINISTIC) != 0;
    const int64_t sz = n_loc * d;
    // ---- the fast path: spex_lightgcn_step_bce_f32's schedule on the partition (see spex_partitioned_dual_task_step_f32 below, whose rec
    //      branch is this with the gate in the middle).  The same choice on every rank: the two schedules differ in their collectives.
    const bool fast = !det && L >= 2 && s->gathered2 != nullptr && (s->graph_push != nullptr || n_loc == 0)
                      && push_mask_matches(s->graph, s->graph_push, n_loc);
    if (fast) {
        if (n_loc)
            SPEX_CHECK_ARG(s->graph_push->n_rows == s->comm->world * s->max_rows && s->graph_push->n_cols == n_loc,
                           "spex_partitioned_step_bce_f32: graph_push must be the (world * max_rows) x n_local transpose of the rank's block of A^T");
        SPEX_CHECK_ARG(s->gathered2 != s->gathered && s->gathered2 != s->gathered1, "spex_partitioned_step_bce_f32: gathered2 is a third table");
        float *Tb[2] = {s->gathered, s->gathered1};
        auto own = [&](float *table) { return own_slot(s->comm, table, s->max_rows, d); };
        // forward layers 1 .. L-1 over the block (plain form for L <= 3: E^1, E^2 stay in the two tables' own slots), E^0 through table 1
        const bool plain = L <= 3;
        for (int32_t l = 0; l + 1 < L; ++l) {
            float *X = Tb[(l + 1) & 1];                                  // E^0 -> gathered1, E^1 -> gathered, E^2 -> gathered1, ...
            SPEX_TRY(exchange_in_place(s->comm, s->rows_per_rank, l == 0 ? s->E0 : nullptr, n_loc, s->max_rows, d, X, stream));
            if (!n_loc) continue;
            if (plain) SPEX_TRY(spex_spmm_f32(s->graph, X, own(Tb[l & 1]), nullptr, 1.0f, nullptr, nullptr, 1.0f, d, stream));
            else SPEX_TRY(spex_spmm_f32(s->graph, X, own(Tb[l & 1]), nullptr, 1.0f, l == 0 ? s->E0 : s->light_out, s->light_out, 1.0f, d, stream));
        }
        // the last layer at the batch's rows on their owners, one all-reduce of 2B rows
        float *Xl = Tb[(L - 2) & 1];
        SPEX_TRY(exchange_in_place(s->comm, s->rows_per_rank, nullptr, n_loc, s->max_rows, d, Xl, stream));
        // (L == 3: E^1 lies in gathered's own slot, E^2 in gathered1's — E^0's copy there was overwritten by layer 2's output, E^0 itself
        //  is the parameter block; L == 2: E^1 in gathered's own slot)
        SPEX_TRY(spex_spmm_owned_rows_f32(s->graph, Xl, pos, 2 * B, lo, plain ? s->E0 : s->light_out, plain ? own(Tb[0]) : nullptr,
                                          plain && L == 3 ? own(Tb[1]) : nullptr, (float)(L + 1), nullptr, s->rows, nullptr, d, stream));
        SPEX_TRY(spex_comm_allreduce_sum_f32(s->comm, s->rows, (int64_t)2 * B * d, stream));
        // scores + BCE + gradient rows on the compact rows (replicated), the owned rows' shares added; the first backward product in
        // push form through the rank's own columns of A — no exchange; P = gathered2's own slot, kept all-zero by the Adam pass
        float *P = own(s->gathered2);
        const bool all_plain = L == 3;
        SPEX_TRY(spex::rows_train_push(n_loc ? s->graph_push : nullptr, nullptr, s->rows, nullptr, nullptr, pos, lo, (int32_t)n_loc, labels, B,
                                       1.0f / (float)B, 1.0f / (float)(L + 1), loss_sum, all_plain ? nullptr : s->g_local, P, nullptr, nullptr, 1,
                                       stream));
        float *Xb = s->gathered2;
        for (int32_t l = L - 2, k = 0; l >= 0; --l, ++k) {
            SPEX_TRY(exchange_in_place(s->comm, s->rows_per_rank, nullptr, n_loc, s->max_rows, d, Xb, stream));
            float *nxt = l == 0 ? s->grad_E0 : own(Tb[k & 1]);
            if (n_loc) {
                if (l == 0 || all_plain) SPEX_TRY(spex_spmm_f32(s->graph_t, Xb, nxt, nullptr, 1.0f, nullptr, nullptr, 1.0f, d, stream));
                else SPEX_TRY(spex_spmm_f32(s->graph_t, Xb, nxt, s->g_local, (float)(L + 1), nullptr, nullptr, 1.0f, d, stream));
            }
            Xb = Tb[k & 1];
        }
        // Adam adds the plain last product's share (g / (L+1); L == 3: the push target P) and clears both tables for the next step
        if (sz)
            SPEX_TRY(spex::adam_step_z2(s->E0, s->grad_E0, s->m, s->v, sz, s->t + 1, s->lr, s->beta1, s->beta2, s->eps,
                                        all_plain ? nullptr : s->g_local, P, stream, nullptr, 0, nullptr, nullptr, all_plain ? P : s->g_local,
                                        all_plain ? 1.0f : (float)(L + 1)));
        s->t += 1;
        return SPEX_OK;
    }
    // ---- forward: L x (exchange, SpMM on the rank's rows), the batch's 2B rows fetched owner-computes (each rank contributes the
    //      rows it owns to a zero-filled buffer, one small all-reduce adds them up), scoring on the compact rows
    SPEX_TRY(spex_partitioned_propagate_f32(s, stream));
    SPEX_TRY(spex_gather_owned_rows_f32(s->light_out, pos, 2 * (int64_t)B, lo, n_loc, d, s->rows, stream));
    SPEX_TRY(spex_comm_allreduce_sum_f32(s->comm, s->rows, (int64_t)2 * B * d, stream));
    const float *users = s->rows, *items = s->rows + (size_t)B * d;    // slot b: the user row of sample b, slot B + b: its item row
    if (det) {
        // per-sample gradient rows (plain stores), added per owned table row in ascending slot order: no float atomics
        // (and per-sample losses, summed in sample order: the gathered table is free between the forward's last SpMM and the
        //  backward's first exchange — its head holds them)
        SPEX_CHECK_ARG((int64_t)s->comm->world * s->max_rows * d >= B, "spex_partitioned_step_bce_f32: gathered table smaller than the batch");
        float *loss_rows = s->gathered;
        SPEX_TRY(spex::score_bce_slots_rows(users, items, d, d, B, B, s->arange, s->arange, labels, B, d, loss_rows, 1.0f / (float)B,
                                            s->grad_rows, d, stream));
        SPEX_TRY(spex::sum_ordered(loss_rows, B, 1.0f, loss_sum, 1, stream));
        SPEX_TRY(spex_reduce_slots_f32(pos, 2 * B, -lo, nullptr, 0, 0, (int32_t)n_loc, s->grad_rows, d, 1.0f, s->g_local, 0, d, stream));
    } else {
        // (per-sample rows by plain stores, then the owned ones added with atomics: no buffer has to be all-zero between steps)
        SPEX_TRY(spex_score_bce_slots_f32(users, items, d, d, B, B, s->arange, s->arange, labels, B, d, loss_sum, nullptr, nullptr, 1.0f / (float)B,
                                          s->grad_rows, d, stream));
        SPEX_TRY(spex_scatter_add_owned_rows_f32(s->grad_rows, pos, 2 * (int64_t)B, lo, n_loc, d, s->g_local, 0, stream));
    }
    // ---- backward: G_L = g / (L + 1);  G_l = g / (L + 1) + A^T G_{l+1} on the row blocks of A^T, one exchange per layer
    SPEX_TRY(spex::scale_div(s->g_local, s->gs, (float)(L + 1), sz, stream));
    float *T[2] = {s->gathered, s->gathered1};
    for (int32_t l = L - 1, k = 0; l >= 0; --l, ++k) {
        SPEX_TRY(exchange_in_place(s->comm, s->rows_per_rank, k == 0 ? s->gs : nullptr, n_loc, s->max_rows, d, T[k & 1], stream));
        float *nxt = l == 0 ? s->grad_E0 : own_slot(s->comm, T[(k + 1) & 1], s->max_rows, d);
        if (n_loc) SPEX_TRY(spex_spmm_f32(s->graph_t, T[k & 1], nxt, s->gs, 1.0f, nullptr, nullptr, 1.0f, d, stream));
    }
    // ---- Adam on the rank's rows (its pass clears the gradient rows again; det: g_local is overwritten per touched row, so it is
    //      cleared densely here as well)
    if (sz)
        SPEX_TRY(spex::adam_step_z2(s->E0, s->grad_E0, s->m, s->v, sz, s->t + 1, s->lr, s->beta1, s->beta2, s->eps, s->g_local, nullptr,
                                    stream));
    s->t += 1;
    return SPEX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// BASELINE config 5 on the row partition (main_auto_expert_s.py:53-91): see include/spex_hip.h, spex_partitioned_dual_step_t.
// Every launch is one of the library's own entry points.  Collectives per step: fast path 2 L - 1 exchanges + one all-reduce (the
// backward's first product needs none: every rank holds the batch's gradient rows), deterministic path 2 L + one all-reduce.
extern "C" int spex_partitioned_dual_task_step_f32(spex_partitioned_dual_step_t *s, const int64_t *pos, const float *labels, int32_t B,
                                                   const int64_t *seq, const int64_t *seq_l, const int64_t *targets, int32_t T, void *stream)
{
    const char *who = "spex_partitioned_dual_task_step_f32";
    SPEX_CHECK_ARG(s && s->graph && s->graph_t && s->comm && s->params && s->m && s->v && s->light && s->g_prop && s->g_raw && s->gs && s->g_E0
                       && s->gathered && s->gathered1 && s->gathered0 && s->user_pos && s->user_table && s->rows && s->mixed_slots && s->grad_slots
                       && s->g_prop_slots && s->g_raw_slots && s->loss_rows && s->att_parts && s->arange && s->g_user && s->g_small && s->a2
                       && s->trust_ws && s->dscore && s->loss_b && s->loss && s->loss_acc && s->precision,
                   "%s: NULL field in the step descriptor", who);
    SPEX_CHECK_ARG(s->gathered != s->gathered1 && s->gathered != s->gathered0 && s->gathered1 != s->gathered0 && s->gathered2 != s->gathered
                       && s->gathered2 != s->gathered1 && s->gathered2 != s->gathered0,
                   "%s: gathered, gathered0, gathered1 and gathered2 are separate tables", who);
    SPEX_CHECK_ARG(pos && labels && B >= 1 && s->slot_capacity >= 2 * B, "%s: batch of %d for a slot capacity of %d", who, B, s->slot_capacity);
    SPEX_CHECK_ARG(T >= 0 && T <= s->path_capacity && (T == 0 || (seq && seq_l && targets)), "%s: T=%d paths (capacity %d) or NULL path pointer", who,
                   T, s->path_capacity);
    const int32_t d = 64, L = s->L, H = s->n_heads, n_u = s->n_user_rows;
    const int64_t n_loc = s->n_local, max_rows = s->max_rows, world = s->comm->world, lo = (int64_t)s->comm->rank * max_rows;
    SPEX_CHECK_ARG(s->d == 64 && L >= 1 && n_loc >= 0 && n_loc <= max_rows && n_u >= 2, "%s: d=%d L=%d n_local=%d max_rows=%d n_user_rows=%d", who, s->d,
                   L, s->n_local, s->max_rows, n_u);
    SPEX_CHECK_ARG(s->graph->n_rows == n_loc && s->graph_t->n_rows == n_loc && s->graph->n_cols == world * max_rows
                       && s->graph_t->n_cols == s->graph->n_cols,
                   "%s: the row blocks must be n_local x (world * max_rows) in the padded layout", who);
    SPEX_CHECK_ARG(s->graph->mask_mode == s->graph_t->mask_mode
                       && (s->graph->mask_mode == 0
                           || (s->graph_t != s->graph && s->graph->keep_prob == s->graph_t->keep_prob && s->graph->seed == s->graph_t->seed
                               && s->graph->keep == s->graph_t->keep)),
                   "%s: edge dropout needs the same mask on graph and graph_t, graph_t a separate handle carrying the edge-id permutation", who);
    SPEX_CHECK_ARG(s->n_local_users >= 0 && s->n_local_users <= n_loc && s->user_lo >= 0 && s->user_lo + s->n_local_users <= n_u,
                   "%s: %d local user rows from user row %d of %d", who, s->n_local_users, s->user_lo, n_u);
    const int64_t n_trust = spex_trust_param_count(d, H);
    SPEX_CHECK_ARG(n_trust > 0, "%s: unsupported number of heads %d", who, H);
    const bool det = (s->flags & SPEX_STEP_DETERMINISTIC) != 0;
    // the fast path (spex_dual_task_step_f32's schedule on the partition): needs the push structure, its zero-kept table and L >= 2
    //  (a rank without rows has nothing to push and needs no structure — but must walk the same sequence of collectives as its peers:
    //   the choice depends on gathered2 and the flags, which the caller sets alike on every rank)
    //  (under edge dropout — model_expert_s.py:104-109, the rec branch only — the fast path needs the mask on the push structure too;
    //   without it the launch-by-launch schedule, whose products are all whole-block launches)
    const bool fast = !det && L >= 2 && s->gathered2 != nullptr && (s->graph_push != nullptr || n_loc == 0)
                      && push_mask_matches(s->graph, s->graph_push, n_loc);
    if (fast && n_loc)
        SPEX_CHECK_ARG(s->graph_push->n_rows == world * max_rows && s->graph_push->n_cols == n_loc,
                       "%s: graph_push must be the (world * max_rows) x n_local transpose of the rank's block of A^T", who);
    const size_t sz = (size_t)n_loc * d;
    float *E0 = s->params, *trust_p = E0 + sz, *att1 = trust_p + n_trust, *att2 = att1 + 4 * d;
    float *g_att1 = s->g_small + n_trust, *g_att2 = g_att1 + 4 * d;
    float *Tb[2] = {s->gathered, s->gathered1};
    auto own = [&](float *table) { return own_slot(s->comm, table, max_rows, d); };

    // ---- layer 1's exchange first: its table is E^0 of every rank — what the trust branch reads
    SPEX_TRY(exchange_in_place(s->comm, s->rows_per_rank, E0, n_loc, max_rows, d, s->gathered0, stream));
    // ---- trust branch (model_expert_s.py:170-192) on the gathered user block, redundantly on every rank; beside the rec branch
    //      when the caller gave a second stream
    const bool two_streams = s->side_stream != nullptr && s->side_stream != stream && T > 0;
    // (g_user is all-zero here: the Adam pass clears the rank's own user rows as it reads them and the other ranks' rows beside them.
    //  The user block is gathered on the CALLER's stream, in front of the fork: with a launch of its own in front of it the trust kernel
    //  reached the GPU while layer 1's product held every CU — its 15 workgroups want a whole CU's registers each and waited the launch
    //  out, ~10 us on the side branch, which at world size 1 is the step's longest chain; forked behind the gather both start together.)
    //  (the launch-by-launch schedule's rec branch is the longer chain by far: there the gather stays on the side branch)
    const bool gather_first = fast && two_streams;
    if (gather_first) SPEX_TRY(spex_gather_owned_rows_f32(s->gathered0, s->user_pos, n_u, 0, world * max_rows, d, s->user_table, stream));
    auto trust_branch = [&](void *st) -> int {
        if (!gather_first) SPEX_TRY(spex_gather_owned_rows_f32(s->gathered0, s->user_pos, n_u, 0, world * max_rows, d, s->user_table, st));
        return spex::trust_head_train(s->user_table, n_u, trust_p, seq, seq_l, targets, T, s->path_len, d, H, s->hybrid, 1.0f, nullptr, s->a2,
                                      s->dscore, s->loss_b, s->trust_ws, s->loss + 1, 0, s->g_small, s->g_user, st == stream && !det ? 8 : 1, st);
    };
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    int rc_trust = SPEX_OK;
    bool forked = false;
    if (two_streams) {
        if (!s->ev_fork) { hipEvent_t e = nullptr; SPEX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); s->ev_fork = e; }
        if (!s->ev_join) { hipEvent_t e = nullptr; SPEX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); s->ev_join = e; }
        fork_ev = (hipEvent_t)s->ev_fork; join_ev = (hipEvent_t)s->ev_join;
        SPEX_HIP(hipEventRecord(fork_ev, (hipStream_t)stream));
        SPEX_HIP(hipStreamWaitEvent((hipStream_t)s->side_stream, fork_ev, 0));
        forked = true;
        rc_trust = trust_branch(s->side_stream);
        if (hipEventRecord(join_ev, (hipStream_t)s->side_stream) != hipSuccess && rc_trust == SPEX_OK) rc_trust = SPEX_ERR_HIP;
    }
    // (the gate gradients' copies of the fast path live in grad_slots — per-sample rows on the deterministic path —, summed and cleared
    //  by the Adam pass, which clears the area after the deterministic path too: a descriptor may change its flags between steps)
    const int32_t att_copies_max = s->slot_capacity / 8 < 64 ? s->slot_capacity / 8 : 64;
    int32_t att_copies_used = 0;
    auto rec_fast = [&]() -> int {
        // ---- forward (model_expert_s.py:95-126): layers 1 .. L-1 over the rank's block, every output written into its own slot of the
        //      next exchange's table; plain form for L <= 3 (the layer tables E^1, E^2 stay in the two tables' own slots and the layer
        //      sum is formed at the batch's rows), running sum in `light` beyond
        const bool plain = L <= 3;
        for (int32_t l = 0; l + 1 < L; ++l) {
            const float *X = l == 0 ? s->gathered0 : Tb[(l - 1) & 1];
            if (l > 0) SPEX_TRY(exchange_in_place(s->comm, s->rows_per_rank, nullptr, n_loc, max_rows, d, Tb[(l - 1) & 1], stream));
            if (!n_loc) continue;
            if (plain) SPEX_TRY(spex_spmm_f32(s->graph, X, own(Tb[l & 1]), nullptr, 1.0f, nullptr, nullptr, 1.0f, d, stream));
            else SPEX_TRY(spex_spmm_f32(s->graph, X, own(Tb[l & 1]), nullptr, 1.0f, l == 0 ? E0 : s->light, s->light, 1.0f, d, stream));
        }
        // ---- the LAST layer at the batch's rows only, on their owners: layer mean + raw rows, compact, zero elsewhere — ONE launch and
        //      ONE all-reduce of 4B rows put the batch's rows of E^0 and of the propagated table on every rank
        float *Xl = Tb[(L - 2) & 1];
        SPEX_TRY(exchange_in_place(s->comm, s->rows_per_rank, nullptr, n_loc, max_rows, d, Xl, stream));
        float *rows_raw = s->rows, *rows_prop = s->rows + (size_t)2 * B * d;
        SPEX_TRY(spex_spmm_owned_rows_f32(s->graph, Xl, pos, 2 * B, lo, plain ? E0 : s->light, plain ? own(Tb[0]) : nullptr,
                                          plain && L == 3 ? own(Tb[1]) : nullptr, (float)(L + 1), E0, rows_prop, rows_raw, d, stream));
        SPEX_TRY(spex_comm_allreduce_sum_f32(s->comm, s->rows, (int64_t)4 * B * d, stream));
        // ---- gate, scores, BCE, the gate's backward on the compact rows — replicated: loss and gate gradients are complete on every
        //      rank —, the owned rows' shares added where they belong, and the backward's first product in push form, WITHOUT an
        //      exchange: every rank holds all 2B gradient rows and pushes them through its own columns of A (graph_push: row = a
        //      position of the padded layout, columns = the rank's rows): P = (g + A^T g) / (L+1) on the rank's rows.  ONE launch
        //      (batch.hip: rows_train_push_kernel); P = the own slot of gathered2, all-zero here (cleared by the previous step's Adam)
        float *P = own(s->gathered2);
        SPEX_TRY(spex::rows_train_push(n_loc ? s->graph_push : nullptr, rows_raw, rows_prop, att1, att2, pos, lo, (int32_t)n_loc, labels, B,
                                       1.0f / (float)B, 1.0f / (float)(L + 1), s->loss, L == 3 ? nullptr : s->g_prop, P, s->g_raw,
                                       att_copies_max >= 1 ? s->grad_slots : g_att1, att_copies_max >= 1 ? att_copies_max : 1, stream));
        att_copies_used = att_copies_max;
        // ---- the L-1 pull-form products on A^T's block: P's table is exchanged in place, the outputs alternate between the two
        //      forward tables' own slots (dead by now); the last one plain — the Adam pass adds its g_prop / (L+1) share; L == 3: both
        //      plain, the Adam pass adds P instead (spex_dual_task_step_f32's schedule)
        float *Xb = s->gathered2;
        for (int32_t l = L - 2, k = 0; l >= 0; --l, ++k) {
            SPEX_TRY(exchange_in_place(s->comm, s->rows_per_rank, nullptr, n_loc, max_rows, d, Xb, stream));
            float *nxt = l == 0 ? s->g_E0 : own(Tb[k & 1]);
            if (n_loc) {
                if (l == 0 || L == 3) SPEX_TRY(spex_spmm_f32(s->graph_t, Xb, nxt, nullptr, 1.0f, nullptr, nullptr, 1.0f, d, stream));
                else SPEX_TRY(spex_spmm_f32(s->graph_t, Xb, nxt, s->g_prop, (float)(L + 1), nullptr, nullptr, 1.0f, d, stream));
            }
            Xb = Tb[k & 1];
        }
        return SPEX_OK;
    };
    auto rec_branch = [&]() -> int {
        // ---- forward (model_expert_s.py:95-126): L x (exchange in place, SpMM on the rank's rows with the running sum fused)
        for (int32_t l = 0; l < L; ++l) {
            const float *X = l == 0 ? s->gathered0 : Tb[(l - 1) & 1];
            if (l > 0) SPEX_TRY(exchange_in_place(s->comm, s->rows_per_rank, nullptr, n_loc, max_rows, d, Tb[(l - 1) & 1], stream));
            const bool last = l == L - 1;
            if (n_loc)
                SPEX_TRY(spex_spmm_f32(s->graph, X, last ? nullptr : own(Tb[l & 1]), nullptr, 1.0f, l == 0 ? E0 : s->light, s->light,
                                       last ? (float)(L + 1) : 1.0f, d, stream));
        }
        // ---- the batch's rows of E^0 and of the propagated table on every rank: owner-computes, ONE all-reduce of 4B rows
        float *rows_raw = s->rows, *rows_prop = s->rows + (size_t)2 * B * d;
        SPEX_TRY(spex_gather_owned_rows_f32(E0, pos, 2 * (int64_t)B, lo, n_loc, d, rows_raw, stream));
        SPEX_TRY(spex_gather_owned_rows_f32(s->light, pos, 2 * (int64_t)B, lo, n_loc, d, rows_prop, stream));
        SPEX_TRY(spex_comm_allreduce_sum_f32(s->comm, s->rows, (int64_t)4 * B * d, stream));
        // ---- gate, scores, BCE, and the gate's backward on the compact rows (slot b: the user row of sample b — att_exp1 —, slot
        //      B + b: its item row — att_exp2): the same 2B rows on every rank, so g_att1 / g_att2 are complete everywhere
        SPEX_TRY(spex_expert_gate_rows_f32(rows_raw, rows_prop, att1, att2, s->arange, B, 0, s->arange, B, B, B, 2 * (int64_t)B, d, s->mixed_slots,
                                           stream));
        SPEX_TRY(spex::score_bce_slots_rows(s->mixed_slots, s->mixed_slots + (size_t)B * d, d, d, B, B, s->arange, s->arange, labels, B, d,
                                            s->loss_rows, 1.0f / (float)B, s->grad_slots, d, stream));
        SPEX_TRY(spex::sum_ordered(s->loss_rows, B, 1.0f, s->loss, 1, stream));
        SPEX_TRY(spex_expert_gate_rows_bwd_det_f32(rows_raw, rows_prop, att1, att2, s->arange, B, 0, s->arange, B, B, B, 2 * (int64_t)B, d,
                                                   s->grad_slots, d, s->g_prop_slots, s->g_raw_slots, s->att_parts, stream));
        const int32_t n_parts = spex_expert_gate_rows_bwd_parts(2 * B);
        SPEX_TRY(spex::sum_parts(s->att_parts, n_parts, 512, 512, g_att1, 0, stream));     // (g_att2 follows g_att1 in g_small: one launch for both)
        (void)g_att2;
        // ---- the rows this rank owns into its gradient blocks
        if (det) {
            SPEX_TRY(spex_reduce_slots_f32(pos, 2 * B, -lo, nullptr, 0, 0, (int32_t)n_loc, s->g_prop_slots, d, 1.0f, s->g_prop, 0, d, stream));
            SPEX_TRY(spex_reduce_slots_f32(pos, 2 * B, -lo, nullptr, 0, 0, (int32_t)n_loc, s->g_raw_slots, d, 1.0f, s->g_raw, 0, d, stream));
        } else {
            SPEX_TRY(spex_scatter_add_owned_rows_f32(s->g_prop_slots, pos, 2 * (int64_t)B, lo, n_loc, d, s->g_prop, 0, stream));
            SPEX_TRY(spex_scatter_add_owned_rows_f32(s->g_raw_slots, pos, 2 * (int64_t)B, lo, n_loc, d, s->g_raw, 0, stream));
        }
        // ---- backward through the propagation: G_L = g / (L + 1);  G_l = g / (L + 1) + A^T G_{l+1} on the blocks of A^T
        SPEX_TRY(spex::scale_div(s->g_prop, s->gs, (float)(L + 1), (int64_t)sz, stream));
        for (int32_t l = L - 1, k = 0; l >= 0; --l, ++k) {
            SPEX_TRY(exchange_in_place(s->comm, s->rows_per_rank, k == 0 ? s->gs : nullptr, n_loc, max_rows, d, Tb[k & 1], stream));
            float *nxt = l == 0 ? s->g_E0 : own(Tb[(k + 1) & 1]);
            if (n_loc) SPEX_TRY(spex_spmm_f32(s->graph_t, Tb[k & 1], nxt, s->gs, 1.0f, nullptr, nullptr, 1.0f, d, stream));
        }
        return SPEX_OK;
    };
    int rc = fast ? rec_fast() : rec_branch();
    if (forked && hipStreamWaitEvent((hipStream_t)stream, join_ev, 0) != hipSuccess && rc == SPEX_OK) rc = SPEX_ERR_HIP;   // joined on every path
    if (rc == SPEX_OK) rc = rc_trust;
    if (rc == SPEX_OK && T > 0 && !two_streams) rc = trust_branch(stream);
    if (rc != SPEX_OK) return rc;
    // ---- uncertainty-weighted sum of both losses + Adam (main_auto_expert_s.py:78-89) over the rank's arena: its table rows
    //      (the first n_local_users of them take the trust head's rows user_lo ..), and the replicated dense parameters.
    //      Fast path: the plain last product's g_prop / (L+1) share (L == 3: the push target P) is added here, the gate gradients'
    //      copies are summed here, and P is cleared for the next step.
    float *P = s->gathered2 ? own(s->gathered2) : nullptr;
    SPEX_TRY(spex::dual_task_adam(s->params, s->m, s->v, s->g_E0, s->g_raw, s->g_user + (size_t)s->user_lo * d, s->g_small, s->g_prop, P, s->loss,
                                  s->loss_acc, s->precision, (int64_t)sz, (int64_t)s->n_local_users * d, n_trust, B, T, s->n_rec, s->t + 1, s->lr,
                                  s->beta1, s->beta2, s->eps, (s->flags & SPEX_STEP_FIXED_TASK_WEIGHTS) != 0, stream,
                                  fast ? (L == 3 ? -1.0f : (float)(L + 1)) : 0.0f, s->grad_slots, att_copies_used, att_copies_max * 512, 0,
                                  fast && L == 3 ? 0 : 1, s->g_user, (int64_t)s->user_lo * d,
                                  s->g_user + (size_t)(s->user_lo + s->n_local_users) * d, (int64_t)(n_u - s->user_lo - s->n_local_users) * d));
    s->t += 1;
    return SPEX_OK;
}
