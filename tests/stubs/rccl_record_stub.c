/* A RECORDING stand-in for librccl.so (test infrastructure; never shipped, never linked into libspexhip.so).
 *
 * libspexhip binds RCCL at run time (spex_amd/csrc/comm.hip: dlopen / dlsym of ten nccl* symbols; SPEX_RCCL_LIB names the
 * library).  A one-GPU box cannot host two RCCL ranks, so the exchange's peer / count / offset arithmetic for world > 1 has
 * never met a wire.  This library exports the same ten symbols, moves NO data, and logs every call — operation, peer, element
 * count, send / receive pointers, stream, group depth — so a test can create a `world = 4, rank = 2` communicator and assert
 * the exact call sequence of spex_comm_allgather_rows_f32 (both forms), spex_comm_allreduce_sum_f32 and the partitioned
 * step.  What it cannot show: that RCCL itself moves the bytes (row 8e stays "unmeasured on hardware").
 *
 * Signatures follow rccl.h (ncclUniqueId is a 128-byte struct passed by value; handles and streams are pointers). */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct stub_comm { int rank, world; } *ncclComm_t;
typedef int ncclResult_t;      /* ncclSuccess == 0 */
typedef int ncclDataType_t;    /* ncclFloat32 == 7 */
typedef int ncclRedOp_t;       /* ncclSum == 0 */

enum { OP_UNIQUE_ID = 1, OP_INIT, OP_DESTROY, OP_ALLGATHER, OP_ALLREDUCE, OP_SEND, OP_RECV, OP_GROUP_START, OP_GROUP_END };

typedef struct {
    int op, peer, dtype, group_depth;     /* group_depth: depth at which the call was made (0 = outside any group) */
    long long count;
    const void *send;
    void *recv;
    void *stream;
} stub_rec;

#define STUB_CAP 65536
static stub_rec g_log[STUB_CAP];
static int g_n, g_depth, g_fail_op, g_fail_after;

static ncclResult_t rec(int op, int peer, long long count, int dtype, const void *send, void *recv, void *stream)
{
    if (g_n < STUB_CAP) {
        stub_rec *r = &g_log[g_n++];
        r->op = op; r->peer = peer; r->dtype = dtype; r->group_depth = g_depth; r->count = count;
        r->send = send; r->recv = recv; r->stream = stream;
    }
    if (g_fail_op == op && g_fail_after-- == 0) { g_fail_op = 0; return 1; /* ncclUnhandledCudaError */ }
    return 0;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0x5A, sizeof(*id)); return rec(OP_UNIQUE_ID, -1, 0, 0, 0, 0, 0); }
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    (void)id;
    *comm = (ncclComm_t)malloc(sizeof(struct stub_comm));
    (*comm)->rank = rank; (*comm)->world = nranks;
    return rec(OP_INIT, rank, nranks, 0, 0, 0, 0);
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { free(comm); return rec(OP_DESTROY, -1, 0, 0, 0, 0, 0); }
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t c, void *stream)
{ (void)c; return rec(OP_ALLGATHER, -1, (long long)count, dt, send, recv, stream); }
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, void *stream)
{ (void)c; return rec(OP_ALLREDUCE, (int)op, (long long)count, dt, send, recv, stream); }
ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, void *stream)
{ (void)c; return rec(OP_SEND, peer, (long long)count, dt, send, 0, stream); }
ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, void *stream)
{ (void)c; return rec(OP_RECV, peer, (long long)count, dt, 0, recv, stream); }
ncclResult_t ncclGroupStart(void) { ncclResult_t r = rec(OP_GROUP_START, -1, 0, 0, 0, 0, 0); ++g_depth; return r; }
ncclResult_t ncclGroupEnd(void) { if (g_depth > 0) --g_depth; return rec(OP_GROUP_END, -1, 0, 0, 0, 0, 0); }
const char *ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : "stub: injected failure"; }

/* ---- the test's side */
int stub_log_count(void) { return g_n; }
int stub_log_get(int i, stub_rec *out) { if (i < 0 || i >= g_n) return -1; *out = g_log[i]; return 0; }
void stub_log_clear(void) { g_n = 0; }
int stub_group_depth(void) { return g_depth; }
/* the (after + 1)-th call of operation `op` from now on returns an error (once) */
void stub_fail(int op, int after) { g_fail_op = op; g_fail_after = after; }
