/* A FUNCTIONAL stand-in for librccl.so over host shared memory (test infrastructure; never shipped).
 *
 * RCCL refuses two ranks on one device, so on a one-GPU box the library's native multi-rank paths (spex_amd/csrc/comm.hip: the
 * grouped send / recv all-gather of the real rows, the equal-shard all-gather, the all-reduce, the one-call partitioned steps)
 * cannot run for world > 1.  This library exports the ten nccl* symbols comm.hip binds (SPEX_RCCL_LIB) and really moves the data —
 * between PROCESSES that may share one GPU — through a file in /dev/shm: every operation synchronises the caller's stream, stages
 * device -> host shared memory, meets the other ranks at a barrier, and copies peers' data host -> device.  Blocking, slow, and
 * deterministic (the all-reduce adds the ranks' buffers in rank order on the host).  It checks what the wire would: matching
 * counts between a send and its receive, every rank in the same collective with the same count.
 * What it does NOT show: anything about RCCL or xGMI.  Row 8e stays "unmeasured on hardware".
 *
 * Build: gcc -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include rccl_shm_stub.c -L/opt/rocm/lib -lamdhip64 -o librccl_shm_stub.so
 * Limits: <= 8 ranks, SPEX_STUB_SLOT_MB (default 48) MiB staged per rank and operation. */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
enum { kMaxRanks = 8, kMaxOps = 64, kFloat32 = 7, kSum = 0, kOk = 0, kErr = 1 };

typedef struct {
    int kind, peer;             /* 1 = send */
    long long count;            /* floats */
    long long offset;           /* into the rank's data region, in floats */
} dir_entry;

typedef struct {
    _Atomic int arrived;        /* barrier */
    _Atomic int generation;
    _Atomic int attached;
    int world;
    long long op_kind[kMaxRanks], op_count[kMaxRanks];     /* what each rank thinks the current collective is */
    int n_dir[kMaxRanks];
    dir_entry dir[kMaxRanks][kMaxOps];
} shm_header;

typedef struct stub_comm {
    int rank, world, fd;
    size_t slot_bytes, total;
    char path[129];
    shm_header *h;
    char *data;                 /* world regions of slot_bytes */
} *ncclComm_t;

typedef struct { int kind, peer; long long count; const void *send; void *recv; ncclComm_t comm; hipStream_t stream; } pending_op;
static __thread pending_op g_pending[kMaxOps];
static __thread int g_n_pending, g_depth;

static float *region(ncclComm_t c, int q) { return (float *)(c->data + (size_t)q * c->slot_bytes); }

static int barrier(ncclComm_t c)
{
    shm_header *h = c->h;
    const int gen = atomic_load(&h->generation);
    if (atomic_fetch_add(&h->arrived, 1) + 1 == c->world) {
        atomic_store(&h->arrived, 0);
        atomic_fetch_add(&h->generation, 1);
        return kOk;
    }
    struct timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (atomic_load(&h->generation) == gen) {
        clock_gettime(CLOCK_MONOTONIC, &t);
        if (t.tv_sec - t0.tv_sec > 120) { fprintf(stderr, "rccl_shm_stub: rank %d waited 120 s at a barrier\n", c->rank); return kErr; }
        usleep(50);
    }
    return kOk;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof(id->internal), "/dev/shm/spex_rccl_stub_%d_%lld", (int)getpid(), (long long)t.tv_nsec);
    return kOk;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return kErr;
    ncclComm_t c = (ncclComm_t)calloc(1, sizeof(struct stub_comm));
    c->rank = rank; c->world = nranks;
    const char *mb = getenv("SPEX_STUB_SLOT_MB");
    c->slot_bytes = (size_t)(mb ? atoi(mb) : 48) << 20;
    c->total = ((sizeof(shm_header) + 4095) & ~(size_t)4095) + c->slot_bytes * (size_t)nranks;
    memcpy(c->path, id.internal, 128);            /* (calloc: path[128] == 0) */
    c->fd = open(c->path, O_RDWR | O_CREAT, 0600);
    if (c->fd < 0 || ftruncate(c->fd, (off_t)c->total) != 0) { free(c); return kErr; }
    void *m = mmap(NULL, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
    if (m == MAP_FAILED) { free(c); return kErr; }
    c->h = (shm_header *)m;
    c->data = (char *)m + ((sizeof(shm_header) + 4095) & ~(size_t)4095);
    c->h->world = nranks;                                   /* (a fresh file is all zero: counters start at 0) */
    atomic_fetch_add(&c->h->attached, 1);
    struct timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (atomic_load(&c->h->attached) < nranks) {         /* creation is collective, like ncclCommInitRank */
        clock_gettime(CLOCK_MONOTONIC, &t);
        if (t.tv_sec - t0.tv_sec > 120) return kErr;
        usleep(100);
    }
    *out = c;
    return kOk;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return kOk;
    if (c->rank == 0) unlink(c->path);
    munmap((void *)c->h, c->total);
    close(c->fd);
    free(c);
    return kOk;
}

/* every rank announces (kind, count); all must agree */
static int agree(ncclComm_t c, long long kind, long long count)
{
    c->h->op_kind[c->rank] = kind; c->h->op_count[c->rank] = count;
    if (barrier(c)) return kErr;
    for (int q = 0; q < c->world; ++q)
        if (c->h->op_kind[q] != kind || c->h->op_count[q] != count) {
            fprintf(stderr, "rccl_shm_stub: rank %d is in collective %lld x %lld, rank %d in %lld x %lld\n", c->rank, kind, count, q, c->h->op_kind[q],
                    c->h->op_count[q]);
            return kErr;
        }
    return kOk;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t stream)
{
    if (dt != kFloat32 || count * 4 > c->slot_bytes) return kErr;
    if (hipStreamSynchronize(stream) != hipSuccess) return kErr;
    if (hipMemcpy(region(c, c->rank), send, count * 4, hipMemcpyDeviceToHost) != hipSuccess) return kErr;
    if (agree(c, 4, (long long)count)) return kErr;
    for (int q = 0; q < c->world; ++q)
        if (hipMemcpy((float *)recv + (size_t)q * count, region(c, q), count * 4, hipMemcpyHostToDevice) != hipSuccess) return kErr;
    return barrier(c);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t stream)
{
    if (dt != kFloat32 || op != kSum || count * 4 > c->slot_bytes) return kErr;
    if (hipStreamSynchronize(stream) != hipSuccess) return kErr;
    if (hipMemcpy(region(c, c->rank), send, count * 4, hipMemcpyDeviceToHost) != hipSuccess) return kErr;
    if (agree(c, 5, (long long)count)) return kErr;
    float *acc = (float *)malloc(count * 4);
    memcpy(acc, region(c, 0), count * 4);
    for (int q = 1; q < c->world; ++q) {                    /* rank order: the same bits on every rank */
        const float *r = region(c, q);
        for (size_t i = 0; i < count; ++i) acc[i] += r[i];
    }
    const int bad = hipMemcpy(recv, acc, count * 4, hipMemcpyHostToDevice) != hipSuccess;
    free(acc);
    if (bad) return kErr;
    return barrier(c);
}

static int flush_group(void)
{
    if (g_n_pending == 0) return kOk;
    ncclComm_t c = g_pending[0].comm;
    shm_header *h = c->h;
    long long off = 0;
    int nd = 0;
    for (int i = 0; i < g_n_pending; ++i) {
        pending_op *p = &g_pending[i];
        if (p->comm != c) return kErr;
        if (hipStreamSynchronize(p->stream) != hipSuccess) return kErr;
        if (p->kind != 1) continue;
        if ((size_t)(off + p->count) * 4 > c->slot_bytes || nd >= kMaxOps) return kErr;
        if (hipMemcpy(region(c, c->rank) + off, p->send, (size_t)p->count * 4, hipMemcpyDeviceToHost) != hipSuccess) return kErr;
        h->dir[c->rank][nd].kind = 1; h->dir[c->rank][nd].peer = p->peer; h->dir[c->rank][nd].count = p->count; h->dir[c->rank][nd].offset = off;
        ++nd;
        off += p->count;
    }
    h->n_dir[c->rank] = nd;
    if (agree(c, 6, 0)) return kErr;
    int taken[kMaxRanks] = {0};
    int rc = kOk;
    for (int i = 0; i < g_n_pending && rc == kOk; ++i) {
        pending_op *p = &g_pending[i];
        if (p->kind != 2) continue;
        const int q = p->peer;
        int found = -1, seen = 0;                           /* the k-th receive from q matches q's k-th send to me */
        for (int e = 0; e < h->n_dir[q]; ++e)
            if (h->dir[q][e].peer == c->rank && seen++ == taken[q]) { found = e; break; }
        if (found < 0 || h->dir[q][found].count != p->count) {
            fprintf(stderr, "rccl_shm_stub: rank %d expects %lld floats from rank %d, which sends %lld\n", c->rank, p->count, q,
                    found < 0 ? -1LL : h->dir[q][found].count);
            rc = kErr;
            break;
        }
        ++taken[q];
        if (hipMemcpy(p->recv, region(c, q) + h->dir[q][found].offset, (size_t)p->count * 4, hipMemcpyHostToDevice) != hipSuccess) rc = kErr;
    }
    /* every send must have been received: count my sends to q against q's receives — checked by q; here only the barrier */
    g_n_pending = 0;
    if (barrier(c)) return kErr;
    return rc;
}

static ncclResult_t queue(int kind, int peer, long long count, const void *send, void *recv, ncclDataType_t dt, ncclComm_t c, hipStream_t stream)
{
    if (dt != kFloat32 || g_n_pending >= kMaxOps || peer < 0 || peer >= c->world) return kErr;
    pending_op *p = &g_pending[g_n_pending++];
    p->kind = kind; p->peer = peer; p->count = count; p->send = send; p->recv = recv; p->comm = c; p->stream = stream;
    return g_depth == 0 ? flush_group() : kOk;
}

ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t stream)
{ return queue(1, peer, (long long)count, send, NULL, dt, c, stream); }
ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t stream)
{ return queue(2, peer, (long long)count, NULL, recv, dt, c, stream); }
ncclResult_t ncclGroupStart(void) { ++g_depth; return kOk; }
ncclResult_t ncclGroupEnd(void)
{
    if (g_depth > 0) --g_depth;
    return g_depth == 0 ? flush_group() : kOk;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : "rccl_shm_stub: failure (see stderr)"; }
