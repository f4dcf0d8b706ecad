// Graph handle: CSR in HBM + the long-row segment table.  See include/spex_hip.h for the contract.
#include <stdarg.h>
#include <stdio.h>

#include <stdlib.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "spex_common.h"

namespace spex {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace spex

extern "C" int spex_version(void) { return 5; }
extern "C" const char *spex_last_error(void) { return spex::g_err; }

// Host-side packing runs on a few threads: SPEX_BUILD_THREADS, default min(16, hardware threads); small inputs stay on
// the calling thread.
static int build_threads(int64_t work)
{
    if (work < ((int64_t)1 << 20)) return 1;
    int n = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("SPEX_BUILD_THREADS")) n = atoi(e);
    return n < 1 ? 1 : (n > 16 ? 16 : n);
}
template <typename F>
static void parallel_parts(int n_parts, F &&body)   // body(part) for part in [0, n_parts), one thread each
{
    if (n_parts <= 1) {
        body(0);
        return;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)n_parts - 1);
    int started = 1;
    try {
        for (; started < n_parts; ++started) th.emplace_back([&body, started]() { body(started); });
    } catch (...) {                                 // no more threads to be had: the rest of the parts run here
    }
    body(0);
    for (int p = started; p < n_parts; ++p) body(p);
    for (auto &t : th) t.join();
}

template <typename T>
static int upload(T **dst, const T *src, size_t n)
{
    *dst = nullptr;
    if (n == 0) return SPEX_OK;
    SPEX_HIP(hipMalloc((void **)dst, n * sizeof(T)));
    SPEX_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return SPEX_OK;
}

extern "C" int spex_graph_destroy(spex_graph_t *g)
{
    if (!g) return SPEX_OK;
    void *ptrs[] = {g->rowptr, g->col, g->val, g->edge_id, g->seg_beg, g->seg_end, g->long_row, g->long_seg0, g->partial,
                    g->task, g->chunk_off, g->chunk_val, g->chunk_mask, g->chunk_eid, g->chunk_row, g->hub_row, g->hub_seg0, g->row_of, g->tile_row, g->chunk_pad, g->wg_rows,
                    g->hub_grp, g->hub_fold, g->hub_ticket};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (g->scratch_ev) (void)hipEventDestroy(g->scratch_ev);
    delete g;
    return SPEX_OK;
}

static int graph_create_impl(const int32_t *h_rowptr, const int32_t *h_col, const float *h_val, const int32_t *h_edge_id, int32_t n_rows,
                             int32_t n_cols, int64_t nnz, int32_t flags, spex_graph_t **out, uint64_t *digest = nullptr,
                             std::vector<int32_t> *hub_dump = nullptr);

// Host only (no device call): the fingerprint of everything spex_graph_create would upload for this matrix — FNV-1a hashes of the
// task table, the chunk arrays, the segment / hub tables — so that a binding, or the CPU test-suite, can check that the threaded
// packer (SPEX_BUILD_THREADS) lays the matrix out exactly as the single-thread planner does.
extern "C" int spex_graph_pack_digest(const int32_t *h_rowptr, const int32_t *h_col, const float *h_val, int32_t n_rows, int32_t n_cols,
                                      int64_t nnz, uint64_t *digest)
{
    SPEX_CHECK_ARG(digest, "spex_graph_pack_digest: NULL output");
    spex_graph_t *unused = nullptr;
    return graph_create_impl(h_rowptr, h_col, h_val, nullptr, n_rows, n_cols, nnz, 0, &unused, digest);
}

// Host only: how the packer laid out the rows beyond kWgRowMax entries (the tables the in-kernel hub fold reads) — for tests of the
// invariants the kernel relies on: every hub starts a workgroup, its groups are 16 adjacent tasks (the last one shorter), the groups'
// partial rows are numbered consecutively per hub.  out: [n_positions, n_hubs, n_positions x (chunks, row, task.w, leader, waves in
// the group, partial row, hub), n_hubs x (first partial row, groups)]; *n_out = ints needed (call with cap = 0 to size).
extern "C" int spex_graph_pack_hub_table(const int32_t *h_rowptr, const int32_t *h_col, const float *h_val, int32_t n_rows, int32_t n_cols,
                                         int64_t nnz, int32_t *out, int64_t cap, int64_t *n_out)
{
    SPEX_CHECK_ARG(n_out && (out || cap == 0), "spex_graph_pack_hub_table: NULL output");
    spex_graph_t *unused = nullptr;
    uint64_t digest[8];
    std::vector<int32_t> dump;
    if (int rc = graph_create_impl(h_rowptr, h_col, h_val, nullptr, n_rows, n_cols, nnz, 0, &unused, digest, &dump)) return rc;
    *n_out = (int64_t)dump.size();
    if (cap >= (int64_t)dump.size()) std::copy(dump.begin(), dump.end(), out);
    return SPEX_OK;
}

extern "C" int spex_graph_create(const int32_t *h_rowptr, const int32_t *h_col, const float *h_val,
                                 const int32_t *h_edge_id, int32_t n_rows, int32_t n_cols, int64_t nnz,
                                 spex_graph_t **out)
{
    return graph_create_impl(h_rowptr, h_col, h_val, h_edge_id, n_rows, n_cols, nnz, 0, out);
}

template <typename T>
static uint64_t fnv1a(const std::vector<T> &v, uint64_t h = 1469598103934665603ull)
{
    const unsigned char *p = reinterpret_cast<const unsigned char *>(v.data());
    for (size_t i = 0, n = v.size() * sizeof(T); i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

static int graph_create_impl(const int32_t *h_rowptr, const int32_t *h_col, const float *h_val, const int32_t *h_edge_id, int32_t n_rows,
                             int32_t n_cols, int64_t nnz, int32_t flags, spex_graph_t **out, uint64_t *digest, std::vector<int32_t> *hub_dump)
{
    SPEX_CHECK_ARG(out, "spex_graph_create: out is NULL");
    *out = nullptr;
    SPEX_CHECK_ARG(h_rowptr && n_rows >= 0 && n_cols >= 0 && nnz >= 0, "spex_graph_create: bad sizes / NULL rowptr");
    SPEX_CHECK_ARG(nnz == 0 || (h_col && h_val), "spex_graph_create: NULL col/val with nnz > 0");
    SPEX_CHECK_ARG(nnz < (int64_t)INT32_MAX, "spex_graph_create: nnz %lld does not fit int32 entry offsets", (long long)nnz);
    SPEX_CHECK_ARG(h_rowptr[0] == 0 && h_rowptr[n_rows] == nnz, "spex_graph_create: rowptr[0] != 0 or rowptr[n_rows] != nnz");
    // Validate on the host what the kernels assume (a bad column index would be an out-of-bounds gather on the GPU).
    const int n_thr = build_threads(nnz);
    int64_t max_edge_id = nnz - 1;
    {
        std::vector<std::string> err((size_t)n_thr);
        std::vector<int64_t> part_max((size_t)n_thr, -1);
        parallel_parts(n_thr, [&](int p) {
            char buf[256];
            const int32_t r_lo = (int32_t)((int64_t)n_rows * p / n_thr), r_hi = (int32_t)((int64_t)n_rows * (p + 1) / n_thr);
            int64_t mx = -1;
            for (int32_t r = r_lo; r < r_hi; ++r) {
                const int32_t b = h_rowptr[r], e_end = h_rowptr[r + 1];
                if (b > e_end || b < 0 || (int64_t)e_end > nnz) {
                    snprintf(buf, sizeof(buf), "spex_graph_create: rowptr not monotone at row %d", r);
                    err[p] = buf;
                    return;
                }
                for (int32_t e = b; e < e_end; ++e) {
                    if (h_col[e] < 0 || h_col[e] >= n_cols) {
                        snprintf(buf, sizeof(buf), "spex_graph_create: column %d out of range at entry %d", h_col[e], e);
                        err[p] = buf;
                        return;
                    }
                    if (e != b && h_col[e - 1] >= h_col[e]) {
                        snprintf(buf, sizeof(buf), "spex_graph_create: columns not strictly ascending in row %d (coalesce first)", r);
                        err[p] = buf;
                        return;
                    }
                    if (h_edge_id) {
                        if (h_edge_id[e] < 0) {
                            snprintf(buf, sizeof(buf), "spex_graph_create: negative edge_id at entry %lld", (long long)e);
                            err[p] = buf;
                            return;
                        }
                        if (h_edge_id[e] > mx) mx = h_edge_id[e];
                    }
                }
            }
            part_max[p] = mx;
        });
        for (int p = 0; p < n_thr; ++p)      // the first failing part = the lowest row
            SPEX_CHECK_ARG(err[p].empty(), "%s", err[p].c_str());
        if (h_edge_id) {
            max_edge_id = -1;
            for (int p = 0; p < n_thr; ++p) max_edge_id = std::max(max_edge_id, part_max[p]);
        }
    }

    spex_graph *g = new spex_graph();
    g->n_rows = n_rows;
    g->n_cols = n_cols;
    g->nnz = nnz;
    g->max_edge_id = max_edge_id;

    // long rows -> segments of kSegLen entries; partial sums are combined in segment order by the fix-up kernel
    std::vector<int32_t> seg_beg, seg_end, long_row, long_seg0;
    for (int32_t r = 0; r < n_rows; ++r) {
        const int32_t b = h_rowptr[r], e = h_rowptr[r + 1];
        if (e - b > spex::kLongRow) {
            long_row.push_back(r);
            long_seg0.push_back((int32_t)seg_beg.size());
            for (int32_t s = b; s < e; s += spex::kSegLen) {
                seg_beg.push_back(s);
                seg_end.push_back(s + spex::kSegLen < e ? s + spex::kSegLen : e);
            }
        }
    }
    long_seg0.push_back((int32_t)seg_beg.size());
    g->n_long = (int32_t)long_row.size();
    g->n_seg = (int32_t)seg_beg.size();

    // Chunked task table (see spex_common.h).
    //   task.w: bits 0-1 kind (0 pack / empty row / null, 1 segment combined in the workgroup, 2 segment through global
    //           scratch), bit 2 workgroup has a barrier, bit 3 leader (first segment of its row), bits 4-7 position of
    //           the wave in its workgroup (= LDS slot), bits 8-12 number of segments of the row; kind 2: bits 4.. slot.
    std::vector<int4> task;
    std::vector<uint32_t> c_off, c_mask, c_eid;
    std::vector<uint8_t> c_pad;                     // padding entries at the end of each chunk (spex_graph_set_values)
    std::vector<int32_t> c_row;
    std::vector<float> c_val;
    std::vector<int32_t> hub_row, hub_seg0;
    std::vector<int4> hub_grp;
    std::vector<int2> hub_fold;
    const bool chunked = true;
    g->row_ids = (int64_t)n_cols * 256 <= ((int64_t)16 << 20);
    (void)flags;
    const int32_t tile_rows = 0;                 // (round 3's tile mode — at most 64 rows per workgroup, for a fused SpMM + NGCF layer launch
                                                 //  that measured no faster than two launches — is gone; the bookkeeping below stays inert)
    std::vector<int32_t> wg_rows;
    // The cache-resident launch is fastest when ALL its workgroups are resident at once (2 per CU: 512): a table that needs a few
    // more than that pays a second dispatch round for them (Epinion2's NGCF adjacency: 518 workgroups, 15.9 us against 14.9).  If
    // the first-fit packing with kOpenTasks open tasks lands just above, it is redone with 32 (denser: 5.5 % padding instead of
    // 6.7 %, slightly less local) and kept if that fits.
    constexpr int kResidentWgs = 512;
    int open_tasks_now = spex::kOpenTasks;
    // Entries a pack of SHORT rows (<= 64 entries each) may hold.  64 = four chunks = four gather round trips per wave.  A table that
    // needs a few more workgroups than are resident pays a whole second dispatch round for them (the Weibo-shaped graph: 520 workgroups,
    // 19.8 us per product instead of ~13); letting the packs hold 80 / 96 / .. entries instead makes some waves run a fifth / sixth
    // chunk (x 1.25 / 1.5 of a wave's chain) and the table fit in one round.  Every row is still ONE fmaf chain in column order, so the
    // products do not change by a bit; rows of 65 .. 1 024 entries keep their 64-entry segments (the row-list and batch kernels sum
    // them the same way).
    int pack_cap = spex::kTaskEntries;
    bool final_pass = false;
    for (int attempt = 0; chunked && attempt < 12; ++attempt) {
        task.clear();
        wg_rows.clear();
        hub_row.clear();
        hub_seg0.clear();
        hub_grp.clear();
        hub_fold.clear();
        // Planning pass (this thread): add_chunks only RECORDS a job — the rows of a pack, or the range [b, e) of row r0
        // (a segment: no end-of-row flags) — and hands out its chunk range; the entries are written afterwards by
        // fill_chunks on several threads, each job into its own range (6 s -> 1 s at 2^24 nodes).
        struct ChunkJob { int64_t first_chunk; int64_t rows_off; int32_t n_rows, b, e, r0; };
        std::vector<ChunkJob> jobs;
        std::vector<int32_t> pack_rows;
        int64_t n_planned = 0;
        jobs.reserve((size_t)nnz / 48 + 64);
        pack_rows.reserve((size_t)n_rows + 64);
        auto add_chunks = [&](const int32_t *rows, int32_t n_rows_in, int32_t b, int32_t e, int32_t r0) -> int2 {
            int64_t entries = 0;
            ChunkJob j{n_planned, (int64_t)pack_rows.size(), 0, b, e, r0};
            if (rows) {
                j.n_rows = n_rows_in;
                for (int32_t i = 0; i < n_rows_in; ++i) {
                    entries += h_rowptr[rows[i] + 1] - h_rowptr[rows[i]];
                    pack_rows.push_back(rows[i]);
                }
            } else {
                entries = e - b;
            }
            const int64_t nch = (entries + spex::kChunk - 1) / spex::kChunk;
            jobs.push_back(j);
            n_planned += nch;
            return make_int2((int32_t)j.first_chunk, (int32_t)nch);
        };
        // a list of rows WITHOUT stored entries (task kind 3): its chunk slots carry the row ids only (c_row), every slot is padding
        auto add_zero_rows = [&](const int32_t *rows, int32_t n) -> int2 {
            ChunkJob j{n_planned, (int64_t)pack_rows.size(), -n, 0, 0, rows[0]};
            for (int32_t i = 0; i < n; ++i) pack_rows.push_back(rows[i]);
            const int64_t nch = (n + spex::kChunk - 1) / spex::kChunk;
            jobs.push_back(j);
            n_planned += nch;
            return make_int2((int32_t)j.first_chunk, (int32_t)nch);
        };
        auto fill_chunks = [&]() {
            const size_t n_slots = (size_t)n_planned * spex::kChunk;
            c_off.resize(n_slots);
            c_val.resize(n_slots);
            c_eid.resize(n_slots);
            if (g->row_ids) c_row.resize(n_slots);
            c_mask.resize((size_t)n_planned);
            c_pad.resize((size_t)n_planned);
            const bool row_ids = g->row_ids;
            auto fill_job = [&](const ChunkJob &j) {
                size_t pos = (size_t)j.first_chunk * spex::kChunk, ci = (size_t)j.first_chunk;
                int32_t in_chunk = 0, last_col = 0, last_eid = 0, last_row = j.r0;
                uint32_t mask = 0;
                auto put = [&](int32_t en, int32_t r, bool last) {
                    last_col = h_col[en];
                    last_eid = h_edge_id ? h_edge_id[en] : en;
                    last_row = r;
                    c_off[pos] = (uint32_t)last_col;
                    c_val[pos] = h_val[en];
                    c_eid[pos] = (uint32_t)last_eid;
                    if (row_ids) c_row[pos] = r;
                    ++pos;
                    if (last) mask |= 1u << in_chunk;
                    if (++in_chunk == spex::kChunk) {
                        c_mask[ci] = mask;
                        c_pad[ci++] = 0;
                        mask = 0;
                        in_chunk = 0;
                    }
                };
                if (j.n_rows < 0) {          // zero-row list: row ids only, all slots padding (value 0 on column 0, never gathered)
                    const int32_t n = -j.n_rows, nch = (n + spex::kChunk - 1) / spex::kChunk;
                    for (int32_t k = 0; k < nch * spex::kChunk; ++k, ++pos) {
                        c_off[pos] = 0u;
                        c_val[pos] = 0.0f;
                        c_eid[pos] = 0u;
                        if (row_ids) c_row[pos] = pack_rows[(size_t)j.rows_off + (k < n ? k : n - 1)];
                    }
                    for (int32_t c = 0; c < nch; ++c) {
                        c_mask[ci + c] = 0u;
                        c_pad[ci + c] = (uint8_t)spex::kChunk;
                    }
                    return;
                }
                if (j.n_rows) {
                    for (int32_t i = 0; i < j.n_rows; ++i) {
                        const int32_t r = pack_rows[(size_t)j.rows_off + i];
                        for (int32_t en = h_rowptr[r]; en < h_rowptr[r + 1]; ++en) put(en, r, en + 1 == h_rowptr[r + 1]);
                    }
                } else {
                    for (int32_t en = j.b; en < j.e; ++en) put(en, j.r0, false);
                }
                if (in_chunk) {  // padding: value 0 on the task's last real source row (a line already being fetched)
                    const uint32_t n_pad = (uint32_t)(spex::kChunk - in_chunk);
                    for (; in_chunk < spex::kChunk; ++in_chunk, ++pos) {
                        c_off[pos] = (uint32_t)last_col;
                        c_val[pos] = 0.0f;
                        c_eid[pos] = (uint32_t)last_eid;
                        if (row_ids) c_row[pos] = last_row;
                    }
                    c_mask[ci] = mask;
                    c_pad[ci] = (uint8_t)n_pad;
                }
            };
            // jobs are in chunk order: split them where the chunk count splits evenly
            std::vector<size_t> cut((size_t)n_thr + 1, jobs.size());
            cut[0] = 0;
            for (int p = 1; p < n_thr; ++p) {
                const int64_t want = n_planned * p / n_thr;
                cut[p] = (size_t)(std::lower_bound(jobs.begin(), jobs.end(), want,
                                                   [](const ChunkJob &j, int64_t v) { return j.first_chunk < v; }) - jobs.begin());
            }
            parallel_parts(n_thr, [&](int p) {
                for (size_t k = cut[p]; k < cut[p + 1]; ++k) fill_job(jobs[k]);
            });
        };
        std::vector<int4> normal;                       // packs of short rows, empty rows
        std::vector<int32_t> normal_nrows;              // rows each of them completes (tile mode)
        struct Mid { int32_t row, b, e, nseg; };
        std::vector<Mid> mids;                          // rows of 65..1024 entries
        std::vector<int4> hubs;                         // 128-entry segments of rows > 1024 entries
        {
            // Short rows (<= 64 entries) are packed first-fit into a few open tasks, in row order: a row goes into the
            // first open task with room, a new task evicts the oldest open one.  Tasks fill to ~60 of 64 entries
            // instead of ~48 with plain next-fit (fewer waves, fewer padding gathers) while a task's rows stay within a
            // short span of the matrix (packing across the whole matrix scatters the epilogue traffic: -6 % on the HBM
            // graph).
            // That is for graphs whose source table sits in cache (<= 16 MiB), where the kernel is bound by wave count
            // and issue slots.  A graph that streams from HBM keeps plain next-fit over ADJACENT rows instead (one open
            // task): its kernel then needs no per-entry row ids (4 B/entry less metadata) and writes whole runs of
            // neighbouring rows — worth 6 % there.
            std::vector<int32_t> empty_rows;
            struct OpenTask { std::vector<int32_t> rows; int room; };
            std::vector<OpenTask> open_tasks;
            const int max_open = g->row_ids ? open_tasks_now : 1;
            auto close_task = [&](size_t k) {
                OpenTask &ot = open_tasks[k];
                const int2 c = add_chunks(ot.rows.data(), (int32_t)ot.rows.size(), 0, 0, ot.rows[0]);
                normal.push_back(make_int4(c.x, c.y, ot.rows[0], 0));
                normal_nrows.push_back((int32_t)ot.rows.size());
                open_tasks.erase(open_tasks.begin() + k);
            };
            auto flush_window = [&]() {
                while (!open_tasks.empty()) close_task(0);
            };
            auto place_row = [&](int32_t r, int32_t deg) {
                if (!g->row_ids && !open_tasks.empty() && open_tasks[0].rows.back() != r - 1) close_task(0);  // adjacency
                for (size_t k = 0; k < open_tasks.size(); ++k) {
                    if (open_tasks[k].room >= deg) {
                        open_tasks[k].rows.push_back(r);
                        open_tasks[k].room -= deg;
                        if (open_tasks[k].room == 0) close_task(k);
                        return;
                    }
                }
                if ((int)open_tasks.size() >= max_open) close_task(0);
                open_tasks.push_back({{r}, pack_cap - deg});
                if (open_tasks.back().room == 0) close_task(open_tasks.size() - 1);
            };
            size_t long_i = 0;
            for (int32_t r = 0; r < n_rows; ++r) {
                const int32_t b = h_rowptr[r], e = h_rowptr[r + 1], deg = e - b;
                while (long_i < long_row.size() && long_row[long_i] < r) ++long_i;
                if (deg == 0) {
                    // a row without stored entries.  Bin-packed tables (row ids per entry) collect them into lists of up to 64 rows
                    // per task (below): one wave per EMPTY row made a graph with a long tail of never-seen items — 2 441 of the
                    // Weibo-shaped graph's 26 813 rows — need 670 workgroups where 512 are resident (two rounds: 21 us per product
                    // instead of 12).  Tile mode and adjacent-row tables keep the one-row zero-fill task.
                    if (g->row_ids && !tile_rows) {
                        empty_rows.push_back(r);
                    } else {
                        normal.push_back(make_int4(0, 0, r, 0));   // zero-fill task
                        normal_nrows.push_back(1);
                    }
                } else if (deg > spex::kWgRowMax) {            // hub: its kLongRow-table segments go through global scratch
                    hub_row.push_back(r);
                    hub_seg0.push_back(long_seg0[long_i]);      // [begin, end) in the kLongRow segment table
                    hub_seg0.push_back(long_seg0[long_i + 1]);
                    for (int32_t sgi = long_seg0[long_i]; sgi < long_seg0[long_i + 1]; ++sgi) {
                        const int2 c = add_chunks(nullptr, 0, seg_beg[sgi], seg_end[sgi], r);
                        hubs.push_back(make_int4(c.x, c.y, r, 2 | (sgi << 4)));
                    }
                } else if (deg > spex::kTaskEntries) {
                    mids.push_back({r, b, e, (deg + spex::kTaskEntries - 1) / spex::kTaskEntries});
                } else {
                    place_row(r, deg);
                }
            }
            flush_window();
            for (size_t k = 0; k < empty_rows.size(); k += (size_t)spex::kTaskEntries) {     // the lightest tasks: last
                const int32_t n = (int32_t)std::min((size_t)spex::kTaskEntries, empty_rows.size() - k);
                const int2 c = add_zero_rows(empty_rows.data() + k, n);
                normal.push_back(make_int4(c.x, c.y, empty_rows[k], 3 | (n << 4)));
                normal_nrows.push_back(n);
            }
        }
        // assemble 16-wave workgroups: hub segments, then rows combined in-workgroup (heaviest first, first fit, the
        // rest of such a workgroup filled with ordinary tasks), then the ordinary tasks
        const int W = spex::kWgWaves;
        size_t next_normal = 0;
        auto null_task = make_int4(0, 0, -1, 0);
        auto fill_wg = [&](bool barrier) {  // complete the current workgroup with ordinary / null tasks
            while (task.size() % W) {
                int4 t = next_normal < normal.size() ? normal[next_normal++] : null_task;
                if (barrier) t.w |= 4;
                task.push_back(t);
            }
        };
        {
        // Hub segments lead the table.  Every hub STARTS a workgroup, so that its segments fall into groups of 16 counted from its own
        // first segment (the last group topped up with ordinary tasks): the association of a hub row's sum — segments in order within
        // a group, groups in order — then depends on the row alone, not on where the packer put it (a row block of the partitioned
        // graph sums its hubs exactly like the whole graph does: tests/test_gpu_dist.py holds the two bit for bit).  spex_common.h:
        // hub_grp (indexed by task position; ordinary tasks in between hold zeros) / hub_fold.  Hub workgroups carry the barrier bit.
        {
            int32_t hub_id = -1, prev_row = -1, n_groups = 0;
            for (size_t j = 0; j < hubs.size(); ++j) {
                int4 h = hubs[j];
                h.w |= 4;
                const bool new_hub = h.z != prev_row;
                if (new_hub) {
                    fill_wg(true);                             // complete the previous hub's last workgroup
                    ++hub_id;
                    prev_row = h.z;
                    hub_fold.push_back(make_int2(n_groups, 0));
                }
                hub_grp.resize(task.size(), make_int4(0, 0, 0, 0));
                if (task.size() % (size_t)W == 0) {           // a group starts: count its waves
                    size_t k = j;
                    while (k < hubs.size() && hubs[k].z == h.z && k - j < (size_t)W) ++k;
                    hub_grp.push_back(make_int4(1, (int32_t)(k - j), n_groups, hub_id));
                    hub_fold[(size_t)hub_id].y += 1;
                    ++n_groups;
                } else {
                    const int4 lead = hub_grp[task.size() - task.size() % (size_t)W];
                    hub_grp.push_back(make_int4(0, lead.y, lead.z, hub_id));
                }
                task.push_back(h);
            }
        }
        fill_wg(!hubs.empty());
        {
            // bucket the rows by segment count; fill each workgroup greedily with the largest row that still fits
            std::vector<std::vector<int32_t>> by_nseg(W + 1);
            for (size_t m = 0; m < mids.size(); ++m) by_nseg[mids[m].nseg].push_back((int32_t)m);
            std::vector<size_t> head(W + 1, 0);
            size_t n_placed = 0;
            while (n_placed < mids.size()) {
                int used = 0;
                for (;;) {
                    int s = W - used;
                    while (s >= 2 && head[s] >= by_nseg[s].size()) --s;
                    if (s < 2) break;
                    const Mid &md = mids[by_nseg[s][head[s]++]];
                    for (int32_t sgi = 0; sgi < md.nseg; ++sgi) {
                        const int32_t sb = md.b + sgi * spex::kTaskEntries;
                        const int32_t se = sb + spex::kTaskEntries < md.e ? sb + spex::kTaskEntries : md.e;
                        const int2 c = add_chunks(nullptr, 0, sb, se, md.row);
                        task.push_back(make_int4(c.x, c.y, md.row,
                                                 1 | 4 | (sgi == 0 ? 8 : 0) | ((used + sgi) << 4) | (md.nseg << 8)));
                    }
                    used += md.nseg;
                    ++n_placed;
                }
                fill_wg(true);
            }
        }
        while (next_normal < normal.size()) task.push_back(normal[next_normal++]);
        fill_wg(false);
            {
                const int n_wgs_now = (int)((task.size() + W - 1) / W);
                const bool pinned = false;
                if (g->row_ids && !pinned && !final_pass && n_wgs_now > kResidentWgs) {
                    if (attempt == 0 && n_wgs_now <= kResidentWgs + kResidentWgs / 16) {       // a denser first fit may do
                        open_tasks_now = 32;
                        continue;
                    }
                    if (n_wgs_now <= 2 * kResidentWgs && pack_cap < 2 * spex::kTaskEntries) {   // larger packs of short rows (see pack_cap)
                        open_tasks_now = 32;
                        pack_cap += spex::kChunk;
                        continue;
                    }
                    open_tasks_now = spex::kOpenTasks;                                          // no use: back to the default
                    pack_cap = spex::kTaskEntries;
                    final_pass = true;
                    continue;
                }
            }
        fill_chunks();
        }
        break;
    }
    g->n_tasks = (int32_t)task.size();
    g->tile_rows = tile_rows;
    g->n_wgs = (int32_t)wg_rows.size();
    if (getenv("SPEX_DEBUG_PACK"))
        fprintf(stderr, "[spex] graph %d x %d nnz %lld: %d tasks = %d workgroups, %lld chunks (%.1f %% padding), tile mode %d (%d workgroups)\n",
                n_rows, n_cols, (long long)nnz, (int)task.size(), (int)task.size() / spex::kWgWaves, (long long)c_mask.size(),
                nnz ? 100.0 * ((double)c_mask.size() * spex::kChunk - (double)nnz) / (double)nnz : 0.0, tile_rows, g->n_wgs);
    g->n_chunks = (int64_t)c_mask.size();
    g->n_hub = (int32_t)hub_row.size();
    g->n_hub_tasks = (int32_t)hub_grp.size();

    // rows starting in each tile of kSoftmaxTile consecutive entries (row-softmax kernels, edge.hip)
    g->tile = (nnz / spex::kSoftmaxTile < 2048) ? 512 : spex::kSoftmaxTile;
    g->n_tiles = (int32_t)((nnz + g->tile - 1) / g->tile);
    std::vector<int32_t> tile_row((size_t)g->n_tiles + 1);
    {
        int32_t r = 0;
        for (int32_t t = 0; t <= g->n_tiles; ++t) {
            const int64_t first = (int64_t)t * g->tile;
            while (r < n_rows && (int64_t)h_rowptr[r] < first) ++r;
            tile_row[t] = r;
        }
    }

    if (digest) {       // spex_graph_pack_digest: fingerprints of the packed arrays, nothing goes to the device
        if (hub_dump) {   // spex_graph_pack_hub_table: the head of the task table (every position the hub groups cover) + the fold table
            hub_dump->clear();
            hub_dump->push_back((int32_t)hub_grp.size());
            hub_dump->push_back((int32_t)hub_fold.size());
            for (size_t k = 0; k < hub_grp.size(); ++k) {
                const int4 t = task[k], q = hub_grp[k];
                for (int32_t v : {t.y, t.z, t.w, q.x, q.y, q.z, q.w}) hub_dump->push_back(v);
            }
            for (const int2 &f : hub_fold) {
                hub_dump->push_back(f.x);
                hub_dump->push_back(f.y);
            }
        }
        digest[0] = fnv1a(task);
        digest[1] = fnv1a(c_off);
        digest[2] = fnv1a(c_val);
        digest[3] = fnv1a(c_mask, fnv1a(c_pad));
        digest[4] = fnv1a(c_eid);
        digest[5] = fnv1a(c_row, fnv1a(wg_rows));
        digest[6] = fnv1a(seg_beg, fnv1a(seg_end, fnv1a(long_row, fnv1a(long_seg0))));
        digest[7] = fnv1a(hub_row, fnv1a(hub_seg0, fnv1a(tile_row, fnv1a(hub_grp, fnv1a(hub_fold))))) ^ (uint64_t)task.size() ^ ((uint64_t)c_mask.size() << 32);
        delete g;
        return SPEX_OK;
    }
    int rc = SPEX_OK;
    if ((rc = upload(&g->tile_row, tile_row.data(), tile_row.size())) || (rc = upload(&g->rowptr, h_rowptr, (size_t)n_rows + 1)) || (rc = upload(&g->col, h_col, (size_t)nnz)) ||
        (rc = upload(&g->val, h_val, (size_t)nnz)) ||
        (h_edge_id && (rc = upload(&g->edge_id, h_edge_id, (size_t)nnz))) ||
        (rc = upload(&g->seg_beg, seg_beg.data(), seg_beg.size())) ||
        (rc = upload(&g->seg_end, seg_end.data(), seg_end.size())) ||
        (rc = upload(&g->long_row, long_row.data(), long_row.size())) ||
        (rc = upload(&g->long_seg0, long_seg0.data(), long_seg0.size())) ||
        (rc = upload(&g->task, task.data(), task.size())) ||
        (rc = upload(&g->chunk_off, c_off.data(), c_off.size())) ||
        (rc = upload(&g->chunk_val, c_val.data(), c_val.size())) ||
        (rc = upload(&g->chunk_mask, c_mask.data(), c_mask.size())) ||
        (rc = upload(&g->chunk_pad, c_pad.data(), c_pad.size())) ||
        (rc = upload(&g->chunk_eid, c_eid.data(), c_eid.size())) ||
        (rc = upload(&g->chunk_row, c_row.data(), c_row.size())) ||
        (rc = upload(&g->wg_rows, wg_rows.data(), wg_rows.size())) ||
        (rc = upload(&g->hub_row, hub_row.data(), hub_row.size())) ||
        (rc = upload(&g->hub_seg0, hub_seg0.data(), hub_seg0.size())) ||
        (rc = upload(&g->hub_grp, hub_grp.data(), hub_grp.size())) ||
        (rc = upload(&g->hub_fold, hub_fold.data(), hub_fold.size()))) {
        spex_graph_destroy(g);
        return rc;
    }
    if (!hub_fold.empty()) {
        hipError_t e = hipMalloc((void **)&g->hub_ticket, hub_fold.size() * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemset(g->hub_ticket, 0, hub_fold.size() * sizeof(unsigned long long));
        if (e != hipSuccess) {
            spex::set_error("hipMalloc(hub_ticket) failed: %s", hipGetErrorString(e));
            spex_graph_destroy(g);
            return SPEX_ERR_HIP;
        }
    }
    if (g->n_seg > 0) {
        g->partial_cap = (int64_t)g->n_seg * 64;
        hipError_t e = hipMalloc((void **)&g->partial, (size_t)g->partial_cap * sizeof(float));
        if (e != hipSuccess) {
            spex::set_error("hipMalloc(partial) failed: %s", hipGetErrorString(e));
            spex_graph_destroy(g);
            return SPEX_ERR_HIP;
        }
    }
    *out = g;
    return SPEX_OK;
}

extern "C" int spex_graph_info(const spex_graph_t *g, int32_t *n_rows, int32_t *n_cols, int64_t *nnz,
                               int32_t *n_long_rows, int32_t *n_segments)
{
    SPEX_CHECK_ARG(g, "spex_graph_info: NULL handle");
    if (n_rows) *n_rows = g->n_rows;
    if (n_cols) *n_cols = g->n_cols;
    if (nnz) *nnz = g->nnz;
    if (n_long_rows) *n_long_rows = g->n_long;
    if (n_segments) *n_segments = g->n_seg;
    return SPEX_OK;
}

extern "C" int spex_graph_set_edge_mask(spex_graph_t *g, int mode, const uint8_t *d_keep, float keep_prob,
                                        uint64_t seed)
{
    SPEX_CHECK_ARG(g, "spex_graph_set_edge_mask: NULL handle");
    SPEX_CHECK_ARG(mode >= 0 && mode <= 2, "spex_graph_set_edge_mask: mode %d not in {0,1,2}", mode);
    SPEX_CHECK_ARG(mode != 1 || d_keep, "spex_graph_set_edge_mask: injected mode needs a device mask");
    SPEX_CHECK_ARG(mode == 0 || (keep_prob > 0.0f), "spex_graph_set_edge_mask: keep_prob must be > 0");
    if (mode != 0 && keep_prob >= 1.0f && mode == 2) mode = 0;
    g->mask_mode = mode;
    g->keep = (mode == 1) ? d_keep : nullptr;
    g->keep_prob = keep_prob;
    g->seed = seed;
    return SPEX_OK;
}

// ------------------------------------------------------------------------------------------------ profiling hook
extern "C" int spex_timer_create(int32_t capacity, int32_t every, spex_timer_t **out)
{
    SPEX_CHECK_ARG(out && capacity > 0 && capacity <= (1 << 20), "spex_timer_create: bad capacity %d", capacity);
    SPEX_CHECK_ARG(every >= 1, "spex_timer_create: every = %d", every);
    spex_timer *t = new spex_timer();
    t->every = every;
    t->start.resize(capacity);
    t->stop.resize(capacity);
    t->launches.assign(capacity, 0);
    for (int32_t i = 0; i < capacity; ++i) {
        SPEX_HIP(hipEventCreate(&t->start[i]));
        SPEX_HIP(hipEventCreate(&t->stop[i]));
    }
    *out = t;
    return SPEX_OK;
}

extern "C" int spex_timer_destroy(spex_timer_t *t)
{
    if (!t) return SPEX_OK;
    for (size_t i = 0; i < t->start.size(); ++i) {
        (void)hipEventDestroy(t->start[i]);
        (void)hipEventDestroy(t->stop[i]);
    }
    delete t;
    return SPEX_OK;
}

extern "C" int spex_timer_attach(spex_graph_t *g, spex_timer_t *t)
{
    SPEX_CHECK_ARG(g, "spex_timer_attach: NULL graph");
    g->timer = t;
    return SPEX_OK;
}

extern "C" int spex_timer_read(spex_timer_t *t, float *h_ms, int32_t *h_launches, int32_t max_count, int32_t *count,
                               int reset)
{
    SPEX_CHECK_ARG(t && count, "spex_timer_read: NULL argument");
    int32_t n = t->used < max_count ? t->used : max_count;
    for (int32_t i = 0; i < n; ++i) {
        SPEX_HIP(hipEventSynchronize(t->stop[i]));
        SPEX_HIP(hipEventElapsedTime(&h_ms[i], t->start[i], t->stop[i]));
        if (h_launches) h_launches[i] = t->launches[i];
    }
    *count = n;
    if (reset) {
        t->used = 0;
        t->seen = 0;
    }
    return SPEX_OK;
}
