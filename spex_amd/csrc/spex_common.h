// Internal helpers shared by the libspexhip.so translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/spex_hip.h"

namespace spex {

void set_error(const char *fmt, ...);

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
constexpr int kLongRow = 128;      // rows with more stored entries are cut into segments
constexpr int kSegLen = 128;       // entries per long-row segment
constexpr int kTaskEntries = 16;   // entry budget of a multi-row quarter-wave task (one 16-lane metadata load)

}  // namespace spex

struct spex_timer {
    std::vector<hipEvent_t> start, stop;
    int32_t used = 0;
};

// The opaque handle.  All pointers are device memory owned by the handle.
struct spex_graph {
    int32_t n_rows = 0, n_cols = 0;
    int64_t nnz = 0;
    int32_t *rowptr = nullptr;   // [n_rows+1]
    int32_t *col = nullptr;      // [nnz]
    float *val = nullptr;        // [nnz]
    int32_t *edge_id = nullptr;  // [nnz] or null (identity)
    // long rows (> kLongRow entries): segment table + per-segment partial rows + per-row fix-up table
    int32_t n_long = 0, n_seg = 0;
    int32_t *seg_beg = nullptr;   // [n_seg] first entry of the segment
    int32_t *seg_end = nullptr;   // [n_seg]
    int32_t *long_row = nullptr;  // [n_long] row index
    int32_t *long_seg0 = nullptr; // [n_long+1] first segment of each long row
    float *partial = nullptr;     // [n_seg * d_cap] scratch, grown on demand
    int64_t partial_cap = 0;      // floats
    // quarter-wave tasks of the d == 64 kernel (16 lanes x float4 per task, 4 tasks per wave): a task is a
    // contiguous entry range that never splits a short row.  x = first entry, y = one-past-last entry,
    // z = partial-row slot (>= 0: the task is one 128-entry segment of a long row) or -1, w = row to zero-fill for an
    // empty row (x == y) or -1.  Sorted by descending 16-entry chunk count so the 4 tasks of a wave are alike.
    int32_t n_tasks = 0;          // multiple of 4
    int4 *task = nullptr;         // [n_tasks]
    int32_t *entry_row = nullptr; // [nnz] row of each stored entry (COO row index, sorted)
    // edge dropout
    int mask_mode = 0;
    const uint8_t *keep = nullptr;
    float keep_prob = 1.0f;
    uint64_t seed = 0;
    spex_timer *timer = nullptr;  // profiling hook (not owned)
};
