// The search plan of the flat index: every dispatch decision of a search as a PURE function of the request shape and
// the index state (flat_plan.hip; split out of flat_index.hip in round 6).  index_search_impl executes a plan,
// prag_plan_search describes one on a host without a GPU, prag_index_last_plan returns the one that ran.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>

#include "flat_internal.h"

namespace prag {

constexpr int kSlotWordsFwd = 64;
constexpr int kSlotWords = kSlotWordsFwd;   // bound slots of the list scan per query: 2 epochs x 32 slots (KC <= 32)

constexpr int kMm8Kc = 256;       // candidates per query of the 8-bit tiled selection (see search_tiled)
constexpr int kMm8CapWg = 128;    // survivors one workgroup can hold per query and segment
constexpr int kMm8Chunk = 1024;   // queries per mm_run call
constexpr int kMm8Growth = 3;     // segment i+1 ends at 3 x the end of segment i (search_tiled)

// Segment growth of the tiled scans (ONE place: plan_search prices the schedule, search_tiled runs it): keep the
// expected survivors of a segment (~(growth-1) * KC per query) inside the per-workgroup regions (64 per query) and the
// compaction's staging buffer (4096).  int8 tiles: a segment yields ~(growth - 1) * 256 survivors per query, each an
// LDS atomic and two stores in the filter while the other wave group waits; short segments keep the bound fresh.
// Measured, 1000 queries, growth 9 / 6 / 4 / 3 / 2: 2.625 M rows 3.81 / 3.66 / 3.57 / 3.39 / 3.96 ms, 21 M rows 19.1 /
// 18.5 / 18.2 / 17.9 / 17.9, 1 M rows 2.13 / 1.95 / 1.97 / 1.90 / 1.94 (same box).
inline int mm_segment_growth(int Bpad, int chunk, int cap_wg, int cu_budget, int kc, bool i8) {
    const int n_qb = std::max(1, std::min(Bpad, chunk) / 256);
    const int g_wg = 1 + (cap_wg / 4) * cu_budget / std::max(1, kc * n_qb);
    const int g_lds = 1 + 3000 / kc;
    int g = std::max(2, std::min(16, std::min(g_wg, g_lds)));
    if (i8) g = std::min(g, kMm8Growth);
    return g;
}

struct PlanEnv {
    int d = 0, metric = PRAG_METRIC_L2, store = PRAG_F16;
    int64_t ntotal = 0;
    int B = 0, k = 0;
    int kc_min = 0, hp_mode = 1, mm_mode = 1, mm8_mode = 1, cert_mode = 1, prepass_mode = -1, wg_cap = 0, n_cu = 256;
    int shadow_mode = 1;
    int64_t mm8_min_rows = 2ll << 20;
    bool shadow_ready = false;    // the index keeps an up-to-date shadow and wants one at this size
    bool mm8_auto_off = false;
    bool allow_mm8 = true;
};

struct SearchPlan {
    int kc = 0;                   // 0: k is beyond the deepest list (PRAG_EUNSUPPORTED)
    bool exact_only = false, mm8_eligible = false, use_mm8 = false, use_mm = false, shadow128 = false, use_qs = false,
         use_shadow = false, use_hp = false, certify = true, use_slots = false, prepass = false;
    int qstride = 0, QT = 32, Bpad = 0, n_tiles = 0, cu_budget = 0, grid = 1, mm_chunk = 0, mm_cap_wg = 0, ex_grid = 1,
        ex_fcap = 1;
    size_t part_need = 0, cand_need = 0;
    const char* family = "";      // kernel of the corpus pass
    int launches = 0;             // corpus passes per search (tiled scans: segments)
    int mm_growth = 0;            // tiled scans: segment growth (first step; later steps: mm_growth_step)
    int64_t last_seg_rows = 0;    // tiled scans: rows of the last (largest) corpus segment - the profiled launch
    int64_t bytes_per_launch = 0; // algorithmic bytes one pass over the shard reads, in the form that is scanned
    size_t ws_bytes = 0;          // device workspace of the groups this plan touches
};

// candidate depth of the per-lane lists for k results (8 / 16 / 32; deeper: the tiled scan's lists; 0: unsupported)
int pick_kc(int k);
// the query-stationary kernel (65-128 queries over fp16 rows) covers this shape
bool qs_supported(int d, int store, int kc);
SearchPlan plan_search(const PlanEnv& e);
int plan_describe(const PlanEnv& e, const SearchPlan& P, char* out, int cap);

}  // namespace prag
