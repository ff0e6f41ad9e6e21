// The flat index handle (struct prag_index) and the few helpers its translation units share: flat_index.hip (add, plan
// execution, the ABI of the single-GPU search) and flat_sharded.hip (the exchange step: RCCL all-gather + merge).
// Split out of flat_index.hip in round 6.  Reference object being replaced: `faiss.IndexFlatL2(768)`
// (make_indexer.py:449-450; exp_rag.py:248, 432-436).
#pragma once

#include <hip/hip_runtime.h>

#include <initializer_list>
#include <string>

#include "flat_internal.h"
#include "flat_plan.h"
#include "prag.h"
#include "tail_gate.h"

using namespace prag;

struct prag_index {
    int d, metric, store;
    int64_t ntotal = 0, cap = 0;
    void* rows = nullptr;
    float* xnorm = nullptr;
    // search workspace
    float* q32 = nullptr;
    _Float16* q16 = nullptr;
    _Float16* q16lo = nullptr;
    int q_cap = 0;
    int hp_mode = 1;   // high-precision selection for <= 32 queries (0 = off)
    float* part_key = nullptr;
    int* part_idx = nullptr;
    size_t part_cap = 0;  // entries
    int* cand = nullptr;
    size_t cand_cap = 0;
    uint32_t* g_tau = nullptr;  // [q_cap]
    uint32_t* g_slot = nullptr; // [q_cap][kSlotWords]
    int prepass_mode = -1;      // PRAG_PREPASS: 1 = always the pre-pass launches, 0 = always the bound slots, unset = by shard size
    // MFMA-tiled scan (> 128 queries): per-query candidate buffers
    uint32_t* mm_cnt = nullptr;
    uint32_t* mm_ovf = nullptr;
    float* mm_ckey = nullptr;
    int* mm_cidx = nullptr;
    uint32_t* mm_wcnt = nullptr;
    float* mm_wkey = nullptr;
    int* mm_widx = nullptr;
    int mm_q_cap = 0;             // queries the per-query arrays hold
    size_t mm_w_entries = 0;      // entries of mm_wkey / mm_widx
    size_t mm_c_entries = 0;      // entries of mm_ckey / mm_cidx, mm_wcnt words
    int mm_mode = 1;   // 0 = never take the MFMA-tiled path (PRAG_SCAN_MM=0)
    int shadow_bound_mode = -1;   // -1 auto, 0 off, 1 on (PRAG_SHADOW_BOUND)
    int shadow_sample_mode = -1;  // -1 auto, 0 off, 1 on (PRAG_SHADOW_SAMPLE): the sampled pre-bound of the two-level search
    int scan_gate_mode = -1;      // prag_search_and_gate: the gate's workgroups in the SCAN's launch: -1 auto, 0 never, 1 always (PRAG_SCAN_GATE)
    // Workgroups of the two-level scan, <= 64 queries: 7/8 of the CUs or all of them - measured on this index's own
    // searches (shadow_scan_wg_cap in flat_internal.h says why it cannot be a constant).  Eight searches alternate the
    // two with timing events around the scan launch (never waited for: a sample is read when the NEXT search finds its
    // event done; nothing while a stream is capturing), the faster minimum wins (all CUs only if >= 2.5 % faster, 5 % when the call carries a gate); a shard
    // that grows or shrinks by 1/8 measures again.  PRAG_SCAN_WG_TUNE=0 / 1: always 7/8 / always every CU.
    struct WgTune {
        int phase = 0;                    // samples taken (8 = decided)
        float best[2] = {1e30f, 1e30f};   // fastest scan launch seen on [0] 7/8 of the CUs, [1] every CU (ms)
        int choice = 0;
        bool pending = false;
        int pending_arm = 0;
        int64_t rows = -1;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
    };
    WgTune wg_tune[2];            // [0] <= 32 queries (two-term tiles), [1] 33-64 queries
    int wg_tune_mode = -1;
    // prag_index_set_adaptive / PRAG_ADAPTIVE at creation.  0 = the deterministic plan: nothing a search launches depends
    // on how earlier searches on the handle went or how long they took - the scan grid stays at its default (7/8 of the
    // CUs for an HBM-bound two-level scan; PRAG_SCAN_WG_TUNE=1 / prag_index_set_scan_workgroups still pin another one),
    // the retry tier is never armed by history (host-io searches still use it on their OWN flag count, which they
    // hold; device-io searches send flagged queries straight to the float64 scan), the sliced gather is always
    // enqueued, the grouped float64 scan is chosen by shape only, the int8 tiles are never switched off.  Every rank
    // and every run of one input then issues the same launches.  Results are the definition's either way.
    int adaptive = 1;
    int64_t scan8_quad_rows = (int64_t)8 << 20;   // PRAG_SCAN8_QUAD_ROWS (tests: 0 = the quad-test scans at every size)
    int mm_shape16 = 1;   // tiled scans on 16 x 16 MFMA tiles (PRAG_MM_SHAPE=32: the 32 x 32 form, A/B timing)
    // 8-bit selection for the MFMA-tiled scan (> 128 queries on an index that keeps a shadow): int8 MFMA over the
    // shadow, kMm8Kc candidates per query, the shadow's error bound in the certificate; the queries that fail that
    // certificate go through a second tier (mm8_second_tier; PRAG_MM8=0: fp16 tiles only)
    int mm8_mode = 1;
    // ... on shards of at least this many rows (shadow mode 2 = "any size": no minimum).  Measured, 1000 queries x
    // 768, int8 tiles against fp16 tiles: 1 M rows 2.04 ms / 1.73 - the 256-deep lists cost sort compactions and
    // 0.14 ms of rerank, and with the bound of a segment coming from <= 166 k rows one score in 650 survives the
    // filter, so the int8 scan itself gains only 7 % -; 2.625 M rows (an 8-GPU shard of 21 M) 3.59 / 3.82;
    // 4 Mi rows 5.27 / 5.92 (before the balanced gather of the compaction); 8 M 8.60 / 11.09; 21 M 19.4 / 28.7
    // (0.76 ns per row and 1000 queries in the last segment against 1.34)
    // (those with segment growth 9; with growth 3: 1 M rows 1.90 / 1.73, 2.625 M 3.36 / 3.90, 21 M 17.8)
    int64_t mm8_min_rows = 2ll << 20;
    float* mm_kq = nullptr;            // [mm_q_cap] key scale of every query
    // The tier decision is taken on the device (mm8_second_tier): every kernel of the second tier is enqueued with a
    // Gate on t2_word[0] = queries that failed the 8-bit certificate.  The count travels to the host asynchronously
    // (pinned word + event) for prag_index_last_tiled8 and the auto-off heuristic: no search waits for it.
    Gate gate;                            // gate of the search being enqueued (second-tier inner searches), else open
    uint32_t* t2_word = nullptr;          // device: [0] failed count of the last 8-bit tiled search
    uint32_t* tier_word_host = nullptr;   // pinned copy
    hipEvent_t tier_event = nullptr;
    bool tier_pending = false;            // a copy is in flight (tier_event)
    int tier_hi_sub = 0;                  // largest failed count the compact-batch tier takes in that search
    int mm8_last_failed = -1;             // queries of the last 8-bit tiled search that failed its certificate
                                          // (-1: it did not take the 8-bit tiles; -2: skipped, see mm8_auto_off; -3: unknown - captured)
    // few failed queries: searched again as a compact batch (mm8_second_tier)
    int* t2_list = nullptr;
    float* t2_q = nullptr;
    float* t2_D = nullptr;
    int64_t* t2_I = nullptr;
    int t2_cap = 0, t2_k = 0;
    // two searches in a row whose whole batch had to be repeated on the fp16 tiles (a corpus the 8-bit bound cannot
    // separate: a few rows of huge norm, look-alikes everywhere): the int8 tiles are skipped from then on, until rows
    // are added or prag_index_set_shadow is called
    int mm8_whole_batch_streak = 0;
    bool mm8_auto_off = false;
    // ... but not for good: after mm8_off_period eligible searches the tiles get ONE probe (a single whole-batch
    // repeat switches them off again and doubles the period, up to 4096 searches)
    int mm8_off_count = 0, mm8_off_period = 64;
    // fp32 indexes: fp16 copy of the rows for the tiled scan's candidate selection (built on the
    // first search with > 128 queries, dropped by add; the rerank always reads the fp32 rows)
    _Float16* rows16 = nullptr;
    int64_t rows16_n = -1, rows16_cap = 0;
    // host-io staging
    // one device block [I int64 | D float | flag count] and its pinned host mirror, plus a pinned/device
    // pair for the queries: a host-io search is one H2D and one D2H transfer
    float* io_q = nullptr;
    char* io_res = nullptr;
    float* io_q_host = nullptr;
    char* io_res_host = nullptr;
    int io_B = 0, io_k = 0;
    int n_cu = 256;
    int wg_cap = 0;  // 0 = use every CU
    int kc_min = 0;  // 0 = default candidate depth for k
    // exactness certificate + exact fallback (flat_internal.h)
    float* qinfo = nullptr;       // [q_cap][4]
    double* qn2 = nullptr;        // [q_cap]
    int* flag_list = nullptr;     // [q_cap]
    uint32_t* cert_words = nullptr;  // [0] = n_flag of the last search, [1] = bits of max ||x||^2
    int64_t xn_max_rows = 0;      // rows already folded into cert_words[1]
    unsigned long long* ex_key = nullptr;
    int* ex_id = nullptr;
    size_t ex_entries = 0;
    unsigned long long* ex_pool = nullptr;   // [f_cap][grid] exact_mfma_kernel: the workgroups' best keys (all-ones between searches)
    size_t ex_pool_entries = 0;
    uint32_t* ex_done = nullptr;  // [ex_done_cap] arrival counters of the exact scan's list merge
    int ex_done_cap = 0;
    // 8-bit shadow (flat_shadow.hip): 0 off, 1 (default) on for shards >= kShadowMinRows when the device has
    // room for it, 2 on at any size
    int shadow_mode = 1;
    bool shadow_failed = false;    // the shadow could not be extended at the end of an add (rows are committed; they are
                                   // scanned directly until prag_index_set_shadow is called again); the message stays
                                   // in prag_last_error
    bool shadow_no_room = false;   // mode 1: the allocation did not fit next to the rows; rows are scanned directly
    signed char* rows8 = nullptr;
    float* sscale = nullptr;
    float* serr = nullptr;
    int64_t shadow_cap = 0, shadow_rows = 0;
    uint32_t* shadow_err_max = nullptr;
    // the shadow's affine map y = (x - mu) / c (flat_shadow.hip): [mu | c | 1/c] d floats each, a [2][d] float64
    // scratch for the column sums behind it, the word max ||y||^2; fitted whenever the shadow is (re)built from row 0
    float* sh_aff = nullptr;
    double* sh_aff_sums = nullptr;
    uint32_t* sh_yn_max = nullptr;
    float* sbias = nullptr;            // [shadow_cap] per-row additive part of the two-level scan's key
    uint32_t* sh_bias_max = nullptr;   // float bits of max |sbias_i|
    // PRAG_SHADOW_AFFINE at creation: 0 identity map (the round 2-4 shadow), 1 (default) rows and queries centred on the
    // column means, c = 1; 2 centred + power-of-two column scales c_j ~ the column's standard deviation.  Measured on
    // embedding-shaped rows, 64 queries x 1 M rows, survivors per query mean / max (profiles/r05c_*, r05d_*):
    // mode 0: 180 000 - 870 000 (every query in the exact scan); mode 2: 1 450 / 2 700 (cosine), 11 200 / 113 000 (L2,
    // one region overflow); mode 1: 2 400 / 3 800 and 4 100 / 10 100.  Why scales lose with ONE int8 query term: queries
    // live in the rows' space, so c_j = sigma_j evens out the rows' grid and squares the disparity on the query's
    // (p_j = q'_j c_j ~ sigma_j^2); with both sides on one grid each, eps ~ ||p|| max|y| + max|p| ||y|| is symmetric
    // under c <-> 1/c and c = 1 is its minimum.  (The 32-query tiles - two query terms - would prefer mode 2 by ~25 %.)
    int shadow_affine_mode = 1;
    double* sh_kshift = nullptr;       // [sh_q_cap] K_q = alpha q.mu of the queries of the running search
    signed char* sh_q8 = nullptr;      // [2][q_cap][d]
    void* sh_sq = nullptr;
    uint32_t* sh_slots = nullptr;
    uint32_t* sh_ovf = nullptr;
    int sh_q_cap = 0;
    int* sh_cand = nullptr;
    int sh_cand_qt = 0;          // query-tile height the candidate store is sized for (64 or 128)
    uint32_t* sh_ccnt = nullptr;
    unsigned long long* sh_pkey = nullptr;
    int* sh_pid = nullptr;
    size_t sh_part_entries = 0;
    int cert_mode = 1;   // 0 = certificate off (PRAG_CERT=0: timing experiments only)
    // row-sharded search in C (prag_index_set_comm / prag_index_search_sharded): the caller's RCCL communicator
    // (borrowed), this rank and the world size; packed exchange buffers [D float32 [B,k] | I int64 [B,k]]
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    char* xch_send = nullptr;
    char* xch_recv = nullptr;
    size_t xch_send_cap = 0, xch_recv_cap = 0;
    EventRing prof_xch;          // HIP events around the all-gather of prag_index_search_sharded (prag_index_profile)
    // retry tier (retry_tier below): armed by what recent searches flagged - the count travels to the host
    // asynchronously and is looked at when the NEXT search is planned, never waited for
    uint32_t* r2_word = nullptr;          // device [4]
    int* r2_list = nullptr;               // [128]
    float* r2_q = nullptr;                // [32][d]
    float* r2_D = nullptr;
    int64_t* r2_I = nullptr;
    int r2_k = 0;
    uint32_t* r2_word_host = nullptr;     // pinned
    hipEvent_t r2_event = nullptr;
    bool r2_pending = false;
    bool retry_armed = false;
    int retry_clean = 0;                  // armed searches in a row that flagged nothing
    int retry_mode = -1;                  // PRAG_RETRY_TIER at creation: -1 adaptive, 0 never, 1 always armed
    int exact_group_mode = -1;            // PRAG_EXACT_GROUP at creation: -1 adaptive, 0 never, 1 always (exact_group_kernel)
    bool exact_group_hint = false;        // recent retry tiers left >= 4 queries for the exact scan
    int exact_mfma_mode = 1;              // PRAG_EXACT_MFMA at creation: 0 = never the float64-MFMA form of the grouped scan
    // the sliced gather behind the exact-bound kernel: the bound kernel finishes every query itself when <= 256 rows stay
    // under its bound (40-100 in practice), and the gather launch is then ~9 us of nothing on the critical path.  It is
    // enqueued while "armed": from the start, and again for 64 searches whenever a search left a query unfinished (that
    // query went through the flag list: retry tier / exact scan); 16 clean searches in a row disarm it.  PRAG_GATHER=1
    // keeps it always (the round 2-4 launch sequence).
    bool gather_armed = true;
    int gather_clean = 0;
    int gather_mode = -1;                 // PRAG_GATHER at creation: -1 adaptive, 1 always
    bool r2_has_unfinished = false;       // the pending statistics record carries an `unfinished` count
    uint32_t* sh_unfin = nullptr;         // device word
    // prag_index_stream_wait_scan: an event recorded right behind the corpus scan of every search (two-level search:
    // after scan8, before the bound kernel / gather / fallback probes), so that independent work of the caller - the
    // gate of the next batch - can start beside the search's low-occupancy tail on another stream
    hipEvent_t scan_done_ev = nullptr;
    bool scan_done_recorded = false;
    TailGate* tail = nullptr;   // prag_search_and_gate: the gate launch the running search may carry beside its bound kernel
    int last_flagged = -1;   // flag count of the last host-io search (-1: last search was device-io)
    std::string last_plan;   // plan_describe of the most recent search (prag_index_last_plan)
    EventRing prof;
};

// Grow-only device workspaces.  Every buffer of a group is released and nulled, then all are allocated; if an
// allocation fails the ones already made are released again.  The caller zeroes the group's capacity before
// the call and sets it after success, so a failed search never leaves a freed pointer behind a non-zero
// capacity, nor a half-allocated group.  (hipFree synchronises the device: growth is rare by design.)
struct WsItem {
    void** ptr;
    size_t bytes;
};
inline int ws_regrow(std::initializer_list<WsItem> items) {
    for (const WsItem& it : items) {
        if (*it.ptr) (void)hipFree(*it.ptr);
        *it.ptr = nullptr;
    }
    for (const WsItem& it : items) {
        const hipError_t e = hipMalloc(it.ptr, it.bytes);
        if (e != hipSuccess) {
            *it.ptr = nullptr;
            for (const WsItem& j : items) {
                if (*j.ptr) (void)hipFree(*j.ptr);
                *j.ptr = nullptr;
            }
            (void)hipGetLastError();
            set_error("prag_index: workspace of %zu bytes: %s", it.bytes, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? PRAG_ENOMEM : PRAG_EHIP;
        }
    }
    return PRAG_OK;
}
template <typename T>
inline void** vpp(T** p) { return reinterpret_cast<void**>(p); }


// prag_index_search's body (flat_index.hip): tag_ids = ids in the exchange format of a row shard
int index_search_impl(prag_index_t* ix, const float* q, int B, int k, int64_t id_offset, float* D, int64_t* I, int io_is_device,
                      void* stream, int tag_ids, bool allow_mm8 = true);
