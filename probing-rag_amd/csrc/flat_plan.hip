// flat_plan.hip - the search plan of the flat index (see flat_plan.h): no device code, no HIP calls.
// Reference call being planned: `D, I = index.search(x, k)` (utils.py:379; exp_rag.py:432-436) on a
// faiss.IndexFlatL2-shaped index (make_indexer.py:449-450) - faiss has one code path; this index picks among six.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>

#include "flat_plan.h"
#include "prag.h"

namespace prag {

// ---------------------------------------------------------------------------------------------------------------
// Search plan: every dispatch decision of a search as a PURE function of the request shape and the index state
// (no pointers, no HIP calls): which kernel family makes the corpus pass, the tile height, the candidate depth,
// the grid, the bound strategy, the workspace it needs.  index_search_impl executes a plan; prag_plan_search
// describes one on a host without a GPU (tests/test_host_logic_cpu.py walks the shape grid); prag_index_last_plan
// is what bench.py prices its roofline with (round 3 re-derived the dispatch in Python).
// ---------------------------------------------------------------------------------------------------------------
int pick_kc(int k) {
    if (k <= 5) return 8;
    if (k <= 12) return 16;
    if (k <= 26) return 32;
    // deeper than a per-lane list can hold in registers: the MFMA-tiled scan keeps KC candidates per
    // query in memory (k plus a margin of near-ties, a multiple of 32, at most kMmMaxKc)
    const int kc = (k + std::max(16, k / 8) + 31) / 32 * 32;
    return kc <= kMmMaxKc ? kc : 0;
}

bool qs_supported(int d, int store, int kc) {
    // (d = 1024 with 16-deep lists: 128 VGPRs of query fragments + the lists spill - those batches take two
    // passes of the 64-query list kernel, or the 128-query shadow tiles when the index keeps a shadow)
    return store == PRAG_F16 && kc <= 16 && (d == 256 || d == 512 || d == 768 || (d == 1024 && kc <= 8));
}

SearchPlan plan_search(const PlanEnv& e) {
    SearchPlan P;
    const int B = e.B, k = e.k, d = e.d;
    int kc = std::max(pick_kc(k), pick_kc(k) ? e.kc_min : 0);
    if (kc == 0) return P;
    // float32 rows with 33..64 queries are rounded to fp16 inside the scan (no high-precision terms at
    // that tile height): the certificate's error bound is ~5e-4 ||q|| ||x||, which an 8-deep list clears
    // only ~99 % of the time at 21 M rows - and a miss costs a 64 GB exact pass.  A 16-deep list does.
    if (e.store == PRAG_F32 && kc == 8 && B > 32) kc = 16;
    // k > 26 on a dimension the MFMA-tiled scan does not cover: straight to the exact float64 scan
    P.exact_only = kc > 32 && !mm_supported(d, PRAG_F16, kc);
    if (P.exact_only) kc = 32;  // (sizes the unused candidate workspace)
    // > 128 queries on an index that keeps an up-to-date shadow: first tier = int8 tiles over the shadow with a
    // deep candidate list; the queries that fail the (much wider) certificate are searched again (mm8_second_tier)
    P.mm8_eligible = e.allow_mm8 && e.mm8_mode && e.mm_mode && !P.exact_only && kc <= 32 && B > 128 && e.ntotal > 0 &&
                     (e.shadow_mode >= 2 || e.ntotal >= e.mm8_min_rows) && e.cert_mode != 0 && e.shadow_ready &&
                     mm8_supported(d, kMm8Kc) && mm_supported(d, PRAG_F16, kMm8Kc) && k <= kMm8Kc / 8;
    P.use_mm8 = P.mm8_eligible && !e.mm8_auto_off;
    if (P.use_mm8) kc = kMm8Kc;
    P.kc = kc;
    P.qstride = (d * 2 + 255) / 256 * 256;
    // 64-query tiles when they fit LDS
    // (64 queries x 32-deep lists = 128 list registers per lane: scratch on either row type - two 32-query tiles)
    const bool wide_ok = 64 * P.qstride + 8 * 4096 + 64 * 12 <= 160 * 1024 && kc < 32;
    // > 128 queries: the contraction bounds the search -> MFMA-tiled scan, 256 queries per tile
    P.use_mm = !P.exact_only && e.ntotal > 0 && mm_supported(d, PRAG_F16, kc) && ((B > 128 && e.mm_mode) || kc > 32);
    // 65..128 queries: one pass over the 8-bit shadow with 128-query tiles when the index keeps one ...
    P.shadow128 = !P.exact_only && !P.use_mm && B > 64 && B <= 128 && e.cert_mode != 0 && e.ntotal > 0 && e.shadow_ready &&
                  shadow_supported(d, kc, k, B) && shadow_tile128_ok(d, kc);
    // ... else the query-stationary kernel over the fp16 rows (128 queries per corpus pass)
    P.use_qs = !P.exact_only && !P.use_mm && !P.shadow128 && B > 64 && qs_supported(d, e.store, kc);
    P.QT = P.use_mm ? 256 : (P.use_qs || P.shadow128) ? 128 : ((B > 32 && wide_ok) ? 64 : 32);
    // <= 32 queries (the reference's call shape): high-precision selection, if two query tiles fit LDS
    P.use_hp = e.hp_mode && P.QT == 32 && 2 * 32 * P.qstride + 8 * 4096 + 32 * 12 <= 160 * 1024;
    P.Bpad = (B + P.QT - 1) / P.QT * P.QT;
    P.n_tiles = (int)((e.ntotal + 31) / 32);
    P.cu_budget = e.wg_cap > 0 ? std::min(e.wg_cap, e.n_cu) : e.n_cu;
    P.grid = P.use_qs ? std::max(1, std::min(P.cu_budget, (P.n_tiles + 3) / 4))
                      : std::max(1, std::min(P.cu_budget, (P.n_tiles + 7) / 8));
    // (the pre-pass may use up to 64 workgroups; the tiled scan sizes its own fallback lists)
    P.part_need = P.use_mm ? 0 : (size_t)std::max(P.grid, 64) * P.QT * kc;
    P.cand_need = (size_t)P.Bpad * kc;
    // queries per mm_run call and survivors one workgroup can hold per query and segment.  Deep lists
    // (k > 26) yield up to KC/8 survivors per 256-row tile right after the first segment: they get
    // 512 slots and 256-query chunks.
    P.mm_chunk = P.use_mm8 ? std::min(P.Bpad, kMm8Chunk) : kc > 32 ? 256 : std::min(P.Bpad, kMmMaxQueries);
    P.mm_cap_wg = P.use_mm8 ? kMm8CapWg : kc > 32 ? 512 : kMmCapWg;
    P.certify = e.cert_mode != 0 || P.exact_only;
    P.ex_grid = (int)std::max<int64_t>(1, std::min<int64_t>(2 * (int64_t)P.cu_budget, (e.ntotal + 31) / 32));
    // flagged queries one round of the exact scan can hold: <= 64 MB of per-workgroup lists
    P.ex_fcap = (int)std::max<int64_t>(1, std::min<int64_t>(B, (64ll << 20) / ((int64_t)P.ex_grid * k * 12)));
    // (the shadow's eps constants read max ||x||^2 and its overflow path needs the exact scan: both exist
    // only with the certificate on - a diag build with PRAG_CERT=0 scans the rows directly)
    P.use_shadow = P.certify && !P.exact_only && !P.use_mm && !P.use_qs && e.ntotal > 0 && e.shadow_ready &&
                   shadow_supported(d, kc, k, B);
    // Bound for the list scan: slots filled inside the launch (no extra launches: best on shards
    // of a few million rows, where two launches are ~6 % of the search) or the pre-pass (its bound
    // is there from the first tile: measured 1.5 % faster at 21 M rows).  Crossover ~8 M rows.
    constexpr int64_t kSample = 8192;
    const bool want_slots = e.prepass_mode < 0 ? e.ntotal <= (8ll << 20) : e.prepass_mode == 0;
    P.use_slots = !P.use_qs && want_slots && !(P.QT == 64 && kc == 32);
    P.prepass = e.ntotal >= 16 * kSample && !P.use_slots;
    // ---- what the corpus pass is and what it has to read ----------------------------------------------------------
    const int64_t N = e.ntotal, l2 = e.metric == PRAG_METRIC_L2 ? 4 * N : 0;
    const int64_t elt_b = e.store == PRAG_F32 ? 4 : 2;
    if (N == 0) { P.family = "none (empty index)"; P.launches = 0; }
    else if (P.exact_only) { P.family = "exact_scan_kernel"; P.launches = B; P.bytes_per_launch = N * d * elt_b; }
    else if (P.use_mm) {
        P.family = P.use_mm8 ? "scan_mm_kernel<int8 tiles over the 8-bit shadow>" : "scan_mm_kernel";
        const int growth = mm_segment_growth(P.Bpad, P.mm_chunk, P.mm_cap_wg, P.cu_budget, kc, P.use_mm8);
        P.mm_growth = growth;
        // the schedule of mm_run (flat_mm.hip): [0, kMmFirstSeg), then every segment ends at g_step x its start
        int segs = 1;
        int64_t lo = 0, hi = std::min<int64_t>(N, kMmFirstSeg);
        while (hi < N) {
            lo = hi;
            const int g_step = mm_growth_step(growth, segs - 1, P.use_mm8);
            hi = std::min<int64_t>(N, hi * (int64_t)std::max(2, std::min(16, g_step)));
            ++segs;
        }
        P.last_seg_rows = hi - lo;
        P.launches = segs * ((P.Bpad + P.mm_chunk - 1) / P.mm_chunk);
        P.bytes_per_launch = P.use_mm8 ? N * (d + 4) + l2 : N * d * 2 + l2;
    } else if (P.use_shadow) {
        P.family = "scan8_kernel";
        P.launches = P.Bpad / P.QT;
        P.bytes_per_launch = N * (d + 12);     // 8-bit row + scale, error bound and the additive part of the key
    } else if (P.use_qs) {
        P.family = "scan_qs_kernel";
        P.launches = P.Bpad / 128;
        P.bytes_per_launch = N * d * 2 + l2;
    } else {
        P.family = "scan_topk_kernel";
        P.launches = P.Bpad / P.QT;
        P.bytes_per_launch = N * d * elt_b + l2;
    }
    // ---- workspace (the groups index_search_impl grows; bytes) ------------------------------------------------------
    size_t ws = (size_t)P.Bpad * (4 * 4 + 8 + 4 + (size_t)d * 4 + (size_t)d * 2 * 2 + 4 + kSlotWords * 4);   // per-query block
    ws += P.part_need * 8 + P.cand_need * 4;
    if (P.use_mm) {
        ws += (size_t)P.Bpad * 12 + 4;
        ws += (size_t)std::max(P.mm_chunk, e.n_cu) * std::max<size_t>(kMmCapQ, e.n_cu) * 12;
        ws += (size_t)e.n_cu * P.mm_chunk * P.mm_cap_wg * 8;
    }
    if (P.certify && N > 0)
        ws += exact_part_entries(P.ex_fcap, P.ex_grid, k) * 12 + (size_t)P.ex_fcap * 4 + exact_part_entries(P.ex_fcap, P.ex_grid, 1) * 8;
    if (P.use_shadow || P.use_mm8) {
        const size_t BpadS = P.use_mm8 ? P.Bpad : (size_t)(B + 63) / 64 * 64;
        ws += BpadS * (2 * (size_t)d + shadow_q_bytes() + shadow_slot_words() * 4 + 8);
        if (P.use_shadow) ws += (size_t)e.n_cu * (P.QT == 128 ? 128 : 64) * 512 * 8 + (size_t)e.n_cu * 128 * 4 + BpadS * shadow_split() * k * 12;
    }
    P.ws_bytes = ws;
    // (what shadow_search launches: the plan's record says the same)
    if (P.use_shadow) P.grid = std::max(1, std::min(shadow_scan_wg_cap(P.cu_budget, e.wg_cap <= 0, P.QT), (P.n_tiles + 7) / 8));
    return P;
}

int plan_describe(const PlanEnv& e, const SearchPlan& P, char* out, int cap) {
    return snprintf(out, cap,
                    "family=%s QT=%d kc=%d Bpad=%d grid=%d launches=%d bytes_per_launch=%lld ws_bytes=%zu hp=%d slots=%d "
                    "prepass=%d shadow=%d tiled=%d int8_tiles=%d exact_only=%d store=%s metric=%d d=%d rows=%lld queries=%d k=%d "
                    "mm_growth=%d last_seg_rows=%lld",
                    P.family, P.QT, P.kc, P.Bpad, P.grid, P.launches, (long long)P.bytes_per_launch, P.ws_bytes, (int)P.use_hp,
                    (int)P.use_slots, (int)P.prepass, (int)P.use_shadow, (int)P.use_mm, (int)P.use_mm8, (int)P.exact_only,
                    e.store == PRAG_F32 ? "f32" : "f16", e.metric, e.d, (long long)e.ntotal, e.B, e.k, P.mm_growth,
                    (long long)P.last_seg_rows);
}

}  // namespace prag

using namespace prag;

extern "C" int prag_plan_search(int d, int metric, int store_dtype, int64_t ntotal, int B, int k, int shadow_ready, int n_cu,
                                char* out, int cap) {
    PRAG_REQUIRE(out != nullptr && cap > 0, PRAG_EINVAL, "prag_plan_search: NULL buffer");
    PRAG_REQUIRE(d >= 16 && d % 16 == 0 && B >= 1 && k >= 1 && ntotal >= 0 && n_cu >= 1, PRAG_EINVAL,
                 "prag_plan_search: d=%d B=%d k=%d ntotal=%lld n_cu=%d", d, B, k, (long long)ntotal, n_cu);
    PlanEnv e;
    e.d = d; e.metric = metric; e.store = store_dtype; e.ntotal = ntotal; e.B = B; e.k = k; e.n_cu = n_cu;
    e.shadow_ready = shadow_ready != 0 && shadow_store_supported(d);
    e.shadow_mode = shadow_ready >= 2 ? 2 : 1;
    const SearchPlan P = plan_search(e);
    PRAG_REQUIRE(P.kc != 0, PRAG_EUNSUPPORTED, "k=%d: at most 911 results per query", k);
    plan_describe(e, P, out, cap);
    return PRAG_OK;
}

