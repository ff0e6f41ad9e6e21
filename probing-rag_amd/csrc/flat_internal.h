// Internal pieces shared by the flat-index translation units (flat_index.hip, flat_mm.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "prag_common.h"

namespace prag {

constexpr int kIdxSentinel = 0x7fffffff;

// monotone float <-> uint maps: selection keys are compared / atomically min-ed as integers
__device__ __forceinline__ uint32_t sortable_u32(float key) {
    const uint32_t u = __float_as_uint(key);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unsortable_f32(uint32_t u) {
    return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
__device__ __forceinline__ unsigned long long pack_key(float key, int idx) {
    return ((unsigned long long)sortable_u32(key) << 32) | (uint32_t)idx;
}
// Minimum of a 64-bit key over aligned groups of 4 / 16 lanes, result in every lane of the group.
// DPP moves (quad_perm, row_half_mirror, row_mirror) instead of __shfl_xor: a 64-bit shuffle is
// two ds_bpermute (~100+ cycles each, serially dependent); the KC selection rounds at the end of
// every scan launch were ~8 us per 32-query tile of pure cross-lane latency.
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_move_u64(unsigned long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, false);
    return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ unsigned long long min_u64(unsigned long long a, unsigned long long b) { return b < a ? b : a; }
__device__ __forceinline__ unsigned long long group_min4_u64(unsigned long long m) {
    m = min_u64(m, dpp_move_u64<0xB1>(m));   // quad_perm [1,0,3,2]
    m = min_u64(m, dpp_move_u64<0x4E>(m));   // quad_perm [2,3,0,1]
    return m;
}
__device__ __forceinline__ unsigned long long group_min16_u64(unsigned long long m) {
    m = group_min4_u64(m);
    m = min_u64(m, dpp_move_u64<0x141>(m));  // row_half_mirror: lane i <-> 7 - i inside each 8
    m = min_u64(m, dpp_move_u64<0x140>(m));  // row_mirror:      lane i <-> 15 - i inside each 16
    return m;
}

constexpr uint32_t kSortablePosInf = 0xFF800000u;  // sortable_u32(+inf)
constexpr uint32_t kSortableNegInf = 0x007FFFFFu;  // sortable_u32(-inf)

// A launch that is enqueued unconditionally and switched on the device: every kernel of the second tier of the
// 8-bit tiled selection carries one and returns at once when it is closed, so the tier decision needs no host
// round trip (a device-io search never waits for its stream, and every rank of a lockstep retrieval issues the
// same launches).  word == nullptr: always open.
struct Gate {
    const uint32_t* word = nullptr;
    uint32_t lo = 0, hi = 0;          // open iff lo <= *word <= hi
};
#ifdef __HIPCC__
__device__ __forceinline__ bool gate_closed(const Gate& g) {
    if (!g.word) return false;
    const uint32_t v = *g.word;       // written by an earlier kernel of the same stream
    return v < g.lo || v > g.hi;
}
#endif

// ---------------------------------------------------------------------------
// MFMA-tiled scan for large query batches (flat_mm.hip): 256 queries x 256 rows per
// workgroup tile, candidates collected by threshold filtering.
// ---------------------------------------------------------------------------
struct MmSearch {
    const _Float16* rows;   // [cap][d] fp16, cap a multiple of 256
    const float* xnorm;     // [cap]
    int64_t N;              // rows in the shard
    int d;
    const _Float16* q16;    // [Bpad][d], Bpad a multiple of 256, zero rows past B
    int B, Bpad;
    float alpha;            // key = (use_norm ? ||x||^2 : 0) + alpha * dot
    int use_norm;
    int kc;                 // candidates kept per query (8, 16 or 32)
    uint32_t* tau;          // [Bpad] sortable pruning bound per query: +inf for real queries, -inf for padding
    int* cand;              // [Bpad][kc] out: candidate row ids (-1 = none), ordered by (key, id)
    // workspace
    uint32_t* cnt;          // [Bpad]               candidates held in ckey/cidx[q][0..cnt)
    float* ckey;            // [Bpad][cap_q]        first segment (slot = row), afterwards the KC best so far
    int* cidx;
    uint32_t* ovf;          // [Bpad] set when a query lost candidates to a full region in some segment
    uint32_t* ovf_any;      // one word: set when any query of the search was flagged
    // (the caller presets cnt[q] = min(N, kMmFirstSeg), ovf[q] = 0, *ovf_any = 0 on the stream)
    int cap_q;              // >= kMmFirstSeg
    uint32_t* wcnt;         // [Bpad][wg_slots]          per-workgroup survivors of the running segment
    float* wkey;            // [wg_slots][Bpad][cap_wg]
    int* widx;
    int cap_wg;
    int wg_slots;           // workgroups the workspace was sized for
    int max_wg;             // workgroups the scan may occupy (<= wg_slots)
    int growth = 16;        // segment i+1 ends at growth x the end of segment i (2..16)
    // 8-bit selection (i8 != 0): the rows' shadow and the first int8 term of the queries on the int8 MFMA
    // (twice the fp16 rate); q16 / rows are not read.  key = xnorm + kq[b] * sscale[i] * (q8[b] . rows8[i])
    int i8 = 0;
    const signed char* rows8 = nullptr;   // [cap/32][d/128][32][128]
    const float* sscale = nullptr;        // [cap]
    const signed char* q8 = nullptr;      // [Bpad][d], zero rows past B
    const float* kq = nullptr;            // [Bpad], 0 past B
    Gate gate;
    bool shape16 = false;                 // 16 x 16 MFMA tiles instead of 32 x 32 (same results; see scan_mm_kernel)
};

constexpr int kMmFirstSeg = 2048;    // rows of the first segment (all of them become candidates); segments grow x16
// Segment boundaries: the first one grows by `growth` (x16 unless the candidate regions are smaller), the later ones
// by at most 4 - a filter group of 1024 scores holds a survivor with probability ~ 1 - exp(-1024 kc / rows seen at the
// last bound update), so a long segment behind a young bound takes the slow path in ~40 % of its groups.  Measured,
// 1000 x 1 M x 768 (profiles/r04n_c3_growth_schedule.txt): x16 everywhere 1.462 ms, x16 then x4 1.431 ms.  int8 tiles
// keep their own (already short) growth.
constexpr int kMmLaterGrowth = 4;
inline int mm_growth_step(int growth, int step, bool i8) {
    return (i8 || step == 0) ? growth : (growth < kMmLaterGrowth ? growth : kMmLaterGrowth);
}
constexpr int kMmCapQ = kMmFirstSeg; // candidate slots per query in ckey/cidx
constexpr int kMmCapWg = 64;         // survivors one workgroup can hold per query and segment
constexpr int kMmMaxQueries = 4096;  // queries per mm_run call (LDS counters); larger batches go in chunks
constexpr int kMmMaxKc = 1024;       // deepest candidate list (k up to 911: see pick_kc)

// In-LDS bitonic sort of n_pad (power of two) 64-bit keys, ascending; all NT threads of the block.
template <int NT>
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* s, int n_pad) {
    for (int size = 2; size <= n_pad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n_pad >> 1); t += NT) {
                const int i = ((t / stride) * 2 * stride) + (t % stride), j = i + stride;
                const unsigned long long a = s[i], b = s[j];
                const bool asc = (i & size) == 0;
                if ((a > b) == asc) {
                    s[i] = b;
                    s[j] = a;
                }
            }
        }
    }
    __syncthreads();
}
// order-preserving maps of float64 scores (exact rerank / exact scan): ascending uint64 == ascending double
__device__ __forceinline__ unsigned long long sortable_u64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return u ^ ((u >> 63) ? ~0ull : (1ull << 63));
}
__device__ __forceinline__ double unsortable_f64(unsigned long long u) {
    return __longlong_as_double((long long)(u ^ ((u >> 63) ? (1ull << 63) : ~0ull)));
}

// ---------------------------------------------------------------------------
// Exactness certificate.  The scans select KC >= k candidates per query by an
// APPROXIMATE key (fp16 operands through the matrix cores, f32 accumulation); the rerank then has
// the candidates' exact float64 scores.  Every row that is NOT a candidate has a selection key
// >= the KC-th candidate's selection key `kth_sel`, and a selection key differs from the exact key
// by at most eps (bound below).  So if the k-th best exact key is < kth_sel - eps no outsider can
// belong to the top k: the result is the definition's.  Otherwise the query is put on the flag list
// and recomputed by the exact float64 scan (flat_exact.hip).
//   eps_dot = rq*nx + nq*(c_row*nx + c_abs) + c_acc*nq*nx        (error of the selection dot product)
//     nq = ||q||, nx = max_i ||x_i||, rq = ||q - (fp16 terms of q the kernel used)|| (measured per query)
//     c_row, c_abs: relative / absolute rounding of the rows inside the kernel (0 for fp16 storage)
//     c_acc = d * n_chains * 2^-23: one rounding (RN or RZ) per product-accumulate, worst case
//   key = -dot (IP, COS) -> eps = eps_dot ;  key = ||x||^2 - 2 dot (L2) -> eps = 2 eps_dot + 2^-22 (xn_max + 2 nq nx)
// ---------------------------------------------------------------------------
// Result ids as they travel between shards ("tagged"): only float32 scores D are exchanged, so two rows of
// different shards whose float64 scores round to the same float32 would be ordered by id in the merge while an
// unsharded search orders them by the float64 value.  A tagged id carries, above the 40 bits of the global row
// id, the top 23 bits of the (order-preserving transform of the) float32 residual score64 - (double)D: the merge
// compares (D, residual, id) and strips the tag.  Resolution ~2^-14 of the residual, i.e. ~2^-38 of the score.
constexpr int kTagShift = 40;
__device__ __forceinline__ int64_t tag_id(int64_t gid, double score, int tag) {
    if (!tag) return gid;
    const float d32 = (float)score;
    const float r32 = (float)(score - (double)d32);
    const unsigned long long rk = sortable_u32(r32) >> 9;      // 23 bits: the result stays non-negative
    return (int64_t)((rk << kTagShift) | ((unsigned long long)gid & ((1ull << kTagShift) - 1)));
}

struct ShadowQ {     // per query, consumed by the scan
    float kscale;    // alpha * sq : key = xnorm_i + kscale * s_i * (acc1 + acc2 / 128)
    float A2, C2;    // two query terms:  eps_i = A e_i + C,  A = |alpha| ||q~||, C = |alpha| rq max||x|| + rounding slack
    float A1, C1;    // first term only (64-query tiles)
    float pad[3];
};

struct CertArgs {
    const float* qinfo;       // [B][4]: ||q||, ||q - q16||, ||q - q16 - q16lo||, unused  (rounded up)
    const double* qn2;        // [B] ||q||^2
    const uint32_t* xn_max;   // bits of max_i ||x_i||^2 (float)
    float c_row, c_abs, c_acc;
    int rq_sel;               // 1: the kernel used q16 only, 2: q16 + q16lo
    uint32_t* n_flag;         // number of flagged queries (zeroed by prep_queries_kernel)
    int* flag_list;           // [B]
    const uint32_t* force;    // optional [B]: non-zero = flag regardless (deep-list overflow), or null
    int tag_ids;              // 1: write tagged ids (prag_index_search_tagged)
    // 8-bit selection (MFMA-tiled scan over the shadow, first int8 query term): the selection key of row i is
    // within A1 e_i + C1 of its exact key (ShadowQ, same bound as the two-level search), so eps = A1 max_i e_i + C1
    const ShadowQ* sq8;       // [B] or null (fp16 selection: the model above)
    const uint32_t* e_max;    // float bits of max_i e_i
    const double* kshift;     // [B] with sq8: K_q = alpha q.mu of the shadow's affine map (exact key - K_q is the
                              // selection's key space, flat_shadow.hip); null: 0
    Gate gate;                // the rerank / merge-rerank / gather kernels return at once when it is closed
};

__device__ __forceinline__ double cert_eps(const CertArgs& c, int b, int metric_l2) {
    if (c.sq8) {   // (key units: |alpha| is inside A1 and C1)
        const double em = (double)__uint_as_float(*c.e_max) * (1.0 + 1e-6);
        return ((double)c.sq8[b].A1 * em + (double)c.sq8[b].C1) * 1.001;
    }
    const double nq = (double)c.qinfo[4 * b + 0];
    const double rq = (double)c.qinfo[4 * b + c.rq_sel];
    const double xn = (double)__uint_as_float(*c.xn_max) * (1.0 + 1e-6);
    const double nx = sqrt(xn);
    const double eps_dot = (rq * nx + nq * ((double)c.c_row * nx + (double)c.c_abs) + (double)c.c_acc * nq * nx) * 1.001;
    return metric_l2 ? 2.0 * eps_dot + 2.384185791015625e-07 * (xn + 2.0 * nq * nx) : eps_dot;
}

// true = the k best of the candidates are provably the k best of the shard
__device__ __forceinline__ bool cert_ok(const CertArgs& c, int b, int metric_l2, double kth_exact_score,
                                        float kth_sel) {
    if (c.force && c.force[b]) return false;
    if (!(kth_sel < INFINITY)) return true;   // the candidate list is not full: every row is in it
    double e = metric_l2 ? kth_exact_score - c.qn2[b] : -kth_exact_score;
    if (c.sq8 && c.kshift) e -= c.kshift[b];      // 8-bit selection over an affine shadow: same key space first
    return e < (double)kth_sel - cert_eps(c, b, metric_l2);
}

__device__ __forceinline__ void cert_flag(const CertArgs& c, int b) {
    const uint32_t slot = atomicAdd(c.n_flag, 1u);
    c.flag_list[slot] = b;
}

// Exact fallback: float64 brute force over every row for the flagged queries (flat_exact.hip).
struct ExactRun {
    const void* rows;       // [N][d] as stored
    int store_f32;
    int64_t N;
    int d;
    int metric_l2;
    const float* q32;       // [B][d] the queries the rerank uses (normalised for cosine)
    const uint32_t* n_flag;
    const int* flag_list;
    int B, k;
    int64_t id_offset;
    float* D;               // [B][k]
    int64_t* I;
    unsigned long long* part_key;   // workspace [f_cap][grid][k]
    int* part_id;
    int f_cap;              // flagged queries one round can hold
    int grid;               // workgroups of the scan
    uint32_t* done;         // [f_cap] zero-initialised arrival counters (the kernel leaves them zero)
    const uint32_t* xn_max = nullptr;       // float bits of max_i ||x_i||^2 over the shard (the inner-product margin of exact_mfma_kernel)
    unsigned long long* gpool = nullptr;    // [f_cap][grid] all-ones words (exact_mfma_kernel: each workgroup's best key; left all-ones)
    int tag_ids;
    Gate gate;
    bool grouped = false;   // several flagged queries expected: eight per pass over the rows (exact_group_kernel)
    bool mfma = true;       // ... sixteen per pass on v_mfma_f64_16x16x4_f64 where the shape allows (PRAG_EXACT_MFMA=0: never)
};
bool exact_mfma_supported(int d, int k);
size_t exact_part_entries(int f_cap, int grid, int k);
// Enqueue ceil(B / f_cap) launches of the exact scan (list merge folded in); each exits at once when no query is flagged.
int exact_run(const ExactRun& r, hipStream_t st);

// ---------------------------------------------------------------------------
// 8-bit shadow, two-level exact search (flat_shadow.hip)
// ---------------------------------------------------------------------------
// The 8-bit rows inside a 4-KiB chunk (32 rows x 128 bytes = 32 x 8 pieces of 16 bytes).  Round 5: the pieces are
// stored in the operand order of v_mfma_i32_32x32x32_i8 - k-step s of the chunk (bytes [32 s, 32 s + 32) of every row)
// is ONE KiB in which lane l = 32 hh + r of a wave finds bytes [32 s + 16 hh, + 16) of row r at byte 16 l - so the
// scan's coalesced 16-byte loads ARE the MFMA's A operand: no staging through LDS, no fragment reads of the rows.
// (-DPRAG_SHADOW_CHUNK_MAJOR: the row-major chunk of rounds 2-4, staged through LDS - kept for same-box A/B runs.)
#ifdef PRAG_SHADOW_CHUNK_MAJOR
constexpr bool kShadowFragMajor = false;
#else
constexpr bool kShadowFragMajor = true;
#endif
// byte offset of 16-byte piece p (0..7) of row r (0..31) inside its chunk
__host__ __device__ constexpr int shadow_piece_off(int r, int p) {
    return kShadowFragMajor ? (((p >> 1) * 64 + (p & 1) * 32 + r) << 4) : r * 128 + p * 16;
}

// Workgroups the two-level scan may use.  Scans of <= 64 queries are HBM-bound and FASTER on fewer workgroups than CUs
// (profiles/r05u_scan_wg_sweep.txt, 21 M rows, scan8 alone, 256 / 224 / 208 / 192 / 176 workgroups: 64 queries 2.547 /
// 2.506 / 2.504 / 2.500 / 2.552 ms, 1 query 2.37-2.45 / 2.339 / 2.327 / 2.325 / 2.377; 2.625 M rows, 64 queries: 0.383 /
// 0.373 / 0.381 / 0.394 / 0.416) - fewer concurrent 4-KiB streams, and CUs left to whatever runs beside the scan.
// 128-query tiles are bound by their epilogue and want every CU (2.83-2.87 / 2.90 / 2.94 / 3.02 / 3.11).  A caller's
// own cap (prag_index_set_scan_workgroups) is kept as given.  Which of the two an index uses is MEASURED on its own
// searches (flat_index.hip, WgTune): on embedding-shaped rows the same 64-query scans lose 5-8 % on 7/8 of the CUs
// (profiles/r05w_bench.json: 2.604 -> 2.729 ms at 21 M rows, 0.554 -> 0.599 at 4 M) where iid rows gain 1.6-3 %.
// Round 6: when the library picks the grid it takes the largest PRIME not above that count (256 CUs: 223 for 7/8, 251
// for "all").  Tiles are dealt to workgroups round-robin (tile t -> workgroup t mod grid: consecutive tiles to different
// workgroups, which is what keeps a run of similar rows out of one candidate region), so rows that repeat with a period
// of P tiles land in grid / gcd(grid, P) workgroups: with 256 workgroups and a period of 4096 rows (128 tiles) in TWO -
// their 512-entry regions overflow and 62 of 64 queries go to the retry tier or the float64 scan (2.2 / 14 ms per
// search against 0.75, profiles/r06i_clustered_interleaved.txt).  A prime grid spreads every period except its own
// multiples (7 136 rows for 223) over all workgroups.
inline int largest_prime_le(int n) {
    for (int c = n; c > 3; --c) {
        bool prime = (c & 1) != 0;
        for (int f = 3; prime && f * f <= c; f += 2) prime = c % f != 0;
        if (prime) return c;
    }
    return n > 1 ? n : 1;
}
inline int shadow_scan_wg_cap(int max_wg, bool auto_wg, int QT, bool every_cu = false) {
    if (!auto_wg) return max_wg;                        // the caller's own cap: kept as given
    const int want = (QT <= 64 && !every_cu) ? (max_wg * 7 / 8 > 1 ? max_wg * 7 / 8 : 1) : max_wg;
#ifdef PRAG_NO_PRIME_GRID       // (A/B builds: `make ab ABFLAGS=-DPRAG_NO_PRIME_GRID`)
    return want;
#else
    return want >= 16 ? largest_prime_le(want) : want;
#endif
}

// Bound slots of the two-level scan, per query: kShadowEpochs epochs x 32 slots filled inside the scan launch
// (after tiles 1, 2, 4, ..., 256) and one more "epoch" filled BEFORE it by prep_queries_kernel from a sample of
// the shard (below).
constexpr int kShadowEpochs = 9;
constexpr int kShadowPreEpoch = kShadowEpochs;                 // index of the sample slots
constexpr int kShadowSlotWords = (kShadowEpochs + 1) * 32;
constexpr int kShadowSampleSlices = 32;                        // one slot each
constexpr int kShadowSampleTiles = 4;                          // 32-row tiles per slice at most: 4096 sample rows (ShadowPrep::sample_tiles)

// what prep_queries_kernel writes for the two-level search (q8a == nullptr: nothing)
struct ShadowPrep {
    signed char* q8a;        // [Bpad][d]
    signed char* q8b;
    ShadowQ* sq;             // [Bpad]
    uint32_t* slots;         // [Bpad][kShadowSlotWords]
    uint32_t* ovf;           // [Bpad]
    uint32_t* done;          // [Bpad] arrival counters of the gather's list merge
    float* kq;               // optional [Bpad]: kscale of every query, contiguous (MFMA-tiled scan over the shadow)
    const uint32_t* xn_max;  // bits of max ||x||^2
    float alpha;
    const float* aff;        // [mu | c | 1/c] of the shadow's affine map (d floats each) or null
    const uint32_t* yn_max;  // bits of max ||y||^2, y = (x - mu) / c (aff null: xn_max is used)
    const uint32_t* bias_max;   // bits of max |sbias_i| (null: xn_max stands in)
    int centre_query;        // 1: the int8 terms approximate (q - mu) c (<= 128-query scans, rows carry sbias_i);
                             // 0: q c (int8 tiles of the > 128-query scans)
    double* kshift;          // [Bpad] out: K_q = alpha q.mu
    uint32_t* unfinished;    // zeroed here: queries the bound kernel leaves to the gather
    // sample for the pre-bound (sample_stride == 0: none, the slots of the pre-epoch stay +inf)
    const signed char* rows8;
    const float* sscale;
    const float* serr;
    const float* sbias;      // per-row additive part of the key (the scan's own array)
    int64_t sample_stride;   // in tiles: slice s, j-th tile = (s * sample_tiles + j) * sample_stride
    int sample_tiles;        // 32-row tiles per slice (1 .. kShadowSampleTiles)
};

#ifdef __HIPCC__
struct ShadowTerms {         // what one wave derives from its query
    float s1;
    double n1, r1, n2, r2;   // ||q~||^2 and ||q - q~||^2 with one / both int8 terms
};
// One wave, one query: v[it] holds elements [4 lane + 256 it, +4) of the query as the rerank will use it (zero
// past d).  Two int8 terms q ~ s1 (q1 + q2 / 128) - written to qa / qb (global or LDS) when non-null - and the
// residual norms of both truncations (float64).
__device__ __forceinline__ ShadowTerms shadow_terms_wave(int d, const f32x4 (&v)[6], int lane, signed char* qa,
                                                         signed char* qb) {
    float mx = 0.f;
#pragma unroll
    for (int it = 0; it < 6; ++it)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(v[it][e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    ShadowTerms t;
    t.s1 = mx > 0.f ? mx / 127.0f : 1.0f;
    const float s1 = t.s1, s2 = s1 * (1.0f / 128.0f);          // exact (power of two)
    double r2 = 0.0, n2 = 0.0, r1 = 0.0, n1 = 0.0;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < d) {
            uint32_t wa = 0, wb = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = v[it][e];
                const float a = fminf(fmaxf(rintf(x / s1), -127.f), 127.f);
                const float rem = x - s1 * a;
                const float bq = fminf(fmaxf(rintf(rem / s2), -127.f), 127.f);
                wa |= ((uint32_t)(int)a & 0xFFu) << (8 * e);
                wb |= ((uint32_t)(int)bq & 0xFFu) << (8 * e);
                const double qt = (double)s1 * (double)a + (double)s2 * (double)bq;
                const double df = (double)x - qt;
                r2 = fma(df, df, r2);
                n2 = fma(qt, qt, n2);
                const double q1 = (double)s1 * (double)a, d1 = (double)x - q1;
                r1 = fma(d1, d1, r1);
                n1 = fma(q1, q1, n1);
            }
            if (qa) *reinterpret_cast<uint32_t*>(qa + c) = wa;
            if (qb) *reinterpret_cast<uint32_t*>(qb + c) = wb;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        r2 += __shfl_xor(r2, o, 64);
        n2 += __shfl_xor(n2, o, 64);
        r1 += __shfl_xor(r1, o, 64);
        n1 += __shfl_xor(n1, o, 64);
    }
    t.n1 = n1; t.r1 = r1; t.n2 = n2; t.r2 = r2;
    return t;
}
// constants of eps_i = A e_i + C (every lane computes the same values)
__device__ __forceinline__ ShadowQ shadow_consts(const ShadowTerms& t, float alpha, const uint32_t* xn_max,
                                                 const uint32_t* yn_max = nullptr, const uint32_t* bias_max = nullptr) {
    const double aa = fabs((double)alpha);
    // the additive part of a key (||x||^2 for L2, plus alpha mu.(x - mu) under the affine map): its float32 rounding
    // and the rounding of key = bias + scale * acc are inside 1e-6 of its largest magnitude
    const double xn0 = (double)__uint_as_float(*xn_max) * (1.0 + 1e-6);
    const double xn = fmax(xn0, bias_max ? (double)__uint_as_float(*bias_max) * (1.0 + 1e-6) : 0.0);
    // the dot product the int8 terms approximate is p.y (p = q c, y = (x - mu) / c): its residual term and its
    // float32 roundings scale with max ||y||; the L2 key also carries ||x||^2 (the xn term of the rounding slack)
    const double nx = yn_max ? sqrt((double)__uint_as_float(*yn_max) * (1.0 + 1e-6)) : sqrt(xn0);
    ShadowQ o;
    o.kscale = alpha * t.s1;
    // eps = A e_i + C;  C = query residual against the largest row + float32 roundings of key / eps
    auto consts = [&](double nq, double rq, float& A, float& C) {
        A = (float)(aa * nq * (1.0 + 1e-5)) + FLT_MIN;
        C = (float)((aa * rq * nx + 1e-6 * (xn + 2.0 * aa * (nq + rq) * nx)) * (1.0 + 1e-5)) + FLT_MIN;
    };
    consts(sqrt(t.n2), sqrt(t.r2), o.A2, o.C2);
    consts(sqrt(t.n1), sqrt(t.r1), o.A1, o.C1);
    o.pad[0] = o.pad[1] = o.pad[2] = 0.f;
    return o;
}
// The wave that owns query b in prep_queries_kernel (`real` false: a padding row - zeros, constants 0).
__device__ __forceinline__ void shadow_prep_wave(const ShadowPrep& p, int b, int d, bool real, const f32x4 (&v)[6],
                                                 int lane) {
    // (the sample slots of the pre-epoch belong to the sampling waves - when there are any)
    const int n_init = p.sample_stride > 0 ? kShadowEpochs * 32 : kShadowSlotWords;
    for (int w = lane; w < n_init; w += 64) p.slots[(int64_t)b * kShadowSlotWords + w] = kSortablePosInf;
    if (lane == 0) {
        p.ovf[b] = 0u;
        p.done[b] = 0u;
    }
    if (!real) {
        for (int c = lane * 4; c < d; c += 256) {
            *reinterpret_cast<uint32_t*>(p.q8a + (int64_t)b * d + c) = 0u;
            *reinterpret_cast<uint32_t*>(p.q8b + (int64_t)b * d + c) = 0u;
        }
        if (lane == 0) {
            p.sq[b] = ShadowQ{0.f, 0.f, 0.f, 0.f, 0.f, {0.f, 0.f, 0.f}};
            if (p.kq) p.kq[b] = 0.f;
            if (p.kshift) p.kshift[b] = 0.0;
        }
        return;
    }
    const ShadowTerms t = shadow_terms_wave(d, v, lane, p.q8a + (int64_t)b * d, p.q8b + (int64_t)b * d);
    if (lane == 0) {
        const ShadowQ c = shadow_consts(t, p.alpha, p.xn_max, p.aff ? p.yn_max : nullptr, p.bias_max);
        p.sq[b] = c;
        if (p.kq) p.kq[b] = c.kscale;
    }
}
// Pre-bound: slice `slice` of the sample - kShadowSampleTiles 32-row tiles of the shadow, spread over the shard -
// scored against the FIRST int8 term of query b by this wave (v_dot4_i32_i8: exact integers, the scan's own
// selection score), slot = min over the slice of key + a eps >= the exact key of that row.  The slices are
// disjoint, so the k'-th smallest slot bounds the k'-th best exact key of the shard: the scan starts with a bound
// near the 0.4 % quantile instead of +inf - no warm-up tiles to visit twice, no flood of early candidates.
// q8_lds: 1024 bytes of this wave's LDS.
constexpr int kShadowSampleRing = 4;
// Lane -> bytes of a chunk: the 16 bytes at 16 lane + 1024 i, i = 0..3.  Fragment-major chunks: piece 2 i + lane / 32
// of row lane % 32 (a row is shared by lanes l and l ^ 32, one metadata triple per lane); row-major chunks: piece
// lane % 8 of row 8 i + lane / 8 (a row is shared by 8 lanes, four triples per lane).
constexpr int kShadowSampleRows = kShadowFragMajor ? 1 : 4;
struct ShadowSample {        // what a sampling wave requests before it even looks at its query
    i32x4 ring[kShadowSampleRing][4];
    float rs[kShadowSampleRows], re[kShadowSampleRows], rx[kShadowSampleRows];     // scale, error bound, bias of the row(s) of the tile being scored
};
__device__ __forceinline__ void shadow_sample_meta(const ShadowPrep& p, int slice, int j, int lane, ShadowSample& sm) {
    const int jc = j < p.sample_tiles ? j : p.sample_tiles - 1;
    const int64_t tile = ((int64_t)slice * p.sample_tiles + jc) * p.sample_stride;
#pragma unroll
    for (int i = 0; i < kShadowSampleRows; ++i) {
        const int64_t row = tile * 32 + (kShadowFragMajor ? (lane & 31) : 8 * i + (lane >> 3));       // (whole tiles below N only)
        sm.rs[i] = p.sscale[row];
        sm.re[i] = p.serr[row];
        sm.rx[i] = p.sbias ? p.sbias[row] : 0.f;
    }
}
__device__ __forceinline__ const signed char* shadow_sample_chunk(const ShadowPrep& p, int d, int slice, int lane, int sidx) {
    const int nch = d >> 7, total = p.sample_tiles * nch;
    const int sc = sidx < total ? sidx : total - 1;
    const int j = sc / nch, ch = sc - j * nch;
    const int64_t tile = ((int64_t)slice * p.sample_tiles + j) * p.sample_stride;
    return p.rows8 + tile * (32 * (int64_t)d) + ch * 4096 + lane * 16;
}
// The sample tiles are cold (the scan of the previous search has been through every cache since) and scattered
// over the shard: a ring of chunks is requested at once, right after the query has been quantised.  (Requesting
// it BEFORE the float64 work on the query was tried: the ring's registers are then live across that work, the
// kernel needs > 168 VGPRs or spills - and 64 queries x 33 waves only fit the chip in one round at 3 waves per SIMD.)
__device__ __forceinline__ void shadow_sample_issue(const ShadowPrep& p, int d, int slice, int lane, ShadowSample& sm) {
#pragma unroll
    for (int u = 0; u < kShadowSampleRing; ++u) {
        const signed char* src = shadow_sample_chunk(p, d, slice, lane, u);
#pragma unroll
        for (int i = 0; i < 4; ++i) sm.ring[u][i] = *reinterpret_cast<const i32x4*>(src + i * 1024);
    }
    shadow_sample_meta(p, slice, 0, lane, sm);
}
__device__ __forceinline__ void shadow_prebound_none(const ShadowPrep& p, int b, int slice, int lane) {
    if (lane == 0) p.slots[(int64_t)b * kShadowSlotWords + kShadowPreEpoch * 32 + slice] = kSortablePosInf;
}
__device__ __forceinline__ void shadow_prebound_wave(const ShadowPrep& p, int b, int d, int slice, const f32x4 (&v)[6],
                                                     int lane, signed char* q8_lds, ShadowSample& sm) {
    uint32_t* slot = p.slots + (int64_t)b * kShadowSlotWords + kShadowPreEpoch * 32 + slice;
    const ShadowTerms t = shadow_terms_wave(d, v, lane, q8_lds, nullptr);
    const ShadowQ c = shadow_consts(t, p.alpha, p.xn_max, p.aff ? p.yn_max : nullptr, p.bias_max);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    shadow_sample_issue(p, d, slice, lane, sm);
    const int nch = d >> 7;
    [[maybe_unused]] const int piece = lane & 7;
    float best = INFINITY;
    // the slice as one stream of 4-KiB chunks (tile-major, chunk-minor), a ring of them in flight
    constexpr int kRing = kShadowSampleRing;
    const int total = p.sample_tiles * nch;
    int acc[kShadowSampleRows] = {};
    int ch = 0, j = 0;
    for (int s0 = 0; s0 < total; s0 += kRing) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {
            if (s0 + u < total) {
                if constexpr (kShadowFragMajor) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const i32x4 qv = *reinterpret_cast<const i32x4*>(q8_lds + ch * 128 + (2 * i + (lane >> 5)) * 16);
#pragma unroll
                        for (int w = 0; w < 4; ++w) acc[0] = __builtin_amdgcn_sdot4(sm.ring[u][i][w], qv[w], acc[0], false);
                    }
                } else {
                    const i32x4 qv = *reinterpret_cast<const i32x4*>(q8_lds + ch * 128 + piece * 16);
#pragma unroll
                    for (int i = 0; i < kShadowSampleRows; ++i)
#pragma unroll
                        for (int w = 0; w < 4; ++w) acc[i] = __builtin_amdgcn_sdot4(sm.ring[u][i][w], qv[w], acc[i], false);
                }
                const signed char* src = shadow_sample_chunk(p, d, slice, lane, s0 + u + kRing);   // (clamped: a harmless re-read)
#pragma unroll
                for (int i = 0; i < 4; ++i) sm.ring[u][i] = *reinterpret_cast<const i32x4*>(src + i * 1024);
                if (++ch == nch) {       // tile finished: the lanes that share a row add their parts (exact integers)
                    ch = 0;
#pragma unroll
                    for (int i = 0; i < kShadowSampleRows; ++i) {
                        int a = acc[i];
                        if constexpr (kShadowFragMajor) {
                            a += __shfl_xor(a, 32, 64);
                        } else {
#pragma unroll
                            for (int o = 1; o < 8; o <<= 1) a += __shfl_xor(a, o, 64);
                        }
                        acc[i] = 0;
                        const float mid = fmaf(c.kscale * sm.rs[i], (float)a, sm.rx[i]);
                        const float eps = fmaf(c.A1, sm.re[i], c.C1);
                        best = fminf(best, mid + eps);
                    }
                    ++j;
                    shadow_sample_meta(p, slice, j, lane, sm);     // (of the next tile: nch chunks ahead of its use)
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fminf(best, __shfl_xor(best, o, 64));
    if (lane == 0) *slot = sortable_u32(best);
}
#endif

struct ShadowStore {
    const void* rows;        // the stored rows (exact scores come from these)
    int store_f32;
    int d;
    signed char* rows8;      // [cap/32][d/128][4 KiB chunk]: 8-bit rows, chunk-major inside 32-row tiles, pieces at shadow_piece_off
    float* sscale;           // [cap] s_i
    float* serr;             // [cap] e_i = ||x_i - s_i x^_i||, rounded up
    uint32_t* err_max;       // float bits of max_i e_i (diagnostic)
    const float* aff = nullptr;   // [mu | c | 1/c] of the affine map y = (x - mu) / c the shadow quantises (null: y = x)
    uint32_t* yn_max = nullptr;   // float bits of max_i ||y_i||^2
    float* sbias = nullptr;       // [cap] per-row additive part of the scan's key: alpha mu.(x_i - mu) [+ ||x_i||^2, L2]
    uint32_t* bias_max = nullptr; // float bits of max_i |sbias_i|
    const float* xnorm_l2 = nullptr;   // ||x_i||^2 of the stored rows when the metric is L2 (build time), else null
    float alpha = -1.0f;          // key = alpha * dot (+ ||x||^2 for L2)
};
struct ShadowSearch {
    ShadowStore store;
    const float* xnorm;
    int64_t N;
    int d, metric_l2;
    float alpha;             // key = (L2 ? ||x||^2 : 0) + alpha * sel
    const float* q32;        // [B][d] as the rerank uses them
    const uint32_t* xn_max;
    int B, Bpad_ws, k, kc;
    int qt_max;              // query-tile height the caller padded for (32, 64 or 128)
    int64_t id_offset;
    float* D;
    int64_t* I;
    uint32_t* g_tau;         // [Bpad] preset by prep_queries_kernel
    // workspace
    signed char* q8a;        // [Bpad][d]
    signed char* q8b;
    void* sq;                // [Bpad] ShadowQ
    uint32_t* slots;         // [Bpad][shadow_slot_words()]
    int* cand;               // [wg_slots][64][cap] x 2 ints (row id, bits of key - a eps)
    uint32_t* ccnt;          // [wg_slots][64]
    int cap, wg_slots, max_wg;
    bool auto_wg = false;    // max_wg is "every CU", not a caller's choice: HBM-bound tiles take 7/8 of it (shadow_scan_wg_cap)
    bool every_cu = false;   // ... unless the index measured that its scans want every CU (WgTune) or PRAG_SCAN_WG_TUNE=1 says so
    unsigned long long* part_key;   // [Bpad][shadow_split()][k]
    int* part_id;
    uint32_t* ovf;           // [Bpad]
    uint32_t* done;          // [Bpad]
    CertArgs cert;           // flag list: queries whose candidate regions overflowed go to the exact scan
    Gate gate;
    const double* kshift = nullptr;   // [Bpad] K_q of every query (prep_queries_kernel) or null
    hipEvent_t scan_done = nullptr;   // recorded behind the scan of the last query tile (prag_index_stream_wait_scan)
    struct TailGate* tail = nullptr;  // a gate launch to carry beside the bound kernel of the last query tile (tail_gate.h)
    hipEvent_t time_ev0 = nullptr, time_ev1 = nullptr;   // recorded around the scan launch of the last query tile (workgroup tuning)
    bool timed_recorded = false;   // out: both events were recorded by this call
    int grid_used = 0;                // out: workgroups of the scan launches
    int scan_gate_mode = -1;          // ... or behind the SCAN's workgroups in its launch (flat_scan_gate.hip): -1 when the gate
                                      // fits under the scan, 0 never, 1 always (PRAG_SCAN_GATE)
    uint32_t* unfinished = nullptr;   // device word: queries the bound kernel left to the gather (zeroed by the prep kernel)
    bool skip_gather = false;         // no sliced gather behind the bound kernel (recent searches never needed one)
    bool exact_bound = false;   // lower g_tau to an exact k-th best before the gather (shadow_bound_kernel)
    int64_t quad_min_rows = (int64_t)8 << 20;   // shards from this size on scan with the quad-test epilogue (flat_shadow.hip)
};
bool shadow_store_supported(int d);
bool shadow_tile128_ok(int d, int kc);
bool shadow_supported(int d, int kc, int k, int B);
size_t shadow_slot_words();
size_t shadow_q_bytes();
int shadow_split();
int shadow_build(const ShadowStore& s, int64_t row0, int64_t row1, hipStream_t st);
int shadow_affine_fit(const ShadowStore& s, int64_t n_rows, int identity, double* sums, hipStream_t st);
int shadow_search(ShadowSearch& s, hipStream_t st, EventRing& prof);

bool mm_supported(int d, int store_dtype, int kc);
bool mm8_supported(int d, int kc);
// Enqueue the segmented scan on `st`; returns PRAG_OK or a negative status.  `prof` brackets
// the launch over the largest segment.
int mm_run(const MmSearch& s, hipStream_t st, EventRing& prof);

}  // namespace prag
