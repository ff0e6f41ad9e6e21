// Internal pieces shared by the flat-index translation units (flat_index.hip, flat_mm.hip).
#pragma once

#include <hip/hip_runtime.h>
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
};

constexpr int kMmFirstSeg = 2048;    // rows of the first segment (all of them become candidates); segments grow x16
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
struct CertArgs {
    const float* qinfo;       // [B][4]: ||q||, ||q - q16||, ||q - q16 - q16lo||, unused  (rounded up)
    const double* qn2;        // [B] ||q||^2
    const uint32_t* xn_max;   // bits of max_i ||x_i||^2 (float)
    float c_row, c_abs, c_acc;
    int rq_sel;               // 1: the kernel used q16 only, 2: q16 + q16lo
    uint32_t* n_flag;         // number of flagged queries (zeroed by prep_queries_kernel)
    int* flag_list;           // [B]
    const uint32_t* force;    // optional [B]: non-zero = flag regardless (deep-list overflow), or null
};

__device__ __forceinline__ double cert_eps(const CertArgs& c, int b, int metric_l2) {
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
    const double e = metric_l2 ? kth_exact_score - c.qn2[b] : -kth_exact_score;
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
};
size_t exact_part_entries(int f_cap, int grid, int k);
// Enqueue ceil(B / f_cap) rounds of {exact scan, merge}; every launch exits at once when no query is flagged.
int exact_run(const ExactRun& r, hipStream_t st);

// ---------------------------------------------------------------------------
// 8-bit shadow, two-level exact search (flat_shadow.hip)
// ---------------------------------------------------------------------------
struct ShadowStore {
    const void* rows;        // the stored rows (exact scores come from these)
    int store_f32;
    int d;
    signed char* rows8;      // [cap/32][d/128][32][128]: 8-bit rows, chunk-major inside 32-row tiles
    float* sscale;           // [cap] s_i
    float* serr;             // [cap] e_i = ||x_i - s_i x^_i||, rounded up
    uint32_t* err_max;       // float bits of max_i e_i (diagnostic)
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
    int qt_max;              // query-tile height the caller padded for (32 or 64)
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
    unsigned long long* part_key;   // [Bpad][shadow_split()][k]
    int* part_id;
    uint32_t* ovf;           // [Bpad]
    CertArgs cert;           // flag list: queries whose candidate regions overflowed go to the exact scan
};
bool shadow_store_supported(int d);
bool shadow_supported(int d, int kc, int k, int B);
size_t shadow_slot_words();
size_t shadow_q_bytes();
int shadow_split();
int shadow_build(const ShadowStore& s, int64_t row0, int64_t row1, hipStream_t st);
int shadow_search(const ShadowSearch& s, hipStream_t st, EventRing& prof);

bool mm_supported(int d, int store_dtype, int kc);
// Enqueue the segmented scan on `st`; returns PRAG_OK or a negative status.  `prof` brackets
// the launch over the largest segment.
int mm_run(const MmSearch& s, hipStream_t st, EventRing& prof);

}  // namespace prag
