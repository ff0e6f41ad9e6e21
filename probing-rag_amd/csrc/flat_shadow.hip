// Two-level exact search: an 8-bit shadow of the stored rows is scanned instead of the rows
// themselves (half the bytes of fp16 storage, a quarter of float32), a proof-carrying filter keeps
// every row that can still belong to the top k, and the survivors are scored exactly (float64) from
// the stored rows.  Results are the definition's (faiss.IndexFlat semantics; /root/reference:
// make_indexer.py:449-450, utils.py:378-380) - the shadow only decides which rows need not be looked at.
//
// Shadow (built on the device from the stored rows):
//   y_i = (x_i - mu) / c   (round 5: a per-INDEX affine map, mu = column means of a sample of the rows, frozen when the
//                           shadow is first built: real embeddings share a mean direction and carry a few outlier
//                           coordinates, and a per-row abs-max int8 grid spent its levels on those instead of on what
//                           separates one row from another.  c = 1 by default; PRAG_SHADOW_AFFINE=2 adds power-of-two
//                           column scales - see shadow_affine_mode in flat_index.hip for why that is not the default)
//   y^_i = rint(y_i / s_i) in [-127,127], s_i = max|y_i| / 127 ;  e_i = ||y_i - s_i y^_i|| (rounded up)
// Query (per search): p = q * c (exact: powers of two), so that q.x_i = q.mu + p.y_i; two int8 terms,
// p ~ sp (p^1 + p^2 / 128), residual rq = ||p - p~|| ~ 2^-15 ||p||.
// Selection score  sel = sp s_i (p^1.y^_i + p^2.y^_i / 128)   (v_mfma_i32_32x32x32_i8: exact integers)
// differs from p.y_i = q.x_i - q.mu by at most
//   eps_i = ||p~|| e_i + rq max||y||     (Cauchy-Schwarz; ~0.8 % of ||p|| ||y|| for 768 Gaussian elements)
// so with key = -sel (IP, COS) or ||x_i||^2 - 2 sel (L2):  key - a eps_i <= exact key - K_q <= key + a eps_i, where
// K_q = alpha q.mu is one constant per query (alpha = -1 / -2): every comparison inside the scan is between keys of
// the same query, and the two places where an EXACT key meets the scan's key space (shadow_bound_kernel, the
// certificate of the int8 tiles) subtract K_q first.  mu = 0, c = 1 is the round 2-4 shadow.
// The <= 128-query scans also take the mean out of the QUERY: q.x_i = q.mu + mu.(x_i - mu) + (q - mu).(x_i - mu); the
// middle term is a per-ROW constant (sbias_i = alpha mu.(x_i - mu) [+ ||x_i||^2 for L2], float64 at build time,
// stored as float32 next to s_i and e_i) and only p' = (q - mu) c meets the int8 rows.  Queries live in the rows'
// space (the same encoder made both, utils.py:365-366, make_indexer.py:447-456): without this a query's own outlier
// coordinates set its int8 scale and the 64-query tiles (ONE query term) resolved the other 760 coordinates with
// three levels - measured on embedding-shaped rows: 180 000 - 870 000 survivors per query of 1 M rows and every query in
// the exact scan (profiles/r05b_embedding_probe_centred_rows_only.txt).
// Filter: tau is an upper bound on the k-th best EXACT key as soon as k rows with key + a eps <= tau
// have been seen (per-lane lists of key_hi = key + a eps, shared through LDS and the chip-wide bound
// slots, exactly as the fp16 scan shares its bound).  A row is dropped only if key - a eps_i > tau,
// i.e. only if it provably is not among the k best; everything else is a candidate.  tau only
// tightens, so rows seen early are tested against a looser bound: a superset, never a miss.
// Candidates go to per-(workgroup, query) regions (LDS counter, fire-and-forget stores); a region
// that overflows flags its query for the exact float64 scan (flat_exact.hip).
// Warm-up: while a wave's bound is still +inf (its first few tiles, until the bound slots of the
// first epoch arrive) nothing can be dropped, so nothing is collected either: those tiles only feed
// the lists, and the wave visits them a second time at the end of its scan - filter only, no list
// pushes (a row must not enter a list twice) - when the bound is tight.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "flat_internal.h"
#include "scan8_body.h"
#include "tail_gate.h"

namespace prag {



// (PRAG_SH_DBG, the timing-only knob of the `make diag` build: scan8_body.h)

// ---------------------------------------------------------------------------
// shadow build: one 256-thread workgroup per 32-row tile.  Thread (row = tid >> 3, piece = tid & 7) owns,
// in every 128-element chunk of its row, the 16 elements [16 piece, 16 piece + 16): it reads them with
// 16-byte loads (8 lanes of a row cover 256 contiguous bytes of fp16, 512 of float32; every load of the
// tile is issued before the first use), the row's max |x| and residual norm are 8-lane butterflies (fixed
// association: the build is bit-reproducible), and the 16 quantised bytes go out as ONE 16-byte store to
// rows8[tile][chunk] + shadow_piece_off(row, piece) (flat_internal.h: the MFMA operand order) - the workgroup writes
// each 4-KiB chunk block whole, 128 contiguous bytes per 8 lanes.
// (Round 2 built one wave per row with 2-byte loads and 1-byte stores: 238 ms for 21 M rows, 0.2 TB/s.)
// ---------------------------------------------------------------------------
template <bool F32, int NCH>
__global__ __launch_bounds__(256) void shadow_build_kernel(const void* __restrict__ rows, int64_t tile0, int64_t n_rows,
                                                          signed char* __restrict__ rows8,
                                                          float* __restrict__ sscale, float* __restrict__ serr,
                                                          uint32_t* __restrict__ err_max, const float* __restrict__ aff,
                                                          uint32_t* __restrict__ yn_max, float* __restrict__ sbias,
                                                          const float* __restrict__ xnorm, float alpha,
                                                          uint32_t* __restrict__ bias_max) {
    constexpr int d = NCH * 128;
    // aff = [mu | c | 1/c] (d floats each) or null: y = (x - mu) * (1/c), one float32 rounding (the subtraction)
    // sbias_i = alpha mu.(x_i - mu) (float64 sum, rounded once) + ||x_i||^2 when xnorm is given (L2)
    const int tid = threadIdx.x;
    const int row_in = tid >> 3, piece = tid & 7;
    const int64_t tile = tile0 + blockIdx.x;
    const int64_t i = tile * 32 + row_in;
    const bool valid = i < n_rows;                 // rows past the end of the shard quantise to zeros
    float v[NCH][16];
    double mdot = 0.0;                             // this thread's part of mu.(x - mu)
    if (valid) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if constexpr (F32) {
                const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(rows) + i * d + c * 128 + piece * 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 x = p[u];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[c][4 * u + e] = x[e];
                }
            } else {
                const half8* p = reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(rows) + i * d + c * 128 + piece * 16);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const half8 x = p[u];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[c][8 * u + e] = (float)x[e];
                }
            }
        }
        if (aff) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f32x4* pm = reinterpret_cast<const f32x4*>(aff + c * 128 + piece * 16);
                const f32x4* pi = reinterpret_cast<const f32x4*>(aff + 2 * d + c * 128 + piece * 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 m4 = pm[u], i4 = pi[u];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ctr = v[c][4 * u + e] - m4[e];
                        mdot = fma((double)m4[e], (double)ctr, mdot);
                        v[c][4 * u + e] = ctr * i4[e];
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) v[c][e] = 0.f;
    }
    float mx = 0.f, yn2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) yn2 = fmaf(v[c][e], v[c][e], yn2);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) yn2 += __shfl_xor(yn2, o, 64);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(v[c][e]));
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float s = mx > 0.f ? mx / 127.0f : 1.0f;
    const float inv = mx > 0.f ? 127.0f / mx : 1.0f;   // (any integer in [-127,127] is a valid quantisation: the
                                                       //  residual below is measured on the one actually stored)
    // residual norm in float32: df = fma(-s, q, x) is the exact difference rounded once, the sum of <= 1024 squares
    // is off by < 1024 * 2^-24 relatively - covered by the 1e-4 the bound is rounded up by.  (The float64 version
    // made the build float64-issue-bound: 31.5 ms for 21 M x 768 rows.)
    float err2 = 0.f;
    signed char* out = rows8 + tile * (32 * (int64_t)d) + shadow_piece_off(row_in, piece);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        u32x4 pk;
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            uint32_t word = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = v[c][4 * wd + e];
                float qv = rintf(x * inv);
                qv = fminf(fmaxf(qv, -127.f), 127.f);
                word |= ((uint32_t)(int)qv & 0xFFu) << (8 * e);
                const float df = fmaf(-s, qv, x);
                err2 = fmaf(df, df, err2);
            }
            pk[wd] = word;
        }
        *reinterpret_cast<u32x4*>(out + c * 4096) = pk;
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) err2 += __shfl_xor(err2, o, 64);
    float e_row = 0.f;
    if (piece == 0 && valid) {
        e_row = sqrtf(err2) * (1.0f + 2e-4f) + FLT_MIN;   // rounded up: it feeds a bound
        sscale[i] = s;
        serr[i] = e_row;
    }
    // one atomic per wave for the (diagnostic) largest residual
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e_row = fmaxf(e_row, __shfl_xor(e_row, o, 64));
    if ((tid & 63) == 0 && e_row > 0.f) atomicMax(err_max, __float_as_uint(e_row));
    // max_i ||y_i||^2 (rounded up: <= 1024 float32 squares): what the query-residual term of eps multiplies
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) mdot += __shfl_xor(mdot, o, 64);
    float b_row = 0.f;
    if (sbias && piece == 0) {
        // one rounding: |sbias_i - exact| <= 2^-24 |sbias_i|, inside the 1e-6 |bias|max slack of the eps constants
        b_row = valid ? (float)((double)alpha * mdot + (xnorm ? (double)xnorm[i] : 0.0)) : 0.f;
        sbias[i] = b_row;
    }
    b_row = fabsf(b_row);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b_row = fmaxf(b_row, __shfl_xor(b_row, o, 64));
    if ((tid & 63) == 0 && b_row > 0.f && bias_max) atomicMax(bias_max, __float_as_uint(b_row));
    float yn = valid ? yn2 * (1.0f + 2e-4f) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) yn = fmaxf(yn, __shfl_xor(yn, o, 64));
    if ((tid & 63) == 0 && yn > 0.f && yn_max) atomicMax(yn_max, __float_as_uint(yn));
}

// ---------------------------------------------------------------------------
// The affine map of the shadow: column sums of a sample of the stored rows (float64), then mu_j = mean,
// c_j = the power of two nearest to the column's standard deviation (1 for a constant column).  Valid for ANY mu and
// c (the filter's bound does not depend on how they were chosen); these make the int8 grid of a row cover what
// differs between rows.
// ---------------------------------------------------------------------------
template <bool F32>
__global__ __launch_bounds__(256) void shadow_colsum_kernel(const void* __restrict__ rows, int d, int64_t n_rows,
                                                           int64_t stride, int64_t n_take, double* __restrict__ sums) {
    // thread t owns columns t, t + 256, ... (d <= 1024); block b takes sample rows b, b + grid, ...
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    for (int64_t j = blockIdx.x; j < n_take; j += gridDim.x) {
        const int64_t row = j * stride;
        if (row >= n_rows) break;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = threadIdx.x + 256 * u;
            if (c < d) {
                const float x = F32 ? reinterpret_cast<const float*>(rows)[row * d + c]
                                    : (float)reinterpret_cast<const _Float16*>(rows)[row * d + c];
                s1[u] += (double)x;
                s2[u] = fma((double)x, (double)x, s2[u]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = threadIdx.x + 256 * u;
        if (c < d) {
            atomicAdd(sums + c, s1[u]);
            atomicAdd(sums + d + c, s2[u]);
        }
    }
}

__global__ __launch_bounds__(256) void shadow_affine_kernel(const double* __restrict__ sums, int d, double n, int identity,
                                                           float* __restrict__ aff) {
    for (int c = threadIdx.x; c < d; c += 256) {
        float mu = 0.f, sc = 1.f;
        if (identity != 1 && n > 0.0) {
            const double m = sums[c] / n;
            const double var = sums[d + c] / n - m * m;
            mu = (float)m;
            if (var > 0.0 && identity != 2) {        // (identity == 2: centre only, c = 1)
                int ex = (int)lrint(log2(sqrt(var)));
                ex = ex < -30 ? -30 : ex > 30 ? 30 : ex;
                sc = ldexpf(1.0f, ex);
            }
            if (!(fabsf(mu) < 3.0e38f)) mu = 0.f;      // (inf / nan rows: no centring on that column)
        }
        aff[c] = mu;
        aff[d + c] = sc;
        aff[2 * d + c] = 1.0f / sc;                    // exact: a power of two
    }
}

// ---------------------------------------------------------------------------
// query terms + per-query constants
// ---------------------------------------------------------------------------
// (ShadowQ and the quantisation itself live in flat_internal.h: prep_queries_kernel runs them for the wave that
// has the query in registers - one launch instead of two)
// ---------------------------------------------------------------------------
// the scan
// ---------------------------------------------------------------------------
// (Scan8Args and the kernel's body: scan8_body.h)
template <int QT, int KC, bool LISTS = true, int NCHS = 0, int ALN = 0, bool QUAD = false>
__global__ __launch_bounds__(512, 1) void scan8_kernel(Scan8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    scan8_body<QT, KC, LISTS, NCHS, ALN, QUAD>(a, smem, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------
// exact float64 scores of the candidates -> top k per query
// ---------------------------------------------------------------------------
// s_waitcnt immediate of the publish step below, gfx9 encoding (gfx90a / gfx942 / gfx950): vmcnt[3:0] | expcnt[6:4] |
// lgkmcnt[11:8] | vmcnt_hi[15:14] - vmcnt = 0, the other counters at their maxima (not waited for).  gfx10+ moved the
// fields: this file is gfx950 code (Makefile: --offload-arch=gfx950) and refuses to build for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "flat_shadow.hip hand-encodes gfx9 s_waitcnt immediates and relies on gfx950 sc1 write-through: build for gfx950 only"
#endif
constexpr int kWaitVmcnt0 = (0 << 14) | (0xF << 8) | (0x7 << 4) | 0x0;
static_assert(kWaitVmcnt0 == 0x0F70, "vmcnt(0) expcnt(7) lgkmcnt(15), gfx9 field layout");
constexpr int kShCap = 1024;
constexpr int kShThreads = 512;
constexpr int kShSplit = 16;     // workgroups per query (small LDS footprint: several per CU); measured in round 3,
                                 // same box: 2 / 4 / 8 / 16 / 32 slices -> 0.645 / 0.575 / 0.513 / 0.522 / 0.540 ms per
                                 // 64-query search at 2.6 M rows, 3.08 / 2.95 / 2.90 / 2.88 / 2.98 ms at 21 M
constexpr int kShIds = 4096;     // candidate ids one workgroup stages in LDS (after the final-bound filter)

struct ShTopK {
    unsigned long long key[kShCap];
    int id[kShCap];
    unsigned long long bound;
    int cnt;
};

__device__ __forceinline__ void sh_cut(ShTopK& t, int k) {
    __syncthreads();
    const int n = t.cnt;
    int n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    for (int i = n + threadIdx.x; i < n_pad; i += kShThreads) {
        t.key[i] = ~0ull;
        t.id[i] = 0x7fffffff;
    }
    for (int size = 2; size <= n_pad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int p = threadIdx.x; p < (n_pad >> 1); p += kShThreads) {
                const int i = ((p / stride) * 2 * stride) + (p % stride), j = i + stride;
                const unsigned long long ka = t.key[i], kb = t.key[j];
                const int ia = t.id[i], ib = t.id[j];
                const bool gt = ka > kb || (ka == kb && ia > ib);
                if (gt == ((i & size) == 0)) {
                    t.key[i] = kb;
                    t.key[j] = ka;
                    t.id[i] = ib;
                    t.id[j] = ia;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        t.cnt = n < k ? n : k;
        if (n >= k) t.bound = t.key[k - 1];
    }
    __syncthreads();
}

template <int CTRL>
__device__ __forceinline__ double sh_dpp_f64(double x) {
    return __longlong_as_double((long long)dpp_move_u64<CTRL>((unsigned long long)__double_as_longlong(x)));
}

struct GatherArgs {
    const void* rows;       // stored rows
    int d;
    int metric_l2;
    const float* q32;       // [B][d]
    const int2* cand;       // [n_wg][QT][cap] of the query tile this launch serves: (row id, key - a eps)
    const uint32_t* ccnt;   // [n_wg][QT]
    const uint32_t* g_tau;  // [QT] final bound of the scan (sortable): upper bound on the k-th best exact key
    int n_wg, QT, cap;
    int q0;                 // first query of the tile
    int k;
    unsigned long long* part_key;   // [B][kShSplit][k]
    int* part_id;
    uint32_t* done;         // [B] slices of the query that have published their list (zeroed by prep_queries_kernel)
    int64_t id_offset;
    float* D;               // [B][k] results (written by the last slice of every query)
    int64_t* I;
    uint32_t* ovf;          // [B] set when a region overflowed (or the id stage did)
    CertArgs cert;          // flag list for the exact fallback
    int dbg;                // timing experiments only (PRAG_SHADOW_DBG bits 32 / 64 / 128; results are WRONG)
    const double* kshift;   // [B] K_q = alpha q.mu: exact key - K_q is the scan's key space (null: 0)
    uint32_t* unfinished;   // one word: queries the bound kernel left to the sliced gather (statistics for the host)
    int no_gather;          // 1: no gather follows this launch - a query the bound kernel cannot finish is flagged
};

// Exact float64 score of one stored row against the staged query: 16 lanes per row (sub = lane & 15), every lane of
// the group returns the sum.  ONE definition for the bound kernel and the gather: both must form the same bits.
template <bool F32>
__device__ __forceinline__ double sh_exact_row(const GatherArgs& a, const float* s_q, int64_t row, bool have, int sub) {
    const int d = a.d;
    double s = 0.0;
    if (have) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {       // d <= 1024
            const int e = sub * 8 + it * 128;
            if (e < d) {
                float xv[8];
                if constexpr (F32) {
                    const float* p = reinterpret_cast<const float*>(a.rows) + row * d + e;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(p);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xv[j] = x0[j]; xv[4 + j] = x1[j]; }
                } else {
                    const half8 h = *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(a.rows) + row * d + e);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = (float)h[j];
                }
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(s_q + e);
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(s_q + e + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double qv = (double)(j < 4 ? q0[j] : q1[j - 4]);
                    const double x = (double)xv[j];
                    if (a.metric_l2) {
                        const double df = qv - x;
                        s = fma(df, df, s);
                    } else {
                        s = fma(qv, x, s);
                    }
                }
            }
        }
    }
    s += sh_dpp_f64<0xB1>(s);
    s += sh_dpp_f64<0x4E>(s);
    s += sh_dpp_f64<0x141>(s);
    s += sh_dpp_f64<0x140>(s);
    return s;
}

// ---------------------------------------------------------------------------
// An EXACT bound for the gather - and, when few rows are left under it, the end of the search in the same launch.
// The scan's own final bound is the maximum over 16 groups of workgroups of the group's best pessimistic key
// (key + a eps): about the 50th best row of the rows seen by the last epoch, two error terms above the k-th best exact
// key.  At the 8-GPU shard size ~2 200 candidates per query pass it (measured, 64 queries x 2.6 M rows: 220 MB of
// stored rows fetched by the gather, 39 of its 58 us; six dependent memory round trips in its publish-and-merge chain).
// Here ONE workgroup per query reads every candidate entry of the query once, keeps the most promising one per thread
// (smallest key - a eps), every 16 lanes score their best one exactly in float64 (the gather's arithmetic) and g_tau
// drops to the k-th best exact key of those 32 rows, moved into the scan's key space and rounded up: the k-th best of
// ANY k rows bounds the k-th best of all.  Nothing is removed that the round-3 argument kept: a row among the k best has
// key - a eps <= its exact key <= the k-th best exact key <= this bound.  The rows within ONE error term of that bound
// are tens, not thousands: up to kShFinish of them are scored right here and D / I written; the sliced gather that
// follows sees the query's `done` word taken and returns after its first load.  More survivors, a region that
// overflowed, or fewer than k rows scored: the sliced gather does the work, under the lower bound.
// ---------------------------------------------------------------------------
constexpr int kShFinish = 256;                 // survivors this kernel scores itself (steps of 32 rows)
constexpr uint32_t kShDoneTaken = 0x80000000u; // done[b]: the query was finished by shadow_bound_kernel
// LDS of one bound workgroup: the query (4 KiB), kShFinish keys and ids, four scalars
constexpr int kShBoundLds = 1024 * 4 + kShFinish * 8 + kShFinish * 4 + 32;
// The body as a device function of (query index inside the tile, LDS block): shadow_bound_kernel runs it alone;
// bound_gate_kernel runs it in the first nq workgroups of a launch whose other workgroups are the gate (below).
template <bool F32>
__device__ __forceinline__ void shadow_bound_body(const GatherArgs& a, uint32_t* __restrict__ g_tau_w, const int qi_,
                                                  char* const lds) {
    if (gate_closed(a.cert.gate)) return;
#ifdef PRAG_MM_DIAG
    unsigned long long stamp[10];
    int n_stamp = 0;
#define SH_STAMP() do { if (a.dbg & 512) stamp[n_stamp++] = wall_clock64(); } while (0)
#else
#define SH_STAMP() do {} while (0)
#endif
    SH_STAMP();
    float* const s_q = reinterpret_cast<float*>(lds);                                          // [1024]
    unsigned long long* const s_key = reinterpret_cast<unsigned long long*>(lds + 4096);       // [kShFinish]
    int* const s_fid = reinterpret_cast<int*>(lds + 4096 + kShFinish * 8);                     // [kShFinish]
    double& s_qn2 = *reinterpret_cast<double*>(lds + 4096 + kShFinish * 12);
    int& s_n = *reinterpret_cast<int*>(lds + 4096 + kShFinish * 12 + 8);
    int& s_over = *reinterpret_cast<int*>(lds + 4096 + kShFinish * 12 + 12);
    float& s_taux = *reinterpret_cast<float*>(lds + 4096 + kShFinish * 12 + 16);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int sub = lane & 15, slot = lane >> 4;
    const int qi = qi_, b = a.q0 + qi, d = a.d;
    const uint32_t tau_bits = a.g_tau[qi];
    const float tau_final = unsortable_f32(tau_bits);
    if (tid == 0) {
        s_n = 0;
        s_over = 0;
        s_taux = tau_final;
    }
    // ---- the most promising entry of every thread: two threads per candidate region ------------------------------------
    // The first 32 slots of the thread's region are fetched WITH its count (one round trip, not two; slots past the count
    // hold entries of earlier searches and are masked) and stay in registers for the second pass.
    constexpr int kKeep = 16;
    const int2 none = int2{0, 0x7fc00000};       // key = NaN: never <= a bound
    int2 ev0[kKeep];
    const int rg0 = tid >> 1, par = tid & 1;
    const bool spec = a.cap >= 2 * kKeep && rg0 < a.n_wg;
    uint32_t c_raw0 = 0;
    if (rg0 < a.n_wg) c_raw0 = a.ccnt[(int64_t)rg0 * a.QT + qi];
    {
        const int2* src = a.cand + ((int64_t)rg0 * a.QT + qi) * a.cap + par;
#pragma unroll
        for (int u = 0; u < kKeep; ++u) ev0[u] = spec ? src[2 * u] : none;
    }
    for (int c = tid; c < 1024; c += kShThreads) s_q[c] = c < d ? a.q32[(int64_t)b * d + c] : 0.f;
    bool over_l = c_raw0 > (uint32_t)a.cap;
    const int c0 = (int)min(c_raw0, (uint32_t)a.cap);
    float best = INFINITY;
    int best_id = 0x7fffffff;
    auto consider = [&](int2 e) {
        const float lo = __uint_as_float((uint32_t)e.y);
        if (lo <= tau_final && (lo < best || (lo == best && e.x < best_id))) {
            best = lo;
            best_id = e.x;
        }
    };
#pragma unroll
    for (int u = 0; u < kKeep; ++u) {
        if (par + 2 * u >= c0) ev0[u] = none;
        consider(ev0[u]);
    }
    const bool simple = spec && c0 <= 2 * kKeep && a.n_wg <= kShThreads / 2;
    // every entry the registers do not hold (long regions, more than 256 regions), eight loads in flight
    auto rest = [&](auto&& fn) {
        for (int rg = rg0; rg < a.n_wg; rg += kShThreads / 2) {
            const uint32_t c_raw = rg == rg0 ? c_raw0 : a.ccnt[(int64_t)rg * a.QT + qi];
            over_l |= c_raw > (uint32_t)a.cap;
            const int c = (int)min(c_raw, (uint32_t)a.cap);
            const int2* src = a.cand + ((int64_t)rg * a.QT + qi) * a.cap;
            for (int j0 = par + (rg == rg0 && spec ? 2 * kKeep : 0); j0 < c; j0 += 16) {
                int2 ev[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) ev[u] = j0 + 2 * u < c ? src[j0 + 2 * u] : none;
#pragma unroll
                for (int u = 0; u < 8; ++u) fn(ev[u]);
            }
        }
    };
    if (!simple) rest(consider);
    // ---- the best of every 16 lanes (DPP, no LDS): the group that found a row scores it --------------------------------
    // (32 rows from 32 disjoint parts of the candidate set: two of the k best in one part cost a place in the bound -
    // the 12th best instead of the 10th, a few more survivors below)
    const unsigned long long pick = group_min16_u64(best < INFINITY ? pack_key(best, best_id) : ~0ull);
    if (over_l) s_over = 1;
    __syncthreads();
    SH_STAMP();
    if (w == 1 && a.metric_l2) {   // ||q||^2 in float64 (L2 keys leave it out)
        double q2 = 0.0;
        for (int c = lane; c < d; c += 64) q2 = fma((double)s_q[c], (double)s_q[c], q2);
        for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o, 64);
        if (lane == 0) s_qn2 = q2;
    }
    {
        const int ci = w * 4 + slot;
        const bool have = pick != ~0ull;
        const int64_t row = have ? (int64_t)(uint32_t)pick : 0;
        const double s = sh_exact_row<F32>(a, s_q, row, have, sub);
        if (sub == 0) {
            s_key[ci] = have ? (a.metric_l2 ? sortable_u64(s) : ~sortable_u64(s)) : ~0ull;
            s_fid[ci] = have ? (int)row : 0x7fffffff;      // the first 32 entries of the final ranking
        }
    }
    __syncthreads();
    SH_STAMP();
    if (tid < 32) {
        const unsigned long long kv = s_key[tid];
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const unsigned long long kj = s_key[j];
            rank += (kj < kv) || (kj == kv && j < tid);
        }
        if (rank == a.k - 1 && kv != ~0ull) {
            const double sc = unsortable_f64(a.metric_l2 ? kv : ~kv);
            // the scan's key space: -score (inner product, cosine), ||x||^2 - 2 q.x = ||q - x||^2 - ||q||^2 (L2)
            // (minus K_q = alpha q.mu when the shadow is an affine image of the rows: see the file comment)
            const double kq = a.kshift ? a.kshift[b] : 0.0;
            const double t = (a.metric_l2 ? sc - s_qn2 : -sc) - kq;
            float tf = __double2float_ru(t);
            tf = fmaf(2.4e-7f, fabsf(tf) + (a.metric_l2 ? (float)s_qn2 : 0.f) + (float)fabs(kq), tf) + 1e-37f;   // upward only: never excludes
            const uint32_t v = sortable_u32(tf);
            if (v < tau_bits) {
                g_tau_w[qi] = v;
                s_taux = tf;
            }
        }
    }
    __syncthreads();
    // ---- second pass: what the exact bound still admits ----------------------------------------------------------------
    const float tau_x = s_taux;
    // (the row this thread's group has scored already is not admitted again: its exact key sits in s_key[0..31])
    const int picked = pick != ~0ull ? (int)(uint32_t)pick : -1;
    auto admit = [&](int2 e) {
        if (__uint_as_float((uint32_t)e.y) <= tau_x && e.x != picked) {
            const int at = 32 + atomicAdd(&s_n, 1);
            if (at < kShFinish) s_fid[at] = e.x;
        }
    };
#pragma unroll
    for (int u = 0; u < kKeep; ++u) admit(ev0[u]);
    if (!simple) rest(admit);
    for (int i = 32 + tid; i < kShFinish; i += kShThreads) s_key[i] = ~0ull;
    __syncthreads();
    SH_STAMP();
    const int n_fin = 32 + s_n;
    // finish here only when it is certain and small: no overflowed region, at most kShFinish survivors (fewer than k
    // is fine: the scan excluded every other row for good - the padding below is what the gather's merge writes)
    if (s_over || n_fin > kShFinish) {
        // left to the sliced gather - or, when the host enqueued none (it does not while recent searches never needed
        // it: shadow_search), to the retry tier / exact scan through the flag list
        if (tid == 0) {
            if (a.unfinished) atomicAdd(a.unfinished, 1u);
            if (a.no_gather && atomicExch(a.ovf + b, 1u) == 0u) cert_flag(a.cert, b);
        }
        return;
    }
    for (int i0 = 32; i0 < n_fin; i0 += 32) {
        const int ci = i0 + w * 4 + slot;
        const bool have = ci < n_fin;
        const int64_t row = have ? s_fid[ci] : 0;
        const double s = sh_exact_row<F32>(a, s_q, row, have, sub);
        if (sub == 0 && have) s_key[ci] = a.metric_l2 ? sortable_u64(s) : ~sortable_u64(s);
    }
    __syncthreads();
    SH_STAMP();
    // rank by counting, 16 lanes per entry (a serial loop over ~50 entries was 6 of the kernel's 20 us: two dependent LDS
    // reads per step); groups without a pick hold (worst key, no id) - what pads the gather's merge when fewer than k
    // rows exist (n_fin >= 32 >= k)
    for (int i0 = 0; i0 < n_fin; i0 += 32) {
        const int ci = i0 + w * 4 + slot;
        const bool have = ci < n_fin;
        const unsigned long long kv = have ? s_key[ci] : 0ull;
        const int iv = have ? s_fid[ci] : 0;
        int rank = 0;
        for (int j = sub; j < n_fin; j += 16) {
            const unsigned long long kj = s_key[j];
            const int ij = s_fid[j];
            rank += ((kj < kv) || (kj == kv && (ij < iv || (ij == iv && j < ci)))) ? 1 : 0;
        }
        rank += __builtin_amdgcn_update_dpp(0, rank, 0xB1, 0xF, 0xF, false);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x4E, 0xF, 0xF, false);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x141, 0xF, 0xF, false);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x140, 0xF, 0xF, false);
        if (have && sub == 0 && rank < a.k) {
            const bool ok = iv != 0x7fffffff;
            const double sc = ok ? unsortable_f64(a.metric_l2 ? kv : ~kv) : 0.0;
            a.D[(int64_t)b * a.k + rank] = ok ? (float)sc : (a.metric_l2 ? FLT_MAX : -FLT_MAX);
            a.I[(int64_t)b * a.k + rank] = ok ? tag_id((int64_t)iv + a.id_offset, sc, a.cert.tag_ids) : -1;
        }
    }
    if (tid == 0) a.done[b] = kShDoneTaken;
    SH_STAMP();
#ifdef PRAG_MM_DIAG
    if ((a.dbg & 512) && tid == 0)
        printf("[bound] query %d: %d ranked; x10 ns: entries+select %llu, 32 rows %llu, bound+filter %llu, survivors' rows %llu, rank+write %llu\n",
               qi, n_fin, stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4]);
#endif
#undef SH_STAMP
}


template <bool F32>
__global__ __launch_bounds__(kShThreads) void shadow_bound_kernel(GatherArgs a, uint32_t* __restrict__ g_tau_w) {
    __shared__ __attribute__((aligned(16))) char lds[kShBoundLds];
    shadow_bound_body<F32>(a, g_tau_w, (int)blockIdx.x, lds);
}

}  // namespace prag

// The gate BESIDE the bound kernel, in one launch (round 5): the first nq workgroups run shadow_bound_body, the others
// prober16_body - the fused prober ensemble over the NEXT batch of pooled states (exp_rag.py:406-415), which depends on
// nothing in the search.  What follows scan8 occupies 64 of 256 CUs for ~30 us; a second stream that waits for the
// scan's event hides only ~7 us of the gate's 40 behind it (the cross-stream dependency costs the rest), a shared
// launch all of it.  Both bodies are the ones their own kernels run (prober16.hip, shadow_bound_kernel above).
#define PSTAMP(i)
#define P16_ABL(bit) 0
#include "prober16_body.h"
#undef PSTAMP
#undef P16_ABL

namespace prag {

template <bool F32, int CT16>
__global__ __launch_bounds__(512, 2) void bound_gate_kernel(GatherArgs g, uint32_t* __restrict__ g_tau_w, int nq, ProberArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char fused_smem[];
    if ((int)blockIdx.x < nq) shadow_bound_body<F32>(g, g_tau_w, (int)blockIdx.x, fused_smem);
    else prober16_body<CT16>(pa, fused_smem, (int)blockIdx.x - nq);
}
static_assert(kShThreads == 512, "bound_gate_kernel: both bodies are written for 512-thread workgroups");

// ... and when the prober already ran behind the scan's workgroups (scan8_gate_kernel), the bound launch finishes the gate:
// workgroups nq.. run gate_kernel's body over the complete logits (exp_rag.py:407-415) - one launch less in the pass.
struct GateFinish {
    const float* logits;
    int L, B, ablation;
    double theta;
    float* probsum;
    int32_t* decision;
};
template <bool F32>
__global__ __launch_bounds__(kShThreads) void bound_finish_kernel(GatherArgs g, uint32_t* __restrict__ g_tau_w, int nq, GateFinish f) {
    extern __shared__ __attribute__((aligned(16))) char fused_smem[];
    if ((int)blockIdx.x < nq) shadow_bound_body<F32>(g, g_tau_w, (int)blockIdx.x, fused_smem);
    else gate_row(f.logits, f.L, f.B, f.ablation, f.theta, f.probsum, f.decision, ((int)blockIdx.x - nq) * kShThreads + (int)threadIdx.x);
}

template <bool F32>
__global__ __launch_bounds__(kShThreads) void shadow_gather_kernel(GatherArgs a) {
    __shared__ ShTopK tk;
    if (gate_closed(a.cert.gate)) return;
    __shared__ int s_ids[kShIds];
    __shared__ __attribute__((aligned(16))) float s_q[1024];
    __shared__ int s_n, s_over;
    __shared__ int s_rc[128];                 // candidates held by each region of this slice
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int sub = lane & 15, slot = lane >> 4;
    const int qi = blockIdx.y;                // query inside the tile
    const int b = a.q0 + qi;
    const int d = a.d;
    if (tid == 0) {
        s_n = 0;
        s_over = 0;
        tk.bound = ~0ull;
        tk.cnt = 0;
    }
    // ---- stage this slice's candidate ids: only those the FINAL bound of the scan still admits -----
    // (rows were collected against the bound of their time; key - a eps > final bound >= k-th best exact
    // key means the row is provably not among the k best)
    // One memory round trip brings the query, the final bound and the region counts of the slice, a second one
    // every filled slot of every region (instead of a count -> entries chain per region).
    const float tau_final = unsortable_f32(a.g_tau[qi]);
    const uint32_t taken = a.done[b];              // (uniform; read with the loads below in flight)
    const int nsplit = (int)gridDim.x;             // slices per query (<= kShSplit)
    const int per = (a.n_wg + nsplit - 1) / nsplit;
    const int wg0 = blockIdx.x * per, wg1 = min(a.n_wg, wg0 + per);
    const int nreg = wg1 - wg0;
    bool over_l = false;
    for (int i = tid; i < nreg; i += kShThreads) {
        const uint32_t c = a.ccnt[(int64_t)(wg0 + i) * a.QT + qi];
        over_l |= c > (uint32_t)a.cap;
        s_rc[i] = (int)min(c, (uint32_t)a.cap);
    }
    for (int c = tid; c < d; c += kShThreads) s_q[c] = a.q32[(int64_t)b * d + c];
    if (taken == kShDoneTaken) return;             // shadow_bound_kernel wrote this query's D / I already
    __syncthreads();
    if (over_l) s_over = 1;
    for (int base = 0; base < ((PRAG_SH_DBG(a.dbg) & 128) ? 0 : nreg * a.cap); base += 8 * kShThreads) {
        int2 ev[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {          // all loads of the batch in flight before the first use
            const int e0 = base + u * kShThreads + tid;
            const int rg = e0 / a.cap, j = e0 - rg * a.cap;
            const bool ok = e0 < nreg * a.cap && j < s_rc[rg < nreg ? rg : 0];
            ev[u] = ok ? a.cand[((int64_t)(wg0 + rg) * a.QT + qi) * a.cap + j] : int2{0, 0x7fc00000};   // key = NaN: never <= the bound
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (__uint_as_float((uint32_t)ev[u].y) <= tau_final) {
                const int at = atomicAdd(&s_n, 1);
                if (at < kShIds) s_ids[at] = ev[u].x;
                else s_over = 1;
            }
        }
    }
    __syncthreads();
    const int n_ids = (PRAG_SH_DBG(a.dbg) & 32) ? 0 : min(s_n, kShIds);
#ifdef PRAG_MM_DIAG
    if ((a.dbg & 256) && tid == 0 && (qi < 2 || (a.dbg & 512))) {
        int tot = 0;
        for (int i = 0; i < nreg; ++i) tot += s_rc[i];
        printf("[gather] query %d slice %d: %d regions hold %d entries (cap %d), %d pass the final bound\n", qi, (int)blockIdx.x, nreg, tot, a.cap, s_n);
    }
#endif
    if (s_over && tid == 0 && atomicExch(a.ovf + b, 1u) == 0u) cert_flag(a.cert, b);   // exact scan recomputes b
    // (Round 3: a second-level filter here - both int8 query terms over the candidate's 8-bit row before its stored
    // row is fetched, for the tiles whose scan used one term - was built and measured: search time unchanged at
    // 21 M rows (2.972 vs 2.973 ms) and at 2.6 M (0.544 vs 0.541).  The gather is a latency chain, not a byte count.)
    // ---- exact scores, 16 lanes per candidate row -------------------------------------------------
    // (45 of the kernel's 58 us at the 8-GPU shard size and 64 queries, tools/gather_probe.py: ~600
    // candidates per slice, one dependent row fetch per step of 32.  The rows in flight per CU are bounded
    // by registers - 4 workgroups x 32 rows x 1.5 KB now; 2 or 4 rows per lane group cost as many resident
    // workgroups as they add rows, tried and dropped)
    for (int i0 = 0; i0 < n_ids; i0 += 32) {
        const int ci = i0 + w * 4 + slot;
        const bool have = ci < n_ids;
        const int64_t row = have ? s_ids[ci] : 0;
        const double s = sh_exact_row<F32>(a, s_q, row, have, sub);
        if (sub == 0 && have) {
            const unsigned long long key = a.metric_l2 ? sortable_u64(s) : ~sortable_u64(s);
            if (key <= tk.bound) {
                const int sl = atomicAdd(&tk.cnt, 1);
                tk.key[sl] = key;
                tk.id[sl] = (int)row;
            }
        }
        if (((i0 >> 5) & 7) == 7) {     // every 8 steps (<= 256 pushes): room check
            __syncthreads();
            const int c = tk.cnt;
            __syncthreads();
            if (c > kShCap - 256) sh_cut(tk, a.k);
        }
    }
    if (!(PRAG_SH_DBG(a.dbg) & 64)) sh_cut(tk, a.k);
    // ---- publish this slice's list; the LAST slice of the query ranks the kShSplit lists and writes D / I --------
    // (one launch instead of gather + merge.  The lists cross XCDs - whose L2s are not coherent with each other -
    // as relaxed device-scope atomics: written through, read around the local L2.  A release / acquire fence pair
    // here would write back and invalidate the whole L2 of every one of the 1024 blocks: measured 524 -> 782 us
    // per search in round 2.)
    const int64_t o = ((int64_t)b * nsplit + blockIdx.x) * a.k;
    for (int j = tid; j < a.k; j += kShThreads) {
        const bool ok = j < tk.cnt;
        __hip_atomic_store(a.part_key + o + j, ok ? tk.key[j] : ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.part_id + o + j, ok ? tk.id[j] : 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(kWaitVmcnt0);   // vmcnt(0): the stores above have reached the coherence point
    __syncthreads();
    if (tid == 0)
        s_n = __hip_atomic_fetch_add(a.done + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)nsplit - 1 ? 1 : 0;
    __syncthreads();
    if (!s_n) return;
    // k <= 32: kShSplit * k <= 512 entries, ranked by counting
    unsigned long long* m_key = tk.key;        // (the block's own list is published: its LDS is free)
    int* m_id = tk.id;
    const int n = nsplit * a.k;
    __syncthreads();
    for (int i = tid; i < n; i += kShThreads) {
        m_key[i] = __hip_atomic_load(a.part_key + (int64_t)b * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        m_id[i] = __hip_atomic_load(a.part_id + (int64_t)b * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int i = tid; i < n; i += kShThreads) {
        const unsigned long long kv = m_key[i];
        const int iv = m_id[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const unsigned long long kj = m_key[j];
            const int ij = m_id[j];
            rank += (kj < kv) || (kj == kv && (ij < iv || (ij == iv && j < i)));
        }
        if (rank < a.k) {
            const bool ok = iv != 0x7fffffff;
            const double sc = ok ? unsortable_f64(a.metric_l2 ? kv : ~kv) : 0.0;
            a.D[(int64_t)b * a.k + rank] = ok ? (float)sc : (a.metric_l2 ? FLT_MAX : -FLT_MAX);
            a.I[(int64_t)b * a.k + rank] = ok ? tag_id((int64_t)iv + a.id_offset, sc, a.cert.tag_ids) : -1;
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// LDS of the scan: query planes (both terms for 32-query tiles), one 4-KiB stage and 384 B of row metadata per
// wave, bounds / counters
// (scan8_lds_bytes: scan8_body.h)

bool shadow_store_supported(int d) { return d % 128 == 0 && d <= 1024; }
// 128-query tiles: the query plane (one int8 term) + stages must fit LDS; lists up to 16 deep
bool shadow_tile128_ok(int d, int kc) {
    return (d == 768 || d == 512) && kc <= 16 && scan8_lds_bytes(128, (d + 255) / 256 * 256) <= 160 * 1024;
}
bool shadow_supported(int d, int kc, int k, int B) {
    return shadow_store_supported(d) && (kc == 8 || kc == 16 || kc == 32) && k <= 32 && B >= 1;
}

template <bool F32>
static void launch_build(int nch, dim3 grid, hipStream_t st, const ShadowStore& s, int64_t tile0, int64_t n_rows) {
#define PRAG_SB(N_) case N_: hipLaunchKernelGGL((shadow_build_kernel<F32, N_>), grid, dim3(256), 0, st, s.rows, tile0, \
                                                n_rows, s.rows8, s.sscale, s.serr, s.err_max, s.aff, s.yn_max, s.sbias, s.xnorm_l2, s.alpha, s.bias_max); break;
    switch (nch) { PRAG_SB(1) PRAG_SB(2) PRAG_SB(3) PRAG_SB(4) PRAG_SB(5) PRAG_SB(6) PRAG_SB(7) PRAG_SB(8) }
#undef PRAG_SB
}

// (re)builds whole 32-row tiles: the tile row0 falls into is rebuilt from its first row (same values as
// before for the rows already covered), rows past row1 inside the last tile become zeros
int shadow_build(const ShadowStore& s, int64_t row0, int64_t row1, hipStream_t st) {
    if (row1 <= row0) return PRAG_OK;
    PRAG_REQUIRE(shadow_store_supported(s.d), PRAG_EUNSUPPORTED, "internal: shadow of d=%d rows", s.d);
    const int64_t tile0 = row0 / 32, tile1 = (row1 + 31) / 32;
    const dim3 grid((unsigned)(tile1 - tile0));
    if (s.store_f32) launch_build<true>(s.d / 128, grid, st, s, tile0, row1);
    else launch_build<false>(s.d / 128, grid, st, s, tile0, row1);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

// mu / c / 1/c of the shadow's affine map from a sample of rows [0, n_rows) (at most 2^18 of them, evenly spread) ->
// aff [3][d]; `sums` is a [2][d] float64 scratch.  identity 1: mu = 0, c = 1 (PRAG_SHADOW_AFFINE=0); 2: c = 1, rows
// centred only (PRAG_SHADOW_AFFINE=1, the default); 0: centred + column scales (PRAG_SHADOW_AFFINE=2).
int shadow_affine_fit(const ShadowStore& s, int64_t n_rows, int identity, double* sums, hipStream_t st) {
    PRAG_REQUIRE(shadow_store_supported(s.d) && s.aff && sums, PRAG_EUNSUPPORTED, "internal: affine map of d=%d rows", s.d);
    const int64_t n_take = std::min<int64_t>(n_rows, (int64_t)1 << 18);
    const int64_t stride = n_take > 0 ? std::max<int64_t>(1, n_rows / n_take) : 1;
    PRAG_HIP(hipMemsetAsync(sums, 0, (size_t)2 * s.d * sizeof(double), st));
    if (identity != 1 && n_take > 0) {
        const dim3 grid((unsigned)std::min<int64_t>(n_take, 2048));
        if (s.store_f32)
            hipLaunchKernelGGL(shadow_colsum_kernel<true>, grid, dim3(256), 0, st, s.rows, s.d, n_rows, stride, n_take, sums);
        else
            hipLaunchKernelGGL(shadow_colsum_kernel<false>, grid, dim3(256), 0, st, s.rows, s.d, n_rows, stride, n_take, sums);
        PRAG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(shadow_affine_kernel, dim3(1), dim3(256), 0, st, sums, s.d, (double)n_take,
                       identity, const_cast<float*>(s.aff));
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

size_t shadow_slot_words() { return kShadowSlotWords; }
size_t shadow_q_bytes() { return sizeof(ShadowQ); }
int shadow_split() { return kShSplit; }

template <int QT, int KC, bool LISTS = true, int NCHS = 0, int ALN = 0, bool QUAD = false>
static int launch_scan8(const Scan8Args& a, int grid, hipStream_t st, EventRing& prof) {
    const int lds = scan8_lds_bytes(QT, a.qstride);
    auto kern = scan8_kernel<QT, KC, LISTS, NCHS, ALN, QUAD>;
    static LdsOptIn lds_opt_in;
    {
        const int rc_ = lds_opt_in.ensure(reinterpret_cast<const void*>(kern), 160 * 1024);
        if (rc_ != PRAG_OK) return rc_;
    }
    prof.begin(st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
    prof.end(st);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

// 64-query tiles, 16-deep lists, rows of a multiple of 384 elements, long shards: the quad-test epilogue with ONE copy
// of it in the loop (three staging sets, a tile = two rounds).  Measured on one box, scan8 alone / shard pass
// (profiles/r04t_scan8_quad_ab.txt): 21 M rows 2.68 -> 2.56 ms (0.76 -> 0.795 of 8 TB/s); 2.625 M rows the pass is
// 0.469 -> 0.481-0.498 ms - short scans spend their time in the early tiles, where most quads hold a candidate and the
// test is extra work - hence the row threshold.
template <bool F32, int CT16>
static int launch_bound_gate_ct(const TailGate& t, const GatherArgs& g, uint32_t* tau, int nq, hipStream_t st) {
    auto kern = bound_gate_kernel<F32, CT16>;
    const int lds = std::max(kShBoundLds, t.lds_bytes);
    static LdsOptIn lds_opt_in;
    const int rc_ = lds_opt_in.ensure(reinterpret_cast<const void*>(kern), 160 * 1024);
    if (rc_ != PRAG_OK) return rc_;
    hipLaunchKernelGGL(kern, dim3(nq + t.n_wg), dim3(kShThreads), lds, st, g, tau, nq, t.pa);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}
static int launch_bound_gate(bool f32, const TailGate& t, const GatherArgs& g, uint32_t* tau, int nq, hipStream_t st) {
#define PRAG_BG(CT_) case CT_: return f32 ? launch_bound_gate_ct<true, CT_>(t, g, tau, nq, st) : launch_bound_gate_ct<false, CT_>(t, g, tau, nq, st);
    switch (t.ct16) { PRAG_BG(2) PRAG_BG(4) PRAG_BG(8) }
#undef PRAG_BG
    set_error("internal: bound_gate_kernel has no %d-row tile", 16 * t.ct16);
    return PRAG_EUNSUPPORTED;
}

// (kScan8Aln: scan8_body.h)
int shadow_search(ShadowSearch& s, hipStream_t st, EventRing& prof) {
    const int qstride = (s.d + 255) / 256 * 256;
    const bool wide = s.qt_max >= 64 && s.B > 32 && scan8_lds_bytes(64, qstride) <= 160 * 1024;
    // 65..128 queries in ONE pass over the shadow (128-query tiles, list-less, lists up to 16 deep)
    const bool wide128 = s.qt_max >= 128 && s.B > 64 && shadow_tile128_ok(s.d, s.kc);
    const int QT = wide128 ? 128 : wide ? 64 : 32;
    // (the caller sizes the candidate store for the tile height: s.cap slots per (workgroup, query) either way -
    // with half the slots the 128-query tiles overflowed on the contiguous-cluster corpus: 3 fallbacks, 5.7 ms)
    const int cap = s.cap;
    const int Bpad = (s.B + QT - 1) / QT * QT;
    PRAG_REQUIRE(Bpad <= s.Bpad_ws, PRAG_EUNSUPPORTED, "internal: shadow workspace too small");
    const int n_tiles = (int)((s.N + 31) / 32);
    const int wg_cap = shadow_scan_wg_cap(s.max_wg, s.auto_wg, QT, s.every_cu);   // (7/8 of the CUs for HBM-bound tiles, a prime count: flat_internal.h)
    const int grid = std::max(1, std::min(wg_cap, (n_tiles + 7) / 8));
    PRAG_REQUIRE(grid <= s.wg_slots, PRAG_EUNSUPPORTED, "internal: shadow candidate regions too few");
    s.grid_used = grid;
    const int nsplit = kShSplit;
    PRAG_REQUIRE(grid <= 128 * nsplit, PRAG_EUNSUPPORTED, "internal: more scan workgroups than the gather's slices hold");
    // (query terms, per-query constants, bound slots and overflow words were written by prep_queries_kernel)
    for (int p0 = 0; p0 < Bpad; p0 += QT) {
        Scan8Args a;
        a.kslots = std::max(1, std::min(s.k, s.kc));
        a.rows8 = s.store.rows8;
        a.sscale = s.store.sscale;
        a.serr = s.store.serr;
        a.sbias = s.store.sbias;
        a.q8a = s.q8a + (size_t)p0 * s.d;
        a.q8b = s.q8b + (size_t)p0 * s.d;
        a.sq = reinterpret_cast<const ShadowQ*>(s.sq) + p0;
        a.N = s.N;
        a.d = s.d;
        a.qstride = qstride;
        a.n_tiles = n_tiles;
        a.g_tau = s.g_tau + p0;
        a.g_slot = s.slots + (size_t)p0 * kShadowSlotWords;
        a.cand = reinterpret_cast<int2*>(s.cand);
        a.ccnt = s.ccnt;
        a.cap = cap;
        a.gate = s.gate;
#ifdef PRAG_MM_DIAG
        static const int dbg_env = getenv("PRAG_SHADOW_DBG") ? atoi(getenv("PRAG_SHADOW_DBG")) : 0;
        a.dbg = dbg_env;
        static const int stamps_env = getenv("PRAG_SCAN8_STAMPS") ? atoi(getenv("PRAG_SCAN8_STAMPS")) : 0;
        static unsigned long long* stamps_dev = nullptr;
        a.stamps = nullptr;
        if (stamps_env) {
            if (!stamps_dev) PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&stamps_dev), (size_t)1024 * 8 * kScan8Stamps * 8));
            PRAG_HIP(hipMemsetAsync(stamps_dev, 0, (size_t)1024 * 8 * kScan8Stamps * 8, st));
            a.stamps = stamps_dev;
        }
#else
        a.dbg = 0;
#endif
        int rc;
        const bool timed = s.time_ev0 && s.time_ev1 && p0 + QT >= Bpad;
        if (timed) PRAG_HIP(hipEventRecord(s.time_ev0, st));
        const bool quad16 = (s.d / 128) % kScan8Aln == 0 && s.N >= s.quad_min_rows;
        const bool quad32 = s.N >= s.quad_min_rows;
        // The gate of the next batch behind the scan's workgroups, in the scan's launch (flat_scan_gate.hip), when the
        // CUs the scan leaves free get through the prober's workgroups while it runs: rounds x ~95 / 60 / 40 us per
        // 128 / 64 / 32-row tile against the scan's bytes at 6 TB/s.  Otherwise the tail carries it (bound_gate_kernel).
        bool scan_gate = false;
        if (QT == 64 && s.tail && !s.tail->taken && p0 + QT >= Bpad && s.scan_gate_mode != 0 &&
            scan8_gate_supported(s.kc, s.tail->ct16) && s.tail->lds_bytes <= 160 * 1024) {
            const int free_cu = std::max(1, s.wg_slots - grid);
            const double gate_us = (double)((s.tail->n_wg + free_cu - 1) / free_cu) * (s.tail->ct16 == 8 ? 95.0 : s.tail->ct16 == 4 ? 60.0 : 40.0);
            const double scan_us = (double)s.N * (s.d + 12) / 6.0e6;
            scan_gate = s.scan_gate_mode > 0 || (grid < s.wg_slots && gate_us <= 0.9 * scan_us);
            // (a launch that is being timed for the workgroup count carries the scan alone - with the prober's
            //  workgroups in it the 7/8 arm would be charged the gate: the tail carries it in those eight searches)
            if (timed && s.scan_gate_mode <= 0) scan_gate = false;
        }
#ifndef PRAG_S8_Q128_NLD
#define PRAG_S8_Q128_NLD 0      // (0: three staging sets for 768-element rows; A/B builds: 6 = a whole tile in flight)
#endif
#ifndef PRAG_S8_Q32_ALN
#define PRAG_S8_Q32_ALN 0       // (0: four sets, run-time chunk loop; A/B builds: 6 with 768-element rows)
#endif
        const bool q32_aln = PRAG_S8_Q32_ALN > 0 && s.d == 128 * (PRAG_S8_Q32_ALN > 0 ? PRAG_S8_Q32_ALN : 1);
        if (QT == 128)
            rc = s.d == 768 ? (s.kc == 8 ? launch_scan8<128, 8, false, 6, PRAG_S8_Q128_NLD, true>(a, grid, st, prof)
                                         : launch_scan8<128, 16, false, 6, PRAG_S8_Q128_NLD, true>(a, grid, st, prof))
                            : (s.kc == 8 ? launch_scan8<128, 8, false, 4, 0, true>(a, grid, st, prof)
                                         : launch_scan8<128, 16, false, 4, 0, true>(a, grid, st, prof));
        else if (QT == 64 && scan_gate)
            rc = launch_scan8_gate(a, grid, s.kc, quad16, *s.tail, st, prof);
        else if (QT == 64)
            rc = s.kc == 8 ? (quad16 ? launch_scan8<64, 8, true, 0, kScan8Aln, true>(a, grid, st, prof) : launch_scan8<64, 8>(a, grid, st, prof))
                 : s.kc == 16 ? (quad16 ? launch_scan8<64, 16, true, 0, kScan8Aln, true>(a, grid, st, prof)
                                        : launch_scan8<64, 16>(a, grid, st, prof))
                              : launch_scan8<64, 32>(a, grid, st, prof);
        else
            rc = quad32 ? (s.kc == 8 ? launch_scan8<32, 8, true, 0, 0, true>(a, grid, st, prof)
                           : s.kc == 16 ? (q32_aln ? launch_scan8<32, 16, true, 0, PRAG_S8_Q32_ALN, true>(a, grid, st, prof)
                                                   : launch_scan8<32, 16, true, 0, 0, true>(a, grid, st, prof))
                                        : launch_scan8<32, 32, true, 0, 0, true>(a, grid, st, prof))
                        : (s.kc == 8 ? launch_scan8<32, 8>(a, grid, st, prof)
                           : s.kc == 16 ? launch_scan8<32, 16>(a, grid, st, prof) : launch_scan8<32, 32>(a, grid, st, prof));
        if (rc != PRAG_OK) return rc;
        if (timed) {
            PRAG_HIP(hipEventRecord(s.time_ev1, st));
            s.timed_recorded = true;
        }
        if (scan_gate) s.tail->taken = s.tail->in_scan = true;
        if (s.scan_done && p0 + QT >= Bpad) PRAG_HIP(hipEventRecord(s.scan_done, st));
#ifdef PRAG_MM_DIAG
        {
            static int stamp_calls = 0;
            if (a.stamps && ++stamp_calls % stamps_env == 0) {      // PRAG_SCAN8_STAMPS=n: every n-th search prints
                PRAG_HIP(hipStreamSynchronize(st));
                std::vector<unsigned long long> h((size_t)grid * 8 * kScan8Stamps);
                PRAG_HIP(hipMemcpy(h.data(), a.stamps, h.size() * 8, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull;
                for (int i = 0; i < grid * 8; ++i)
                    if (h[(size_t)i * kScan8Stamps + 14]) t0 = std::min(t0, h[(size_t)i * kScan8Stamps]);
                const char* names[13] = {"entry", "prologue", "tile1", "tile2", "tile4", "tile8", "tile16", "tile32", "tile64",
                                         "tile128", "last_first_visit", "loop_end", "exit"};
                fprintf(stderr, "[scan8 stamps] grid %d QT %d rows %lld: us since the first wave's entry (min / median / max over waves)\n",
                        grid, QT, (long long)s.N);
                for (int k_ = 0; k_ < 13; ++k_) {
                    std::vector<double> v;
                    for (int i = 0; i < grid * 8; ++i) {
                        const unsigned long long x = h[(size_t)i * kScan8Stamps + k_];
                        if (x && h[(size_t)i * kScan8Stamps + 14]) v.push_back((double)(x - t0) / 100.0);   // 100 MHz wall clock
                    }
                    if (v.empty()) continue;
                    std::sort(v.begin(), v.end());
                    fprintf(stderr, "  %-17s %8.1f %8.1f %8.1f   (%zu waves)\n", names[k_], v.front(), v[v.size() / 2], v.back(), v.size());
                }
                double redo = 0, nmy = 0, cand = 0;
                for (int i = 0; i < grid * 8; ++i) {
                    redo += (double)h[(size_t)i * kScan8Stamps + 13];
                    nmy += (double)h[(size_t)i * kScan8Stamps + 14];
                    if (i % 8 == 0) cand += (double)h[(size_t)i * kScan8Stamps + 15];
                }
                fprintf(stderr, "  tiles %.0f, second visits %.0f (%.1f %%), candidates appended %.0f (%.0f per query)\n", nmy, redo,
                        100.0 * redo / std::max(1.0, nmy), cand, cand / QT);
            }
        }
#endif
        const int nq = std::min(QT, s.B - p0);
        GatherArgs g;
        g.rows = s.store.rows;
        g.d = s.d;
        g.metric_l2 = s.metric_l2;
        g.q32 = s.q32;
        g.cand = reinterpret_cast<const int2*>(s.cand);
        g.ccnt = s.ccnt;
        g.g_tau = s.g_tau + p0;
        g.n_wg = grid;
        g.QT = QT;
        g.cap = cap;
        g.q0 = p0;
        g.k = s.k;
        g.part_key = s.part_key;
        g.part_id = s.part_id;
        g.done = s.done;
        g.id_offset = s.id_offset;
        g.D = s.D;
        g.I = s.I;
        g.ovf = s.ovf;
        g.cert = s.cert;
        g.cert.gate = s.gate;
        g.dbg = a.dbg;
        g.kshift = s.kshift;
        g.unfinished = s.unfinished;
        const bool skip_gather = s.exact_bound && s.skip_gather;
        g.no_gather = skip_gather ? 1 : 0;
        if (s.exact_bound) {     // (k <= 32: the kernel scores 32 rows)
            TailGate* tg = s.tail && !s.tail->taken && p0 + QT >= Bpad ? s.tail : nullptr;
            TailGate* tf = s.tail && s.tail->in_scan && !s.tail->finished && !s.tail->gate_folded && p0 + QT >= Bpad ? s.tail : nullptr;
            if (tg) {            // the gate of the next batch in the same launch (bound_gate_kernel)
                const int rc_t = launch_bound_gate(s.store.store_f32 != 0, *tg, g, s.g_tau + p0, nq, st);
                if (rc_t != PRAG_OK) return rc_t;
                tg->taken = true;
            } else if (tf) {     // the prober ran in the scan's launch: this one finishes the gate (bound_finish_kernel)
                const GateFinish f{tf->pa.logits, tf->pa.n_run, tf->pa.B, tf->fin_ablation, tf->fin_theta, tf->fin_probsum, tf->fin_decision};
                const int n_fin = (tf->pa.B + kShThreads - 1) / kShThreads;
                if (s.store.store_f32)
                    hipLaunchKernelGGL(bound_finish_kernel<true>, dim3(nq + n_fin), dim3(kShThreads), kShBoundLds, st, g, s.g_tau + p0, nq, f);
                else
                    hipLaunchKernelGGL(bound_finish_kernel<false>, dim3(nq + n_fin), dim3(kShThreads), kShBoundLds, st, g, s.g_tau + p0, nq, f);
                PRAG_LAUNCH_CHECK();
                tf->finished = true;
            } else {
                if (s.store.store_f32)
                    hipLaunchKernelGGL(shadow_bound_kernel<true>, dim3(nq), dim3(kShThreads), 0, st, g, s.g_tau + p0);
                else
                    hipLaunchKernelGGL(shadow_bound_kernel<false>, dim3(nq), dim3(kShThreads), 0, st, g, s.g_tau + p0);
                PRAG_LAUNCH_CHECK();
            }
        }
        if (!skip_gather) {
            if (s.store.store_f32)
                hipLaunchKernelGGL(shadow_gather_kernel<true>, dim3(nsplit, nq), dim3(kShThreads), 0, st, g);
            else
                hipLaunchKernelGGL(shadow_gather_kernel<false>, dim3(nsplit, nq), dim3(kShThreads), 0, st, g);
            PRAG_LAUNCH_CHECK();
        }
    }
    return PRAG_OK;
}

}  // namespace prag
