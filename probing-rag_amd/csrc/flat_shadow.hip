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
#include "tail_gate.h"

namespace prag {

typedef int i32x16 __attribute__((ext_vector_type(16)));



// Timing experiments that give WRONG results (warm-up off, nothing collected, gather stages off) exist only
// in the `make diag` build (libprag_diag.so, -DPRAG_MM_DIAG); in libprag.so the knob is the constant 0 and
// the environment is never read.
#ifdef PRAG_MM_DIAG
#define PRAG_SH_DBG(x) (x)
#else
#define PRAG_SH_DBG(x) 0
#endif

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
struct Scan8Args {
    int kslots;                 // bound slots in use per epoch = k (<= KC): see the slot comment in scan8_kernel
    const signed char* rows8;   // [N/32][d/128][4 KiB]: chunk-major inside 32-row tiles, pieces at shadow_piece_off
    const float* sscale;        // [roundup(N,32)]
    const float* serr;
    const float* sbias;         // [roundup(N,32)] per-row additive part of the key: alpha mu.(x_i - mu) [+ ||x_i||^2 for L2]
    const signed char* q8a;     // [QT][d] this pass's query tile
    const signed char* q8b;
    const ShadowQ* sq;          // [QT]
    int64_t N;
    int d;
    int qstride;                // LDS bytes per query row (multiple of 256)
    int n_tiles;                // ceil(N / 32)
    uint32_t* g_tau;            // [QT] chip-wide bound (sortable), +inf at start, -inf for padding
    uint32_t* g_slot;           // [QT][kShadowEpochs + 1][32]
    int2* cand;                 // [grid][QT][cap]  (row id, bits of key - a eps)
    uint32_t* ccnt;             // [grid][QT]   (> cap: the region overflowed, the query goes to the exact scan)
    int cap;

    int dbg;                    // timing experiments only (PRAG_SHADOW_DBG; results are WRONG): bit 0 no warm-up
                                // (no second visits), bit 1 nothing is collected
    Gate gate;
#ifdef PRAG_MM_DIAG
    unsigned long long* stamps; // [grid][8 waves][kScan8Stamps] wall-clock stamps (PRAG_SCAN8_STAMPS=1, `make diag` only)
#endif
};
// stamps of one wave: 0 kernel entry, 1 prologue done (queries in LDS, first loads issued), 2..9 after its tile 1, 2, 4,
// 8, 16, 32, 64, 128, 10 after its last first-visit tile, 11 after the second visits, 12 kernel exit, 13 = redo,
// 14 = n_my, 15 = candidates appended by the workgroup (wave 0)
[[maybe_unused]] constexpr int kScan8Stamps = 16;

// (kShadowEpochs = 9 in flat_internal.h: bound slots refreshed after tiles 1, 2, 4, ..., 256; per query
// kShadowEpochs + 1 rows of 32 words, the last one the sample slots written by prep_queries_kernel)
constexpr int kShadowSlotRows = kShadowEpochs + 1;

template <int KC>
struct KeyList {   // sorted KC smallest keys (no ids: the lists only feed the bound)
    float k[KC];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < KC; ++j) k[j] = INFINITY;
    }
    __device__ __forceinline__ void push(float key, float tau) {
        if (key < k[KC - 1] && key <= tau) {
            float prev = -INFINITY;
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                const float cur = k[j];
                k[j] = __builtin_amdgcn_fmed3f(prev, cur, key);
                prev = cur;
            }
        }
    }
};

// 64-query tiles run the FIRST int8 query term only (32-query tiles both).  Both terms at 64 queries were built
// in round 3 (32 more accumulator registers -> three staging buffers instead of five): it paid on shards of a
// few million rows while the scan opened with a warm-up (0.50 vs 0.58 ms per search at 2.6 M rows); with the
// sampled pre-bound the single term wins at every size (2.6 M rows: 0.477 vs 0.535 ms, 21 M: 2.8 vs 3.6).
//
// Tried in round 3 and dropped (same box, alternating runs): the row length as a template parameter with the
// chunks of a tile unrolled - six staging buffers refilled at a fixed distance, COUNTED vmcnt waits in every
// chunk instead of this loop's drain at the head of every batch of NLD chunks (2.87 vs 2.80 ms at 21 M rows,
// 0.468 vs 0.418 ms at 2.6 M), and the same with the whole next tile requested in one 24-KiB burst (2.88 ms).
// The prefetch structure is not what limits this loop.
// LISTS false: no per-lane lists - a lane keeps only the best key_hi it has seen per query (what the bound slots
// are fed with) and the bound comes from the slot epochs alone.  128-query tiles need it (4 x 16 list registers
// on top of 64 accumulator registers do not fit); at 64 queries it was measured and is no faster (21 M rows:
// 2.82 - 2.89 vs 2.81 - 2.85 ms; 2.6 M rows: 0.449 - 0.460 vs 0.425 - 0.441).
// NCHS > 0 (128-query tiles, d = 128 NCHS): the chunks of a tile are unrolled with ONE copy of the epilogue behind
// them instead of one per staging buffer - the only form in which four query columns per lane fit the register
// file.  (For 64-query tiles this loop form measured 2-3 % slower than the run-time one, see above.)
template <int QT, int KC, bool LISTS = true, int NCHS = 0, int ALN = 0, bool QUAD = false>
__global__ __launch_bounds__(512, 1) void scan8_kernel(Scan8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gate_closed(a.gate)) return;
#ifdef PRAG_MM_DIAG
    const unsigned long long stamp_entry = a.stamps ? wall_clock64() : 0ull;
#define S8_STAMP(i) do { if (a.stamps && (threadIdx.x & 63) == 0) \
        a.stamps[((int64_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * kScan8Stamps + (i)] = wall_clock64(); } while (0)
#define S8_VALUE(i, v) do { if (a.stamps && (threadIdx.x & 63) == 0) \
        a.stamps[((int64_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * kScan8Stamps + (i)] = (unsigned long long)(v); } while (0)
#else
#define S8_STAMP(i) do {} while (0)
#define S8_VALUE(i, v) do {} while (0)
#endif
    constexpr int NQ = QT / 32;
    // 128-query tiles, epilogue order.  Rounds 3-4: query column outermost, its three constants read from LDS per
    // column (the row-major order kept 16 constants + the row metadata live beside the staging loop's registers and
    // spilled).  Without the LDS staging there is room: -DPRAG_S8_Q128_ROW_EPI=1 takes the 64-query tiles' order
    // (row quad outermost: the quad's metadata is read once, not once per column).
#ifndef PRAG_S8_IMAX_TEST
#define PRAG_S8_IMAX_TEST 1     // (0 in A/B builds: the 64-query quad test on converted keys, as in round 4)
#endif
#ifndef PRAG_S8_Q128_ROW_EPI
#define PRAG_S8_Q128_ROW_EPI 0
#endif
    constexpr bool COL_EPI = QT == 128 && !PRAG_S8_Q128_ROW_EPI;
    // 32-query tiles carry the query as two int8 terms (the HBM-bound loop has matrix-pipe slack for the
    // second MFMA); 64-query tiles use the first term only and pay with a wider eps (more candidates)
    constexpr int TERMS = QT == 32 ? 2 : 1;
    static_assert(QT == 32 || QT == 64 || QT == 128, "query tile");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int d = NCHS > 0 ? NCHS * 128 : a.d;
    const int NCH = d >> 7;                       // 128-byte chunks per row
    const int qstride = a.qstride;
    char* s_qa = smem;
    char* s_qb = smem + QT * qstride;           // (second term: 32-query tiles only)
    const int planes_b = TERMS * QT * qstride;
    char* s_st = smem + planes_b + w * 4096;
    float* s_meta = reinterpret_cast<float*>(smem + planes_b + 8 * 4096) + w * 96;   // [3][32] per wave
    uint32_t* s_tau = reinterpret_cast<uint32_t*>(smem + planes_b + 8 * 4096 + 8 * 384);
    uint32_t* s_best = s_tau + QT;
    uint32_t* s_ccnt = s_best + QT;
    uint32_t* s_slot_ok = s_ccnt + QT;     // set once a poll found every query's slot bound finite
    uint32_t* s_arrive = s_slot_ok + 1;    // waves that have fed epoch 0 (kShadowEpochs words)
    uint32_t* s_pre = s_arrive + kShadowEpochs + 1;   // [QT] the sampled pre-bound of every query
    float* s_sqc = reinterpret_cast<float*>(s_pre + QT + 5);   // [QT][3] kscale, A, C (128-query tiles only)
    if (tid <= kShadowEpochs) s_slot_ok[tid] = 0u;   // the flag and the arrival counters behind it
    // ---- the bound the search starts with: the k-th smallest of the 32 sample slots prep_queries_kernel filled.
    // The slices are disjoint row sets, so k rows have exact keys at or below it.  (The MAX over the slots
    // is valid too but has a bad tail - one slice without a good row loosens the bound of that query - and a loose
    // start floods the candidate regions of the first tiles: one query in a few searches overflowed a region.)
    {
        uint32_t* s_ps = reinterpret_cast<uint32_t*>(smem + planes_b);       // [QT][32] in the (still unused) stages
        for (int i = tid; i < QT * 32; i += 512)
            s_ps[i] = a.g_slot[((i >> 5) * kShadowSlotRows + kShadowPreEpoch) * 32 + (i & 31)];
        __syncthreads();
        for (int tt = tid; tt < QT * 8; tt += 512) {   // 8 threads per query, 4 slots each: rank by counting
            const int q = tt >> 3;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = (tt & 7) * 4 + u;
                const uint32_t v = s_ps[q * 32 + i];
                int rank = 0;
                for (int j = 0; j < 32; ++j) {
                    const uint32_t x = s_ps[q * 32 + j];
                    rank += (x < v || (x == v && j < i)) ? 1 : 0;
                }
                if (rank == a.kslots - 1) s_pre[q] = v;
            }
        }
        __syncthreads();
    }
    if constexpr (COL_EPI) {
        if (tid < QT) {
            const ShadowQ sq_ = a.sq[tid];
            s_sqc[3 * tid] = sq_.kscale;
            s_sqc[3 * tid + 1] = sq_.A1;
            s_sqc[3 * tid + 2] = sq_.C1;
        }
    }
    if (tid < 64) {      // wave 0 (QT / 64 queries per lane)
        bool missing = false;
        for (int qq = tid; qq < QT; qq += 64) {
            const uint32_t t0 = a.g_tau[qq];
            const uint32_t m = s_pre[qq];
            missing |= m == kSortablePosInf && t0 != kSortableNegInf;   // (padding queries do not count)
            s_tau[qq] = m < t0 ? m : t0;
            s_best[qq] = 0xFFFFFFFFu;
            s_ccnt[qq] = 0u;
        }
        // every query has a finite bound: no warm-up (nothing to visit twice)
        if (__builtin_amdgcn_ballot_w64(missing) == 0 && tid == 0) *s_slot_ok = 1u;
    }
    // ---- query tiles -> LDS (swizzled 16-B pieces, as the fp16 scan) -----------------------------
    {
        const int ppr = d >> 4;
        const int total = QT * ppr;
#pragma unroll
        for (int plane = 0; plane < TERMS; ++plane) {
            const signed char* qsrc = plane == 0 ? a.q8a : a.q8b;
            char* qdst = plane == 0 ? s_qa : s_qb;
            for (int e0 = tid; e0 < total; e0 += 512 * 4) {
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * 512;
                    const int ec = e < total ? e : total - 1;
                    const int row = ec / ppr, pc = ec - row * ppr;
                    v[u] = *reinterpret_cast<const u32x4*>(qsrc + (int64_t)row * d + 16 * pc);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * 512;
                    if (e < total) {
                        const int row = e / ppr, pc = e - row * ppr;
                        *reinterpret_cast<u32x4*>(qdst + row * qstride + (((pc & ~15) | ((pc ^ row) & 15)) << 4)) = v[u];
                    }
                }
            }
        }
    }
    __syncthreads();

    const int nW = gridDim.x * 8;
    // tile t belongs to workgroup t mod grid (wave (t / grid) mod 8): CONSECUTIVE tiles go to different
    // workgroups, so a run of similar rows (a corpus in article order: 1024 contiguous near-duplicates are 32
    // tiles) spreads over 32 candidate regions instead of filling the regions of 4 workgroups (measured on such
    // a corpus with the wave-major map: all 64 queries overflowed into the exact scan, 104 ms per search
    // against 1.1 ms for the direct scan)
    const int gw = w * (int)gridDim.x + (int)blockIdx.x;
    const int n_my = gw < a.n_tiles ? (a.n_tiles - gw + nW - 1) / nW : 0;

    KeyList<LISTS ? KC : 1> top[NQ];    // (LISTS false: k[0] = the lane's best key)
#ifndef PRAG_S8_Q128_REGCONST
#define PRAG_S8_Q128_REGCONST 1     // (the 128-query tiles' per-column constants in registers for 768-element rows: 2.98 -> 2.96 ms; 0 in A/B builds)
#endif
    constexpr bool REG_CONST = !COL_EPI || (PRAG_S8_Q128_REGCONST && NCHS == 6);
    float kscale[REG_CONST ? NQ : 1], cA[REG_CONST ? NQ : 1], cC[REG_CONST ? NQ : 1];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        top[t].init();
        if constexpr (REG_CONST) {
            const ShadowQ s = a.sq[32 * t + r];
            kscale[t] = s.kscale;
            cA[t] = TERMS == 2 ? s.A2 : s.A1;
            cC[t] = TERMS == 2 ? s.C2 : s.C1;
        }
    }

    // staging geometry: 4 x 16-B loads per lane per 128-byte chunk of 32 rows
    int st_doc[4], st_dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int doc = 8 * i + (lane >> 3), q = lane & 7;
        st_doc[i] = doc;
        st_dst[i] = doc * 128 + ((q ^ ((doc >> 1) & 7)) << 4);
    }
    const int64_t row_bytes = d;
    const char* rows = reinterpret_cast<const char*>(a.rows8);
    // chunks in flight per wave (4 KiB each): four for 32-query tiles, five for 64-query tiles (20 KiB per
    // wave, 160 KiB per CU) - with the chunk-major row layout the fifth is worth 2-3 % at 64 queries (2.875 ->
    // 2.827 ms at 21 M rows, same box, alternating runs) and nothing at 32 or 1; with two the loop starved.
    // 64 queries x 32-deep lists have no registers beyond three.
    constexpr int NLD = ALN > 0 ? ALN :
                        NCHS > 0 ? (QUAD ? (NCHS % 3 == 0 ? 3 : 4) : 2) :   // (128-query tiles; before the quad test freed registers: two)
                         QT == 64 ? (KC == 32 ? 3 : 5) : 4;   // (six at 64 queries fit - 253 VGPRs - and are slower: 2.92 vs 2.89 ms)
    u32x4 ld[NLD][4];
    // wave-uniform tile base (scalar registers) + a per-lane 32-bit offset: no 64-bit vector address
    // arithmetic and no per-row clamp in the loop - the shadow is allocated in multiples of 256 rows, so
    // the rows of the last, partial tile past N are readable (their scores are masked in the epilogue)
    // (lane -> byte (8 i + lane / 8) * 128 + (lane % 8) * 16 = 1024 i + 16 lane of the chunk: one unsigned 32-bit lane
    // offset on a scalar base, the four pieces at immediate offsets)
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto issue = [&](u32x4 (&ldr)[4], int tile, int c) __attribute__((always_inline)) {
        const char* base = rows + (int64_t)tile * (32 * row_bytes) + c * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // read-once stream: non-temporal loads leave L2 / the Infinity Cache to what IS reused between launches
            // (in a retrieve-decide pass: the gate's 22 MB of weights and states).  Same box, plain / nt loads
            // (profiles/r04h_scan8_nt_loads_ab.txt): 2.625 M-row shard 0.400-0.426 -> 0.362-0.390 ms and the gate that
            // follows it 55.8 -> 40.1 us; 21 M rows 2.78 -> 2.69 ms (0.733 -> 0.757 of 8 TB/s).
#ifdef PRAG_SCAN_PLAIN_LOADS
            ldr[i] = *reinterpret_cast<const u32x4*>(base + lane16 + 1024 * i);
#else
            ldr[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + lane16 + 1024 * i));
#endif
        }
    };
    const int a_off = r * 128;
    const int a_sw = (r >> 1) & 7;
    const int xq0 = (r ^ hh) & 15;                  // (32 t does not reach the low four bits)
    int q_base[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) q_base[t] = (32 * t + r) * qstride;

    i32x16 acc1[NQ], acc2[TERMS == 2 ? NQ : 1];
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc1[t][e] = 0;
            if (TERMS == 2) acc2[t][e] = 0;
        }

    // virtual tile sequence of this wave: its n_my tiles, then the first `redo` of them again
    constexpr int kWarmMax = 16;
    int redo = 0;
    bool warm = !(PRAG_SH_DBG(a.dbg) & 1) && *s_slot_ok == 0u;   // (read after the barrier below the query tiles)
    auto vtile = [&](int vt) __attribute__((always_inline)) {
        int at = vt < n_my ? vt : vt - n_my;
        at = at < n_my ? at : n_my - 1;             // prefetch past the end: any valid tile
        return gw + at * nW;
    };
    int vt_cur = 0, c_cur = 0;
    int vt_nx = 0, c_nx = 0;
    auto advance = [&](int& t, int& c) __attribute__((always_inline)) {
        const bool wrap = (c + 1 == NCH);
        c = wrap ? 0 : c + 1;
        t = wrap ? t + 1 : t;
    };
    float m_s = 1.f, m_e = 0.f, m_x = 0.f;   // metadata of row (tile*32 + r), requested at the tile's first chunk
    int tiles_done = 0;

    auto load_meta = [&](int tile) __attribute__((always_inline)) {
        if (PRAG_SH_DBG(a.dbg) & 32768) return;         // timing only: no row metadata
        const int64_t row = (int64_t)tile * 32 + r;     // (arrays are padded to a multiple of 32 rows)
        m_s = a.sscale[row];
        m_e = a.serr[row];
        m_x = a.sbias[row];
    };
    // stage one 4-KiB chunk (32 rows x 128 bytes), refill its registers with chunk (tile_nx, c_nx), 4 k-steps
    auto chunk_step = [&](u32x4 (&ldr)[4], int c, int tile_nx, int c_nx) __attribute__((always_inline)) {
        if (PRAG_SH_DBG(a.dbg) & 4096) {      // timing only: the stream alone (loads into registers, nothing else)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ldr[i]));
            issue(ldr, tile_nx, c_nx);
            return;
        }
        if constexpr (!kShadowFragMajor) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(s_st + st_dst[i]) = ldr[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (staged by some lanes, read as fragments by others)
            __builtin_amdgcn_wave_barrier();
            issue(ldr, tile_nx, c_nx);
        }
        // query-fragment slot of k-step piece P = 8 c + 2 s + hh in row qrow = 32 t + r:
        //   (P & ~15) | ((P ^ qrow) & 15)  =  (P & ~15) | (((8 c + 2 s) & 15) ^ xq),  xq = (r ^ hh) & 15
        int xq = xq0;
        // (unrolled chunks: hoisted out of the tile loop these offsets would be 4 NCHS registers)
        if constexpr (NCHS > 0) asm volatile("" : "+v"(xq));
        if (PRAG_SH_DBG(a.dbg) & 2048) {           // timing only: stream + staging writes, no fragment reads, no MFMAs
            if constexpr (kShadowFragMajor) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ldr[i]));
                issue(ldr, tile_nx, c_nx);
            }
            return;
        }
        // fragment-major chunks: the KiB a lane group loaded IS k-step s's A operand; its registers are refilled with
        // the same KiB of chunk (tile_nx, c_nx) as soon as the step's MFMAs have been issued (they read their
        // operands at issue)
        [[maybe_unused]] const char* nx_base = rows + (int64_t)tile_nx * (32 * row_bytes) + c_nx * 4096;
#pragma unroll
        for (int s = 0; s < 4; ++s) {               // 4 k-steps of 32 elements per 128-byte chunk
            i32x4 av;
            if constexpr (kShadowFragMajor) av = __builtin_bit_cast(i32x4, ldr[s]);
            else av = *reinterpret_cast<const i32x4*>(s_st + a_off + (((2 * s + hh) ^ a_sw) << 4));
            const int P0 = c * 8 + 2 * s;           // (bit 0 = hh lives in xq)
            const int q_sw = ((P0 & ~15) | (((P0 & 15) ^ xq))) << 4;
            // the first k-step of a tile starts from the constant 0 (an inline operand of the MFMA): the epilogue does not
            // clear 16 (32) accumulator registers per query column
            if (s == 0 && c == 0) {
                const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < NQ; ++t) {
                    const int q_addr = q_base[t] + q_sw;
                    const i32x4 b1 = *reinterpret_cast<const i32x4*>(s_qa + q_addr);
                    acc1[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b1, zero, 0, 0, 0);
                    if constexpr (TERMS == 2) {
                        const i32x4 b2 = *reinterpret_cast<const i32x4*>(s_qb + q_addr);
                        acc2[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b2, zero, 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < NQ; ++t) {
                    const int q_addr = q_base[t] + q_sw;
                    const i32x4 b1 = *reinterpret_cast<const i32x4*>(s_qa + q_addr);
                    acc1[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b1, acc1[t], 0, 0, 0);
                    if constexpr (TERMS == 2) {
                        const i32x4 b2 = *reinterpret_cast<const i32x4*>(s_qb + q_addr);
                        acc2[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b2, acc2[t], 0, 0, 0);
                    }
                }
            }
            if constexpr (kShadowFragMajor) {
#ifdef PRAG_SCAN_PLAIN_LOADS
                ldr[s] = *reinterpret_cast<const u32x4*>(nx_base + lane16 + 1024 * s);
#else
                ldr[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(nx_base + lane16 + 1024 * s));
#endif
            }
        }
    };
    // ---- epilogue of a tile: 16 rows x this lane's queries ----------------------------------------
    auto epilogue = [&](int tile_cur, bool second) __attribute__((always_inline)) {   // second visit of a warm-up tile: filter only
        {
            // ---- epilogue: 16 rows x this lane's queries ---------------------------------------
            const bool collect = (second || !warm) && !(PRAG_SH_DBG(a.dbg) & 2);
            if (PRAG_SH_DBG(a.dbg) & 1024) {       // timing only: no epilogue arithmetic at all
                ++tiles_done;
                warm = false;
                return;
            }
            if (hh == 0) {
                s_meta[r] = m_s;
                s_meta[32 + r] = m_e;
                s_meta[64 + r] = m_x;
            }
            // same-wave exchange through LDS: DS operations of a wave complete in order; the fence keeps the
            // compiler from moving the reads above the stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int64_t doc0 = (int64_t)tile_cur * 32;
            // one (row, query) pair: list / best-key update, filter, append
            // key of one (row, query) pair and its error bound - ONE definition for the group test and the per-pair code
            auto key_eps = [&](int t, int ge, float rs_e, float re_e, float rx_e, float ks, float cA_t, float cC_t, float& mid,
                               float& eps) __attribute__((always_inline)) {
                float dq = (float)acc1[t][ge];
                if constexpr (TERMS == 2) dq = fmaf((float)acc2[t][ge], 1.0f / 128.0f, dq);
                mid = fmaf(ks * rs_e, dq, rx_e);
                eps = fmaf(cA_t, re_e, cC_t);
            };
            auto pair = [&](int t, int ge, int64_t doc, float rs_e, float re_e, float rx_e, float ks, float cA_t, float cC_t,
                            float tau_t) __attribute__((always_inline)) {
                const bool valid = doc < a.N;
                float mid, eps;
                key_eps(t, ge, rs_e, re_e, rx_e, ks, cA_t, cC_t, mid, eps);
                if (!second) {
                    if constexpr (LISTS) top[t].push(valid ? mid + eps : INFINITY, tau_t);
                    else top[t].k[0] = fminf(top[t].k[0], valid ? mid + eps : INFINITY);
                }
                const float lo = mid - eps;
                if (collect && valid && lo <= tau_t) {     // cannot be excluded: candidate
                    // appended HERE, at each of the sites.  Round 3 measured the alternatives on one box (21 M
                    // rows, 64 queries): all candidates through a per-lane pending list in LDS and one copy of
                    // the append code behind the unrolled part 2.90 ms vs 2.82; an else branch at every site
                    // that sends what a full region cannot take to a per-workgroup overflow pool 3.34 - 3.44 ms.
                    // A region that overflows flags its query instead.
                    const uint32_t slot = atomicAdd(&s_ccnt[32 * t + r], 1u);
                    if (slot < (uint32_t)a.cap)
                        a.cand[((int64_t)blockIdx.x * QT + 32 * t + r) * a.cap + slot] = int2{(int)doc, (int)__float_as_uint(lo)};
                }
            };
            float tau[NQ];
            if constexpr (COL_EPI) {
                // query tile outermost, its constants read from LDS: with four query columns per lane the
                // row-major order below keeps 16 constants and the row metadata live at once and spills
                // Quad test on the INTEGER accumulators (round 5): with kscale <= 0 (alpha is -1 or -2) and s_e >= 0,
                //   key_e - eps_e >= rx_min - |kscale| s_max max(acc_0..3, 0) - (A e_max + C)
                // - every step is monotone in its operands and rounds once, exactly as the per-pair code does, so no pair
                // the per-pair code would take is skipped (a few more quads are taken).  The three per-quad extremes are
                // formed once per tile (not per query column), a quad of a column then costs two v_max3_i32, one convert,
                // one multiply, two fmas, a subtraction and the compare - it was 4 converts, 4 multiplies, 5 fmas, 6 min /
                // max.  profiles/r05o_scan8_ablation.txt: the epilogue's issue time adds to the stream's, 0.58 ms of a
                // 3.06 ms scan at 128 queries.
                float rs_max[4], rx_min[4], re_max[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 rs = *reinterpret_cast<const f32x4*>(s_meta + 8 * g + 4 * hh);
                    const f32x4 re = *reinterpret_cast<const f32x4*>(s_meta + 32 + 8 * g + 4 * hh);
                    const f32x4 rx = *reinterpret_cast<const f32x4*>(s_meta + 64 + 8 * g + 4 * hh);
                    rs_max[g] = fmaxf(fmaxf(rs[0], rs[1]), fmaxf(rs[2], rs[3]));
                    re_max[g] = fmaxf(fmaxf(re[0], re[1]), fmaxf(re[2], re[3]));
                    rx_min[g] = fminf(fminf(rx[0], rx[1]), fminf(rx[2], rx[3]));
                    // (the last, partial tile of a shard: every quad takes the per-pair code, which masks rows past N)
                    if (!QUAD || doc0 + 32 > a.N) rx_min[g] = -INFINITY;
                }
#pragma unroll
                for (int t = 0; t < NQ; ++t) {
                    tau[t] = unsortable_f32(s_tau[32 * t + r]);
                    float ks, cA_t, cC_t;
                    if constexpr (REG_CONST) {
                        ks = kscale[t]; cA_t = cA[t]; cC_t = cC[t];
                    } else {
                        ks = s_sqc[3 * (32 * t + r)]; cA_t = s_sqc[3 * (32 * t + r) + 1]; cC_t = s_sqc[3 * (32 * t + r) + 2];
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // (list-less: a lane's best key + eps per query is only refreshed in quads that hold something
                        //  under the bound - a skipped value is above the bound and could not have lowered any slot
                        //  below it; the stale one stays valid (larger))
                        int am = max(max(acc1[t][4 * g], acc1[t][4 * g + 1]), max(acc1[t][4 * g + 2], acc1[t][4 * g + 3]));
                        am = max(am, 0);
                        const float m = fmaf(ks * rs_max[g], (float)am, rx_min[g]) - fmaf(cA_t, re_max[g], cC_t);
                        if (!(m > tau[t])) {       // (also when a NaN got in: the per-pair code decides)
                            const f32x4 rs = *reinterpret_cast<const f32x4*>(s_meta + 8 * g + 4 * hh);
                            const f32x4 re = *reinterpret_cast<const f32x4*>(s_meta + 32 + 8 * g + 4 * hh);
                            const f32x4 rx = *reinterpret_cast<const f32x4*>(s_meta + 64 + 8 * g + 4 * hh);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                pair(t, 4 * g + e, doc0 + 8 * g + 4 * hh + e, rs[e], re[e], rx[e], ks, cA_t, cC_t, tau[t]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < NQ; ++t) tau[t] = unsortable_f32(s_tau[32 * t + r]);
                // Quad test first: the smallest key - eps of four rows of one query column.  Above the bound none of
                // the four is a candidate and none can enter the bound list either (key + eps >= key - eps > tau): the
                // list push and the append - two compares and two branches per pair - run for the quads that hold
                // something (a few per cent of them per wave; profiles/r04s_scan8_ablation.txt: every instruction of
                // this epilogue is exposed - two waves per SIMD do not hide it).  Rows past N (last tile) take the
                // per-pair code, which masks them.
                constexpr bool kQuadTest = QUAD;   // (64 queries x 32-deep lists: no registers for it)
                const bool partial = doc0 + 32 > a.N;
                if (kQuadTest && !partial && !(PRAG_SH_DBG(a.dbg) & 65536)) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // (the metadata of a row quad is read where it is used: without the clobber all four quads' 48
                        //  values are loaded ahead of the first branch and the kernel spills)
                        asm volatile("" ::: "memory");
                        const f32x4 rs = *reinterpret_cast<const f32x4*>(s_meta + 8 * g + 4 * hh);
                        const f32x4 re = *reinterpret_cast<const f32x4*>(s_meta + 32 + 8 * g + 4 * hh);
                        const f32x4 rx = *reinterpret_cast<const f32x4*>(s_meta + 64 + 8 * g + 4 * hh);
                        // the test uses the quad's largest error bound for all four rows: min(key) - max(eps) <= every
                        // key - eps (a few more quads take the exact per-pair code, none is missed), and a pair costs
                        // convert, multiply, fma, min instead of two more fmas and a subtraction
                        const float re_max = fmaxf(fmaxf(re[0], re[1]), fmaxf(re[2], re[3]));
                        // single-term tiles: the test on the integer accumulators (see the 128-query form above)
                        [[maybe_unused]] const float rs_max = fmaxf(fmaxf(rs[0], rs[1]), fmaxf(rs[2], rs[3]));
                        [[maybe_unused]] const float rx_min = fminf(fminf(rx[0], rx[1]), fminf(rx[2], rx[3]));
#pragma unroll
                        for (int t = 0; t < NQ; ++t) {
                            float mid[4], eps[4];
                            float m;
                            if constexpr (TERMS == 1 && PRAG_S8_IMAX_TEST) {
                                int am = max(max(acc1[t][4 * g], acc1[t][4 * g + 1]), max(acc1[t][4 * g + 2], acc1[t][4 * g + 3]));
                                am = max(am, 0);
                                m = fmaf(kscale[t] * rs_max, (float)am, rx_min) - fmaf(cA[t], re_max, cC[t]);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float dq = (float)acc1[t][4 * g + e];
                                    if constexpr (TERMS == 2) dq = fmaf((float)acc2[t][4 * g + e], 1.0f / 128.0f, dq);
                                    mid[e] = fmaf(kscale[t] * rs[e], dq, rx[e]);          // (key_eps's arithmetic)
                                }
                                m = fminf(fminf(mid[0], mid[1]), fminf(mid[2], mid[3])) - fmaf(cA[t], re_max, cC[t]);
                            }
                            if (m <= tau[t]) {
                                if constexpr (TERMS == 1 && PRAG_S8_IMAX_TEST) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) mid[e] = fmaf(kscale[t] * rs[e], (float)acc1[t][4 * g + e], rx[e]);
                                }
#pragma unroll
                                for (int e = 0; e < 4; ++e) eps[e] = fmaf(cA[t], re[e], cC[t]);
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (!second) {
                                        if constexpr (LISTS) top[t].push(mid[e] + eps[e], tau[t]);
                                        else top[t].k[0] = fminf(top[t].k[0], mid[e] + eps[e]);
                                    }
                                    const float lo = mid[e] - eps[e];
                                    if (collect && lo <= tau[t]) {
                                        // (per-lane pieces rebuilt HERE from an opaque copy of the lane id: hoisted out of
                                        //  the tile loop they are 20 registers the loop does not have - spilled, and a
                                        //  reload waits for vmcnt(0), i.e. drains the prefetch queue)
                                        int le = lane;
                                        asm volatile("" : "+v"(le));
                                        const int rq = le & 31, row_off = 4 * (le >> 5) + 8 * g + e;
                                        const uint32_t slot = atomicAdd(&s_ccnt[32 * t + rq], 1u);
                                        if (slot < (uint32_t)a.cap)
                                            a.cand[((int64_t)blockIdx.x * QT + 32 * t + rq) * a.cap + slot] =
                                                int2{(int)doc0 + row_off, (int)__float_as_uint(lo)};
                                    }
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);   // (one quad at a time: all eight in flight spill)
                        }
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 rs = *reinterpret_cast<const f32x4*>(s_meta + 8 * g + 4 * hh);
                        const f32x4 re = *reinterpret_cast<const f32x4*>(s_meta + 32 + 8 * g + 4 * hh);
                        const f32x4 rx = *reinterpret_cast<const f32x4*>(s_meta + 64 + 8 * g + 4 * hh);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int t = 0; t < NQ; ++t)
                                pair(t, 4 * g + e, doc0 + 8 * g + 4 * hh + e, rs[e], re[e], rx[e], kscale[t], cA[t], cC[t], tau[t]);
                    }
                }
            }
            if (!second && !(PRAG_SH_DBG(a.dbg) & 16384)) {     // (16384, timing only: no bound maintenance)
#pragma unroll
                for (int t = 0; t < NQ; ++t)
                    if constexpr (LISTS)
                        if (top[t].k[KC - 1] < tau[t]) atomicMin(&s_tau[32 * t + r], sortable_u32(top[t].k[KC - 1]));
                ++tiles_done;
#ifdef PRAG_MM_DIAG
                if (a.stamps && (tiles_done & (tiles_done - 1)) == 0 && tiles_done <= 128) S8_STAMP(2 + (31 - __builtin_clz(tiles_done)));
                if (a.stamps && tiles_done == n_my) S8_STAMP(10);
#endif
                if (warm) {
                    ++redo;
                    // leave the warm-up when a chip-wide slot bound has arrived for every query (a lane's own
                    // 16th-best-of-16 is a bound too, but one that lets nearly every row through)
                    if (*s_slot_ok != 0u || redo >= kWarmMax) warm = false;
                }
                // bound slots (see flat_index.hip kSlotWords): epoch e is fed after tile 2^e
                if ((tiles_done & (tiles_done - 1)) == 0 && tiles_done <= (1 << (kShadowEpochs - 1))) {
#pragma unroll
                    for (int t = 0; t < NQ; ++t) {
                        const float k0 = top[t].k[0];
                        if (k0 < INFINITY) atomicMin(&s_best[32 * t + r], sortable_u32(k0));
                    }
                    if (tiles_done == 1) {
                        // epoch 0: the LAST wave of the workgroup to get here publishes at once (every list
                        // has fed by then: LDS operations of a wave complete in order), so the bound can
                        // be polled one tile earlier: the warm-up is a tile shorter and fewer early rows
                        // pass the filter (2.6 M-row shard: scan 474 -> 412 us).  Doing the same for the
                        // later epochs, with a third poll each, made the whole scan 8 % SLOWER (3.12 ->
                        // 3.38 ms at 21 M rows; the extra in-loop polls, presumably their vmcnt(0)).
                        const int n_active = min(8, max(0, (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x));   // waves of this workgroup that own a tile
                        uint32_t old = 0;
                        if (lane == 0) old = atomicAdd(s_arrive, 1u);
                        old = (uint32_t)__shfl((int)old, 0, 64);
                        if ((int)old + 1 == n_active) {
                            for (int qq = lane; qq < QT; qq += 64) {
                                const uint32_t v = s_best[qq];
                                if (v != 0xFFFFFFFFu)
                                    (void)__hip_atomic_fetch_min(a.g_slot + (qq * kShadowSlotRows + 0) * 32 + ((int)blockIdx.x % a.kslots), v,
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                }
                {
                    // (wave 0 publishing epoch 0 right away, before the other waves had fed, was tried: the
                    // warm-up then ends on a weaker bound and the early tiles flood the candidate regions,
                    // 0.53 -> 0.57 ms on a 2.6 M-row shard - hence the last-arriver rule above)
                    const int tm = tiles_done - 1;          // published one tile after the lists fed s_best
                    if (w == 0 && tm >= 1 && (tm & (tm - 1)) == 0 && tm <= (1 << (kShadowEpochs - 1))) {
                        const int epoch = 31 - __builtin_clz(tm);
                        for (int qq = lane; qq < QT; qq += 64) {
                            const uint32_t v = s_best[qq];
                            if (v != 0xFFFFFFFFu)
                                (void)__hip_atomic_fetch_min(a.g_slot + (qq * kShadowSlotRows + epoch) * 32 + ((int)blockIdx.x % a.kslots), v,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    // polls 2 and 3 tiles after an epoch was fed (epoch 0 also after 1), waves take turns
#pragma unroll
                    for (int i = -1; i < 2; ++i) {
                        const int tp = tiles_done - 2 - i;
                        if (tp >= 1 && (tp & (tp - 1)) == 0 && tp <= (1 << (kShadowEpochs - 1)) && (i >= 0 || tp == 1)) {
                            const int epoch = 31 - __builtin_clz(tp);
                            if (w == ((2 * epoch + i + 5) & 7)) {
                                bool missing = false;
                                for (int qq = lane; qq < QT; qq += 64) {
                                    const uint32_t* sl = a.g_slot + (qq * kShadowSlotRows + epoch) * 32;
                                    uint32_t m = 0u;
                                    // k slots, not KC: the maximum over m slots fed by disjoint sets of workgroups bounds
                                    // the m-th best key, and it sits at about rank m H(m) of the rows seen - 29 for 10
                                    // slots, 54 for 16.  The bound only has to cover the k-th best.
#pragma unroll 4
                                    for (int s2 = 0; s2 < a.kslots; ++s2) {
                                        const uint32_t v = __hip_atomic_load(sl + s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        m = v > m ? v : m;
                                    }
                                    if (m < s_tau[qq]) atomicMin(&s_tau[qq], m);
                                    // padding queries (bound -inf) never get slot values: they do not count
                                    missing |= m == kSortablePosInf && s_tau[qq] != kSortableNegInf;
                                }
                                if (__builtin_amdgcn_ballot_w64(missing) == 0 && lane == 0) *s_slot_ok = 1u;
                            }
                        }
                    }
                }
                if (w == 0 && (tiles_done & 7) == 0 && hh == 0) {
#pragma unroll
                    for (int t = 0; t < NQ; ++t) {
                        const uint32_t loc = s_tau[32 * t + r];
                        const uint32_t glob = __hip_atomic_load(a.g_tau + 32 * t + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (loc < glob)
                            __hip_atomic_fetch_min(a.g_tau + 32 * t + r, loc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else if (glob < loc)
                            atomicMin(&s_tau[32 * t + r], glob);
                    }
                }
            }
        }
    };
    // the wave that streams and feeds the matrix pipe is served ahead of its SIMD partner's epilogue (same box, 21 M rows:
    // 128-query tiles 2.886-2.908 -> 2.864-2.878 ms per search, 64-query tiles 2.560 -> 2.550 ms per launch)
#ifndef PRAG_S8_NO_SETPRIO
#define S8_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define S8_PRIO(x) do {} while (0)
#endif
    auto body = [&](u32x4 (&ldr)[4], bool may_end) __attribute__((always_inline)) {      // may_end: a constant at every call site
        const int tile_cur = vtile(vt_cur);
        if (c_cur == 0) load_meta(tile_cur);
        chunk_step(ldr, c_cur, vtile(vt_nx), c_nx);
        advance(vt_nx, c_nx);
        if (may_end && c_cur == NCH - 1) {
            S8_PRIO(0);
            epilogue(tile_cur, vt_cur >= n_my);
            S8_PRIO(1);
        }
        advance(vt_cur, c_cur);
    };

#ifdef PRAG_MM_DIAG
    if (a.stamps && lane == 0) a.stamps[((int64_t)blockIdx.x * 8 + w) * kScan8Stamps + 0] = stamp_entry;
    S8_STAMP(1);
#endif
    if constexpr (NCHS > 0) {
        static_assert(NCHS == 0 || NCHS % NLD == 0, "staging buffers rotate with the chunks of a tile");
        if (n_my > 0) {
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                issue(ld[u], vtile(0), u);
                __builtin_amdgcn_sched_barrier(0);   // in this order: the counted waits in the loop rely on it
            }
            for (int vt = 0; vt < n_my + redo; ++vt) {
                const int tile_cur = vtile(vt), tile_next = vtile(vt + 1);
                load_meta(tile_cur);         // NCHS refills older than its use in the epilogue
                S8_PRIO(1);
#pragma unroll
                for (int c = 0; c < NCHS; ++c) {
                    const int cn = c + NLD;  // the chunk this buffer holds next
                    chunk_step(ld[c % NLD], c, cn < NCHS ? tile_cur : tile_next, cn < NCHS ? cn : cn - NCHS);
                    __builtin_amdgcn_sched_barrier(0);
                }
                S8_PRIO(0);
                epilogue(tile_cur, vt >= n_my);
            }
        }
    } else if (n_my > 0) {
        S8_PRIO(1);
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            issue(ld[u], vtile(vt_nx), c_nx);
            advance(vt_nx, c_nx);
            // in this order (left alone, the compiler requests the LAST set first and the loop then opens every
            // round of NLD chunks with s_waitcnt vmcnt(0): the waits are counted from the issue order)
            if constexpr (kShadowFragMajor) __builtin_amdgcn_sched_barrier(0);
        }
        // (`redo` grows during the first tiles only, long before the loop bound is reached - or, on a
        // shard of a few tiles, up to n_my: every tile is then visited twice)
        // (`redo` grows during the first tiles only, long before the loop bound is reached - or, on a
        // shard of a few tiles, up to n_my: every tile is then visited twice)
        if constexpr (ALN > 0) {
            // ALN: rows of a whole number of rounds (the host checks (d / 128) % ALN == 0): a tile can only end in the LAST staging set of a round - ONE copy of the
            // epilogue in the loop instead of NLD (85 KB of code for 64-query tiles against a 64-KB instruction cache
            // shared by two CUs; profiles/r04s_scan8_ablation.txt: code that was only skipped, not removed, made the
            // loop faster)
            for (int it = 0; it < (n_my + redo) * NCH; it += NLD) {
                body(ld[0], NLD == 1);
                if constexpr (NLD > 1) body(ld[1], NLD == 2);
                if constexpr (NLD > 2) body(ld[2], NLD == 3);
                if constexpr (NLD > 3) body(ld[3], NLD == 4);
                if constexpr (NLD > 4) body(ld[4], NLD == 5);
                if constexpr (NLD > 5) body(ld[5], NLD == 6);
            }
        } else {
            // (written out: left to `#pragma unroll`, one instantiation came back with the loop over the staging sets
            //  NOT unrolled - ld[u] indexed at run time, i.e. 320 B of scratch - without a diagnostic)
            static_assert(NLD >= 2 && NLD <= 6, "staging sets written out below");
            for (int it = 0; it < (n_my + redo) * NCH; it += NLD) {
                body(ld[0], true);
                if (it + 1 < (n_my + redo) * NCH) body(ld[1], true);
                if constexpr (NLD > 2) { if (it + 2 < (n_my + redo) * NCH) body(ld[2], true); }
                if constexpr (NLD > 3) { if (it + 3 < (n_my + redo) * NCH) body(ld[3], true); }
                if constexpr (NLD > 4) { if (it + 4 < (n_my + redo) * NCH) body(ld[4], true); }
                if constexpr (NLD > 5) { if (it + 5 < (n_my + redo) * NCH) body(ld[5], true); }
            }
        }
    }
    S8_PRIO(0);
    S8_STAMP(11);
    S8_VALUE(13, redo);
    S8_VALUE(14, n_my);
    __syncthreads();
    if (tid < QT) {
        a.ccnt[(int64_t)blockIdx.x * QT + tid] = s_ccnt[tid];
        // the final bound of this workgroup: the gather drops candidates that a later, tighter bound excludes
        (void)__hip_atomic_fetch_min(a.g_tau + tid, s_tau[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef PRAG_MM_DIAG
    if (a.stamps && tid == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < QT; ++i) tot += s_ccnt[i];
        a.stamps[((int64_t)blockIdx.x * 8) * kScan8Stamps + 15] = tot;
    }
#endif
    S8_STAMP(12);
#undef S8_STAMP
#undef S8_VALUE
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
static int scan8_lds_bytes(int QT, int qstride) {
    return (QT == 32 ? 2 : 1) * QT * qstride + 8 * 4096 + 8 * 384 + 4 * QT * 4 + 64 + (QT == 128 ? 3 * QT * 4 + 64 : 0);
}

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

constexpr int kScan8Aln = 3;
int shadow_search(const ShadowSearch& s, hipStream_t st, EventRing& prof) {
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
    const int grid = std::max(1, std::min(s.max_wg, (n_tiles + 7) / 8));
    PRAG_REQUIRE(grid <= s.wg_slots, PRAG_EUNSUPPORTED, "internal: shadow candidate regions too few");
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
        const bool quad16 = (s.d / 128) % kScan8Aln == 0 && s.N >= s.quad_min_rows;
        const bool quad32 = s.N >= s.quad_min_rows;
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
            if (tg) {            // the gate of the next batch in the same launch (bound_gate_kernel)
                const int rc_t = launch_bound_gate(s.store.store_f32 != 0, *tg, g, s.g_tau + p0, nq, st);
                if (rc_t != PRAG_OK) return rc_t;
                tg->taken = true;
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
