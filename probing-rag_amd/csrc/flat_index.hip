// Flat (exact brute-force) index for MI355X (gfx950, CDNA4): the replacement for
// faiss.IndexFlatL2.add/search on the Probing-RAG retrieval path
// (/root/reference: make_indexer.py:449-457, utils.py:378-380, exp_rag.py:432-436).
//
// search = 4 launches on the caller's stream (DESIGN.md "Kernel 2"):
//   1. prep_queries   : fp32 queries -> (cosine: normalise) -> fp16 tile + fp32 copy
//   2. scan_topk      : ONE pass over the stored rows per tile of <=64 queries.
//        HBM-bound: every wave streams its own 32-row slices with coalesced
//        128-B-line loads -> XOR-swizzled per-wave LDS stage -> ds_read_b128
//        fragments -> v_mfma_f32_32x32x16_f16 against the LDS-resident query
//        tile.  The 32x32 score tile never leaves registers: each lane keeps a
//        sorted top-KC list for its (query, row-half) stream; [B,N] is never
//        materialised.  Waves never synchronise with each other in the loop.
//   3. merge_lists    : per query, merge the per-lane lists into KC candidates
//        ordered by (key, row id).
//   4. rerank         : exact fp64 distance / inner product of the KC candidates
//        against the fp32 query, final (score, id) order, write D (f32) / I (i64).
// Step 4 makes the result independent of MFMA accumulation order: scores are the
// float32 rounding of the float64 value, ties break by lowest row id.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "exchange.h"
#include "flat_internal.h"
#include "flat_index_state.h"
#include "flat_plan.h"
#include "tail_gate.h"

namespace prag {

// ---------------------------------------------------------------------------
// counter-based synthetic rows: bit-identical to oracle_np.synth_rows
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

__host__ __device__ __forceinline__ float synth_value(uint32_t seed, uint64_t row, uint32_t col, uint32_t d) {
    const uint64_t ctr = row * (uint64_t)d + col;
    const uint32_t lo = (uint32_t)ctr, hi = (uint32_t)(ctr >> 32);
    const uint32_t h1 = mix32(lo ^ mix32(hi ^ seed));
    const uint32_t h2 = mix32(h1 ^ 0x9E3779B9u);
    const int t = (int)(h1 & 0xFFFFu) + (int)(h1 >> 16) + (int)(h2 & 0xFFFFu) + (int)(h2 >> 16) - 131070;
    return (float)t * (float)(1.0 / 37837.22);
}

// ---------------------------------------------------------------------------
// add: one wave per row.  cosine -> normalise in fp64, round to f32; store as
// f32 or f16; keep ||stored row||^2 (fp64 sum, rounded) for the L2 scan key.
// ---------------------------------------------------------------------------
template <bool STORE_F32, bool SYNTH>
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ src, uint32_t seed,
                                                      int64_t synth_row0, int64_t n, int d, int normalise,
                                                      void* __restrict__ rows, float* __restrict__ xnorm,
                                                      int64_t dst_row0) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* s = SYNTH ? nullptr : src + i * d;
    auto get = [&](int c) -> float {
        if constexpr (SYNTH) return synth_value(seed, (uint64_t)(synth_row0 + i), (uint32_t)c, (uint32_t)d);
        else return s[c];
    };
    double nrm = 1.0;
    if (normalise) {
        double ss = 0.0;
        for (int c = lane; c < d; c += 64) {
            const double v = (double)get(c);
            ss = fma(v, v, ss);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        nrm = sqrt(ss);
        if (!(nrm > 0.0)) nrm = 1.0;
    }
    double q = 0.0;
    for (int c = lane; c < d; c += 64) {
        float v = get(c);
        if (normalise) v = (float)((double)v / nrm);
        double kept;
        if constexpr (STORE_F32) {
            reinterpret_cast<float*>(rows)[(dst_row0 + i) * d + c] = v;
            kept = (double)v;
        } else {
            const _Float16 h = (_Float16)v;
            reinterpret_cast<_Float16*>(rows)[(dst_row0 + i) * d + c] = h;
            kept = (double)h;
        }
        q = fma(kept, kept, q);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if (lane == 0) xnorm[dst_row0 + i] = (float)q;
}

// ---------------------------------------------------------------------------
// queries: q32[b] = (cosine ? q/||q|| : q) in f32 ; q16 = fp16(q32), rows padded
// with zeros up to a multiple of the tile height.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_queries_kernel(const float* __restrict__ q, int B, int Bpad, int d,
                                                          int normalise, float* __restrict__ q32,
                                                          _Float16* __restrict__ q16, _Float16* __restrict__ q16lo,
                                                          uint32_t* __restrict__ g_tau,
                                                          uint32_t* __restrict__ mm_cnt, uint32_t* __restrict__ mm_ovf,
                                                          uint32_t mm_first_rows, float* __restrict__ qinfo,
                                                          double* __restrict__ qn2, uint32_t* __restrict__ n_flag,
                                                          int* __restrict__ flag_list, int flag_all,
                                                          uint32_t* __restrict__ g_slot, ShadowPrep sp, Gate gate) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= Bpad || gate_closed(gate)) return;
    // blockIdx.y > 0 (two-level search only): this wave scores slice blockIdx.y - 1 of the shadow's sample
    // against its query and writes one bound slot; everything else is the work of the y == 0 wave
    const bool sampler = blockIdx.y > 0;
    __shared__ __attribute__((aligned(16))) signed char s_q8[4][1024];
    if (sampler && (b >= B || sp.sample_stride == 0)) {
        shadow_prebound_none(sp, b, (int)blockIdx.y - 1, lane);
        return;
    }
    ShadowSample sm;
    if (!sampler) {
    // certificate bookkeeping: the flag list starts empty (or holds every query when the search goes
    // straight to the exact scan)
    if (b == 0 && lane == 0) *n_flag = flag_all ? (uint32_t)B : 0u;
    if (b == 0 && lane == 0 && sp.unfinished) *sp.unfinished = 0u;
    if (flag_all && b < B && lane == 0) flag_list[b] = b;
    if (mm_cnt && lane == 0) {  // MFMA-tiled scan: candidate counts / overflow flags (word Bpad: "any")
        mm_cnt[b] = mm_first_rows;
        mm_ovf[b] = 0u;
        if (b == 0) mm_ovf[Bpad] = 0u;
    }
    // the scan's chip-wide pruning bound: +inf for real queries; padding rows never collect anything
    if (lane == 0) g_tau[b] = b < B ? kSortablePosInf : kSortableNegInf;
    g_slot[(int64_t)b * kSlotWordsFwd + lane] = kSortablePosInf;   // bound slots of the list scan (2 epochs x 32)
    if (b >= B) {
        for (int c = lane * 4; c < d; c += 256) {
            *reinterpret_cast<half4*>(q16 + (int64_t)b * d + c) = half4{0, 0, 0, 0};
            *reinterpret_cast<half4*>(q16lo + (int64_t)b * d + c) = half4{0, 0, 0, 0};
        }
        if (sp.q8a) {
            const f32x4 zero[6] = {};
            shadow_prep_wave(sp, b, d, false, zero, lane);
        }
        return;
    }
    }
    const float* s = q + (int64_t)b * d;
    // d is a multiple of 64: 4 consecutive elements per lane per step, at most 6 steps (d <= 1536)
    f32x4 v[6];
    double ss = 0.0;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int c = lane * 4 + it * 256;
        v[it] = c < d ? *reinterpret_cast<const f32x4*>(s + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) ss = fma((double)v[it][e], (double)v[it][e], ss);
    }
    double nrm = 1.0;
    if (normalise) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        nrm = sqrt(ss);
        if (!(nrm > 0.0)) nrm = 1.0;
    }
    // norms for the exactness certificate: ||q||, ||q - q16||, ||q - q16 - q16lo|| of the query as used
    double n2 = 0.0, r1 = 0.0, r2 = 0.0;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < d) {
            f32x4 o;
            half4 h, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = normalise ? (float)((double)v[it][e] / nrm) : v[it][e];
                h[e] = (_Float16)o[e];
                lo[e] = (_Float16)(o[e] - (float)h[e]);
                const double od = (double)o[e];
                const double e1 = od - (double)h[e];
                const double e2 = e1 - (double)lo[e];
                n2 = fma(od, od, n2);
                r1 = fma(e1, e1, r1);
                r2 = fma(e2, e2, r2);
            }
            if (!sampler) {
                *reinterpret_cast<f32x4*>(q32 + (int64_t)b * d + c) = o;
                *reinterpret_cast<half4*>(q16 + (int64_t)b * d + c) = h;
                *reinterpret_cast<half4*>(q16lo + (int64_t)b * d + c) = lo;
            }
            v[it] = o;             // the query as the rerank uses it: what the 8-bit terms below approximate
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        n2 += __shfl_xor(n2, o, 64);
        r1 += __shfl_xor(r1, o, 64);
        r2 += __shfl_xor(r2, o, 64);
    }
    // the two-level search over an affine image of the rows (flat_shadow.hip): its int8 terms approximate p = q c
    // (c: powers of two, exact), and K_q = alpha q.mu moves an exact key into the scan's key space
    if (sp.q8a && sp.aff) {
        double kq = 0.0;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int c = lane * 4 + it * 256;
            if (c < d) {
                const f32x4 m4 = *reinterpret_cast<const f32x4*>(sp.aff + c);
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(sp.aff + d + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    kq = fma((double)v[it][e], (double)m4[e], kq);
                    // (centre_query: the rows carry alpha mu.(x - mu) themselves, the int8 terms see q - mu)
                    v[it][e] = (sp.centre_query ? v[it][e] - m4[e] : v[it][e]) * c4[e];
                }
            }
        }
        if (!sampler) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) kq += __shfl_xor(kq, o, 64);
            if (lane == 0 && sp.kshift) sp.kshift[b] = (double)sp.alpha * kq;
        }
    }
    if (sampler) {
        shadow_prebound_wave(sp, b, d, (int)blockIdx.y - 1, v, lane, s_q8[threadIdx.x >> 6], sm);
        return;
    }
    if (lane == 0) {
        qn2[b] = n2;
        // float roundings upwards: these feed an error BOUND
        qinfo[4 * b + 0] = (float)(sqrt(n2) * (1.0 + 1e-6));
        qinfo[4 * b + 1] = (float)(sqrt(r1) * (1.0 + 1e-6));
        qinfo[4 * b + 2] = (float)(sqrt(r2) * (1.0 + 1e-6));
        qinfo[4 * b + 3] = 0.f;
    }
    if (sp.q8a) shadow_prep_wave(sp, b, d, true, v, lane);
}

// max_i ||x_i||^2 over rows [row0, row1) folded into *out (float bits; non-negative floats order like
// uints): the certificate's bound on the norm of a row that is not among the candidates
__global__ __launch_bounds__(256) void xnorm_max_kernel(const float* __restrict__ xnorm, int64_t row0, int64_t row1,
                                                       uint32_t* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = row0 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < row1; i += (int64_t)gridDim.x * 256)
        m = fmaxf(m, xnorm[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

// ---------------------------------------------------------------------------
// scan + fused per-lane top-KC
// ---------------------------------------------------------------------------
struct ScanArgs {
    const void* rows;        // [N][d] f16 or f32
    const float* xnorm;      // [roundup(N,32)]
    const _Float16* q16;     // [QT][d] this pass's query tile (fp16 rounding of the f32 query)
    const _Float16* q16lo;   // [QT][d] fp16(q32 - q16): second term of the high-precision selection
    int64_t N;
    int d;
    int qstride;             // LDS bytes per query row (multiple of 256)
    int n_tiles;             // ceil(N / 32)
    float alpha;             // key = bias + alpha * dot  (L2: -2, IP: -1)
    int use_norm;            // L2: bias = ||x||^2
    float* out_key;          // [n_lists][QT][KC]
    int* out_idx;
    uint32_t* g_tau;         // [QT] chip-wide pruning bound per query (sortable-uint keys, +inf at start)
    uint32_t* g_slot;        // [QT][2 epochs][32] bound slots (below), or null: bound from g_tau only
    Gate gate;               // flat_internal.h: the kernels return at once when it is closed
};

// Chip-wide pruning bound without a pre-pass.  Twice per launch ("epochs": after a wave's 1st and 4th
// tile) every lane min-s the best key of its list into an LDS word per query; a tile later wave 0
// min-s the workgroup's best key of each query into global slot (workgroup mod KC) of that query and
// epoch.  The slots of one epoch receive from DISJOINT sets of workgroups, i.e. of rows: whenever all
// KC slots are set, KC distinct rows of the shard have keys <= max(slots), so the KC-th best key of the
// shard is <= max(slots) - at any time, whatever the interleaving (values only decrease).  One global
// atomic instruction per workgroup and epoch; with 240 workgroups a slot is the best of ~4 k rows after
// one tile (~16 k after four), the max over 16 slots the ~0.09 % (0.02 %) quantile: tighter than the
// 0.2 % the round-1 pre-pass bought with two extra launches.  The waves take turns polling the slots.

template <int KC>
struct TopList {
    float k[KC];
    int i[KC];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            k[j] = INFINITY;
            i[j] = kIdxSentinel;
        }
    }
    // strict '<': among equal keys the earlier (lower row id) entry stays in front
    // `tau` is a workgroup-wide upper bound on the KC-th best key of this query
    // (some lane already holds KC keys <= tau): anything above it cannot survive.
    __device__ __forceinline__ void push(float key, int idx, float tau) {
        if (key < k[KC - 1] && key <= tau) {
            // sorted insertion, dropping the last entry: new_k[j] = clamp(key, old[j-1], old[j])
            // = v_med3_f32; the id follows the same choice via the two compares
            bool prev = false;       // key < old[j-1]
            float ok = -INFINITY;    // old[j-1]
            int oi = 0;
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                const float cur_k = k[j];
                const int cur_i = i[j];
                const bool c = key < cur_k;
                k[j] = __builtin_amdgcn_fmed3f(ok, cur_k, key);
                i[j] = prev ? oi : (c ? idx : cur_i);
                prev = c;
                ok = cur_k;
                oi = cur_i;
            }
        }
    }
};

// HP ("high-precision selection", QT == 32 only): the query is carried as q16 + q16lo and fp32
// rows as hi + lo halves, 2 (fp16 rows) or 3 (fp32 rows) MFMAs per k-step - the matrix pipe is
// mostly idle in this HBM-bound kernel - so the selection keys are accurate to f32 accumulation
// (~1e-7 relative) instead of the fp16 rounding of the operands (~3e-5): the reference's own call
// shape (a handful of queries, fp32 rows) no longer depends on how many near-ties surround the
// k-th result.
template <int QT, int KC, bool F32, bool HP>
__device__ __forceinline__ void scan_topk_body(const ScanArgs& a, char* smem) {
    static_assert(!HP || QT == 32, "high-precision selection is built for one 32-query tile");
    constexpr int NQ = QT / 32;
    constexpr bool SPLIT_ROWS = HP && F32;          // rows staged as hi + lo halves
    constexpr int CHK = SPLIT_ROWS ? 32 : 64;       // elements per staged chunk (4 KiB of LDS per wave either way)
    constexpr int KS = CHK / 16;                    // MFMA k-steps per chunk
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int d = a.d;
    const int NCH = d / CHK;
    const int qstride = a.qstride;
    constexpr int NQT = HP ? 2 : 1;                 // query tiles in LDS: q16 (+ q16lo)
    char* s_q = smem;
    char* s_ql = smem + QT * qstride;               // HP only
    char* s_st = smem + NQT * QT * qstride + w * 4096;
    uint32_t* s_tau = reinterpret_cast<uint32_t*>(smem + NQT * QT * qstride + 8 * 4096);  // [QT] sortable keys
    uint32_t* s_best = s_tau + QT;   // [2 epochs][QT] best key this workgroup has seen per query (bound slots)
    if (tid < QT) {
        s_tau[tid] = a.g_tau[tid];  // +inf, or the pre-pass bound (valid: subset of the shard)
        s_best[tid] = 0xFFFFFFFFu;
        s_best[QT + tid] = 0xFFFFFFFFu;
    }

    // ---- query tile(s) -> LDS (swizzled 16-B pieces); loads issued in batches of 8 so their
    //      latencies overlap (a one-load-at-a-time loop costs ~25 us per launch) -----------
    {
        const int ppr = d >> 3;  // pieces per row
        const int total = QT * ppr;
#pragma unroll
        for (int plane = 0; plane < NQT; ++plane) {
            const _Float16* qsrc = plane == 0 ? a.q16 : a.q16lo;
            char* qdst = plane == 0 ? s_q : s_ql;
            for (int e0 = tid; e0 < total; e0 += 512 * 8) {
                u32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * 512;
                    const int ec = e < total ? e : total - 1;
                    const int row = ec / ppr, pc = ec - row * ppr;
                    v[u] = *reinterpret_cast<const u32x4*>(qsrc + (int64_t)row * d + 8 * pc);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
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
    const int gw = blockIdx.x * 8 + w;
    const int n_my = gw < a.n_tiles ? (a.n_tiles - gw + nW - 1) / nW : 0;

    TopList<KC> top[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) top[t].init();

    // staging geometry: NLD 16-B loads per lane per chunk; a staged row is 2*CHK bytes of fp16 and
    // its 16-B pieces are XOR-swizzled so that the fragment reads below are conflict-free
    //   64-element chunks (128-B rows): slot = piece ^ ((row>>1)&7)
    //   32-element chunks ( 64-B rows): slot = piece ^ ((row>>2)&3)
    constexpr int NLD = SPLIT_ROWS ? 4 : (F32 ? 8 : 4);
    int st_doc[NLD], st_dst[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        if constexpr (SPLIT_ROWS) {   // 32 rows x 128 B of f32 per chunk: lane -> 4 floats -> 8 B of hi, 8 B of lo
            const int doc = 8 * i + (lane >> 3), q4 = lane & 7;
            st_doc[i] = doc;
            st_dst[i] = doc * 64 + (((q4 >> 1) ^ ((doc >> 2) & 3)) << 4) + (q4 & 1) * 8;
        } else if constexpr (F32) {
            const int doc = 4 * i + (lane >> 4), q4 = lane & 15;
            st_doc[i] = doc;
            st_dst[i] = doc * 128 + (((q4 >> 1) ^ ((doc >> 1) & 7)) << 4) + (q4 & 1) * 8;
        } else {
            const int doc = 8 * i + (lane >> 3), q = lane & 7;
            st_doc[i] = doc;
            st_dst[i] = doc * 128 + ((q ^ ((doc >> 1) & 7)) << 4);
        }
    }
    const int col_b = SPLIT_ROWS ? (lane & 7) * 16 : (F32 ? (lane & 15) * 16 : (lane & 7) * 16);  // byte offset inside the chunk
    const int64_t row_bytes = (int64_t)d * (F32 ? 4 : 2);
    const int chunk_bytes = CHK * (F32 ? 4 : 2);
    const char* rows = reinterpret_cast<const char*>(a.rows);

    // Two chunks of every lane's 16-B pieces are kept in flight (ldA / ldB).  The
    // refill is unconditional and branch-free - past the last tile it re-reads the
    // (clamped) last row - so the registers never merge across control flow and
    // the compiler keeps counted vmcnt waits instead of draining the queue.
    // float32 rows with 32-deep lists and no high-precision query tile (d >= 1024: two query tiles do not fit LDS):
    // 64 list registers + two sets of 8 staging registers + the prefetched norms do not fit 256 VGPRs (280 B of
    // scratch in round 3).  That shape keeps ONE chunk in flight and reads the row norms in the epilogue.
    constexpr bool ONE_SET = F32 && !HP && KC == 32;
    u32x4 ldA[NLD], ldB[ONE_SET ? 1 : NLD];
    // wave-uniform tile base (scalar registers) + a per-lane 32-bit offset: no 64-bit vector address
    // arithmetic and no per-row clamp in the loop.  The rows are allocated in multiples of 256, so the
    // rows of the last, partial tile past N are readable; their scores are masked in the epilogue.
    // Load i of a lane sits 32/NLD rows below load 0: ONE lane offset, the rest is wave-uniform (an array of
    // NLD offsets cost NLD - 1 registers the 32-deep lists and the float32 rows did not have: scratch).
    const int lane_off0 = st_doc[0] * (int)row_bytes + col_b;
    const int64_t ld_step = (int64_t)(32 / NLD) * row_bytes;
    const int tile_last = a.n_tiles - 1;
    auto issue = [&](u32x4 (&ld)[NLD], int tile, int c) {
        const int tc = tile < tile_last ? tile : tile_last;     // prefetch past the end: re-read the last tile
        const char* base = rows + (int64_t)tc * (32 * row_bytes) + c * chunk_bytes;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            // (read-once stream: non-temporal, like the 8-bit scan - flat_shadow.hip)
#ifdef PRAG_SCAN_PLAIN_LOADS
            ld[i] = *reinterpret_cast<const u32x4*>(base + i * ld_step + lane_off0);
#else
            ld[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + i * ld_step + lane_off0));
#endif
        }
    };

    const int a_off = r * 128;
    const int a_sw = (r >> 1) & 7;

    f32x16 acc[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // (tile, chunk) cursors: cur = being consumed, nx = next to be requested
    int tile_cur = gw, c_cur = 0;
    int tile_nx = gw, c_nx = 0;
    auto advance = [&](int& t, int& c) {
        const bool wrap = (c + 1 == NCH);
        c = wrap ? 0 : c + 1;
        t = wrap ? t + nW : t;
    };
    f32x4 xn[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) xn[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    int tiles_done = 0;

    auto body = [&](u32x4 (&ld)[NLD]) {
        // norms of this tile's rows: requested at its first chunk, i.e. OLDER than
        // every prefetch issued below, so waiting for them does not drain the queue
        if (a.use_norm && (ONE_SET ? c_cur == NCH - 1 : c_cur == 0)) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                xn[g] = *reinterpret_cast<const f32x4*>(a.xnorm + (int64_t)tile_cur * 32 + 8 * g + 4 * hh);
        }

#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const u32x4 v = ld[i];
            if constexpr (SPLIT_ROWS) {
                const f32x4 f = __builtin_bit_cast(f32x4, v);
                half4 h, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = (_Float16)f[e];
                    lo[e] = (_Float16)(f[e] - (float)h[e]);
                }
                *reinterpret_cast<half4*>(s_st + st_dst[i]) = h;
                *reinterpret_cast<half4*>(s_st + 2048 + st_dst[i]) = lo;
            } else if constexpr (F32) {
                const f32x4 f = __builtin_bit_cast(f32x4, v);
                half4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (_Float16)f[e];
                *reinterpret_cast<half4*>(s_st + st_dst[i]) = h;
            } else {
                *reinterpret_cast<u32x4*>(s_st + st_dst[i]) = v;
            }
        }
        issue(ld, tile_nx, c_nx);
        advance(tile_nx, c_nx);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int a_addr = SPLIT_ROWS ? (r * 64 + (((2 * s + hh) ^ ((r >> 2) & 3)) << 4))
                                          : (a_off + (((2 * s + hh) ^ a_sw) << 4));
            const half8 av = *reinterpret_cast<const half8*>(s_st + a_addr);
            const int P = c_cur * (CHK / 8) + 2 * s + hh;
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                const int qrow = 32 * t + r;
                const int q_addr = qrow * qstride + (((P & ~15) | ((P ^ qrow) & 15)) << 4);
                const half8 bv = *reinterpret_cast<const half8*>(s_q + q_addr);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[t], 0, 0, 0);
                if constexpr (HP) {
                    const half8 bl = *reinterpret_cast<const half8*>(s_ql + q_addr);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bl, acc[t], 0, 0, 0);
                    if constexpr (SPLIT_ROWS) {
                        const half8 al = *reinterpret_cast<const half8*>(s_st + 2048 + a_addr);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bv, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        if (c_cur == NCH - 1) {
            // ---- epilogue: 16 rows x this lane's queries -> running top-KC -------
            const int64_t doc0 = (int64_t)tile_cur * 32;
            float tau[NQ];
#pragma unroll
            for (int t = 0; t < NQ; ++t) tau[t] = unsortable_f32(s_tau[32 * t + r]);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t doc = doc0 + 8 * g + 4 * hh + e;
                    const bool valid = doc < a.N;
#pragma unroll
                    for (int t = 0; t < NQ; ++t) {
                        const float key = valid ? fmaf(a.alpha, acc[t][4 * g + e], xn[g][e]) : INFINITY;
                        top[t].push(key, (int)doc, tau[t]);
                        acc[t][4 * g + e] = 0.f;
                    }
                }
            // publish this lane's KC-th best when it tightened the workgroup bound
#pragma unroll
            for (int t = 0; t < NQ; ++t)
                if (top[t].k[KC - 1] < tau[t]) atomicMin(&s_tau[32 * t + r], sortable_u32(top[t].k[KC - 1]));
            // chip-wide pruning bound: wave 0 of every workgroup trades its LDS bound with the
            // global one every 8th tile (any lane's KC-th best anywhere is a valid upper bound;
            // a stale value is only looser).  Kept rare: 2048 waves on 64 hot words serialise.
            ++tiles_done;
            // (not for 64 queries x 32-deep lists: 128 list registers leave no room, the whole list
            // state went to scratch; that shape keeps the pre-pass)
            constexpr bool SLOTS = !(QT == 64 && KC == 32);
            if (SLOTS && a.g_slot) {
                if (tiles_done == 1 || tiles_done == 4) {        // every list: best key -> LDS
                    const int epoch = tiles_done == 1 ? 0 : 1;
#pragma unroll
                    for (int t = 0; t < NQ; ++t) {
                        const float k0 = top[t].k[0];
                        if (k0 < INFINITY) atomicMin(&s_best[epoch * QT + 32 * t + r], sortable_u32(k0));
                    }
                }
                if (w == 0 && (tiles_done == 2 || tiles_done == 5) && lane < QT) {   // workgroup -> its slot
                    const int epoch = tiles_done == 2 ? 0 : 1;
                    const uint32_t v = s_best[epoch * QT + lane];
                    if (v != 0xFFFFFFFFu)
                        (void)__hip_atomic_fetch_min(a.g_slot + (lane * 2 + epoch) * 32 + (blockIdx.x % KC), v,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // polls: epoch 0 at tiles 3..5, epoch 1 at tiles 6..9, one wave each (spreads the stall)
                const int poll = tiles_done - 3;
                if (poll >= 0 && poll < 7 && w == poll && lane < QT) {
                    const int epoch = poll < 3 ? 0 : 1;
                    const uint32_t* sl = a.g_slot + (lane * 2 + epoch) * 32;
                    uint32_t m = 0u;
#pragma unroll
                    for (int s2 = 0; s2 < KC; ++s2) {
                        const uint32_t v = __hip_atomic_load(sl + s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        m = v > m ? v : m;                       // max over the KC slots
                    }
                    if (m < s_tau[lane]) atomicMin(&s_tau[lane], m);
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
        advance(tile_cur, c_cur);
    };

    const int n_it = n_my * NCH;
    if constexpr (ONE_SET) {
        if (n_it > 0) {
            issue(ldA, tile_nx, c_nx);
            advance(tile_nx, c_nx);
            for (int it = 0; it < n_it; ++it) body(ldA);
        }
    } else if (n_it > 0) {
        issue(ldA, tile_nx, c_nx);
        advance(tile_nx, c_nx);
        issue(ldB, tile_nx, c_nx);
        advance(tile_nx, c_nx);
        int it = 0;
        for (; it + 1 < n_it; it += 2) {
            body(ldA);
            body(ldB);
        }
        if (it < n_it) body(ldA);
    }

    // ---- merge the 16 per-lane lists of every query inside the workgroup -----------
    // (LDS is free now: one query tile at a time, 16 lists x 32 queries x KC x 8 B)
    unsigned long long* s_m = reinterpret_cast<unsigned long long*>(smem);
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        __syncthreads();
        {
            unsigned long long* dst = s_m + ((size_t)(w * 2 + hh) * 32 + r) * KC;
#pragma unroll
            for (int j = 0; j < KC; ++j) dst[j] = pack_key(top[t].k[j], top[t].i[j]);
        }
        __syncthreads();
        // 16 consecutive lanes = the 16 source lists of one query; KC rounds of 16-way min
        const int q = tid >> 4, src = tid & 15;
        const unsigned long long* mine = s_m + ((size_t)src * 32 + q) * KC;
        int head = 0;
        unsigned long long cur = mine[0];
        const int64_t o = ((int64_t)blockIdx.x * QT + 32 * t + q) * KC;
        for (int round = 0; round < KC; ++round) {
            const unsigned long long m = group_min16_u64(cur);
            if (cur == m && (uint32_t)m != (uint32_t)kIdxSentinel) {  // unique owner pops
                ++head;
                cur = head < KC ? mine[head] : ~0ull;
            }
            if (src == 0) {
                a.out_idx[o + round] = (int)(uint32_t)m;
                // undo the monotone mapping for the key
                a.out_key[o + round] = unsortable_f32((uint32_t)(m >> 32));
            }
        }
    }
}

template <int QT, int KC, bool F32, bool HP>
__global__ __launch_bounds__(512, 1) void scan_topk_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gate_closed(a.gate)) return;
    scan_topk_body<QT, KC, F32, HP>(a, smem);
}

// Re-run of the query groups (QT queries each) whose candidate buffers overflowed in the
// MFMA-tiled scan (flat_mm.hip): per-lane lists cannot overflow.  One launch walks all groups
// and skips the clean ones, so the common case costs one empty kernel.  Group g writes its
// lists at out_* + g * part_stride.
template <int QT, int KC>
__global__ __launch_bounds__(512, 1) void scan_topk_flagged_kernel(ScanArgs a, const uint32_t* __restrict__ q_flag,
                                                                   int n_groups, int64_t part_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_any;
    if (gate_closed(a.gate)) return;
    if (q_flag[(int64_t)n_groups * QT] == 0) return;  // the "any query flagged" word: the common case
    for (int g = 0; g < n_groups; ++g) {
        if (threadIdx.x == 0) s_any = 0;
        __syncthreads();
        if (threadIdx.x < QT && q_flag[g * QT + threadIdx.x] != 0) s_any = 1;
        __syncthreads();
        const int any = s_any;
        __syncthreads();
        if (!any) continue;
        ScanArgs b = a;
        b.q16 = a.q16 + (int64_t)g * QT * a.d;
        b.g_tau = a.g_tau + g * QT;
        b.g_slot = nullptr;
        b.out_key = a.out_key + g * part_stride;
        b.out_idx = a.out_idx + g * part_stride;
        scan_topk_body<QT, KC, false, false>(b, smem);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// query-stationary scan for large query batches (fp16 rows): 128 queries per pass.
// Eight waves (two per SIMD, so one wave's LDS/barrier latency hides under the
// other's MFMAs) each keep 16 queries as v_mfma_f32_16x16x32_f16 B fragments in
// registers (d/32 x 4 VGPRs); the workgroup stages 128-row x 64-element chunks of
// the corpus ONCE through a double-buffered LDS tile that all waves read, so one
// HBM pass serves 128 queries.  Loads run 4 chunks ahead in four named register
// sets (static rotation: d/64 is a multiple of 4), one raw barrier per chunk.
// ---------------------------------------------------------------------------
// No ordinary VMEM load or returning atomic may sit inside the scan loop: its result can only be
// waited for with s_waitcnt vmcnt(0)-like counts that drain the DMA ring (loads return in order).
// The per-group side data - row norms (L2) and the chip-wide bounds - therefore travel by LDS-DMA
// too (two 512-B transfers per group, issued by waves 0 and 1 a whole group ahead of their use).
// (the corpus chunks are read once: non-temporal LDS-DMA, aux = 2, like the other scans' streams)
#ifdef PRAG_SCAN_PLAIN_LOADS
constexpr int kQsAux = 0;
#else
constexpr int kQsAux = 2;
#endif
template <int NKS /* d/32 */, int KC, bool L2>
__global__ __launch_bounds__(512, 2) void scan_qs_kernel(ScanArgs a) {
    constexpr int kSideXn = 4 * 128 * 128;          // LDS: [128] f32 row norms of the running group
    constexpr int kSideTau = 4 * 128 * 128 + 1024;  //      [128] sortable-uint bounds (one group old)
    constexpr int NCH = NKS / 2;  // 64-element chunks per row
    static_assert(NCH % 4 == 0, "register-set rotation needs d % 256 == 0");
    constexpr int DG = 128;       // rows per workgroup step
    constexpr int STAGE = DG * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gate_closed(a.gate)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g4 = lane >> 4;
    const int d = NKS * 32;

    // ---- this wave's 16 queries -> registers (B operand: k = 32s + 8*g4 + j) --------
    half8 qf[NKS];
    {
        const _Float16* qrow = a.q16 + (int64_t)(16 * w + r16) * d + 8 * g4;
#pragma unroll
        for (int s = 0; s < NKS; ++s) qf[s] = *reinterpret_cast<const half8*>(qrow + 32 * s);
    }

    const int n_groups = (int)((a.N + DG - 1) / DG);
    const int nWG = gridDim.x;
    const int n_my = (int)blockIdx.x < n_groups ? (n_groups - (int)blockIdx.x + nWG - 1) / nWG : 0;

    // LDS-DMA staging (no staging registers): one wave-instruction writes 1 KiB of LDS
    // linearly (lane x 16 B), so the XOR swizzle is applied to the per-lane SOURCE address:
    // wave w, piece j covers rows 8*(2w+j) .. +8 of the [128 rows x 128 B] chunk; lane l
    // fills slot l&7 of row 8*(2w+j) + (l>>3) with global piece (slot ^ ((row>>1)&7)).
    typedef const __attribute__((address_space(1))) char* gcptr;
    const gcptr rows = (gcptr) reinterpret_cast<const char*>(a.rows);
    const int64_t row_bytes = (int64_t)d * 2;
    gcptr p_cur[2];
    gcptr p_nxt[2];
    // (per-lane staging geometry is rebuilt from an opaque copy of the lane id once per group: kept
    // live across the unrolled chunk loop it is spilled, and a spill reload waits behind
    // s_waitcnt vmcnt(0) - a full drain of the DMA ring twice per group)
    auto group_ptrs = [&](gcptr (&ptr)[2], int group) {
        int le = lane;
        asm volatile("" : "+v"(le));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int doc = 8 * (2 * w + j) + (le >> 3);
            const int col = (((le & 7) ^ ((doc >> 1) & 7)) << 4);
            int64_t row = (int64_t)group * DG + doc;
            row = row < a.N ? row : a.N - 1;
            ptr[j] = rows + row * row_bytes + col;
        }
    };
    typedef __attribute__((address_space(3))) char* lptr;
    const lptr lds0 = (lptr)smem;
    // chunk c of the group behind `ptr` -> ring stage `stg` (c, stg compile-time)
#define PRAG_DMA(ptr, c_, stg_)                                                                          \
    {                                                                                                   \
        __builtin_amdgcn_global_load_lds(ptr[0] + (c_) * 128, lds0 + (stg_) * STAGE + (2 * w) * 1024, 16, 0, kQsAux);     \
        __builtin_amdgcn_global_load_lds(ptr[1] + (c_) * 128, lds0 + (stg_) * STAGE + (2 * w + 1) * 1024, 16, 0, kQsAux); \
    }

    TopList<KC> top;
    top.init();
    uint32_t gtau = a.g_tau[16 * w + r16];  // +inf, or the pre-pass bound
    const gcptr xn_g = (gcptr) reinterpret_cast<const char*>(a.xnorm);
    const gcptr tau_g = (gcptr) reinterpret_cast<const char*>(a.g_tau);
    f32x4 acc[8];  // 8 tiles of 16 rows; C layout: query = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // swizzled fragment offsets: row = 16t + r16, so the swizzle term ((row>>1)&7) does not depend
    // on the tile t: address = base[s2] + t*2048 (+ stage offset), both folded into ds_read immediates.
    // k-step 1 is k-step 0 with byte-bit 6 flipped ((4 | g4) ^ sw == 4 ^ (g4 ^ sw)).
    const int a_base0 = r16 * 128 + ((g4 ^ ((r16 >> 1) & 7)) << 4);
    const int a_base1 = a_base0 ^ 64;

    if (n_my > 0) {
        const int g0 = blockIdx.x;
        group_ptrs(p_cur, g0);
        group_ptrs(p_nxt, g0 + nWG);
        // prologue: chunks 0..2 in flight into ring stages 0..2
        PRAG_DMA(p_cur, 0, 0)
        PRAG_DMA(p_cur, 1, 1)
        PRAG_DMA(p_cur, 2, 2)

        for (int gi = 0; gi < n_my; ++gi) {
            const int g = g0 + gi * nWG;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                // chunk `c` has landed once all but the 2 younger chunks (4 DMAs) are done; the
                // barrier makes every wave's part visible and frees stage (c+3)%4 (read last step)
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (c + 3 < NCH) PRAG_DMA(p_cur, (c + 3) % NCH, (c + 3) % 4)
                else PRAG_DMA(p_nxt, (c + 3) % NCH, (c + 3) % 4)
                if (c == 0 && w < 2 && (L2 || w == 1)) {
                    // side data of THIS group (every wave is past the previous group's epilogue now)
                    int le = lane;
                    asm volatile("" : "+v"(le));
                    if (le < 32) {
                        const gcptr src = w == 0 ? xn_g + ((int64_t)g * DG) * 4 + le * 16 : tau_g + le * 16;
                        const lptr dst = lds0 + (w == 0 ? kSideXn : kSideTau);
                        __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
                    }
                }

                const char* xs = smem + (c % 4) * STAGE;
                // four fragment reads in flight at a time (LDS latency would otherwise be paid per MFMA)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int tb = 0; tb < 8; tb += 4) {
                        half8 av[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            av[u] = *reinterpret_cast<const half8*>(xs + (s2 ? a_base1 : a_base0) + (tb + u) * 2048);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            acc[tb + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[u], qf[2 * c + s2], acc[tb + u], 0, 0, 0);
                        // pin the order: all four LDS reads first (their latencies overlap), then the MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
            // next group's pointers become current; compute the one after
            p_cur[0] = p_nxt[0];
            p_cur[1] = p_nxt[1];
            group_ptrs(p_nxt, g + 2 * nWG);
            // ---- epilogue: 8 x 4 rows against this lane's query -----------------------
            int le = lane;
            asm volatile("" : "+v"(le));
            const int g4 = le >> 4;   // (shadows the loop-invariant copy on purpose, see group_ptrs)
            const int64_t doc0 = (int64_t)g * DG;
            // the chip-wide bound as of the start of this group (LDS copy; a stale value is only a
            // looser bound - the word never increases)
            gtau = *reinterpret_cast<const uint32_t*>(smem + kSideTau + (16 * w + (le & 15)) * 4);
            float tau = top.k[KC - 1];
            tau = fminf(tau, __shfl_xor(tau, 16, 64));
            tau = fminf(tau, __shfl_xor(tau, 32, 64));
            tau = fminf(tau, unsortable_f32(gtau));
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                f32x4 nv = {0.f, 0.f, 0.f, 0.f};
                if constexpr (L2) nv = *reinterpret_cast<const f32x4*>(smem + kSideXn + (16 * t + 4 * g4) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t doc = doc0 + 16 * t + 4 * g4 + e;
                    const float key = doc < a.N ? fmaf(a.alpha, acc[t][e], nv[e]) : INFINITY;
                    top.push(key, (int)doc, tau);
                }
                acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if ((gi & 3) == 3) {  // every 4th group: publish this query's bound (no value comes back)
                float mine_f = top.k[KC - 1];
                mine_f = fminf(mine_f, __shfl_xor(mine_f, 16, 64));
                mine_f = fminf(mine_f, __shfl_xor(mine_f, 32, 64));
                const uint32_t mine = sortable_u32(mine_f);
                if (g4 == 0 && mine < gtau)
                    (void)__hip_atomic_fetch_min(a.g_tau + 16 * w + (le & 15), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail DMAs before LDS is reused
    }
#undef PRAG_DMA

    // ---- merge the four row-quarter lists of every query: one list per (wg, query) ------
    __syncthreads();
    unsigned long long* s_m = reinterpret_cast<unsigned long long*>(smem);
    {
        unsigned long long* dst = s_m + ((size_t)(16 * w + r16) * 4 + g4) * KC;
#pragma unroll
        for (int j = 0; j < KC; ++j) dst[j] = pack_key(top.k[j], top.i[j]);
    }
    __syncthreads();
    {
        const int q = tid >> 2, src = tid & 3;
        const unsigned long long* mine = s_m + ((size_t)q * 4 + src) * KC;
        int head = 0;
        unsigned long long cur = mine[0];
        const int64_t o = ((int64_t)blockIdx.x * 128 + q) * KC;
        for (int round = 0; round < KC; ++round) {
            const unsigned long long m = group_min4_u64(cur);
            if (cur == m && (uint32_t)m != (uint32_t)kIdxSentinel) {
                ++head;
                cur = head < KC ? mine[head] : ~0ull;
            }
            if (src == 0) {
                a.out_idx[o + round] = (int)(uint32_t)m;
                a.out_key[o + round] = unsortable_f32((uint32_t)(m >> 32));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// merge the per-lane lists of one query tile: grid = queries, 256 threads
// ---------------------------------------------------------------------------

// KC smallest (key,id) pairs of one query out of n_lists sorted lists of KC (sentinel-terminated).
// Latency-lean: no per-thread lists, no selection rounds.
//   1. the list heads go to LDS; tau = the KC-th smallest head (rank by counting).  KC heads are <= tau,
//      so the KC-th best of the union is <= tau, and only the KC lists whose head is <= tau can hold
//      anything <= tau: at most KC*KC entries survive (usually KC plus a few);
//   2. every entry <= tau is appended to an LDS buffer (atomic counter);
//   3. the survivors are ranked by counting (row ids are unique, so packed values are).
// Result: s_out[rank] = packed (key, id) for rank < KC, ~0 where there are fewer than KC rows.
// All NT threads of the block; s_head [kMergeMaxLists], s_surv [KC*KC], s_misc [4].
constexpr int kMergeMaxLists = 1024;
template <int KC, int NT>
__device__ __forceinline__ void merge_lists_block(const float* __restrict__ part_key, const int* __restrict__ part_idx,
                                                  int n_lists, int QT, int q, unsigned long long* s_head,
                                                  unsigned long long* s_surv, unsigned long long* s_misc,
                                                  unsigned long long* s_out /*[KC]*/) {
    const int tid = threadIdx.x;
    const unsigned long long kInf = ~0ull;
    int* s_cnt = reinterpret_cast<int*>(s_misc + 1);
    for (int l = tid; l < n_lists; l += NT) {
        const int64_t o = ((int64_t)l * QT + q) * KC;
        const int idx = part_idx[o];
        s_head[l] = idx == kIdxSentinel ? kInf : pack_key(part_key[o], idx);
    }
    if (tid == 0) {
        s_misc[0] = kInf;   // tau: everything passes unless KC heads exist
        *s_cnt = 0;
    }
    if (tid < KC) s_out[tid] = kInf;
    __syncthreads();
    for (int l = tid; l < n_lists; l += NT) {
        const unsigned long long h = s_head[l];
        if (h != kInf) {
            int rank = 0;
            for (int j = 0; j < n_lists; ++j) rank += s_head[j] < h;   // broadcast reads
            if (rank == KC - 1) s_misc[0] = h;
        }
    }
    __syncthreads();
    const unsigned long long tau = s_misc[0];
    const int total = n_lists * KC;
    for (int e = tid; e < total; e += NT) {
        const int l = e / KC, j = e - l * KC;
        if (s_head[l] <= tau && s_head[l] != kInf) {       // only KC lists get past this
            const int64_t o = ((int64_t)l * QT + q) * KC + j;
            const int idx = part_idx[o];
            if (idx != kIdxSentinel) {
                const unsigned long long v = pack_key(part_key[o], idx);
                if (v <= tau) s_surv[atomicAdd(s_cnt, 1)] = v;
            }
        }
    }
    __syncthreads();
    const int S = *s_cnt;
    for (int e = tid; e < S; e += NT) {
        const unsigned long long v = s_surv[e];
        int rank = 0;
        for (int j = 0; j < S; ++j) rank += s_surv[j] < v;
        if (rank < KC) s_out[rank] = v;
    }
    __syncthreads();
}

template <int KC>
__global__ __launch_bounds__(256) void merge_lists_kernel(const float* __restrict__ part_key,
                                                         const int* __restrict__ part_idx, int n_lists,
                                                         int QT, int* __restrict__ cand_idx /*[q][KC]*/,
                                                         uint32_t* __restrict__ tau_out /*[q] or null*/,
                                                         const uint32_t* __restrict__ q_flag /*or null*/,
                                                         int64_t part_stride, int any_idx, Gate gate) {
    __shared__ unsigned long long s_head[kMergeMaxLists];
    __shared__ unsigned long long s_surv[KC * KC];
    __shared__ unsigned long long s_misc[4];
    __shared__ unsigned long long s_out[KC];
    if (gate_closed(gate)) return;
    // flagged mode (fallback of the MFMA-tiled scan): block = global query, lists of its group of QT
    // queries start at part_stride * group; groups without a flagged query keep their candidates
    int q = blockIdx.x;
    const int64_t qglob = blockIdx.x;
    if (q_flag) {
        if (q_flag[any_idx] == 0) return;  // nothing was flagged in this search
        const int grp = q / QT;
        bool any = false;
        for (int j = 0; j < QT; ++j) any |= q_flag[grp * QT + j] != 0;
        if (!any) return;
        part_key += grp * part_stride;
        part_idx += grp * part_stride;
        q -= grp * QT;
    }
    merge_lists_block<KC, 256>(part_key, part_idx, n_lists, QT, q, s_head, s_surv, s_misc, s_out);
    if ((int)threadIdx.x < KC) {
        const unsigned long long v = s_out[threadIdx.x];
        cand_idx[qglob * KC + threadIdx.x] = (v == ~0ull) ? -1 : (int)(uint32_t)v;
        // pre-pass use: the KC-th best key of the rows seen = a valid pruning bound for the
        // full scan (those rows are a subset of the shard)
        if (tau_out && threadIdx.x == KC - 1 && v != ~0ull) tau_out[qglob] = (uint32_t)(v >> 32);
    }
}

// ---------------------------------------------------------------------------
// exact rerank of one query's KC candidates: one wave per candidate, 16-B row loads, float64
// ---------------------------------------------------------------------------
template <bool F32, typename CandFn>
__device__ __forceinline__ void rerank_block(const void* __restrict__ rows, int d, int metric_l2,
                                             const float* __restrict__ q, CandFn cand, int KC, int k,
                                             int64_t id_offset, float* __restrict__ Db, int64_t* __restrict__ Ib,
                                             double* s_score /*[64]*/, int* s_idx /*[64]*/, const CertArgs& cert,
                                             int b, float kth_sel /* selection key of the KC-th candidate, +inf: list not full */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    for (int c = w; c < KC; c += nw) {
        const int idx = cand(c);
        double s = 0.0;
        if (idx >= 0) {
            // 8 consecutive elements per lane per step (d is a multiple of 64)
            for (int e = lane * 8; e < d; e += 512) {
                float xv[8];
                if constexpr (F32) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(rows) + (int64_t)idx * d + e);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(rows) + (int64_t)idx * d + e + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xv[j] = a0[j]; xv[4 + j] = a1[j]; }
                } else {
                    const half8 h = *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(rows) + (int64_t)idx * d + e);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = (float)h[j];
                }
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(q + e);
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(q + e + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double qv = (double)(j < 4 ? q0[j] : q1[j - 4]);
                    const double x = (double)xv[j];
                    if (metric_l2) {
                        const double t = qv - x;
                        s = fma(t, t, s);
                    } else {
                        s = fma(qv, x, s);
                    }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        }
        if (lane == 0) {
            s_score[c] = s;
            s_idx[c] = idx;
        }
    }
    __syncthreads();
    // order by (score better, id lower), invalid (-1) entries last in candidate order: every candidate
    // finds its rank by counting (KC <= 32 threads, no serial sort)
    if ((int)threadIdx.x < KC) {
        const int c = threadIdx.x;
        const double sv = s_score[c];
        const int iv = s_idx[c];
        int rank = 0;
        for (int j = 0; j < KC; ++j) {
            const double sj = s_score[j];
            const int ij = s_idx[j];
            bool before;  // does (sj,ij) go before (sv,iv)?
            if (ij < 0) before = iv < 0 && j < c;
            else if (iv < 0) before = true;
            else if (sj != sv) before = metric_l2 ? (sj < sv) : (sj > sv);
            else before = ij < iv;
            rank += (j != c) && before;
        }
        const bool ok = iv >= 0;
        if (rank < k) {
            Db[rank] = ok ? (float)sv : (metric_l2 ? FLT_MAX : -FLT_MAX);
            Ib[rank] = ok ? tag_id((int64_t)iv + id_offset, sv, cert.tag_ids) : -1;
        }
        // exactness certificate (the k-th result's owner): can a row outside the candidates still
        // belong to the top k?  Fewer than k valid candidates = the list is not full = every row is in it.
        if (rank == k - 1 && !cert_ok(cert, b, metric_l2, ok ? sv : 0.0, ok ? kth_sel : INFINITY)) cert_flag(cert, b);
    }
    for (int j = KC + (int)threadIdx.x; j < k; j += blockDim.x) {   // k > KC never happens on these paths; pad anyway
        Db[j] = metric_l2 ? FLT_MAX : -FLT_MAX;
        Ib[j] = -1;
    }
}

template <bool F32>
__global__ __launch_bounds__(1024) void rerank_kernel(const void* __restrict__ rows, int d, int metric_l2,
                                                     const float* __restrict__ q32,
                                                     const int* __restrict__ cand_idx, int KC, int k,
                                                     int64_t id_offset, float* __restrict__ D,
                                                     int64_t* __restrict__ I, CertArgs cert,
                                                     const uint32_t* __restrict__ kth_sel /*[B] sortable, or null*/) {
    __shared__ double s_score[64];
    __shared__ int s_idx[64];
    if (gate_closed(cert.gate)) return;
    const int b = blockIdx.x;
    rerank_block<F32>(rows, d, metric_l2, q32 + (int64_t)b * d, [&](int c) { return cand_idx[(int64_t)b * KC + c]; }, KC,
                      k, id_offset, D + (int64_t)b * k, I + (int64_t)b * k, s_score, s_idx, cert, b,
                      kth_sel ? unsortable_f32(kth_sel[b]) : INFINITY);
}

// Deep candidate lists (k > 26 -> KC up to 1024, tiled scan only): exact float64 scores by one
// wave per candidate, then a bitonic sort of (order-preserving score bits, id) pairs in LDS.
template <bool F32>
__global__ __launch_bounds__(1024) void rerank_sort_kernel(const void* __restrict__ rows, int d, int metric_l2,
                                                          const float* __restrict__ q32,
                                                          const int* __restrict__ cand_idx, int KC, int k,
                                                          int64_t id_offset, float* __restrict__ D,
                                                          int64_t* __restrict__ I, CertArgs cert,
                                                          const uint32_t* __restrict__ kth_sel /*[B] sortable*/) {
    __shared__ unsigned long long s_key[kMmMaxKc];
    __shared__ int s_id[kMmMaxKc];
    if (gate_closed(cert.gate)) return;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* q = q32 + (int64_t)b * d;
    int n_pad = 2;
    while (n_pad < KC) n_pad <<= 1;
    // (four candidates of a wave in flight at once - ids, then rows, requested before the first product - were
    // measured: 152.7 us against 144 for 1000 queries x 256 candidates; the scattered 1.5-KB row reads, 2.6 TB/s,
    // bound this loop, not its latency chain)
    for (int c = w; c < n_pad; c += 16) {
        const int idx = c < KC ? cand_idx[(int64_t)b * KC + c] : -1;
        double s = 0.0;
        if (idx >= 0) {
            for (int e = lane * 8; e < d; e += 512) {
                float xv[8];
                if constexpr (F32) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(rows) + (int64_t)idx * d + e);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(rows) + (int64_t)idx * d + e + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xv[j] = a0[j]; xv[4 + j] = a1[j]; }
                } else {
                    const half8 h = *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(rows) + (int64_t)idx * d + e);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = (float)h[j];
                }
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(q + e);
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(q + e + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double qv = (double)(j < 4 ? q0[j] : q1[j - 4]);
                    const double x = (double)xv[j];
                    if (metric_l2) {
                        const double t = qv - x;
                        s = fma(t, t, s);
                    } else {
                        s = fma(qv, x, s);
                    }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        }
        if (lane == 0) {
            // ascending sort key: L2 smaller is better, inner product larger is better; invalid last
            s_key[c] = idx < 0 ? ~0ull : (metric_l2 ? sortable_u64(s) : ~sortable_u64(s));
            s_id[c] = idx < 0 ? 0x7fffffff : idx;
        }
    }
    // bitonic sort of (key, id) pairs, ascending
    for (int size = 2; size <= n_pad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n_pad >> 1); t += 1024) {
                const int i = ((t / stride) * 2 * stride) + (t % stride), j = i + stride;
                const unsigned long long ka = s_key[i], kb = s_key[j];
                const int ia = s_id[i], ib = s_id[j];
                const bool gt = ka > kb || (ka == kb && ia > ib);
                if (gt == ((i & size) == 0)) {
                    s_key[i] = kb; s_key[j] = ka;
                    s_id[i] = ib; s_id[j] = ia;
                }
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 1024) {
        const bool ok = j < KC && s_key[j] != ~0ull;
        const double sc = ok ? unsortable_f64(metric_l2 ? s_key[j] : ~s_key[j]) : 0.0;
        D[(int64_t)b * k + j] = ok ? (float)sc : (metric_l2 ? FLT_MAX : -FLT_MAX);
        I[(int64_t)b * k + j] = ok ? tag_id((int64_t)s_id[j] + id_offset, sc, cert.tag_ids) : -1;
    }
    if (threadIdx.x == 0) {
        const bool full = k <= KC && s_key[k - 1] != ~0ull;
        const double sk = full ? unsortable_f64(metric_l2 ? s_key[k - 1] : ~s_key[k - 1]) : 0.0;
        if (!cert_ok(cert, b, metric_l2, sk, full ? unsortable_f32(kth_sel[b]) : INFINITY)) cert_flag(cert, b);
    }
}

// merge of the per-workgroup lists + exact rerank of the survivors in one launch (the end of every
// search on the per-lane-list paths): grid = queries, 16 waves
template <int KC, bool F32>
__global__ __launch_bounds__(1024) void merge_rerank_kernel(const float* __restrict__ part_key,
                                                           const int* __restrict__ part_idx, int n_lists, int QT,
                                                           int q0, const void* __restrict__ rows, int d,
                                                           int metric_l2, const float* __restrict__ q32, int k,
                                                           int64_t id_offset, float* __restrict__ D,
                                                           int64_t* __restrict__ I, CertArgs cert) {
    __shared__ unsigned long long s_head[kMergeMaxLists];
    __shared__ unsigned long long s_surv[KC * KC];
    __shared__ unsigned long long s_misc[4];
    __shared__ unsigned long long s_out[KC];
    __shared__ double s_score[64];
    __shared__ int s_idx[64];
    if (gate_closed(cert.gate)) return;
    const int b = q0 + blockIdx.x;  // global query; blockIdx.x = its slot in this pass's query tile
    merge_lists_block<KC, 1024>(part_key, part_idx, n_lists, QT, (int)blockIdx.x, s_head, s_surv, s_misc, s_out);
    rerank_block<F32>(rows, d, metric_l2, q32 + (int64_t)b * d,
                      [&](int c) { return s_out[c] == ~0ull ? -1 : (int)(uint32_t)s_out[c]; }, KC, k, id_offset,
                      D + (int64_t)b * k, I + (int64_t)b * k, s_score, s_idx, cert, b,
                      s_out[KC - 1] == ~0ull ? INFINITY : unsortable_f32((uint32_t)(s_out[KC - 1] >> 32)));
}

template <bool F32>
__global__ __launch_bounds__(256) void reconstruct_kernel(const void* __restrict__ rows, int64_t n_elems,
                                                         float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_elems; i += stride) {
        if constexpr (F32) out[i] = reinterpret_cast<const float*>(rows)[i];
        else out[i] = (float)reinterpret_cast<const _Float16*>(rows)[i];
    }
}

__global__ __launch_bounds__(256) void rows_to_f16_kernel(const float* __restrict__ src, int64_t n_vec4,
                                                         _Float16* __restrict__ dst) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_vec4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        reinterpret_cast<half4*>(dst)[i] = half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    }
}

}  // namespace prag

// ---------------------------------------------------------------------------
// second tier of the 8-bit tiled selection when only a few queries failed its certificate: those queries are
// gathered into a compact batch, searched on their own (<= 128 queries: the two-level / list kernels), and their
// results scattered back over the rows of the batch
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_queries_kernel(const float* __restrict__ q, const int* __restrict__ list, int n,
                                                            int d, float* __restrict__ out, prag::Gate gate) {
    const int i = blockIdx.x;
    if (i >= n || prag::gate_closed(gate)) return;
    const int b = list[i];
    if (b < 0) {
        // padding slot of the retry tier: a fixed pseudo-random query (any ordinary query certifies like the unflagged
        // ones of the batch did; a COPY of a flagged query would be flagged again and cost an exact pass of its own)
        for (int c = threadIdx.x * 4; c < d; c += 1024) {
            prag::f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t h = (uint32_t)(i * 8191 + c + e) * 2654435761u;
                h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                v[e] = (float)(int32_t)h * (1.0f / 2147483648.0f);
            }
            *reinterpret_cast<prag::f32x4*>(out + (int64_t)i * d + c) = v;
        }
        return;
    }
    const float* src = q + (int64_t)b * d;
    for (int c = threadIdx.x * 4; c < d; c += 1024)
        *reinterpret_cast<prag::f32x4*>(out + (int64_t)i * d + c) = *reinterpret_cast<const prag::f32x4*>(src + c);
}
// rows [0, *n_dev) of the compact batch go back to their rows of the result (the rest were padding)
__global__ __launch_bounds__(64) void scatter_results_kernel(const float* __restrict__ Ds, const int64_t* __restrict__ Is,
                                                            const int* __restrict__ list, const uint32_t* __restrict__ n_dev, int k,
                                                            float* __restrict__ D, int64_t* __restrict__ I, prag::Gate gate) {
    const int i = blockIdx.x;
    if (prag::gate_closed(gate) || (uint32_t)i >= *n_dev) return;
    const int64_t b = list[i];
    for (int j = threadIdx.x; j < k; j += 64) {
        D[b * k + j] = Ds[(int64_t)i * k + j];
        I[b * k + j] = Is[(int64_t)i * k + j];
    }
}
// The tier decision, on the device: t2_word[0] = queries that failed the 8-bit certificate (the gates of the second
// tier compare it with ranges chosen by the host); the failed rows are copied out of the flag list (the inner
// searches reuse it) and the compact batch is padded to `cap` entries with its first row (duplicate queries, whose
// results are not scattered back).
__global__ __launch_bounds__(64) void mm8_tier_plan_kernel(const uint32_t* __restrict__ n_flag, const int* __restrict__ flag_list,
                                                          uint32_t* __restrict__ t2_word, int* __restrict__ t2_list, int cap) {
    const uint32_t n = *n_flag;
    if (threadIdx.x == 0) t2_word[0] = n;
    if (n == 0) return;
    const int first = flag_list[0];
    for (int i = threadIdx.x; i < cap; i += 64) t2_list[i] = (uint32_t)i < n ? flag_list[i] : first;
}

// Retry tier of the <= 128-query searches (round 5): the queries on the flag list - a candidate region of the two-level
// search overflowed, or the certificate of a 33-128-query direct scan did not clear - are searched again 32 at a time by
// the <= 32-query kernels (BOTH int8 query terms over the shadow / hi + lo fp16 terms over the rows: a far tighter
// filter and certificate) before anything goes to the exact float64 scan.  r2_word[0] = flagged count, r2_list = the
// flagged batch rows, padded with -1 (gather_queries_kernel writes a fixed pseudo-random query there) to a multiple of 32;
// [1] accumulates what the inner searches still flag.
__global__ __launch_bounds__(128) void retry_plan_kernel(uint32_t* __restrict__ n_flag, const int* __restrict__ flag_list,
                                                        uint32_t* __restrict__ r2_word, int* __restrict__ r2_list, int cap) {
    const uint32_t n = *n_flag;
    if (threadIdx.x == 0) {
        r2_word[0] = n;
        r2_word[1] = 0u;
    }
    if (n == 0) return;
    for (int i = threadIdx.x; i < cap; i += 128) r2_list[i] = (uint32_t)i < n ? flag_list[i] : -1;   // -1: padding query
}
__global__ __launch_bounds__(64) void retry_scatter_kernel(const float* __restrict__ Ds, const int64_t* __restrict__ Is,
                                                          const int* __restrict__ list, uint32_t* __restrict__ r2_word,
                                                          const uint32_t* __restrict__ n_flag_inner, int part0, int k,
                                                          float* __restrict__ D, int64_t* __restrict__ I, prag::Gate gate) {
    const int i = blockIdx.x;
    if (prag::gate_closed(gate)) return;
    if (i == 0 && threadIdx.x == 0) r2_word[1] += *n_flag_inner;     // (one part at a time on the stream: no race)
    if ((uint32_t)(part0 + i) >= r2_word[0]) return;                  // padding rows
    const int64_t b = list[part0 + i];
    for (int j = threadIdx.x; j < k; j += 64) {
        D[b * k + j] = Ds[(int64_t)i * k + j];
        I[b * k + j] = Is[(int64_t)i * k + j];
    }
}
__global__ void retry_finish_kernel(uint32_t* __restrict__ r2_word, uint32_t* __restrict__ n_flag) {
    r2_word[2] = r2_word[0] != 0u ? 1u : 0u;          // the tier did something in this search
    if (r2_word[0] != 0u) *n_flag = r2_word[1];       // what the exact scan recomputed in the end
}

// ===========================================================================
// host side
// ===========================================================================
using namespace prag;


static size_t elt(const prag_index* ix) { return ix->store == PRAG_F32 ? 4 : 2; }

static int ensure_capacity(prag_index* ix, int64_t want) {
    if (want <= ix->cap) return PRAG_OK;
    int64_t ncap = std::max<int64_t>(want, ix->cap + ix->cap / 2);
    ncap = (ncap + 255) / 256 * 256;  // the MFMA-tiled scan reads whole 256-row tiles
    void* nrows = nullptr;
    float* nnorm = nullptr;
    PRAG_HIP(hipDeviceSynchronize());   // searches still reading the old buffers on other streams
    PRAG_HIP(hipMalloc(&nrows, (size_t)ncap * ix->d * elt(ix)));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&nnorm), (size_t)ncap * sizeof(float));
    if (e == hipSuccess) e = hipMemset(nnorm, 0, (size_t)ncap * sizeof(float));
    if (e == hipSuccess && ix->ntotal > 0) {
        e = hipMemcpy(nrows, ix->rows, (size_t)ix->ntotal * ix->d * elt(ix), hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(nnorm, ix->xnorm, (size_t)ix->ntotal * sizeof(float), hipMemcpyDeviceToDevice);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {   // nothing leaks on the error path
        (void)hipFree(nrows);
        if (nnorm) (void)hipFree(nnorm);
        set_error("prag_index: growing to %lld rows failed: %s", (long long)ncap, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? PRAG_ENOMEM : PRAG_EHIP;
    }
    if (ix->rows) (void)hipFree(ix->rows);
    if (ix->xnorm) (void)hipFree(ix->xnorm);
    ix->rows = nrows;
    ix->xnorm = nnorm;
    ix->cap = ncap;
    return PRAG_OK;
}

// ---- 8-bit shadow maintenance ---------------------------------------------------------------------
constexpr int64_t kShadowMinRows = 1 << 20;

// Does this index keep a shadow at its current size / mode?  (mode 1: shards of >= 2^20 rows when the
// device has room for d + 12 more bytes per row with 2 GB to spare; mode 2: any size)
static bool shadow_wanted(prag_index* ix) {
    if (!ix->shadow_mode || ix->ntotal == 0 || !shadow_store_supported(ix->d) || ix->shadow_failed) return false;
    if (ix->shadow_mode >= 2) return true;
    if (ix->ntotal < kShadowMinRows || ix->shadow_no_room) return false;
    if (ix->shadow_cap < ix->cap) {
        size_t free_b = 0, total_b = 0;
        const size_t have = ix->rows8 ? (size_t)ix->shadow_cap * (ix->d + 12) : 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess ||
            free_b + have < (size_t)ix->cap * (ix->d + 12) + ((size_t)2 << 30)) {
            ix->shadow_no_room = true;
            return false;
        }
    }
    return true;
}

// Bring the shadow up to date with the stored rows (allocation follows the row capacity; rows added since
// the last call are quantised; the max ||x||^2 word the shadow's error constants read is refreshed).
// Called at the end of every add (so that no search pays for it), from prag_index_prepare, and - a no-op
// then - from the search itself.
static int shadow_ensure(prag_index* ix, hipStream_t st) {
    if (ix->xn_max_rows < ix->ntotal) {  // rows added since the last refresh
        const int blocks = (int)std::min<int64_t>(1024, (ix->ntotal - ix->xn_max_rows + 255) / 256);
        hipLaunchKernelGGL(xnorm_max_kernel, dim3(blocks), dim3(256), 0, st, ix->xnorm, ix->xn_max_rows, ix->ntotal,
                           ix->cert_words + 1);
        PRAG_LAUNCH_CHECK();
        ix->xn_max_rows = ix->ntotal;
    }
    if (!shadow_wanted(ix)) {
        if ((ix->shadow_no_room || ix->shadow_failed) && ix->rows8) {   // an undersized shadow nobody will read again
            PRAG_HIP(hipStreamSynchronize(st));                          // (a search may still be reading it)
            for (void* p : {(void*)ix->rows8, (void*)ix->sscale, (void*)ix->serr, (void*)ix->sbias})
                if (p) (void)hipFree(p);
            ix->rows8 = nullptr; ix->sscale = nullptr; ix->serr = nullptr; ix->sbias = nullptr; ix->shadow_cap = 0; ix->shadow_rows = 0;
        }
        return PRAG_OK;
    }
    if (ix->shadow_cap < ix->cap) {    // (re)allocate with the rows; rebuilt from row 0
        PRAG_HIP(hipStreamSynchronize(st));   // a search may still be reading the old shadow on this stream
        for (void* p : {(void*)ix->rows8, (void*)ix->sscale, (void*)ix->serr, (void*)ix->sbias})
            if (p) (void)hipFree(p);
        ix->rows8 = nullptr; ix->sscale = nullptr; ix->serr = nullptr; ix->sbias = nullptr; ix->shadow_cap = 0; ix->shadow_rows = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ix->rows8), (size_t)ix->cap * ix->d);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ix->sscale), (size_t)ix->cap * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ix->serr), (size_t)ix->cap * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ix->sbias), (size_t)ix->cap * sizeof(float));
        if (e == hipSuccess && !ix->shadow_err_max) {
            e = hipMalloc(reinterpret_cast<void**>(&ix->shadow_err_max), sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemsetAsync(ix->shadow_err_max, 0, sizeof(uint32_t), st);
        }
        if (e == hipSuccess && !ix->sh_aff) {
            e = hipMalloc(reinterpret_cast<void**>(&ix->sh_aff), (size_t)3 * ix->d * sizeof(float));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ix->sh_aff_sums), (size_t)2 * ix->d * sizeof(double));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ix->sh_yn_max), sizeof(uint32_t));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ix->sh_bias_max), sizeof(uint32_t));
        }
        if (e != hipSuccess) {   // nothing half-allocated is left behind; rows are scanned directly from now on
            for (void* p : {(void*)ix->rows8, (void*)ix->sscale, (void*)ix->serr, (void*)ix->sbias})
                if (p) (void)hipFree(p);
            ix->rows8 = nullptr; ix->sscale = nullptr; ix->serr = nullptr; ix->sbias = nullptr;
            (void)hipGetLastError();
            if (ix->shadow_mode == 1) { ix->shadow_no_room = true; return PRAG_OK; }
            set_error("prag_index: allocating the 8-bit shadow failed: %s", hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? PRAG_ENOMEM : PRAG_EHIP;
        }
        ix->shadow_cap = ix->cap;
    }
    if (ix->shadow_rows < ix->ntotal) {
        ShadowStore ss;
        ss.rows = ix->rows;
        ss.store_f32 = ix->store == PRAG_F32;
        ss.d = ix->d;
        ss.rows8 = ix->rows8;
        ss.sscale = ix->sscale;
        ss.serr = ix->serr;
        ss.err_max = ix->shadow_err_max;
        ss.aff = ix->sh_aff;
        ss.yn_max = ix->sh_yn_max;
        ss.sbias = ix->sbias;
        ss.bias_max = ix->sh_bias_max;
        ss.xnorm_l2 = ix->metric == PRAG_METRIC_L2 ? ix->xnorm : nullptr;
        ss.alpha = ix->metric == PRAG_METRIC_L2 ? -2.0f : -1.0f;
        if (ix->shadow_rows == 0) {
            PRAG_HIP(hipMemsetAsync(ix->sh_bias_max, 0, sizeof(uint32_t), st));
            // a shadow built from its first row: fit the affine map to the rows that are there (frozen for the rows
            // added later - any map is valid, the fit only decides how tight the filter is)
            PRAG_HIP(hipMemsetAsync(ix->sh_yn_max, 0, sizeof(uint32_t), st));
            PRAG_HIP(hipMemsetAsync(ix->shadow_err_max, 0, sizeof(uint32_t), st));
            const int rc_a = shadow_affine_fit(ss, ix->ntotal, ix->shadow_affine_mode == 0 ? 1 : ix->shadow_affine_mode == 1 ? 2 : 0, ix->sh_aff_sums, st);
            if (rc_a != PRAG_OK) return rc_a;
        }
        const int rc = shadow_build(ss, ix->shadow_rows, ix->ntotal, st);
        if (rc != PRAG_OK) return rc;
        ix->shadow_rows = ix->ntotal;
        ix->mm8_whole_batch_streak = 0;     // new rows: the 8-bit tiles get another chance
        ix->mm8_auto_off = false;
        ix->mm8_off_count = 0;
        ix->mm8_off_period = 64;
    }
    return PRAG_OK;
}

// At the end of an add the rows are already committed: a shadow that cannot follow them (allocation, build) must not
// make the add fail - a caller that retried would add the rows twice.  The index scans its rows directly from then on
// (shadow_failed; prag_index_set_shadow clears it), the reason stays readable through prag_last_error.
static int shadow_after_add(prag_index* ix, hipStream_t st) {
    const int rc = shadow_ensure(ix, st);
    if (rc != PRAG_OK) {
        (void)hipGetLastError();
        ix->shadow_failed = true;
        (void)shadow_ensure(ix, st);      // frees what is left of the shadow
    }
    return PRAG_OK;
}

extern "C" int prag_index_prepare(prag_index_t* ix, void* stream) {
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    return shadow_ensure(ix, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int prag_index_create(prag_index_t** out, int d, int metric, int store_dtype,
                                 int64_t capacity_rows) {
    PRAG_REQUIRE(out != nullptr, PRAG_EINVAL, "prag_index_create: out is NULL");
    PRAG_REQUIRE(d >= 64 && d % 64 == 0 && d <= 1536, PRAG_EUNSUPPORTED,
                 "d=%d: the scan kernel needs a multiple of 64 in [64,1536]", d);
    PRAG_REQUIRE(metric == PRAG_METRIC_L2 || metric == PRAG_METRIC_IP || metric == PRAG_METRIC_COS,
                 PRAG_EINVAL, "metric=%d", metric);
    PRAG_REQUIRE(store_dtype == PRAG_F32 || store_dtype == PRAG_F16, PRAG_EINVAL, "store_dtype=%d",
                 store_dtype);
    PRAG_REQUIRE(capacity_rows >= 0, PRAG_EINVAL, "capacity_rows < 0");
    prag_index* ix = new (std::nothrow) prag_index();
    PRAG_REQUIRE(ix != nullptr, PRAG_ENOMEM, "out of host memory");
    ix->d = d;
    ix->metric = metric;
    ix->store = store_dtype;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        ix->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // Performance-only switches (every setting returns the exact result): PRAG_SCAN_MM=0 never takes the
    // MFMA-tiled scan, PRAG_PREPASS, PRAG_SHADOW.  The switches that trade exactness for a timing
    // experiment (certificate off, tiled-scan overflow left unrepaired) exist only in `make diag`.
    if (const char* e = getenv("PRAG_SCAN_MM")) ix->mm_mode = atoi(e) != 0;
    if (const char* e = getenv("PRAG_SHADOW_AFFINE")) ix->shadow_affine_mode = atoi(e);
    if (const char* e = getenv("PRAG_RETRY_TIER")) ix->retry_mode = atoi(e);
    if (const char* e = getenv("PRAG_EXACT_GROUP")) ix->exact_group_mode = atoi(e);
    if (const char* e = getenv("PRAG_EXACT_MFMA")) ix->exact_mfma_mode = atoi(e);
    if (const char* e = getenv("PRAG_GATHER")) ix->gather_mode = atoi(e);
#ifdef PRAG_MM_DIAG
    if (const char* e = getenv("PRAG_SCAN_MM")) ix->mm_mode = atoi(e);
    if (const char* e = getenv("PRAG_CERT")) ix->cert_mode = atoi(e);
#endif
    if (const char* e = getenv("PRAG_MM_SHAPE")) ix->mm_shape16 = atoi(e) != 32;
    if (const char* e = getenv("PRAG_SHADOW_BOUND")) ix->shadow_bound_mode = atoi(e);
    if (const char* e = getenv("PRAG_SHADOW_SAMPLE")) ix->shadow_sample_mode = atoi(e);
    if (const char* e = getenv("PRAG_SCAN_GATE")) ix->scan_gate_mode = atoi(e);
    if (const char* e = getenv("PRAG_SCAN_WG_TUNE")) ix->wg_tune_mode = atoi(e);
    if (const char* e = getenv("PRAG_ADAPTIVE")) ix->adaptive = atoi(e) != 0;
    if (const char* e = getenv("PRAG_SCAN8_QUAD_ROWS")) ix->scan8_quad_rows = atoll(e);
    if (const char* e = getenv("PRAG_MM8")) ix->mm8_mode = atoi(e) != 0;
    if (const char* e = getenv("PRAG_MM8_MIN_ROWS")) ix->mm8_min_rows = atoll(e);
    if (const char* e = getenv("PRAG_PREPASS")) ix->prepass_mode = atoi(e);
    if (const char* e = getenv("PRAG_SHADOW")) ix->shadow_mode = atoi(e);
    {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ix->cert_words), 2 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(ix->cert_words, 0, 2 * sizeof(uint32_t));
        if (e != hipSuccess) {
            set_error("prag_index_create: %s", hipGetErrorString(e));
            delete ix;
            return PRAG_EHIP;
        }
    }
    if (capacity_rows > 0) {
        int rc = ensure_capacity(ix, capacity_rows);
        if (rc != PRAG_OK) {
            (void)hipFree(ix->cert_words);
            delete ix;
            return rc;
        }
    }
    *out = ix;
    return PRAG_OK;
}

static int launch_add(prag_index* ix, const float* src_dev, bool synth, uint32_t seed, int64_t synth_row0,
                      int64_t n, hipStream_t st) {
    const int normalise = ix->metric == PRAG_METRIC_COS;
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
#define PRAG_ADD(F32_, SY_)                                                                          \
    hipLaunchKernelGGL((add_rows_kernel<F32_, SY_>), grid, block, 0, st, src_dev, seed, synth_row0, n, \
                       ix->d, normalise, ix->rows, ix->xnorm, ix->ntotal)
    if (ix->store == PRAG_F32) {
        if (synth) PRAG_ADD(true, true); else PRAG_ADD(true, false);
    } else {
        if (synth) PRAG_ADD(false, true); else PRAG_ADD(false, false);
    }
#undef PRAG_ADD
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

extern "C" int prag_index_add(prag_index_t* ix, const float* x, int64_t n, int src_is_device, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    PRAG_REQUIRE(n >= 0, PRAG_EINVAL, "n=%lld", (long long)n);
    if (n == 0) return PRAG_OK;  // index.add of an empty batch is a no-op
    PRAG_REQUIRE(x != nullptr, PRAG_EINVAL, "prag_index_add: x is NULL");
    PRAG_REQUIRE(ix->ntotal + n <= 0x7fffffffLL - 64, PRAG_EUNSUPPORTED,
                 "more than 2^31 rows per shard: shard the corpus across devices");
    int rc = ensure_capacity(ix, ix->ntotal + n);
    if (rc != PRAG_OK) return rc;
    if (src_is_device) {
        // on the caller's stream: x may still be in flight there (an encoder's output)
        rc = launch_add(ix, x, false, 0, 0, n, st);
        if (rc != PRAG_OK) return rc;
        ix->ntotal += n;
        rc = shadow_after_add(ix, st);
        PRAG_HIP(hipStreamSynchronize(st));
        return rc;
    }
    // host rows: stream through a bounded device staging buffer
    const int64_t chunk = std::min<int64_t>(n, std::max<int64_t>(1, (64ll << 20) / ((int64_t)ix->d * 4)));
    float* stage = nullptr;
    PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&stage), (size_t)chunk * ix->d * sizeof(float)));
    for (int64_t o = 0; o < n; o += chunk) {
        const int64_t m = std::min(chunk, n - o);
        hipError_t e = hipMemcpyAsync(stage, x + o * ix->d, (size_t)m * ix->d * sizeof(float), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            rc = launch_add(ix, stage, false, 0, 0, m, st);
            if (rc == PRAG_OK) e = hipStreamSynchronize(st);
        }
        if (e != hipSuccess || rc != PRAG_OK) {
            (void)hipFree(stage);
            if (e != hipSuccess) set_error("prag_index_add: %s", hipGetErrorString(e));
            return e != hipSuccess ? PRAG_EHIP : rc;
        }
        ix->ntotal += m;
    }
    (void)hipFree(stage);
    rc = shadow_after_add(ix, st);
    PRAG_HIP(hipStreamSynchronize(st));
    return rc;
}

extern "C" int prag_index_add_synthetic(prag_index_t* ix, uint32_t seed, int64_t row0, int64_t n) {
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    PRAG_REQUIRE(n >= 0 && row0 >= 0, PRAG_EINVAL, "row0=%lld n=%lld", (long long)row0, (long long)n);
    if (n == 0) return PRAG_OK;
    PRAG_REQUIRE(ix->ntotal + n <= 0x7fffffffLL - 64, PRAG_EUNSUPPORTED, "more than 2^31 rows per shard");
    int rc = ensure_capacity(ix, ix->ntotal + n);
    if (rc != PRAG_OK) return rc;
    rc = launch_add(ix, nullptr, true, seed, row0, n, nullptr);
    if (rc != PRAG_OK) return rc;
    ix->ntotal += n;
    rc = shadow_after_add(ix, nullptr);
    PRAG_HIP(hipDeviceSynchronize());
    return rc;
}

extern "C" int64_t prag_index_ntotal(const prag_index_t* ix) { return ix ? ix->ntotal : -1; }
extern "C" int prag_index_d(const prag_index_t* ix) { return ix ? ix->d : -1; }

template <int QT, int KC, bool F32, bool HP = false>
static int launch_scan(const ScanArgs& a, int grid, hipStream_t st, EventRing& prof) {
    const int lds_loop = (HP ? 2 : 1) * QT * a.qstride + 8 * 4096 + QT * 12;   // queries + stages + thresholds + best keys
    const int lds_merge = 16 * 32 * KC * 8;                     // in-workgroup list merge
    const int lds = lds_loop > lds_merge ? lds_loop : lds_merge;
    auto kern = scan_topk_kernel<QT, KC, F32, HP>;
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

template <int QT, bool F32, bool HP = false>
static int dispatch_scan_kc(int kc, const ScanArgs& a, int grid, hipStream_t st, EventRing& prof) {
    switch (kc) {
        case 8: return launch_scan<QT, 8, F32, HP>(a, grid, st, prof);
        case 16: return launch_scan<QT, 16, F32, HP>(a, grid, st, prof);
        case 32:
            if constexpr (QT == 32) return launch_scan<QT, 32, F32, HP>(a, grid, st, prof);
            break;   // 64-query tiles never carry 32-deep lists (index_search_impl: wide_ok)
    }
    set_error("internal: KC=%d", kc);
    return PRAG_EUNSUPPORTED;
}

template <int QT, int KC>
static int launch_flagged(const ScanArgs& a, int grid, const uint32_t* q_flag, int n_groups, int64_t part_stride,
                          hipStream_t st) {
    const int lds_loop = QT * a.qstride + 8 * 4096 + QT * 12;
    const int lds_merge = 16 * 32 * KC * 8;
    const int lds = lds_loop > lds_merge ? lds_loop : lds_merge;
    auto kern = scan_topk_flagged_kernel<QT, KC>;
    static LdsOptIn lds_opt_in;
    {
        const int rc_ = lds_opt_in.ensure(reinterpret_cast<const void*>(kern), 160 * 1024 - 64);
        if (rc_ != PRAG_OK) return rc_;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a, q_flag, n_groups, part_stride);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

template <int QT>
static int dispatch_flagged(int kc, const ScanArgs& a, int grid, const uint32_t* q_flag, int n_groups,
                            int64_t part_stride, hipStream_t st) {
    switch (kc) {
        case 8: return launch_flagged<QT, 8>(a, grid, q_flag, n_groups, part_stride, st);
        case 16: return launch_flagged<QT, 16>(a, grid, q_flag, n_groups, part_stride, st);
        case 32:
            if constexpr (QT == 32) return launch_flagged<QT, 32>(a, grid, q_flag, n_groups, part_stride, st);
            break;
    }
    set_error("internal: KC=%d", kc);
    return PRAG_EUNSUPPORTED;
}

template <int NKS, int KC, bool L2>
static int launch_qs(const ScanArgs& a, int grid, hipStream_t st, EventRing& prof) {
    // 4-stage ring + 2 KiB of side data, reused by the final list merge
    const int lds = 4 * 128 * 128 + 2048 > 128 * 4 * KC * 8 ? 4 * 128 * 128 + 2048 : 128 * 4 * KC * 8;
    auto kern = scan_qs_kernel<NKS, KC, L2>;
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

// query-stationary kernel: fp16 rows, d in {256,512,768,1024}, lists up to 16 deep
static int dispatch_qs(int d, int kc, const ScanArgs& a, int grid, hipStream_t st, EventRing& prof) {
#define PRAG_QS(D_)                                                          \
    if (d == D_) {                                                           \
        if (a.use_norm) {                                                    \
            if (kc == 8) return launch_qs<D_ / 32, 8, true>(a, grid, st, prof);  \
            return launch_qs<D_ / 32, 16, true>(a, grid, st, prof);          \
        }                                                                    \
        if (kc == 8) return launch_qs<D_ / 32, 8, false>(a, grid, st, prof); \
        return launch_qs<D_ / 32, 16, false>(a, grid, st, prof);             \
    }
    PRAG_QS(256) PRAG_QS(512) PRAG_QS(768)
#undef PRAG_QS
    if (d == 1024 && kc == 8) return a.use_norm ? launch_qs<32, 8, true>(a, grid, st, prof) : launch_qs<32, 8, false>(a, grid, st, prof);
    set_error("internal: query-stationary scan does not cover d=%d", d);
    return PRAG_EUNSUPPORTED;
}

static int launch_merge(int kc, const float* pk, const int* pi, int n_lists, int QT, int nq, int* cand,
                        uint32_t* tau_out, hipStream_t st, const uint32_t* q_flag = nullptr,
                        int64_t part_stride = 0, int any_idx = 0, Gate gate = Gate{}) {
    switch (kc) {
        case 8: hipLaunchKernelGGL(merge_lists_kernel<8>, dim3(nq), dim3(256), 0, st, pk, pi, n_lists, QT, cand, tau_out, q_flag, part_stride, any_idx, gate); break;
        case 16: hipLaunchKernelGGL(merge_lists_kernel<16>, dim3(nq), dim3(256), 0, st, pk, pi, n_lists, QT, cand, tau_out, q_flag, part_stride, any_idx, gate); break;
        case 32: hipLaunchKernelGGL(merge_lists_kernel<32>, dim3(nq), dim3(256), 0, st, pk, pi, n_lists, QT, cand, tau_out, q_flag, part_stride, any_idx, gate); break;
        default: set_error("internal: KC=%d", kc); return PRAG_EUNSUPPORTED;
    }
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

template <int KC>
static void launch_merge_rerank_kc(bool f32, const float* pk, const int* pi, int n_lists, int QT, int nq, int q0,
                                   const void* rows, int d, int metric_l2, const float* q32, int k, int64_t id_offset,
                                   float* D, int64_t* I, const CertArgs& cert, hipStream_t st) {
    if (f32)
        hipLaunchKernelGGL((merge_rerank_kernel<KC, true>), dim3(nq), dim3(1024), 0, st, pk, pi, n_lists, QT, q0, rows, d,
                           metric_l2, q32, k, id_offset, D, I, cert);
    else
        hipLaunchKernelGGL((merge_rerank_kernel<KC, false>), dim3(nq), dim3(1024), 0, st, pk, pi, n_lists, QT, q0, rows, d,
                           metric_l2, q32, k, id_offset, D, I, cert);
}

static int launch_merge_rerank(int kc, bool f32, const float* pk, const int* pi, int n_lists, int QT, int nq, int q0,
                               const void* rows, int d, int metric_l2, const float* q32, int k, int64_t id_offset,
                               float* D, int64_t* I, const CertArgs& cert, hipStream_t st) {
    switch (kc) {
        case 8: launch_merge_rerank_kc<8>(f32, pk, pi, n_lists, QT, nq, q0, rows, d, metric_l2, q32, k, id_offset, D, I, cert, st); break;
        case 16: launch_merge_rerank_kc<16>(f32, pk, pi, n_lists, QT, nq, q0, rows, d, metric_l2, q32, k, id_offset, D, I, cert, st); break;
        case 32: launch_merge_rerank_kc<32>(f32, pk, pi, n_lists, QT, nq, q0, rows, d, metric_l2, q32, k, id_offset, D, I, cert, st); break;
        default: set_error("internal: KC=%d", kc); return PRAG_EUNSUPPORTED;
    }
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

// candidates per query of the 8-bit tiled selection: the certificate needs the KC-th selection key to clear the
// k-th exact key by the shadow's error bound (~0.55 sigma of the score distribution on 768 Gaussian elements with
// the worst row's residual): rank ~110 at 1 M and at 21 M rows for k = 10; 256 leaves 0.2 sigma of margin
// (kMm8Kc = 256, kMm8CapWg, kMm8Chunk, kMm8Growth: with the search plan below)

// > 128 queries on fp16 rows: MFMA-tiled scan (flat_mm.hip) in chunks of kMmMaxQueries, then the
// device-flagged fallback through the per-lane-list kernel for queries whose candidate store
// overflowed.  Leaves the candidate ids in ix->cand.
static int search_tiled(prag_index* ix, int B, int Bpad, int kc, int qstride, int n_tiles, int cu_budget,
                        int chunk, int cap_wg, hipStream_t st, bool i8) {
    const int metric_l2 = ix->metric == PRAG_METRIC_L2;
    const _Float16* rows16 = reinterpret_cast<const _Float16*>(ix->rows);
    if (ix->store == PRAG_F32 && !i8) {
        // candidate selection runs on an fp16 copy of the rows (what the list kernels do on the fly)
        if (ix->rows16_cap < ix->cap) {
            ix->rows16_cap = 0; ix->rows16_n = -1;
            const int rc_ws = ws_regrow({{vpp(&ix->rows16), (size_t)ix->cap * ix->d * sizeof(_Float16)}});
            if (rc_ws != PRAG_OK) return rc_ws;
            ix->rows16_cap = ix->cap;
        }
        if (ix->rows16_n != ix->ntotal) {
            hipLaunchKernelGGL(rows_to_f16_kernel, dim3(4096), dim3(256), 0, st, reinterpret_cast<const float*>(ix->rows),
                               ix->ntotal * (int64_t)ix->d / 4, ix->rows16);
            PRAG_LAUNCH_CHECK();
            ix->rows16_n = ix->ntotal;
        }
        rows16 = ix->rows16;
    }
    MmSearch m;
    m.rows = rows16;
    m.xnorm = ix->xnorm;
    m.N = ix->ntotal;
    m.d = ix->d;
    m.alpha = metric_l2 ? -2.0f : -1.0f;
    m.use_norm = metric_l2;
    m.kc = kc;
    m.ckey = ix->mm_ckey;
    m.cidx = ix->mm_cidx;
    m.cap_q = kMmCapQ;
    m.wcnt = ix->mm_wcnt;
    m.wkey = ix->mm_wkey;
    m.widx = ix->mm_widx;
    m.cap_wg = cap_wg;
    m.wg_slots = ix->n_cu;
    m.max_wg = cu_budget;
    m.gate = ix->gate;
    m.shape16 = ix->mm_shape16 != 0;
    m.growth = mm_segment_growth(Bpad, chunk, cap_wg, cu_budget, kc, i8);     // (one place: the plan prices this schedule)
    int rc = PRAG_OK;
    for (int c0 = 0; c0 < B; c0 += chunk) {  // chunks of <= 4096 queries (LDS counters)
        m.B = std::min(B - c0, chunk);
        m.Bpad = std::min(Bpad - c0, chunk);
        m.q16 = ix->q16 + (size_t)c0 * ix->d;
        if (i8) {
            m.i8 = 1;
            m.rows8 = ix->rows8;
            m.sscale = ix->sscale;
            m.q8 = ix->sh_q8 + (size_t)c0 * ix->d;     // first int8 term of the queries
            m.kq = ix->mm_kq + c0;
        }
        m.tau = ix->g_tau + c0;
        m.cand = ix->cand + (size_t)c0 * kc;
        m.cnt = ix->mm_cnt + c0;
        m.ovf = ix->mm_ovf + c0;
        m.ovf_any = ix->mm_ovf + Bpad;
        static EventRing no_prof_mm;     // a gated second-tier search is not part of the profiled launches
        rc = mm_run(m, st, ix->gate.word ? no_prof_mm : ix->prof);
        if (rc != PRAG_OK) return rc;
    }
    // Queries whose candidate buffer overflowed (flag set on the device): their groups go
    // through the per-lane-list kernel again; with no flag set this is two empty launches.
    // deep lists have no list-kernel fallback: a query whose candidate store overflowed keeps its flag
    // in mm_ovf, the rerank puts it on the certificate's flag list and the exact scan recomputes it
    if (kc > 32) return PRAG_OK;
    if (ix->mm_mode == 2) return PRAG_OK;  // PRAG_SCAN_MM=2 (tests of the tests): overflow goes unrepaired
    const bool fb64 = 64 * qstride + 8 * 4096 + 64 * 12 + 64 <= 160 * 1024 - 64 && kc < 32;
    const int fq = fb64 ? 64 : 32;
    const int fb_grid = std::max(1, std::min(cu_budget, (n_tiles + 7) / 8));
    const int n_groups = Bpad / fq;
    const int64_t part_stride = (int64_t)fb_grid * fq * kc;
    const size_t fb_need = (size_t)n_groups * part_stride;
    if (fb_need > ix->part_cap) {
        ix->part_cap = 0;
        const int rc_ws = ws_regrow({{vpp(&ix->part_key), fb_need * sizeof(float)}, {vpp(&ix->part_idx), fb_need * sizeof(int)}});
        if (rc_ws != PRAG_OK) return rc_ws;
        ix->part_cap = fb_need;
    }
    ScanArgs a;
    a.rows = rows16;
    a.xnorm = ix->xnorm;
    a.q16 = ix->q16;
    a.q16lo = ix->q16lo;
    a.N = ix->ntotal;
    a.d = ix->d;
    a.qstride = qstride;
    a.n_tiles = n_tiles;
    a.alpha = m.alpha;
    a.use_norm = metric_l2;
    a.out_key = ix->part_key;
    a.out_idx = ix->part_idx;
    a.g_tau = ix->g_tau;
    a.g_slot = nullptr;
    a.gate = ix->gate;
    rc = fb64 ? dispatch_flagged<64>(kc, a, fb_grid, ix->mm_ovf, n_groups, part_stride, st)
              : dispatch_flagged<32>(kc, a, fb_grid, ix->mm_ovf, n_groups, part_stride, st);
    if (rc != PRAG_OK) return rc;
    rc = launch_merge(kc, ix->part_key, ix->part_idx, fb_grid, fq, B, ix->cand, ix->g_tau, st, ix->mm_ovf,
                      part_stride, Bpad, ix->gate);
    if (rc != PRAG_OK) return rc;
    return PRAG_OK;
}

static void consume_tier_stats(prag_index* ix, bool wait);


extern "C" int prag_index_search(prag_index_t* ix, const float* q, int B, int k, int64_t id_offset, float* D,
                                 int64_t* I, int io_is_device, void* stream) {
    return index_search_impl(ix, q, B, k, id_offset, D, I, io_is_device, stream, 0);
}

extern "C" int prag_index_search_tagged(prag_index_t* ix, const float* q, int B, int k, int64_t id_offset, float* D,
                                        int64_t* I, int io_is_device, void* stream) {
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    PRAG_REQUIRE(id_offset >= 0 && id_offset + ix->ntotal < (1ll << kTagShift), PRAG_EUNSUPPORTED,
                 "tagged ids hold %d-bit global row ids", kTagShift);
    return index_search_impl(ix, q, B, k, id_offset, D, I, io_is_device, stream, 1);
}

// Allocate every workspace a search of this shape needs NOW (the index grows its workspaces lazily, by shape class, with
// hipFree / hipMalloc - i.e. a device synchronisation - inside the first search that needs more): throw-away searches
// of B pseudo-random queries on `stream`, waited for.  Afterwards a search of up to B queries and this k allocates
// nothing, never waits for the device on the device-io path, and can be captured into a graph from its first call.
// Round 5: the dummy queries are a fixed non-degenerate pattern - all-zero queries tie every row, no
// certificate can pass, and every query went through the exact float64 scan of the whole shard (or failed the whole
// int8 tier and fed the auto-off heuristic a synthetic whole-batch repeat).  The tier statistics are put back as they
// were, and a shape that can take the int8 tiles is searched under BOTH settings of the auto-off switch, so that the
// plan it flips to later finds its workspaces too.
extern "C" int prag_index_reserve(prag_index_t* ix, int B, int k, void* stream) {
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    PRAG_REQUIRE(B >= 1 && k >= 1, PRAG_EINVAL, "prag_index_reserve: B=%d k=%d", B, k);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* q = nullptr;
    float* D = nullptr;
    int64_t* I = nullptr;
    const size_t nq = (size_t)B * ix->d;
    std::vector<float> hq(nq);
    uint32_t lcg = 0x9E3779B9u;
    for (size_t i = 0; i < nq; ++i) {          // uniform in [-1, 1): distinct directions, no ties
        lcg = lcg * 1664525u + 1013904223u;
        hq[i] = (float)(int32_t)lcg * (1.0f / 2147483648.0f);
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&q), nq * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&D), (size_t)B * k * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&I), (size_t)B * k * sizeof(int64_t));
    int rc = PRAG_OK;
    // the statistics of the int8-tile heuristic are the caller's searches' business, not a sizing run's
    consume_tier_stats(ix, true);
    const int sv_streak = ix->mm8_whole_batch_streak, sv_count = ix->mm8_off_count, sv_period = ix->mm8_off_period,
              sv_failed = ix->mm8_last_failed;
    const bool sv_off = ix->mm8_auto_off;
    const std::string sv_plan = ix->last_plan;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("prag_index_reserve: %s", hipGetErrorString(e));
        rc = e == hipErrorOutOfMemory ? PRAG_ENOMEM : PRAG_EHIP;
    } else {
        e = hipMemcpyAsync(q, hq.data(), nq * sizeof(float), hipMemcpyHostToDevice, st);
        const int sv_retry = ix->retry_mode;
        const bool sv_armed = ix->retry_armed;
        if (ix->retry_mode != 0) ix->retry_mode = 1;       // a 33-128-query shape also sizes its retry tier (retry_tier)
        for (int pass = 0; pass < 2 && e == hipSuccess && rc == PRAG_OK; ++pass) {
            ix->mm8_auto_off = pass == 0 ? false : true;
            ix->mm8_off_count = 0;             // (no probe flips the switch back inside the sizing search)
            rc = index_search_impl(ix, q, B, k, 0, D, I, 1, stream, 0);
            const hipError_t e2 = hipStreamSynchronize(st);
            if (e2 != hipSuccess) e = e2;
            ix->tier_pending = false;
            ix->r2_pending = false;
            if (ix->last_plan.find("int8_tiles=1") == std::string::npos && pass == 0) break;   // not an int8-tile shape
        }
        ix->retry_mode = sv_retry;
        ix->retry_armed = sv_armed;
        if (rc == PRAG_OK && e != hipSuccess) {
            set_error("prag_index_reserve: %s", hipGetErrorString(e));
            rc = PRAG_EHIP;
        }
    }
    ix->mm8_whole_batch_streak = sv_streak;
    ix->mm8_off_count = sv_count;
    ix->mm8_off_period = sv_period;
    ix->mm8_auto_off = sv_off;
    ix->mm8_last_failed = sv_failed;
    ix->last_plan = sv_plan;
    ix->tier_pending = false;
    for (void* p : {(void*)q, (void*)D, (void*)I})
        if (p) (void)hipFree(p);
    return rc;
}

// Second tier of the 8-bit tiled selection.  The queries that failed its certificate are on the device (flag count
// in *flag_word, batch rows in ix->flag_list) and so is the decision what to do about them: BOTH continuations are
// enqueued, each behind a Gate on the failed count, and the one that does not apply returns at the top of every kernel
// (~2 us per launch; a > 128-query search is >= 0.1 ms of GPU time):
//   1 .. hi_sub failed  (hi_sub = min(kMm8SubsetMax, B / 4): the odd query sitting in a cluster of look-alikes) - they
//                       are gathered into a compact batch of kMm8SubsetMax queries (padded with duplicates), searched by
//                       the <= 64-query kernels - the two-level search over the shadow, with its own certificate and
//                       exact fallback - and scattered back over their rows of the result;
//   more than that      the WHOLE batch is repeated on the fp16 tiles.
// Round 3 read the count back (hipMemcpyAsync + hipStreamSynchronize) and branched on the host: the one place a
// device-io search waited for its stream - not capturable into a graph, a stall for a pipelined caller, and in the
// lockstep multi-rank path every rank took its tier decision alone.  Now the count only travels to the host
// asynchronously, for prag_index_last_tiled8 and the auto-off heuristic (consume_tier_stats).
constexpr int kMm8SubsetMax = 64;
static void consume_tier_stats(prag_index* ix, bool wait) {
    if (!ix->tier_pending) return;
    if (wait) {
        if (hipEventSynchronize(ix->tier_event) != hipSuccess) return;
    } else if (hipEventQuery(ix->tier_event) != hipSuccess) {
        (void)hipGetLastError();          // hipErrorNotReady: look again at the next search
        return;
    }
    ix->tier_pending = false;
    const int n = (int)*ix->tier_word_host;
    ix->mm8_last_failed = n;
    if (n > ix->tier_hi_sub) {            // the whole batch went through both tiers
        if (++ix->mm8_whole_batch_streak >= 2) {
            ix->mm8_auto_off = true;
            ix->mm8_off_count = 0;
        }
    } else {
        ix->mm8_whole_batch_streak = 0;
        if (n == 0) ix->mm8_off_period = 64;
    }
}

struct GateScope {          // ix->gate is reset on every path out of the second tier
    prag_index* ix;
    ~GateScope() { ix->gate = Gate{}; }
};

static int mm8_second_tier(prag_index* ix, const float* q_dev, int B, int k, int64_t id_offset, float* D_dev, int64_t* I_dev,
                           void* stream, int tag_ids, const uint32_t* flag_word) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (ix->t2_cap < kMm8SubsetMax || ix->t2_k < k || !ix->t2_word) {
        const int nk = std::max(k, ix->t2_k);
        ix->t2_cap = 0; ix->t2_k = 0;
        const int rc_ws = ws_regrow({{vpp(&ix->t2_list), (size_t)kMm8SubsetMax * sizeof(int)},
                                     {vpp(&ix->t2_q), (size_t)kMm8SubsetMax * ix->d * sizeof(float)},
                                     {vpp(&ix->t2_D), (size_t)kMm8SubsetMax * nk * sizeof(float)},
                                     {vpp(&ix->t2_I), (size_t)kMm8SubsetMax * nk * sizeof(int64_t)},
                                     {vpp(&ix->t2_word), 4 * sizeof(uint32_t)}});
        if (rc_ws != PRAG_OK) return rc_ws;
        ix->t2_cap = kMm8SubsetMax;
        ix->t2_k = nk;
    }
    if (!ix->tier_word_host) {
        PRAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&ix->tier_word_host), 4 * sizeof(uint32_t)));
        PRAG_HIP(hipEventCreateWithFlags(&ix->tier_event, hipEventDisableTiming));
    }
    hipLaunchKernelGGL(mm8_tier_plan_kernel, dim3(1), dim3(64), 0, st, flag_word, ix->flag_list, ix->t2_word, ix->t2_list,
                       kMm8SubsetMax);
    PRAG_LAUNCH_CHECK();
    const uint32_t hi_sub = (uint32_t)std::min(kMm8SubsetMax, B / 4);
    {   // statistics, never waited for by a search; a capturing stream records nothing (the count stays unknown)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        ix->tier_pending = false;
        if (!capturing) {
            PRAG_HIP(hipMemcpyAsync(ix->tier_word_host, ix->t2_word, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PRAG_HIP(hipEventRecord(ix->tier_event, st));
            ix->tier_pending = true;
            ix->tier_hi_sub = (int)hi_sub;
        } else {
            ix->mm8_last_failed = -3;
        }
    }
    GateScope scope{ix};
    // ---- a few failed queries: compact batch through the <= 64-query kernels -----------------------------------------
    ix->gate = Gate{ix->t2_word, 1u, hi_sub};
    hipLaunchKernelGGL(gather_queries_kernel, dim3(kMm8SubsetMax), dim3(256), 0, st, q_dev, ix->t2_list, kMm8SubsetMax, ix->d,
                       ix->t2_q, ix->gate);
    PRAG_LAUNCH_CHECK();
    int rc = index_search_impl(ix, ix->t2_q, kMm8SubsetMax, k, id_offset, ix->t2_D, ix->t2_I, 1, stream, tag_ids, false);
    if (rc != PRAG_OK) return rc;
    hipLaunchKernelGGL(scatter_results_kernel, dim3(kMm8SubsetMax), dim3(64), 0, st, ix->t2_D, ix->t2_I, ix->t2_list, ix->t2_word,
                       k, D_dev, I_dev, ix->gate);
    PRAG_LAUNCH_CHECK();
    // ---- more than that: the whole batch on the fp16 tiles ------------------------------------------------------------
    ix->gate = Gate{ix->t2_word, hi_sub + 1u, 0xFFFFFFFFu};
    rc = index_search_impl(ix, q_dev, B, k, id_offset, D_dev, I_dev, 1, stream, tag_ids, false);
    return rc;
}

// The flag count of the previous <= 128-query search, if it has arrived: something flagged arms the retry tier, 64 armed
// searches in a row without a flag disarm it (an armed search with nothing flagged pays ~7 early-exit launches per 32
// queries, ~40 us at 64 queries: not something the common case should carry).
static void consume_retry_stats(prag_index* ix, bool wait) {
    if (!ix->r2_pending) return;
    if (wait) {
        if (hipEventSynchronize(ix->r2_event) != hipSuccess) return;
    } else if (hipEventQuery(ix->r2_event) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    ix->r2_pending = false;
    if (ix->r2_word_host[0] > 0u) {
        ix->retry_armed = true;
        ix->retry_clean = 0;
    } else if (ix->retry_armed && ++ix->retry_clean >= 64) {
        ix->retry_armed = false;
        ix->exact_group_hint = false;
    }
    if (ix->r2_has_unfinished) {            // [3] = queries the bound kernel left to the gather in that search
        ix->r2_has_unfinished = false;
        if (ix->r2_word_host[3] > 0u) {
            ix->gather_armed = true;
            ix->gather_clean = -48;             // (armed for 64 searches)
        } else if (ix->gather_armed && ++ix->gather_clean >= 16) {
            ix->gather_armed = false;
        }
    }
    if (ix->r2_word_host[2] != 0u) {        // the tier ran in that search: [1] = what its inner searches still flagged
        ix->exact_group_hint = ix->r2_word_host[1] >= 4u;
        ix->r2_word_host[2] = 0u;
    }
}

static int retry_tier(prag_index* ix, const float* q_dev, int B, int k, int64_t id_offset, float* D_dev, int64_t* I_dev,
                      void* stream, int tag_ids, uint32_t* flag_word) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    constexpr int kPart = 32, kMaxB = 128;
    if (!ix->r2_word || ix->r2_k < k) {
        const int nk = std::max(k, ix->r2_k);
        ix->r2_k = 0;
        const int rc_ws = ws_regrow({{vpp(&ix->r2_list), (size_t)kMaxB * sizeof(int)},
                                     {vpp(&ix->r2_q), (size_t)kPart * ix->d * sizeof(float)},
                                     {vpp(&ix->r2_D), (size_t)kPart * nk * sizeof(float)},
                                     {vpp(&ix->r2_I), (size_t)kPart * nk * sizeof(int64_t)},
                                     {vpp(&ix->r2_word), 4 * sizeof(uint32_t)}});
        if (rc_ws != PRAG_OK) return rc_ws;
        ix->r2_k = nk;
    }
    hipLaunchKernelGGL(retry_plan_kernel, dim3(1), dim3(128), 0, st, flag_word, ix->flag_list, ix->r2_word, ix->r2_list, kMaxB);
    PRAG_LAUNCH_CHECK();
    GateScope scope{ix};
    const int n_parts = (std::min(B, kMaxB) + kPart - 1) / kPart;
    for (int p = 0; p < n_parts; ++p) {
        ix->gate = Gate{ix->r2_word, (uint32_t)(kPart * p) + 1u, 0xFFFFFFFFu};
        hipLaunchKernelGGL(gather_queries_kernel, dim3(kPart), dim3(256), 0, st, q_dev, ix->r2_list + kPart * p, kPart, ix->d,
                           ix->r2_q, ix->gate);
        PRAG_LAUNCH_CHECK();
        // (32 queries: two-term tiles over the shadow / high-precision list scan over the rows, its own certificate and -
        //  for what even that cannot clear - its own exact scan; the inner search resets and refills the flag list)
        const int rc = index_search_impl(ix, ix->r2_q, kPart, k, id_offset, ix->r2_D, ix->r2_I, 1, stream, tag_ids, false);
        if (rc != PRAG_OK) return rc;
        hipLaunchKernelGGL(retry_scatter_kernel, dim3(kPart), dim3(64), 0, st, ix->r2_D, ix->r2_I, ix->r2_list, ix->r2_word,
                           ix->cert_words /* the inner (device-io) search's flag count */, kPart * p, k, D_dev, I_dev, ix->gate);
        PRAG_LAUNCH_CHECK();
    }
    ix->gate = Gate{};
    hipLaunchKernelGGL(retry_finish_kernel, dim3(1), dim3(1), 0, st, ix->r2_word, flag_word);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// One search = a plan (plan_search) and the stages below, each a function of its own (round 5; rounds 1-4 grew one
// 600-line function).  SearchRun is what the stages share.  Every workspace a search needs is sized BEFORE its first
// launch, in two places: search_workspaces (query block, lists, candidates, tiled-scan stores, exact-scan lists) and
// search_shadow_args (the two-level search's query terms, regions and slice lists); the tiers that run a search of
// their own (mm8_second_tier, retry_tier) size theirs where they start.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kShadowCap = 512;   // candidates per (scan workgroup, query) region of the two-level search (search_shadow_args)

struct SearchRun {
    prag_index* ix;
    SearchPlan P;
    int B, k;
    int64_t id_offset;
    int io_is_device;
    void* stream;
    int tag_ids;
    bool allow_mm8;
    hipStream_t st;
    const float* q_dev;        // device views of the caller's queries / results (host i/o: the staging blocks)
    float* D_dev;
    int64_t* I_dev;
    CertArgs cert;
    uint32_t* flag_word;       // flag count of this search (device i/o: cert_words[0]; host i/o: behind the results)
    EventRing* prof;
    ShadowPrep sprep;
    bool reranked;             // the executor wrote D / I itself (two-level search, list scan)
};
// the names the stages were written with
#define SEARCH_BASE(r)                                                                                               \
    [[maybe_unused]] prag_index* const ix = (r).ix;                                                                  \
    [[maybe_unused]] const SearchPlan& P = (r).P;                                                                    \
    [[maybe_unused]] const int B = (r).B, k = (r).k, io_is_device = (r).io_is_device, tag_ids = (r).tag_ids;         \
    [[maybe_unused]] const int64_t id_offset = (r).id_offset;                                                        \
    [[maybe_unused]] void* const stream = (r).stream;                                                                \
    [[maybe_unused]] const bool allow_mm8 = (r).allow_mm8;                                                           \
    [[maybe_unused]] hipStream_t const st = (r).st;                                                                  \
    [[maybe_unused]] const float*& q_dev = (r).q_dev;                                                                \
    [[maybe_unused]] float*& D_dev = (r).D_dev;                                                                      \
    [[maybe_unused]] int64_t*& I_dev = (r).I_dev;
#define SEARCH_PLAN(r)                                                                                               \
    [[maybe_unused]] const int kc = P.kc, qstride = P.qstride, QT = P.QT, Bpad = P.Bpad, n_tiles = P.n_tiles,        \
                               cu_budget = P.cu_budget, grid = P.grid, n_lists = P.grid, mm_chunk = P.mm_chunk,      \
                               mm_cap_wg = P.mm_cap_wg, ex_grid = P.ex_grid, ex_fcap = P.ex_fcap;                    \
    [[maybe_unused]] const bool exact_only = P.exact_only, use_mm8 = P.use_mm8, use_mm = P.use_mm, use_qs = P.use_qs, \
                                use_hp = P.use_hp, certify = P.certify, use_shadow = P.use_shadow;                   \
    [[maybe_unused]] const size_t part_need = P.part_need, cand_need = P.cand_need;                                  \
    [[maybe_unused]] const int metric_l2 = (r).ix->metric == PRAG_METRIC_L2;
#define SEARCH_CERT(r)                                                                                               \
    [[maybe_unused]] CertArgs& cert = (r).cert;                                                                      \
    [[maybe_unused]] uint32_t* const flag_word = (r).flag_word;                                                      \
    [[maybe_unused]] EventRing& prof = *(r).prof;                                                                    \
    [[maybe_unused]] ShadowPrep& sprep = (r).sprep;                                                                  \
    [[maybe_unused]] bool& reranked = (r).reranked;

// host i/o: queries into the pinned / device staging block, results come back through it (search_finish)
static int search_stage_io(SearchRun& r, const float* q, float* D, int64_t* I) {
    SEARCH_BASE(r)
    // ---- host i/o staging -----------------------------------------------------
    q_dev = q;
    D_dev = D;
    I_dev = I;
    if (!io_is_device) {
        if (B > ix->io_B || k > ix->io_k) {
            if (ix->io_q) (void)hipFree(ix->io_q);
            if (ix->io_res) (void)hipFree(ix->io_res);
            if (ix->io_q_host) (void)hipHostFree(ix->io_q_host);
            if (ix->io_res_host) (void)hipHostFree(ix->io_res_host);
            ix->io_q = nullptr; ix->io_res = nullptr; ix->io_q_host = nullptr; ix->io_res_host = nullptr;
            const int nb = std::max(B, ix->io_B), nk = std::max(k, ix->io_k);
            ix->io_B = ix->io_k = 0;
            const size_t res_bytes = (size_t)nb * nk * 12 + 8;
            PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&ix->io_q), (size_t)nb * ix->d * sizeof(float)));
            PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&ix->io_res), res_bytes));
            PRAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&ix->io_q_host), (size_t)nb * ix->d * sizeof(float)));
            PRAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&ix->io_res_host), res_bytes));
            ix->io_B = nb;
            ix->io_k = nk;
        }
        memcpy(ix->io_q_host, q, (size_t)B * ix->d * sizeof(float));
        PRAG_HIP(hipMemcpyAsync(ix->io_q, ix->io_q_host, (size_t)B * ix->d * sizeof(float), hipMemcpyHostToDevice, st));
        q_dev = ix->io_q;
        I_dev = reinterpret_cast<int64_t*>(ix->io_res);
        D_dev = reinterpret_cast<float*>(ix->io_res + (size_t)B * k * 8);
    }

    return PRAG_OK;
}

// every per-search workspace of the plan's kernel family (and of the exact scan behind it), before the first launch
static int search_workspaces(SearchRun& r) {
    SEARCH_BASE(r)
    const int Bpad = P.Bpad;
    const bool use_mm = P.use_mm;
    if (Bpad > ix->q_cap) {
        ix->q_cap = 0;
        const int rc_ws = ws_regrow({{vpp(&ix->qinfo), (size_t)Bpad * 4 * sizeof(float)},
                                     {vpp(&ix->qn2), (size_t)Bpad * sizeof(double)},
                                     {vpp(&ix->flag_list), (size_t)Bpad * sizeof(int)},
                                     {vpp(&ix->q32), (size_t)Bpad * ix->d * sizeof(float)},
                                     {vpp(&ix->q16), (size_t)Bpad * ix->d * sizeof(_Float16)},
                                     {vpp(&ix->q16lo), (size_t)Bpad * ix->d * sizeof(_Float16)},
                                     {vpp(&ix->g_tau), (size_t)Bpad * sizeof(uint32_t)},
                                     {vpp(&ix->g_slot), (size_t)Bpad * kSlotWords * sizeof(uint32_t)}});
        if (rc_ws != PRAG_OK) return rc_ws;
        ix->q_cap = Bpad;
    }
    const size_t part_need = P.part_need;      // one merged list per workgroup and query
    if (part_need > ix->part_cap) {
        ix->part_cap = 0;
        const int rc_ws = ws_regrow({{vpp(&ix->part_key), part_need * sizeof(float)}, {vpp(&ix->part_idx), part_need * sizeof(int)}});
        if (rc_ws != PRAG_OK) return rc_ws;
        ix->part_cap = part_need;
    }
    const size_t cand_need = P.cand_need;
    if (cand_need > ix->cand_cap) {
        ix->cand_cap = 0;
        const int rc_ws = ws_regrow({{vpp(&ix->cand), cand_need * sizeof(int)}});
        if (rc_ws != PRAG_OK) return rc_ws;
        ix->cand_cap = cand_need;
    }

    // queries per mm_run call and survivors one workgroup can hold per query and segment.  Deep lists
    // (k > 26) yield up to KC/8 survivors per 256-row tile right after the first segment: they get
    // 512 slots and 256-query chunks.
    const int mm_chunk = P.mm_chunk, mm_cap_wg = P.mm_cap_wg;
    if (use_mm) {
        if (Bpad > ix->mm_q_cap) {
            ix->mm_q_cap = 0;
            const int rc_ws = ws_regrow({{vpp(&ix->mm_cnt), (size_t)Bpad * sizeof(uint32_t)},
                                         {vpp(&ix->mm_kq), (size_t)Bpad * sizeof(float)},
                                         {vpp(&ix->mm_ovf), ((size_t)Bpad + 1) * sizeof(uint32_t)}});
            if (rc_ws != PRAG_OK) return rc_ws;
            ix->mm_q_cap = Bpad;
        }
        const size_t c_need = (size_t)std::max(mm_chunk, ix->n_cu) * std::max<size_t>(kMmCapQ, ix->n_cu);
        if ((size_t)mm_chunk * kMmCapQ > ix->mm_c_entries || (size_t)ix->n_cu * mm_chunk > ix->mm_c_entries) {
            ix->mm_c_entries = 0;
            const int rc_ws = ws_regrow({{vpp(&ix->mm_ckey), c_need * sizeof(float)}, {vpp(&ix->mm_cidx), c_need * sizeof(int)},
                                         {vpp(&ix->mm_wcnt), c_need * sizeof(uint32_t)}});
            if (rc_ws != PRAG_OK) return rc_ws;
            ix->mm_c_entries = c_need;
        }
        const size_t w_need = (size_t)ix->n_cu * mm_chunk * mm_cap_wg;
        if (w_need > ix->mm_w_entries) {
            ix->mm_w_entries = 0;
            const int rc_ws = ws_regrow({{vpp(&ix->mm_wkey), w_need * sizeof(float)}, {vpp(&ix->mm_widx), w_need * sizeof(int)}});
            if (rc_ws != PRAG_OK) return rc_ws;
            ix->mm_w_entries = w_need;
        }
    }

    // ---- exact-scan workspace (the certificate's fallback) --------------------------------------------------------
    const bool certify = P.certify;
    const int ex_grid = P.ex_grid, ex_fcap = P.ex_fcap;
    if (certify && ix->ntotal > 0) {
        const size_t need = exact_part_entries(ex_fcap, ex_grid, k);
        if (need > ix->ex_entries) {
            ix->ex_entries = 0;
            const int rc_ws = ws_regrow({{vpp(&ix->ex_key), need * sizeof(unsigned long long)}, {vpp(&ix->ex_id), need * sizeof(int)}});
            if (rc_ws != PRAG_OK) return rc_ws;
            ix->ex_entries = need;
        }
        // every workgroup's best key per flagged query (exact_mfma_kernel's chip-wide bound): all-ones between searches
        const size_t pool_need = exact_part_entries(ex_fcap, ex_grid, 1);
        if (pool_need > ix->ex_pool_entries) {
            ix->ex_pool_entries = 0;
            const int rc_ws = ws_regrow({{vpp(&ix->ex_pool), pool_need * sizeof(unsigned long long)}});
            if (rc_ws != PRAG_OK) return rc_ws;
            PRAG_HIP(hipMemsetAsync(ix->ex_pool, 0xff, pool_need * sizeof(unsigned long long), st));
            ix->ex_pool_entries = pool_need;
        }
        if (ex_fcap > ix->ex_done_cap) {
            ix->ex_done_cap = 0;
            const int rc_ws = ws_regrow({{vpp(&ix->ex_done), (size_t)ex_fcap * sizeof(uint32_t)}});
            if (rc_ws != PRAG_OK) return rc_ws;
            PRAG_HIP(hipMemsetAsync(ix->ex_done, 0, (size_t)ex_fcap * sizeof(uint32_t), st));
            ix->ex_done_cap = ex_fcap;
        }
    }
    return PRAG_OK;
}

// the certificate's error model for the scan that will run (flat_internal.h "Exactness certificate")
static void search_certificate(SearchRun& r) {
    SEARCH_BASE(r)
    SEARCH_PLAN(r)
    CertArgs cert;
    cert.qinfo = ix->qinfo;
    cert.qn2 = ix->qn2;
    cert.xn_max = ix->cert_words + 1;
    // host i/o: the flag count lives behind the results so that one transfer brings everything back
    uint32_t* flag_word = io_is_device ? ix->cert_words : reinterpret_cast<uint32_t*>(ix->io_res + (size_t)B * k * 12);
    cert.n_flag = flag_word;
    cert.flag_list = ix->flag_list;
    cert.force = nullptr;
    cert.tag_ids = tag_ids;
    cert.sq8 = nullptr;        // (set below, once the query workspace of the 8-bit selection exists)
    cert.e_max = ix->shadow_err_max;
    cert.kshift = nullptr;
    cert.gate = ix->gate;
    static EventRing no_prof_gated;   // a gated second-tier search is not part of the profiled launches
    EventRing& prof = ix->gate.word ? no_prof_gated : ix->prof;
    {
        // which operands the selection kernel rounds (flat_internal.h "Exactness certificate")
        const bool hp = !use_mm && !use_qs && use_hp;
        const bool rows_rounded = ix->store == PRAG_F32;          // fp32 rows -> fp16 terms inside the scan
        const double u16 = 1.0 / 2048.0, sub = std::sqrt((double)ix->d) * 2.9802322387695312e-08 /* 2^-25 */;
        cert.rq_sel = hp ? 2 : 1;
        if (!rows_rounded) { cert.c_row = 0.f; cert.c_abs = 0.f; }
        else if (hp) { cert.c_row = (float)(3.0 * u16 * u16); cert.c_abs = (float)(2.0 * sub); }
        else { cert.c_row = (float)(u16 * (1.0 + u16) * 1.001); cert.c_abs = (float)(sub * 1.001); }
        const int chains = hp ? (rows_rounded ? 3 : 2) : 1;
        cert.c_acc = (float)((double)ix->d * chains * 1.1920928955078125e-07 /* 2^-23 */ * 1.001);
        if (!certify) {  // timing experiments: everything certifies
            cert.c_row = cert.c_abs = cert.c_acc = -1e30f;
        }
    }
    r.cert = cert;
    r.flag_word = flag_word;
    r.prof = &prof;
}

// the two-level search's (and the int8 tiles') query-side workspace and what prep_queries_kernel writes into it
static int search_shadow_args(SearchRun& r) {
    SEARCH_BASE(r)
    SEARCH_PLAN(r)
    SEARCH_CERT(r)
    // ---- two-level search through the 8-bit shadow (HBM-bound batches on large shards) --------------
    // candidates per (scan workgroup, query) region: 64 MB at 256 workgroups x 64 queries.  128 (round 2) sent
    // every query of a corpus with contiguous clusters to the exact scan (104 ms per search against 1.1 ms for
    // the direct scan: whole tiles of a loosely bounded query's look-alikes arrive at once); 512 holds them
    // (0.77 ms) and costs nothing on the i.i.d. corpus (same box: 2.84 / 2.85 / 2.84 ms at 128 / 256 / 512)
    // (the shadow's eps constants read max ||x||^2 and its overflow path needs the exact scan: both exist
    // only with the certificate on - a diag build with PRAG_CERT=0 scans the rows directly)
    sprep = ShadowPrep{};
    if (use_shadow || use_mm8) {   // workspace of the two-level search; its query terms come out of prep_queries_kernel
        const int BpadS = use_mm8 ? Bpad : (B + 63) / 64 * 64;
        if (BpadS > ix->sh_q_cap) {
            ix->sh_q_cap = 0;
            const int rc_ws = ws_regrow({{vpp(&ix->sh_q8), (size_t)2 * BpadS * ix->d},
                                         {&ix->sh_sq, (size_t)BpadS * shadow_q_bytes()},
                                         {vpp(&ix->sh_kshift), (size_t)BpadS * sizeof(double)},
                                         {vpp(&ix->sh_unfin), 4 * sizeof(uint32_t)},
                                         {vpp(&ix->sh_slots), (size_t)BpadS * shadow_slot_words() * sizeof(uint32_t)},
                                         {vpp(&ix->sh_ovf), (size_t)2 * BpadS * sizeof(uint32_t)}});   // + arrival counters
            if (rc_ws != PRAG_OK) return rc_ws;
            ix->sh_q_cap = BpadS;
        }
        const int cand_qt = QT == 128 ? 128 : 64;     // regions of kShadowCap slots for every query of a tile
        if (use_shadow && (!ix->sh_cand || !ix->sh_ccnt || ix->sh_cand_qt < cand_qt)) {
            ix->sh_cand_qt = 0;
            const int rc_ws = ws_regrow({{vpp(&ix->sh_cand), (size_t)ix->n_cu * cand_qt * kShadowCap * 2 * sizeof(int)},
                                         {vpp(&ix->sh_ccnt), (size_t)ix->n_cu * 128 * sizeof(uint32_t)}});
            if (rc_ws != PRAG_OK) return rc_ws;
            ix->sh_cand_qt = cand_qt;
        }
        const size_t pe = (size_t)BpadS * shadow_split() * k;
        if (use_shadow && pe > ix->sh_part_entries) {
            ix->sh_part_entries = 0;
            const int rc_ws = ws_regrow({{vpp(&ix->sh_pkey), pe * sizeof(unsigned long long)}, {vpp(&ix->sh_pid), pe * sizeof(int)}});
            if (rc_ws != PRAG_OK) return rc_ws;
            ix->sh_part_entries = pe;
        }
        sprep.q8a = ix->sh_q8;
        sprep.q8b = ix->sh_q8 + (size_t)ix->sh_q_cap * ix->d;
        sprep.sq = reinterpret_cast<ShadowQ*>(ix->sh_sq);
        sprep.slots = ix->sh_slots;
        sprep.ovf = ix->sh_ovf;
        sprep.done = ix->sh_ovf + ix->sh_q_cap;
        sprep.xn_max = ix->cert_words + 1;
        sprep.alpha = metric_l2 ? -2.0f : -1.0f;
        sprep.aff = ix->sh_aff;
        sprep.yn_max = ix->sh_yn_max;
        sprep.bias_max = ix->sh_bias_max;
        sprep.centre_query = use_shadow ? 1 : 0;     // (the int8 tiles keep the query as it is: their rows carry no bias term)
        sprep.kshift = ix->sh_kshift;
        sprep.unfinished = ix->sh_unfin;
        // sample for the pre-bound: kShadowSampleSlices x kShadowSampleTiles whole tiles spread over the shard
        sprep.rows8 = ix->rows8;
        sprep.sscale = ix->sscale;
        sprep.serr = ix->serr;
        sprep.sbias = ix->sbias;
        const int64_t whole_tiles = ix->ntotal / 32;
        // (PRAG_SHADOW_SAMPLE=2: > 32 queries sample ONE tile per slice - a quarter of the sampling work for a bound
        //  at the ~1 % quantile instead of the 0.25 % one; 3: two tiles; <= 32 queries keep four)
        sprep.sample_tiles = (ix->shadow_sample_mode == 2 && B > 32) ? 1 : (ix->shadow_sample_mode == 3 && B > 32) ? 2 : kShadowSampleTiles;
        sprep.sample_stride = whole_tiles / (kShadowSampleSlices * sprep.sample_tiles);   // 0: shard too small
        // The sampled bound is used for batches of <= 32 queries (kernel trace at 2.6 M rows: the sampling waves
        // take the prep kernel from 7 to 19 us and the scan from 399 to 376 us; at 64 queries - one query term,
        // twice the error band - the candidates of the loosely bounded first tiles cost the scan and the gather
        // what the warm-up's second visits cost: prep +13 us, gather +6 us, scan +-0, and one query in a few
        // searches overflowed a region).  PRAG_SHADOW_SAMPLE=0|1 forces it off / on for A/B runs (exact either way).
        const int sample_env = ix->shadow_sample_mode;
        // ... and on shards so small that a wave owns fewer than 6 tiles: the chip-wide bound reaches a wave
        // through the slot epochs it polls after its 2nd and 3rd tile, so without a starting bound such a wave
        // filters against its own lists only and nearly every row passes (65 537 rows, 64 queries: every query
        // overflowed the gather's staging and went to the exact scan; found by tools/fuzz_shadow.py)
        const bool few_tiles = n_tiles < 48 * cu_budget;
        // Round 5 (profiles/r05b_scan8_stamps_shard.txt, r05c_shard_ab.txt): at the 8-GPU shard size a wave owns 40
        // tiles and re-visits 3 of them when the scan opens without a bound (7.8 % of the scan, 18 us); with the gate
        // of the next batch running beside the search's tail the sampling waves' 13 us are off the critical path -
        // 0.459 -> 0.420 ms per pass.  > 32 queries sample when a wave owns fewer than 128 tiles (~8 M rows).
        const bool short_scan = n_tiles < 128 * 8 * cu_budget;
        if (sample_env == 0 || (sample_env < 0 && B > 32 && !few_tiles && !short_scan)) sprep.sample_stride = 0;
        if (use_mm8) {   // the tiled scan wants the first int8 term, the key scales as one array, and no sample
            sprep.sample_stride = 0;
            sprep.kq = ix->mm_kq;
            cert.sq8 = reinterpret_cast<const ShadowQ*>(ix->sh_sq);
            cert.kshift = ix->sh_kshift;
        }
    }
    return PRAG_OK;
}

static int search_prep(SearchRun& r) {
    SEARCH_BASE(r)
    SEARCH_PLAN(r)
    SEARCH_CERT(r)
    hipLaunchKernelGGL(prep_queries_kernel, dim3((Bpad + 3) / 4, sprep.sample_stride > 0 ? 1 + kShadowSampleSlices : 1), dim3(256), 0,
                       st, q_dev, B, Bpad, ix->d,
                       ix->metric == PRAG_METRIC_COS ? 1 : 0, ix->q32, ix->q16, ix->q16lo, ix->g_tau,
                       use_mm ? ix->mm_cnt : nullptr, ix->mm_ovf,
                       (uint32_t)std::min<int64_t>(ix->ntotal, kMmFirstSeg), ix->qinfo, ix->qn2, flag_word,
                       ix->flag_list, exact_only && ix->ntotal > 0 ? 1 : 0, ix->g_slot, sprep, ix->gate);
    PRAG_LAUNCH_CHECK();

    return PRAG_OK;
}

// <= 128 queries over the 8-bit shadow: scan8 -> exact bound -> gather (flat_shadow.hip)
static int exec_two_level(SearchRun& r) {
    SEARCH_BASE(r)
    SEARCH_PLAN(r)
    SEARCH_CERT(r)
    int rc = PRAG_OK;
    ShadowSearch ss;
    ss.store.rows = ix->rows;
    ss.store.store_f32 = ix->store == PRAG_F32;
    ss.store.d = ix->d;
    ss.store.rows8 = ix->rows8;
    ss.store.sscale = ix->sscale;
    ss.store.serr = ix->serr;
    ss.store.err_max = ix->shadow_err_max;
    ss.store.aff = ix->sh_aff;
    ss.store.yn_max = ix->sh_yn_max;
    ss.store.sbias = ix->sbias;
    ss.store.bias_max = ix->sh_bias_max;
    ss.kshift = ix->sh_kshift;
    ss.xnorm = ix->xnorm;
    ss.N = ix->ntotal;
    ss.d = ix->d;
    ss.metric_l2 = metric_l2;
    ss.alpha = metric_l2 ? -2.0f : -1.0f;
    ss.q32 = ix->q32;
    ss.xn_max = ix->cert_words + 1;
    ss.B = B;
    ss.Bpad_ws = std::min(ix->sh_q_cap, Bpad);
    ss.qt_max = QT;
    ss.k = k;
    ss.kc = kc;
    ss.id_offset = id_offset;
    ss.D = D_dev;
    ss.I = I_dev;
    ss.g_tau = ix->g_tau;
    ss.q8a = ix->sh_q8;
    ss.q8b = ix->sh_q8 + (size_t)ix->sh_q_cap * ix->d;
    ss.sq = ix->sh_sq;
    ss.slots = ix->sh_slots;
    ss.cand = ix->sh_cand;
    ss.ccnt = ix->sh_ccnt;
    ss.cap = kShadowCap;
    ss.wg_slots = ix->n_cu;
    ss.max_wg = cu_budget;
    // 7/8 of the CUs or all of them (<= 64 queries; WgTune above)
    prag_index::WgTune* tune = nullptr;
    bool every_cu = ix->wg_tune_mode == 1;
    if (ix->wg_cap <= 0 && B <= 64 && ix->wg_tune_mode < 0 && ix->adaptive && allow_mm8 && !ix->gate.word) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        prag_index::WgTune& T = ix->wg_tune[B > 32 ? 1 : 0];
        if (!capturing) {
            if (T.pending) {
                if (hipEventQuery(T.ev1) == hipSuccess) {
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, T.ev0, T.ev1) == hipSuccess && ms > 0.f) {
                        T.best[T.pending_arm] = std::min(T.best[T.pending_arm], ms);
                        ++T.phase;
                    }
                    T.pending = false;
                } else {
                    (void)hipGetLastError();
                }
            }
            if (T.rows >= 0 && !T.pending && (ix->ntotal > T.rows + T.rows / 8 || ix->ntotal < T.rows - T.rows / 8)) {
                T.phase = 0;
                T.best[0] = T.best[1] = 1e30f;
            }
            // (shards of under a million rows scan in ~0.1 ms: launch noise, and nothing to gain either way)
            if (T.phase < 8 && !T.pending && ix->ntotal >= (1ll << 20)) {
                if (!T.ev0) {
                    PRAG_HIP(hipEventCreate(&T.ev0));
                    PRAG_HIP(hipEventCreate(&T.ev1));
                }
                if (T.phase == 0) T.rows = ix->ntotal;
                tune = &T;
            }
        }
        // Every CU only when clearly faster - by 2.5 % (embedding-shaped rows show 5-8 %, the noise of four samples is
        // ~1 %), by 5 % when this call has a gate to carry: on 7/8 of the CUs the gate's workgroups run behind the
        // scan's in the scan's launch, which is worth 1-2 % of a 21 M-row pass and 3-5 % of a 2.6 M-row one.
        if (T.phase >= 8) T.choice = T.best[1] < (ix->tail ? 0.95f : 0.975f) * T.best[0] ? 1 : 0;
        every_cu = tune ? (T.phase & 1) != 0 : T.choice == 1;
    }
    ss.auto_wg = ix->wg_cap <= 0;
    ss.every_cu = every_cu;
    if (tune) {
        ss.time_ev0 = tune->ev0;
        ss.time_ev1 = tune->ev1;
        tune->pending = true;
        tune->pending_arm = every_cu ? 1 : 0;
    }
    ss.part_key = ix->sh_pkey;
    ss.part_id = ix->sh_pid;
    ss.ovf = ix->sh_ovf;
    ss.done = ix->sh_ovf + ix->sh_q_cap;
    ss.cert = cert;
    ss.gate = ix->gate;
    // one more small launch; pays once the candidate lists are long (measured: profiles/r04p_exact_bound_ab.txt)
    ss.quad_min_rows = ix->scan8_quad_rows;
    ss.scan_done = allow_mm8 && !ix->gate.word ? ix->scan_done_ev : nullptr;   // (not the gated inner searches)
    ss.tail = allow_mm8 && !ix->gate.word ? ix->tail : nullptr;
    ss.scan_gate_mode = ix->scan_gate_mode;
    ss.unfinished = ix->sh_unfin;
    // (only where the statistics that re-arm it travel: outer 33-128-query searches with device i/o - search_finish)
    ss.skip_gather = ix->adaptive && ix->gather_mode != 1 && !ix->gather_armed && allow_mm8 && !ix->gate.word && io_is_device && B > 32 &&
                     B <= 128 && ix->retry_mode != 0;
    if (ss.scan_done) ix->scan_done_recorded = true;
    ss.exact_bound = k <= 32 && (ix->shadow_bound_mode < 0 ? ix->ntotal >= (1ll << 19) : ix->shadow_bound_mode != 0);
    rc = shadow_search(ss, st, prof);
    if (tune && (rc != PRAG_OK || !ss.timed_recorded)) tune->pending = false;   // no event pair was recorded: nothing to read later
    if (rc != PRAG_OK) return rc;
    if (allow_mm8 && !ix->gate.word) {      // the plan on record says what was launched
        const size_t at = ix->last_plan.find(" grid=");
        if (at != std::string::npos) {
            const size_t end = ix->last_plan.find(' ', at + 1);
            ix->last_plan.replace(at, (end == std::string::npos ? ix->last_plan.size() : end) - at, " grid=" + std::to_string(ss.grid_used));
        }
    }
    reranked = true;
    return PRAG_OK;
}

// <= 128 queries over the stored rows: per-lane-list scan (or the query-stationary kernel) + merge / rerank / certificate
static int exec_list_scan(SearchRun& r) {
    SEARCH_BASE(r)
    SEARCH_PLAN(r)
    SEARCH_CERT(r)
    ScanArgs a;
    a.rows = ix->rows;
    a.xnorm = ix->xnorm;
    a.N = ix->ntotal;
    a.d = ix->d;
    a.qstride = qstride;
    a.n_tiles = n_tiles;
    a.alpha = metric_l2 ? -2.0f : -1.0f;
    a.use_norm = metric_l2;
    a.out_key = ix->part_key;
    a.out_idx = ix->part_idx;
    a.gate = ix->gate;
    auto run_scan = [&](const ScanArgs& sa, int g, EventRing& ring) -> int {
        if (use_qs) return dispatch_qs(ix->d, kc, sa, g, st, ring);
        if (QT == 32 && use_hp)
            return ix->store == PRAG_F32 ? dispatch_scan_kc<32, true, true>(kc, sa, g, st, ring)
                                         : dispatch_scan_kc<32, false, true>(kc, sa, g, st, ring);
        if (QT == 32)
            return ix->store == PRAG_F32 ? dispatch_scan_kc<32, true>(kc, sa, g, st, ring)
                                         : dispatch_scan_kc<32, false>(kc, sa, g, st, ring);
        return ix->store == PRAG_F32 ? dispatch_scan_kc<64, true>(kc, sa, g, st, ring)
                                     : dispatch_scan_kc<64, false>(kc, sa, g, st, ring);
    };
    // Pre-pass over the first kSample rows (same kernels, a few workgroups): per query, the
    // KC-th best key of that SUBSET is a valid upper bound on the shard's KC-th best, so the
    // full scan starts pruned (~0.2 % quantile) instead of inserting at every slot while its
    // per-lane lists warm up.  Only worth it when the shard is much larger than the sample.
    constexpr int64_t kSample = 8192;
    // (the list scan now gets its bound from the slots filled inside the launch; the pre-pass remains
    // for the query-stationary kernel and behind PRAG_PREPASS=1 for A/B timing)
    // Bound for the list scan: slots filled inside the launch (no extra launches: best on shards
    // of a few million rows, where two launches are ~6 % of the search) or the pre-pass (its bound
    // is there from the first tile: measured 1.5 % faster at 21 M rows).  Crossover ~8 M rows.
    const bool use_slots = P.use_slots, prepass = P.prepass;
    static EventRing no_prof;  // the pre-pass is not part of the profiled scan launches
    for (int p0 = 0; p0 < Bpad; p0 += QT) {
        a.q16 = ix->q16 + (size_t)p0 * ix->d;
        a.q16lo = ix->q16lo + (size_t)p0 * ix->d;
        a.g_tau = ix->g_tau + p0;
        a.g_slot = use_slots ? ix->g_slot + (size_t)p0 * kSlotWords : nullptr;
        const int nq = std::min(QT, B - p0);
        int rc;
        if (prepass) {
            ScanArgs pre = a;
            pre.N = kSample;
            pre.n_tiles = (int)(kSample / 32);
            const int pre_grid = use_qs ? (int)(kSample / 128) : (int)(kSample / 256);
            rc = run_scan(pre, pre_grid, no_prof);
            if (rc != PRAG_OK) return rc;
            rc = launch_merge(kc, ix->part_key, ix->part_idx, pre_grid, QT, nq, ix->cand + (size_t)p0 * kc,
                              ix->g_tau + p0, st, nullptr, 0, 0, ix->gate);
            if (rc != PRAG_OK) return rc;
        }
        rc = run_scan(a, grid, prof);
        if (rc != PRAG_OK) return rc;
        // the end of the search for this query tile: list merge + exact rerank + certificate in one launch
        rc = launch_merge_rerank(kc, ix->store == PRAG_F32, ix->part_key, ix->part_idx, n_lists, QT, nq, p0,
                                 ix->rows, ix->d, metric_l2, ix->q32, k, id_offset, D_dev, I_dev, cert, st);
        if (rc != PRAG_OK) return rc;
    }
    reranked = true;
    return PRAG_OK;
}

// candidates (tiled scans, deep lists, the all -1 list of an empty / exact-only search) -> float64 scores, D / I, flags
static int exec_rerank(SearchRun& r) {
    SEARCH_BASE(r)
    SEARCH_PLAN(r)
    SEARCH_CERT(r)
    if (!reranked && kc > 32) {  // deep lists (k > 26): scores + sort in one 1024-thread block per query
        CertArgs dc = cert;
        dc.force = ix->mm_ovf;   // a query whose candidate store overflowed is recomputed by the exact scan
        if (ix->store == PRAG_F32)
            hipLaunchKernelGGL(rerank_sort_kernel<true>, dim3(B), dim3(1024), 0, st, ix->rows, ix->d, metric_l2, ix->q32,
                               ix->cand, kc, k, id_offset, D_dev, I_dev, dc, ix->g_tau);
        else
            hipLaunchKernelGGL(rerank_sort_kernel<false>, dim3(B), dim3(1024), 0, st, ix->rows, ix->d, metric_l2, ix->q32,
                               ix->cand, kc, k, id_offset, D_dev, I_dev, dc, ix->g_tau);
        PRAG_LAUNCH_CHECK();
    } else if (!reranked) {  // empty index / exact-only (all candidates -1), or candidates from the tiled scan
        CertArgs rc_ = cert;
        const uint32_t* kth = (ix->ntotal == 0 || exact_only) ? nullptr : ix->g_tau;
        if (exact_only) rc_.c_acc = -1e30f;   // nothing to certify: every query is already on the flag list
        if (ix->store == PRAG_F32)
            hipLaunchKernelGGL(rerank_kernel<true>, dim3(B), dim3(64 * std::min(kc, 16)), 0, st, ix->rows, ix->d, metric_l2,
                               ix->q32, ix->cand, kc, k, id_offset, D_dev, I_dev, rc_, kth);
        else
            hipLaunchKernelGGL(rerank_kernel<false>, dim3(B), dim3(64 * std::min(kc, 16)), 0, st, ix->rows, ix->d, metric_l2,
                               ix->q32, ix->cand, kc, k, id_offset, D_dev, I_dev, rc_, kth);
        PRAG_LAUNCH_CHECK();
    }
    return PRAG_OK;
}

// what follows the scan: second tier of the int8 tiles, retry tier, exact scan of the flagged queries, host copy-out
static int search_finish(SearchRun& r, float* D, int64_t* I) {
    SEARCH_BASE(r)
    SEARCH_PLAN(r)
    SEARCH_CERT(r)
    if (ix->scan_done_ev && !use_shadow && allow_mm8 && !ix->gate.word) {    // (the two-level search recorded it behind scan8)
        PRAG_HIP(hipEventRecord(ix->scan_done_ev, st));
        ix->scan_done_recorded = true;
    }
    // ---- exact float64 scan for the queries on the flag list --------------------------------------
    // Device i/o: always enqueued, the kernels return at once when the list is empty (~3 us each, no
    // host round trip).  Host i/o synchronises anyway: look at the flag count first and launch only
    // when something was flagged.
    ExactRun er;
    er.rows = ix->rows;
    er.store_f32 = ix->store == PRAG_F32;
    er.N = ix->ntotal;
    er.d = ix->d;
    er.metric_l2 = metric_l2;
    er.q32 = ix->q32;
    er.n_flag = flag_word;
    er.flag_list = ix->flag_list;
    er.B = B;
    er.k = k;
    er.id_offset = id_offset;
    er.D = D_dev;
    er.I = I_dev;
    er.part_key = ix->ex_key;
    er.part_id = ix->ex_id;
    er.f_cap = ex_fcap;
    er.grid = ex_grid;
    er.done = ix->ex_done;
    er.gpool = ix->ex_pool;
    er.xn_max = ix->cert_words + 1;
    er.tag_ids = tag_ids;
    er.gate = ix->gate;
    // several flagged queries expected - every query goes to the exact scan by construction, or recent searches on this
    // handle flagged some (the armed retry tier's inner searches included): eight queries per pass over the rows
    // (one flagged query is the common case and the single-query kernel is 3.7x faster for it - profiles/
    //  r05f_exact_group_bench.txt -, so "several" means: by construction, or the retry tier's inner searches have been
    //  leaving >= 4 queries flagged lately)
    er.mfma = ix->exact_mfma_mode != 0;
    // ... or a shape the matrix-pipe form covers (round 6): it decides on the device - up to eight flagged queries of a
    // group share a pass on the four-block float64 MFMA, nine to sixteen on the full tile, and ONE costs 1.23-1.29 ms over
    // 4 M x 640 fp16 rows where the one-query kernel takes 1.43-1.46 (float32 rows 1.98 against 2.45) - so no history is
    // needed, the launch sequence is the same whatever earlier searches flagged, and a single query takes it too
    er.grouped = ix->exact_group_mode != 0 &&
                 (ix->exact_group_mode == 1 || (exact_only && B >= 2) || (ix->adaptive && ix->exact_group_hint) ||
                  (er.mfma && exact_mfma_supported(ix->d, k)));
    const bool may_flag = certify && ix->ntotal > 0;
    if (use_mm8) {
        // second tier, decided on the device (mm8_second_tier): no read-back, no host branch
        const int rc = mm8_second_tier(ix, q_dev, B, k, id_offset, D_dev, I_dev, stream, tag_ids, flag_word);
        if (rc != PRAG_OK) return rc;
        ix->last_flagged = -1;
        if (io_is_device) return PRAG_OK;
        // host i/o: one transfer back, then the statistics are there too
        PRAG_HIP(hipMemcpyAsync(ix->io_res_host, ix->io_res, (size_t)B * k * 12, hipMemcpyDeviceToHost, st));
        PRAG_HIP(hipStreamSynchronize(st));
        consume_tier_stats(ix, true);
        memcpy(I, ix->io_res_host, (size_t)B * k * sizeof(int64_t));
        memcpy(D, ix->io_res_host + (size_t)B * k * 8, (size_t)B * k * sizeof(float));
        return PRAG_OK;
    }
    // Retry tier (retry_tier above): 33-128 queries, outer searches only.  Device i/o: armed by what earlier searches
    // flagged (the count travels back asynchronously); host i/o: decided on the count that came back with the results.
    const bool retry_shape = allow_mm8 && !ix->gate.word && may_flag && !exact_only && !use_mm && B > 32 && B <= 128 &&
                             ix->retry_mode != 0;
    if (io_is_device) {
        ix->last_flagged = -1;
        const bool retry_now = retry_shape && ((ix->adaptive && ix->retry_armed) || ix->retry_mode == 1);
        bool recorded_now = false;
        if (retry_shape && ix->adaptive) {      // statistics for the next search's decision; never waited for, nothing while capturing
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
            // (a record still on its way is not overwritten: a caller that enqueues passes faster than the device runs
            //  them would never see one arrive - the next record is made once this one has been looked at)
            if (!capturing && !ix->r2_pending) {
                recorded_now = true;
                if (!ix->r2_word_host) {
                    PRAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&ix->r2_word_host), 4 * sizeof(uint32_t)));
                    memset(ix->r2_word_host, 0, 4 * sizeof(uint32_t));
                    PRAG_HIP(hipEventCreateWithFlags(&ix->r2_event, hipEventDisableTiming));
                }
                PRAG_HIP(hipMemcpyAsync(ix->r2_word_host, flag_word, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                if (use_shadow && ix->sh_unfin) {      // ... and how many queries the bound kernel left to the gather
                    PRAG_HIP(hipMemcpyAsync(ix->r2_word_host + 3, ix->sh_unfin, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                    ix->r2_has_unfinished = true;
                }
                PRAG_HIP(hipEventRecord(ix->r2_event, st));
                ix->r2_pending = true;
            }
        }
        if (retry_now) {
            const int rc = retry_tier(ix, q_dev, B, k, id_offset, D_dev, I_dev, stream, tag_ids, flag_word);
            if (rc != PRAG_OK) return rc;
            if (recorded_now) {         // (not capturing) the tier's own outcome travels back with the flag count
                PRAG_HIP(hipMemcpyAsync(ix->r2_word_host + 1, ix->r2_word + 1, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                PRAG_HIP(hipEventRecord(ix->r2_event, st));
            }
        } else if (may_flag) {
            const int rc = exact_run(er, st);
            if (rc != PRAG_OK) return rc;
        }
        return PRAG_OK;
    }
    const size_t res_bytes = (size_t)B * k * 12 + 4;
    auto fetch = [&]() -> int {
        PRAG_HIP(hipMemcpyAsync(ix->io_res_host, ix->io_res, res_bytes, hipMemcpyDeviceToHost, st));
        PRAG_HIP(hipStreamSynchronize(st));
        return PRAG_OK;
    };
    int rc_io = fetch();
    if (rc_io != PRAG_OK) return rc_io;
    uint32_t n_flag = 0;
    memcpy(&n_flag, ix->io_res_host + (size_t)B * k * 12, sizeof(n_flag));
    ix->last_flagged = (int)n_flag;
    if (may_flag && n_flag > 0) {
        if (retry_shape) {
            ix->retry_armed = ix->adaptive != 0;   // (device-io searches on this handle start armed too)
            ix->retry_clean = 0;
            const int rc = retry_tier(ix, q_dev, B, k, id_offset, D_dev, I_dev, stream, tag_ids, flag_word);
            if (rc != PRAG_OK) return rc;
        } else {
            const int rc = exact_run(er, st);
            if (rc != PRAG_OK) return rc;
        }
        rc_io = fetch();
        if (rc_io != PRAG_OK) return rc_io;
        if (retry_shape) {      // what the exact scan recomputed in the end (retry_finish_kernel)
            memcpy(&n_flag, ix->io_res_host + (size_t)B * k * 12, sizeof(n_flag));
            ix->last_flagged = (int)n_flag;
            ix->exact_group_hint = n_flag >= 4u;
        }
    }
    memcpy(I, ix->io_res_host, (size_t)B * k * sizeof(int64_t));
    memcpy(D, ix->io_res_host + (size_t)B * k * 8, (size_t)B * k * sizeof(float));
    return PRAG_OK;
}

int index_search_impl(prag_index_t* ix, const float* q, int B, int k, int64_t id_offset, float* D, int64_t* I,
                             int io_is_device, void* stream, int tag_ids, bool allow_mm8) {
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    PRAG_REQUIRE(B >= 0 && k >= 1, PRAG_EINVAL, "B=%d k=%d", B, k);
    if (B == 0) return PRAG_OK;
    PRAG_REQUIRE(q && D && I, PRAG_EINVAL, "prag_index_search: NULL pointer");
    // ---- plan: every dispatch decision, as a pure function of the shape and the index state (plan_search) --------
    PlanEnv env;
    env.d = ix->d; env.metric = ix->metric; env.store = ix->store; env.ntotal = ix->ntotal; env.B = B; env.k = k;
    env.kc_min = ix->kc_min; env.hp_mode = ix->hp_mode; env.mm_mode = ix->mm_mode; env.mm8_mode = ix->mm8_mode;
    env.cert_mode = ix->cert_mode; env.prepass_mode = ix->prepass_mode; env.wg_cap = ix->wg_cap; env.n_cu = ix->n_cu;
    env.shadow_mode = ix->shadow_mode; env.mm8_min_rows = ix->mm8_min_rows; env.allow_mm8 = allow_mm8;
    env.shadow_ready = ix->ntotal > 0 && ix->rows8 != nullptr && ix->shadow_rows == ix->ntotal && shadow_wanted(ix);
    if (allow_mm8) {
        // (not while `stream` is being captured: hipEventQuery is not a capture-safe call - it invalidated the capture of
        //  a search that followed an uncaptured one; the statistics wait for the next uncaptured search)
        hipStreamCaptureStatus cs0 = hipStreamCaptureStatusNone;
        const bool capturing0 = hipStreamIsCapturing(reinterpret_cast<hipStream_t>(stream), &cs0) == hipSuccess &&
                                cs0 != hipStreamCaptureStatusNone;
        if (!capturing0) {
            consume_tier_stats(ix, false);     // the previous search's failed count, if it has arrived
            consume_retry_stats(ix, false);    // ... and the flag count of the previous <= 128-query search
        }
    }
    if (!ix->adaptive) ix->mm8_auto_off = false;
    env.mm8_auto_off = ix->mm8_auto_off;
    SearchPlan P = plan_search(env);
    PRAG_REQUIRE(P.kc != 0, PRAG_EUNSUPPORTED, "k=%d: at most 911 results per query", k);
    if (P.mm8_eligible && ix->mm8_auto_off && ++ix->mm8_off_count >= ix->mm8_off_period) {
        // one probe after a while: a serving index must not lose the 8-bit tiles for good over two bad batches
        ix->mm8_auto_off = false;
        ix->mm8_whole_batch_streak = 1;      // a single whole-batch repeat switches them off again ...
        ix->mm8_off_period = std::min(4096, ix->mm8_off_period * 2);   // ... for twice as long
        ix->mm8_off_count = 0;
        env.mm8_auto_off = false;
        P = plan_search(env);
    }
    if (allow_mm8) {                         // (second-tier inner searches keep the outer search's plan on record)
        char buf[640];
        plan_describe(env, P, buf, (int)sizeof(buf));
        ix->last_plan = buf;
        ix->last_plan += ix->adaptive ? " adaptive=1" : " adaptive=0";
        if (!P.use_mm8) {
            ix->tier_pending = false;
            ix->mm8_last_failed = P.mm8_eligible ? -2 : -1;
        }
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);


    SearchRun r{};
    r.ix = ix; r.P = P; r.B = B; r.k = k; r.id_offset = id_offset; r.io_is_device = io_is_device; r.stream = stream;
    r.tag_ids = tag_ids; r.allow_mm8 = allow_mm8; r.st = st; r.prof = &ix->prof; r.reranked = false;
    int rc = search_stage_io(r, q, D, I);
    if (rc == PRAG_OK) rc = search_workspaces(r);
    // max ||x||^2 and the shadow are kept up to date by add / prepare: a no-op unless set_shadow changed the mode
    if (rc == PRAG_OK) rc = shadow_ensure(ix, st);
    if (rc != PRAG_OK) return rc;
    search_certificate(r);
    rc = search_shadow_args(r);
    if (rc == PRAG_OK) rc = search_prep(r);
    if (rc != PRAG_OK) return rc;
    // ---- the corpus pass of the plan's kernel family ------------------------------------------------------------------
    if (P.use_shadow) {
        rc = exec_two_level(r);
    } else if (ix->ntotal == 0 || P.exact_only) {
        PRAG_HIP(hipMemsetAsync(ix->cand, 0xFF, P.cand_need * sizeof(int), st));  // all -1
    } else if (P.use_mm) {
        rc = search_tiled(ix, B, P.Bpad, P.kc, P.qstride, P.n_tiles, P.cu_budget, P.mm_chunk, P.mm_cap_wg, st, P.use_mm8);
    } else {
        rc = exec_list_scan(r);
    }
    if (rc == PRAG_OK) rc = exec_rerank(r);
    if (rc != PRAG_OK) return rc;
    return search_finish(r, D, I);
}

extern "C" int prag_index_reconstruct(prag_index_t* ix, int64_t row0, int64_t n, float* out_host) {
    PRAG_REQUIRE(ix != nullptr && out_host != nullptr, PRAG_EINVAL, "prag_index_reconstruct: NULL pointer");
    PRAG_REQUIRE(row0 >= 0 && n >= 0 && row0 + n <= ix->ntotal, PRAG_EINVAL, "rows [%lld,%lld) outside [0,%lld)",
                 (long long)row0, (long long)(row0 + n), (long long)ix->ntotal);
    if (n == 0) return PRAG_OK;
    if (ix->store == PRAG_F32) {  // stored as given: straight copy, no device temporary
        PRAG_HIP(hipMemcpy(out_host, reinterpret_cast<const float*>(ix->rows) + (size_t)row0 * ix->d,
                           (size_t)n * ix->d * sizeof(float), hipMemcpyDeviceToHost));
        return PRAG_OK;
    }
    // fp16 rows: widen through a bounded (<= 64 MB) device staging buffer, chunk by chunk
    const int64_t chunk = std::min<int64_t>(n, std::max<int64_t>(1, (64ll << 20) / ((int64_t)ix->d * 4)));
    float* tmp = nullptr;
    PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)chunk * ix->d * sizeof(float)));
    hipError_t e = hipSuccess;
    for (int64_t o = 0; o < n && e == hipSuccess; o += chunk) {
        const int64_t m = std::min(chunk, n - o), ne = m * ix->d;
        const char* src = reinterpret_cast<const char*>(ix->rows) + (size_t)(row0 + o) * ix->d * elt(ix);
        const int blocks = (int)std::min<int64_t>((ne + 255) / 256, 4096);
        hipLaunchKernelGGL(reconstruct_kernel<false>, dim3(blocks), dim3(256), 0, 0, src, ne, tmp);
        e = hipMemcpy(out_host + (size_t)o * ix->d, tmp, (size_t)ne * sizeof(float), hipMemcpyDeviceToHost);
    }
    (void)hipFree(tmp);
    PRAG_HIP(e);
    return PRAG_OK;
}

extern "C" int prag_index_last_fallbacks(prag_index_t* ix, void* stream, int* n_out) {
    PRAG_REQUIRE(ix != nullptr && n_out != nullptr, PRAG_EINVAL, "prag_index_last_fallbacks: NULL pointer");
    if (ix->last_flagged >= 0) {   // host-io search: the count came back with the results
        *n_out = ix->last_flagged;
        return PRAG_OK;
    }
    uint32_t n = 0;
    PRAG_HIP(hipMemcpyAsync(&n, ix->cert_words, sizeof(n), hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)));
    PRAG_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    *n_out = (int)n;
    return PRAG_OK;
}

extern "C" int prag_index_last_plan(prag_index_t* ix, char* out, int cap) {
    PRAG_REQUIRE(ix != nullptr && out != nullptr && cap > 0, PRAG_EINVAL, "prag_index_last_plan: bad argument");
    snprintf(out, cap, "%s", ix->last_plan.c_str());
    return PRAG_OK;
}

extern "C" int prag_index_last_tiled8(prag_index_t* ix, int* n_failed_out) {
    PRAG_REQUIRE(ix != nullptr && n_failed_out != nullptr, PRAG_EINVAL, "prag_index_last_tiled8: NULL pointer");
    consume_tier_stats(ix, true);     // the count travels asynchronously: a measurement hook may wait for it
    *n_failed_out = ix->mm8_last_failed;
    return PRAG_OK;
}

extern "C" int prag_index_set_candidate_depth(prag_index_t* ix, int depth) {
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    PRAG_REQUIRE(depth == 0 || depth == 8 || depth == 16 || depth == 32, PRAG_EINVAL,
                 "candidate depth %d: use 0 (default), 8, 16 or 32", depth);
    ix->kc_min = depth;
    return PRAG_OK;
}

extern "C" int prag_index_set_shadow(prag_index_t* ix, int mode) {
    PRAG_REQUIRE(ix != nullptr && mode >= 0 && mode <= 2, PRAG_EINVAL, "prag_index_set_shadow: mode %d (0, 1 or 2)", mode);
    ix->shadow_mode = mode;
    ix->shadow_no_room = false;
    ix->shadow_failed = false;
    ix->mm8_whole_batch_streak = 0;
    ix->mm8_auto_off = false;
    ix->mm8_off_count = 0;
    ix->mm8_off_period = 64;
    return PRAG_OK;
}

extern "C" int prag_index_set_adaptive(prag_index_t* ix, int on) {
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    ix->adaptive = on != 0;
    if (!ix->adaptive) {        // history is dropped with the switch: the next search plans as a fresh handle would
        ix->retry_armed = false;
        ix->retry_clean = 0;
        ix->exact_group_hint = false;
        ix->gather_armed = true;
        ix->gather_clean = 0;
        ix->mm8_auto_off = false;
        ix->mm8_off_count = 0;
        ix->mm8_off_period = 64;
        ix->mm8_whole_batch_streak = 0;
    }
    return PRAG_OK;
}

extern "C" int prag_index_set_scan_workgroups(prag_index_t* ix, int n_workgroups) {
    PRAG_REQUIRE(ix != nullptr && n_workgroups >= 0, PRAG_EINVAL, "prag_index_set_scan_workgroups: bad argument");
    ix->wg_cap = n_workgroups;
    return PRAG_OK;
}

// One pass of the retrieval-gating hot path as ONE call: the top-k of B queries over this index AND the gate over the
// next batch of pooled states (exp_rag.py:406-415, 432-436).  The two are independent; when the search is a two-level
// search the gate's prober workgroups ride in the launch of its bound kernel (bound_gate_kernel, flat_shadow.hip) - the
// search's tail occupies a quarter of the chip - otherwise (direct scans, tiled scans, gate shapes prober16_kernel does
// not take) the call is prag_index_search followed by prag_gate.  Same results as the two calls, always.
extern "C" int prag_search_and_gate(prag_index_t* ix, const float* q_dev, int B, int k, int64_t id_offset, float* D_dev,
                                    int64_t* I_dev, prag_prober_t* p, const void* x_dev, int x_dtype, int64_t x_layer_stride,
                                    int Bg, int ablation, double theta, float* logits_dev, float* probsum_dev,
                                    int32_t* decision_dev, int flags, void* stream) {
    PRAG_REQUIRE(ix != nullptr && p != nullptr, PRAG_EINVAL, "prag_search_and_gate: NULL handle");
    const int tag_ids = flags & 1;
    PRAG_REQUIRE(!tag_ids || (id_offset >= 0 && id_offset + ix->ntotal < (1ll << kTagShift)), PRAG_EUNSUPPORTED,
                 "tagged ids hold %d-bit global row ids", kTagShift);
    PRAG_REQUIRE(logits_dev && probsum_dev && decision_dev, PRAG_EINVAL, "prag_search_and_gate: NULL gate output");
    TailGate tg;
    const bool have = Bg >= 1 && prober_describe_tail(p, x_dev, x_dtype, x_layer_stride, Bg, logits_dev, ablation, theta,
                                                      probsum_dev, decision_dev, &tg);
    ix->tail = have ? &tg : nullptr;
    const int rc = index_search_impl(ix, q_dev, B, k, id_offset, D_dev, I_dev, 1, stream, tag_ids);
    ix->tail = nullptr;
    if (rc != PRAG_OK) return rc;
    // (the record of what ran: the scan's launch was scan8_gate_kernel / the bound kernel's was bound_gate_kernel)
    if (have && tg.taken) ix->last_plan += tg.in_scan ? " gate_launch=scan8_gate_kernel" : " gate_launch=bound_gate_kernel";
    if (Bg < 1) return PRAG_OK;
    if (have && tg.taken) {     // the logits are on their way: softmax / sum over layers / threshold (exp_rag.py:407-415)
        if (tg.gate_folded || tg.finished) return PRAG_OK;     // ... done by the last prober workgroup of every row tile / by the bound launch
        return prag_gate_from_logits(logits_dev, tg.pa.n_run, Bg, ablation, theta, probsum_dev, decision_dev, stream);
    }
    return prag_gate(p, x_dev, x_dtype, x_layer_stride, Bg, ablation, theta, logits_dev, probsum_dev, decision_dev, stream);
}

extern "C" int prag_index_stream_wait_scan(prag_index_t* ix, void* other_stream) {
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    if (!ix->scan_done_ev) {     // first call: searches record the event from now on; nothing to wait for yet
        PRAG_HIP(hipEventCreateWithFlags(&ix->scan_done_ev, hipEventDisableTiming));
        ix->scan_done_recorded = false;
        return PRAG_OK;
    }
    if (ix->scan_done_recorded)
        PRAG_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(other_stream), ix->scan_done_ev, 0));
    return PRAG_OK;
}

// Measurement hook: candidates the scan of the most recent two-level search (<= 128 queries: its LAST query tile)
// handed to the exact rerank - per query, summed over the scan's workgroups - i.e. how tight the proof-carrying filter
// was on this corpus.  Synchronises `stream`.  n_queries_out = 0: the last search did not take the two-level path.
extern "C" int prag_index_last_survivors(prag_index_t* ix, void* stream, int64_t* total_out, int* max_per_query_out,
                                         int* n_queries_out) {
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    if (total_out) *total_out = 0;
    if (max_per_query_out) *max_per_query_out = 0;
    if (n_queries_out) *n_queries_out = 0;
    if (ix->last_plan.find("family=scan8_kernel") == std::string::npos || !ix->sh_ccnt) return PRAG_OK;
    int QT = 64, grid = 0, B = 0;
    {
        const char* p = strstr(ix->last_plan.c_str(), " QT=");
        if (p) QT = atoi(p + 4);
        p = strstr(ix->last_plan.c_str(), " grid=");
        if (p) grid = atoi(p + 6);
        p = strstr(ix->last_plan.c_str(), " queries=");
        if (p) B = atoi(p + 9);
    }
    PRAG_REQUIRE(grid >= 1 && grid <= ix->n_cu && (QT == 32 || QT == 64 || QT == 128), PRAG_ESTATE, "unreadable plan");
    PRAG_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    std::vector<uint32_t> h((size_t)grid * QT);
    PRAG_HIP(hipMemcpy(h.data(), ix->sh_ccnt, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    const int nq = std::min(QT, B - (B - 1) / QT * QT);       // real queries of the last tile
    int64_t total = 0;
    int mx = 0;
    for (int q = 0; q < nq; ++q) {
        int64_t c = 0;
        for (int g = 0; g < grid; ++g) c += h[(size_t)g * QT + q];
        total += c;
        mx = (int)std::max<int64_t>(mx, c);
    }
    if (total_out) *total_out = total;
    if (max_per_query_out) *max_per_query_out = mx;
    if (n_queries_out) *n_queries_out = nq;
    return PRAG_OK;
}

extern "C" int prag_index_profile(prag_index_t* ix, int slots) {
    PRAG_REQUIRE(ix != nullptr && slots >= 0 && slots <= 4096, PRAG_EINVAL, "prag_index_profile: bad argument");
    if (slots == 0) {
        ix->prof.disable();
        ix->prof_xch.disable();
        return PRAG_OK;
    }
    if (ix->comm) {                 // sharded searches: the exchange step gets a ring of its own
        const int rc = ix->prof_xch.enable(slots);
        if (rc != PRAG_OK) return rc;
    }
    return ix->prof.enable(slots);
}

extern "C" int prag_index_profile_read(prag_index_t* ix, float* ms, int cap, int* n_out) {
    PRAG_REQUIRE(ix != nullptr && ms != nullptr && cap >= 0, PRAG_EINVAL, "prag_index_profile_read: bad argument");
    return ix->prof.read(ms, cap, n_out);
}

extern "C" void prag_index_destroy(prag_index_t* ix) {
    if (!ix) return;
    ix->prof.disable();
    ix->prof_xch.disable();
    if (ix->scan_done_ev) (void)hipEventDestroy(ix->scan_done_ev);
    for (auto& T : ix->wg_tune)
        for (hipEvent_t ev : {T.ev0, T.ev1})
            if (ev) (void)hipEventDestroy(ev);
    if (ix->r2_event) (void)hipEventDestroy(ix->r2_event);
    if (ix->r2_word_host) (void)hipHostFree(ix->r2_word_host);
    for (void* p : {(void*)ix->r2_word, (void*)ix->r2_list, (void*)ix->r2_q, (void*)ix->r2_D, (void*)ix->r2_I})
        if (p) (void)hipFree(p);
    void* ptrs[] = {ix->rows, ix->xnorm, ix->q32, ix->q16, ix->q16lo, ix->g_tau, ix->part_key, ix->part_idx, ix->cand,
                    ix->io_q, ix->io_res, ix->mm_cnt, ix->mm_ovf, ix->mm_kq, ix->mm_ckey, ix->mm_cidx, ix->t2_list, ix->t2_q,
                    ix->t2_D, ix->t2_I, ix->t2_word, ix->xch_send, ix->xch_recv,
                    ix->mm_wcnt, ix->mm_wkey, ix->mm_widx, ix->rows16, ix->qinfo, ix->qn2, ix->flag_list, ix->g_slot,
                    ix->cert_words, ix->ex_key, ix->ex_id, ix->rows8, ix->sscale, ix->serr, ix->shadow_err_max, ix->sh_aff, ix->sh_aff_sums, ix->sh_yn_max, ix->sh_bias_max, ix->sbias, ix->sh_kshift, ix->sh_unfin, ix->sh_q8,
                    ix->sh_sq, ix->sh_slots, ix->sh_ovf, ix->sh_cand, ix->sh_ccnt, ix->sh_pkey, ix->sh_pid, ix->ex_done, ix->ex_pool};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (ix->io_q_host) (void)hipHostFree(ix->io_q_host);
    if (ix->io_res_host) (void)hipHostFree(ix->io_res_host);
    if (ix->tier_word_host) (void)hipHostFree(ix->tier_word_host);
    if (ix->tier_event) (void)hipEventDestroy(ix->tier_event);
    delete ix;
}
