// scan8_kernel's body as a device function (round 5): one workgroup of the two-level search's scan over the 8-bit shadow.
// It lives in a header because TWO kernels run it: scan8_kernel (flat_shadow.hip) and scan8_gate_kernel
// (flat_scan_gate.hip), which carries the gate's prober workgroups in the SAME launch behind the scan's - an HBM-bound
// scan is fastest on 7/8 of the CUs, and the gate of the next batch depends on nothing in the search.  `block_idx` /
// `grid_dim` replace blockIdx.x / gridDim.x (the scan's workgroups only); `smem` is the workgroup's dynamic LDS
// (scan8_lds_bytes(QT, qstride) bytes, 16-byte aligned).
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "flat_internal.h"

namespace prag {

typedef int i32x16 __attribute__((ext_vector_type(16)));

// LDS of the scan: query planes (both terms for 32-query tiles), 4 KiB + 384 B of row metadata per wave, bounds /
// counters.  (The 4 KiB per wave were the staging buffers of the row-major chunks; with the shadow in MFMA operand
// order the rows never pass through LDS and the block only holds the pre-bound's 32 sample slots per query while
// the kernel starts - and the stages of the -DPRAG_SHADOW_CHUNK_MAJOR A/B build.)
inline int scan8_lds_bytes(int QT, int qstride) {
    return (QT == 32 ? 2 : 1) * QT * qstride + 8 * 4096 + 8 * 384 + 4 * QT * 4 + 64 + (QT == 128 ? 3 * QT * 4 + 64 : 0);
}
// staging sets of the one-epilogue-copy form of the 64-query quad-test tiles (rows of a multiple of 128 * 3 elements)
constexpr int kScan8Aln = 3;

// Timing experiments that give WRONG results (warm-up off, nothing collected, gather stages off) exist only
// in the `make diag` build (libprag_diag.so, -DPRAG_MM_DIAG); in libprag.so the knob is the constant 0 and
// the environment is never read.
#ifdef PRAG_MM_DIAG
#define PRAG_SH_DBG(x) (x)
#else
#define PRAG_SH_DBG(x) 0
#endif

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
// flat_scan_gate.hip: the scan (64-query tiles, 8- or 16-deep lists) with the gate's prober workgroups behind it in one launch
struct TailGate;
bool scan8_gate_supported(int kc, int ct16);
int launch_scan8_gate(const Scan8Args& a, int grid, int kc, bool quad, const TailGate& t, hipStream_t st, EventRing& prof);

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
__device__ __forceinline__ void scan8_body(const Scan8Args& a, char* const smem, const int block_idx, const int grid_dim) {
    if (gate_closed(a.gate)) return;
#ifdef PRAG_MM_DIAG
    const unsigned long long stamp_entry = a.stamps ? wall_clock64() : 0ull;
#define S8_STAMP(i) do { if (a.stamps && (threadIdx.x & 63) == 0) \
        a.stamps[((int64_t)block_idx * 8 + (threadIdx.x >> 6)) * kScan8Stamps + (i)] = wall_clock64(); } while (0)
#define S8_VALUE(i, v) do { if (a.stamps && (threadIdx.x & 63) == 0) \
        a.stamps[((int64_t)block_idx * 8 + (threadIdx.x >> 6)) * kScan8Stamps + (i)] = (unsigned long long)(v); } while (0)
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

    const int nW = grid_dim * 8;
    // tile t belongs to workgroup t mod grid (wave (t / grid) mod 8): CONSECUTIVE tiles go to different
    // workgroups, so a run of similar rows (a corpus in article order: 1024 contiguous near-duplicates are 32
    // tiles) spreads over 32 candidate regions instead of filling the regions of 4 workgroups (measured on such
    // a corpus with the wave-major map: all 64 queries overflowed into the exact scan, 104 ms per search
    // against 1.1 ms for the direct scan)
    const int gw = w * grid_dim + block_idx;
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

    // staging geometry of the row-major chunks (A/B build only): 4 x 16-B loads per lane per 128-byte chunk of 32 rows
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
    // one 4-KiB chunk (32 rows x 128 bytes), 4 k-steps: the registers its loads filled ARE the A operands (row-major
    // chunks of the A/B build pass through LDS first); they are refilled with chunk (tile_nx, c_nx)
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
                        a.cand[((int64_t)block_idx * QT + 32 * t + r) * a.cap + slot] = int2{(int)doc, (int)__float_as_uint(lo)};
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
                                            a.cand[((int64_t)block_idx * QT + 32 * t + rq) * a.cap + slot] =
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
                        const int n_active = min(8, max(0, (a.n_tiles - block_idx + grid_dim - 1) / grid_dim));   // waves of this workgroup that own a tile
                        uint32_t old = 0;
                        if (lane == 0) old = atomicAdd(s_arrive, 1u);
                        old = (uint32_t)__shfl((int)old, 0, 64);
                        if ((int)old + 1 == n_active) {
                            for (int qq = lane; qq < QT; qq += 64) {
                                const uint32_t v = s_best[qq];
                                if (v != 0xFFFFFFFFu)
                                    (void)__hip_atomic_fetch_min(a.g_slot + (qq * kShadowSlotRows + 0) * 32 + (block_idx % a.kslots), v,
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
                                (void)__hip_atomic_fetch_min(a.g_slot + (qq * kShadowSlotRows + epoch) * 32 + (block_idx % a.kslots), v,
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
    if (a.stamps && lane == 0) a.stamps[((int64_t)block_idx * 8 + w) * kScan8Stamps + 0] = stamp_entry;
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
        a.ccnt[(int64_t)block_idx * QT + tid] = s_ccnt[tid];
        // the final bound of this workgroup: the gather drops candidates that a later, tighter bound excludes
        (void)__hip_atomic_fetch_min(a.g_tau + tid, s_tau[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef PRAG_MM_DIAG
    if (a.stamps && tid == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < QT; ++i) tot += s_ccnt[i];
        a.stamps[((int64_t)block_idx * 8) * kScan8Stamps + 15] = tot;
    }
#endif
    S8_STAMP(12);
#undef S8_STAMP
#undef S8_VALUE
}

}  // namespace prag
