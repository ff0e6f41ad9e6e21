// MFMA-tiled flat scan for large query batches (> 128 queries) and deep result lists (k > 26) on
// MI355X (gfx950):
// BASELINE config 3 (1k queries x 1M rows x 768, "MFMA tile") - the regime where the
// [queries x d] x [d x rows] contraction, not the HBM stream, bounds index.search
// (reference call site: utils.py:378-380 batch_topk_sim -> faiss IndexFlat.search).
//
// scan_mm_kernel: persistent workgroups (8 waves = 2 per SIMD, 1 workgroup per CU) walk
// 256-row x 256-query tiles in an XCD-aware order (the workgroups of one XCD take the query
// blocks of the same row tile at the same time, so the rows come from HBM once and from that
// XCD's L2 afterwards).  Both operands stream through LDS by LDS-DMA (global_load_lds_dwordx4,
// source-side XOR swizzle, no staging registers) in 64-wide K tiles, double buffered; every
// wave owns a 128-row x 64-query block of accumulators (8 x v_mfma_f32_32x32x16_f16 tiles).
// A K tile is four phases {fragment reads for one 64x32 quadrant | 16 KiB of DMA for a K tile
// two ahead | counted vmcnt | barrier | 8 MFMAs | barrier}; the two wave groups (row halves)
// run one barrier apart, so one group's MFMAs cover the other group's LDS reads.  Every LDS
// region is read in exactly one phase, restaged two or three phases later and waited for one
// phase before its next read: four to five phases of flight time per DMA, uniform vmcnt(8).
//
// [B,N] never exists: after the last K tile each lane compares its 128 scores with the
// per-query pruning bound and appends the rare survivors (key, row) to its workgroup's own
// region of a candidate store: the slot comes from a counter in LDS, the stores are
// fire-and-forget, so a survivor costs no global round trip.  The corpus is scanned in segments
// of geometrically growing size (2048 rows - all of them candidates - then x16):
// mm_compact_kernel reduces every query's candidates to its KC best after each segment and
// tightens the bound to the KC-th best key seen so far, so a segment contributes ~15*KC
// candidates per query when rows are exchangeable.  A region that overflows anyway (rows sorted
// by decreasing distance, say) raises a per-query flag and the caller re-runs those queries
// through the per-lane-list kernels, which cannot overflow (k <= 26; deeper lists report it).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "flat_internal.h"

namespace prag {

struct MmArgs {
    const _Float16* rows;
    const float* xnorm;
    const _Float16* q16;
    int64_t row0, row1;   // this segment: rows [row0,row1), row0 % 256 == 0
    int n_qb;             // query blocks of 256
    int n_tiles;          // row tiles x query blocks
    float alpha;
    int use_norm;
    const uint32_t* tau;  // [n_qb*256]
    float inv_alpha;      // 1 / alpha (exact: alpha is -1 or -2)
    float* ckey;          // [Bpad][cap_q]
    int* cidx;
    int cap_q;
    int Bpad;             // n_qb * 256
    int wg_stride;        // workgroup slots per query in wcnt
    uint32_t* wcnt;       // [Bpad][wg_stride]     survivors per (query, workgroup) of this launch
    float* wkey;          // [grid][Bpad][cap_wg]  their keys / rows
    int* widx;
    int cap_wg;
    unsigned long long* dbg;  // diagnostic builds: {core-clock ticks, 100 MHz ticks} of block 0; else null
    // 8-bit selection (I8 kernels): shadow rows, their scales, first int8 query term, per-query key scale
    const signed char* rows8;
    const float* sscale;
    const signed char* q8;
    const float* kq;
    Gate gate;
};

// LDS map (bytes): A even/odd K tile at 0 / 32 KiB, B even/odd at 64 / 96 KiB ([256 rows][128 B],
// 16-B pieces XOR-swizzled with (row>>1)&7), then 1 KiB of row norms and 1 KiB of bounds.
// (8-bit selection: two more KiB - the row scales of the tile and the key scales of the query block)
constexpr int kMmLdsXn = 131072;
constexpr int kMmLdsTau = 131072 + 1024;
constexpr int kMmLdsSs = 131072 + 2048;
constexpr int kMmLdsKq = 131072 + 3072;
constexpr int kMmLdsCnt = 131072 + 4096;   // [Bpad] survivor counters of this workgroup

// ABL != 0: timing-only ablations (wrong results) for tools/mm_ablate.py, built with -DPRAG_MM_DIAG;
// bit 0 no MFMAs, bit 1 no LDS-DMA, bit 2 no fragment reads, bit 3 no filter, bit 4 vmcnt(14)
// instead of vmcnt(8) (reads may race the DMA), bit 5 no barrier after the MFMA block, bit 6 the
// int8 MFMA on the same bytes (with NKT = d/128: what an 8-bit shadow of the rows would cost)
// MODE 0: the launch over the first segment (no bound yet; every row becomes a candidate);
// MODE 1: inner product / cosine (key = -score); MODE 2: squared L2 (key = ||x||^2 - 2 score).
// I8: the same pipeline over the 8-bit shadow of the rows (chunk-major inside 32-row tiles: a 128-byte K tile of
// 32 consecutive rows is one contiguous 4-KiB block, its 16-byte pieces in the MFMA operand order of the two-level
// scan - copied to LDS verbatim, one contiguous KiB per DMA instruction) and the first int8 term of the queries, K tiles of 128
// elements on v_mfma_i32_32x32x32_i8 - twice the fp16 rate, exact integer dot products.  The selection key
//   key = ||x_i||^2 (L2) + kq_b * (s_i * dot)        (two float roundings, the same expression wherever it is formed)
// is within A1 e_i + C1 of the exact key (flat_internal.h, ShadowQ): the caller keeps a deeper candidate list and
// the certificate uses that bound.
// S16: the same tiles on v_mfma_f32_16x16x32_f16 / v_mfma_i32_16x16x64_i8 (a wave's 128 rows x 64 queries = 8 x 4
// tiles of 16 x 16 instead of 4 x 2 of 32 x 32).  Same LDS image, same 16-B fragment reads (a K-32 step of a 16-row
// tile is one ds_read_b128 per lane: piece 4 s + lane / 16 of row lane % 16), same MFMA cycles per flop; the chip
// holds a higher clock on this shape under matrix load (tools/micro/mfma_shape.hip: bare loops 2.06 against 1.70 GHz
// with all 256 CUs issuing; cdna_hip_programming.md section 5.4 rule 28).  The selection keys differ from the 32 x 32
// form in the last bits (another accumulation order); the results do not - the rerank scores in float64 and
// certifies (flat_internal.h).
template <int NKT /* d / 64 (I8: d / 128), even */, int MODE = 1, int ABL = 0, bool I8 = false, bool S16 = false>
__global__ __launch_bounds__(512, 2) void scan_mm_kernel(MmArgs a) {
    constexpr bool FIRST = MODE == 0;
    static_assert(NKT % 2 == 0 && NKT >= 4, "K tiles are consumed in even/odd pairs");
    constexpr int64_t RB = (int64_t)NKT * 128;  // bytes per row (fp16, or int8 with 128-element K tiles)
    constexpr int KA = I8 ? 4096 : 128;         // bytes from one K tile of a row to the next in global memory
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gate_closed(a.gate)) return;
    typedef const __attribute__((address_space(1))) char* gcptr;
    typedef __attribute__((address_space(3))) char* lptr;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;  // row half / 64-query strip of this wave
    const int r = lane & 31, h = lane >> 5;

    // blocks b, b+8, b+16, ... share an XCD: give each XCD a contiguous run of tiles so the
    // workgroups that run together on it work on the same rows (bijective for any grid size)
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = (int)blockIdx.x & 7;
    const int v = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + ((int)blockIdx.x >> 3);
    const int n_my = v < a.n_tiles ? (a.n_tiles - v + nwg - 1) / nwg : 0;
    if (n_my == 0) return;  // whole workgroup (grid <= tiles, so this does not happen)
#ifdef PRAG_MM_DIAG
    const unsigned long long dbg_c0 = clock64(), dbg_w0 = wall_clock64();
#endif

    // ---- fragment read offsets: A row = 128*wr + 32*mt + r, B row = 64*wc + 32*nt + r; the
    //      swizzle term (row>>1)&7 only depends on r, k-step ks flips byte bits 5..6 ---------
    // FM (I8 over fragment-major shadow chunks, flat_internal.h shadow_piece_off): the A buffer holds the 8 shadow
    // tiles of the 256 rows verbatim - 4 KiB each, k-step s = one KiB in lane order - so a DMA instruction copies one
    // contiguous KiB and a 32 x 32 fragment read is 16 lane + 1024 s: conflict-free without a swizzle
    constexpr bool FM = I8 && kShadowFragMajor;
    const int swz = (r >> 1) & 7;
    int a_off[4], b_off[4];
    if constexpr (!S16) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int piece = ((2 * ks + h) ^ swz) << 4;
            a_off[ks] = FM ? 16384 * wr + ((ks * 64 + lane) << 4) : (128 * wr + r) * 128 + piece;
            b_off[ks] = 65536 + (64 * wc + r) * 128 + piece;
        }
    } else {
        // 16 x 16 tiles: lane (c16, q4) reads row c16 of its tile, piece 4 s + q4 of K-32 step s; the swizzle term
        // ((row >> 1) & 7) = (c16 >> 1) & 7 for every tile (tiles start at multiples of 16 rows)
        const int c16 = lane & 15, q4 = lane >> 4, swz16 = (c16 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int piece = (((4 * (ks & 1) + q4) ^ swz16) << 4);
            a_off[ks] = FM ? 16384 * wr + shadow_piece_off(c16, 4 * (ks & 1) + q4)
                           : (128 * wr + c16) * 128 + piece;          // (only [0] and [1] are used)
            b_off[ks] = 65536 + (64 * wc + c16) * 128 + piece;
        }
    }

    // ---- LDS-DMA geometry: one wave instruction fills 8 rows x 128 B linearly, so lane l
    //      fetches global piece (l&7) ^ swizzle(row) of row 8*chunk + (l>>3) ------------------
    //      (the per-lane parts are rebuilt inside tile_ptr, once per tile, from an opaque copy of
    //      the lane id: kept live across the main loop they get spilled, and a reload waits
    //      behind s_waitcnt vmcnt(0))
    const gcptr rows_g = (gcptr)(I8 ? reinterpret_cast<const char*>(a.rows8) : reinterpret_cast<const char*>(a.rows));
    const gcptr q_g = (gcptr)(I8 ? reinterpret_cast<const char*>(a.q8) : reinterpret_cast<const char*>(a.q16));
    const lptr lds0 = (lptr)smem;
    const lptr dA = lds0 + w * 1024;
    const lptr dB = lds0 + 65536 + (64 * (w >> 2) + 8 * (w & 3)) * 128;

    struct TilePtr {
        gcptr pa, pb;
        int64_t row0;
        int q0;
    };
    auto tile_ptr = [&](int t) {
        t = t < a.n_tiles ? t : a.n_tiles - 1;  // past the end: re-stage the last tile (never read)
        const int rt = t / a.n_qb, qb = t - rt * a.n_qb;
        int le = lane;
        asm volatile("" : "+v"(le));
        const int l8 = le >> 3;
        const int colb = (((le & 7) ^ ((4 * (w & 1) + (le >> 4)) & 7)) << 4);
        const int a_thr = 8 * w + l8;                        // A rows 8w.. (+128: second instruction)
        const int b_thr = 64 * (w >> 2) + 8 * (w & 3) + l8;  // B strips (w>>2) and (w>>2)+2, first 32 rows
        TilePtr p;
        p.row0 = a.row0 + (int64_t)rt * 256;
        p.q0 = qb * 256;
        if constexpr (FM)   // wave w: k-step w & 3 of shadow tile w >> 2 of the 256 rows (SA0: +4 tiles; SA1: +2, +6)
            p.pa = rows_g + ((p.row0 >> 5) + (w >> 2)) * (32 * RB) + (w & 3) * 1024 + le * 16;
        else if constexpr (I8)   // (row0 is a multiple of 256: the row's place inside its 32-row tile is a_thr & 31)
            p.pa = rows_g + ((p.row0 + a_thr) >> 5) * (32 * RB) + (a_thr & 31) * 128 + colb;
        else
            p.pa = rows_g + (p.row0 + a_thr) * RB + colb;
        p.pb = q_g + (int64_t)(p.q0 + b_thr) * RB + colb;
        return p;
    };

#define MM_GLDS(gp_, lp_)                                                              \
    if constexpr (!(ABL & 2)) __builtin_amdgcn_global_load_lds((gp_), (lp_), 16, 0, 0)
    // the four 16-KiB staging steps of one K tile (kt_) into buffer buf_ (0 even, 1 odd)
#define MM_SA0(tp_, kt_, buf_)                                    \
    {                                                             \
        const gcptr g0_ = (tp_).pa + (kt_) * KA;                  \
        const gcptr g1_ = g0_ + 128 * RB;                         \
        const lptr l0_ = dA + (buf_) * 32768;                     \
        const lptr l1_ = l0_ + 16384;                             \
        MM_GLDS(g0_, l0_);                                        \
        MM_GLDS(g1_, l1_);                                        \
    }
#define MM_SA1(tp_, kt_, buf_)                                    \
    {                                                             \
        const gcptr g0_ = (tp_).pa + 64 * RB + (kt_) * KA;        \
        const gcptr g1_ = g0_ + 128 * RB;                         \
        const lptr l0_ = dA + (buf_) * 32768 + 8192;              \
        const lptr l1_ = l0_ + 16384;                             \
        MM_GLDS(g0_, l0_);                                        \
        MM_GLDS(g1_, l1_);                                        \
    }
#define MM_SB0(tp_, kt_, buf_)                                    \
    {                                                             \
        const gcptr g0_ = (tp_).pb + (kt_) * 128;                 \
        const gcptr g1_ = g0_ + 128 * RB;                         \
        const lptr l0_ = dB + (buf_) * 32768;                     \
        const lptr l1_ = l0_ + 16384;                             \
        MM_GLDS(g0_, l0_);                                        \
        MM_GLDS(g1_, l1_);                                        \
    }
#define MM_SB1(tp_, kt_, buf_)                                    \
    {                                                             \
        const gcptr g0_ = (tp_).pb + 32 * RB + (kt_) * 128;       \
        const gcptr g1_ = g0_ + 128 * RB;                         \
        const lptr l0_ = dB + (buf_) * 32768 + 4096;              \
        const lptr l1_ = l0_ + 16384;                             \
        MM_GLDS(g0_, l0_);                                        \
        MM_GLDS(g1_, l1_);                                        \
    }
#define MM_LDSR(off_) ((ABL & 4) ? hz : *reinterpret_cast<const half8*>(smem + (off_)))
    // the 8 A fragments of a 64-row quadrant (base_ = byte offset of its first row in the buffer) and the 4 B
    // fragments of a 32-query half: 32 x 32 shape = [row tile of 32][k-step of 16]; 16 x 16 shape = fragment
    // f = 2 * (tile of 16) + (k-step of 32), kept in the same registers
#define MM_READ_A(base_)                                                                              \
    _Pragma("unroll") for (int f = 0; f < 8; ++f) {                                                   \
        if constexpr (S16) A[f >> 2][f & 3] = MM_LDSR(a_off[f & 1] + (FM ? (f >> 2) * 4096 + ((f >> 1) & 1) * 256 : (f >> 1) * 2048) + (base_)); \
        else A[f >> 2][f & 3] = MM_LDSR(a_off[f & 3] + (f >> 2) * 4096 + (base_));                    \
    }
#define MM_READ_B(BF_, base_)                                                                         \
    _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                   \
        if constexpr (S16) BF_[f] = MM_LDSR(b_off[f & 1] + (f >> 1) * 2048 + (base_));                \
        else BF_[f] = MM_LDSR(b_off[f] + (base_));                                                    \
    }
    // everything staged four or more phases ago has landed; the barrier publishes it
#define MM_WAIT_BAR()                                     \
    {                                                     \
        if constexpr (ABL & 16) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  \
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                        \
        __builtin_amdgcn_s_barrier();                     \
        __builtin_amdgcn_sched_barrier(0);                \
    }
#define MM_MFMA8(MQ_, NQ_, AF_, BF_, first_)                                                          \
    if constexpr (S16) {                                                                              \
        /* quadrant = 4 row tiles x 2 query tiles of 16 x 16, two K-32 steps: 16 MFMAs of 16 cycles */ \
        __builtin_amdgcn_s_setprio(1);                                                                \
        _Pragma("unroll") for (int sk = 0; sk < 2; ++sk)                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
            f32x4& t_ = acc16[4 * (MQ_) + i][2 * (NQ_) + j];                                          \
            const int fa_ = 2 * i + sk, fb_ = 2 * j + sk;                                             \
            if constexpr (ABL & 1) {                                                                  \
                asm volatile("" ::"v"(AF_[fa_ >> 2][fa_ & 3]), "v"(BF_[fb_]));                        \
            } else if constexpr (I8) {                                                                \
                t_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(                 \
                    __builtin_bit_cast(i32x4, AF_[fa_ >> 2][fa_ & 3]), __builtin_bit_cast(i32x4, BF_[fb_]), \
                    __builtin_bit_cast(i32x4, ((first_) && sk == 0) ? zero4 : t_), 0, 0, 0));          \
            } else {                                                                                  \
                t_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(AF_[fa_ >> 2][fa_ & 3], BF_[fb_],         \
                                                            ((first_) && sk == 0) ? zero4 : t_, 0, 0, 0); \
            }                                                                                         \
        }                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();                                      \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } else {                                                                                          \
        f32x16& c0_ = acc[2 * (MQ_)][NQ_];                                                            \
        f32x16& c1_ = acc[2 * (MQ_) + 1][NQ_];                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                            \
            if constexpr (ABL & 1) {                                                                  \
                asm volatile("" ::"v"(AF_[0][ks]), "v"(AF_[1][ks]), "v"(BF_[ks]));                    \
            } else if constexpr (I8 || (ABL & 64)) {  /* (ABL 64: timing probe, fp16 bytes through the int8 MFMA) */ \
                typedef int i32x16_ __attribute__((ext_vector_type(16)));                              \
                c0_ = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(                \
                    __builtin_bit_cast(i32x4, AF_[0][ks]), __builtin_bit_cast(i32x4, BF_[ks]),         \
                    __builtin_bit_cast(i32x16_, ((first_) && ks == 0) ? zero16 : c0_), 0, 0, 0));      \
                c1_ = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(                \
                    __builtin_bit_cast(i32x4, AF_[1][ks]), __builtin_bit_cast(i32x4, BF_[ks]),         \
                    __builtin_bit_cast(i32x16_, ((first_) && ks == 0) ? zero16 : c1_), 0, 0, 0));      \
            } else {                                                                                  \
                c0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF_[0][ks], BF_[ks], ((first_) && ks == 0) ? zero16 : c0_, 0, 0, 0); \
                c1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF_[1][ks], BF_[ks], ((first_) && ks == 0) ? zero16 : c1_, 0, 0, 0); \
            }                                                                                         \
        }                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();                                      \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc[4][2];     // 32 x 32 tiles: [row tile][query tile]
    f32x4 acc16[8][4];    // 16 x 16 tiles (S16): [row tile][query tile]; only one of the two sets is ever used
    half8 A[2][4], Bx[4], By[4];
    half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};  // ablation 3 only: an opaque constant in place of the fragment reads
    if constexpr (ABL & 4) asm volatile("" : "+v"(hz));
    if constexpr (ABL & (1 | 64)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i >> 1][i & 1] = zero16;
    }
    static_assert(!S16 || ABL == 0, "the timing-only ablations exist for the 32 x 32 form");

    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem + kMmLdsCnt);
    for (int i = tid; i < a.Bpad; i += 512) s_cnt[i] = 0;  // published by the prologue's barrier

    TilePtr cur = tile_ptr(v), nxt = tile_ptr(v + nwg);

    // ---- prologue: K tile 0 complete in the even buffer, first half of K tile 1 in the odd one
    MM_SA0(cur, 0, 0)
    MM_SB0(cur, 0, 0)
    MM_SB1(cur, 0, 0)
    MM_SA1(cur, 0, 0)
    MM_SB0(cur, 1, 1)
    MM_SA0(cur, 1, 1)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    MM_READ_B(Bx, 0)
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind

    for (int ti = 0; ti < n_my; ++ti) {
#pragma unroll
        for (int it = 0; it < NKT / 2; ++it) {
            const int ke = 2 * it, ko = 2 * it + 1;             // K tiles in the even / odd buffer
            const bool wrap = ke + 2 >= NKT;                    // staging moves on to the next tile
            const int ke2 = wrap ? ke + 2 - NKT : ke + 2;
            const int ko2 = wrap ? ko + 2 - NKT : ko + 2;
            const TilePtr& tn = wrap ? nxt : cur;
            // Fragment reads are spread 8/4/8/4 over the phases (a 12-read phase does not fit the
            // other group's 8-MFMA window): the first query tile of the NEXT K tile is fetched in
            // phase 4 into the register set the second query tile no longer needs, so the two sets
            // Bx/By swap roles every K tile.
            // ===== even K tile: queries 0-31 in Bx (read one phase ago), 32-63 in By =====
            // phase 1: quadrant (rows 0-63, queries 0-31)
            MM_READ_A(0)
            MM_SB1(cur, ko, 1)
            MM_WAIT_BAR()
            MM_MFMA8(0, 0, A, Bx, it == 0)
            // phase 2: (rows 0-63, queries 32-63)
            MM_READ_B(By, 4096)
            MM_SA1(cur, ko, 1)
            if (it == NKT / 2 - 2) {
                // row norms and bounds of this tile, well ahead of the filter.  Not in phase 1: at
                // d = 256 this is the tile's first K-tile pair and the lagging wave group may still
                // be reading the previous tile's copy until it passes the barrier that ends phase 1.
                if (w < (I8 ? 4 : 2)) {
                    // (the lane offset is rebuilt here on purpose: hoisted out of the tile loop it costs
                    // a VGPR pair the loop does not have, and a spilled pointer reloads behind vmcnt(0))
                    int l16;
                    asm volatile("v_lshlrev_b32 %0, 4, %1" : "=v"(l16) : "v"(lane));
                    const gcptr src = w == 0   ? (gcptr) reinterpret_cast<const char*>(a.xnorm + cur.row0)
                                      : w == 1 ? (gcptr) reinterpret_cast<const char*>(a.tau + cur.q0)
                                      : w == 2 ? (gcptr) reinterpret_cast<const char*>(a.sscale + cur.row0)
                                               : (gcptr) reinterpret_cast<const char*>(a.kq + cur.q0);
                    const gcptr gx = src + l16;
                    const lptr lx = lds0 + (w == 0 ? kMmLdsXn : w == 1 ? kMmLdsTau : w == 2 ? kMmLdsSs : kMmLdsKq);
                    MM_GLDS(gx, lx);
                }
            }
            MM_WAIT_BAR()
            MM_MFMA8(0, 1, A, By, it == 0)
            // phase 3: (rows 64-127, queries 32-63)
            MM_READ_A(8192)
            MM_SB0(tn, ke2, 0)
            MM_WAIT_BAR()
            MM_MFMA8(1, 1, A, By, it == 0)
            // phase 4: (rows 64-127, queries 0-31); By <- queries 0-31 of the odd K tile
            MM_READ_B(By, 32768)
            MM_SA0(tn, ke2, 0)
            MM_WAIT_BAR()
            MM_MFMA8(1, 0, A, Bx, it == 0)
            // ===== odd K tile: queries 0-31 in By, 32-63 in Bx =====
            MM_READ_A(32768)
            MM_SB1(tn, ke2, 0)
            MM_WAIT_BAR()
            MM_MFMA8(0, 0, A, By, false)
            MM_READ_B(Bx, 32768 + 4096)
            MM_SA1(tn, ke2, 0)
            MM_WAIT_BAR()
            MM_MFMA8(0, 1, A, Bx, false)
            MM_READ_A(32768 + 8192)
            MM_SB0(tn, ko2, 1)
            MM_WAIT_BAR()
            MM_MFMA8(1, 1, A, Bx, false)
            // phase 8: Bx <- queries 0-31 of the next even K tile (next tile's first one at the end)
            MM_READ_B(Bx, 0)
            MM_SA0(tn, ko2, 1)
            MM_WAIT_BAR()
            MM_MFMA8(1, 0, A, By, false)
        }

        // ---- filter: 128 scores per lane against the bound of their query ----------------
        if constexpr (ABL & 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(acc[i >> 1][i & 1]));
        } else {
            // (everything per-lane below is rebuilt from an opaque copy of the lane id: hoisted out of
            // the tile loop these values would sit in VGPRs the main loop does not have, and a spilled
            // value reloads behind s_waitcnt vmcnt(0), i.e. drains the DMA pipeline once per tile)
            int le = lane;
            asm volatile("" : "+v"(le));
            // A lane's 128 scores as groups of 16 for ONE query: 32 x 32 tiles are such groups (row tile mt, query
            // tile nt: rows 32 mt + 8 g + 4 h + e, query 32 nt + r).  With 16 x 16 tiles a group is gathered from
            // four of them: NMT = 2 row blocks of 64, NNT = 4 query tiles of 16; rows 64 mt + 16 g + 4 q4 + e.
            constexpr int NMT = S16 ? 2 : 4, NNT = S16 ? 4 : 2, RT_ROWS = S16 ? 64 : 32, G_ROWS = S16 ? 16 : 8,
                          QT_COLS = S16 ? 16 : 32;
            const int r = S16 ? (le & 15) : (le & 31), h = S16 ? (le >> 4) : (le >> 5);
            const int64_t rbase = cur.row0 + 128 * wr + 4 * h;
            const int qbase = cur.q0 + 64 * wc + r;
            // per-query constants: the 32 x 32 form keeps its two queries' values across the row tiles; the 16 x 16
            // form (four queries per lane) reads them from LDS where they are used - twelve more live registers
            // spilled the int8 squared-L2 variant
            float tauf[NNT], thr[NNT], al[NNT];
            auto q_consts = [&](int nt) {
                tauf[nt] = unsortable_f32(*reinterpret_cast<const uint32_t*>(smem + kMmLdsTau + (64 * wc + QT_COLS * nt + r) * 4));
                thr[nt] = tauf[nt] * a.inv_alpha;  // key <= tau  <=>  score >= tau / alpha  (alpha < 0)
                // I8: key = xn + kq_b * (s_i * dot); the query's kq = alpha * (its int8 scale) stands where alpha does
                al[nt] = I8 ? *reinterpret_cast<const float*>(smem + kMmLdsKq + (64 * wc + QT_COLS * nt + r) * 4) : a.alpha;
            };
            if constexpr (!S16) {
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt) q_consts(nt);
            }
            // The other wave group waits at the next barrier while this one filters, so every
            // instruction here is exposed twice per tile: the common case is kept to ~12 (inner
            // product) / ~26 (L2) VALU operations per 16 scores.  v_max3/v_min3 by hand: fmaxf()
            // adds a canonicalising v_max per operand in IEEE mode.  The s_nops cover the MFMA ->
            // VALU read hazard the compiler cannot see through inline asm.
            asm volatile("s_nop 15\n\ts_nop 15");
#define MM_MAX3(d_, x_, y_, z_) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d_) : "v"(x_), "v"(y_), "v"(z_))
#define MM_MIN3(d_, x_, y_, z_) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d_) : "v"(x_), "v"(y_), "v"(z_))
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                f32x4 xn[4], ss[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if ((FIRST && a.use_norm) || MODE == 2)
                        xn[g] = *reinterpret_cast<const f32x4*>(smem + kMmLdsXn + (128 * wr + RT_ROWS * mt + G_ROWS * g + 4 * h) * 4);
                    else
                        xn[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (I8)
                        ss[g] = *reinterpret_cast<const f32x4*>(smem + kMmLdsSs + (128 * wr + RT_ROWS * mt + G_ROWS * g + 4 * h) * 4);
                }
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt) {
                    if constexpr (S16) q_consts(nt);
                    // 16 x 16 tiles: row tile g of the 64-row block mt, query tile nt -> c[4 g ..]
                    // I8: t = s_i * dot (the integer converts exactly: |dot| <= 128 * 127^2 * NKT < 2^24)
                    f32x16 c;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float raw;
                            if constexpr (S16) raw = acc16[4 * mt + g][nt][e];
                            else raw = acc[mt][nt][4 * g + e];
                            c[4 * g + e] = I8 ? ss[g][e] * (float)__builtin_bit_cast(int, raw) : raw;
                        }
                    const int q = qbase + QT_COLS * nt;
                    const int64_t rb = rbase + RT_ROWS * mt;
                    if constexpr (FIRST) {
                        // no bound yet: every row is a candidate, slot = row - row0 (no counters);
                        // the four rows of a register quad are consecutive slots -> 16-B stores
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int64_t row = rb + G_ROWS * g;
                            const int64_t o = (int64_t)q * a.cap_q + (row - a.row0);
                            f32x4 kv;
                            i32x4 iv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                kv[e] = fmaf(al[nt], c[4 * g + e], xn[g][e]);
                                iv[e] = (int)row + e;
                            }
                            if (row + 3 < a.row1) {
                                *reinterpret_cast<f32x4*>(a.ckey + o) = kv;
                                *reinterpret_cast<i32x4*>(a.cidx + o) = iv;
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (row + e < a.row1) {
                                        a.ckey[o + e] = kv[e];
                                        a.cidx[o + e] = iv[e];
                                    }
                            }
                        }
                    } else {
                        // does any of the 16 scores reach the bound?
                        bool any;
                        if constexpr (MODE == 1) {
                            float m;
                            MM_MAX3(m, c[0], c[1], c[2]);
#pragma unroll
                            for (int e = 3; e < 15; e += 2) MM_MAX3(m, m, c[e], c[e + 1]);
                            // (I8: the key itself is formed - one multiply - so that this test and the
                            //  survivor's key below are the same arithmetic; kq < 0: the largest t has the smallest key)
                            any = I8 ? al[nt] * fmaxf(m, c[15]) <= tauf[nt] : fmaxf(m, c[15]) >= thr[nt];
                        } else {
                            float kk[16];
#pragma unroll
                            for (int e = 0; e < 16; ++e) kk[e] = fmaf(al[nt], c[e], xn[e >> 2][e & 3]);
                            float m;
                            MM_MIN3(m, kk[0], kk[1], kk[2]);
#pragma unroll
                            for (int e = 3; e < 15; e += 2) MM_MIN3(m, m, kk[e], kk[e + 1]);
                            any = fminf(m, kk[15]) <= tauf[nt];
                        }
                        if (__builtin_amdgcn_ballot_w64(any) != 0) {  // rare once the bound is warm
                            // survivors go to this workgroup's own region of the candidate store: the
                            // slot comes from an LDS counter (no global round trip), the stores are
                            // fire-and-forget.  Only the register quads that hold one are expanded.
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                float key[4];
                                bool hit = false;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    key[e] = fmaf(al[nt], c[4 * g + e], xn[g][e]);
                                    hit |= key[e] <= tauf[nt];
                                }
                                if (__builtin_amdgcn_ballot_w64(hit) == 0) continue;
                                const int64_t row0q = rb + G_ROWS * g;
                                int np = 0;
#pragma unroll
                                for (int e = 0; e < 4; ++e) np += (key[e] <= tauf[nt] && row0q + e < a.row1) ? 1 : 0;
                                if (np > 0) {
                                    // (inline asm on purpose: in front of a compiler-visible LDS atomic hipcc
                                    // puts s_waitcnt vmcnt(0) - the LDS-DMA in flight might alias it - which
                                    // drains the DMA pipeline on every survivor; the counters are no DMA target)
                                    uint32_t slot;
                                    const uint32_t cnt_addr = (uint32_t)(uintptr_t)(lptr)(reinterpret_cast<char*>(s_cnt + q));
                                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                                                 : "=v"(slot)
                                                 : "v"(cnt_addr), "v"((uint32_t)np)
                                                 : "memory");
                                    const int64_t o = ((int64_t)blockIdx.x * a.Bpad + q) * a.cap_wg;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        if (key[e] <= tauf[nt] && row0q + e < a.row1) {
                                            if (slot < (uint32_t)a.cap_wg) {
                                                a.wkey[o + slot] = key[e];
                                                a.widx[o + slot] = (int)(row0q + e);
                                            }
                                            ++slot;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
#undef MM_MAX3
#undef MM_MIN3
        }
        cur = nxt;
        nxt = tile_ptr(v + (ti + 2) * nwg);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail DMAs must not outlive the workgroup's LDS
    if (wr == 0) __builtin_amdgcn_s_barrier();         // pair the second group's extra barrier
    __syncthreads();                                   // both groups are through their last filter
#ifdef PRAG_MM_DIAG
    if (a.dbg && blockIdx.x == 0 && tid == 0) {
        a.dbg[0] = clock64() - dbg_c0;
        a.dbg[1] = wall_clock64() - dbg_w0;
    }
#endif
    if constexpr (!FIRST)
        for (int i = tid; i < a.Bpad; i += 512) a.wcnt[(int64_t)i * a.wg_stride + blockIdx.x] = s_cnt[i];
#undef MM_GLDS
#undef MM_SA0
#undef MM_SA1
#undef MM_SB0
#undef MM_SB1
#undef MM_LDSR
#undef MM_READ_A
#undef MM_READ_B
#undef MM_WAIT_BAR
#undef MM_MFMA8
}

// ---------------------------------------------------------------------------
// after each segment: the query's KC best so far (ckey/cidx[q][0..cnt)) + the survivors every
// workgroup collected for it in this segment -> its KC best by (key, id), in place;
// bound <- KC-th best key; cand[q][0..KC) <- row ids.  One 256-thread block per query.
// ---------------------------------------------------------------------------
constexpr int kMmCompactCap = 4096;  // entries staged in LDS; a query with more is flagged as overflowed

// This segment's survivors of query q -> s_v[n ...): 256 workgroups per step.  The entries of a step are numbered
// by a block prefix sum over the workgroups' counts and dealt round-robin to the threads (8-step binary search in
// the prefix array for the owning workgroup): every thread issues the same number of independent loads.  (One
// thread per workgroup copying its own region serially - the first version - left 64 threads with ~32 dependent
// round trips each when the candidate lists are deep: 122 us per compaction of 1000 queries at 256 candidates.)
// Returns true when a region or the staging buffer overflowed; n is advanced.  All 256 threads call it.
__device__ __forceinline__ bool mm_gather_survivors(unsigned long long* s_v, int& n, int q, const uint32_t* __restrict__ wcnt,
                                                    const float* __restrict__ wkey, const int* __restrict__ widx,
                                                    int cap_wg, int n_wg, int wg_stride, int Bpad, int* s_tot, int* s_incl) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    bool over = false;
    for (int w0 = 0; w0 < n_wg; w0 += 256) {
        const int wg = w0 + tid;
        const uint32_t cw = wg < n_wg ? wcnt[(int64_t)q * wg_stride + wg] : 0u;
        over |= cw > (uint32_t)cap_wg;
        const int mine = cw < (uint32_t)cap_wg ? (int)cw : cap_wg;
        int incl = mine;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int up = __shfl_up(incl, sft, 64);
            if (lane >= sft) incl += up;
        }
        if (lane == 63) s_tot[w] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            before += j < w ? s_tot[j] : 0;
            total += s_tot[j];
        }
        s_incl[tid] = before + incl;            // inclusive prefix over the 256 workgroups of this step
        __syncthreads();
        for (int e = tid; e < total; e += 256) {
            int lo = 0, hi = 255;               // first t with s_incl[t] > e
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int mid = (lo + hi) >> 1;
                if (s_incl[mid] > e) hi = mid; else lo = mid + 1;
            }
            const int t = lo;
            const int j = e - (t > 0 ? s_incl[t - 1] : 0);
            const int64_t wo = ((int64_t)(w0 + t) * Bpad + q) * cap_wg;
            if (n + e < kMmCompactCap) s_v[n + e] = pack_key(wkey[wo + j], widx[wo + j]);
            else over = true;
        }
        n = n + total < kMmCompactCap ? n + total : kMmCompactCap;
        __syncthreads();
    }
    return over;
}


__global__ __launch_bounds__(256) void mm_compact_kernel(uint32_t* __restrict__ cnt, float* __restrict__ ckey,
                                                        int* __restrict__ cidx, int cap_q, int KC,
                                                        uint32_t* __restrict__ tau, int* __restrict__ cand,
                                                        uint32_t* __restrict__ ovf, uint32_t* __restrict__ ovf_any,
                                                        const uint32_t* __restrict__ wcnt,
                                                        const float* __restrict__ wkey, const int* __restrict__ widx,
                                                        int cap_wg, int n_wg, int wg_stride, int Bpad, Gate gate) {
    __shared__ unsigned long long s_v[kMmCompactCap];
    __shared__ unsigned long long s_m[2][4];
    if (gate_closed(gate)) return;
    __shared__ int s_tot[4];
    __shared__ int s_incl[256];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t c = cnt[q];
    int n = c < (uint32_t)cap_q ? (int)c : cap_q;
    bool over = c > (uint32_t)cap_q || n > kMmCompactCap;
    n = n < kMmCompactCap ? n : kMmCompactCap;
    const int64_t o = (int64_t)q * cap_q;
    for (int i = tid; i < n; i += 256) s_v[i] = pack_key(ckey[o + i], cidx[o + i]);
    // this segment's survivors: 256 workgroups per step, offsets by a block prefix sum
    over |= mm_gather_survivors(s_v, n, q, wcnt, wkey, widx, cap_wg, n_wg, wg_stride, Bpad, s_tot, s_incl);
    __syncthreads();
    unsigned long long prev = 0;
    for (int round = 0; round < KC; ++round) {
        // smallest packed value above the previous pick (row ids are unique, so values are)
        unsigned long long m = ~0ull;
        for (int i = tid; i < n; i += 256) {
            const unsigned long long x = s_v[i];
            if ((round == 0 || x > prev) && x < m) m = x;
        }
        m = group_min16_u64(m);                       // DPP inside the 16-lane rows,
#pragma unroll
        for (int sft = 16; sft < 64; sft <<= 1) {    // two shuffles across them
            const unsigned long long other = __shfl_xor(m, sft, 64);
            m = other < m ? other : m;
        }
        if (lane == 0) s_m[round & 1][w] = m;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long other = s_m[round & 1][j];
            m = other < m ? other : m;
        }
        if (tid == 0) {
            const bool ok = m != ~0ull;
            cand[(int64_t)q * KC + round] = ok ? (int)(uint32_t)m : -1;
            if (ok) {
                ckey[o + round] = unsortable_f32((uint32_t)(m >> 32));
                cidx[o + round] = (int)(uint32_t)m;
                if (round == KC - 1) tau[q] = (uint32_t)(m >> 32);
            }
        }
        prev = m;  // once exhausted (m == ~0) nothing is above it: the remaining rounds write -1
    }
    const int any_over = __syncthreads_or(over ? 1 : 0);
    if (tid == 0) {
        cnt[q] = n < KC ? n : KC;
        if (any_over) {
            ovf[q] = 1u;
            *ovf_any = 1u;
        }
    }
}

// The same step for deep lists (k > 26 -> KC up to 1024): the gathered entries are sorted in LDS
// (bitonic, 256 threads) instead of selected round by round.
__global__ __launch_bounds__(256) void mm_compact_sort_kernel(uint32_t* __restrict__ cnt, float* __restrict__ ckey,
                                                             int* __restrict__ cidx, int cap_q, int KC,
                                                             uint32_t* __restrict__ tau, int* __restrict__ cand,
                                                             uint32_t* __restrict__ ovf, uint32_t* __restrict__ ovf_any,
                                                             const uint32_t* __restrict__ wcnt,
                                                             const float* __restrict__ wkey,
                                                             const int* __restrict__ widx, int cap_wg, int n_wg,
                                                             int wg_stride, int Bpad, Gate gate) {
    __shared__ unsigned long long s_v[kMmCompactCap];
    __shared__ int s_tot[4];
    if (gate_closed(gate)) return;
    __shared__ int s_incl[256];
    const int q = blockIdx.x, tid = threadIdx.x;
    const uint32_t c = cnt[q];
    int n = c < (uint32_t)cap_q ? (int)c : cap_q;
    bool over = c > (uint32_t)cap_q || n > kMmCompactCap;
    n = n < kMmCompactCap ? n : kMmCompactCap;
    const int64_t o = (int64_t)q * cap_q;
    for (int i = tid; i < n; i += 256) s_v[i] = pack_key(ckey[o + i], cidx[o + i]);
    over |= mm_gather_survivors(s_v, n, q, wcnt, wkey, widx, cap_wg, n_wg, wg_stride, Bpad, s_tot, s_incl);
    int n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    for (int i = n + tid; i < n_pad; i += 256) s_v[i] = ~0ull;
    bitonic_sort_u64<256>(s_v, n_pad);
    for (int r = tid; r < KC; r += 256) {
        const bool ok = r < n;
        const unsigned long long m = ok ? s_v[r] : ~0ull;
        cand[(int64_t)q * KC + r] = ok ? (int)(uint32_t)m : -1;
        if (ok) {
            ckey[o + r] = unsortable_f32((uint32_t)(m >> 32));
            cidx[o + r] = (int)(uint32_t)m;
            if (r == KC - 1) tau[q] = (uint32_t)(m >> 32);
        }
    }
    const int any_over = __syncthreads_or(over ? 1 : 0);
    if (tid == 0) {
        cnt[q] = n < KC ? n : KC;
        if (any_over) {
            ovf[q] = 1u;
            *ovf_any = 1u;
        }
    }
}

bool mm_supported(int d, int store_dtype, int kc) {
    return store_dtype == PRAG_F16 && (d == 256 || d == 512 || d == 768 || d == 1024) && kc >= 8 && kc <= kMmMaxKc;
}

bool mm8_supported(int d, int kc) {
    return (d == 512 || d == 768 || d == 1024) && kc >= 8 && kc <= kMmMaxKc;   // d / 128 K tiles, an even number >= 4
}

template <int NKT, int MODE, int ABL = 0, bool I8 = false, bool S16 = false>
static int launch_mm_impl(const MmArgs& a, int grid, hipStream_t st) {
    auto kern = scan_mm_kernel<NKT, MODE, ABL, I8, S16>;
    static LdsOptIn lds_opt_in;
    {
        const int rc_ = lds_opt_in.ensure(reinterpret_cast<const void*>(kern), 160 * 1024);
        if (rc_ != PRAG_OK) return rc_;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), kMmLdsCnt + 4 * a.Bpad, st, a);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

template <int NKT, int ABL = 0>
static int launch_mm(const MmArgs& a, bool first, int grid, hipStream_t st, bool s16 = false) {
    if constexpr (ABL == 0) {
        if (s16) {
            if (first) return launch_mm_impl<NKT, 0, 0, false, true>(a, grid, st);
            if (a.use_norm) return launch_mm_impl<NKT, 2, 0, false, true>(a, grid, st);
            return launch_mm_impl<NKT, 1, 0, false, true>(a, grid, st);
        }
    }
    if (first) return launch_mm_impl<NKT, 0, 0>(a, grid, st);
    if (a.use_norm) return launch_mm_impl<NKT, 2, 0>(a, grid, st);
    return launch_mm_impl<NKT, 1, ABL>(a, grid, st);
}

template <int NKT8>
static int launch_mm8(const MmArgs& a, bool first, int grid, hipStream_t st, bool s16 = false) {
    // (squared L2 on the int8 tiles stays on the 32 x 32 form: its filter needs row norms AND row scales, and the
    // 16 x 16 variant of it spilled five registers)
    if (s16 && !a.use_norm) {
        if (first) return launch_mm_impl<NKT8, 0, 0, true, true>(a, grid, st);
        return launch_mm_impl<NKT8, 1, 0, true, true>(a, grid, st);
    }
    if (first) return launch_mm_impl<NKT8, 0, 0, true>(a, grid, st);
    if (a.use_norm) return launch_mm_impl<NKT8, 2, 0, true>(a, grid, st);
    return launch_mm_impl<NKT8, 1, 0, true>(a, grid, st);
}

int mm_run(const MmSearch& s, hipStream_t st, EventRing& prof) {
    PRAG_REQUIRE((s.i8 ? mm8_supported(s.d, s.kc) && s.rows8 && s.sscale && s.q8 && s.kq
                       : mm_supported(s.d, PRAG_F16, s.kc)) && s.Bpad % 256 == 0 && s.Bpad <= kMmMaxQueries &&
                     s.cap_q >= kMmFirstSeg && s.max_wg >= 1 && s.max_wg <= s.wg_slots,
                 PRAG_EUNSUPPORTED, "internal: MFMA-tiled scan called outside its envelope");
    const int64_t first_rows = std::min<int64_t>(s.N, kMmFirstSeg);
    // cnt[q] = first_rows, ovf[q] = 0, *ovf_any = 0 were set by the caller (prep_queries_kernel)
    MmArgs a;
    a.rows = s.rows;
    a.xnorm = s.xnorm;
    a.q16 = s.q16;
    a.n_qb = s.Bpad / 256;
    a.alpha = s.alpha;
    a.inv_alpha = 1.0f / s.alpha;
    a.use_norm = s.use_norm;
    a.tau = s.tau;
    a.ckey = s.ckey;
    a.cidx = s.cidx;
    a.cap_q = s.cap_q;
    a.Bpad = s.Bpad;
    a.wcnt = s.wcnt;
    a.wg_stride = s.wg_slots;
    a.wkey = s.wkey;
    a.widx = s.widx;
    a.cap_wg = s.cap_wg;
    a.dbg = nullptr;
    a.rows8 = s.rows8;
    a.sscale = s.sscale;
    a.q8 = s.q8;
    a.kq = s.kq;
    a.gate = s.gate;
#ifdef PRAG_MM_DIAG
    static unsigned long long* dbg_dev = nullptr;
    if (getenv("PRAG_MM_CLOCK")) {
        if (!dbg_dev) PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&dbg_dev), 16));
        a.dbg = dbg_dev;
    }
#endif
    int64_t lo = 0;
    int64_t hi = first_rows;  // first segment: every row is a candidate
    int seg_i = 0;
    while (lo < s.N) {
        a.row0 = lo;
        a.row1 = hi;
        const bool first = lo == 0;
        const int64_t n_rt = (hi - lo + 255) / 256;
        a.n_tiles = (int)(n_rt * a.n_qb);
        const int grid = (int)std::min<int64_t>(a.n_tiles, std::max(1, s.max_wg));
        const bool biggest = hi == s.N;  // segments grow x16: the last one dominates
        if (biggest) prof.begin(st);
        int rc;
        if (s.i8) {
            switch (s.d) {
                case 512: rc = launch_mm8<4>(a, first, grid, st, s.shape16); break;
                case 768:
#ifdef PRAG_MM_DIAG
                {   // timing-only ablations of the int8 tiles (wrong results): 8 no filter, 1 no MFMAs, 9 neither
                    const char* e = getenv("PRAG_MM_ABLATE");
                    const int abl = e ? atoi(e) : 0;
                    if (!first && !a.use_norm && abl == 8) { rc = launch_mm_impl<6, 1, 8, true>(a, grid, st); break; }
                    if (!first && !a.use_norm && abl == 1) { rc = launch_mm_impl<6, 1, 1, true>(a, grid, st); break; }
                    if (!first && !a.use_norm && abl == 9) { rc = launch_mm_impl<6, 1, 9, true>(a, grid, st); break; }
                    if (!first && !a.use_norm && abl == 2) { rc = launch_mm_impl<6, 1, 2, true>(a, grid, st); break; }
                    if (!first && !a.use_norm && abl == 4) { rc = launch_mm_impl<6, 1, 4, true>(a, grid, st); break; }
                }
#endif
                    rc = launch_mm8<6>(a, first, grid, st, s.shape16); break;
                default: rc = launch_mm8<8>(a, first, grid, st, s.shape16); break;
            }
        } else
        switch (s.d) {
            case 256: rc = launch_mm<4>(a, first, grid, st, s.shape16); break;
            case 512: rc = launch_mm<8>(a, first, grid, st, s.shape16); break;
            case 768:
#ifdef PRAG_MM_DIAG
            {
                const char* e = getenv("PRAG_MM_ABLATE");
                switch (e ? atoi(e) : 0) {
                    case 8: rc = launch_mm<12, 8>(a, first, grid, st); break;
                    case 9: rc = launch_mm<12, 9>(a, first, grid, st); break;
                    case 10: rc = launch_mm<12, 10>(a, first, grid, st); break;
                    case 12: rc = launch_mm<12, 12>(a, first, grid, st); break;
                    case 14: rc = launch_mm<12, 14>(a, first, grid, st); break;
                    case 15: rc = launch_mm<12, 15>(a, first, grid, st); break;
                    case 24: rc = launch_mm<12, 24>(a, first, grid, st); break;
                    case 40: rc = launch_mm<12, 40>(a, first, grid, st); break;
                    case 56: rc = launch_mm<12, 56>(a, first, grid, st); break;
                    case 72: rc = launch_mm<6, 72>(a, first, grid, st); break;   // 768 int8 per row
                    default: rc = launch_mm<12>(a, first, grid, st, s.shape16); break;
                }
                break;
            }
#else
                rc = launch_mm<12>(a, first, grid, st, s.shape16);
                break;
#endif
            default: rc = launch_mm<16>(a, first, grid, st, s.shape16); break;
        }
        if (biggest) prof.end(st);
        if (rc != PRAG_OK) return rc;
#ifdef PRAG_MM_DIAG
        if (a.dbg && biggest) {
            unsigned long long h[2];
            PRAG_HIP(hipMemcpy(h, a.dbg, 16, hipMemcpyDeviceToHost));
            fprintf(stderr, "[mm diag] block 0: %llu core ticks in %.1f us -> %.0f MHz\n", h[0], h[1] / 100.0,
                    h[1] ? h[0] / (h[1] / 100.0) : 0.0);
        }
#endif
        if (s.kc <= 32)
            hipLaunchKernelGGL(mm_compact_kernel, dim3(s.B), dim3(256), 0, st, s.cnt, s.ckey, s.cidx, s.cap_q, s.kc, s.tau,
                               s.cand, s.ovf, s.ovf_any, s.wcnt, s.wkey, s.widx, s.cap_wg, first ? 0 : grid, s.wg_slots,
                               s.Bpad, s.gate);
        else
            hipLaunchKernelGGL(mm_compact_sort_kernel, dim3(s.B), dim3(256), 0, st, s.cnt, s.ckey, s.cidx, s.cap_q, s.kc,
                               s.tau, s.cand, s.ovf, s.ovf_any, s.wcnt, s.wkey, s.widx, s.cap_wg, first ? 0 : grid,
                               s.wg_slots, s.Bpad, s.gate);
        PRAG_LAUNCH_CHECK();
        lo = hi;
        const int g_step = mm_growth_step(s.growth, seg_i, s.i8 != 0);
        ++seg_i;
        hi = std::min<int64_t>(s.N, hi * (int64_t)std::max(2, std::min(16, g_step)));
    }
    return PRAG_OK;
}

}  // namespace prag
