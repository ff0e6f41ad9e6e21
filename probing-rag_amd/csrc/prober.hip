// Fused prober ensemble + gate for MI355X (gfx950, CDNA4).
//
// What it computes (reference: /root/reference, utils.py:29-57 ImprovedProbe,
// exp_rag.py:381-389, 406-415):
//   LN(d) -> fc1(d->512) -> SiLU -> LN(512) -> fc2(512->512) -> SiLU -> LN(512)
//   -> fc3(512->2), one weight set per probed layer, then per-layer softmax,
//   sum over layers and the threshold compare.
//
// How (DESIGN.md "Kernel 1"):
//   * One workgroup = NWV waves (8 = two per SIMD for 64/128-row tiles, 4 for 32-row tiles) =
//     one tile of 32*CT batch rows of one layer; XCD-aware block -> (layer, tile) placement.
//     The GEMMs are computed TRANSPOSED, H^T[n, m] = W[n, :] . x[m, :] on
//     v_mfma_f32_32x32x16_f16: the weight rows sit on the MFMA A operand, the batch rows on
//     the B operand, so every wave owns 512/NWV hidden units for all of the tile's rows.
//   * Weights are pre-packed at load time in MFMA-fragment-major order (one fully coalesced
//     1 KiB buffer_load_dwordx4 per fragment per wave, straight to VGPRs - a wave's weight rows
//     are not shared with the other waves, so an LDS round trip would be pure overhead).
//     LayerNorm affines are folded into the following Linear at load time.
//   * Activations (shared by all waves) are staged through LDS in full 128-B lines with an
//     XOR swizzle that makes the ds_read_b128 fragment reads conflict-free; double-buffered,
//     one raw s_barrier per 64-wide K step, fragments of the next sub-step prefetched.
//   * No LayerNorm is applied element-wise.  fp16 activations are fed to the MFMA *raw*
//     (exact) and LN0 becomes  rstd*(W~x - mu*rowsum(W~)) + b~  in the epilogue, with
//     sum x / sum x^2 from v_dot2_f32_f16 on the staging registers; LN1 and LN2 are folded the
//     same way into the fc2 / fc3 epilogues from one-pass sums of the SiLU outputs.
//   * The fc1 accumulator tile has the hidden index in its registers and the batch row on its
//     lane - exactly the B-operand layout of the next MFMA (k order permuted; W2 is
//     pre-permuted to match), so SiLU runs lane-locally and fc2's operand crosses waves
//     through LDS as ready-made fragments: an fp16 hi term plus a lo term (fp16 with fp32-parity
//     weights; fp8 against an fp8 copy of W2 with fp16 weights - half the matrix-pipe time).
//   * 128-row tiles run fc2 in two passes; the two waves of a SIMD take the VALU phase (SiLU
//     epilogues) and the MFMA phase of a pass in opposite orders, so the pipe stays fed.
//   * fc3 (512->2) partial dot products, the cross-wave reductions and the logits store finish
//     the launch; a second tiny kernel does softmax / sum over layers / threshold in the
//     reference's own order (deterministic, no float atomics).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "prag_common.h"
#include "prober_internal.h"
#include "tail_gate.h"
#include "prober_small.h"

namespace prag {

#ifdef PRAG_MM_DIAG
// timing-only build (make diag): s_memtime stamps of three workgroups, read by tools/prober_stamps.py
// second half of the buffer: s_memrealtime (100 MHz) at the same points - the clock the chip holds inside the
// kernel is d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md "DVFS give-back" item 6)
__device__ unsigned long long g_pstamp[2 * 3 * 8 * 32];
#define PSTAMP_T(var, var2)                                                                   \
    {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(var), "=s"(var2)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    }
#define PSTAMP(i)                                                                             \
    if (ps_sel >= 0) {                                                                        \
        unsigned long long t_, r_;                                                            \
        PSTAMP_T(t_, r_)                                                                      \
        if (lane == 0) {                                                                      \
            g_pstamp[(ps_sel * 8 + w) * 32 + (i)] = t_;                                       \
            g_pstamp[3 * 8 * 32 + (ps_sel * 8 + w) * 32 + (i)] = r_;                          \
        }                                                                                     \
    }
#else
#define PSTAMP(i)
#endif

// column tiles per fc2 pass: two wherever the tile has two (every W2 fragment then feeds 4 MFMAs and the
// 512 KiB of W2 stream through the CU half as often); the 4-wave 128-row variant keeps one (its fc2 phase
// would hold 128 + 64 accumulator VGPRs)
template <int CT, int NWV>
constexpr int fc2_group() {
    return (CT >= 2 && !(CT == 4 && NWV == 4)) ? 2 : 1;
}

// fc2's second activation term (the low 11 bits of the SiLU outputs) in fp8: v_mfma_f32_32x32x64_f8f6f4 does
// the 512-wide contraction in a quarter of the k-steps at half the rate (64 cycles each), i.e. in half the
// matrix-pipe time of the fp16 term; |lo| <= 2^-10 |s|, so 3 mantissa bits on it and on its copy of W2 leave
// ~2^-15 of relative error per product (the hi term is rounded to nearest here, so |lo| <= 2^-12 |s|).
// fp16-weight mode only; every tile shape uses it, so a row's logits do not depend on the batch it is in.
// Two row tiles of a wave make exactly one 32-byte operand per lane: k block kb = hidden units 64kb..64kb+63.
template <int NA, int NWV>
constexpr bool fc2_lo8() {
    return NA == 1;
}
template <int NA, int NWV, int G>
constexpr int exch_bytes() {
    return fc2_lo8<NA, NWV>() ? (16 * 2 * G + 8 * G * 2) * 1024     // hi fragments + fp8 lo operands
                              : 2 * 16 * 2 * G * 1024;            // hi/lo x 16 row tiles x 2 k-steps x G tiles
}

template <int NA, int NB, int CT, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV / 4) void prober_fused_kernel(ProberArgs a) {
    constexpr int NT = 64 * NWV;        // threads
    constexpr int RT = 16 / NWV;        // 32-row hidden tiles per wave
    constexpr int G = fc2_group<CT, NWV>();
    constexpr int NG = CT / G;
    constexpr bool RAW = (NB == 1);
    constexpr int ROWS = 32 * CT;
    constexpr int XPART = ROWS * 128;            // bytes of one staged part (64 halves per row)
    constexpr int XSTAGE = NB * XPART;
    constexpr bool LO8 = fc2_lo8<NA, NWV>();
    static_assert(!LO8 || RT % 2 == 0, "two row tiles per fp8 operand");
    constexpr int EXCH = exch_bytes<NA, NWV, G>();
    constexpr int REGION_A = (kRing * XSTAGE > EXCH) ? kRing * XSTAGE : EXCH;   // activation ring | exchange area
    constexpr int NPASS = (256 * CT + NT - 1) / NT;  // 16-B pieces per thread per staged part

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_x = smem;                                          // staging ring (fc1)
    char* s_ex = smem;                                         // exchange fragments (fc2), same bytes
    float* s_red = reinterpret_cast<float*>(smem + REGION_A);  // [2][NWV][ROWS] LN1 sums, then [4][NWV][64]
    float* s_mu0 = s_red + 2 * NWV * ROWS + 4 * NWV * 64;      // [ROWS]
    float* s_rs0 = s_mu0 + ROWS;                               // [ROWS]
    float* s_cst = s_rs0 + ROWS;  // [6][512]: wsum1, b1, b2, W3[0], W3[1], w2sum (epilogue constants)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int hh = lane >> 5;
    // XCD-aware placement (speed only): blocks b and b+8 share an XCD, so give every XCD a
    // contiguous chunk of the layer-major work list - its 4 MiB L2 then holds one or two
    // layers' weights instead of all of them.  Blocks past the list exit.
    const int per_xcd = (a.n_tiles * a.n_run + 7) / 8;
    const int vidx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (vidx >= a.n_tiles * a.n_run) return;
    const int lrun = vidx / a.n_tiles;
    const LayerDev& L = a.layers[a.layer0 + lrun];
    const int m0 = (vidx - lrun * a.n_tiles) * ROWS;
    // the layer record, read once before anything is stored (scalar loads) and pinned to SGPRs
    const __amdgpu_buffer_rsrc_t rs_w1 = make_rsrc(uniform_p(L.W1f), (size_t)NA * a.d * 1024);
    const __amdgpu_buffer_rsrc_t rs_w2 = make_rsrc(uniform_p(L.W2f), (size_t)NA * 32 * 16 * 1024);
    const __amdgpu_buffer_rsrc_t rs_w2q = make_rsrc(uniform_p(L.W2q), LO8 ? (size_t)8 * 16 * 2048 : 0);
    const float L_sc1 = uniform_f(L.sc1), L_sc2 = uniform_f(L.sc2);
    const float L_b3[2] = {uniform_f(L.b3[0]), uniform_f(L.b3[1])};
    const float L_w3sum[2] = {uniform_f(L.w3sum[0]), uniform_f(L.w3sum[1])};
    const int d = a.d;
    const int S16 = d >> 4;
    const int T = d >> 6;
#ifdef PRAG_MM_DIAG
    const int ps_sel = !a.stamps ? -1 : vidx == 0 ? 0 : vidx == 17 ? 1 : vidx == a.n_tiles * a.n_run - 1 ? 2 : -1;
    PSTAMP(0)
#endif

    // epilogue constants -> LDS once, so no epilogue ever waits on a global load.  The loads are issued
    // right behind the first activation tile's and stored after the weight prologue (their latency used to
    // sit in front of the first tile's: two dependent memory round trips before the first MFMA).
    constexpr int NCST = (6 * kHidden / 4 + NT - 1) / NT;
    f32x4 cst_v[NCST];
    auto cst_load = [&]() {
#pragma unroll
        for (int u = 0; u < NCST; ++u) {
            const int i = tid + u * NT;
            const int ic = i < 6 * kHidden / 4 ? i : 6 * kHidden / 4 - 1;
            const int arr = ic / (kHidden / 4), o = (ic % (kHidden / 4)) * 4;
            const gptr_f32 src = as_global(arr == 0 ? L.wsum1 : arr == 1 ? L.b1 : arr == 2 ? L.b2
                                           : arr == 5 ? L.w2sum : (L.W3 + (arr - 3) * kHidden));
            typedef const __attribute__((address_space(1))) f32x4* gptr_f32x4;
            cst_v[u] = *(gptr_f32x4)(src + o);
        }
    };
    auto cst_store = [&]() {
#pragma unroll
        for (int u = 0; u < NCST; ++u) {
            const int i = tid + u * NT;
            if (i < 6 * kHidden / 4) {
                const int arr = i / (kHidden / 4), o = (i % (kHidden / 4)) * 4;
                *reinterpret_cast<f32x4*>(s_cst + arr * kHidden + o) = cst_v[u];
            }
        }
    };

    // ---- activation staging: thread -> 16-B piece(s) of the [ROWS x 64] tile ----
    // (threads beyond the tile duplicate an in-range piece: no divergent loads)
    static_assert((256 * CT) % NT == 0, "every 16-B piece of a staged tile has exactly one owner");
    __amdgpu_buffer_rsrc_t rs_x[NB];   // this tile's rows of this layer
    {
        const int64_t tile0 = (int64_t)lrun * a.x_layer_stride + (int64_t)m0 * d;
        const size_t bytes = (size_t)(a.B - m0) * d * 2;
        rs_x[0] = make_rsrc(uniform_p(a.xh + tile0), bytes);
        if constexpr (NB == 2) rs_x[1] = make_rsrc(uniform_p(a.xl + tile0), bytes);
    }
    unsigned x_off[NPASS];
    int st_off[NPASS];
#pragma unroll
    for (int c = 0; c < NPASS; ++c) {
        const int e = tid + c * NT;
        const int row = e >> 3, q = e & 7;
        const int lrow = m0 + row < a.B ? row : a.B - 1 - m0;   // rows past the batch re-read its last row
        x_off[c] = (unsigned)(lrow * d + 8 * q) * 2u;
        st_off[c] = row * 128 + ((q ^ ((row >> 1) & 7)) << 4);
    }
    // fragment read offsets: lane (r,hh), tile c, sub-step -> piece 2*sub+hh of row 32c+r
    int rd_row_off[CT], rd_sw[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int row = 32 * c + r;
        rd_row_off[c] = row * 128;
        rd_sw[c] = (row >> 1) & 7;
    }

    u32x4 xreg[NB][NPASS];
    auto x_load_r = [&](u32x4 (&xr)[NB][NPASS], int t) {
#pragma unroll
        for (int p = 0; p < NB; ++p)
#pragma unroll
            for (int c = 0; c < NPASS; ++c) xr[p][c] = buf_load16(rs_x[p], x_off[c], (unsigned)t * 128u);
    };
    auto x_store_r = [&](u32x4 (&xr)[NB][NPASS], int stage) {
#pragma unroll
        for (int p = 0; p < NB; ++p)
#pragma unroll
            for (int c = 0; c < NPASS; ++c)
                *reinterpret_cast<u32x4*>(s_x + stage * XSTAGE + p * XPART + st_off[c]) = xr[p][c];
    };
    auto x_load = [&](int t) { x_load_r(xreg, t); };
    auto x_store = [&](int stage) { x_store_r(xreg, stage); };

    // ---- weight fragments: straight from global, 1 KiB per wave-load --------------
    const unsigned lane16 = lane * 16;
    const unsigned w_frag0 = RT * w;   // fragment ((part*S16 + s16)*16 + RT*w + rti), 1 KiB each
    half8 afr[4][NA][RT];
    auto a_load = [&](int slot, int s16) {
#pragma unroll
        for (int p = 0; p < NA; ++p)
#pragma unroll
            for (int rti = 0; rti < RT; ++rti) {
                const u32x4 v = buf_load16(rs_w1, lane16, ((unsigned)(p * S16 + s16) * 16u + w_frag0 + rti) << 10);
                afr[slot][p][rti] = __builtin_bit_cast(half8, v);
            }
    };

    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][c][e] = 0.f;

    // in-flight LayerNorm-0 statistics (RAW only): every thread sums the 16-B pieces it stages, straight
    // from the staging registers (each element of the tile is touched once per workgroup; the 8 lanes that
    // share a row are combined after the loop)
    float st_s[NPASS], st_q2[NPASS];
#pragma unroll
    for (int c = 0; c < NPASS; ++c) st_s[c] = st_q2[c] = 0.f;
    const half2_t kOnes2 = {(_Float16)1.f, (_Float16)1.f};
    auto x_stats_r = [&](u32x4 (&xr)[NB][NPASS]) {
        // LayerNorm-0 sums with v_dot2_f32_f16: products of halves are exact in f32,
        // accumulation is f32 (error ~1e-6 * (1 + mean^2/var) on the variance)
#pragma unroll
        for (int c = 0; c < NPASS; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned int u = xr[0][c][j];   // (bit_cast of the element lvalue itself reads element 0)
                const half2_t xv = __builtin_bit_cast(half2_t, u);
                st_s[c] = __builtin_amdgcn_fdot2(xv, kOnes2, st_s[c], false);
                st_q2[c] = __builtin_amdgcn_fdot2(xv, xv, st_q2[c], false);
            }
    };
    auto x_stats = [&]() { x_stats_r(xreg); };

    // ---- prologue ---------------------------------------------------------------
    // (issue order mirrors the loop body - activations first, then the four weight
    // slots - so the counted vmcnt waits at the loop head hold on entry too)
    // Activation tiles live in a kRing-stage ring and are stored kAhead K steps before they are read, so one
    // barrier per kAhead K steps orders everything: tile t (stage t % kRing) is written during K step
    // t - kAhead, the barrier in front of every kAhead-th K step publishes the tiles of the coming group, and
    // the stage a K step overwrites was last read kAhead K steps - at least one barrier - earlier.
    {
        // every load of the prologue is issued before the first one is waited for: the first kAhead tiles,
        // the epilogue constants, tile kAhead (the loop's staging registers) and the four weight slots
        u32x4 xpro[kAhead][NB][NPASS];
#pragma unroll
        for (int i = 0; i < kAhead; ++i) x_load_r(xpro[i], i < T ? i : T - 1);
        __builtin_amdgcn_sched_barrier(0);
        cst_load();
        __builtin_amdgcn_sched_barrier(0);
        x_load(kAhead < T ? kAhead : T - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            __builtin_amdgcn_sched_barrier(0);  // pin the issue order (see above)
            a_load(s, s);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < kAhead; ++i) {
            if constexpr (RAW)
                if (i < T) x_stats_r(xpro[i]);
            x_store_r(xpro[i], i);
        }
        cst_store();
        __builtin_amdgcn_sched_barrier(0);
    }
    PSTAMP(1)

    // ---- fc1 main loop: one barrier per kAhead 64-wide K steps ---------------------
    for (int t = 0; t < T; ++t) {
        if ((t & (kAhead - 1)) == 0) {
            // LDS-only hand-off: drain this wave's LDS ops and meet at a raw barrier.
            // (__syncthreads() also carries a fence that makes hipcc drain vmcnt to 0,
            // which would serialise the weight prefetch.)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (RAW)
            if (t + kAhead < T) x_stats();                 // tile t+kAhead; the tail's re-reads do not count
        x_store((t + kAhead) & (kRing - 1));           // tile t+kAhead (loaded one step ago)
        x_load(t + kAhead + 1 < T ? t + kAhead + 1 : T - 1);   // clamped: the tail re-reads the last tile
        const char* xs = s_x + (t & (kRing - 1)) * XSTAGE;
        const int s16n = 4 * (t + 1 < T ? t + 1 : T - 1);
        // B fragments of sub-step s+1 are read while the MFMAs of sub-step s run: only the first read
        // of a K step is exposed
        half8 bfr[2][NB][CT];
        auto b_read = [&](int buf, int sub) {
#pragma unroll
            for (int p = 0; p < NB; ++p)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const int off = p * XPART + rd_row_off[c] + (((2 * sub + hh) ^ rd_sw[c]) << 4);
                    bfr[buf][p][c] = *reinterpret_cast<const half8*>(xs + off);
                }
        };
        b_read(0, 0);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const int cb = sub & 1;
            if (sub < 3) b_read(cb ^ 1, sub + 1);
#pragma unroll
            for (int rti = 0; rti < RT; ++rti)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    acc[rti][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[sub][0][rti], bfr[cb][0][c],
                                                                         acc[rti][c], 0, 0, 0);
                    if constexpr (NB == 2)
                        acc[rti][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            afr[sub][0][rti], bfr[cb][1][c], acc[rti][c], 0, 0, 0);
                    if constexpr (NA == 2)
                        acc[rti][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            afr[sub][1][rti], bfr[cb][0][c], acc[rti][c], 0, 0, 0);
                }
            a_load(sub, s16n + sub);  // refill this slot for the next K step
            if constexpr (NA == 1 && NB == 1) {
                // one fragment read / one weight load in the shadow of each MFMA instead of a block of them
                // after the MFMAs (tools/micro/fc1_loop.hip: 2781 -> 2653 cycles per K step)
                constexpr int n_mf = RT * CT, n_ds = CT < n_mf ? CT : n_mf;
                if (sub < 3) {
#pragma unroll
                    for (int i = 0; i < n_ds; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                    }
                }
                constexpr int n_vm = (n_mf - n_ds) < RT ? (n_mf - n_ds) : RT;
#pragma unroll
                for (int i = 0; i < n_vm; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // VMEM read
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep sub-steps apart: caps live fragments (no spills)
        }
    }

    PSTAMP(2)
    // ---- LayerNorm-0 statistics -> LDS -------------------------------------------
    if constexpr (RAW) {
#pragma unroll
        for (int c = 0; c < NPASS; ++c) {
            // lanes 8j..8j+7 staged the eight pieces of one row
            float s1 = st_s[c], s2 = st_q2[c];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            const float mean = s1 / (float)d;
            const float var = fmaxf(s2 / (float)d - mean * mean, 0.f);
            if ((tid & 7) == 0) {
                const int row = (tid + c * NT) >> 3;
                s_mu0[row] = mean;
                s_rs0[row] = rsqrt_fast(var + kLnEps);
            }
        }
    }
    __syncthreads();  // stats visible; every wave is done with the staging ring
    PSTAMP(3)

    // ---- after fc1: epilogue 1 (LN0 fold, bias, SiLU, LN1 sums), fc2, epilogue 2, fc3 -----------
    // LayerNorm-1 is NOT applied element-wise: fc2 consumes the raw SiLU outputs and
    // its epilogue applies  rstd1*(W2~ s - mean1*rowsum(W2~)) + b2~  (same algebra as LN0).
    float* bufS1 = s_red;               // [NWV][ROWS] partial sums of s
    float* bufS2 = s_red + NWV * ROWS;  // [NWV][ROWS] partial sums of s*s
    float* setT0 = s_red + 2 * NWV * ROWS;  // [4][NWV][64]: sum s2, sum s2^2, fc3 class 0 / 1 partials
    float* setT1 = s_red;                   // second set of the same (128-row tiles, once bufS is dead)

    // epilogue 1 on column tiles [c0, c0 + nc): SiLU in place, partial LN1 sums -> bufS
    auto ep1_cols = [&](const int c0, const int nc) {
        float mu[CT], rs[CT], S1[CT], S2[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (c >= c0 && c < c0 + nc) {
                mu[c] = RAW ? s_mu0[32 * c + r] : 0.f;
                rs[c] = (RAW ? s_rs0[32 * c + r] : 1.f) * L_sc1;
                S1[c] = S2[c] = 0.f;
            }
        const int epg = nc >= 4 ? 1 : nc == 2 ? 2 : 4;   // rows of the accumulator per group: 4 chains
#pragma unroll
        for (int rti = 0; rti < RT; ++rti)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int nb = 32 * (RT * w + rti) + 8 * g4 + 4 * hh;
                const f32x4 ws = *reinterpret_cast<const f32x4*>(s_cst + nb);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(s_cst + kHidden + nb);
#pragma unroll
                for (int e0 = 0; e0 < 4; e0 += epg) {
                    float sv[4][CT];
#pragma unroll
                    for (int e = e0; e < e0 + epg; ++e)
#pragma unroll
                        for (int c = 0; c < CT; ++c)
                            if (c >= c0 && c < c0 + nc) {
                                float v = acc[rti][c][4 * g4 + e];
                                if constexpr (RAW) v = fmaf(-mu[c], ws[e], v);
                                sv[e][c] = silu_f(fmaf(rs[c], v, bb[e]));
                            }
                    // four independent SiLU chains per group, and the group pinned where it is written:
                    // pure arithmetic is free to move across sched_barrier at IR level - without the empty
                    // asm the compiler ran one chain per group and left the other 3*32 in one block at the
                    // end (serial dependent VALU, transcendental hazards exposed, spills)
#pragma unroll
                    for (int e = e0; e < e0 + epg; ++e)
#pragma unroll
                        for (int c = 0; c < CT; ++c)
                            if (c >= c0 && c < c0 + nc) asm volatile("" : "+v"(sv[e][c]));
#pragma unroll
                    for (int e = e0; e < e0 + epg; ++e)
#pragma unroll
                        for (int c = 0; c < CT; ++c)
                            if (c >= c0 && c < c0 + nc) {
                                acc[rti][c][4 * g4 + e] = sv[e][c];
                                S1[c] += sv[e][c];
                                S2[c] = fmaf(sv[e][c], sv[e][c], S2[c]);
                            }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (c >= c0 && c < c0 + nc) {
                const float t1 = xor32(S1[c]), t2 = xor32(S2[c]);
                if (hh == 0) {
                    bufS1[w * ROWS + 32 * c + r] = t1;
                    bufS2[w * ROWS + 32 * c + r] = t2;
                }
            }
    };

    // LN1 statistics of column tiles [c0, c0 + nc) from the partial sums of all waves
    float mean1[CT], rstd1[CT];
    auto mean_cols = [&](const int c0, const int nc) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (c >= c0 && c < c0 + nc) {
                const int m = 32 * c + r;
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int ww = 0; ww < NWV; ++ww) {
                    t1 += bufS1[ww * ROWS + m];
                    t2 += bufS2[ww * ROWS + m];
                }
                mean1[c] = t1 * (1.0f / kHidden);
                const float var = fmaxf(t2 * (1.0f / kHidden) - mean1[c] * mean1[c], 0.f);
                rstd1[c] = rsqrt_fast(var + kLnEps);
                // one column tile's 2*NWV reads at a time (hoisted together they went to scratch)
                asm volatile("" : "+v"(mean1[c]), "+v"(rstd1[c]));
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    // fc2 weight fragments: two slots, refilled right after use.  The fragments do not depend on the column
    // tile, so the refills at the end of one pass wrap around to the first k-steps and the next pass starts with
    // them already in registers.  (Round 2 gave the 128-row fp16 variant four slots - a wave that runs the k loop
    // alone on its SIMD has only 512 cycles of prefetch with two - at the price of 21 spilled registers / 84 B of
    // scratch; round 3 measured both on one box: back to back 85.4 vs 86.7 us, behind a cache flush 101.4 vs
    // 98.4 us, in `bench.py` 95.5-96.5 vs 96.2 us.  Two slots: 255 VGPRs, no scratch.)
    constexpr int A2S = 2;
    half8 a2[A2S][NA][RT];
    // fp8 copy of this wave's W2 rows for the lo term: two slots of one k block (64 hidden units) each
    i32x8 q2[2][RT];
    auto q2_load = [&](int slot, int kb) {
        if constexpr (LO8) {
#pragma unroll
            for (int rti = 0; rti < RT; ++rti) {
                const unsigned frag = ((unsigned)kb * 16u + w_frag0 + rti) << 11;
                const u32x4 v0 = buf_load16(rs_w2q, lane16, frag);
                const u32x4 v1 = buf_load16(rs_w2q, lane16, frag + 1024u);
                q2[slot][rti] = i32x8{(int)v0[0], (int)v0[1], (int)v0[2], (int)v0[3],
                                      (int)v1[0], (int)v1[1], (int)v1[2], (int)v1[3]};
            }
        }
    };
    auto a2_load = [&](int slot, int ks) {
#pragma unroll
        for (int p = 0; p < NA; ++p)
#pragma unroll
            for (int rti = 0; rti < RT; ++rti) {
                const u32x4 v = buf_load16(rs_w2, lane16, ((unsigned)(p * 32 + ks) * 16u + w_frag0 + rti) << 10);
                a2[slot][p][rti] = __builtin_bit_cast(half8, v);
            }
    };

    // publish this wave's SiLU outputs of pass g as ready-made B fragments (hi + lo halves, RTZ split)
    auto publish = [&](const int g) {
#pragma unroll
        for (int rti = 0; rti < RT; ++rti)
#pragma unroll
            for (int c2 = 0; c2 < G; ++c2) {
                const int c = g * G + c2;
                u32x4 lo8;   // LO8: this row tile's 16 lo values as fp8, byte e of the lane's operand half rti
#pragma unroll
                for (int s2i = 0; s2i < 2; ++s2i) {
                    u32x4 hi, lo;
                    float lq[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v0 = acc[rti][c][8 * s2i + 2 * j], v1 = acc[rti][c][8 * s2i + 2 * j + 1];
                        half2_t h01;
                        if constexpr (LO8) {   // nearest (v_cvt_pk_f16_f32): halves |lo|
                            typedef float float2_t __attribute__((ext_vector_type(2)));
                            h01 = __builtin_convertvector(float2_t{v0, v1}, half2_t);
                        } else {
                            h01 = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(v0, v1));
                        }
                        hi[j] = __builtin_bit_cast(unsigned int, h01);
                        if constexpr (LO8) {
                            // (v - hi) * 2^13, clamped to the e4m3 range (an overflow converts to NaN)
                            const float kS = (float)(1 << kLoShift);
                            lq[2 * j] = __builtin_amdgcn_fmed3f(fmaf(v0, kS, -kS * (float)h01[0]), -448.f, 448.f);
                            lq[2 * j + 1] = __builtin_amdgcn_fmed3f(fmaf(v1, kS, -kS * (float)h01[1]), -448.f, 448.f);
                        } else {
                            const auto l01 = __builtin_amdgcn_cvt_pkrtz(v0 - (float)h01[0], v1 - (float)h01[1]);
                            lo[j] = __builtin_bit_cast(unsigned int, l01);
                        }
                    }
                    if constexpr (LO8) {
#pragma unroll
                        for (int dd = 0; dd < 2; ++dd) {
                            int wd = 0;
                            wd = __builtin_amdgcn_cvt_pk_fp8_f32(lq[4 * dd], lq[4 * dd + 1], wd, false);
                            wd = __builtin_amdgcn_cvt_pk_fp8_f32(lq[4 * dd + 2], lq[4 * dd + 3], wd, true);
                            lo8[2 * s2i + dd] = (unsigned)wd;
                        }
                    }
                    const int fi = (((RT * w + rti) * 2 + s2i) * G + c2) * 64 + lane;
                    *reinterpret_cast<u32x4*>(s_ex + (size_t)fi * 16) = hi;
                    if constexpr (!LO8) *reinterpret_cast<u32x4*>(s_ex + (size_t)(16 * 2 * G * 64 + fi) * 16) = lo;
                }
                if constexpr (LO8)   // [k block][c2][half][lane] x 16 B behind the hi fragments
                    *reinterpret_cast<u32x4*>(s_ex + (size_t)16 * 2 * G * 1024 +
                                              (size_t)(((((RT / 2) * w + (rti >> 1)) * G + c2) * 2 + (rti & 1)) * 64 + lane) * 16) = lo8;
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    auto zero2 = [&](f32x16 (&acc2)[RT][G]) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int c = 0; c < G; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[i][c][e] = 0.f;
    };

    // the fc2 k loop of one pass; exchange fragments of k-step ks+1 are read while the MFMAs of ks run
    auto fc2_loop = [&](f32x16 (&acc2)[RT][G]) {
        half8 b2h[2][G], b2l[LO8 ? 1 : 2][LO8 ? 1 : G];
        auto b2_read = [&](int buf, int ks) {
#pragma unroll
            for (int c2 = 0; c2 < G; ++c2) {
                const int fi = (ks * G + c2) * 64 + lane;
                b2h[buf][c2] = *reinterpret_cast<const half8*>(s_ex + (size_t)fi * 16);
                if constexpr (!LO8)
                    b2l[buf][c2] = *reinterpret_cast<const half8*>(s_ex + (size_t)(16 * 2 * G * 64 + fi) * 16);
            }
        };
        b2_read(0, 0);
        auto hi_iter = [&](const int ks2) {
#pragma unroll
            for (int u = 0; u < A2S; ++u) {
                const int ks = ks2 + u;
                b2_read((u & 1) ^ 1, (ks + 1) & 31);  // the last one wraps to k-step 0: valid bytes, never used
#pragma unroll
                for (int rti = 0; rti < RT; ++rti)
#pragma unroll
                    for (int c2 = 0; c2 < G; ++c2) {
                        acc2[rti][c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            a2[u][0][rti], b2h[u & 1][c2], acc2[rti][c2], 0, 0, 0);
                        if constexpr (!LO8)
                            acc2[rti][c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                a2[u][0][rti], b2l[u & 1][c2], acc2[rti][c2], 0, 0, 0);
                        if constexpr (NA == 2)
                            acc2[rti][c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                a2[u][1][rti], b2h[u & 1][c2], acc2[rti][c2], 0, 0, 0);
                    }
                a2_load(u, (ks + A2S) & 31);  // refill, wrapping to the next pass's first fragments: no branch
                // one LDS read / one weight load in the shadow of each MFMA: a wave that has the SIMD to
                // itself (the other one is in its VALU phase) then keeps the pipe at one MFMA per 32 cycles
                if constexpr (NA == 1) {
                    constexpr int n_ds = LO8 ? G : 2 * G, n_mf = LO8 ? RT * G : 2 * RT * G;
                    constexpr int n_pair = n_ds < n_mf ? n_ds : n_mf;
#pragma unroll
                    for (int i = 0; i < n_pair; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                    }
                    constexpr int n_vm = (n_mf - n_pair) < RT ? (n_mf - n_pair) : RT;
#pragma unroll
                    for (int i = 0; i < n_vm; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                    }
                    if constexpr (n_mf - n_pair - n_vm > 0)
                        __builtin_amdgcn_sched_group_barrier(0x008, n_mf - n_pair - n_vm, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (LO8) {
#pragma unroll 1
            for (int ks2 = 0; ks2 < 32 - A2S; ks2 += A2S) hi_iter(ks2);
            q2_load(0, 0);   // the lo term's first weight blocks, one iteration before they are used
            q2_load(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            hi_iter(32 - A2S);
            // lo term: eight k blocks of 64 hidden units (one per pair of published row tiles), fp8 x fp8
            const char* s_lo = s_ex + (size_t)16 * 2 * G * 1024;
            i32x8 bq[2][G];
            auto bq_read = [&](int buf, int kb) {
#pragma unroll
                for (int c2 = 0; c2 < G; ++c2) {
                    const char* pz = s_lo + (size_t)(((kb * G + c2) * 2) * 64 + lane) * 16;
                    const u32x4 v0 = *reinterpret_cast<const u32x4*>(pz);
                    const u32x4 v1 = *reinterpret_cast<const u32x4*>(pz + 1024);
                    bq[buf][c2] = i32x8{(int)v0[0], (int)v0[1], (int)v0[2], (int)v0[3],
                                        (int)v1[0], (int)v1[1], (int)v1[2], (int)v1[3]};
                }
            };
            bq_read(0, 0);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (kb + 1 < 8) bq_read((kb & 1) ^ 1, kb + 1);
#pragma unroll
                for (int rti = 0; rti < RT; ++rti)
#pragma unroll
                    for (int c2 = 0; c2 < G; ++c2)
                        acc2[rti][c2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                            q2[kb & 1][rti], bq[kb & 1][c2], acc2[rti][c2], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                if (kb + 2 < 8) q2_load(kb & 1, kb + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 1
            for (int ks2 = 0; ks2 < 32; ks2 += A2S) hi_iter(ks2);
        }
    };

    // epilogue 2 of pass g: LN1 fold, bias, SiLU, one-pass LN2 sums, fc3 partial dot products -> set
    auto ep2 = [&](const int g, f32x16 (&acc2)[RT][G], float* set) {
        float T1[G], T2[G], P0[G], P1[G], m1[G], r1[G];
#pragma unroll
        for (int c2 = 0; c2 < G; ++c2) {
            T1[c2] = T2[c2] = P0[c2] = P1[c2] = 0.f;
            m1[c2] = mean1[g * G + c2];
            r1[c2] = rstd1[g * G + c2] * L_sc2;
        }
#pragma unroll
        for (int rti = 0; rti < RT; ++rti)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int nb = 32 * (RT * w + rti) + 8 * g4 + 4 * hh;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(s_cst + 2 * kHidden + nb);
                const f32x4 w2s = *reinterpret_cast<const f32x4*>(s_cst + 5 * kHidden + nb);
                const f32x4 w30 = *reinterpret_cast<const f32x4*>(s_cst + 3 * kHidden + nb);
                const f32x4 w31 = *reinterpret_cast<const f32x4*>(s_cst + 4 * kHidden + nb);
                float sv[4][G];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c2 = 0; c2 < G; ++c2) {
                        const float v = fmaf(-m1[c2], w2s[e], acc2[rti][c2][4 * g4 + e]);
                        sv[e][c2] = silu_f(fmaf(r1[c2], v, bb[e]));
                    }
                // 4*G independent chains per group, pinned here (see epilogue 1)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c2 = 0; c2 < G; ++c2) asm volatile("" : "+v"(sv[e][c2]));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c2 = 0; c2 < G; ++c2) {
                        T1[c2] += sv[e][c2];
                        T2[c2] = fmaf(sv[e][c2], sv[e][c2], T2[c2]);
                        P0[c2] = fmaf(sv[e][c2], w30[e], P0[c2]);
                        P1[c2] = fmaf(sv[e][c2], w31[e], P1[c2]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
        for (int c2 = 0; c2 < G; ++c2) {
            const float t1 = xor32(T1[c2]), t2 = xor32(T2[c2]), q0 = xor32(P0[c2]), q1 = xor32(P1[c2]);
            if (hh == 0) {
                set[(0 * NWV + w) * 64 + 32 * c2 + r] = t1;
                set[(1 * NWV + w) * 64 + 32 * c2 + r] = t2;
                set[(2 * NWV + w) * 64 + 32 * c2 + r] = q0;
                set[(3 * NWV + w) * 64 + 32 * c2 + r] = q1;
            }
        }
    };

    // logits of row (32*G*g + i) from the cross-wave sums of a set:
    // logits = W3~ . LN2(s2) + b3~ = rstd2 * (W3~.s2 - mean2 * rowsum(W3~)) + b3~
    auto logits_row = [&](const int g, const int i, const float* set) {
        const int row = m0 + 32 * g * G + i;
        if (row < a.B) {
            float t1 = 0.f, t2 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < NWV; ++ww) {
                t1 += set[(0 * NWV + ww) * 64 + i];
                t2 += set[(1 * NWV + ww) * 64 + i];
                q0 += set[(2 * NWV + ww) * 64 + i];
                q1 += set[(3 * NWV + ww) * 64 + i];
            }
            const float mean2 = t1 * (1.0f / kHidden);
            const float var2 = fmaxf(t2 * (1.0f / kHidden) - mean2 * mean2, 0.f);
            const float rstd2 = rsqrt_fast(var2 + kLnEps);
            float2 o;
            o.x = fmaf(rstd2, q0 - mean2 * L_w3sum[0], L_b3[0]);
            o.y = fmaf(rstd2, q1 - mean2 * L_w3sum[1], L_b3[1]);
            *reinterpret_cast<float2*>(a.logits + ((size_t)lrun * a.B + row) * 2) = o;
        }
    };

    if constexpr (NG == 2 && ROWS == 128) {
        // Two fc2 passes, and the two waves of every SIMD (w and w + NWV/2) take the VALU phase and the
        // MFMA phase of a pass in opposite orders: while one runs the k loop of pass 0 the other applies
        // epilogue 1 to the column tiles of pass 1, and epilogue 2 of pass 0 runs beside the k loop of
        // pass 1.  (In one order for everybody the matrix pipe idles through every epilogue.)
        f32x16 acc2a[RT][G], acc2b[RT][G];
        const bool loop_first = w >= NWV / 2;
        ep1_cols(0, G);
        PSTAMP(4)
#pragma unroll
        for (int u = 0; u < A2S; ++u) {
            a2_load(u, u);
            __builtin_amdgcn_sched_barrier(0);
        }
        publish(0);
        zero2(acc2a);
        PSTAMP(5)
        __syncthreads();   // fragments and the LN1 partial sums of pass 0 are in LDS
        mean_cols(0, G);
        PSTAMP(6)
        if (loop_first) {
            __builtin_amdgcn_s_setprio(2);   // finish the k loop first: the other wave then has the pipe alone
            fc2_loop(acc2a);
            __builtin_amdgcn_s_setprio(0);
        }
        ep1_cols(G, G);
        if (!loop_first) fc2_loop(acc2a);
        PSTAMP(7)
        __syncthreads();   // every wave is done with the exchange area; LN1 partial sums of pass 1 are in LDS
        publish(1);
        zero2(acc2b);
        __syncthreads();
        mean_cols(G, G);
        __syncthreads();   // bufS is dead from here on: its bytes become the second set of cross-wave sums
        PSTAMP(8)
        if (!loop_first) {
            __builtin_amdgcn_s_setprio(2);
            fc2_loop(acc2b);
            __builtin_amdgcn_s_setprio(0);
        }
        ep2(0, acc2a, setT0);
        if (loop_first) fc2_loop(acc2b);
        PSTAMP(9)
        ep2(1, acc2b, setT1);
        PSTAMP(10)
        __syncthreads();
        if (tid < 64) logits_row(0, tid, setT0);
        else if (tid < 128) logits_row(1, tid - 64, setT1);
        PSTAMP(11)
    } else {
        ep1_cols(0, CT);
        PSTAMP(4)
        a2_load(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a2_load(1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g > 0) __syncthreads();  // previous pass finished reading the exchange area / the sums
            publish(g);
            f32x16 acc2[RT][G];
            zero2(acc2);
            __syncthreads();  // fragments (and, first time, the LN1 partial sums) of all waves are in LDS
            if (g == 0) mean_cols(0, CT);
            fc2_loop(acc2);
            ep2(g, acc2, setT0);
            __syncthreads();
            if (tid < 32 * G) logits_row(g, tid, setT0);
        }
        PSTAMP(11)
    }
}

#ifdef PRAG_MM_DIAG
extern "C" int prag_diag_prober_stamps(unsigned long long* out, int n) {
    if (n > 2 * 3 * 8 * 32) n = 2 * 3 * 8 * 32;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -2;
}
#endif

// ---------------------------------------------------------------------------
// fp32 activations -> LayerNorm-0-normalised hi/lo fp16 terms (one wave per row)
// ---------------------------------------------------------------------------
// T = float or bf16 bits (unsigned short)
template <typename T>
__device__ __forceinline__ f32x4 load4_as_f32(const T* p) {
    if constexpr (sizeof(T) == 4) {
        return *reinterpret_cast<const f32x4*>(p);
    } else {  // bf16: the upper half of an f32
        const ushort4 u = *reinterpret_cast<const ushort4*>(p);
        return f32x4{__uint_as_float((uint32_t)u.x << 16), __uint_as_float((uint32_t)u.y << 16),
                     __uint_as_float((uint32_t)u.z << 16), __uint_as_float((uint32_t)u.w << 16)};
    }
}

template <typename T>
__global__ __launch_bounds__(256) void prenorm_split_kernel(const T* __restrict__ x,
                                                           int64_t x_layer_stride, int B, int d,
                                                           int n_rows_total, _Float16* __restrict__ xh,
                                                           _Float16* __restrict__ xl) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows_total) return;
    const int l = row / B, b = row - l * B;
    const T* src = x + (int64_t)l * x_layer_stride + (int64_t)b * d;
    _Float16* dh = xh + (int64_t)row * d;
    _Float16* dl = xl + (int64_t)row * d;
    // Round 6: a row of up to 4096 elements is read ONCE into registers (16 x 16 B per lane) - rounds 1-5 read it three
    // times, three dependent memory round trips per row (the pre-pass of the reference-precision gate at B = 4096: ~115 us
    // for 400 MB of traffic).  Same sums in the same order: bit-identical outputs.
    if (d <= 4096) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = lane * 4 + u * 256;
            v[u] = i < d ? load4_as_f32(src + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (lane * 4 + u * 256 < d) s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)d;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (lane * 4 + u * 256 < d) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dv = v[u][e] - mean;
                    q = fmaf(dv, dv, q);
                }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = 1.0f / sqrtf(q / (float)d + kLnEps);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = lane * 4 + u * 256;
            if (i < d) {
                half4 h, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float n = (v[u][e] - mean) * rstd;
                    const _Float16 h16 = (_Float16)n;
                    h[e] = h16;
                    lo[e] = (_Float16)(n - (float)h16);
                }
                *reinterpret_cast<half4*>(dh + i) = h;
                *reinterpret_cast<half4*>(dl + i) = lo;
            }
        }
        return;
    }
    // pass 1: mean ; pass 2: centred second moment (rows are L2/L1 hot on re-read)
    float s = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        const f32x4 v = load4_as_f32(src + i);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)d;
    float q = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        const f32x4 v = load4_as_f32(src + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dv = v[e] - mean;
            q = fmaf(dv, dv, q);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)d + kLnEps);
    for (int i = lane * 4; i < d; i += 256) {
        const f32x4 v = load4_as_f32(src + i);
        half4 h, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float n = (v[e] - mean) * rstd;
            const _Float16 h16 = (_Float16)n;
            h[e] = h16;
            lo[e] = (_Float16)(n - (float)h16);
        }
        *reinterpret_cast<half4*>(dh + i) = h;
        *reinterpret_cast<half4*>(dl + i) = lo;
    }
}

// attention-masked mean over the sequence: one block column per 4 features
template <typename T>
__global__ __launch_bounds__(256) void pool_masked_mean_kernel(const T* __restrict__ h,
                                                              const int64_t* __restrict__ mask, int Tlen, int d,
                                                              float* __restrict__ out) {
    const int b = blockIdx.y;
    const int col = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (col >= d) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    float cnt = 0.f;
    for (int t = 0; t < Tlen; ++t) {
        const float m = (float)mask[(int64_t)b * Tlen + t];
        if (m != 0.f) {
            const T* p = h + ((int64_t)b * Tlen + t) * d + col;
            f32x4 v;
            if constexpr (sizeof(T) == 4) v = *reinterpret_cast<const f32x4*>(p);
            else if constexpr (std::is_same<T, _Float16>::value) {
                const half4 hv = *reinterpret_cast<const half4*>(p);
                v = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
            } else v = load4_as_f32(p);
            s += v * m;
            cnt += m;
        }
    }
    *reinterpret_cast<f32x4*>(out + (int64_t)b * d + col) = s / fmaxf(cnt, 1.0f);   // clamp(min=1e-9) ~ no-op for int masks
}

// ---------------------------------------------------------------------------
// gate: exp_rag.py:407-415 in the reference's own order (float32, layer by layer)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ logits, int L, int B,
                                                  int ablation, double theta,
                                                  float* __restrict__ probsum,
                                                  int32_t* __restrict__ decision) {
    gate_row(logits, L, B, ablation, theta, probsum, decision, blockIdx.x * blockDim.x + threadIdx.x);   // (prober_internal.h)
}

// ---------------------------------------------------------------------------
// pooling: on-device replacements for the hook cache
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pool_accumulate_kernel(float* __restrict__ acc,
                                                             const T* __restrict__ h, int64_t n,
                                                             int assign) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        f32x4 v;
        if constexpr (sizeof(T) == 4) {
            v = *reinterpret_cast<const f32x4*>(h + i);
        } else if constexpr (std::is_same<T, _Float16>::value) {
            const half4 hv = *reinterpret_cast<const half4*>(h + i);
            v = {(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        } else {   // bf16 bits
            v = load4_as_f32(h + i);
        }
        f32x4* dst = reinterpret_cast<f32x4*>(acc + i);
        *dst = assign ? v : (*dst + v);
    }
}

// the same for every probed layer of one decode step in ONE launch: layer l's activations sit wherever the model
// left them (pointer table passed by value), its accumulator at acc + l * n
struct PoolLayerPtrs {
    const void* h[64];
};
template <typename T>
__global__ __launch_bounds__(256) void pool_accumulate_layers_kernel(float* __restrict__ acc, PoolLayerPtrs ptrs,
                                                                    int64_t n, int assign) {
    const T* __restrict__ h = reinterpret_cast<const T*>(ptrs.h[blockIdx.y]);
    float* __restrict__ dst_l = acc + (int64_t)blockIdx.y * n;
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        f32x4 v;
        if constexpr (sizeof(T) == 4) {
            v = *reinterpret_cast<const f32x4*>(h + i);
        } else if constexpr (std::is_same<T, _Float16>::value) {
            const half4 hv = *reinterpret_cast<const half4*>(h + i);
            v = {(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        } else {   // bf16 bits
            v = load4_as_f32(h + i);
        }
        f32x4* dst = reinterpret_cast<f32x4*>(dst_l + i);
        *dst = assign ? v : (*dst + v);
    }
}

// out[b,:] = sum or mean of the last pred_lens[b] positions of acts[b] ([B,T,d])
template <typename T>
__global__ __launch_bounds__(256) void pool_ragged_kernel(const T* __restrict__ acts, int Tlen, int d,
                                                         const int64_t* __restrict__ pred_lens,
                                                         int scale_mean, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int col = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (col >= d) return;
    int64_t n = pred_lens[b];
    n = n < 0 ? 0 : (n > Tlen ? Tlen : n);
    const T* base = acts + ((int64_t)b * Tlen + (Tlen - n)) * d + col;
    // sequential accumulation in position order, like torch.sum/mean over dim 0
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int64_t t = 0; t < n; ++t) {
        if constexpr (sizeof(T) == 4) {
            s += *reinterpret_cast<const f32x4*>(base + t * d);
        } else if constexpr (std::is_same<T, _Float16>::value) {
            const half4 hv = *reinterpret_cast<const half4*>(base + t * d);
            s += f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        } else {   // bf16 bits
            s += load4_as_f32(base + t * d);
        }
    }
    if (scale_mean && n > 0) s = s / (float)n;
    *reinterpret_cast<f32x4*>(out + (int64_t)b * d + col) = s;
}

// train.py:153-162 / utils.py:134-143 (`input_tensor_method1`, the `each_token` method - train.py:354's default):
// the last pred_lens[b] positions of every sample, concatenated in sample order -> out [n_rows][d] float32, and
// the sample's label repeated for each of its rows (torch.repeat_interleave).  row_off[b] = rows of the samples
// before b (row_off[B] = n_rows); blockIdx.x = output row, blockIdx.y = 1024-column chunk.
template <typename T>
__global__ __launch_bounds__(256) void pool_each_token_kernel(const T* __restrict__ acts, int B, int Tlen, int d,
                                                             const int64_t* __restrict__ row_off,
                                                             const int32_t* __restrict__ labels, float* __restrict__ out,
                                                             int32_t* __restrict__ labels_out) {
    const int64_t r = blockIdx.x;
    int lo = 0, hi = B;                         // the sample b with row_off[b] <= r < row_off[b + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (row_off[mid] <= r) lo = mid; else hi = mid;
    }
    const int b = lo;
    const int64_t n = row_off[b + 1] - row_off[b];
    const int64_t t = (int64_t)Tlen - n + (r - row_off[b]);
    if (labels_out && labels && blockIdx.y == 0 && threadIdx.x == 0) labels_out[r] = labels[b];
    const int col = (blockIdx.y * blockDim.x + threadIdx.x) * 4;
    if (col >= d) return;
    const T* src = acts + ((int64_t)b * Tlen + t) * d + col;
    f32x4 v;
    if constexpr (sizeof(T) == 4) {
        v = *reinterpret_cast<const f32x4*>(src);
    } else if constexpr (std::is_same<T, _Float16>::value) {
        const half4 hv = *reinterpret_cast<const half4*>(src);
        v = {(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
    } else {   // bf16 bits
        v = load4_as_f32(src);
    }
    *reinterpret_cast<f32x4*>(out + r * d + col) = v;
}

}  // namespace prag

// ===========================================================================
// host side
// ===========================================================================
using namespace prag;

struct HostLayer {  // kept for prag_prober_effective_weights
    std::vector<float> W1, b1, W2, b2, W3, b3;
};

struct prag_prober {
    int n_layers, d, na;
    std::vector<bool> loaded;
    std::vector<LayerDev> h_layers;
    std::vector<HostLayer> eff;
    std::vector<std::vector<void*>> allocs;  // device buffers per layer (freed when the layer is reloaded)
    int upload_layer = 0;
    LayerDev* d_layers = nullptr;
    // small-batch path (prober_small.hip): f32 effective weights + two [L][8][512] workspaces
    std::vector<SmallLayer> h_small;
    SmallLayer* d_small = nullptr;
    float* small_ws = nullptr;
    void* small_sync = nullptr;   // arrival counters of the one-launch form (zero between launches)
    // PRAG_PROBER_SMALL=0: always the MFMA-tiled kernel; 1 (default): three short launches; 3: the same stages in
    // ONE launch with in-launch hand-offs (round 4: built, bit-identical, and SLOWER - 30.2 against 25.2 us per gate
    // at one pooled state, profiles/r04c_latency.txt: a hand-off costs a write-through drain, an atomic, a poll and
    // a load from beyond L2, ~5 us, where a kernel boundary costs ~2 us and leaves the data in L2).  Round 6: the
    // one-launch form is what a decode step of the LM ends with (prag_pool_step_gate below) - there the launch count
    // per token is what the host pays for)
    int small_mode = 1;
    int ct_force = 0;            // PRAG_PROBER_CT at creation: row-tile height override (tuning runs)
    _Float16* ws_h = nullptr;  // [L][maxB][d] hi / lo workspace for fp32 activations
    _Float16* ws_l = nullptr;
    int64_t ws_rows = 0;
    int n_cu = 256;            // compute units of the device the handle lives on (tile-height choice)
    int shape16 = 1;           // fp16 x fp16 mode on 16 x 16 MFMA tiles (PRAG_PROBER_SHAPE=32: the 32 x 32 kernel)
    EventRing prof;
    // prag_gate_decide: outputs owned by the handle.  Up to kDecideDirect rows the kernels write probsum / decision
    // straight into mapped, coherent host memory (no copy node, no second launch); larger batches go through device
    // buffers and ONE copy into the same pinned block.
    float* dec_logits = nullptr;        // device [L][dec_cap][2]
    char* dec_host = nullptr;           // pinned + mapped: int32 decision[dec_cap] | float probsum[dec_cap][2]
    char* dec_host_dev = nullptr;       // the same block as the device sees it
    char* dec_dev = nullptr;            // device staging for batches beyond kDecideDirect
    int dec_cap = 0;
    int dec_spin = 1;                   // PRAG_DECIDE_SPIN=0 at creation: always hipStreamSynchronize
    // prag_pool_step_gate: the result block of the most recent step (mapped pinned memory the kernel writes):
    // uint64 tag | int32 decision[8] | float probsum[8][2]
    uint64_t* step_host = nullptr;
    uint64_t* step_host_dev = nullptr;
    uint64_t step_tag = 0;              // tag of the most recent step enqueued on this handle
    // tickets of the gate folded into prober16_body: one word per row tile, zero between launches
    uint32_t* tile_cnt = nullptr;
    int tile_cnt_cap = 0;
    // PRAG_GATE_FOLD=1 at creation: the gate inside the prober launch.  Built, bit-identical, and SLOWER as it stands
    // (profiles/r05i_shard_ab.txt: the launch that carries the 512-row gate 37.5 -> 57.8 us, pass 0.443 -> 0.460 ms): the
    // hand-off costs every workgroup a device-scope fence (an L2 write-back on this chip) and the last arriver twelve
    // dependent loads from beyond its L2 per row, where gate_kernel's launch costs ~9 us.  Off by default.
    int gate_fold = 0;
};
constexpr int kDecideDirect = 256;

static int pick_scale_exp(double maxabs) {
    if (!(maxabs > 0.0)) return 0;
    // bring the largest magnitude into [2^13, 2^14): no fp16 subnormal terms, no overflow
    return 13 - (int)std::floor(std::log2(maxabs));
}

// pack an [512][K] (double, already scaled) matrix into MFMA A-fragment order
// kmap(step, half, j) gives the source column of element j of lane-half `half` at k-step `step`
template <typename KMap>
static void pack_fragments(const std::vector<double>& Ws, int K, int n_steps, int na, KMap kmap,
                           std::vector<_Float16>& out, std::vector<double>& deq) {
    out.assign((size_t)na * n_steps * 16 * 64 * 8, (_Float16)0);
    deq.assign((size_t)kHidden * K, 0.0);
    for (int step = 0; step < n_steps; ++step)
        for (int rt = 0; rt < 16; ++rt)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = 32 * rt + (lane & 31), half = lane >> 5;
                for (int j = 0; j < 8; ++j) {
                    const int col = kmap(step, half, j);
                    const double v = Ws[(size_t)row * K + col];
                    const _Float16 hi = (_Float16)v;
                    const size_t o = ((((size_t)step) * 16 + rt) * 64 + lane) * 8 + j;
                    out[o] = hi;
                    double q = (double)hi;
                    if (na == 2) {
                        const _Float16 lo = (_Float16)(v - (double)hi);
                        out[(size_t)n_steps * 16 * 64 * 8 + o] = lo;
                        q += (double)lo;
                    }
                    deq[(size_t)row * K + col] = q;
                }
            }
}

// the same for 16 x 16 tiles (prober16.hip): fragment (step, hidden tile g of 16 rows), lane (c16 = lane & 15,
// q4 = lane >> 4) holds row 16 g + c16, element j -> column kmap(step, q4, j); fp16 only
template <typename KMap>
static void pack_fragments16(const std::vector<double>& Ws, int K, int n_steps, KMap kmap, std::vector<_Float16>& out) {
    out.assign((size_t)n_steps * 32 * 64 * 8, (_Float16)0);
    for (int step = 0; step < n_steps; ++step)
        for (int g = 0; g < 32; ++g)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = 16 * g + (lane & 15), q4 = lane >> 4;
                for (int j = 0; j < 8; ++j)
                    out[((((size_t)step) * 32 + g) * 64 + lane) * 8 + j] = (_Float16)Ws[(size_t)row * K + kmap(step, q4, j)];
            }
}

// OCP fp8 e4m3 (no infinities, max 448), round to nearest even, saturating: what v_cvt_pk_fp8_f32 produces
static unsigned char to_e4m3(double v) {
    const unsigned char sgn = std::signbit(v) ? 0x80 : 0x00;
    double a = std::fabs(v);
    if (!(a == a)) return 0x7f;
    if (a >= 448.0) return sgn | 0x7e;
    if (a < std::ldexp(1.0, -6)) return sgn | (unsigned char)std::nearbyint(a * 512.0);   // subnormals: k * 2^-9 (8 = 2^-6)
    int e;
    (void)std::frexp(a, &e);
    int E = e - 1;                                                  // a = 1.xxx * 2^E
    int q = (int)std::nearbyint(std::ldexp(a, 3 - E)) - 8;          // 3 mantissa bits
    if (q == 8) {
        q = 0;
        ++E;
    }
    const int code = ((E + 7) << 3) | q;
    return sgn | (unsigned char)(code > 0x7e ? 0x7e : code);
}

// fp8 copy of the (scaled) fc2 matrix for the lo term: [8 k blocks][16 row tiles][2 halves][64 lanes][16 B].
// Byte j of half `hf` of lane (row r, lane half h) holds hidden unit 32*(2*kb + hf) + 8*(j>>2) + 4*h + (j&3):
// the order in which a wave's two accumulator tiles sit in its lanes (see `publish`).
static void pack_w2q(const std::vector<double>& Ws, std::vector<unsigned char>& out) {
    out.assign((size_t)8 * 16 * 2 * 64 * 16, 0);
    for (int kb = 0; kb < 8; ++kb)
        for (int rt = 0; rt < 16; ++rt)
            for (int hf = 0; hf < 2; ++hf)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 16; ++j) {
                        const int row = 32 * rt + (lane & 31), h = lane >> 5;
                        const int col = 32 * (2 * kb + hf) + 8 * (j >> 2) + 4 * h + (j & 3);
                        out[((((size_t)kb * 16 + rt) * 2 + hf) * 64 + lane) * 16 + j] =
                            to_e4m3(std::ldexp(Ws[(size_t)row * kHidden + col], -kLoShift));
                    }
}

// fp8 copy of the (scaled) fc2 matrix for the 16 x 16 kernel: [4 k blocks][32 hidden tiles][2 halves][64 lanes][16 B];
// byte b of half hf of lane (c16, q4): hidden unit 128 kb + 64 hf + 16 (b >> 2) + 4 q4 + (b & 3)
static void pack_w2q16(const std::vector<double>& Ws, std::vector<unsigned char>& out) {
    out.assign((size_t)4 * 32 * 2 * 64 * 16, 0);
    for (int kb = 0; kb < 4; ++kb)
        for (int g = 0; g < 32; ++g)
            for (int hf = 0; hf < 2; ++hf)
                for (int lane = 0; lane < 64; ++lane)
                    for (int b = 0; b < 16; ++b) {
                        const int row = 16 * g + (lane & 15), q4 = lane >> 4;
                        const int col = 128 * kb + 64 * hf + 16 * (b >> 2) + 4 * q4 + (b & 3);
                        out[((((size_t)kb * 32 + g) * 2 + hf) * 64 + lane) * 16 + b] =
                            to_e4m3(std::ldexp(Ws[(size_t)row * kHidden + col], -kLoShift));
                    }
}

template <typename T>
static int upload(prag_prober* p, const std::vector<T>& v, const T** out) {
    void* dptr = nullptr;
    PRAG_HIP(hipMalloc(&dptr, v.size() * sizeof(T)));
    p->allocs[p->upload_layer].push_back(dptr);
    PRAG_HIP(hipMemcpy(dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T*>(dptr);
    return PRAG_OK;
}

extern "C" int prag_prober_create(prag_prober_t** out, int n_layers, int d_model, int d_hidden,
                                  int n_classes, int weight_mode) {
    PRAG_REQUIRE(out != nullptr, PRAG_EINVAL, "prag_prober_create: out is NULL");
    PRAG_REQUIRE(n_layers >= 1 && n_layers <= 64, PRAG_EINVAL, "n_layers=%d out of range [1,64]", n_layers);
    PRAG_REQUIRE(d_hidden == kHidden, PRAG_EUNSUPPORTED,
                 "d_hidden=%d: the reference prober has hidden_size=512 (utils.py:30)", d_hidden);
    PRAG_REQUIRE(n_classes == kClasses, PRAG_EUNSUPPORTED,
                 "n_classes=%d: the reference uses num_classes=2 (utils.py:289)", n_classes);
    PRAG_REQUIRE(d_model >= 128 && d_model % 64 == 0, PRAG_EUNSUPPORTED,
                 "d_model=%d must be a multiple of 64 and >= 128", d_model);
    PRAG_REQUIRE(weight_mode == PRAG_W_F16 || weight_mode == PRAG_W_F32, PRAG_EINVAL,
                 "weight_mode=%d", weight_mode);
    prag_prober* p = new (std::nothrow) prag_prober();
    PRAG_REQUIRE(p != nullptr, PRAG_ENOMEM, "out of host memory");
    p->n_layers = n_layers;
    p->d = d_model;
    p->na = (weight_mode == PRAG_W_F32) ? 2 : 1;
    p->loaded.assign(n_layers, false);
    p->h_layers.resize(n_layers);
    p->eff.resize(n_layers);
    p->allocs.resize(n_layers);
    p->h_small.resize(n_layers);
    if (const char* ev = getenv("PRAG_PROBER_SMALL")) p->small_mode = atoi(ev);
    if (const char* ev = getenv("PRAG_PROBER_CT")) p->ct_force = atoi(ev);
    if (const char* ev = getenv("PRAG_PROBER_SHAPE")) p->shape16 = atoi(ev) != 32;
    if (const char* ev = getenv("PRAG_DECIDE_SPIN")) p->dec_spin = atoi(ev) != 0;
    if (const char* ev = getenv("PRAG_GATE_FOLD")) p->gate_fold = atoi(ev) != 0;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            prop.multiProcessorCount > 0)
            p->n_cu = prop.multiProcessorCount;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->d_layers), sizeof(LayerDev) * n_layers);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_small), sizeof(SmallLayer) * n_layers);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&p->small_ws), sizeof(float) * 2 * n_layers * kSmallMaxB * kHidden);
    if (e == hipSuccess) e = hipMalloc(&p->small_sync, small_sync_bytes());
    if (e == hipSuccess) e = hipMemset(p->small_sync, 0, small_sync_bytes());
    if (e != hipSuccess) {
        if (p->d_layers) (void)hipFree(p->d_layers);
        if (p->d_small) (void)hipFree(p->d_small);
        if (p->small_ws) (void)hipFree(p->small_ws);
        set_error("hipMalloc failed: %s", hipGetErrorString(e));
        delete p;
        return PRAG_EHIP;
    }
    *out = p;
    return PRAG_OK;
}

extern "C" int prag_prober_load_layer(prag_prober_t* p, int li, const float* ln0_w, const float* ln0_b,
                                      const float* W1, const float* b1, const float* ln1_w,
                                      const float* ln1_b, const float* W2, const float* b2,
                                      const float* ln2_w, const float* ln2_b, const float* W3,
                                      const float* b3) {
    PRAG_REQUIRE(p != nullptr, PRAG_EINVAL, "prober handle is NULL");
    PRAG_REQUIRE(li >= 0 && li < p->n_layers, PRAG_EINVAL, "layer_idx=%d out of range", li);
    PRAG_REQUIRE(ln0_w && ln0_b && W1 && b1 && ln1_w && ln1_b && W2 && b2 && ln2_w && ln2_b && W3 && b3,
                 PRAG_EINVAL, "prag_prober_load_layer: NULL parameter array");
    const int d = p->d, H = kHidden;
    LayerDev& L = p->h_layers[li];
    HostLayer& E = p->eff[li];
    // a reload (checkpoint swap, load_state_dict per epoch) replaces the layer's buffers: wait for
    // launches that may still read the old ones, then free them
    if (!p->allocs[li].empty()) {
        PRAG_HIP(hipDeviceSynchronize());
        for (void* q : p->allocs[li]) (void)hipFree(q);
        p->allocs[li].clear();
        p->loaded[li] = false;
    }
    p->upload_layer = li;

    // ---- fc1: fold ln0 affine, scale, pack ---------------------------------
    std::vector<double> Wg((size_t)H * d);
    double mx = 0.0;
    for (int n = 0; n < H; ++n)
        for (int k = 0; k < d; ++k) {
            const double v = (double)W1[(size_t)n * d + k] * (double)ln0_w[k];
            Wg[(size_t)n * d + k] = v;
            mx = std::max(mx, std::fabs(v));
        }
    const int e1 = pick_scale_exp(mx);
    for (auto& v : Wg) v = std::ldexp(v, e1);
    std::vector<_Float16> packed;
    std::vector<double> deq;
    pack_fragments(Wg, d, d / 16, p->na,
                   [](int step, int half, int j) { return 16 * step + 8 * half + j; }, packed, deq);
    std::vector<float> wsum(H), b1e(H);
    E.W1.resize((size_t)H * d);
    E.b1.resize(H);
    for (int n = 0; n < H; ++n) {
        double s = 0.0, bb = (double)b1[n];
        for (int k = 0; k < d; ++k) {
            s += deq[(size_t)n * d + k];
            bb += (double)W1[(size_t)n * d + k] * (double)ln0_b[k];
            E.W1[(size_t)n * d + k] = (float)std::ldexp(deq[(size_t)n * d + k], -e1);
        }
        wsum[n] = (float)s;
        b1e[n] = (float)bb;
        E.b1[n] = b1e[n];
    }
    int rc;
    const _Float16* dW1 = nullptr;
    if ((rc = upload(p, packed, &dW1)) != PRAG_OK) return rc;
    L.W1f = reinterpret_cast<const u32x4*>(dW1);
    L.W1g = L.W2g = L.W2qg = nullptr;
    if (p->na == 1) {   // the same rounded weights in 16 x 16 fragment order (prober16.hip)
        std::vector<_Float16> p16;
        pack_fragments16(Wg, d, d / 32, [](int step, int q4, int j) { return 32 * step + 8 * q4 + j; }, p16);
        const _Float16* dW1g = nullptr;
        if ((rc = upload(p, p16, &dW1g)) != PRAG_OK) return rc;
        L.W1g = reinterpret_cast<const u32x4*>(dW1g);
    }
    if ((rc = upload(p, wsum, &L.wsum1)) != PRAG_OK) return rc;
    if ((rc = upload(p, b1e, &L.b1)) != PRAG_OK) return rc;
    L.sc1 = (float)std::ldexp(1.0, -e1);

    // ---- fc2: fold ln1 affine; k order = accumulator-register order of fc1 ----
    std::vector<double> W2g((size_t)H * H);
    mx = 0.0;
    for (int n = 0; n < H; ++n)
        for (int k = 0; k < H; ++k) {
            const double v = (double)W2[(size_t)n * H + k] * (double)ln1_w[k];
            W2g[(size_t)n * H + k] = v;
            mx = std::max(mx, std::fabs(v));
        }
    const int e2 = pick_scale_exp(mx);
    for (auto& v : W2g) v = std::ldexp(v, e2);
    pack_fragments(W2g, H, 32, p->na,
                   [](int step, int half, int j) {
                       // step = 2*row_tile + s ; element j of lane-half `half` holds hidden unit
                       // 32*row_tile + 16*s + 8*(j>>2) + 4*half + (j&3)  (32x32 C/D layout)
                       return 32 * (step >> 1) + 16 * (step & 1) + 8 * (j >> 2) + 4 * half + (j & 3);
                   },
                   packed, deq);
    std::vector<float> b2e(H), w2sum(H);
    E.W2.resize((size_t)H * H);
    E.b2.resize(H);
    for (int n = 0; n < H; ++n) {
        double bb = (double)b2[n], sm = 0.0;
        for (int k = 0; k < H; ++k) {
            bb += (double)W2[(size_t)n * H + k] * (double)ln1_b[k];
            sm += deq[(size_t)n * H + k];
            E.W2[(size_t)n * H + k] = (float)std::ldexp(deq[(size_t)n * H + k], -e2);
        }
        b2e[n] = (float)bb;
        w2sum[n] = (float)sm;
        E.b2[n] = b2e[n];
    }
    if ((rc = upload(p, w2sum, &L.w2sum)) != PRAG_OK) return rc;
    const _Float16* dW2 = nullptr;
    if ((rc = upload(p, packed, &dW2)) != PRAG_OK) return rc;
    L.W2f = reinterpret_cast<const u32x4*>(dW2);
    if ((rc = upload(p, b2e, &L.b2)) != PRAG_OK) return rc;
    L.sc2 = (float)std::ldexp(1.0, -e2);
    L.W2q = nullptr;
    if (p->na == 1) {
        std::vector<unsigned char> q8;
        pack_w2q(W2g, q8);
        const unsigned char* dq = nullptr;
        if ((rc = upload(p, q8, &dq)) != PRAG_OK) return rc;
        L.W2q = reinterpret_cast<const u32x4*>(dq);
        // 16 x 16 tiles: K-32 step ks = 2 w + p, element j of lane quarter q4 -> hidden unit
        // 64 w + 16 (2 p + (j >> 2)) + 4 q4 + (j & 3)
        std::vector<_Float16> p16;
        pack_fragments16(W2g, H, 16,
                         [](int step, int q4, int j) { return 64 * (step >> 1) + 16 * (2 * (step & 1) + (j >> 2)) + 4 * q4 + (j & 3); },
                         p16);
        const _Float16* dW2g = nullptr;
        if ((rc = upload(p, p16, &dW2g)) != PRAG_OK) return rc;
        L.W2g = reinterpret_cast<const u32x4*>(dW2g);
        std::vector<unsigned char> q16;
        pack_w2q16(W2g, q16);
        const unsigned char* dq16 = nullptr;
        if ((rc = upload(p, q16, &dq16)) != PRAG_OK) return rc;
        L.W2qg = reinterpret_cast<const u32x4*>(dq16);
    }

    // ---- fc3: fp32 VALU, fold ln2 affine --------------------------------------
    std::vector<float> W3e((size_t)kClasses * H);
    E.W3.resize((size_t)kClasses * H);
    E.b3.resize(kClasses);
    for (int c = 0; c < kClasses; ++c) {
        double bb = (double)b3[c], sm = 0.0;
        for (int k = 0; k < H; ++k) {
            W3e[(size_t)c * H + k] = (float)((double)W3[(size_t)c * H + k] * (double)ln2_w[k]);
            bb += (double)W3[(size_t)c * H + k] * (double)ln2_b[k];
            sm += (double)W3e[(size_t)c * H + k];
            E.W3[(size_t)c * H + k] = W3e[(size_t)c * H + k];
        }
        L.b3[c] = (float)bb;
        L.w3sum[c] = (float)sm;
        E.b3[c] = L.b3[c];
    }
    if ((rc = upload(p, W3e, &L.W3)) != PRAG_OK) return rc;

    // ---- small-batch path: the same effective weights as plain f32 rows ------------------------
    {
        SmallLayer& S = p->h_small[li];
        if ((rc = upload(p, E.W1, &S.W1)) != PRAG_OK) return rc;
        if ((rc = upload(p, E.W2, &S.W2)) != PRAG_OK) return rc;
        S.b1 = L.b1;
        S.b2 = L.b2;
        S.W3 = L.W3;
        S.b3[0] = L.b3[0];
        S.b3[1] = L.b3[1];
        PRAG_HIP(hipMemcpy(p->d_small + li, &S, sizeof(SmallLayer), hipMemcpyHostToDevice));
    }

    PRAG_HIP(hipMemcpy(p->d_layers + li, &L, sizeof(LayerDev), hipMemcpyHostToDevice));
    PRAG_HIP(hipDeviceSynchronize());
    p->loaded[li] = true;
    return PRAG_OK;
}

extern "C" int prag_prober_effective_weights(prag_prober_t* p, int li, float* W1, float* b1, float* W2,
                                             float* b2, float* W3, float* b3) {
    PRAG_REQUIRE(p != nullptr, PRAG_EINVAL, "prober handle is NULL");
    PRAG_REQUIRE(li >= 0 && li < p->n_layers, PRAG_EINVAL, "layer_idx=%d out of range", li);
    PRAG_REQUIRE(p->loaded[li], PRAG_ESTATE, "layer %d has no weights loaded", li);
    const HostLayer& E = p->eff[li];
    if (W1) memcpy(W1, E.W1.data(), E.W1.size() * sizeof(float));
    if (b1) memcpy(b1, E.b1.data(), E.b1.size() * sizeof(float));
    if (W2) memcpy(W2, E.W2.data(), E.W2.size() * sizeof(float));
    if (b2) memcpy(b2, E.b2.data(), E.b2.size() * sizeof(float));
    if (W3) memcpy(W3, E.W3.data(), E.W3.size() * sizeof(float));
    if (b3) memcpy(b3, E.b3.data(), E.b3.size() * sizeof(float));
    return PRAG_OK;
}

static int ensure_tile_cnt(prag_prober* p, int B);

extern "C" int prag_prober_reserve(prag_prober_t* p, int max_B) {
    PRAG_REQUIRE(p != nullptr, PRAG_EINVAL, "prober handle is NULL");
    PRAG_REQUIRE(max_B >= 1, PRAG_EINVAL, "max_B=%d", max_B);
    {
        const int rc_t = ensure_tile_cnt(p, max_B);      // tickets of the folded gate
        if (rc_t != PRAG_OK) return rc_t;
    }
    const int64_t rows = (int64_t)p->n_layers * max_B;
    if (rows <= p->ws_rows) return PRAG_OK;
    if (p->ws_h) (void)hipFree(p->ws_h);
    if (p->ws_l) (void)hipFree(p->ws_l);
    p->ws_h = p->ws_l = nullptr;
    p->ws_rows = 0;
    PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&p->ws_h), (size_t)rows * p->d * sizeof(_Float16)));
    PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&p->ws_l), (size_t)rows * p->d * sizeof(_Float16)));
    p->ws_rows = rows;
    return PRAG_OK;
}

// tickets for the folded gate: one word per 32-row tile of the largest batch seen (grown outside any capture: a
// first call of a larger batch allocates, like every workspace of the handle)
static int ensure_tile_cnt(prag_prober* p, int B) {
    const int need = (B + 31) / 32 + 8;
    if (need <= p->tile_cnt_cap) return PRAG_OK;
    if (p->tile_cnt) (void)hipFree(p->tile_cnt);
    p->tile_cnt = nullptr;
    p->tile_cnt_cap = 0;
    PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&p->tile_cnt), (size_t)need * sizeof(uint32_t)));
    PRAG_HIP(hipMemset(p->tile_cnt, 0, (size_t)need * sizeof(uint32_t)));
    p->tile_cnt_cap = need;
    return PRAG_OK;
}

template <int NA, int NB, int CT, int NWV>
static int launch_fused(const ProberArgs& a, int n_run, hipStream_t st, EventRing& prof) {
    constexpr int G = fc2_group<CT, NWV>();
    constexpr int ROWS = 32 * CT;
    constexpr int XSTAGE = NB * ROWS * 128;
    constexpr int EXCH = exch_bytes<NA, NWV, G>();
    constexpr int REGION_A = (kRing * XSTAGE > EXCH) ? kRing * XSTAGE : EXCH;
    constexpr int LDS = REGION_A + (2 * NWV * ROWS + 4 * NWV * 64 + 2 * ROWS + 6 * kHidden) * (int)sizeof(float);
    static_assert(LDS <= 160 * 1024, "workgroup LDS");
    auto kern = prober_fused_kernel<NA, NB, CT, NWV>;
    static LdsOptIn lds_opt_in;
    {
        const int rc_ = lds_opt_in.ensure(reinterpret_cast<const void*>(kern), LDS);
        if (rc_ != PRAG_OK) return rc_;
    }
    ProberArgs b = a;
#ifdef PRAG_MM_DIAG
    b.stamps = getenv("PRAG_PROBER_STAMPS") ? atoi(getenv("PRAG_PROBER_STAMPS")) : 0;
#endif
    b.n_tiles = (a.B + ROWS - 1) / ROWS;
    b.n_run = n_run;
    const int per_xcd = (b.n_tiles * n_run + 7) / 8;
    prof.begin(st);
    hipLaunchKernelGGL(kern, dim3(8 * per_xcd), dim3(64 * NWV), LDS, st, b);
    prof.end(st);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

static int pick_ct(int B, int n_run, int max_ct, int n_cu) {
    // smallest tile that still fills the chip: more workgroups pull weights in parallel
    int ct = 1;
    while (ct < max_ct && (int64_t)((B + 32 * ct - 1) / (32 * ct)) * n_run > n_cu) ct *= 2;
    return ct;
}

struct GateOut {   // fused gate of the small-batch path
    int ablation;
    double theta;
    float* probsum;
    int32_t* decision;
};

static int forward_impl(prag_prober_t* p, const void* x_dev, int x_dtype, int64_t x_layer_stride, int layer0,
                        int n_run, int B, float* logits_dev, void* stream, const GateOut* gate, bool* gate_done) {
    PRAG_REQUIRE(p != nullptr, PRAG_EINVAL, "prober handle is NULL");
    PRAG_REQUIRE(x_dev && logits_dev, PRAG_EINVAL, "prag_prober_forward: NULL device pointer");
    PRAG_REQUIRE(B >= 1, PRAG_EINVAL, "B=%d must be >= 1", B);
    PRAG_REQUIRE(layer0 >= 0 && n_run >= 1 && layer0 + n_run <= p->n_layers, PRAG_EINVAL,
                 "layers [%d,%d) outside [0,%d)", layer0, layer0 + n_run, p->n_layers);
    PRAG_REQUIRE(x_dtype == PRAG_F32 || x_dtype == PRAG_F16 || x_dtype == PRAG_BF16, PRAG_EINVAL, "x_dtype=%d",
                 x_dtype);
    PRAG_REQUIRE(n_run == 1 || x_layer_stride >= (int64_t)B * p->d, PRAG_EINVAL,
                 "x_layer_stride=%lld smaller than B*d_model", (long long)x_layer_stride);
    for (int l = layer0; l < layer0 + n_run; ++l)
        PRAG_REQUIRE(p->loaded[l], PRAG_ESTATE, "layer %d has no weights loaded", l);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

    // a handful of rows (the reference's call shape is ONE): weight rows spread over the whole chip
    // (measured: 28 us at one row against 64 us / 34 us for the tiled kernel with f32-parity / fp16
    // weights; the VALU dot products cross over at ~4 / ~2 rows)
    if (p->small_mode && small_supported(B, p->d) && B <= (p->na == 2 ? 4 : 2)) {
        SmallRun r;
        r.layers = p->d_small;
        r.x = x_dev;
        r.x_dtype = x_dtype;
        r.x_layer_stride = x_layer_stride;
        r.layer0 = layer0;
        r.n_run = n_run;
        r.B = B;
        r.d = p->d;
        r.h1 = p->small_ws;
        r.h2 = p->small_ws + (size_t)p->n_layers * kSmallMaxB * kHidden;
        r.logits = logits_dev;
        const bool fuse = gate && layer0 == 0 && n_run == p->n_layers;
        r.ablation = fuse ? gate->ablation : 0;
        r.theta = fuse ? gate->theta : 0.0;
        r.probsum = fuse ? gate->probsum : nullptr;
        r.decision = fuse ? gate->decision : nullptr;
        r.sync = p->small_sync;
        r.fused = p->small_mode == 3;
        p->prof.begin(st);
        const int rc = small_run(r, st);
        p->prof.end(st);
        if (rc == PRAG_OK && fuse && gate_done) *gate_done = true;
        return rc;
    }

    ProberArgs a{};
    a.layers = p->d_layers;
    a.layer0 = layer0;
    a.B = B;
    a.d = p->d;
    a.logits = logits_dev;
    int nb;
    if (x_dtype == PRAG_F16) {
        a.xh = reinterpret_cast<const _Float16*>(x_dev);
        a.xl = nullptr;
        a.x_layer_stride = x_layer_stride;
        nb = 1;
    } else {
        if ((int64_t)n_run * B > p->ws_rows) {
            int rc = prag_prober_reserve(p, B);
            if (rc != PRAG_OK) return rc;
        }
        const int rows = n_run * B;
        if (x_dtype == PRAG_F32)
            hipLaunchKernelGGL(prenorm_split_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, st,
                               reinterpret_cast<const float*>(x_dev), x_layer_stride, B, p->d, rows, p->ws_h, p->ws_l);
        else
            hipLaunchKernelGGL(prenorm_split_kernel<unsigned short>, dim3((rows + 3) / 4), dim3(256), 0, st,
                               reinterpret_cast<const unsigned short*>(x_dev), x_layer_stride, B, p->d, rows,
                               p->ws_h, p->ws_l);
        PRAG_LAUNCH_CHECK();
        a.xh = p->ws_h;
        a.xl = p->ws_l;
        a.x_layer_stride = (int64_t)B * p->d;
        nb = 2;
    }
    const int max_ct = (p->na == 1 && nb == 1) ? 4 : 2;
    int ct = pick_ct(B, n_run, max_ct, p->n_cu);
    const int ct_force = p->ct_force;
    if (ct_force == 1 || ct_force == 2 || (ct_force == 4 && max_ct >= 4)) ct = ct_force;
#define PRAG_DISPATCH(NA_, NB_)                                                      \
    if (p->na == NA_ && nb == NB_) {                                                 \
        if (ct == 1) return launch_fused<NA_, NB_, 1, 4>(a, n_run, st, p->prof);     \
        if (ct == 2) return launch_fused<NA_, NB_, 2, 8>(a, n_run, st, p->prof);     \
    }
    // (a 4-wave form of the 128-row tile - one wave per SIMD, 512 registers - was kept behind an environment knob
    // through round 3: slower, and 132 B of scratch per lane; removed)
    // fp16 weights x fp16 activations (the throughput mode): 16 x 16 MFMA tiles (prober16.hip) at every tile height;
    // PRAG_PROBER_SHAPE=32 keeps the 32 x 32 kernel for A/B timing
    if (p->na == 1 && nb == 1 && p->shape16) {
        // the gate folded into the launch when the whole ensemble runs (prag_gate): one launch less per decision batch
        if (gate && gate_done && p->gate_fold && layer0 == 0 && n_run == p->n_layers && gate->decision) {
            const int rc_t = ensure_tile_cnt(p, B);
            if (rc_t != PRAG_OK) return rc_t;
            a.probsum = gate->probsum;
            a.decision = gate->decision;
            a.tile_cnt = p->tile_cnt;
            a.ablation = gate->ablation;
            a.theta = gate->theta;
            const int rc_l = prober16_launch(a, n_run, 32 * ct, st, p->prof);
            if (rc_l == PRAG_OK) *gate_done = true;
            return rc_l;
        }
        return prober16_launch(a, n_run, 32 * ct, st, p->prof);
    }
    if (p->na == 1 && nb == 1 && ct == 4) return launch_fused<1, 1, 4, 8>(a, n_run, st, p->prof);
    PRAG_DISPATCH(1, 1)
    PRAG_DISPATCH(1, 2)
    PRAG_DISPATCH(2, 1)
    PRAG_DISPATCH(2, 2)
#undef PRAG_DISPATCH
    set_error("internal: no kernel for na=%d nb=%d ct=%d", p->na, nb, ct);
    return PRAG_EUNSUPPORTED;
}

// prag_gate's prober launch as data (tail_gate.h): only the throughput shape - fp16 activations, fp16 weights, the
// 16 x 16 kernel, more rows than the small-batch path takes
bool prag::prober_describe_tail(prag_prober* p, const void* x_dev, int x_dtype, int64_t x_layer_stride, int B, float* logits_dev,
                                int ablation, double theta, float* probsum_dev, int32_t* decision_dev, TailGate* out) {
    if (!p || !x_dev || !logits_dev || !out || B < 1 || x_dtype != PRAG_F16 || p->na != 1 || !p->shape16) return false;
    if (ablation < 0 || ablation > p->n_layers) return false;
    // (arguments prag_gate would refuse are refused by prag_gate: the fused launch never sees them)
    if (p->n_layers > 1 && x_layer_stride < (int64_t)B * p->d) return false;
    for (int l = 0; l < p->n_layers; ++l)
        if (!p->loaded[l]) return false;
    if (p->small_mode && small_supported(B, p->d) && B <= (p->na == 2 ? 4 : 2)) return false;
    int ct = pick_ct(B, p->n_layers, 4, p->n_cu);
    if (p->ct_force == 1 || p->ct_force == 2 || p->ct_force == 4) ct = p->ct_force;
    TailGate t;
    t.pa = ProberArgs{};
    t.pa.layers = p->d_layers;
    t.pa.xh = reinterpret_cast<const _Float16*>(x_dev);
    t.pa.xl = nullptr;
    t.pa.x_layer_stride = x_layer_stride;
    t.pa.layer0 = 0;
    t.pa.B = B;
    t.pa.d = p->d;
    t.pa.logits = logits_dev;
    t.ct16 = 2 * ct;
    t.pa.n_tiles = (B + 16 * t.ct16 - 1) / (16 * t.ct16);
    t.pa.n_run = p->n_layers;
    t.n_wg = 8 * ((t.pa.n_tiles * p->n_layers + 7) / 8);
    t.lds_bytes = prober16_lds_bytes(t.ct16);
    t.taken = false;
    t.gate_folded = false;
    t.fin_probsum = probsum_dev;
    t.fin_decision = decision_dev;
    t.fin_ablation = ablation;
    t.fin_theta = theta;
    if (p->gate_fold && decision_dev && ensure_tile_cnt(p, B) == PRAG_OK) {     // the gate rides along too
        t.pa.probsum = probsum_dev;
        t.pa.decision = decision_dev;
        t.pa.tile_cnt = p->tile_cnt;
        t.pa.ablation = ablation;
        t.pa.theta = theta;
        t.gate_folded = true;
    }
    *out = t;
    return t.lds_bytes > 0;
}

extern "C" int prag_prober_forward(prag_prober_t* p, const void* x_dev, int x_dtype,
                                   int64_t x_layer_stride, int layer0, int n_run, int B,
                                   float* logits_dev, void* stream) {
    return forward_impl(p, x_dev, x_dtype, x_layer_stride, layer0, n_run, B, logits_dev, stream, nullptr, nullptr);
}

extern "C" int prag_gate_from_logits(const float* logits_dev, int L, int B, int ablation, double theta,
                                     float* probsum_dev, int32_t* decision_dev, void* stream) {
    PRAG_REQUIRE(logits_dev != nullptr, PRAG_EINVAL, "logits_dev is NULL");
    PRAG_REQUIRE(L >= 1 && B >= 1, PRAG_EINVAL, "L=%d B=%d", L, B);
    PRAG_REQUIRE(ablation >= 0 && ablation <= L, PRAG_EINVAL, "ablation=%d outside [0,%d]", ablation, L);
    hipLaunchKernelGGL(gate_kernel, dim3((B + 255) / 256), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), logits_dev, L, B, ablation, theta,
                       probsum_dev, decision_dev);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

extern "C" int prag_gate(prag_prober_t* p, const void* x_dev, int x_dtype, int64_t x_layer_stride, int B,
                         int ablation, double theta, float* logits_dev, float* probsum_dev,
                         int32_t* decision_dev, void* stream) {
    PRAG_REQUIRE(p != nullptr, PRAG_EINVAL, "prober handle is NULL");
    PRAG_REQUIRE(ablation >= 0 && ablation <= p->n_layers, PRAG_EINVAL, "ablation=%d outside [0,%d]", ablation,
                 p->n_layers);
    const GateOut g{ablation, theta, probsum_dev, decision_dev};
    bool done = false;
    int rc = forward_impl(p, x_dev, x_dtype, x_layer_stride, 0, p->n_layers, B, logits_dev, stream, &g, &done);
    if (rc != PRAG_OK || done) return rc;
    return prag_gate_from_logits(logits_dev, p->n_layers, B, ablation, theta, probsum_dev, decision_dev,
                                 stream);
}

// The gate as the retrieve-decide loop consumes it (exp_rag.py:393, 406-415: `logits = return_prober_logit_gemma_2b(...)`,
// softmax / sum / threshold in Python, then `if logit_sum[0] + theta < logit_sum[1]` on the HOST): one call that ends
// with the decisions in the caller's host memory.  Everything the call needs lives in the handle (device logits, a
// pinned + mapped result block), so the host side allocates nothing per decision; for <= kDecideDirect rows the gate's
// last kernel writes the decisions into the mapped block itself and the host waits for THOSE WORDS (a sentinel it put
// there first) instead of for an interrupt - the stream is not otherwise synchronised.  probsum_host != NULL also
// returns the two sums the reference prints (exp_rag.py:420): that variant waits for the stream.
static int decide_reserve(prag_prober* p, int B) {
    if (B <= p->dec_cap) return PRAG_OK;
    const int cap = std::max(B, 64);
    if (p->dec_logits) (void)hipFree(p->dec_logits);
    if (p->dec_host) (void)hipHostFree(p->dec_host);
    if (p->dec_dev) (void)hipFree(p->dec_dev);
    p->dec_logits = nullptr; p->dec_host = nullptr; p->dec_host_dev = nullptr; p->dec_dev = nullptr; p->dec_cap = 0;
    PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&p->dec_logits), (size_t)p->n_layers * cap * 2 * sizeof(float)));
    PRAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->dec_host), (size_t)cap * 12,
                           hipHostMallocMapped | hipHostMallocCoherent));
    PRAG_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->dec_host_dev), p->dec_host, 0));
    if (cap > kDecideDirect) PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&p->dec_dev), (size_t)cap * 12));
    p->dec_cap = cap;
    return PRAG_OK;
}

extern "C" int prag_gate_decide(prag_prober_t* p, const void* x_dev, int x_dtype, int64_t x_layer_stride, int B,
                                int ablation, double theta, int32_t* decision_host, float* probsum_host, void* stream) {
    PRAG_REQUIRE(p != nullptr, PRAG_EINVAL, "prober handle is NULL");
    PRAG_REQUIRE(B >= 0 && (B == 0 || decision_host != nullptr), PRAG_EINVAL, "prag_gate_decide: B=%d, decision_host NULL", B);
    if (B == 0) return PRAG_OK;
    int rc = decide_reserve(p, B);
    if (rc != PRAG_OK) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool direct = B <= kDecideDirect;
    volatile int32_t* dec_h = reinterpret_cast<volatile int32_t*>(p->dec_host);
    float* ps_h = reinterpret_cast<float*>(p->dec_host + (size_t)p->dec_cap * 4);
    char* base = direct ? p->dec_host_dev : p->dec_dev;
    int32_t* dec_d = reinterpret_cast<int32_t*>(base);
    float* ps_d = reinterpret_cast<float*>(base + (size_t)p->dec_cap * 4);
    const bool spin = direct && p->dec_spin && probsum_host == nullptr;
    if (spin)
        for (int b = 0; b < B; ++b) dec_h[b] = -1;           // no decision is -1: the kernel's store replaces it
    rc = prag_gate(p, x_dev, x_dtype, x_layer_stride, B, ablation, theta, p->dec_logits, ps_d, dec_d, stream);
    if (rc != PRAG_OK) return rc;
    if (!direct) PRAG_HIP(hipMemcpyAsync(p->dec_host, p->dec_dev, (size_t)p->dec_cap * 12, hipMemcpyDeviceToHost, st));
    bool done = false;
    if (spin) {
        // ~200 us of polling (a B = 1 gate is ~20 us of kernels), then the ordinary wait
        for (int it = 0; it < 200000 && !done; ++it) {
            done = true;
            for (int b = 0; b < B; ++b)
                if (dec_h[b] == -1) { done = false; break; }
            if (!done) __builtin_ia32_pause();
        }
    }
    if (!done) PRAG_HIP(hipStreamSynchronize(st));
    for (int b = 0; b < B; ++b) decision_host[b] = dec_h[b];
    if (probsum_host) memcpy(probsum_host, ps_h, (size_t)B * 2 * sizeof(float));
    return PRAG_OK;
}

// One decode step of the hooked layers AND the gate on the sums so far, in ONE launch.  The reference's hooks copy every
// layer's activations to the host on every forward pass (exp_rag.py:317-329) and the gate - cat / sum / six probers /
// softmax / threshold, exp_rag.py:381-389, 406-415 - starts when `generate` has returned, on a host whose caches the LM's
// 140 ms of Python have emptied: measured 87-135 us per prag_gate_decide call inside the loop against 30 back to back
// (profiles/r06b_gate_in_loop.txt; the kernels themselves take 25 us there, the rest is the cold host path of three
// launches and a poll).  Here the launch that adds a decode step's activations to the running sums (the pool's one
// launch per token since round 4) also runs the gate on the sums it has just formed and leaves the decision in pinned
// host memory: when `generate` returns, the decision of its last step is already on the host.
//   acc_in / acc_out  float32 [L][B][d_model], DIFFERENT buffers (ping-pong; acc_in is not read when assign != 0)
//   h_dev_ptrs        host array of L device pointers, [B][d_model] of h_dtype each
//   *tag_out          identifies this step for prag_gate_step_result
// PRAG_EUNSUPPORTED outside the small-batch envelope (B rows the small path serves, d_model <= 4096, L <= 16): the
// caller then adds with prag_pool_accumulate_layers and decides with prag_gate_decide.
extern "C" int prag_pool_step_gate(prag_prober_t* p, const float* acc_in_dev, float* acc_out_dev,
                                   const void* const* h_dev_ptrs, int h_dtype, int B, int assign, int ablation,
                                   double theta, uint64_t* tag_out, void* stream) {
    PRAG_REQUIRE(p != nullptr, PRAG_EINVAL, "prober handle is NULL");
    PRAG_REQUIRE(acc_out_dev && h_dev_ptrs && tag_out && (assign || acc_in_dev), PRAG_EINVAL,
                 "prag_pool_step_gate: NULL pointer");
    PRAG_REQUIRE(acc_in_dev != acc_out_dev, PRAG_EINVAL, "prag_pool_step_gate: acc_in and acc_out must be different buffers");
    PRAG_REQUIRE(h_dtype == PRAG_F32 || h_dtype == PRAG_F16 || h_dtype == PRAG_BF16, PRAG_EINVAL, "h_dtype=%d", h_dtype);
    PRAG_REQUIRE(ablation >= 0 && ablation <= p->n_layers, PRAG_EINVAL, "ablation=%d outside [0,%d]", ablation, p->n_layers);
    PRAG_REQUIRE(B >= 1, PRAG_EINVAL, "B=%d must be >= 1", B);
    for (int l = 0; l < p->n_layers; ++l) {
        PRAG_REQUIRE(p->loaded[l], PRAG_ESTATE, "layer %d has no weights loaded", l);
        PRAG_REQUIRE(h_dev_ptrs[l] != nullptr, PRAG_EINVAL, "prag_pool_step_gate: layer %d pointer is NULL", l);
    }
    // the rows prag_gate would hand to the small-batch kernels, so that the step's decision IS prag_gate_decide's
    PRAG_REQUIRE(p->small_mode && small_supported(B, p->d) && B <= (p->na == 2 ? 4 : 2) && p->d <= 4096 &&
                     p->n_layers <= kSmallStepMaxLayers && p->small_sync != nullptr,
                 PRAG_EUNSUPPORTED, "prag_pool_step_gate: B=%d d_model=%d layers=%d is outside the small-batch gate", B, p->d,
                 p->n_layers);
    int rc = decide_reserve(p, B);
    if (rc != PRAG_OK) return rc;
    if (!p->step_host) {
        PRAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->step_host), 128, hipHostMallocMapped | hipHostMallocCoherent));
        memset(p->step_host, 0, 128);
        PRAG_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->step_host_dev), p->step_host, 0));
    }
    SmallRun r;
    r.layers = p->d_small;
    r.x = nullptr;
    r.x_dtype = h_dtype;
    r.x_layer_stride = 0;
    r.layer0 = 0;
    r.n_run = p->n_layers;
    r.B = B;
    r.d = p->d;
    r.h1 = p->small_ws;
    r.h2 = p->small_ws + (size_t)p->n_layers * kSmallMaxB * kHidden;
    r.logits = p->dec_logits;
    r.ablation = ablation;
    r.theta = theta;
    r.probsum = nullptr;
    r.decision = nullptr;
    r.sync = p->small_sync;
    r.fused = 1;
    SmallStep sp;
    sp.h = h_dev_ptrs;
    sp.acc_in = acc_in_dev;
    sp.acc_out = acc_out_dev;
    sp.assign = assign;
    sp.tag = ++p->step_tag;
    sp.host_dev = p->step_host_dev;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    p->prof.begin(st);
    rc = small_step_run(r, sp, st);
    p->prof.end(st);
    if (rc != PRAG_OK) return rc;
    *tag_out = sp.tag;
    return PRAG_OK;
}

// The decision the step tagged `tag` left on the host (exp_rag.py:393, 406-415's `if`): decision_host int32 [B],
// probsum_host float32 [B,2] or NULL.  Only the MOST RECENT step of a handle can be asked for (PRAG_ESTATE otherwise:
// a later step has overwritten the block).  Polls the tag word for ~200 us, then waits for `stream` (the stream the
// step was enqueued on).  PRAG_ESTATE as well when a hand-off inside that launch timed out (another kernel held the
// chip for ~50 ms): the sums are fine - the caller falls back to prag_gate_decide on them.
extern "C" int prag_gate_step_result(prag_prober_t* p, uint64_t tag, int B, int32_t* decision_host, float* probsum_host,
                                     void* stream) {
    PRAG_REQUIRE(p != nullptr && decision_host != nullptr, PRAG_EINVAL, "prag_gate_step_result: NULL pointer");
    PRAG_REQUIRE(B >= 1 && B <= kSmallMaxB, PRAG_EINVAL, "B=%d outside [1,%d]", B, kSmallMaxB);
    PRAG_REQUIRE(p->step_host != nullptr && tag != 0 && tag == p->step_tag, PRAG_ESTATE,
                 "prag_gate_step_result: tag %llu is not the handle's most recent step (%llu)", (unsigned long long)tag,
                 (unsigned long long)p->step_tag);
    volatile uint64_t* tw = p->step_host;
    constexpr uint64_t kVoid = 1ull << 63;
    bool done = false;
    for (int it = 0; it < 200000 && !done; ++it) {
        done = (__atomic_load_n(tw, __ATOMIC_ACQUIRE) & ~kVoid) == tag;
        if (!done) __builtin_ia32_pause();
    }
    if (!done) {
        PRAG_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
        done = (__atomic_load_n(tw, __ATOMIC_ACQUIRE) & ~kVoid) == tag;
    }
    PRAG_REQUIRE(done, PRAG_ESTATE, "prag_gate_step_result: the step's launch finished without publishing tag %llu",
                 (unsigned long long)tag);
    PRAG_REQUIRE((*tw & kVoid) == 0, PRAG_ESTATE, "prag_gate_step_result: a hand-off inside the step's launch timed out");
    const int32_t* hd = reinterpret_cast<const int32_t*>(p->step_host + 1);
    const float* hp = reinterpret_cast<const float*>(hd + kSmallMaxB);
    for (int b = 0; b < B; ++b) decision_host[b] = hd[b];
    if (probsum_host) memcpy(probsum_host, hp, (size_t)B * 2 * sizeof(float));
    return PRAG_OK;
}

extern "C" int prag_prober_profile(prag_prober_t* p, int slots) {
    PRAG_REQUIRE(p != nullptr && slots >= 0 && slots <= 4096, PRAG_EINVAL, "prag_prober_profile: bad argument");
    if (slots == 0) {
        p->prof.disable();
        return PRAG_OK;
    }
    return p->prof.enable(slots);
}

extern "C" int prag_prober_profile_read(prag_prober_t* p, float* ms, int cap, int* n_out) {
    PRAG_REQUIRE(p != nullptr && ms != nullptr && cap >= 0, PRAG_EINVAL, "prag_prober_profile_read: bad argument");
    return p->prof.read(ms, cap, n_out);
}

extern "C" void prag_prober_destroy(prag_prober_t* p) {
    if (!p) return;
    p->prof.disable();
    for (auto& layer : p->allocs)
        for (void* q : layer) (void)hipFree(q);
    if (p->d_layers) (void)hipFree(p->d_layers);
    if (p->d_small) (void)hipFree(p->d_small);
    if (p->small_ws) (void)hipFree(p->small_ws);
    if (p->small_sync) (void)hipFree(p->small_sync);
    if (p->ws_h) (void)hipFree(p->ws_h);
    if (p->ws_l) (void)hipFree(p->ws_l);
    if (p->tile_cnt) (void)hipFree(p->tile_cnt);
    if (p->dec_logits) (void)hipFree(p->dec_logits);
    if (p->dec_host) (void)hipHostFree(p->dec_host);
    if (p->dec_dev) (void)hipFree(p->dec_dev);
    if (p->step_host) (void)hipHostFree(p->step_host);
    delete p;
}

extern "C" int prag_pool_accumulate(float* acc_dev, const void* h_dev, int h_dtype, int64_t n_elems,
                                    int assign, void* stream) {
    PRAG_REQUIRE(acc_dev && h_dev, PRAG_EINVAL, "prag_pool_accumulate: NULL device pointer");
    PRAG_REQUIRE(n_elems >= 0 && n_elems % 4 == 0, PRAG_EINVAL, "n_elems=%lld must be a multiple of 4",
                 (long long)n_elems);
    PRAG_REQUIRE(h_dtype == PRAG_F32 || h_dtype == PRAG_F16 || h_dtype == PRAG_BF16, PRAG_EINVAL, "h_dtype=%d",
                 h_dtype);
    if (n_elems == 0) return PRAG_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int blocks = (int)std::min<int64_t>((n_elems / 4 + 255) / 256, 2048);
    if (h_dtype == PRAG_F32)
        hipLaunchKernelGGL(pool_accumulate_kernel<float>, dim3(blocks), dim3(256), 0, st, acc_dev,
                           reinterpret_cast<const float*>(h_dev), n_elems, assign);
    else if (h_dtype == PRAG_F16)
        hipLaunchKernelGGL(pool_accumulate_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, acc_dev,
                           reinterpret_cast<const _Float16*>(h_dev), n_elems, assign);
    else
        hipLaunchKernelGGL(pool_accumulate_kernel<unsigned short>, dim3(blocks), dim3(256), 0, st, acc_dev,
                           reinterpret_cast<const unsigned short*>(h_dev), n_elems, assign);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

extern "C" int prag_pool_accumulate_layers(float* acc_dev, const void* const* h_dev_ptrs, int n_layers, int h_dtype,
                                           int64_t n_elems, int assign, void* stream) {
    PRAG_REQUIRE(acc_dev && h_dev_ptrs, PRAG_EINVAL, "prag_pool_accumulate_layers: NULL pointer");
    PRAG_REQUIRE(n_layers >= 1 && n_layers <= 64, PRAG_EINVAL, "n_layers=%d outside [1,64]", n_layers);
    PRAG_REQUIRE(n_elems >= 0 && n_elems % 4 == 0, PRAG_EINVAL, "n_elems=%lld must be a multiple of 4",
                 (long long)n_elems);
    PRAG_REQUIRE(h_dtype == PRAG_F32 || h_dtype == PRAG_F16 || h_dtype == PRAG_BF16, PRAG_EINVAL, "h_dtype=%d",
                 h_dtype);
    PoolLayerPtrs ptrs{};
    for (int l = 0; l < n_layers; ++l) {
        PRAG_REQUIRE(h_dev_ptrs[l] != nullptr, PRAG_EINVAL, "prag_pool_accumulate_layers: layer %d pointer is NULL", l);
        ptrs.h[l] = h_dev_ptrs[l];
    }
    if (n_elems == 0) return PRAG_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)std::min<int64_t>((n_elems / 4 + 255) / 256, 512), (unsigned)n_layers);
    if (h_dtype == PRAG_F32)
        hipLaunchKernelGGL(pool_accumulate_layers_kernel<float>, grid, dim3(256), 0, st, acc_dev, ptrs, n_elems, assign);
    else if (h_dtype == PRAG_F16)
        hipLaunchKernelGGL(pool_accumulate_layers_kernel<_Float16>, grid, dim3(256), 0, st, acc_dev, ptrs, n_elems, assign);
    else
        hipLaunchKernelGGL(pool_accumulate_layers_kernel<unsigned short>, grid, dim3(256), 0, st, acc_dev, ptrs, n_elems,
                           assign);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

extern "C" int prag_pool_masked_mean(const void* hidden_dev, int dtype, const int64_t* mask_dev, int B, int T, int d,
                                     float* out_dev, void* stream) {
    PRAG_REQUIRE(hidden_dev && mask_dev && out_dev, PRAG_EINVAL, "prag_pool_masked_mean: NULL device pointer");
    PRAG_REQUIRE(B >= 1 && T >= 1 && d >= 4 && d % 4 == 0, PRAG_EINVAL, "B=%d T=%d d=%d", B, T, d);
    PRAG_REQUIRE(dtype == PRAG_F32 || dtype == PRAG_F16 || dtype == PRAG_BF16, PRAG_EINVAL, "dtype=%d", dtype);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((d / 4 + 255) / 256, B);
    if (dtype == PRAG_F32)
        hipLaunchKernelGGL(pool_masked_mean_kernel<float>, grid, dim3(256), 0, st,
                           reinterpret_cast<const float*>(hidden_dev), mask_dev, T, d, out_dev);
    else if (dtype == PRAG_F16)
        hipLaunchKernelGGL(pool_masked_mean_kernel<_Float16>, grid, dim3(256), 0, st,
                           reinterpret_cast<const _Float16*>(hidden_dev), mask_dev, T, d, out_dev);
    else
        hipLaunchKernelGGL(pool_masked_mean_kernel<unsigned short>, grid, dim3(256), 0, st,
                           reinterpret_cast<const unsigned short*>(hidden_dev), mask_dev, T, d, out_dev);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

extern "C" int prag_pool_ragged(const void* acts_dev, int dtype, int B, int T, int d,
                                const int64_t* pred_lens_dev, int scale_mean, float* out_dev,
                                void* stream) {
    PRAG_REQUIRE(acts_dev && pred_lens_dev && out_dev, PRAG_EINVAL, "prag_pool_ragged: NULL device pointer");
    PRAG_REQUIRE(B >= 1 && T >= 1 && d >= 4 && d % 4 == 0, PRAG_EINVAL, "B=%d T=%d d=%d", B, T, d);
    PRAG_REQUIRE(dtype == PRAG_F32 || dtype == PRAG_F16 || dtype == PRAG_BF16, PRAG_EINVAL, "dtype=%d", dtype);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((d / 4 + 255) / 256, B);
    if (dtype == PRAG_F32)
        hipLaunchKernelGGL(pool_ragged_kernel<float>, grid, dim3(256), 0, st,
                           reinterpret_cast<const float*>(acts_dev), T, d, pred_lens_dev, scale_mean, out_dev);
    else if (dtype == PRAG_F16)
        hipLaunchKernelGGL(pool_ragged_kernel<_Float16>, grid, dim3(256), 0, st,
                           reinterpret_cast<const _Float16*>(acts_dev), T, d, pred_lens_dev, scale_mean,
                           out_dev);
    else
        hipLaunchKernelGGL(pool_ragged_kernel<unsigned short>, grid, dim3(256), 0, st,
                           reinterpret_cast<const unsigned short*>(acts_dev), T, d, pred_lens_dev, scale_mean,
                           out_dev);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

extern "C" int prag_pool_each_token(const void* acts_dev, int dtype, int B, int T, int d, const int64_t* row_offsets_dev,
                                    int64_t n_rows, const int32_t* labels_dev, float* out_dev, int32_t* labels_out_dev,
                                    void* stream) {
    PRAG_REQUIRE(acts_dev && row_offsets_dev && out_dev, PRAG_EINVAL, "prag_pool_each_token: NULL device pointer");
    PRAG_REQUIRE(B >= 1 && T >= 1 && d >= 4 && d % 4 == 0, PRAG_EINVAL, "B=%d T=%d d=%d", B, T, d);
    PRAG_REQUIRE(n_rows >= 0 && n_rows <= (int64_t)B * T && n_rows <= 0x7fffffff, PRAG_EINVAL,
                 "n_rows=%lld outside [0, B*T]", (long long)n_rows);
    PRAG_REQUIRE(dtype == PRAG_F32 || dtype == PRAG_F16 || dtype == PRAG_BF16, PRAG_EINVAL, "dtype=%d", dtype);
    if (n_rows == 0) return PRAG_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)n_rows, (d / 4 + 255) / 256);
    if (dtype == PRAG_F32)
        hipLaunchKernelGGL(pool_each_token_kernel<float>, grid, dim3(256), 0, st, reinterpret_cast<const float*>(acts_dev),
                           B, T, d, row_offsets_dev, labels_dev, out_dev, labels_out_dev);
    else if (dtype == PRAG_F16)
        hipLaunchKernelGGL(pool_each_token_kernel<_Float16>, grid, dim3(256), 0, st,
                           reinterpret_cast<const _Float16*>(acts_dev), B, T, d, row_offsets_dev, labels_dev, out_dev,
                           labels_out_dev);
    else
        hipLaunchKernelGGL(pool_each_token_kernel<unsigned short>, grid, dim3(256), 0, st,
                           reinterpret_cast<const unsigned short*>(acts_dev), B, T, d, row_offsets_dev, labels_dev,
                           out_dev, labels_out_dev);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}
