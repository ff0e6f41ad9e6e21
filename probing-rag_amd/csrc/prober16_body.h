// prober16_kernel's body as a device function (round 5): one workgroup of the fused prober ensemble on 16 x 16 MFMA tiles
// (see prober16.hip for the geometry).  It lives in a header because TWO kernels run it: prober16_kernel (prober16.hip)
// and bound_gate_kernel (flat_shadow.hip), which carries the gate's workgroups in the SAME launch as the two-level
// search's bound kernel - the gate of the next batch depends on nothing in the search, and the search's tail leaves 3/4
// of the chip idle.  `block_idx` replaces blockIdx.x; `smem` is the workgroup's dynamic LDS (p16_lds_bytes<CT16>() bytes,
// 16-byte aligned).  The includer defines PSTAMP(i) and P16_ABL(bit) first (timing stamps / ablations of the `make diag`
// build; empty and 0 otherwise).
#pragma once

#include <hip/hip_runtime.h>

#include "prag_common.h"
#include "prober_internal.h"

namespace prag {

template <int CT16>
constexpr int p16_lds_bytes() {
    constexpr int ROWS = 16 * CT16, GC = CT16 < 4 ? CT16 : 4;
    constexpr int XSTAGE = ROWS * 128, EXCH = 24 * GC * 1024;
    constexpr int REGION_A = (kRing * XSTAGE > EXCH) ? kRing * XSTAGE : EXCH;
    return REGION_A + (2 * 8 * ROWS + 4 * 8 * 64 + 2 * ROWS + 6 * kHidden) * (int)sizeof(float);
}

template <int CT16>
__device__ __forceinline__ void prober16_body(const ProberArgs& a, char* const smem, const int block_idx) {
    constexpr int NT = 512, NWV = 8, HT = 4;
    constexpr int ROWS = 16 * CT16;
    constexpr int GC = CT16 < 4 ? CT16 : 4;        // column tiles per fc2 pass and per fc1 fragment group
    constexpr int NG = CT16 / GC;                  // fc2 passes / fragment groups per K-32 step
    constexpr int XSTAGE = ROWS * 128;             // bytes of one staged [ROWS x 64] tile
    constexpr int EXCH = 24 * GC * 1024;           // hi fragments [16 ks][GC] + fp8 lo operands [4 kb][GC][2] KiB
    constexpr int REGION_A = (kRing * XSTAGE > EXCH) ? kRing * XSTAGE : EXCH;
    constexpr int NPIECE = 8 * ROWS;               // 16-B pieces of a staged tile
    constexpr int NPASS = (NPIECE + NT - 1) / NT;

    char* s_x = smem;                                          // staging ring (fc1)
    char* s_ex = smem;                                         // exchange fragments (fc2), same bytes
    float* s_red = reinterpret_cast<float*>(smem + REGION_A);  // [2][NWV][ROWS] LN1 sums, then [4][NWV][64]
    float* s_mu0 = s_red + 2 * NWV * ROWS + 4 * NWV * 64;      // [ROWS]
    float* s_rs0 = s_mu0 + ROWS;                               // [ROWS]
    float* s_cst = s_rs0 + ROWS;  // [6][512]: wsum1, b1, b2, W3[0], W3[1], w2sum (epilogue constants)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    // XCD-aware placement (speed only): blocks b and b+8 share an XCD - contiguous chunks of the layer-major work list
    const int per_xcd = (a.n_tiles * a.n_run + 7) / 8;
    const int vidx = (block_idx & 7) * per_xcd + (block_idx >> 3);
    if (vidx >= a.n_tiles * a.n_run) return;
    const int lrun = vidx / a.n_tiles;
    const LayerDev& L = a.layers[a.layer0 + lrun];
    const int tile_idx = vidx - lrun * a.n_tiles;
    const int m0 = tile_idx * ROWS;
    const __amdgpu_buffer_rsrc_t rs_w1 = make_rsrc(uniform_p(L.W1g), (size_t)a.d * 1024);
    const __amdgpu_buffer_rsrc_t rs_w2 = make_rsrc(uniform_p(L.W2g), (size_t)16 * 32 * 1024);
    const __amdgpu_buffer_rsrc_t rs_w2q = make_rsrc(uniform_p(L.W2qg), (size_t)4 * 32 * 2048);
    const float L_sc1 = uniform_f(L.sc1), L_sc2 = uniform_f(L.sc2);
    const float L_b3[2] = {uniform_f(L.b3[0]), uniform_f(L.b3[1])};
    const float L_w3sum[2] = {uniform_f(L.w3sum[0]), uniform_f(L.w3sum[1])};
    const int d = a.d;
    const int T = d >> 6;
#ifdef PRAG_MM_DIAG
    const int ps_sel = !a.stamps ? -1 : vidx == 0 ? 0 : vidx == 17 ? 1 : vidx == a.n_tiles * a.n_run - 1 ? 2 : -1;
#endif
    PSTAMP(0)

    // epilogue constants -> LDS once (loads issued behind the first activation tile's, stored after the weight prologue)
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

    // ---- activation staging: thread -> 16-B piece(s) of the [ROWS x 64] tile (32-row tiles: waves 4-7 stage nothing)
    __amdgpu_buffer_rsrc_t rs_x;
    {
        const int64_t tile0 = (int64_t)lrun * a.x_layer_stride + (int64_t)m0 * d;
        rs_x = make_rsrc(uniform_p(a.xh + tile0), (size_t)(a.B - m0) * d * 2);
    }
    unsigned x_off[NPASS];
    int st_off[NPASS];
    bool st_on[NPASS];
#pragma unroll
    for (int c = 0; c < NPASS; ++c) {
        const int e = tid + c * NT;
        st_on[c] = e < NPIECE;
        const int ec = st_on[c] ? e : NPIECE - 1;
        const int row = ec >> 3, q = ec & 7;
        const int lrow = m0 + row < a.B ? row : a.B - 1 - m0;   // rows past the batch re-read its last row
        x_off[c] = (unsigned)(lrow * d + 8 * q) * 2u;
        st_off[c] = row * 128 + ((q ^ ((row >> 1) & 7)) << 4);
    }
    // B fragment of column tile ct at K-32 sub-step `sub`: row 16 ct + c16, piece 4 sub + q4; (row >> 1) & 7 does not
    // depend on ct, so one lane offset per sub-step and ct as an immediate
    int rd_off[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) rd_off[sub] = c16 * 128 + (((4 * sub + q4) ^ ((c16 >> 1) & 7)) << 4);

    u32x4 xreg[NPASS];
    auto x_load_r = [&](u32x4 (&xr)[NPASS], int t) {
#pragma unroll
        for (int c = 0; c < NPASS; ++c) xr[c] = buf_load16(rs_x, x_off[c], (unsigned)t * 128u);
    };
    auto x_store_r = [&](u32x4 (&xr)[NPASS], int stage) {
#pragma unroll
        for (int c = 0; c < NPASS; ++c)
            if (st_on[c]) *reinterpret_cast<u32x4*>(s_x + stage * XSTAGE + st_off[c]) = xr[c];
    };
    // weight fragments: (K-32 step s32, hidden tile 4 w + ht) -> 1 KiB, straight to registers
    const unsigned lane16 = lane * 16;
    half8 afr[2][HT];
    auto a_load = [&](int slot, int s32) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            const u32x4 v = buf_load16(rs_w1, lane16, ((unsigned)s32 * 32u + 4u * w + ht) << 10);
            afr[slot][ht] = __builtin_bit_cast(half8, v);
        }
    };

    f32x4 acc[HT][CT16];
#pragma unroll
    for (int i = 0; i < HT; ++i)
#pragma unroll
        for (int c = 0; c < CT16; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LayerNorm-0 statistics in flight: every staging thread sums its own 16-B pieces (exact-product dot2, f32 sums)
    float st_s[NPASS], st_q2[NPASS];
#pragma unroll
    for (int c = 0; c < NPASS; ++c) st_s[c] = st_q2[c] = 0.f;
    const half2_t kOnes2 = {(_Float16)1.f, (_Float16)1.f};
    auto x_stats_r = [&](u32x4 (&xr)[NPASS]) {
#pragma unroll
        for (int c = 0; c < NPASS; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned int u = xr[c][j];
                const half2_t xv = __builtin_bit_cast(half2_t, u);
                st_s[c] = __builtin_amdgcn_fdot2(xv, kOnes2, st_s[c], false);
                st_q2[c] = __builtin_amdgcn_fdot2(xv, xv, st_q2[c], false);
            }
    };

    // ---- prologue: every load is issued before the first one is waited for ---------------------------------------
    {
        u32x4 xpro[kAhead][NPASS];
#pragma unroll
        for (int i = 0; i < kAhead; ++i) x_load_r(xpro[i], i < T ? i : T - 1);
        __builtin_amdgcn_sched_barrier(0);
        cst_load();
        __builtin_amdgcn_sched_barrier(0);
        x_load_r(xreg, kAhead < T ? kAhead : T - 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            __builtin_amdgcn_sched_barrier(0);  // pin the issue order: the loop's counted waits hold on entry too
            a_load(s, s);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < kAhead; ++i) {
            if (i < T) x_stats_r(xpro[i]);
            x_store_r(xpro[i], i);
        }
        cst_store();
        __builtin_amdgcn_sched_barrier(0);
    }

    PSTAMP(1)
    // ---- fc1 main loop: one barrier per kAhead 64-wide K steps ------------------------------------------------------
    for (int t = 0; t < T; ++t) {
        if ((t & (kAhead - 1)) == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (t + kAhead < T) x_stats_r(xreg);                   // tile t+kAhead; the tail's re-reads do not count
        x_store_r(xreg, (t + kAhead) & (kRing - 1));           // tile t+kAhead (loaded one step ago)
        x_load_r(xreg, t + kAhead + 1 < T ? t + kAhead + 1 : T - 1);
        const char* xs = s_x + (t & (kRing - 1)) * XSTAGE;
        const int s32n = 2 * (t + 1 < T ? t + 1 : T - 1);
        half8 bfr[2][GC];
        auto b_read = [&](int buf, int hs) {   // half sub-step hs = sub * NG + grp
            const int sub = hs / NG, grp = hs % NG;
#pragma unroll
            for (int c = 0; c < GC; ++c)
                bfr[buf][c] = *reinterpret_cast<const half8*>(xs + rd_off[sub] + (grp * GC + c) * 2048);
        };
        b_read(0, 0);
#pragma unroll
        for (int hs = 0; hs < 2 * NG; ++hs) {
            const int sub = hs / NG, grp = hs % NG, cb = hs & 1;
            if (hs + 1 < 2 * NG) b_read(cb ^ 1, hs + 1);
#pragma unroll
            for (int ht = 0; ht < HT; ++ht)
#pragma unroll
                for (int c = 0; c < GC; ++c)
                    acc[ht][grp * GC + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[sub][ht], bfr[cb][c],
                                                                                   acc[ht][grp * GC + c], 0, 0, 0);
            if (grp == NG - 1) a_load(sub, s32n + sub);     // refill this sub-step's fragments for the next K step
            {   // one fragment read / one weight load in the shadow of the MFMAs instead of a block behind them
                constexpr int n_mf = HT * GC;
                int used = 0;
                if (hs + 1 < 2 * NG) {
#pragma unroll
                    for (int i = 0; i < GC; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                    }
                    used += 2 * GC;
                }
                if (grp == NG - 1 && used + 2 * HT <= n_mf) {
#pragma unroll
                    for (int i = 0; i < HT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep half sub-steps apart: caps live fragments
        }
    }

    PSTAMP(2)
    // ---- LayerNorm-0 statistics -> LDS -------------------------------------------------------------------------------
#pragma unroll
    for (int c = 0; c < NPASS; ++c) {
        float s1 = st_s[c], s2 = st_q2[c];     // lanes 8j..8j+7 staged the eight pieces of one row
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
        }
        const float mean = s1 / (float)d;
        const float var = fmaxf(s2 / (float)d - mean * mean, 0.f);
        if ((tid & 7) == 0 && st_on[c]) {
            const int row = (tid + c * NT) >> 3;
            s_mu0[row] = mean;
            s_rs0[row] = rsqrt_fast(var + kLnEps);
        }
    }
    __syncthreads();  // stats visible; every wave is done with the staging ring
    PSTAMP(3)

    float* bufS1 = s_red;               // [NWV][ROWS] partial sums of s
    float* bufS2 = s_red + NWV * ROWS;  // [NWV][ROWS] partial sums of s*s
    float* setT0 = s_red + 2 * NWV * ROWS;  // [4][NWV][64]: sum s2, sum s2^2, fc3 class 0 / 1 partials
    float* setT1 = s_red;                   // second set of the same (two-pass tiles, once bufS is dead)

    // sum over the four lanes (q4 = 0..3) that hold the same batch row
    auto rsum4 = [](float v) {
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        return v;
    };

    // Every per-lane LDS address of the epilogues is rebuilt where it is used, from an opaque copy of the lane id:
    // left to itself the compiler computes them all once, keeps ~30 address registers live from the top of the kernel
    // and spills them around the fc2 loops (a reload waits behind s_waitcnt vmcnt(0): the weight prefetch drains).
#define P16_LOCAL_LANE()                          \
    int le_ = lane;                               \
    asm volatile("" : "+v"(le_));                 \
    const int c16 = le_ & 15, q4 = le_ >> 4;      \
    (void)c16;                                    \
    (void)q4;

    // epilogue 1 on column tiles [c0, c0 + nc): LN0 fold, bias, SiLU in place, partial LN1 sums -> bufS; up to four
    // column tiles at a time (sixteen independent SiLU chains per hidden tile)
    auto ep1_cols = [&](const int c0, const int nc) {
        if (P16_ABL(1)) return;
        constexpr int E1 = CT16 < 4 ? CT16 : 4;
#pragma unroll
        for (int cb = 0; cb < CT16; cb += E1) {
            if (cb < c0 || cb >= c0 + nc) continue;
            P16_LOCAL_LANE()
            // (memory clobber: without it the per-hidden-tile constants below - the same LDS words for every column
            // group - are merged across the groups and 32 registers of them stay live through the whole epilogue)
            asm volatile("" ::: "memory");
            float mu[E1], rs[E1], S1[E1], S2[E1];
#pragma unroll
            for (int u = 0; u < E1; ++u) {
                mu[u] = s_mu0[16 * (cb + u) + c16];
                rs[u] = s_rs0[16 * (cb + u) + c16] * L_sc1;
                S1[u] = S2[u] = 0.f;
            }
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) {
                const int nb = 64 * w + 16 * ht + 4 * q4;
                const f32x4 ws = *reinterpret_cast<const f32x4*>(s_cst + nb);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(s_cst + kHidden + nb);
                float sv[E1][4];
#pragma unroll
                for (int u = 0; u < E1; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = fmaf(-mu[u], ws[e], acc[ht][cb + u][e]);
                        sv[u][e] = silu_f(fmaf(rs[u], v, bb[e]));
                    }
                // independent SiLU chains, pinned where they are written (see prober.hip)
#pragma unroll
                for (int u = 0; u < E1; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(sv[u][e]));
#pragma unroll
                for (int u = 0; u < E1; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[ht][cb + u][e] = sv[u][e];
                        S1[u] += sv[u][e];
                        S2[u] = fmaf(sv[u][e], sv[u][e], S2[u]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < E1; ++u) {
                const float t1 = rsum4(S1[u]), t2 = rsum4(S2[u]);
                if (q4 == 0) {
                    bufS1[w * ROWS + 16 * (cb + u) + c16] = t1;
                    bufS2[w * ROWS + 16 * (cb + u) + c16] = t2;
                }
            }
        }
    };

    // LN1 statistics of column tiles [c0, c0 + nc) from the partial sums of all waves.  They go to LDS, over the
    // LayerNorm-0 statistics of the same rows (consumed by epilogue 1 of that tile, a barrier ago): with 16-row tiles a
    // lane carries twice as many per-row values as with 32-row tiles, and 16 of them live across both fc2 loops were
    // registers the 128-row tile does not have.  Every wave writes the same bits and reads them back itself.
    auto mean_cols = [&](const int c0, const int nc) {
        if (P16_ABL(8)) return;
        P16_LOCAL_LANE()
#pragma unroll
        for (int c = 0; c < CT16; ++c)
            if (c >= c0 && c < c0 + nc) {
                const int m = 16 * c + c16;
                // (lane quarter q4 adds waves 2 q4, 2 q4 + 1; the four quarters hold the same batch row)
                float t1 = bufS1[(2 * q4) * ROWS + m] + bufS1[(2 * q4 + 1) * ROWS + m];
                float t2 = bufS2[(2 * q4) * ROWS + m] + bufS2[(2 * q4 + 1) * ROWS + m];
                t1 = rsum4(t1);
                t2 = rsum4(t2);
                const float mean1 = t1 * (1.0f / kHidden);
                const float var = fmaxf(t2 * (1.0f / kHidden) - mean1 * mean1, 0.f);
                s_mu0[m] = mean1;
                s_rs0[m] = rsqrt_fast(var + kLnEps);
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    // fc2 weight fragments: two slots of HALF a K-32 step (two of the wave's four hidden tiles) each, refilled right
    // after use; the first step's fragments of the NEXT pass are requested at the end of a pass.  (Whole steps per slot and double-buffered B fragments - the 32 x 32 kernel's scheme at
    // twice the bytes per step - cost 32 more VGPRs than the 128-row tile has.)
    half8 a2[2][2];
    auto a2_load = [&](int slot, int hsf) {   // half step hsf = 2 ks + pair: tiles 2 pair, 2 pair + 1 of K-32 step ks
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4 v = buf_load16(rs_w2, lane16, ((unsigned)(hsf >> 1) * 32u + 4u * w + 2 * (hsf & 1) + i) << 10);
            a2[slot][i] = __builtin_bit_cast(half8, v);
        }
    };
    // fp8 copy of this wave's W2 rows for the lo term: four slots of ONE (k block, hidden tile) operand each, refilled
    // four operands ahead (a k block's operands of all four tiles in two slots each - 64 VGPRs - spilled the
    // 128-row tile: per K step the 16 x 16 forms hold twice the fragment bytes of the 32 x 32 ones)
    i32x8 q2[4];
    auto q2_load = [&](int slot, int n) {   // operand n = 4 kb + ht
        const unsigned frag = ((unsigned)(n >> 2) * 32u + 4u * w + (n & 3)) << 11;
        const u32x4 v0 = buf_load16(rs_w2q, lane16, frag);
        const u32x4 v1 = buf_load16(rs_w2q, lane16, frag + 1024u);
        q2[slot] = i32x8{(int)v0[0], (int)v0[1], (int)v0[2], (int)v0[3], (int)v1[0], (int)v1[1], (int)v1[2], (int)v1[3]};
    };

    // publish this wave's SiLU outputs of pass g: two K-32 hi fragments per column tile (tiles 2 p, 2 p + 1 of the
    // wave; rounded to nearest, so |lo| <= 2^-12 |s|) and the wave's half of a k block's fp8 lo operand
    auto publish = [&](const int g) {
        if (P16_ABL(2)) return;
        int lane_p = lane;
        asm volatile("" : "+v"(lane_p));
#pragma unroll
        for (int c2 = 0; c2 < GC; ++c2) {
            const int c = g * GC + c2;
            u32x4 lo8;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                u32x4 hi;
#pragma unroll
                for (int hh2 = 0; hh2 < 2; ++hh2) {       // hidden tile 2 p + hh2 -> elements 4 hh2 .. 4 hh2 + 3
                    const int ht = 2 * p + hh2;
                    float lq[4];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const float v0 = acc[ht][c][2 * jj], v1 = acc[ht][c][2 * jj + 1];
                        typedef float float2_t __attribute__((ext_vector_type(2)));
                        const half2_t h01 = __builtin_convertvector(float2_t{v0, v1}, half2_t);   // nearest
                        hi[2 * hh2 + jj] = __builtin_bit_cast(unsigned int, h01);
                        // (v - hi) * 2^13, clamped to the e4m3 range (an overflow converts to NaN)
                        const float kS = (float)(1 << kLoShift);
                        lq[2 * jj] = __builtin_amdgcn_fmed3f(fmaf(v0, kS, -kS * (float)h01[0]), -448.f, 448.f);
                        lq[2 * jj + 1] = __builtin_amdgcn_fmed3f(fmaf(v1, kS, -kS * (float)h01[1]), -448.f, 448.f);
                    }
                    int wd = 0;
                    wd = __builtin_amdgcn_cvt_pk_fp8_f32(lq[0], lq[1], wd, false);
                    wd = __builtin_amdgcn_cvt_pk_fp8_f32(lq[2], lq[3], wd, true);
                    lo8[ht] = (unsigned)wd;               // bytes 4 ht .. 4 ht + 3 of this wave's half operand
                }
                const int fi = ((2 * w + p) * GC + c2) * 64 + lane_p;
                *reinterpret_cast<u32x4*>(s_ex + (size_t)fi * 16) = hi;
            }
            // lo: [k block w >> 1][c2][half w & 1][lane] x 16 B behind the hi fragments
            *reinterpret_cast<u32x4*>(s_ex + (size_t)16 * GC * 1024 +
                                      (size_t)((((w >> 1) * GC + c2) * 2 + (w & 1)) * 64 + lane_p) * 16) = lo8;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto zero2 = [&](f32x4 (&acc2)[HT][GC]) {
#pragma unroll
        for (int i = 0; i < HT; ++i)
#pragma unroll
            for (int c = 0; c < GC; ++c) acc2[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // the fc2 k loop of one pass.  A K-32 step is two half steps (tiles 0-1, then 2-3 of the wave): the B fragment of
    // column tile c2 is single-buffered and re-read for the next step right behind its last MFMA of this one
    auto fc2_loop = [&](f32x4 (&acc2)[HT][GC]) {
        int lane_f = lane;
        asm volatile("" : "+v"(lane_f));
        half8 b2h[GC];
        auto b2_read1 = [&](int c2, int ks) {
            b2h[c2] = *reinterpret_cast<const half8*>(s_ex + (size_t)((ks * GC + c2) * 64 + lane_f) * 16);
        };
#pragma unroll
        for (int c2 = 0; c2 < GC; ++c2) b2_read1(c2, 0);
        auto hi_iter = [&](const int ks, const bool refill) {
            // tiles 0, 1
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c2 = 0; c2 < GC; ++c2)
                    acc2[i][c2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[0][i], b2h[c2], acc2[i][c2], 0, 0, 0);
            if (refill) a2_load(0, 2 * ks + 2);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
            }
            __builtin_amdgcn_sched_barrier(0);
            // tiles 2, 3; then the next step's B fragments (the last one wraps to step 0: valid bytes, never used)
#pragma unroll
            for (int c2 = 0; c2 < GC; ++c2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc2[2 + i][c2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[1][i], b2h[c2], acc2[2 + i][c2], 0, 0, 0);
                b2_read1(c2, (ks + 1) & 15);
            }
            if (refill) a2_load(1, 2 * ks + 3);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll 1
        for (int ks = 0; ks < 15; ++ks) hi_iter(ks, true);
#pragma unroll
        for (int n = 0; n < 4; ++n) q2_load(n, n);   // the lo term's first weight operands, one iteration before they are used
        __builtin_amdgcn_sched_barrier(0);
        hi_iter(15, false);    // (the next pass's first fragments are requested behind the lo term: 16 registers it needs)
        // lo term: four k blocks of 128 hidden units, fp8 x fp8; sixteen (k block, hidden tile) operands of this wave
        const char* s_lo = s_ex + (size_t)16 * GC * 1024;
        i32x8 bq[GC];
        auto bq_read = [&](int kb) {
#pragma unroll
            for (int c2 = 0; c2 < GC; ++c2) {
                const char* pz = s_lo + (size_t)(((kb * GC + c2) * 2) * 64 + lane_f) * 16;
                const u32x4 v0 = *reinterpret_cast<const u32x4*>(pz);
                const u32x4 v1 = *reinterpret_cast<const u32x4*>(pz + 1024);
                bq[c2] = i32x8{(int)v0[0], (int)v0[1], (int)v0[2], (int)v0[3], (int)v1[0], (int)v1[1], (int)v1[2], (int)v1[3]};
            }
        };
        bq_read(0);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int ht = n & 3, kb = n >> 2;
#pragma unroll
            for (int c2 = 0; c2 < GC; ++c2) {
                acc2[ht][c2] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(q2[n & 3], bq[c2], acc2[ht][c2], 0, 0, 0,
                                                                                0x7f7f7f7f, 0, 0x7f7f7f7f);
                if (ht == 3 && kb + 1 < 4) {   // the next k block's operand of this column tile, behind its last use
                    const char* pz = s_lo + (size_t)((((kb + 1) * GC + c2) * 2) * 64 + lane_f) * 16;
                    const u32x4 v0 = *reinterpret_cast<const u32x4*>(pz);
                    const u32x4 v1 = *reinterpret_cast<const u32x4*>(pz + 1024);
                    bq[c2] = i32x8{(int)v0[0], (int)v0[1], (int)v0[2], (int)v0[3], (int)v1[0], (int)v1[1], (int)v1[2], (int)v1[3]};
                }
            }
            if (n + 4 < 16) q2_load(n & 3, n + 4);
            __builtin_amdgcn_sched_barrier(0);
        }
        a2_load(0, 0);     // the next pass starts with its first weight fragments already in registers
        a2_load(1, 1);
    };

    // epilogue 2 of pass g: LN1 fold, bias, SiLU, one-pass LN2 sums, fc3 partial dot products -> set.  Column tiles in
    // groups of EG (per-row running sums are registers: two groups of two instead of one of four)
    auto ep2 = [&](const int g, f32x4 (&acc2)[HT][GC], float* set) {
        if (P16_ABL(4)) {          // (the accumulators must stay live: one cheap store of their sum)
            float keep = 0.f;
#pragma unroll
            for (int i = 0; i < HT; ++i)
#pragma unroll
                for (int c = 0; c < GC; ++c) keep += acc2[i][c][0] + acc2[i][c][3];
            if (keep == 12345.678f) set[lane] = keep;
            return;
        }
        constexpr int EG = GC;      // (two groups of two re-read the per-hidden constants: +1.3 k cycles per call)
#pragma unroll
        for (int cg = 0; cg < GC; cg += EG) {
            asm volatile("" ::: "memory");   // (keeps the constants' LDS loads of each group apart: see epilogue 1)
            P16_LOCAL_LANE()
            float T1[EG], T2[EG], P0[EG], P1[EG], m1[EG], r1[EG];
#pragma unroll
            for (int u = 0; u < EG; ++u) {
                T1[u] = T2[u] = P0[u] = P1[u] = 0.f;
                m1[u] = s_mu0[16 * (g * GC + cg + u) + c16];
                r1[u] = s_rs0[16 * (g * GC + cg + u) + c16] * L_sc2;
            }
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) {
                const int nb = 64 * w + 16 * ht + 4 * q4;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(s_cst + 2 * kHidden + nb);
                const f32x4 w2s = *reinterpret_cast<const f32x4*>(s_cst + 5 * kHidden + nb);
                const f32x4 w30 = *reinterpret_cast<const f32x4*>(s_cst + 3 * kHidden + nb);
                const f32x4 w31 = *reinterpret_cast<const f32x4*>(s_cst + 4 * kHidden + nb);
                float sv[4][EG];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int u = 0; u < EG; ++u) {
                        const float v = fmaf(-m1[u], w2s[e], acc2[ht][cg + u][e]);
                        sv[e][u] = silu_f(fmaf(r1[u], v, bb[e]));
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int u = 0; u < EG; ++u) asm volatile("" : "+v"(sv[e][u]));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int u = 0; u < EG; ++u) {
                        T1[u] += sv[e][u];
                        T2[u] = fmaf(sv[e][u], sv[e][u], T2[u]);
                        P0[u] = fmaf(sv[e][u], w30[e], P0[u]);
                        P1[u] = fmaf(sv[e][u], w31[e], P1[u]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < EG; ++u) {
                const float t1 = rsum4(T1[u]), t2 = rsum4(T2[u]), p0 = rsum4(P0[u]), p1 = rsum4(P1[u]);
                if (q4 == 0) {
                    set[(0 * NWV + w) * 64 + 16 * (cg + u) + c16] = t1;
                    set[(1 * NWV + w) * 64 + 16 * (cg + u) + c16] = t2;
                    set[(2 * NWV + w) * 64 + 16 * (cg + u) + c16] = p0;
                    set[(3 * NWV + w) * 64 + 16 * (cg + u) + c16] = p1;
                }
            }
        }
    };

    // logits of row (16 * GC * g + i) from the cross-wave sums of a set:
    // logits = W3~ . LN2(s2) + b3~ = rstd2 * (W3~.s2 - mean2 * rowsum(W3~)) + b3~
    auto logits_row = [&](const int g, const int i, const float* set) {
        const int row = m0 + 16 * GC * g + i;
        if (row < a.B) {
            float t1 = 0.f, t2 = 0.f, p0 = 0.f, p1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < NWV; ++ww) {
                t1 += set[(0 * NWV + ww) * 64 + i];
                t2 += set[(1 * NWV + ww) * 64 + i];
                p0 += set[(2 * NWV + ww) * 64 + i];
                p1 += set[(3 * NWV + ww) * 64 + i];
            }
            const float mean2 = t1 * (1.0f / kHidden);
            const float var2 = fmaxf(t2 * (1.0f / kHidden) - mean2 * mean2, 0.f);
            const float rstd2 = rsqrt_fast(var2 + kLnEps);
            float2 o;
            o.x = fmaf(rstd2, p0 - mean2 * L_w3sum[0], L_b3[0]);
            o.y = fmaf(rstd2, p1 - mean2 * L_w3sum[1], L_b3[1]);
            *reinterpret_cast<float2*>(a.logits + ((size_t)lrun * a.B + row) * 2) = o;
        }
    };

    if constexpr (NG == 2) {
        // Two fc2 passes; the two waves of every SIMD (w and w + 4) take the VALU phase and the MFMA phase of a pass
        // in opposite orders (see prober.hip: in one order for everybody the matrix pipe idles through every epilogue)
        f32x4 acc2a[HT][GC], acc2b[HT][GC];
        const bool loop_first = w >= NWV / 2;
        ep1_cols(0, GC);
        PSTAMP(4)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            a2_load(u, u);
            __builtin_amdgcn_sched_barrier(0);
        }
        publish(0);
        zero2(acc2a);
        PSTAMP(5)
        __syncthreads();   // fragments and the LN1 partial sums of pass 0 are in LDS
        mean_cols(0, GC);
        PSTAMP(6)
        if (loop_first) {
            __builtin_amdgcn_s_setprio(2);   // finish the k loop first: the other wave then has the pipe alone
            fc2_loop(acc2a);
            __builtin_amdgcn_s_setprio(0);
        }
        ep1_cols(GC, GC);
        if (!loop_first) fc2_loop(acc2a);
        PSTAMP(7)
        __syncthreads();   // every wave is done with the exchange area; LN1 partial sums of pass 1 are in LDS
        publish(1);
        zero2(acc2b);
        __syncthreads();
        mean_cols(GC, GC);
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
        f32x4 acc2[HT][GC];
        ep1_cols(0, CT16);
        a2_load(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a2_load(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        publish(0);
        zero2(acc2);
        __syncthreads();  // fragments and the LN1 partial sums of all waves are in LDS
        mean_cols(0, CT16);
        fc2_loop(acc2);
        ep2(0, acc2, setT0);
        __syncthreads();
        if (tid < 16 * GC) logits_row(0, tid, setT0);
    }
    if (a.decision) {
        // ---- the gate, folded in (one launch less per pass): ticket per row tile; the last of the n_run workgroups of
        // the tile sums softmax(logits[n, row]) over the layers n >= ablation and thresholds - gate_kernel's arithmetic
        // (prober.hip), float32 in layer order, the comparison in double.  Logits of the other workgroups are read with
        // agent-scope loads behind the fence pair (they may come from another XCD's L2).
        __threadfence();
        __syncthreads();                       // every logit of this workgroup is stored; s_red is free
        int* s_ticket = reinterpret_cast<int*>(s_red);
        if (tid == 0)
            *s_ticket = __hip_atomic_fetch_add(a.tile_cnt + tile_idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                                (uint32_t)a.n_run - 1u ? 1 : 0;
        __syncthreads();
        if (*s_ticket) {
            __threadfence();
            for (int i = tid; i < ROWS; i += NT) {
                const int row = m0 + i;
                if (row >= a.B) continue;
                float s0 = 0.f, s1 = 0.f;
                for (int n = a.ablation; n < a.n_run; ++n) {
                    const float* zp = a.logits + ((size_t)n * a.B + row) * 2;
                    const float z0 = __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(zp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    const float z1 = __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(zp + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    const float m = fmaxf(z0, z1);
                    const float e0 = expf(z0 - m), e1 = expf(z1 - m);
                    const float inv = 1.0f / (e0 + e1);
                    s0 += e0 * inv;
                    s1 += e1 * inv;
                }
                if (a.probsum) {
                    a.probsum[2 * row] = s0;
                    a.probsum[2 * row + 1] = s1;
                }
                a.decision[row] = ((double)s0 + a.theta < (double)s1) ? 0 : 1;
            }
            if (tid == 0) __hip_atomic_store(a.tile_cnt + tile_idx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#undef P16_LOCAL_LANE

}  // namespace prag
