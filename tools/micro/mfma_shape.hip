// Microbenchmark: which fp16 MFMA shape should the prober's fc1 loop use?  (cdna_hip_programming.md §5.4 rule 28,
// MI355X_MICROARCH.md "DVFS give-back" item 7: at equal cycles per FLOP the chip can hold a higher clock on one
// shape than on the other, so build both at the same output tile per wave and keep the faster by wall, on random
// data.)  Same workgroup as prober_fused_kernel<1,1,4,8>: 8 waves, 128 batch rows x 512 hidden units, 64-wide K
// steps, weights straight from global memory as ready-made 1-KiB fragments, activations through a 4-stage
// swizzled LDS ring (one barrier per two K steps), LayerNorm-0 sums from the staging registers.
//   SHAPE 32: v_mfma_f32_32x32x16_f16, wave tile = 2 hidden tiles x 4 column tiles of 32x32  (the shipped loop)
//   SHAPE 16: v_mfma_f32_16x16x32_f16, wave tile = 4 hidden tiles x 8 column tiles of 16x16  (same 64 x 128 outputs)
//   FULL 0: bare MFMA loop (operands in registers), FULL 1: the whole loop
// Every wave stamps s_memtime (core clock) and s_memrealtime (100 MHz) around its loop: the in-kernel clock is
// d(memtime) / d(memrealtime) x 100 MHz.  Numerical results are meaningless; only time, cycles and clock matter.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_shape.hip -o tools/micro/mfma_shape
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NWV = 8, ROWS = 128, XSTAGE = ROWS * 128, NT = 512, NPASS = 2;

struct Stamps {
    unsigned long long cyc, rt;
};

template <int SHAPE, int FULL>
__global__ __launch_bounds__(512, 2) void loop_kernel(const u32x4* __restrict__ W, const _Float16* __restrict__ x,
                                                      int d, float* __restrict__ out, Stamps* __restrict__ st) {
    extern __shared__ __attribute__((aligned(16))) char s_x[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = d >> 6;
    const int m0 = (blockIdx.x % 32) * ROWS;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 4096 * 2048 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 512 * 2048 * 2, 0x00020000);
    int st_off[NPASS];
    unsigned xoff[NPASS];
#pragma unroll
    for (int c = 0; c < NPASS; ++c) {
        const int e = tid + c * NT;
        const int row = e >> 3, q = e & 7;
        xoff[c] = (unsigned)(((m0 + row) * d + 8 * q) * 2);
        st_off[c] = row * 128 + ((q ^ ((row >> 1) & 7)) << 4);
    }
    u32x4 xreg[NPASS];
    auto x_load = [&](int t) {
#pragma unroll
        for (int c = 0; c < NPASS; ++c) xreg[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff[c], t * 128, 0);
    };
    auto x_store = [&](int stage) {
#pragma unroll
        for (int c = 0; c < NPASS; ++c) *reinterpret_cast<u32x4*>(s_x + stage * XSTAGE + st_off[c]) = xreg[c];
    };
    float st_s[NPASS] = {0.f, 0.f}, st_q[NPASS] = {0.f, 0.f};
    const half2_t kOnes2 = {(_Float16)1.f, (_Float16)1.f};
    auto x_stats = [&]() {
#pragma unroll
        for (int c = 0; c < NPASS; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned u = xreg[c][j];
                const half2_t xv = __builtin_bit_cast(half2_t, u);
                st_s[c] = __builtin_amdgcn_fdot2(xv, kOnes2, st_s[c], false);
                st_q[c] = __builtin_amdgcn_fdot2(xv, xv, st_q[c], false);
            }
    };
    // a wave's weight fragments: 8 per K step in both shapes (1 KiB each)
    half8 afr[8];
    auto a_load1 = [&](int slot, int t) {   // fragment `slot` (0..7) of K step t
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, ((unsigned)(t * 8 + slot) * 8u + w) << 10, 0);
        afr[slot] = __builtin_bit_cast(half8, v);
    };
    unsigned long long c0, c1, r0, r1;
    float s = 0.f;

    if constexpr (FULL) {
        x_load(0);
        x_store(0);
        x_store(1);
        x_load(2);
#pragma unroll
        for (int i = 0; i < 8; ++i) a_load1(i, 0);
        __syncthreads();
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) a_load1(i, 0);
    }

    if constexpr (SHAPE == 32) {
        const int r = lane & 31, hh = lane >> 5;
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][c][e] = 0.f;
        int rd_row_off[4], rd_sw[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 32 * c + r;
            rd_row_off[c] = row * 128;
            rd_sw[c] = (row >> 1) & 7;
        }
        half8 bfr[2][4];
        auto b_read = [&](const char* xs, int buf, int sub) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                bfr[buf][c] = *reinterpret_cast<const half8*>(xs + rd_row_off[c] + (((2 * sub + hh) ^ rd_sw[c]) << 4));
        };
        if constexpr (!FULL) {
#pragma unroll
            for (int c = 0; c < 4; ++c) bfr[0][c] = bfr[1][c] = afr[c];
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
        for (int t = 0; t < T; ++t) {
            if constexpr (FULL) {
                if ((t & 1) == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                if (t + 2 < T) x_stats();
                x_store((t + 2) & 3);
                x_load(t + 3 < T ? t + 3 : T - 1);
            }
            const char* xs = s_x + (t & 3) * XSTAGE;
            const int tn = t + 1 < T ? t + 1 : T - 1;
            if constexpr (FULL) b_read(xs, 0, 0);
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const int cb = sub & 1;
                if constexpr (FULL)
                    if (sub < 3) b_read(xs, cb ^ 1, sub + 1);
#pragma unroll
                for (int rti = 0; rti < 2; ++rti)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[rti][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[2 * sub + rti], bfr[cb][c], acc[rti][c], 0, 0, 0);
                if constexpr (FULL) {
                    a_load1(2 * sub, tn);
                    a_load1(2 * sub + 1, tn);
                    if (sub < 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][c][e];
    } else {
        const int c16 = lane & 15, q = lane >> 4;
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][c][e] = 0.f;
        // lane (c16, q) reads batch row 16*ct + c16, 16-B piece 4*sub + q of its 128-B line; the swizzle term
        // ((row >> 1) & 7) does not depend on ct, so one address per sub-step and ct as an immediate offset
        const int sw = (c16 >> 1) & 7;
        int rd_off[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) rd_off[sub] = c16 * 128 + (((4 * sub + q) ^ sw) << 4);
        half8 bfr[2][4];
        auto b_read = [&](const char* xs, int buf, int sub, int half) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                bfr[buf][c] = *reinterpret_cast<const half8*>(xs + rd_off[sub] + (4 * half + c) * 2048);
        };
        if constexpr (!FULL) {
#pragma unroll
            for (int c = 0; c < 4; ++c) bfr[0][c] = bfr[1][c] = afr[c];
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
        for (int t = 0; t < T; ++t) {
            if constexpr (FULL) {
                if ((t & 1) == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                if (t + 2 < T) x_stats();
                x_store((t + 2) & 3);
                x_load(t + 3 < T ? t + 3 : T - 1);
            }
            const char* xs = s_x + (t & 3) * XSTAGE;
            const int tn = t + 1 < T ? t + 1 : T - 1;
            if constexpr (FULL) b_read(xs, 0, 0, 0);
#pragma unroll
            for (int hs = 0; hs < 4; ++hs) {   // half sub-steps: (K-32 sub-step, column half)
                const int sub = hs >> 1, half = hs & 1, cb = hs & 1;
                if constexpr (FULL)
                    if (hs < 3) b_read(xs, cb ^ 1, (hs + 1) >> 1, (hs + 1) & 1);
#pragma unroll
                for (int ht = 0; ht < 4; ++ht)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[ht][4 * half + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[4 * sub + ht], bfr[cb][c],
                                                                                       acc[ht][4 * half + c], 0, 0, 0);
                if constexpr (FULL) {
                    if (half == 1) {
#pragma unroll
                        for (int ht = 0; ht < 4; ++ht) a_load1(4 * sub + ht, tn);
                    }
                    if (hs < 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    }
                    if (half == 1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) s += acc[i][c][e];
    }
    s += st_s[0] + st_s[1] + st_q[0] + st_q[1];
    out[(size_t)blockIdx.x * NT + tid] = s;
    if (lane == 0) {
        st[blockIdx.x * NWV + w].cyc = c1 - c0;
        st[blockIdx.x * NWV + w].rt = r1 - r0;
    }
}

template <int SHAPE, int FULL>
static void run(const u32x4* W, const _Float16* x, int d, float* out, Stamps* st, int grid, double soak_s) {
    const int lds = 4 * XSTAGE;
    auto kern = loop_kernel<SHAPE, FULL>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    // soak: back-to-back launches so that the clock the chip holds under THIS load is the one measured
    int n_soak = 0;
    hipEventRecord(a, 0);
    for (;;) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, W, x, d, out, st);
        n_soak += 200;
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms > soak_s * 1e3) break;
    }
    const int n = 200;
    hipEventRecord(a, 0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, W, x, d, out, st);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<Stamps> h(grid * NWV);
    hipMemcpy(h.data(), st, h.size() * sizeof(Stamps), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (auto& v : h) {
        cyc.push_back((double)v.cyc);
        clk.push_back(v.rt ? (double)v.cyc / (double)v.rt * 100.0 : 0.0);
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const int T = d / 64;
    const double us = ms * 1e3 / n;
    const double flop = 2.0 * 128 * 512 * d * grid;
    printf("shape %2d %-4s grid %3d: %7.2f us per launch = %6.1f TF/s | loop cycles per K step: median %6.0f max %6.0f "
           "(floor 2048) | in-kernel clock: median %5.0f MHz (min %5.0f max %5.0f)\n",
           SHAPE, FULL ? "full" : "bare", grid, us, flop / us * 1e-6, cyc[cyc.size() / 2] / T, cyc.back() / T,
           clk[clk.size() / 2], clk.front(), clk.back());
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int d = 2048, B = 4096;
    const double soak = argc > 1 ? atof(argv[1]) : 1.5;
    u32x4* W;
    _Float16* x;
    float* out;
    Stamps* st;
    hipMalloc(&W, (size_t)6 * 512 * d * 2);
    hipMalloc(&x, (size_t)B * d * 2);
    hipMalloc(&out, (size_t)256 * NT * 4);
    hipMalloc(&st, 256 * NWV * sizeof(Stamps));
    std::vector<_Float16> hx((size_t)B * d);
    srand(1);
    for (auto& v : hx) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    for (int l = 0; l < 3; ++l) hipMemcpy((char*)W + (size_t)l * B * d * 2 / 4, hx.data(), (size_t)512 * d * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int grid : {192, 256}) {
            run<32, 0>(W, x, d, out, st, grid, soak);
            run<16, 0>(W, x, d, out, st, grid, soak);
            run<32, 1>(W, x, d, out, st, grid, soak);
            run<16, 1>(W, x, d, out, st, grid, soak);
        }
    return 0;
}
