// Microbenchmark: the fc1 main loop of prober_fused_kernel<1,1,4,8> (128 batch rows x 512 hidden units per
// workgroup, 8 waves, 64-wide K steps) with its ingredients switchable, to see which of them keeps the
// matrix pipe below 32 cycles per MFMA.  Results are meaningless numerically; only the cycle counts matter.
//   bit 0: B fragments (activations) read from LDS every sub-step (else: registers loaded once)
//   bit 1: A fragments (weights) streamed from global memory (else: registers loaded once)
//   bit 2: activation staging global -> registers -> LDS every K step
//   bit 3: one s_barrier per K step
//   bit 4: LayerNorm-0 sums (v_dot2c) on one extra fragment per sub-step
//   bit 5: global loads addressed as uniform base (SGPR pair) + 32-bit lane offset instead of 64-bit lane pointers
//   bit 6: global loads as buffer_load_dwordx4 (resource + lane offset VGPR + SGPR offset): no address VALU
//   bit 7: LayerNorm-0 sums taken from the staging registers instead (16 v_dot2c per K step, no extra LDS read)
//   bit 8: one LDS read / one weight load placed in the shadow of each MFMA (sched_group_barrier)
//   bit 9: (with bits 7, 8) no branch around the sums (last K step peeled) and sums / staging stores / staging
//          loads of a K step placed in the MFMA shadows of its first sub-step
//   bit 10: four-stage activation ring, staged two K steps ahead, ONE barrier per TWO K steps
//   bit 11: (with bit 10) eight stages, four K steps ahead, one barrier per FOUR K steps
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/fc1_loop.hip -o tools/micro/fc1_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr;

constexpr int NWV = 8, RT = 2, CT = 4, ROWS = 128, XSTAGE = ROWS * 128, NT = 512, NPASS = 2;

template <int MODE>
__global__ __launch_bounds__(512, 2) void fc1_loop(const u32x4* __restrict__ W, const _Float16* __restrict__ x,
                                                   int d, float* __restrict__ out,
                                                   unsigned long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(16))) char s_x[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int T = d >> 6;
    const int m0 = (blockIdx.x % 32) * ROWS;
    const _Float16* xsrc[NPASS];
    int st_off[NPASS];
    unsigned xoff[NPASS];
#pragma unroll
    for (int c = 0; c < NPASS; ++c) {
        const int e = (tid + c * NT) % (256 * CT);
        const int row = e >> 3, q = e & 7;
        xsrc[c] = x + (size_t)(m0 + row) * d + 8 * q;
        xoff[c] = (unsigned)(((m0 + row) * d + 8 * q) * 2);
        st_off[c] = row * 128 + ((q ^ ((row >> 1) & 7)) << 4);
    }
    int rd_row_off[CT], rd_sw[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int row = 32 * c + r;
        rd_row_off[c] = row * 128;
        rd_sw[c] = (row >> 1) & 7;
    }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 4096 * 2048 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 512 * 2048 * 2, 0x00020000);
    u32x4 xreg[NPASS];
    auto x_load = [&](int t) {
#pragma unroll
        for (int c = 0; c < NPASS; ++c) {
            if constexpr (MODE & 64) {
                xreg[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff[c], t * 128, 0);
            } else if constexpr (MODE & 32) {
                const char* base = reinterpret_cast<const char*>(x) + (size_t)t * 128;   // uniform
                xreg[c] = *reinterpret_cast<const u32x4*>(base + xoff[c]);
            } else {
                xreg[c] = *reinterpret_cast<const u32x4*>(xsrc[c] + 64 * t);
            }
        }
    };
    auto x_store = [&](int stage) {
#pragma unroll
        for (int c = 0; c < NPASS; ++c) *reinterpret_cast<u32x4*>(s_x + stage * XSTAGE + st_off[c]) = xreg[c];
    };
    const gptr w1p = (gptr)W + (size_t)(RT * w) * 64 + lane;
    half8 afr[4][RT];
    auto a_load = [&](int slot, int s16) {
#pragma unroll
        for (int rti = 0; rti < RT; ++rti) {
            u32x4 v;
            if constexpr (MODE & 64) {
                v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, (s16 * 16 + RT * w + rti) << 10, 0);
            } else if constexpr (MODE & 32) {
                const char* base = reinterpret_cast<const char*>(W) + ((size_t)(s16 * 16 + RT * w + rti) << 10);  // uniform
                v = *reinterpret_cast<const u32x4*>(base + (unsigned)(lane * 16));
            } else {
                v = w1p[((size_t)s16 * 16 + rti) * 64];
            }
            afr[slot][rti] = __builtin_bit_cast(half8, v);
        }
    };
    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][c][e] = 0.f;
    const int stat_row = 32 * (w % CT) + r;
    const int stat_off = stat_row * 128, stat_sw = (stat_row >> 1) & 7;
    float st_s = 0.f, st_q2 = 0.f;
    float st_s2[NPASS] = {0.f, 0.f}, st_q22[NPASS] = {0.f, 0.f};
    const half2_t kOnes2 = {(_Float16)1.f, (_Float16)1.f};

    x_load(0);
    x_store(0);
    x_store(1);
    x_store(2);
    x_store(3);
    x_store(4);
    x_store(5);
    x_store(6);
    x_store(7);
    x_load(1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        a_load(s, s);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    half8 bfr[2][CT];
    half8 sfr[2];
    auto b_read = [&](const char* xs, int buf, int sub) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
            bfr[buf][c] = *reinterpret_cast<const half8*>(xs + rd_row_off[c] + (((2 * sub + hh) ^ rd_sw[c]) << 4));
        if constexpr (MODE & 16)
            sfr[buf] = *reinterpret_cast<const half8*>(xs + stat_off + (((2 * sub + hh) ^ stat_sw) << 4));
    };
    b_read(s_x, 0, 0);
    b_read(s_x, 1, 1);
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    auto kstep = [&](const int t, const bool sums) {
        if constexpr (MODE & 8) {
            if (!(MODE & 1024) || (t & ((MODE & 2048) ? 3 : 1)) == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        const char* xs = s_x + ((MODE & 2048) ? (t & 7) : (MODE & 1024) ? (t & 3) : (t & 1)) * XSTAGE;
        const int s16n = 4 * (t + 1 < T ? t + 1 : T - 1);
        if constexpr (MODE & 512) {
            if constexpr (MODE & 1) b_read(xs, 0, 0);
        }
        if constexpr (MODE & 4) {
            if constexpr (MODE & 128) {
                if ((MODE & 512) ? sums : (t + 1 < T)) {
#pragma unroll
                    for (int c = 0; c < NPASS; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned u = xreg[c][j];
                            const half2_t xv = __builtin_bit_cast(half2_t, u);
                            st_s2[c] = __builtin_amdgcn_fdot2(xv, kOnes2, st_s2[c], false);
                            st_q22[c] = __builtin_amdgcn_fdot2(xv, xv, st_q22[c], false);
                        }
                }
            }
            x_store((MODE & 2048) ? ((t + 4) & 7) : (MODE & 1024) ? ((t + 2) & 3) : ((t + 1) & 1));
            x_load(t + 3 < T ? t + 3 : T - 1);
        }
        if constexpr (!(MODE & 512)) {
            if constexpr (MODE & 1) b_read(xs, 0, 0);
        }
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const int cb = sub & 1;
            if constexpr (MODE & 1)
                if (sub < 3) b_read(xs, cb ^ 1, sub + 1);
            if constexpr (MODE & 16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const half2_t xv = half2_t{sfr[cb][2 * j], sfr[cb][2 * j + 1]};
                    st_s = __builtin_amdgcn_fdot2(xv, kOnes2, st_s, false);
                    st_q2 = __builtin_amdgcn_fdot2(xv, xv, st_q2, false);
                }
            }
#pragma unroll
            for (int rti = 0; rti < RT; ++rti)
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    acc[rti][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[sub][rti], bfr[cb][c], acc[rti][c], 0, 0, 0);
            if constexpr (MODE & 2) a_load(sub, s16n + sub);
            if constexpr (MODE & 256) {
                if ((MODE & 512) && sub == 0) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // first fragment of this sub-step
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (sums) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (sums) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // staging store
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // staging load + weight refill
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (MODE & 512) {
        for (int t = 0; t < T - 1; ++t) kstep(t, true);
        kstep(T - 1, false);
    } else {
        for (int t = 0; t < T; ++t) kstep(t, false);
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = st_s + st_q2 + st_s2[0] + st_s2[1] + st_q22[0] + st_q22[1];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][c][e];
    out[(size_t)blockIdx.x * NT + tid] = s;
    if (lane == 0) cyc[blockIdx.x * NWV + w] = t1 - t0;
}

template <int MODE>
static void run(const u32x4* W, const _Float16* x, int d, float* out, unsigned long long* cyc, int grid) {
    const int lds = 8 * XSTAGE;
    hipFuncSetAttribute((const void*)fc1_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fc1_loop<MODE>, dim3(grid), dim3(NT), lds, 0, W, x, d, out, cyc);
    hipDeviceSynchronize();
    const int n = 20;
    hipEventRecord(a, 0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(fc1_loop<MODE>, dim3(grid), dim3(NT), lds, 0, W, x, d, out, cyc);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(grid * NWV);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (auto v : h) {
        sum += (double)v;
        if ((double)v > mx) mx = (double)v;
    }
    const int T = d / 64;
    printf("mode %4d [%s%s%s%s%s%s%s%s%s%s%s] grid %3d: %7.1f us per launch | loop cycles per K step: mean %6.0f max %6.0f (64 MFMAs per SIMD: floor 2048)\n",
           MODE, MODE & 1 ? "lds " : "", MODE & 2 ? "wts " : "", MODE & 4 ? "stage " : "", MODE & 8 ? "barrier " : "",
           MODE & 16 ? "stats " : "", MODE & 32 ? "saddr " : "", MODE & 64 ? "buffer " : "", MODE & 128 ? "stats-from-staging " : "", MODE & 256 ? "interleaved " : "", MODE & 512 ? "sums+staging in sub-step 0 " : "", MODE & 1024 ? "4-stage ring, barrier every 2nd K step" : "", grid, ms * 1e3 / n, sum / h.size() / T, mx / T);
}

int main(int argc, char** argv) {
    const int d = 2048, B = 4096, grid = argc > 1 ? atoi(argv[1]) : 192;
    u32x4* W;
    _Float16* x;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&W, (size_t)6 * 512 * d * 2);
    hipMalloc(&x, (size_t)B * d * 2);
    hipMalloc(&out, (size_t)256 * NT * 4);
    hipMalloc(&cyc, 256 * NWV * 8);
    std::vector<_Float16> hx((size_t)B * d);
    srand(1);
    for (auto& v : hx) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    for (int l = 0; l < 3; ++l) hipMemcpy((char*)W + (size_t)l * B * d * 2 / 4, hx.data(), (size_t)512 * d * 2, hipMemcpyHostToDevice);
    run<0>(W, x, d, out, cyc, grid);
    run<1>(W, x, d, out, cyc, grid);
    run<2>(W, x, d, out, cyc, grid);
    run<3>(W, x, d, out, cyc, grid);
    run<8>(W, x, d, out, cyc, grid);
    run<9>(W, x, d, out, cyc, grid);
    run<11>(W, x, d, out, cyc, grid);
    run<15>(W, x, d, out, cyc, grid);
    run<17>(W, x, d, out, cyc, grid);
    run<31>(W, x, d, out, cyc, grid);
    run<34>(W, x, d, out, cyc, grid);
    run<63>(W, x, d, out, cyc, grid);
    run<66>(W, x, d, out, cyc, grid);
    run<67>(W, x, d, out, cyc, grid);
    run<79>(W, x, d, out, cyc, grid);
    run<95>(W, x, d, out, cyc, grid);
    run<207>(W, x, d, out, cyc, grid);
    run<463>(W, x, d, out, cyc, grid);
    run<975>(W, x, d, out, cyc, grid);
    run<1487>(W, x, d, out, cyc, grid);
    run<3535>(W, x, d, out, cyc, grid);
    return 0;
}
