// Issue rates of the float64 vector instructions the exact fallback scan is made of (gfx950), one and two waves per SIMD:
// v_fma_f64, v_add_f64, v_cvt_f64_f32, v_cvt_f32_f16, and ds_read_b128 where the four 16-lane groups of a wave read the SAME
// 16 x 64 bytes (the query slice of a grouped exact scan).  Why: round 3 measured a batched exact scan as float64-issue
// bound without saying which instruction; the grouping only pays if the per-(row element, query) work is ONE fma.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/f64_valu_probe.hip -o tools/micro/f64_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(512) void k_rate(double* out, unsigned long long* cyc, int n) {
    __shared__ __attribute__((aligned(16))) double s_q[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) s_q[i] = 1.0 + 1e-3 * i;
    __syncthreads();
    double a[8], acc[8];
    float f[8];
    _Float16 h[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0 + threadIdx.x * 1e-3 + i; acc[i] = 0.0; f[i] = 1.0f + i + threadIdx.x; h[i] = (_Float16)(1.0f + i); }
    const int sub = threadIdx.x & 15;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < n; ++it) {
        if constexpr (OP == 0) {          // 8 independent fma chains
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fma(a[i], a[(i + 1) & 7], acc[i]);
        } else if constexpr (OP == 1) {   // 8 adds
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = acc[i] + a[i];
        } else if constexpr (OP == 2) {   // 8 cvt f32 -> f64 (+ a cheap dependency so they are not hoisted)
#pragma unroll
            for (int i = 0; i < 8; ++i) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(acc[i]) : "v"(f[i])); }
        } else if constexpr (OP == 3) {   // 8 cvt f16 -> f32
#pragma unroll
            for (int i = 0; i < 8; ++i) { asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f[i]) : "v"(h[i])); }
        } else if constexpr (OP == 4) {   // ds_read_b128 x 4: 64 B per lane, same addresses in the 4 lane groups; + 8 fma
            typedef double d2 __attribute__((ext_vector_type(2)));
            const int g = it & 7;
            d2 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const d2*>(s_q + g * 128 + sub * 8 + 2 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc[2 * u] = fma(a[2 * u], q[u][0], acc[2 * u]); acc[2 * u + 1] = fma(a[2 * u + 1], q[u][1], acc[2 * u + 1]); }
        } else {                          // the L2 pair: sub + fma
#pragma unroll
            for (int i = 0; i < 8; ++i) { const double df = a[i] - acc[(i + 3) & 7] * 0.0 - a[(i + 1) & 7]; acc[i] = fma(df, df, acc[i]); }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i] + (double)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name, int threads, double ops_per_iter) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8);
    const int n = 20000;
    hipLaunchKernelGGL(k_rate<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, n);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, n);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 256.0;
    // lane-ops per second over the chip, and cycles (at 2.0 GHz nominal) one SIMD spends per wave instruction
    const double lane_ops = 256.0 * threads * ops_per_iter * n;
    printf("%-34s %d waves/SIMD: %.3f ms, %.2f T lane-ops/s, %.1f ns per wave-instruction per SIMD\n", name, (int)waves_per_simd, ms,
           lane_ops / (ms * 1e-3) / 1e12, ms * 1e6 / (ops_per_iter * n * waves_per_simd));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("v_fma_f64 x8", threads, 8);
        run<1>("v_add_f64 x8", threads, 8);
        run<2>("v_cvt_f64_f32 x8", threads, 8);
        run<3>("v_cvt_f32_f16 x8", threads, 8);
        run<4>("4 ds_read_b128 (shared) + 8 fma_f64", threads, 8);
        run<5>("L2 pair: 2 v_add_f64 + v_mul + fma x8", threads, 8);
    }
    return 0;
}
