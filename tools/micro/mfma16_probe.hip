// Probe for the 16 x 16 MFMA forms the prober's fused kernel uses (gfx950):
//  (1) v_mfma_f32_16x16x32_f16: lane (c = l & 15, q = l >> 4) holds A[row c][k = 8 q + j] / B[k = 8 q + j][col c];
//      D: col = l & 15, row = 4 (l >> 4) + e;
//  (2) v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (32 bytes per lane): which k a byte belongs to only
//      matters in that A and B agree - D[i][j] = sum over (lane quarter q, byte b) of A_lane(i, q)[b] * B_lane(j, q)[b] -
//      and its accumulator layout is the f16 instruction's;
//  (3) issue rate of both (cycles per MFMA, one wave per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma16_probe.hip -o tools/micro/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// A: [16 rows][128 k], B: [16 cols][128 k] (k contiguous), values exactly representable in e4m3
__global__ void k_check(const float* A, const float* B, float* C8, float* C16) {
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {   // byte 4 w + t of lane quarter q <-> k = 32 q + 4 w + t
        int wa = 0, wb = 0;
        wa = __builtin_amdgcn_cvt_pk_fp8_f32(A[c * 128 + 32 * q + 4 * w + 0], A[c * 128 + 32 * q + 4 * w + 1], wa, false);
        wa = __builtin_amdgcn_cvt_pk_fp8_f32(A[c * 128 + 32 * q + 4 * w + 2], A[c * 128 + 32 * q + 4 * w + 3], wa, true);
        wb = __builtin_amdgcn_cvt_pk_fp8_f32(B[c * 128 + 32 * q + 4 * w + 0], B[c * 128 + 32 * q + 4 * w + 1], wb, false);
        wb = __builtin_amdgcn_cvt_pk_fp8_f32(B[c * 128 + 32 * q + 4 * w + 2], B[c * 128 + 32 * q + 4 * w + 3], wb, true);
        a[w] = wa;
        b[w] = wb;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int e = 0; e < 4; ++e) C8[lane * 4 + e] = acc[e];
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < 4; ++s) {   // the f16 instruction over the same 128 k
        half8 ha, hb;
        for (int j = 0; j < 8; ++j) {
            ha[j] = (_Float16)A[c * 128 + 32 * s + 8 * q + j];
            hb[j] = (_Float16)B[c * 128 + 32 * s + 8 * q + j];
        }
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d, 0, 0, 0);
    }
    for (int e = 0; e < 4; ++e) C16[lane * 4 + e] = d[e];
}

template <int F8>
__global__ __launch_bounds__(256) void k_rate(float* out, unsigned long long* cyc, int n) {
    i32x8 a, b;
    half8 ha, hb;
    for (int w = 0; w < 8; ++w) {
        a[w] = 0x38383838 + threadIdx.x;
        b[w] = 0x38383838;
        ha[w] = (_Float16)1.f;
        hb[w] = (_Float16)(0.001f * threadIdx.x);
    }
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (F8) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            else c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c[i], 0, 0, 0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 4; ++e) s += c[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float fp8_exact(int i) {
    static const float tab[] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, -0.5f, -1.f, -2.f, 0.25f, -0.25f, 4.f, -3.f, 0.75f, -1.5f, 6.f};
    return tab[i & 15];
}

int main() {
    std::vector<float> A(16 * 128), B(16 * 128);
    srand(3);
    for (auto& v : A) v = fp8_exact(rand());
    for (auto& v : B) v = fp8_exact(rand());
    float *dA, *dB, *dC8, *dC16;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC8, 64 * 4 * 4); hipMalloc(&dC16, 64 * 4 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dB, dC8, dC16);
    std::vector<float> C8(256), C16(256);
    hipMemcpy(C8.data(), dC8, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(C16.data(), dC16, 1024, hipMemcpyDeviceToHost);
    int bad8 = 0, bad16 = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int col = lane & 15, row = 4 * (lane >> 4) + e;   // D[row i of A][col j of B]
            double ref = 0;
            for (int k = 0; k < 128; ++k) ref += (double)A[row * 128 + k] * (double)B[col * 128 + k];
            if (C8[lane * 4 + e] != (float)ref) ++bad8;
            if (C16[lane * 4 + e] != (float)ref) ++bad16;
        }
    printf("16x16x128 fp8 (e4m3, unit scales): %d of 256 results differ from the host sum (0 = operand bytes pair up by (lane quarter, byte); layout col = l & 15, row = 4 (l >> 4) + e)\n", bad8);
    printf("16x16x32 f16 over the same k: %d of 256 differ\n", bad16);
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4);
    hipMalloc(&cyc, 1024 * 8);
    const int n = 2000;
    for (int f8 = 0; f8 < 2; ++f8) {
        for (int rep = 0; rep < 2; ++rep) {
            if (f8) hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(256), 0, 0, out, cyc, n);
            else hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(256), 0, 0, out, cyc, n);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256);
        hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        printf("%s: %.1f cycles per MFMA (one wave per SIMD, 8 accumulators)\n", f8 ? "16x16x128 fp8 scaled" : "16x16x32 f16", s / 256 / n / 8);
    }
    return 0;
}
