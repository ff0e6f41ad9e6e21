// Microbenchmark / probe: v_mfma_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands on gfx950 -
// (1) which k an operand byte belongs to only matters in that A and B agree: C[i][j] = sum over the
//     (lane half, byte) pairs; checked against a host sum on random data in fp8-exact values;
// (2) the accumulator layout equals that of v_mfma_f32_32x32x16_f16 (row = 8*(e/4) + 4*half + e%4 ...);
// (3) issue rate against the f16 instruction (cycles per MFMA, one wave per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/fp8_mfma.hip -o tools/micro/fp8_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__global__ void k_check(const float* A, const float* B, float* C8, float* C16) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    i32x8 a, b;
    for (int q = 0; q < 8; ++q) {
        int wa = 0, wb = 0;
        wa = __builtin_amdgcn_cvt_pk_fp8_f32(A[r * 64 + 32 * h + 4 * q + 0], A[r * 64 + 32 * h + 4 * q + 1], wa, false);
        wa = __builtin_amdgcn_cvt_pk_fp8_f32(A[r * 64 + 32 * h + 4 * q + 2], A[r * 64 + 32 * h + 4 * q + 3], wa, true);
        wb = __builtin_amdgcn_cvt_pk_fp8_f32(B[r * 64 + 32 * h + 4 * q + 0], B[r * 64 + 32 * h + 4 * q + 1], wb, false);
        wb = __builtin_amdgcn_cvt_pk_fp8_f32(B[r * 64 + 32 * h + 4 * q + 2], B[r * 64 + 32 * h + 4 * q + 3], wb, true);
        a[q] = wa;
        b[q] = wb;
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int e = 0; e < 16; ++e) C8[lane * 16 + e] = c[e];
    // the f16 instruction on the first 16 k of each half, for the accumulator layout
    f32x16 d = {0};
    for (int s = 0; s < 4; ++s) {
        half8 ha, hb;
        for (int j = 0; j < 8; ++j) {
            ha[j] = (_Float16)A[r * 64 + 16 * s + 8 * h + j];
            hb[j] = (_Float16)B[r * 64 + 16 * s + 8 * h + j];
        }
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, d, 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) C16[lane * 16 + e] = d[e];
}

template <int F8>
__global__ __launch_bounds__(256) void k_rate(float* out, unsigned long long* cyc, int n) {
    i32x8 a, b;
    half8 ha, hb;
    for (int q = 0; q < 8; ++q) {
        a[q] = 0x38383838 + threadIdx.x;
        b[q] = 0x38383838;
        ha[q] = (_Float16)1.f;
        hb[q] = (_Float16)(0.001f * threadIdx.x);
    }
    f32x16 c[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (F8) c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            else c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c[i], 0, 0, 0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += c[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_cvt(const float* in, unsigned char* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) {
        const int w = __builtin_amdgcn_cvt_pk_fp8_f32(in[2 * i], in[2 * i + 1], 0, false);
        out[2 * i] = (unsigned char)(w & 0xff);
        out[2 * i + 1] = (unsigned char)((w >> 8) & 0xff);
    }
}

// the host-side conversion libprag uses for the fp8 copy of W2 (prober.hip: to_e4m3)
static unsigned char to_e4m3(double v) {
    const unsigned char sgn = std::signbit(v) ? 0x80 : 0x00;
    double a = std::fabs(v);
    if (!(a == a)) return 0x7f;
    if (a >= 448.0) return sgn | 0x7e;
    if (a < std::ldexp(1.0, -6)) return sgn | (unsigned char)std::nearbyint(a * 512.0);
    int e;
    (void)std::frexp(a, &e);
    int E = e - 1;
    int q = (int)std::nearbyint(std::ldexp(a, 3 - E)) - 8;
    if (q == 8) {
        q = 0;
        ++E;
    }
    const int code = ((E + 7) << 3) | q;
    return sgn | (unsigned char)(code > 0x7e ? 0x7e : code);
}

static float fp8_exact(int i) {  // values exactly representable in e4m3
    static const float tab[] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, -0.5f, -1.f, -2.f, 0.25f, -0.25f, 4.f, -3.f, 0.75f, -1.5f, 6.f};
    return tab[i & 15];
}

int main() {
    std::vector<float> A(32 * 64), B(32 * 64);
    srand(3);
    for (auto& v : A) v = fp8_exact(rand());
    for (auto& v : B) v = fp8_exact(rand());
    float *dA, *dB, *dC8, *dC16;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC8, 64 * 16 * 4); hipMalloc(&dC16, 64 * 16 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dB, dC8, dC16);
    std::vector<float> C8(1024), C16(1024);
    hipMemcpy(C8.data(), dC8, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(C16.data(), dC16, 4096, hipMemcpyDeviceToHost);
    int bad8 = 0, bad16 = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 16; ++e) {
            const int j = lane & 31, i = 8 * (e / 4) + 4 * (lane >> 5) + (e & 3);   // f16 MFMA layout: lane = column
            double ref = 0;
            for (int k = 0; k < 64; ++k) ref += (double)A[i * 64 + k] * B[j * 64 + k];
            if (std::fabs(C8[lane * 16 + e] - ref) > 1e-3) ++bad8;
            if (std::fabs(C16[lane * 16 + e] - ref) > 1e-3) ++bad16;
        }
    printf("fp8 32x32x64: %d of 1024 outputs differ from the host sum; f16 x4: %d differ\n", bad8, bad16);
    {   // (4) v_cvt_pk_fp8_f32 against the host conversion, on magnitudes from the subnormals to the clamp
        const int n = 1 << 16;
        std::vector<float> h(n);
        for (int i = 0; i < n; ++i) {
            const double mag = std::ldexp(1.0 + (rand() % 4096) / 4096.0, (rand() % 22) - 12);   // 2^-12 .. 2^10
            h[i] = (float)((rand() & 1) ? mag : -mag);
        }
        h[0] = 0.f; h[1] = 448.f; h[2] = 447.9f; h[3] = 464.f; h[4] = 1e-9f; h[5] = -0.f;
        float* di; unsigned char* dq;
        hipMalloc(&di, n * 4); hipMalloc(&dq, n);
        hipMemcpy(di, h.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_cvt, dim3(n / 2 / 256), dim3(256), 0, 0, di, dq, n);
        std::vector<unsigned char> q(n);
        hipMemcpy(q.data(), dq, n, hipMemcpyDeviceToHost);
        int bad = 0, shown = 0, over = 0, over_nan = 0;
        for (int i = 0; i < n; ++i) {
            const unsigned char w = to_e4m3((double)h[i]);
            if (std::fabs(h[i]) >= 464.f) {      // beyond the largest value that rounds to 448: the device gives NaN
                ++over;
                over_nan += (q[i] & 0x7f) == 0x7f;
                continue;
            }
            const bool same = w == q[i] || ((w & 0x7f) == 0 && (q[i] & 0x7f) == 0);   // +-0
            if (!same) {
                ++bad;
                if (shown++ < 8) printf("   %g: device 0x%02x host 0x%02x\n", h[i], q[i], w);
            }
        }
        printf("v_cvt_pk_fp8_f32 vs host to_e4m3: %d of %d in-range values differ; %d of %d out-of-range values convert to NaN "
               "on the device (the kernel clamps first, the host saturates)\n", bad, n - over, over_nan, over);
    }
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int n = 2000;
    for (int f8 = 0; f8 < 2; ++f8) {
        for (int rep = 0; rep < 2; ++rep) {
            if (f8) hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(256), 0, 0, out, cyc, n);
            else hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(256), 0, 0, out, cyc, n);
            hipDeviceSynchronize();
        }
        unsigned long long h[256];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        printf("%s: %.1f cycles per MFMA (one wave per SIMD, 4 accumulators)\n", f8 ? "fp8 32x32x64 (scaled form, scales 1.0)" : "f16 32x32x16", s / 256 / n / 4);
    }
    return 0;
}
