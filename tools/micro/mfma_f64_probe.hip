// Probe for v_mfma_f64_16x16x4_f64 (gfx950): operand / accumulator layout and issue rate.
// Hypothesis checked: lane l (c = l & 15, q = l >> 4) holds A[row c][k = q] and B[k = q][col c]; D[row = 4 q + e][col = c].
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_f64_probe.hip -o tools/micro/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void k_check(const double* A, const double* B, double* D) {   // A [16][4], B [4][16]
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    f64x4 acc = {0., 0., 0., 0.};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[c * 4 + q], B[q * 16 + c], acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) D[lane * 4 + e] = acc[e];
}

__global__ __launch_bounds__(256) void k_rate(double* out, unsigned long long* cyc, int n) {
    double a = 1.0 + threadIdx.x, b = 0.5;
    f64x4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = f64x4{0., 0., 0., 0.};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    std::vector<double> A(64), B(64), D(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i + 0.01 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 1 + 0.5 * j + 7 * k;
    double *dA, *dB, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 256 * 8);
    hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        const int row = 4 * (l >> 4) + e, col = l & 15;
        double want = 0;
        for (int k = 0; k < 4; ++k) want += A[row * 4 + k] * B[k * 16 + col];
        if (D[l * 4 + e] != want) { if (bad < 5) printf("lane %d e %d: got %.6f want %.6f\n", l, e, D[l * 4 + e], want); ++bad; }
    }
    printf("layout A[row=l&15][k=l>>4], B[k=l>>4][col=l&15], D[row=4(l>>4)+e][col=l&15]: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 256 * 8);
    const int n = 20000;
    hipLaunchKernelGGL(k_rate, dim3(256), dim3(256), 0, 0, out, cyc, n);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate, dim3(256), dim3(256), 0, 0, out, cyc, n);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    const double mfmas = 4.0 * n;
    printf("rate: %.1f s_memtime ticks per MFMA (one wave per SIMD, 256 workgroups); chip: %.1f TFLOP/s f64 (%.3f ms)\n",
           c0 / mfmas, 256.0 * 4 * mfmas * 16 * 16 * 4 * 2 / (ms * 1e-3) / 1e12, ms);
    return bad != 0;
}
