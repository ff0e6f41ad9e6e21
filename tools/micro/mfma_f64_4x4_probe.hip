// Probe for v_mfma_f64_4x4x4_4b_f64 (gfx950): operand / result layout and issue rate (round 6: would small groups of flagged
// queries - <= 4 or <= 8 - run faster on four 4 x 4 x 4 blocks than on one 16 x 16 x 4 tile with most columns unused?).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_f64_4x4_probe.hip -o tools/micro/mfma_f64_4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_check(const double* A, const double* B, double* D) {   // per lane one A, one B, one D
    const int l = threadIdx.x;
    double acc = 0.0;
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], acc, 0, 0, 0);
    D[l] = acc;
}

__global__ __launch_bounds__(256) void k_rate(double* out, int n) {
    double a = 1.0 + threadIdx.x, b = 0.5;
    double c[8];
    for (int i = 0; i < 8; ++i) c[i] = 0.0;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    // distinct values per lane in both operands, one instruction, compare every output with the hypothesised layout
    std::vector<double> A(64), B(64), D(64);
    for (int l = 0; l < 64; ++l) { A[l] = 1.0 + l; B[l] = 100.0 + 3 * l; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 64 * 8);
    hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 64 * 8, hipMemcpyDeviceToHost);
    // layout (found from D[0], D[1], D[4] of a first run, checked here on all 64 outputs): every operand has k (A, B) or i (D)
    // in lane >> 4, the block in (lane >> 2) & 3 and the remaining index in lane & 3 - i.e. A is the 16 x 16 x 4 tile's A
    // operand (row = lane & 15 = 4 block + i, k = lane >> 4), B holds q[k][j = lane & 3] once per block, D[i][j] of block b
    // sits in lane 16 i + 4 b + j
    int bad = 0;
    for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double want = 0;
        for (int k = 0; k < 4; ++k) want += A[16 * k + 4 * b + i] * B[16 * k + 4 * b + j];
        if (D[16 * i + 4 * b + j] != want) ++bad;
    }
    printf("layout A[16k+4b+i] B[16k+4b+j] D[16i+4b+j]: %s (%d of 64 outputs differ)\n", bad ? "NO" : "confirmed", bad);
    printf("D[0..7] = "); for (int l = 0; l < 8; ++l) printf("%.0f ", D[l]); printf("\n");
    double* out; hipMalloc(&out, 256 * 256 * 8);
    const int n = 20000;
    hipLaunchKernelGGL(k_rate, dim3(256), dim3(256), 0, 0, out, n);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate, dim3(256), dim3(256), 0, 0, out, n);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = 8.0 * n;
    printf("rate: chip %.1f TFLOP/s f64 on 4x4x4_4b (%.3f ms; 512 flops per instruction; one wave per SIMD, 8 chains)\n",
           256.0 * 4 * mfmas * 512 / (ms * 1e-3) / 1e12, ms);
    return 0;
}
