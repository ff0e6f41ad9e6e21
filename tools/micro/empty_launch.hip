// Microbenchmark: cost of a kernel that reads one word and exits, as a function of grid / block /
// static LDS - what the exact-fallback launches cost when no query is flagged.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/empty_launch.hip -o /tmp/empty_launch
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS>
__global__ void k_exit(const unsigned* flag, unsigned* out) {
    __shared__ unsigned s[LDS / 4 + 1];
    if (*flag == 0) return;
    s[threadIdx.x % (LDS / 4 + 1)] = threadIdx.x;
    __syncthreads();
    out[blockIdx.x] = s[0];
}
template <int LDS>
static void run(int grid, int block, unsigned* flag, unsigned* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_exit<LDS>, dim3(grid), dim3(block), 0, 0, flag, out);
    hipDeviceSynchronize();
    const int n = 2000;
    hipEventRecord(a, 0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_exit<LDS>, dim3(grid), dim3(block), 0, 0, flag, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("grid %5d block %4d lds %6d : %.2f us per launch\n", grid, block, LDS, ms * 1e3 / n);
}
int main() {
    unsigned *flag, *out;
    hipMalloc(&flag, 4); hipMemset(flag, 0, 4); hipMalloc(&out, 1 << 20);
    for (int grid : {1, 64, 256, 512, 1024}) {
        for (int block : {64, 256, 512, 1024}) {
            run<1024>(grid, block, flag, out);
        }
    }
    run<30000>(512, 512, flag, out);
    run<30000>(256, 256, flag, out);
    run<60000>(256, 512, flag, out);
    return 0;
}
