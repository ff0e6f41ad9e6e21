// How fast can 8 waves per CU stream rows when every wave instruction touches 16 rows x 64 B (exact_mfma_kernel's operand
// layout: lane (r, p) loads the 16-byte piece p of row r's current 64-byte segment) instead of 1 KB of contiguous bytes?
// Same grid (256 x 512 threads), same register ring (H loads of 16 B in flight per lane), same total bytes.
//   A: the kernel's pattern, non-temporal loads      B: wave-contiguous 1 KB per instruction, non-temporal
//   C: the kernel's pattern, ordinary loads          D: lane reads 64 contiguous bytes per 4 steps (row r: 256 B per 4 steps)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/row_piece_stream.hip -o tools/micro/row_piece_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRowB = 1280, kRows = 128, kSteps = kRowB / 64;      // fp16 rows of 640 elements, 20 segments of 64 B

template <int PAT, int H>
__global__ __launch_bounds__(512) void stream_kernel(const char* rows, int64_t n_tiles, uint32_t* out) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, p = lane >> 4;
    u32x4 ring[H];
    u32x4 acc = {0, 0, 0, 0};
    auto addr = [&](int64_t tile, int it) {
        const char* tb = rows + tile * (int64_t)kRows * kRowB;
        if (PAT == 1) return tb + (int64_t)w * 16 * kRowB + it * 1024 + lane * 16;
        if (PAT == 3) return tb + (int64_t)(w * 16 + r) * kRowB + (it >> 2) * 256 + p * 64 + (it & 3) * 16;
        return tb + (int64_t)(w * 16 + r) * kRowB + it * 64 + p * 16;
    };
    auto ld = [&](const char* a) {
        if (PAT == 2) return *reinterpret_cast<const u32x4*>(a);
        return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a));
    };
    if ((int64_t)blockIdx.x < n_tiles)
#pragma unroll
        for (int it = 0; it < H; ++it) ring[it] = ld(addr(blockIdx.x, it));
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t tn = t + gridDim.x < n_tiles ? t + gridDim.x : t;
#pragma unroll
        for (int it = 0; it < kSteps; ++it) {
            const int slot = it % H;
            acc ^= ring[slot];
            ring[slot] = it + H < kSteps ? ld(addr(t, it + H)) : ld(addr(tn, it + H - kSteps));
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int PAT, int H>
static void run(const char* name, const char* rows, int64_t n_tiles, uint32_t* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((stream_kernel<PAT, H>), dim3(256), dim3(512), 0, 0, rows, n_tiles, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i && ms < best) best = ms;
    }
    printf("%-58s H=%2d: %.3f ms = %.2f TB/s\n", name, H, best, (double)n_tiles * kRows * kRowB / (best * 1e-3) / 1e12);
}

int main() {
    const int64_t n_tiles = 4000000 / kRows;
    const size_t bytes = (size_t)n_tiles * kRows * kRowB;
    char* rows; uint32_t* out;
    hipMalloc(&rows, bytes); hipMalloc(&out, 4);
    hipMemset(rows, 1, bytes);
    run<0, 10>("A  16 rows x 64 B per instruction, non-temporal", rows, n_tiles, out);
    run<1, 10>("B  1 KB contiguous per instruction, non-temporal", rows, n_tiles, out);
    run<2, 10>("C  16 rows x 64 B per instruction, ordinary loads", rows, n_tiles, out);
    run<3, 10>("D  lane reads 64 contiguous B over 4 steps, non-temporal", rows, n_tiles, out);
    run<0, 5>("A  16 rows x 64 B per instruction, non-temporal", rows, n_tiles, out);
    run<0, 20>("A  16 rows x 64 B per instruction, non-temporal", rows, n_tiles, out);
    run<1, 20>("B  1 KB contiguous per instruction, non-temporal", rows, n_tiles, out);
    run<3, 20>("D  lane reads 64 contiguous B over 4 steps, non-temporal", rows, n_tiles, out);
    return 0;
}
