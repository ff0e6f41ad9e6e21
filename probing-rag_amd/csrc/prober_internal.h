// Shared by the fused prober kernels (prober.hip: 32 x 32 MFMA tiles, every weight / activation mode; prober16.hip:
// 16 x 16 tiles, the fp16 throughput mode): layer records, launch arguments, device helpers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prag_common.h"

namespace prag {

constexpr int kHidden = 512;
constexpr int kClasses = 2;
constexpr float kLnEps = 1e-5f;

struct LayerDev {
    const u32x4* W1f;    // [NA][d/16][16 row tiles][64 lanes] x 16 B
    const u32x4* W2f;    // [NA][32 k steps][16 row tiles][64 lanes] x 16 B
    const u32x4* W2q;    // fp16-weight mode: fp8 (e4m3) copy for the lo term, [8 k blocks][16 row tiles][2][64 lanes] x 16 B
    // fp16-weight mode, 16 x 16 tiles (prober16.hip): [d/32 K-32 steps][32 hidden tiles][64 lanes] x 16 B,
    // [16 K-32 steps][32][64] x 16 B in the k order of fc1's 16 x 16 accumulators, fp8 copy [4 k blocks][32][2][64] x 16 B
    const u32x4* W1g;
    const u32x4* W2g;
    const u32x4* W2qg;
    const float* wsum1;  // [512] row sums of the packed (scaled) W1
    const float* b1;     // [512] b1 + W1 . ln0_b
    const float* b2;     // [512] b2 + W2 . ln1_b
    const float* w2sum;  // [512] row sums of the packed (scaled) W2
    const float* W3;     // [2][512] W3 * ln2_w
    float b3[2];         // b3 + W3 . ln2_b
    float w3sum[2];      // row sums of W3 * ln2_w
    float sc1, sc2;      // 2^-e of the packed fc1 / fc2 weights
};

struct ProberArgs {
    const LayerDev* layers;
    const _Float16* xh;  // raw fp16 activations, or hi part of the normalised ones
    const _Float16* xl;  // lo part (NB == 2) or nullptr
    int64_t x_layer_stride;
    int layer0;
    int B;
    int d;
    int n_tiles;    // row tiles per layer (set by the launcher)
    int n_run;      // layers in this launch
    float* logits;  // [n_run][B][2]
    // gate folded into the launch (prober16_body; decision == nullptr: logits only).  The LAST of the n_run workgroups
    // of a row tile to publish its logits - a ticket per tile in tile_cnt, which it hands back at zero - runs
    // exp_rag.py:407-415 for the tile's rows: gate_kernel's arithmetic, in gate_kernel's order.
    float* probsum = nullptr;       // [B][2] or null
    int32_t* decision = nullptr;    // [B]
    uint32_t* tile_cnt = nullptr;   // [n_tiles] zero between launches
    int ablation = 0;
    double theta = 0.0;
#ifdef PRAG_MM_DIAG
    int stamps;     // 1: phase stamps of three workgroups
    int ablate;     // PRAG_PROBER_ABLATE, timing only (WRONG results): bit 0 no epilogue-1 arithmetic, bit 1 no
                    // publish (hi / lo split + LDS stores), bit 2 no epilogue-2 arithmetic, bit 3 no LN statistics /
                    // logits - what is left is the loads, the LDS staging and every MFMA
#endif
};


// exp_rag.py:407-415 for pooled state b, in the reference's own order (float32, layer by layer): gate_kernel's body, also
// run by the workgroups the two-level search's bound launch adds for it (flat_shadow.hip bound_finish_kernel)
__device__ __forceinline__ void gate_row(const float* __restrict__ logits, int L, int B, int ablation, double theta,
                                         float* __restrict__ probsum, int32_t* __restrict__ decision, int b) {
    if (b >= B) return;
    float s0 = 0.f, s1 = 0.f;
    for (int n = ablation; n < L; ++n) {
        const float2 z = *reinterpret_cast<const float2*>(logits + ((size_t)n * B + b) * 2);
        const float m = fmaxf(z.x, z.y);
        const float e0 = expf(z.x - m), e1 = expf(z.y - m);
        const float inv = 1.0f / (e0 + e1);
        s0 += e0 * inv;
        s1 += e1 * inv;
    }
    if (probsum) {
        probsum[2 * b] = s0;
        probsum[2 * b + 1] = s1;
    }
    // the reference compares Python floats: `s[0].item() + threshold < s[1].item()` (exp_rag.py:414)
    // - float32 sums widened to double, theta a double
    if (decision) decision[b] = ((double)s0 + theta < (double)s1) ? 0 : 1;
}

__device__ __forceinline__ float silu_f(float h) {
    // h * sigmoid(h) as v_mul + v_exp_f32 + v_add + v_rcp_f32 + v_mul; the two transcendentals
    // are ~1 ulp, far inside the 1e-4 budget.  (__frcp_rn / "1.0f / x" expand to the full IEEE
    // division sequence - v_div_scale, v_div_fmas, v_div_fixup: ~10 extra VALU per element.)
    return h * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * h));
}

// Pointers fetched from a struct in memory have no provable address space and
// compile to flat_load (which counts on lgkmcnt too and forces full drains at
// every barrier); these helpers pin them to global memory.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) u32x4* gptr_u32x4;
typedef const __attribute__((address_space(1))) float* gptr_f32;
__device__ __forceinline__ gptr_u32x4 as_global(const u32x4* p) { return (gptr_u32x4)p; }
__device__ __forceinline__ gptr_f32 as_global(const float* p) { return (gptr_f32)p; }

__device__ __forceinline__ float xor32(float v) { return v + __shfl_xor(v, 32, 64); }

// 1/sqrt(v) for the LayerNorm scales inside the fused kernel: v_rsq_f32 (1 ulp) instead of the IEEE
// sqrt + division sequence (~35 dependent VALU each, on the critical path between the fc2 phases)
__device__ __forceinline__ float rsqrt_fast(float v) { return __builtin_amdgcn_rsqf(v); }

// Streams with a uniform base go through buffer loads: resource in SGPRs, one loop-invariant 32-bit lane
// offset in a VGPR, the moving part of the address in an SGPR.  (global_load with 64-bit lane pointers
// needs a v_lshl_add_u64 per load and moves 512 B of addresses per instruction: in the fc1 loop that alone
// cost ~800 of ~3200 cycles per K step - tools/micro/fc1_loop.hip.)  Out-of-range reads return 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0,
                                             bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes, 0x00020000);
}
// Values that are the same for the whole workgroup but reach it through a vector load (anything read from
// the layer table after the first store or barrier): pin them to SGPRs.  A buffer resource the compiler
// cannot prove uniform is otherwise "waterfalled" - a readfirstlane loop around every load.
__device__ __forceinline__ float uniform_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
template <typename T>
__device__ __forceinline__ const T* uniform_p(const T* p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uni_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uni_off, 0);
}


// fc1's activation ring: tile t sits in stage t % kRing and is written kAhead K steps before it is read, so
// one barrier per kAhead K steps orders everything (see the prologue of the kernel)
constexpr int kRing = 4, kAhead = kRing / 2;
constexpr int kLoShift = 13;   // lo is scaled by 2^13 before the fp8 conversion (and W2's copy by 2^-13)

// prober16.hip: the launch on 16 x 16 tiles (rows_per_tile = 32, 64 or 128)
int prober16_launch(const ProberArgs& a, int n_run, int rows_per_tile, hipStream_t st, EventRing& prof);
int prober16_lds_bytes(int ct16);     // dynamic LDS of prober16_body<ct16> (0: no such tile height)

}  // namespace prag
