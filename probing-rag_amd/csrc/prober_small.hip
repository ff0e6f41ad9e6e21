// Small-batch prober: LN -> fc1 -> SiLU -> LN -> fc2 -> SiLU -> LN -> fc3 (utils.py:45-57) for up to 8
// rows per layer, all in f32 on the vector ALUs - see prober_small.h.  The arithmetic is the
// reference's, on the LayerNorm-folded weights: LayerNorm WITHOUT affine on the input of each Linear
// (biased variance, eps 1e-5 inside the square root), then  y = W~ xn + b~.
//   small_fc_kernel<T>  one workgroup = 16 output units of one layer (4 per wave); the B input rows are
//                       normalised into LDS once per workgroup; every weight row is read once, 16 B per
//                       lane, two rows (16 loads per lane) in flight per wave; 64-lane butterfly sums.
//                       Used for fc1 (K = d_model, input = caller's activations) and fc2 (K = 512).
//   small_head_kernel   one workgroup: LN2 + fc3 for every (layer, row), then - optionally - the gate
//                       (softmax / sum over layers / threshold, float32 in layer order, exp_rag.py:407-415).
//   small_fused_kernel  (round 4; `make diag` build only since round 5, PRAG_PROBER_SMALL=3) the three stages in ONE launch: 32 workgroups per layer run
//                       fc1, hand their 16 hidden units to the layer's other workgroups through global memory
//                       (write-through stores, one arrival counter per layer), run fc2 the same way, and the workgroup
//                       that arrives last at the final counter runs the head and the gate.  Same arithmetic, same
//                       order: bit-identical logits.  Measured SLOWER than the three launches (30.2 against 25.2 us
//                       per gate, profiles/r04c_latency.txt): each in-launch hand-off is a write-through drain, an
//                       atomic, a poll and a load from beyond L2 (~5 us); a kernel boundary is ~2 us and leaves the
//                       data in L2.  Not the default.
// Round 4 also took two dependent memory round trips out of every stage of BOTH forms: the weight rows of a wave are
// requested before the LayerNorm of its input (they do not depend on it) and an input row is read once, not three
// times (28.6 -> 25.2 us for the three launches).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "prag_common.h"
#include "prober_small.h"

namespace prag {

constexpr int kSmH = 512;
constexpr float kSmEps = 1e-5f;

template <typename T>
__device__ __forceinline__ f32x4 sm_load4(const T* p) {
    if constexpr (sizeof(T) == 4) {
        return *reinterpret_cast<const f32x4*>(p);
    } else if constexpr (std::is_same<T, _Float16>::value) {
        const half4 h = *reinterpret_cast<const half4*>(p);
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    } else {  // bf16 bits: the upper half of an f32
        const ushort4 u = *reinterpret_cast<const ushort4*>(p);
        return f32x4{__uint_as_float((uint32_t)u.x << 16), __uint_as_float((uint32_t)u.y << 16),
                     __uint_as_float((uint32_t)u.z << 16), __uint_as_float((uint32_t)u.w << 16)};
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float sm_silu(float h) { return h / (1.0f + __expf(-h)); }

// LayerNorm (no affine: folded into W / bias) of the B input rows -> LDS.  A row is read ONCE into registers
// (K <= 4096: 16 x 16 B per lane) - round 3 read it three times, three dependent memory round trips in front of
// the first weight - mean, centred second moment and the normalised values come from the registers, in the
// order of the three-pass form (same sums, same results).
// (the K <= 4096 form: the row sits in registers, 16 x 16 B per lane)
__device__ __forceinline__ void small_ln_regs(const f32x4 (&v)[16], int K, int lane, float* dst) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if (lane * 4 + u * 256 < K) s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    const float mean = wave_sum(s) / (float)K;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if (lane * 4 + u * 256 < K) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dv = v[u][e] - mean;
                q = fmaf(dv, dv, q);
            }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + kSmEps);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int i = lane * 4 + u * 256;
        if (i < K)
            *reinterpret_cast<f32x4*>(dst + i) =
                f32x4{(v[u][0] - mean) * rstd, (v[u][1] - mean) * rstd, (v[u][2] - mean) * rstd, (v[u][3] - mean) * rstd};
    }
}

template <typename T>
__device__ __forceinline__ void small_ln_rows(const T* __restrict__ in_rows, int B, int K, float* s_xn, int w, int lane) {
    for (int b = w; b < B; b += 4) {
        const T* src = in_rows + (int64_t)b * K;
        if (K <= 4096) {
            f32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = lane * 4 + u * 256;
                v[u] = i < K ? sm_load4(src + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            small_ln_regs(v, K, lane, s_xn + b * K);
        } else {
            float s = 0.f;
            for (int i = lane * 4; i < K; i += 256) {
                const f32x4 v = sm_load4(src + i);
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
            const float mean = wave_sum(s) / (float)K;
            float q = 0.f;
            for (int i = lane * 4; i < K; i += 256) {
                const f32x4 v = sm_load4(src + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dv = v[e] - mean;
                    q = fmaf(dv, dv, q);
                }
            }
            const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + kSmEps);
            for (int i = lane * 4; i < K; i += 256) {
                const f32x4 v = sm_load4(src + i);
                *reinterpret_cast<f32x4*>(s_xn + b * K + i) =
                    f32x4{(v[0] - mean) * rstd, (v[1] - mean) * rstd, (v[2] - mean) * rstd, (v[3] - mean) * rstd};
            }
        }
    }
}

// The decode-step form (round 6, K <= 4096): the input row of the gate is the running sum of the hooked activations,
// acc_out = (assign ? 0 : acc_in) + h - pool_accumulate_layers_kernel's arithmetic, element by element - formed in
// registers on the way into the LayerNorm; `acc_out` != null (one workgroup per layer) also stores it.
template <typename T>
__device__ __forceinline__ void small_ln_rows_step(const float* __restrict__ acc_in, const T* __restrict__ h,
                                                   float* __restrict__ acc_out, int assign, int B, int K, float* s_xn, int w,
                                                   int lane) {
    for (int b = w; b < B; b += 4) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = lane * 4 + u * 256;
            if (i < K) {
                const f32x4 hv = sm_load4(h + (int64_t)b * K + i);
                v[u] = assign ? hv : (*reinterpret_cast<const f32x4*>(acc_in + (int64_t)b * K + i) + hv);
                if (acc_out) *reinterpret_cast<f32x4*>(acc_out + (int64_t)b * K + i) = v[u];
            } else {
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        small_ln_regs(v, K, lane, s_xn + b * K);
    }
}

// The first 2048 columns of a wave's four weight rows (two pairs x two rows x 8 x 16 B per lane = 128 VGPRs): they
// do not depend on the input, so they are requested BEFORE the LayerNorm of the input rows and arrive under it
// (round 3: LayerNorm, then pair 0's loads, then pair 1's - four dependent memory round trips per launch).
struct SmallWPre {
    f32x4 a[2][8], b[2][8];
};
__device__ __forceinline__ void small_w_preload(SmallWPre& p, const float* __restrict__ W, int K, int n0, int lane) {
#pragma unroll
    for (int pair = 0; pair < 2; ++pair) {
        const float* wa = W + (int64_t)(n0 + 2 * pair) * K;
        const float* wb = wa + K;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane * 4 + u * 256;
            const bool ok = i < K;
            p.a[pair][u] = ok ? *reinterpret_cast<const f32x4*>(wa + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            p.b[pair][u] = ok ? *reinterpret_cast<const f32x4*>(wb + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// one lane waits until *cnt >= target; false (and *giveup = code) after ~50 ms
__device__ __forceinline__ bool wait_counter(uint32_t* cnt, uint32_t target, uint32_t* giveup, uint32_t code) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(2);
        if (spins > (1u << 22)) {
            __hip_atomic_store(giveup, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
}

// 16 output units of a layer (4 per wave, as two pairs) from the B normalised rows in LDS: SiLU(W xn + bias) ->
// out_rows[b][n] (SC1: write-through stores, for the one-launch form's hand-off).  `pre` holds the first 2048
// columns of both pairs' weight rows (small_w_preload); wider inputs load the rest step by step.
template <int B_MAX, bool SC1>
__device__ __forceinline__ void small_fc_tile(const float* __restrict__ W, const float* __restrict__ bias, const float* s_xn,
                                              int B, int K, int n0, int lane, float* out_rows, const SmallWPre& pre) {
#pragma unroll
    for (int pair = 0; pair < 2; ++pair) {
        const int na = n0 + 2 * pair, nb = na + 1;
        const float* wa = W + (int64_t)na * K;
        const float* wb = W + (int64_t)nb * K;
        float acc_a[B_MAX], acc_b[B_MAX];
#pragma unroll
        for (int b = 0; b < B_MAX; ++b) acc_a[b] = acc_b[b] = 0.f;
        auto fma_step = [&](const f32x4 (&va)[8], const f32x4 (&vb)[8], int i0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                if (i < K) {
#pragma unroll
                    for (int b = 0; b < B_MAX; ++b) {
                        if (b < B) {
                            const f32x4 x = *reinterpret_cast<const f32x4*>(s_xn + b * K + i);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc_a[b] = fmaf(va[u][e], x[e], acc_a[b]);
                                acc_b[b] = fmaf(vb[u][e], x[e], acc_b[b]);
                            }
                        }
                    }
                }
            }
        };
        fma_step(pre.a[pair], pre.b[pair], lane * 4);
        for (int i0 = lane * 4 + 2048; i0 < K; i0 += 2048) {      // 8 x 16-B loads per row per lane per step
            f32x4 va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                const bool ok = i < K;
                va[u] = ok ? *reinterpret_cast<const f32x4*>(wa + i) : f32x4{0.f, 0.f, 0.f, 0.f};
                vb[u] = ok ? *reinterpret_cast<const f32x4*>(wb + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            fma_step(va, vb, i0);
        }
#pragma unroll
        for (int b = 0; b < B_MAX; ++b) {
            if (b < B) {
                const float ha = wave_sum(acc_a[b]) + bias[na];
                const float hb = wave_sum(acc_b[b]) + bias[nb];
                if (lane == 0) {
                    float* o = out_rows + (int64_t)b * kSmH;
                    if constexpr (SC1) {
                        st_sc1(o + na, sm_silu(ha));
                        st_sc1(o + nb, sm_silu(hb));
                    } else {
                        o[na] = sm_silu(ha);
                        o[nb] = sm_silu(hb);
                    }
                }
            }
        }
    }
}

// in: [n_run][B][K] (element type T, layer stride in_layer_stride); W of layer l: [512][K]; out: [n_run][B][512]
template <typename T, int B_MAX>
__global__ __launch_bounds__(256) void small_fc_kernel(const SmallLayer* __restrict__ layers, int layer0, int which,
                                                      const T* __restrict__ in, int64_t in_layer_stride, int B, int K,
                                                      float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s_xn[];   // [B][K] normalised input rows
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lrun = blockIdx.y;
    const SmallLayer& L = layers[layer0 + lrun];
    const float* W = which == 0 ? L.W1 : L.W2;
    const float* bias = which == 0 ? L.b1 : L.b2;
    const int n0 = blockIdx.x * 16 + w * 4;
    SmallWPre pre;
    small_w_preload(pre, W, K, n0, lane);
    small_ln_rows<T>(in + (int64_t)lrun * in_layer_stride, B, K, s_xn, w, lane);
    __syncthreads();
    small_fc_tile<B_MAX, false>(W, bias, s_xn, B, K, n0, lane, out + (int64_t)lrun * B * kSmH, pre);
}

__global__ __launch_bounds__(256) void small_head_kernel(const SmallLayer* __restrict__ layers, int layer0, int n_run,
                                                        int B, const float* __restrict__ h2, float* __restrict__ logits,
                                                        int ablation, double theta, float* __restrict__ probsum,
                                                        int32_t* __restrict__ decision) {
    __shared__ float s_logit[64 * kSmallMaxB * 2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int pr = w; pr < n_run * B; pr += 4) {
        const int lrun = pr / B;
        const SmallLayer& L = layers[layer0 + lrun];
        const float* s = h2 + (int64_t)pr * kSmH;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(s + lane * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(s + lane * 8 + 4);
        const float mean = wave_sum((v0[0] + v0[1]) + (v0[2] + v0[3]) + (v1[0] + v1[1]) + (v1[2] + v1[3])) / (float)kSmH;
        float q = 0.f, n8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            n8[e] = (e < 4 ? v0[e] : v1[e - 4]) - mean;
            q = fmaf(n8[e], n8[e], q);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)kSmH + kSmEps);
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xn = n8[e] * rstd;
            d0 = fmaf(L.W3[lane * 8 + e], xn, d0);
            d1 = fmaf(L.W3[kSmH + lane * 8 + e], xn, d1);
        }
        d0 = wave_sum(d0) + L.b3[0];
        d1 = wave_sum(d1) + L.b3[1];
        if (lane == 0) {
            logits[(int64_t)pr * 2] = d0;
            logits[(int64_t)pr * 2 + 1] = d1;
            s_logit[pr * 2] = d0;
            s_logit[pr * 2 + 1] = d1;
        }
    }
    if (!decision && !probsum) return;
    __syncthreads();
    if (tid < B) {   // the gate: same arithmetic and order as gate_kernel (prober.hip)
        float s0 = 0.f, s1 = 0.f;
        for (int n = ablation; n < n_run; ++n) {
            const float z0 = s_logit[(n * B + tid) * 2], z1 = s_logit[(n * B + tid) * 2 + 1];
            const float m = fmaxf(z0, z1);
            const float e0 = expf(z0 - m), e1 = expf(z1 - m);
            const float inv = 1.0f / (e0 + e1);
            s0 += e0 * inv;
            s1 += e1 * inv;
        }
        if (probsum) {
            probsum[2 * tid] = s0;
            probsum[2 * tid + 1] = s1;
        }
        if (decision) decision[tid] = ((double)s0 + theta < (double)s1) ? 0 : 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct SmallSync {          // device words of one prober handle, zero between launches
    uint32_t cnt1[64];      // per layer: workgroups whose fc1 outputs are stored
    uint32_t cnt2;          // workgroups whose fc2 outputs are stored (all layers)
    uint32_t giveup;        // != 0: a wait timed out (the outputs of that launch are not valid)
    uint32_t pad[2];
};

// One launch for the whole small-batch gate.
// Hand-offs follow cdna_hip_programming.md Guideline 16 in its counter form, with write-through payloads:
//   producer: every value another workgroup will read is stored `sc1` (relaxed agent-scope atomic store = one
//             global_store_dword sc1), every storing wave drains (`s_waitcnt vmcnt(0)`), the workgroup meets at a
//             barrier, ONE lane adds to the arrival counter (relaxed, agent scope);
//   consumer: ONE lane polls the counter with relaxed agent-scope loads (`sc1`, s_sleep between polls, bounded),
//             the workgroup meets at a barrier, then EVERY load of handed-off bytes is an `sc1` load to registers
//             (relaxed agent-scope atomic load) - no plain load ever touches h1 / h2, so no L1 line can be stale.
// Placement-independent: nothing depends on dispatch order or on which XCD a workgroup runs; every workgroup of the
// grid (32 x n_run <= 2048, one or more per CU) is resident or becomes resident as others finish only AFTER their
// last wait, and a workgroup that waits occupies one CU slot of many, so waiting cannot starve the producers it
// waits for unless another kernel holds the rest of the chip - the spin is bounded and sets a give-up word then.
// The arrival counters return to zero inside the launch (the last arriver resets them after every poll and ticket
// of this launch has happened), so a replayed graph node starts from clean state; they are zeroed once at creation.
// ---------------------------------------------------------------------------------------------------------------

//
// Round 6 - STEP: the launch a decode step of the LM ends with (prag_pool_step_gate).  Stage 1's input rows are the
// running sums of the hooked activations, formed on the way in (acc_out = acc_in + h, stored by the first workgroup of
// each layer; acc_in and acc_out are DIFFERENT buffers - the layer's other workgroups still read acc_in), and the last
// stage leaves decision, sums and the step's tag in pinned host memory: the decision is on the host before
// `generate` returns, computed in the shadow of the LM's last layers.  Standing alone the one-launch form is ~5 us
// slower on the GPU than three launches (profiles/r04c_latency.txt); inside the loop the launch count per token is
// what the host pays for, and it stays ONE (it replaces pool_accumulate_layers_kernel's).
struct SmallStepArgs {
    const void* h[kSmallStepMaxLayers];   // where the model left each layer's activations of this step ([B][K], type T)
    const float* acc_in;                  // [n_run][B][K] sums before this step (not read when assign)
    float* acc_out;                       // [n_run][B][K] sums after it
    int assign;
    uint64_t tag;                         // written to host[0] last (bit 63 set: a hand-off timed out, outputs void)
    uint64_t* host;                       // mapped pinned block: tag | int32 decision[8] | float probsum[8][2]
};

template <typename T, int B_MAX, bool STEP>
__global__ __launch_bounds__(256) void small_fused_kernel(const SmallLayer* __restrict__ layers, int layer0, int n_run,
                                                         const T* __restrict__ in, int64_t in_layer_stride, int B, int K,
                                                         float* h1, float* h2, SmallSync* sync,
                                                         float* __restrict__ logits, int ablation, double theta,
                                                         float* __restrict__ probsum, int32_t* __restrict__ decision,
                                                         SmallStepArgs step) {
    extern __shared__ __attribute__((aligned(16))) float s_xn[];   // [B][max(K, 512)] normalised rows | head scratch
    // (the flag word sits BEHIND the rows in the dynamic region: a static __shared__ object in front would move
    // the region's base off its 16-byte alignment - cdna_hip_programming.md Guideline 17)
    int& s_flag = *reinterpret_cast<int*>(s_xn + (size_t)B * (K > kSmH ? K : kSmH));
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lrun = blockIdx.y;
    const SmallLayer& L = layers[layer0 + lrun];
    const int n0 = blockIdx.x * 16 + w * 4;
    // ---- stage 1: LayerNorm of the caller's rows -> LDS, fc1 + SiLU -> h1 (write-through) ----------------------
    SmallWPre pre;
    small_w_preload(pre, L.W1, K, n0, lane);
    if constexpr (STEP) {
        const int64_t lo = (int64_t)lrun * B * K;
        small_ln_rows_step<T>(step.acc_in + lo, reinterpret_cast<const T*>(step.h[lrun]),
                              blockIdx.x == 0 ? step.acc_out + lo : nullptr, step.assign, B, K, s_xn, w, lane);
    } else {
        small_ln_rows<T>(in + (int64_t)lrun * in_layer_stride, B, K, s_xn, w, lane);
    }
    __syncthreads();
    small_fc_tile<B_MAX, true>(L.W1, L.b1, s_xn, B, K, n0, lane, h1 + (int64_t)lrun * B * kSmH, pre);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(&sync->cnt1[lrun], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (a timed-out wait goes on with whatever is there - the give-up word says the outputs are void - so that
        // the tickets below are still drawn and the counters return to zero)
    }
    // fc2's weight rows do not depend on the hand-off: request them before waiting for it
    small_w_preload(pre, L.W2, kSmH, n0, lane);
    if (tid == 0) (void)wait_counter(&sync->cnt1[lrun], gridDim.x, &sync->giveup, 1u);
    __syncthreads();
    // ---- stage 2: the layer's 512 fc1 outputs (sc1 loads only), LayerNorm -> LDS, fc2 + SiLU -> h2 ---------------
    for (int b = w; b < B; b += 4) {
        const float* src = h1 + ((int64_t)lrun * B + b) * kSmH;
        // the elements and the summation order of small_fc_kernel's own LayerNorm (K = 512: two steps of four)
        float v[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * u + e] = ld_sc1(src + u * 256 + lane * 4 + e);
        float s = 0.f;
        s += (v[0] + v[1]) + (v[2] + v[3]);
        s += (v[4] + v[5]) + (v[6] + v[7]);
        const float mean = wave_sum(s) / (float)kSmH;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float dv = v[e] - mean;
            q = fmaf(dv, dv, q);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)kSmH + kSmEps);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            *reinterpret_cast<f32x4*>(s_xn + b * kSmH + u * 256 + lane * 4) =
                f32x4{(v[4 * u] - mean) * rstd, (v[4 * u + 1] - mean) * rstd, (v[4 * u + 2] - mean) * rstd, (v[4 * u + 3] - mean) * rstd};
    }
    __syncthreads();
    small_fc_tile<B_MAX, true>(L.W2, L.b2, s_xn, B, kSmH, n0, lane, h2 + (int64_t)lrun * B * kSmH, pre);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint32_t n_wg = gridDim.x * gridDim.y;
    if (tid == 0) s_flag = __hip_atomic_fetch_add(&sync->cnt2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_wg - 1 ? 1 : 0;
    __syncthreads();
    if (!s_flag) return;
    __syncthreads();
    // ---- stage 3 (the workgroup whose ticket was the last): LN2 + fc3 for every (layer, row), then the gate ------
    // every other workgroup has passed its last wait and drawn its ticket: the counters can go back to zero
    if (tid < (int)gridDim.y) __hip_atomic_store(&sync->cnt1[tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_store(&sync->cnt2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float* s_logit = s_xn;                                   // [n_run * B][2]
    for (int pr = w; pr < n_run * B; pr += 4) {
        const int lr = pr / B;
        const SmallLayer& LL = layers[layer0 + lr];
        const float* s = h2 + (int64_t)pr * kSmH;
        float n8[8];
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {                         // lane's elements 8*lane .. 8*lane+7, as small_head_kernel
            n8[e] = ld_sc1(s + lane * 8 + e);
        }
        sum = (n8[0] + n8[1]) + (n8[2] + n8[3]) + (n8[4] + n8[5]) + (n8[6] + n8[7]);
        const float mean = wave_sum(sum) / (float)kSmH;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            n8[e] -= mean;
            q = fmaf(n8[e], n8[e], q);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)kSmH + kSmEps);
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xn = n8[e] * rstd;
            d0 = fmaf(LL.W3[lane * 8 + e], xn, d0);
            d1 = fmaf(LL.W3[kSmH + lane * 8 + e], xn, d1);
        }
        d0 = wave_sum(d0) + LL.b3[0];
        d1 = wave_sum(d1) + LL.b3[1];
        if (lane == 0) {
            logits[(int64_t)pr * 2] = d0;
            logits[(int64_t)pr * 2 + 1] = d1;
            s_logit[pr * 2] = d0;
            s_logit[pr * 2 + 1] = d1;
        }
    }
    if constexpr (!STEP) {
        if (!decision && !probsum) return;
    }
    __syncthreads();
    if (tid < B) {   // the gate: same arithmetic and order as gate_kernel (prober.hip)
        float s0 = 0.f, s1 = 0.f;
        for (int n = ablation; n < n_run; ++n) {
            const float z0 = s_logit[(n * B + tid) * 2], z1 = s_logit[(n * B + tid) * 2 + 1];
            const float m = fmaxf(z0, z1);
            const float e0 = expf(z0 - m), e1 = expf(z1 - m);
            const float inv = 1.0f / (e0 + e1);
            s0 += e0 * inv;
            s1 += e1 * inv;
        }
        const int32_t dec = ((double)s0 + theta < (double)s1) ? 0 : 1;
        if (probsum) {
            probsum[2 * tid] = s0;
            probsum[2 * tid + 1] = s1;
        }
        if (decision) decision[tid] = dec;
        if constexpr (STEP) {
            int32_t* hd = reinterpret_cast<int32_t*>(step.host + 1);
            float* hp = reinterpret_cast<float*>(hd + kSmallMaxB);
            hd[tid] = dec;
            hp[2 * tid] = s0;
            hp[2 * tid + 1] = s1;
            __threadfence_system();           // the results are on their way to the host before the tag
        }
    }
    if constexpr (STEP) {
        __syncthreads();
        if (tid == 0) {
            // a hand-off of this launch that timed out voids the outputs: the tag says so and the word is handed back
            const uint32_t gave_up = __hip_atomic_load(&sync->giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gave_up) __hip_atomic_store(&sync->giveup, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(step.host, step.tag | (gave_up ? (1ull << 63) : 0ull), __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

size_t small_sync_bytes() { return sizeof(SmallSync); }

template <typename T, int BM, bool STEP>
static int launch_fused_small_bm(const SmallRun& r, const SmallStepArgs& step, hipStream_t st) {
    const dim3 grid(kSmH / 16, r.n_run), block(256);
    const size_t lds = (size_t)r.B * (r.d > kSmH ? r.d : kSmH) * sizeof(float) + 16;   // + the flag word
    auto kern = small_fused_kernel<T, BM, STEP>;
    static LdsOptIn lds_opt_in;     // 8 rows x 2048 (or 4 x 4096) floats + the flag word is just over 64 KiB
    {
        const int rc_ = lds_opt_in.ensure(reinterpret_cast<const void*>(kern), kSmallMaxElems * (int)sizeof(float) + 16);
        if (rc_ != PRAG_OK) return rc_;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, r.layers, r.layer0, r.n_run, reinterpret_cast<const T*>(r.x),
                       r.x_layer_stride, r.B, r.d, r.h1, r.h2, reinterpret_cast<SmallSync*>(r.sync), r.logits, r.ablation,
                       r.theta, r.probsum, r.decision, step);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

template <typename T, bool STEP>
static int launch_fused_small(const SmallRun& r, const SmallStepArgs& step, hipStream_t st) {
    if (r.B <= 1) return launch_fused_small_bm<T, 1, STEP>(r, step, st);
    if (r.B <= 2) return launch_fused_small_bm<T, 2, STEP>(r, step, st);
    if (r.B <= 4) return launch_fused_small_bm<T, 4, STEP>(r, step, st);
    return launch_fused_small_bm<T, 8, STEP>(r, step, st);
}

// One decode step + the gate on the sums so far, one launch (prag_pool_step_gate).  r.x is not read.
int small_step_run(const SmallRun& r, const SmallStep& sp, hipStream_t st) {
    PRAG_REQUIRE(small_supported(r.B, r.d) && r.d <= 4096 && r.d % 4 == 0 && r.layer0 == 0 && r.n_run >= 1 &&
                     r.n_run <= kSmallStepMaxLayers && r.sync != nullptr,
                 PRAG_EUNSUPPORTED, "internal: step gate called outside its envelope (B=%d d=%d layers=%d)", r.B, r.d, r.n_run);
    SmallStepArgs a{};
    for (int l = 0; l < r.n_run; ++l) a.h[l] = sp.h[l];
    a.acc_in = sp.acc_in;
    a.acc_out = sp.acc_out;
    a.assign = sp.assign;
    a.tag = sp.tag;
    a.host = sp.host_dev;
    if (r.x_dtype == PRAG_F32) return launch_fused_small<float, true>(r, a, st);
    if (r.x_dtype == PRAG_F16) return launch_fused_small<_Float16, true>(r, a, st);
    return launch_fused_small<unsigned short, true>(r, a, st);
}

template <typename T>
static void launch_fc1(const SmallRun& r, hipStream_t st) {
    const dim3 grid(kSmH / 16, r.n_run), block(256);
    const size_t lds = (size_t)r.B * r.d * sizeof(float);
    const T* x = reinterpret_cast<const T*>(r.x);
#define PRAG_SM(BM) hipLaunchKernelGGL((small_fc_kernel<T, BM>), grid, block, lds, st, r.layers, r.layer0, 0, x, \
                                       r.x_layer_stride, r.B, r.d, r.h1)
    if (r.B <= 1) PRAG_SM(1);
    else if (r.B <= 2) PRAG_SM(2);
    else if (r.B <= 4) PRAG_SM(4);
    else PRAG_SM(8);
#undef PRAG_SM
}

int small_run(const SmallRun& r, hipStream_t st) {
    PRAG_REQUIRE(small_supported(r.B, r.d) && r.n_run >= 1 && r.n_run <= 64 && r.d % 4 == 0, PRAG_EUNSUPPORTED,
                 "internal: small-batch prober called outside its envelope (B=%d d=%d)", r.B, r.d);
    if (r.sync && r.fused) {        // one launch (PRAG_PROBER_SMALL=3: A/B timing; standing alone three launches are faster)
        const SmallStepArgs none{};
        if (r.x_dtype == PRAG_F32) return launch_fused_small<float, false>(r, none, st);
        if (r.x_dtype == PRAG_F16) return launch_fused_small<_Float16, false>(r, none, st);
        return launch_fused_small<unsigned short, false>(r, none, st);
    }
    if (r.x_dtype == PRAG_F32) launch_fc1<float>(r, st);
    else if (r.x_dtype == PRAG_F16) launch_fc1<_Float16>(r, st);
    else launch_fc1<unsigned short>(r, st);
    PRAG_LAUNCH_CHECK();
    const dim3 grid(kSmH / 16, r.n_run), block(256);
    const size_t lds = (size_t)r.B * kSmH * sizeof(float);
    const int64_t hs = (int64_t)r.B * kSmH;
#define PRAG_SM(BM) hipLaunchKernelGGL((small_fc_kernel<float, BM>), grid, block, lds, st, r.layers, r.layer0, 1, \
                                       (const float*)r.h1, hs, r.B, kSmH, r.h2)
    if (r.B <= 1) PRAG_SM(1);
    else if (r.B <= 2) PRAG_SM(2);
    else if (r.B <= 4) PRAG_SM(4);
    else PRAG_SM(8);
#undef PRAG_SM
    PRAG_LAUNCH_CHECK();
    hipLaunchKernelGGL(small_head_kernel, dim3(1), dim3(256), 0, st, r.layers, r.layer0, r.n_run, r.B, (const float*)r.h2,
                       r.logits, r.ablation, r.theta, r.probsum, r.decision);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

}  // namespace prag
