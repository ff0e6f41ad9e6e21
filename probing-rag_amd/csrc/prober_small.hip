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
    // ---- LayerNorm (no affine: folded into W / bias) of the B input rows -> LDS -------------------
    for (int b = w; b < B; b += 4) {
        const T* src = in + (int64_t)lrun * in_layer_stride + (int64_t)b * K;
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
    __syncthreads();
    // ---- 16 output units per workgroup, 4 per wave, two weight rows in flight ----------------------
    const int n0 = blockIdx.x * 16 + w * 4;
#pragma unroll
    for (int pair = 0; pair < 2; ++pair) {
        const int na = n0 + 2 * pair, nb = na + 1;
        const float* wa = W + (int64_t)na * K;
        const float* wb = W + (int64_t)nb * K;
        float acc_a[B_MAX], acc_b[B_MAX];
#pragma unroll
        for (int b = 0; b < B_MAX; ++b) acc_a[b] = acc_b[b] = 0.f;
        for (int i0 = lane * 4; i0 < K; i0 += 2048) {      // 8 x 16-B loads per row per lane per step
            f32x4 va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                const bool ok = i < K;
                va[u] = ok ? *reinterpret_cast<const f32x4*>(wa + i) : f32x4{0.f, 0.f, 0.f, 0.f};
                vb[u] = ok ? *reinterpret_cast<const f32x4*>(wb + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
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
        }
#pragma unroll
        for (int b = 0; b < B_MAX; ++b) {
            if (b < B) {
                const float ha = wave_sum(acc_a[b]) + bias[na];
                const float hb = wave_sum(acc_b[b]) + bias[nb];
                if (lane == 0) {
                    float* o = out + ((int64_t)lrun * B + b) * kSmH;
                    o[na] = sm_silu(ha);
                    o[nb] = sm_silu(hb);
                }
            }
        }
    }
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
