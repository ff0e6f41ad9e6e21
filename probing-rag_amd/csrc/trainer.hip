// Prober training step on MI355X (gfx950): the replacement for the reference's
//   loss, logit = _method_2_util(model, activations, labels, pred_lens)   (train.py:199-208, utils.py:181-189)
//   loss.backward(); optim.step(); scheduler.step(); optim.zero_grad()    (train.py:210-220, utils.py:191-197)
// with ImprovedProbe in train mode (dropout p = 0.1 after both hidden LayerNorms, utils.py:39-56),
// CrossEntropyLoss applied to the softmax PROBABILITIES (train.py:149-150), torch.optim.AdamW and
// ExponentialLR(gamma = 0.995) (train.py:131-135).
//
// Everything is fp32 like the reference.  A step is launch- and weight-traffic-bound (B = 8 in
// train_prober.sh: 50 MFLOP against 1.3 M parameters with two moments each), so the design goal
// is to touch every parameter once: gradients of the weight matrices are never materialised -
// adamw_matrix_kernel recomputes dW[n,k] = sum_b dOut[b,n] * In[b,k] (B fused multiply-adds) in
// the thread that updates W[n,k], exp_avg[n,k], exp_avg_sq[n,k].  All reductions run in a fixed
// order (no float atomics): a step is reproducible bit for bit.
// Dropout masks come from a counter hash keyed by (seed, step, site, row, unit) - identical to
// oracle_np.dropout_keep, which is also what the golden generator feeds the reference module.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "prag_common.h"

namespace prag {

constexpr float kLnEps = 1e-5f;  // torch.nn.LayerNorm default

__host__ __device__ __forceinline__ uint32_t tmix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[w] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// ---- forward ---------------------------------------------------------------
// LayerNorm of the input rows: xh = (x - mu) * rstd, y = xh * g + b.  grid = B, block = 256.
__global__ __launch_bounds__(256) void tr_ln_input_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ b, int d,
                                                         float* __restrict__ xh, float* __restrict__ y) {
    __shared__ float s_red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + (int64_t)row * d;
    float s = 0.f;
    for (int c = tid; c < d; c += 256) s += xr[c];
    const float mu = block_sum_256(s, s_red) / (float)d;
    float q = 0.f;
    for (int c = tid; c < d; c += 256) {
        const float t = xr[c] - mu;
        q = fmaf(t, t, q);
    }
    const float rstd = 1.0f / sqrtf(block_sum_256(q, s_red) / (float)d + kLnEps);
    for (int c = tid; c < d; c += 256) {
        const float h = (xr[c] - mu) * rstd;
        xh[(int64_t)row * d + c] = h;
        y[(int64_t)row * d + c] = fmaf(h, g[c], b[c]);
    }
}

// out[b,n] = W[n,:] . in[b,:] + bias[n]: one wave per output unit n, 8 rows at a time.
__global__ __launch_bounds__(256) void tr_fc_forward_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                           const float* __restrict__ in, int B, int K, int N,
                                                           float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = W + (int64_t)n * K;
    for (int b0 = 0; b0 < B; b0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = lane * 4; k < K; k += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (b0 + j < B) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(in + (int64_t)(b0 + j) * K + k);
                    acc[j] = fmaf(wv[0], xv[0], fmaf(wv[1], xv[1], fmaf(wv[2], xv[2], fmaf(wv[3], xv[3], acc[j]))));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0 && b0 + j < B) out[(int64_t)(b0 + j) * N + n] = v + bias[n];
        }
    }
}

__device__ __forceinline__ float dropout_scale_keep(uint32_t site_key, uint32_t thresh, float scale, int row, int n,
                                                    int N) {
    const uint32_t h = tmix32((uint32_t)(row * N + n) ^ site_key);
    return h >= thresh ? scale : 0.0f;
}

// s = silu(h); LayerNorm over the row; dropout.  Keeps sh = normalised s and rstd for backward.
// grid = B, block = 256.
__global__ __launch_bounds__(256) void tr_act_ln_dropout_kernel(const float* __restrict__ h, const float* __restrict__ g,
                                                               const float* __restrict__ be, int N, uint32_t site_key,
                                                               uint32_t thresh, float scale, float* __restrict__ sh,
                                                               float* __restrict__ rstd_out, float* __restrict__ dout) {
    __shared__ float s_red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* hr = h + (int64_t)row * N;
    float s = 0.f;
    for (int c = tid; c < N; c += 256) {
        const float v = hr[c];
        s += v / (1.0f + expf(-v));
    }
    const float mu = block_sum_256(s, s_red) / (float)N;
    float q = 0.f;
    for (int c = tid; c < N; c += 256) {
        const float v = hr[c];
        const float t = v / (1.0f + expf(-v)) - mu;
        q = fmaf(t, t, q);
    }
    const float rstd = 1.0f / sqrtf(block_sum_256(q, s_red) / (float)N + kLnEps);
    if (tid == 0) rstd_out[row] = rstd;
    for (int c = tid; c < N; c += 256) {
        const float v = hr[c];
        const float z = (v / (1.0f + expf(-v)) - mu) * rstd;
        sh[(int64_t)row * N + c] = z;
        const float keep = thresh ? dropout_scale_keep(site_key, thresh, scale, row, c, N) : 1.0f;
        dout[(int64_t)row * N + c] = fmaf(z, g[c], be[c]) * keep;
    }
}

// probabilities, double-softmax cross entropy (mean over the batch) and dL/dlogits.  One block.
__global__ __launch_bounds__(256) void tr_loss_kernel(const float* __restrict__ z, const int* __restrict__ labels, int B,
                                                     int C, float* __restrict__ probs, float* __restrict__ dz,
                                                     float* __restrict__ loss_out) {
    __shared__ float s_red[4];
    float local = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* zr = z + (int64_t)b * C;
        float mx = zr[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, zr[c]);
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += expf(zr[c] - mx);
        float p[16], e2 = 0.f;
        for (int c = 0; c < C; ++c) {
            p[c] = expf(zr[c] - mx) / den;
            probs[(int64_t)b * C + c] = p[c];
            e2 += expf(p[c]);
        }
        // labels are validated by the host layer (torch's CrossEntropyLoss raises on a class outside
        // [0,C)); the clamp only keeps a stray value from indexing outside p[]
        const int y = min(max(labels[b], 0), C - 1);
        local += logf(e2) - p[y];
        // d loss / d p = (softmax(p) - onehot) / B ; through the first softmax: dz = p * (gp - sum(gp * p))
        float gp[16], dot = 0.f;
        for (int c = 0; c < C; ++c) {
            gp[c] = (expf(p[c]) / e2 - (c == y ? 1.0f : 0.0f)) / (float)B;
            dot = fmaf(gp[c], p[c], dot);
        }
        for (int c = 0; c < C; ++c) dz[(int64_t)b * C + c] = p[c] * (gp[c] - dot);
    }
    const float total = block_sum_256(local, s_red);
    if (threadIdx.x == 0) *loss_out = total / (float)B;
}

// ---- backward ----------------------------------------------------------------
// din[b,k] = sum_n dout[b,n] * W[n,k].  grid = K / 64, block = 1024: a block owns 64 consecutive k
// (one 256-B segment of every W row per wave load); its 16 waves split the n range, keep 8 row
// loads of W in flight, read dout from LDS (a uniform global load per multiply was the whole cost
// of the first version) and their partial sums are added in wave order (deterministic).
__global__ __launch_bounds__(1024) void tr_fc_backward_input_kernel(const float* __restrict__ dout,
                                                                   const float* __restrict__ W, int B, int K, int N,
                                                                   float* __restrict__ din) {
    extern __shared__ float s_dyn[];            // [8][N] dout rows, then [16][8][64] partial sums
    float* s_do = s_dyn;
    float* s_part = s_dyn + 8 * N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    const int per = (N + 15) / 16;
    const int n0 = w * per < N ? w * per : N, n1 = n0 + per < N ? n0 + per : N;
    for (int b0 = 0; b0 < B; b0 += 8) {
        const int nb = B - b0 < 8 ? B - b0 : 8;
        __syncthreads();
        for (int i = threadIdx.x; i < 8 * N; i += 1024) s_do[i] = i < nb * N ? dout[(int64_t)b0 * N + i] : 0.f;
        __syncthreads();
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int n = n0;
        for (; n + 8 <= n1; n += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = W[(int64_t)(n + u) * K + k];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(s_do[j * N + n + u], wv[u], acc[j]);
        }
        for (; n < n1; ++n) {
            const float wv = W[(int64_t)n * K + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(s_do[j * N + n], wv, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s_part[(w * 8 + j) * 64 + lane] = acc[j];
        __syncthreads();
        if (w < nb) {  // wave j finishes row b0 + j
            float t = 0.f;
            for (int u = 0; u < 16; ++u) t += s_part[(u * 8 + w) * 64 + lane];
            din[(int64_t)(b0 + w) * K + k] = t;
        }
    }
}

// through dropout, LayerNorm and SiLU of one hidden layer:
//   dn = dd * keep (stored over dd: the LayerNorm parameter gradients need it), dsh = dn * g,
//   ds = rstd * (dsh - mean(dsh) - sh * mean(dsh * sh)),  dh = ds * silu'(h).   grid = B, block = 256.
__global__ __launch_bounds__(256) void tr_ln_act_backward_kernel(float* __restrict__ dd, const float* __restrict__ g,
                                                                const float* __restrict__ sh,
                                                                const float* __restrict__ rstd_in,
                                                                const float* __restrict__ h, int N, uint32_t site_key,
                                                                uint32_t thresh, float scale, float* __restrict__ dh) {
    __shared__ float s_red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float a = 0.f, bsum = 0.f;
    for (int c = tid; c < N; c += 256) {
        const int64_t i = (int64_t)row * N + c;
        const float keep = thresh ? dropout_scale_keep(site_key, thresh, scale, row, c, N) : 1.0f;
        const float dn = dd[i] * keep;
        dd[i] = dn;
        const float dsh = dn * g[c];
        a += dsh;
        bsum = fmaf(dsh, sh[i], bsum);
    }
    const float m1 = block_sum_256(a, s_red) / (float)N;
    const float m2 = block_sum_256(bsum, s_red) / (float)N;
    const float rstd = rstd_in[row];
    for (int c = tid; c < N; c += 256) {
        const int64_t i = (int64_t)row * N + c;
        const float ds = rstd * (dd[i] * g[c] - m1 - sh[i] * m2);
        const float v = h[i];
        const float sg = 1.0f / (1.0f + expf(-v));
        dh[i] = ds * sg * (1.0f + v * (1.0f - sg));
    }
}

struct AdamStep {
    float decay;      // 1 - lr * weight_decay
    float beta1, beta2;
    float step_size;  // lr / (1 - beta1^t)
    float bc2_sqrt;   // sqrt(1 - beta2^t)
    float eps;
};

__device__ __forceinline__ void adamw_apply(float& p, float& m, float& v, float g, const AdamStep& a) {
    p *= a.decay;                              // param.mul_(1 - lr * weight_decay)
    m = fmaf(g - m, 1.0f - a.beta1, m);        // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(v, a.beta2, (1.0f - a.beta2) * g * g);
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= a.step_size * (m / denom);            // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// weight matrices: gradient on the fly, one thread per element, k fastest (coalesced).
__global__ __launch_bounds__(256) void tr_adamw_matrix_kernel(float* __restrict__ W, float* __restrict__ m,
                                                             float* __restrict__ v, const float* __restrict__ dout,
                                                             const float* __restrict__ in, int B, int K, int N,
                                                             AdamStep a) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)N * K) return;
    const int n = (int)(e / K), k = (int)(e - (int64_t)n * K);
    float g = 0.f;
    for (int b = 0; b < B; ++b) g = fmaf(dout[(int64_t)b * N + n], in[(int64_t)b * K + k], g);
    float p = W[e], mm = m[e], vv = v[e];
    adamw_apply(p, mm, vv, g, a);
    W[e] = p;
    m[e] = mm;
    v[e] = vv;
}

// vectors (biases, LayerNorm affines): g[j] = sum_b a[b,j] * (mult ? mult[b,j] : 1).  All nine
// vectors of the model in one launch: blockIdx.y picks the vector.
struct VecJob {
    float* p;
    float* m;
    float* v;
    const float* a;
    const float* mult;
    int n;
};
struct VecJobs {
    VecJob j[9];
};
__global__ __launch_bounds__(256) void tr_adamw_vectors_kernel(VecJobs jobs, int B, AdamStep a) {
    const VecJob job = jobs.j[blockIdx.y];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= job.n) return;
    float g = 0.f;
    for (int b = 0; b < B; ++b) {
        const float t = job.a[(int64_t)b * job.n + j];
        g = job.mult ? fmaf(t, job.mult[(int64_t)b * job.n + j], g) : g + t;
    }
    float pp = job.p[j], mm = job.m[j], vv = job.v[j];
    adamw_apply(pp, mm, vv, g, a);
    job.p[j] = pp;
    job.m[j] = mm;
    job.v[j] = vv;
}

}  // namespace prag

// ===========================================================================
// host side
// ===========================================================================
using namespace prag;

struct prag_trainer {
    int d, H, C;
    double lr0, beta1, beta2, eps, weight_decay, gamma, dropout_p;
    uint32_t seed;
    int64_t step = 0;  // optimiser steps taken
    bool loaded = false;
    bool training = true;  // false: dropout is the identity (`probe.eval()`)
    // parameters / moments: one buffer each, state-dict order
    float* params = nullptr;
    float* exp_avg = nullptr;
    float* exp_avg_sq = nullptr;
    size_t off[12], len[12], total = 0;
    // activations for up to b_cap rows
    int b_cap = 0;
    float* ws = nullptr;
    int* labels = nullptr;
    float *xh0, *y0, *h1, *sh1, *r1, *d1, *h2, *sh2, *r2, *d2, *z, *probs, *dz, *dd2, *dh2, *dd1, *dh1, *dy0, *loss;
};

enum { P_LN0_W, P_LN0_B, P_W1, P_B1, P_LN1_W, P_LN1_B, P_W2, P_B2, P_LN2_W, P_LN2_B, P_W3, P_B3 };

extern "C" int prag_trainer_create(prag_trainer_t** out, int d_model, int d_hidden, int n_classes, double lr,
                                   double beta1, double beta2, double eps, double weight_decay, double gamma,
                                   double dropout_p, uint32_t seed) {
    PRAG_REQUIRE(out != nullptr, PRAG_EINVAL, "prag_trainer_create: out is NULL");
    PRAG_REQUIRE(d_model >= 256 && d_model % 256 == 0 && d_hidden >= 256 && d_hidden % 256 == 0, PRAG_EUNSUPPORTED,
                 "d_model=%d d_hidden=%d: multiples of 256 only", d_model, d_hidden);
    PRAG_REQUIRE(n_classes >= 2 && n_classes <= 16, PRAG_EUNSUPPORTED, "n_classes=%d outside [2,16]", n_classes);
    PRAG_REQUIRE(dropout_p >= 0.0 && dropout_p < 1.0 && lr > 0.0 && gamma > 0.0, PRAG_EINVAL,
                 "dropout_p=%g lr=%g gamma=%g", dropout_p, lr, gamma);
    prag_trainer* t = new (std::nothrow) prag_trainer();
    PRAG_REQUIRE(t != nullptr, PRAG_ENOMEM, "out of host memory");
    t->d = d_model;
    t->H = d_hidden;
    t->C = n_classes;
    t->lr0 = lr;
    t->beta1 = beta1;
    t->beta2 = beta2;
    t->eps = eps;
    t->weight_decay = weight_decay;
    t->gamma = gamma;
    t->dropout_p = dropout_p;
    t->seed = seed;
    const size_t d = d_model, H = d_hidden, C = n_classes;
    const size_t lens[12] = {d, d, H * d, H, H, H, H * H, H, H, H, C * H, C};
    size_t o = 0;
    for (int i = 0; i < 12; ++i) {
        t->off[i] = o;
        t->len[i] = lens[i];
        o += (lens[i] + 63) / 64 * 64;
    }
    t->total = o;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&t->params), o * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->exp_avg), o * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->exp_avg_sq), o * sizeof(float));
    if (e == hipSuccess) e = hipMemset(t->exp_avg, 0, o * sizeof(float));
    if (e == hipSuccess) e = hipMemset(t->exp_avg_sq, 0, o * sizeof(float));
    if (e != hipSuccess) {
        set_error("prag_trainer_create: %s", hipGetErrorString(e));
        prag_trainer_destroy(t);
        return PRAG_EHIP;
    }
    *out = t;
    return PRAG_OK;
}

// host fp32 tensors in state-dict order (utils.py:302-329): copied; resets the optimiser state
extern "C" int prag_trainer_load(prag_trainer_t* t, const float* ln0_w, const float* ln0_b, const float* W1,
                                 const float* b1, const float* ln1_w, const float* ln1_b, const float* W2,
                                 const float* b2, const float* ln2_w, const float* ln2_b, const float* W3,
                                 const float* b3) {
    PRAG_REQUIRE(t != nullptr, PRAG_EINVAL, "trainer handle is NULL");
    const float* src[12] = {ln0_w, ln0_b, W1, b1, ln1_w, ln1_b, W2, b2, ln2_w, ln2_b, W3, b3};
    for (int i = 0; i < 12; ++i) {
        PRAG_REQUIRE(src[i] != nullptr, PRAG_EINVAL, "prag_trainer_load: tensor %d is NULL", i);
        PRAG_HIP(hipMemcpy(t->params + t->off[i], src[i], t->len[i] * sizeof(float), hipMemcpyHostToDevice));
    }
    PRAG_HIP(hipMemset(t->exp_avg, 0, t->total * sizeof(float)));
    PRAG_HIP(hipMemset(t->exp_avg_sq, 0, t->total * sizeof(float)));
    t->step = 0;
    t->loaded = true;
    return PRAG_OK;
}

extern "C" int prag_trainer_export(prag_trainer_t* t, float* ln0_w, float* ln0_b, float* W1, float* b1, float* ln1_w,
                                   float* ln1_b, float* W2, float* b2, float* ln2_w, float* ln2_b, float* W3,
                                   float* b3) {
    PRAG_REQUIRE(t != nullptr && t->loaded, PRAG_ESTATE, "prag_trainer_export before prag_trainer_load");
    float* dst[12] = {ln0_w, ln0_b, W1, b1, ln1_w, ln1_b, W2, b2, ln2_w, ln2_b, W3, b3};
    PRAG_HIP(hipDeviceSynchronize());
    for (int i = 0; i < 12; ++i) {
        PRAG_REQUIRE(dst[i] != nullptr, PRAG_EINVAL, "prag_trainer_export: tensor %d is NULL", i);
        PRAG_HIP(hipMemcpy(dst[i], t->params + t->off[i], t->len[i] * sizeof(float), hipMemcpyDeviceToHost));
    }
    return PRAG_OK;
}

extern "C" double prag_trainer_lr(const prag_trainer_t* t) {
    return t ? t->lr0 * pow(t->gamma, (double)t->step) : -1.0;  // what the next step will use
}
extern "C" int64_t prag_trainer_steps(const prag_trainer_t* t) { return t ? t->step : -1; }

extern "C" int prag_trainer_set_training(prag_trainer_t* t, int training) {
    PRAG_REQUIRE(t != nullptr, PRAG_EINVAL, "trainer handle is NULL");
    t->training = training != 0;
    return PRAG_OK;
}

static int trainer_reserve(prag_trainer* t, int B) {
    if (B <= t->b_cap) return PRAG_OK;
    if (t->ws) (void)hipFree(t->ws);
    if (t->labels) (void)hipFree(t->labels);
    t->ws = nullptr;
    t->labels = nullptr;
    t->b_cap = 0;
    const size_t d = t->d, H = t->H, C = t->C, b = (size_t)(B + 7) / 8 * 8;
    const size_t need = b * (3 * d + 10 * H + 3 * C + 2) + 64;
    PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&t->ws), need * sizeof(float)));
    PRAG_HIP(hipMalloc(reinterpret_cast<void**>(&t->labels), b * sizeof(int)));
    float* p = t->ws;
    auto take = [&](size_t n) { float* r = p; p += n; return r; };
    t->xh0 = take(b * d); t->y0 = take(b * d); t->dy0 = take(b * d);
    t->h1 = take(b * H); t->sh1 = take(b * H); t->d1 = take(b * H); t->dd1 = take(b * H); t->dh1 = take(b * H);
    t->h2 = take(b * H); t->sh2 = take(b * H); t->d2 = take(b * H); t->dd2 = take(b * H); t->dh2 = take(b * H);
    t->r1 = take(b); t->r2 = take(b);
    t->z = take(b * C); t->probs = take(b * C); t->dz = take(b * C);
    t->loss = take(64);
    t->b_cap = (int)b;
    return PRAG_OK;
}

// One optimiser step on x_dev [B,d_model] fp32 (the pooled hidden states) and labels_dev int32 [B].
// loss_dev (1 float) and probs_dev [B,n_classes] are optional device outputs of the forward pass
// (what _method_2_util returns).  Everything is enqueued on `stream`.
extern "C" int prag_trainer_step(prag_trainer_t* t, const float* x_dev, const int32_t* labels_dev, int B,
                                 float* loss_dev, float* probs_dev, void* stream) {
    PRAG_REQUIRE(t != nullptr && t->loaded, PRAG_ESTATE, "prag_trainer_step before prag_trainer_load");
    PRAG_REQUIRE(x_dev && labels_dev && B >= 1, PRAG_EINVAL, "prag_trainer_step: NULL input or B=%d", B);
    int rc = trainer_reserve(t, B);
    if (rc != PRAG_OK) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int d = t->d, H = t->H, C = t->C;
    float* P = t->params;
    auto par = [&](int i) { return P + t->off[i]; };
    const int64_t step = t->step + 1;
    // dropout keys (oracle_np.dropout_keep)
    const uint32_t key = tmix32(t->seed ^ (uint32_t)((uint64_t)step * 0x9E3779B9ull));
    const uint32_t site_key[2] = {tmix32(key ^ 1u), tmix32(key ^ 2u)};
    uint32_t thresh = 0;
    float scale = 1.0f;
    if (t->training && t->dropout_p > 0.0) {
        const double th = nearbyint(t->dropout_p * 4294967296.0);
        thresh = th >= 4294967295.0 ? 4294967295u : (uint32_t)th;
        scale = (float)(1.0 / (1.0 - t->dropout_p));
    }
    // ---- forward ----
    hipLaunchKernelGGL(tr_ln_input_kernel, dim3(B), dim3(256), 0, st, x_dev, par(P_LN0_W), par(P_LN0_B), d, t->xh0, t->y0);
    hipLaunchKernelGGL(tr_fc_forward_kernel, dim3((H + 3) / 4), dim3(256), 0, st, par(P_W1), par(P_B1), t->y0, B, d, H, t->h1);
    hipLaunchKernelGGL(tr_act_ln_dropout_kernel, dim3(B), dim3(256), 0, st, t->h1, par(P_LN1_W), par(P_LN1_B), H,
                       site_key[0], thresh, scale, t->sh1, t->r1, t->d1);
    hipLaunchKernelGGL(tr_fc_forward_kernel, dim3((H + 3) / 4), dim3(256), 0, st, par(P_W2), par(P_B2), t->d1, B, H, H, t->h2);
    hipLaunchKernelGGL(tr_act_ln_dropout_kernel, dim3(B), dim3(256), 0, st, t->h2, par(P_LN2_W), par(P_LN2_B), H,
                       site_key[1], thresh, scale, t->sh2, t->r2, t->d2);
    hipLaunchKernelGGL(tr_fc_forward_kernel, dim3((C + 3) / 4), dim3(256), 0, st, par(P_W3), par(P_B3), t->d2, B, H, C, t->z);
    hipLaunchKernelGGL(tr_loss_kernel, dim3(1), dim3(256), 0, st, t->z, labels_dev, B, C, t->probs, t->dz, t->loss);
    if (loss_dev) PRAG_HIP(hipMemcpyAsync(loss_dev, t->loss, sizeof(float), hipMemcpyDeviceToDevice, st));
    if (probs_dev) PRAG_HIP(hipMemcpyAsync(probs_dev, t->probs, (size_t)B * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    // eval mode (`probe.eval()`, train.py:299): the reference's validation pass is a forward only - no backward,
    // no optimiser step, no scheduler step; a validation loop written against step() must not train on the dev set
    if (!t->training) return PRAG_OK;
    // ---- backward (activations first: every weight is still the pre-step value) ----
    const size_t part_lds = 16 * 8 * 64 * sizeof(float);
    hipLaunchKernelGGL(tr_fc_backward_input_kernel, dim3(H / 64), dim3(1024), 8 * C * sizeof(float) + part_lds, st, t->dz,
                       par(P_W3), B, H, C, t->dd2);
    hipLaunchKernelGGL(tr_ln_act_backward_kernel, dim3(B), dim3(256), 0, st, t->dd2, par(P_LN2_W), t->sh2, t->r2, t->h2, H,
                       site_key[1], thresh, scale, t->dh2);
    hipLaunchKernelGGL(tr_fc_backward_input_kernel, dim3(H / 64), dim3(1024), 8 * H * sizeof(float) + part_lds, st, t->dh2,
                       par(P_W2), B, H, H, t->dd1);
    hipLaunchKernelGGL(tr_ln_act_backward_kernel, dim3(B), dim3(256), 0, st, t->dd1, par(P_LN1_W), t->sh1, t->r1, t->h1, H,
                       site_key[0], thresh, scale, t->dh1);
    hipLaunchKernelGGL(tr_fc_backward_input_kernel, dim3(d / 64), dim3(1024), 8 * H * sizeof(float) + part_lds, st, t->dh1,
                       par(P_W1), B, d, H, t->dy0);
    // ---- AdamW (torch.optim.AdamW.step), lr of ExponentialLR after `step - 1` scheduler steps ----
    const double lr = t->lr0 * pow(t->gamma, (double)(step - 1));
    AdamStep a;
    a.decay = (float)(1.0 - lr * t->weight_decay);
    a.beta1 = (float)t->beta1;
    a.beta2 = (float)t->beta2;
    a.step_size = (float)(lr / (1.0 - pow(t->beta1, (double)step)));
    a.bc2_sqrt = (float)sqrt(1.0 - pow(t->beta2, (double)step));
    a.eps = (float)t->eps;
    float* M = t->exp_avg;
    float* V = t->exp_avg_sq;
    auto mat = [&](int i, const float* dout, const float* in, int K, int N) {
        hipLaunchKernelGGL(tr_adamw_matrix_kernel, dim3((unsigned)(((size_t)N * K + 255) / 256)), dim3(256), 0, st,
                           P + t->off[i], M + t->off[i], V + t->off[i], dout, in, B, K, N, a);
    };
    VecJobs jobs;
    int nj = 0;
    auto vec = [&](int i, const float* g, const float* mult, int n) {
        jobs.j[nj++] = VecJob{P + t->off[i], M + t->off[i], V + t->off[i], g, mult, n};
    };
    mat(P_W3, t->dz, t->d2, H, C);
    mat(P_W2, t->dh2, t->d1, H, H);
    mat(P_W1, t->dh1, t->y0, d, H);
    vec(P_B3, t->dz, nullptr, C);
    vec(P_LN2_W, t->dd2, t->sh2, H);   // dd2 now holds dn2 (after the dropout mask)
    vec(P_LN2_B, t->dd2, nullptr, H);
    vec(P_B2, t->dh2, nullptr, H);
    vec(P_LN1_W, t->dd1, t->sh1, H);
    vec(P_LN1_B, t->dd1, nullptr, H);
    vec(P_B1, t->dh1, nullptr, H);
    vec(P_LN0_W, t->dy0, t->xh0, d);
    vec(P_LN0_B, t->dy0, nullptr, d);
    hipLaunchKernelGGL(tr_adamw_vectors_kernel, dim3((std::max(d, H) + 255) / 256, 9), dim3(256), 0, st, jobs, B, a);
    PRAG_LAUNCH_CHECK();
    t->step = step;
    return PRAG_OK;
}

extern "C" void prag_trainer_destroy(prag_trainer_t* t) {
    if (!t) return;
    void* ptrs[] = {t->params, t->exp_avg, t->exp_avg_sq, t->ws, t->labels};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete t;
}
