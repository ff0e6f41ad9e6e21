// Small-batch prober path (prober_small.hip): the reference's own call shape is ONE pooled state per
// probed layer (exp_rag.py:381-389, batch_size 1).  At a handful of rows the MFMA-tiled kernel has one
// workgroup per layer pulling that layer's 5 MB of weights through a single CU (~60 us); here the
// weight rows are spread over the whole chip instead: three short launches of plain f32 dot products.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace prag {

struct SmallLayer {        // LayerNorm-folded ("effective") weights of one layer, f32, row-major [out][in]
    const float* W1;       // [512][d]   W1 * diag(ln0_w)
    const float* b1;       // [512]      b1 + W1 . ln0_b
    const float* W2;       // [512][512] W2 * diag(ln1_w)
    const float* b2;       // [512]
    const float* W3;       // [2][512]   W3 * diag(ln2_w)
    float b3[2];
};

struct SmallRun {
    const SmallLayer* layers;   // device array, all layers of the handle
    const void* x;              // activations of layer `layer0` onwards
    int x_dtype;                // PRAG_F32 | PRAG_F16 | PRAG_BF16
    int64_t x_layer_stride;
    int layer0, n_run, B, d;
    float* h1;                  // workspace [n_run][B][512]
    float* h2;                  // workspace [n_run][B][512]
    float* logits;              // out [n_run][B][2]
    // gate fused into the last launch (exp_rag.py:407-415), only when probsum/decision are given
    int ablation;
    double theta;
    float* probsum;             // [B][2] or null
    int32_t* decision;          // [B] or null
    void* sync = nullptr;       // device block of small_sync_bytes() bytes, zero between launches (one-launch form)
    int fused = 0;              // 1: one launch (needs `sync`; measured slower, prober.hip); 0: three launches
};

constexpr int kSmallMaxB = 8;
constexpr int kSmallStepMaxLayers = 16;   // layers of one prag_pool_step_gate launch (pointer table passed by value)

// One decode step of the hooked layers + the gate on the sums so far (prag_pool_step_gate)
struct SmallStep {
    const void* const* h;       // host array [n_run] of device pointers: each layer's activations of this step, [B][d]
    const float* acc_in;        // device [n_run][B][d]: sums before this step (not read when assign)
    float* acc_out;             // device [n_run][B][d]: sums after it; a different buffer
    int assign;
    uint64_t tag;               // > 0, < 2^63
    uint64_t* host_dev;         // device view of the mapped pinned block: tag | int32 decision[8] | float probsum[8][2]
};
int small_step_run(const SmallRun& r, const SmallStep& sp, hipStream_t st);
constexpr int kSmallMaxElems = 16384;   // B * d staged in LDS as f32 (64 KiB)
inline bool small_supported(int B, int d) { return B >= 1 && B <= kSmallMaxB && (int64_t)B * d <= kSmallMaxElems; }
// Enqueue the gate on `st`: one launch (r.sync given), or the three launches of rounds 2-3.
int small_run(const SmallRun& r, hipStream_t st);
size_t small_sync_bytes();

}  // namespace prag
