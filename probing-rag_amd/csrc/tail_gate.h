// The gate's prober launch described as data, so that the two-level search can carry it in the launch of its bound
// kernel (flat_shadow.hip bound_gate_kernel; prag_search_and_gate in flat_index.hip).  Reference: the retrieve-decide
// loop alternates generation -> gate -> retrieval (exp_rag.py:396-474); with batches in flight the gate of the next
// batch and the retrieval of this one are independent.
#pragma once

#include "prober_internal.h"

struct prag_prober;

namespace prag {

struct TailGate {
    ProberArgs pa;       // complete arguments of prober16_body (n_tiles and n_run set)
    int ct16 = 0;        // row tile = 16 * ct16 batch rows (2, 4 or 8)
    int n_wg = 0;        // workgroups of the launch (8 * ceil(n_tiles * n_run / 8))
    int lds_bytes = 0;   // p16_lds_bytes<ct16>()
    bool taken = false;  // set by whoever launched it
    bool in_scan = false;   // ... behind the scan's workgroups in the scan's launch (scan8_gate_kernel), not beside the bound kernel
    // exp_rag.py:407-415 over the logits (gate_kernel's arithmetic): what whoever finishes the gate needs.  When the
    // prober ran in the scan's launch the logits are complete before the bound kernel starts, and that launch finishes
    // the gate on workgroups of its own (bound_finish_kernel) - one launch less in the pass.
    float* fin_probsum = nullptr;
    int32_t* fin_decision = nullptr;
    int fin_ablation = 0;
    double fin_theta = 0.0;
    bool finished = false;  // probsum / decision are written (set by the launcher that did it)
    bool gate_folded = false;   // pa carries the gate's outputs: the launch also does exp_rag.py:407-415
};

// prober.hip: describe the prober launch of prag_gate(p, x, ...) - false when that call would not run prober16_kernel
// (small batches, float32 activations or weights, the 32 x 32 shape): the caller then runs prag_gate as usual.
bool prober_describe_tail(prag_prober* p, const void* x_dev, int x_dtype, int64_t x_layer_stride, int B, float* logits_dev,
                          int ablation, double theta, float* probsum_dev, int32_t* decision_dev, TailGate* out);

}  // namespace prag
