// scan8_gate_kernel: the two-level search's scan over the 8-bit shadow (64-query tiles) and the gate's prober ensemble
// in ONE launch (round 5).  The first n_scan workgroups run scan8_kernel's body, the others prober16_kernel's.
//
// Why: an HBM-bound scan is fastest on about 7/8 of the CUs (profiles/r05u_scan_wg_sweep.txt), so 32 CUs are free for
// the whole scan; the gate of the NEXT batch of generations (exp_rag.py:381-389, 406-415) depends on nothing in this
// retrieval (make_indexer.py:447-457) and takes 0.1-0.5 ms of those CUs while the scan runs 0.35-2.5 ms.  Workgroups
// of one launch are placed in index order, so the scan's settle on their CUs first and the prober's take what is left
// - two launches on two streams give no such order (the prober's 192 workgroups, placed first, hold the CUs the scan's
// statically assigned tiles are waiting for).  The search's tail then runs shadow_bound_kernel alone.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "flat_internal.h"
#include "scan8_body.h"
#include "tail_gate.h"

#define PSTAMP(i)
#define P16_ABL(bit) 0
#include "prober16_body.h"
#undef PSTAMP
#undef P16_ABL

namespace prag {

template <int KC, bool QUAD, int CT16>
__global__ __launch_bounds__(512, 1) void scan8_gate_kernel(Scan8Args a, int n_scan, ProberArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < n_scan)
        scan8_body<64, KC, true, 0, QUAD ? kScan8Aln : 0, QUAD>(a, smem, (int)blockIdx.x, n_scan);
    else
        prober16_body<CT16>(pa, smem, (int)blockIdx.x - n_scan);
}

template <int KC, bool QUAD, int CT16>
static int launch_ct(const Scan8Args& a, int grid, const TailGate& t, hipStream_t st, EventRing& prof) {
    auto kern = scan8_gate_kernel<KC, QUAD, CT16>;
    const int lds = std::max(scan8_lds_bytes(64, a.qstride), t.lds_bytes);
    static LdsOptIn lds_opt_in;
    const int rc_ = lds_opt_in.ensure(reinterpret_cast<const void*>(kern), 160 * 1024);
    if (rc_ != PRAG_OK) return rc_;
    prof.begin(st);
    hipLaunchKernelGGL(kern, dim3(grid + t.n_wg), dim3(512), lds, st, a, grid, t.pa);
    prof.end(st);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

// Round 6: four forms instead of twelve.  Measured under the deterministic plan (profiles/r06d_shard_fused_ab.txt), one
// rank's pass of the 8-GPU job: gate behind the scan's workgroups 0.403-0.428 ms, gate beside the bound kernel 0.417-0.436,
// the two calls 0.443-0.461 - carrying the gate in a launch of the search is worth 6-8 %, WHICH launch carries it ~1 %
// (1.4 % at 21 M rows).  The forms kept are the ones the BASELINE shapes take: 16-deep lists (5 < k <= 12), the 128-row
// prober tile of a >= 2731-row gate and the 32-row tile of a <= 682-row one; every other shape rides beside the bound
// kernel (bound_gate_kernel covers all tile heights).
bool scan8_gate_supported(int kc, int ct16) { return kc == 16 && (ct16 == 2 || ct16 == 8); }

int launch_scan8_gate(const Scan8Args& a, int grid, int kc, bool quad, const TailGate& t, hipStream_t st, EventRing& prof) {
#define PRAG_SG(KC_, Q_, CT_) if (kc == KC_ && quad == Q_ && t.ct16 == CT_) return launch_ct<KC_, Q_, CT_>(a, grid, t, st, prof);
    PRAG_SG(16, true, 8) PRAG_SG(16, true, 2)
    PRAG_SG(16, false, 8) PRAG_SG(16, false, 2)
#undef PRAG_SG
    set_error("internal: scan8_gate_kernel has no (kc %d, %d-row tile) form", kc, 16 * t.ct16);
    return PRAG_EUNSUPPORTED;
}

}  // namespace prag
