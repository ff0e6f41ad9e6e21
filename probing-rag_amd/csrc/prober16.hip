// Fused prober ensemble on 16 x 16 MFMA tiles (fp16 weights, fp16 activations: the throughput mode).
//
// Same computation and the same skeleton as prober_fused_kernel (prober.hip; reference: utils.py:29-57 ImprovedProbe,
// exp_rag.py:381-389) - transposed GEMMs with the weights on the MFMA A operand straight from global memory, raw fp16
// activations through a swizzled 4-stage LDS ring, LayerNorms folded into the epilogues, fc1's accumulators handed to
// fc2 as ready-made B fragments (fp16 hi term + fp8 lo term) - on v_mfma_f32_16x16x32_f16 and
// v_mfma_scale_f32_16x16x128_f8f6f4 instead of the 32 x 32 forms.  Why: the kernel is power-limited, not issue-limited
// (tools/micro/mfma_shape.hip, profiles/r04a_mfma_shape_clock.txt: with the same loop at the same cycles per K step
// the chip holds 2.23 GHz on 16 x 16 tiles against 2.01 GHz on 32 x 32; the tiled scans gained 11-13 % from the same
// switch, profiles/r04g_mm_shape_ab.txt).  cdna_hip_programming.md section 5.4 rule 28.
//
// Geometry: one workgroup = 8 waves = ROWS = 16 * CT16 batch rows of one layer (CT16 = 2, 4, 8); wave w owns hidden
// units [64 w, 64 w + 64) = four 16-unit tiles.  An accumulator tile f32x4 of lane (c16 = lane & 15, q4 = lane >> 4):
// element e <-> hidden unit 64 w + 16 ht + 4 q4 + e, batch row 16 ct + c16.
//   fc1   K-32 step s of hidden tile g: one 1-KiB fragment, lane (c16, q4) holds W1~[16 g + c16][32 s + 8 q4 + j]
//   fc2   K-32 step ks = 2 w' + p consumes the SiLU outputs wave w' holds in its tiles 2 p, 2 p + 1: element j of lane
//         (c16, q4) is hidden unit 64 w' + 16 (2 p + (j >> 2)) + 4 q4 + (j & 3); W2 is packed in that k order
//   lo    k block kb = hidden units [128 kb, 128 kb + 128): byte 16 hf + 4 ht + e of lane (c16, q4) is hidden unit
//         128 kb + 64 hf + 16 ht + 4 q4 + e (waves 2 kb + hf); the fp8 copy of W2 is packed the same way
//         (tools/micro/mfma16_probe.hip checks that operand bytes pair up by (lane quarter, byte))
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "prag_common.h"
#include "prober_internal.h"

namespace prag {

#ifdef PRAG_MM_DIAG
// timing-only build (make diag): s_memtime / s_memrealtime stamps of three workgroups, read by tools/prober_stamps.py
__device__ unsigned long long g_pstamp16[2 * 3 * 8 * 32];
#define PSTAMP(i)                                                                                                  \
    if (ps_sel >= 0) {                                                                                             \
        unsigned long long t_, r_;                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        if (lane == 0) {                                                                                           \
            g_pstamp16[(ps_sel * 8 + w) * 32 + (i)] = t_;                                                          \
            g_pstamp16[3 * 8 * 32 + (ps_sel * 8 + w) * 32 + (i)] = r_;                                             \
        }                                                                                                          \
    }
#define P16_ABL(bit) (a.ablate & (bit))
#else
#define PSTAMP(i)
#define P16_ABL(bit) 0
#endif

}  // namespace prag

#include "prober16_body.h"

namespace prag {

template <int CT16>
__global__ __launch_bounds__(512, 2) void prober16_kernel(ProberArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    prober16_body<CT16>(a, smem, (int)blockIdx.x);
}


#ifdef PRAG_MM_DIAG
extern "C" int prag_diag_prober16_stamps(unsigned long long* out, int n) {
    if (n > 2 * 3 * 8 * 32) n = 2 * 3 * 8 * 32;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamp16), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -2;
}
#endif

template <int CT16>
static int launch_p16(const ProberArgs& a, int n_run, hipStream_t st, EventRing& prof) {
    constexpr int LDS = p16_lds_bytes<CT16>();
    static_assert(LDS <= 160 * 1024, "workgroup LDS");
    auto kern = prober16_kernel<CT16>;
    static LdsOptIn lds_opt_in;
    {
        const int rc_ = lds_opt_in.ensure(reinterpret_cast<const void*>(kern), LDS);
        if (rc_ != PRAG_OK) return rc_;
    }
    ProberArgs b = a;
#ifdef PRAG_MM_DIAG
    b.stamps = getenv("PRAG_PROBER_STAMPS") ? atoi(getenv("PRAG_PROBER_STAMPS")) : 0;
    b.ablate = getenv("PRAG_PROBER_ABLATE") ? atoi(getenv("PRAG_PROBER_ABLATE")) : 0;
#endif
    b.n_tiles = (a.B + 16 * CT16 - 1) / (16 * CT16);
    b.n_run = n_run;
    const int per_xcd = (b.n_tiles * n_run + 7) / 8;
    prof.begin(st);
    hipLaunchKernelGGL(kern, dim3(8 * per_xcd), dim3(512), LDS, st, b);
    prof.end(st);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

int prober16_lds_bytes(int ct16) {
    return ct16 == 2 ? p16_lds_bytes<2>() : ct16 == 4 ? p16_lds_bytes<4>() : ct16 == 8 ? p16_lds_bytes<8>() : 0;
}

int prober16_launch(const ProberArgs& a, int n_run, int rows_per_tile, hipStream_t st, EventRing& prof) {
    switch (rows_per_tile) {
        case 32: return launch_p16<2>(a, n_run, st, prof);
        case 64: return launch_p16<4>(a, n_run, st, prof);
        case 128: return launch_p16<8>(a, n_run, st, prof);
    }
    set_error("internal: prober16_launch rows_per_tile=%d", rows_per_tile);
    return PRAG_EUNSUPPORTED;
}

}  // namespace prag
