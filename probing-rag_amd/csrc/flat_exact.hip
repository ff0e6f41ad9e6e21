// Exact float64 brute-force search for the queries the certificate could not clear
// (flat_internal.h "Exactness certificate"), for deep-list overflows and for k > 26 on dimensions
// the MFMA-tiled scan does not cover.  Replaces faiss.IndexFlatL2.search's guarantee
// (/root/reference: make_indexer.py:449-450, utils.py:378-380 - exact brute force, always) on the
// rare inputs where the matrix-core selection cannot prove its own result: rows within the
// selection error of the k-th result (dense near-duplicates, squared-L2 cancellation).
//
// exact_scan_kernel: every workgroup streams its share of the rows once per flagged query
//   (16 lanes per row, direct sum (q - x)^2 or q.x in float64 - no ||x||^2 - 2 q.x cancellation),
//   keeps the rows that can still reach its top k in an LDS buffer (threshold filter against the
//   k-th best (score, id) so far; the buffer is sorted and cut to k when it fills) and writes its k
//   best.  HBM-bound like the list scan: one extra pass over the shard per flagged query.
//   The last workgroup to finish a query folds the per-workgroup lists the same way and writes D (float32
//   rounding of the float64 score) / I (row id + offset), faiss padding.
// The kernel reads the flag count first and returns at once when it is zero (the common case): one ~3 us launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>

#include "flat_internal.h"

namespace prag {

constexpr int kExCap = 2048;     // LDS slots per workgroup (k <= 1024 leaves >= 768 free after a cut)
constexpr int kExCheck = 8;      // row tiles (of 32 rows) between capacity checks
constexpr int kExThreads = 512;

// Row loads of exact_mfma_kernel: ORDINARY loads.  The float64 scans streamed their rows non-temporally like the first-pass
// scans since round 4; on THIS kernel's access pattern - 16 B pieces of 16 rows per wave instruction, two adjacent 16 B loads
// per lane on float32 rows - that cost 8 % (fp16 rows) to 20 % (float32 rows) of the pass (the bare pattern:
// tools/micro/row_piece_stream.hip, 5.85 against 6.34 TB/s); the one-query and eight-query kernels, whose wave instructions
// cover 256 contiguous bytes of a row, are 1-2 % faster WITH the hint and keep it (profiles/r06y_exact_loads_ab.txt).
__device__ __forceinline__ u32x4 exm_load(const u32x4* p) {
#ifdef PRAG_EX_NONTEMPORAL
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

struct ExTopK {
    unsigned long long key[kExCap];
    int id[kExCap];
    unsigned long long bound;    // k-th best key so far (~0: none yet); ties on the key are kept
    int cnt;
};

__device__ __forceinline__ void ex_init(ExTopK& t) {
    if (threadIdx.x == 0) {
        t.bound = ~0ull;
        t.cnt = 0;
    }
}

__device__ __forceinline__ void ex_push(ExTopK& t, unsigned long long key, int id) {
    if (key <= t.bound) {
        const int slot = atomicAdd(&t.cnt, 1);
        t.key[slot] = key;   // callers keep cnt <= kExCap between cuts
        t.id[slot] = id;
    }
}

// sort the buffered entries by (key, id), keep the k best, tighten the bound.  All threads.
__device__ __forceinline__ void ex_cut(ExTopK& t, int k) {
    __syncthreads();
    const int n = t.cnt;
    int n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    for (int i = n + threadIdx.x; i < n_pad; i += kExThreads) {
        t.key[i] = ~0ull;
        t.id[i] = 0x7fffffff;
    }
    for (int size = 2; size <= n_pad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int p = threadIdx.x; p < (n_pad >> 1); p += kExThreads) {
                const int i = ((p / stride) * 2 * stride) + (p % stride), j = i + stride;
                const unsigned long long ka = t.key[i], kb = t.key[j];
                const int ia = t.id[i], ib = t.id[j];
                const bool gt = ka > kb || (ka == kb && ia > ib);
                if (gt == ((i & size) == 0)) {
                    t.key[i] = kb;
                    t.key[j] = ka;
                    t.id[i] = ib;
                    t.id[j] = ia;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        t.cnt = n < k ? n : k;
        if (n >= k) t.bound = t.key[k - 1];
    }
    __syncthreads();
}

struct ExactArgs {
    const void* rows;
    int64_t N;
    int d;
    int metric_l2;
    const float* q32;
    const uint32_t* n_flag;
    const int* flag_list;
    int f0, f_cap;   // this round handles flag slots [f0, f0 + f_cap)
    int k;
    unsigned long long* part_key;   // [f_cap][n_lists][k]
    int* part_id;
    int n_lists;
    int64_t id_offset;
    float* D;
    int64_t* I;
    uint32_t* done;   // [f_cap] workgroups that have written their list of flag slot f (zero between searches)
    unsigned long long* gpool;    // [f_cap][n_lists] exact_mfma_kernel: every workgroup's best key of flag slot f (~0 between searches)
    const uint32_t* xn_max;       // float bits of max_i ||x_i||^2 (exact_mfma_kernel's inner-product margin; null: per-row norms)
    int tag_ids;
    Gate gate;
};

template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double x) {
    return __longlong_as_double((long long)dpp_move_u64<CTRL>((unsigned long long)__double_as_longlong(x)));
}
// sum over the 16 lanes of a DPP row, same association in every row (bitwise reproducible)
__device__ __forceinline__ double dpp_add16_f64(double v) {
    v += dpp_move_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move_f64<0x141>(v);   // row_half_mirror
    v += dpp_move_f64<0x140>(v);   // row_mirror
    return v;
}

// fold the n_lists per-workgroup lists of flag slot `fs` (query b) and write D / I; all threads of the block
__device__ __forceinline__ void exact_merge_lists(const ExactArgs& a, ExTopK& tk, int fs, int b) {
    const int tid = threadIdx.x;
    __syncthreads();
    ex_init(tk);
    __syncthreads();
    const int64_t total = (int64_t)a.n_lists * a.k;
    const int64_t o = (int64_t)fs * total;
    for (int64_t base = 0; base < total; base += kExThreads) {
        const int64_t i = base + tid;
        if (i < total) {
            const int id = a.part_id[o + i];
            if (id != 0x7fffffff) ex_push(tk, a.part_key[o + i], id);
        }
        __syncthreads();
        const int c = tk.cnt;
        __syncthreads();
        if (c > kExCap - kExThreads) ex_cut(tk, a.k);
    }
    ex_cut(tk, a.k);
    for (int j = tid; j < a.k; j += kExThreads) {
        const bool ok = j < tk.cnt;
        const double sc = ok ? unsortable_f64(a.metric_l2 ? tk.key[j] : ~tk.key[j]) : 0.0;
        a.D[(int64_t)b * a.k + j] = ok ? (float)sc : (a.metric_l2 ? FLT_MAX : -FLT_MAX);
        a.I[(int64_t)b * a.k + j] = ok ? tag_id((int64_t)tk.id[j] + a.id_offset, sc, a.tag_ids) : -1;
    }
}

template <bool F32>
__global__ __launch_bounds__(kExThreads) void exact_scan_kernel(ExactArgs a) {
    __shared__ ExTopK tk;
    __shared__ __attribute__((aligned(16))) float s_q[1536];
    __shared__ int s_last;
    if (gate_closed(a.gate)) return;
    const uint32_t nf = *a.n_flag;
    if ((uint32_t)a.f0 >= nf) return;
    const int f1 = (int)std::min<uint32_t>(nf, (uint32_t)(a.f0 + a.f_cap));
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int sub = lane & 15, slot = lane >> 4;
    const int d = a.d;
    const int64_t n_tiles = (a.N + 31) / 32;
    for (int f = a.f0; f < f1; ++f) {
        const int b = a.flag_list[f];
        __syncthreads();
        for (int c = tid; c < d; c += kExThreads) s_q[c] = a.q32[(int64_t)b * d + c];
        ex_init(tk);
        __syncthreads();
        int since = 0;
        for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
            const int64_t row = t * 32 + w * 4 + slot;
            double s = 0.0;
            if (row < a.N) {
                // 8 consecutive elements per lane per step, 16 lanes per row: steps of 128 elements
                constexpr int MAXS = 12;   // d <= 1536
                u32x4 v0[MAXS], v1[MAXS];
#pragma unroll
                for (int it = 0; it < MAXS; ++it) {
                    const int e = sub * 8 + it * 128;
                    if (e < d) {
                        if constexpr (F32) {
                            const float* p = reinterpret_cast<const float*>(a.rows) + row * d + e;
                            v0[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
                            v1[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 4));
                        } else {
                            v0[it] = __builtin_nontemporal_load(
                                reinterpret_cast<const u32x4*>(reinterpret_cast<const _Float16*>(a.rows) + row * d + e));
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < MAXS; ++it) {
                    const int e = sub * 8 + it * 128;
                    if (e < d) {
                        float xv[8];
                        if constexpr (F32) {
                            const f32x4 x0 = __builtin_bit_cast(f32x4, v0[it]);
                            const f32x4 x1 = __builtin_bit_cast(f32x4, v1[it]);
#pragma unroll
                            for (int j = 0; j < 4; ++j) { xv[j] = x0[j]; xv[4 + j] = x1[j]; }
                        } else {
                            const half8 h = __builtin_bit_cast(half8, v0[it]);
#pragma unroll
                            for (int j = 0; j < 8; ++j) xv[j] = (float)h[j];
                        }
                        const f32x4 q0 = *reinterpret_cast<const f32x4*>(s_q + e);
                        const f32x4 q1 = *reinterpret_cast<const f32x4*>(s_q + e + 4);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const double qv = (double)(j < 4 ? q0[j] : q1[j - 4]);
                            const double x = (double)xv[j];
                            if (a.metric_l2) {
                                const double df = qv - x;
                                s = fma(df, df, s);
                            } else {
                                s = fma(qv, x, s);
                            }
                        }
                    }
                }
            }
            s = dpp_add16_f64(s);
            if (sub == 0 && row < a.N)
                ex_push(tk, a.metric_l2 ? sortable_u64(s) : ~sortable_u64(s), (int)row);
            if (++since == kExCheck) {
                since = 0;
                __syncthreads();
                const int c = tk.cnt;
                __syncthreads();
                if (c > kExCap - 32 * kExCheck) ex_cut(tk, a.k);
            }
        }
        ex_cut(tk, a.k);
        const int64_t o = ((int64_t)(f - a.f0) * a.n_lists + blockIdx.x) * a.k;
        for (int j = tid; j < a.k; j += kExThreads) {
            const bool ok = j < tk.cnt;
            a.part_key[o + j] = ok ? tk.key[j] : ~0ull;
            a.part_id[o + j] = ok ? tk.id[j] : 0x7fffffff;
        }
        // the LAST workgroup to finish this query folds the per-workgroup lists (one launch instead of two:
        // with no query flagged the whole exact path is a single early-exit launch).  Only flagged queries get
        // here, so the device-scope fences (L2 write-back / invalidate across XCDs) cost nothing otherwise.
        __threadfence();
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(a.done + (f - a.f0), 1u) == gridDim.x - 1 ? 1 : 0;
        __syncthreads();
        if (s_last) {
            __threadfence();
            exact_merge_lists(a, tk, f - a.f0, b);
            if (tid == 0) a.done[f - a.f0] = 0u;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Grouped form (round 5): up to kExGroup flagged queries per pass over the rows.  A row is loaded and converted to
// float64 ONCE per pass; the queries of the group sit in LDS as float64 (no conversion per use: four ds_read_b128 - the
// same 64 bytes in each of the wave's four 16-lane groups, a broadcast - feed eight v_fma_f64); the per-(element, query)
// work is ONE fma (inner product) or a subtraction and an fma (L2).  tools/micro/f64_valu_probe.hip
// (profiles/r05d_f64_valu_rates.txt): v_fma_f64, v_add_f64, v_cvt_f64_f32 all issue at ~2.3-2.8 ns per wave instruction
// per SIMD and the shared LDS reads hide behind the fmas - 2.6e13 lane-fmas/s over the chip, i.e. eight queries x 21 M x
// 768 in ~6 ms against 8 x 5 ms one by one.  Every lane forms the sums of the single-query kernel in the same order:
// the float64 scores - and D / I - are bit-identical.  One small threshold list per query (kExGroupCap slots, k <=
// kExGroupCap / 4); the group's lists alias the big list the merge of a query needs afterwards.
// The single-query kernel stays: ONE flagged query is the common case, and 100 KB of LDS per workgroup would halve its
// occupancy (what round 3's batched attempt - float32 queries in LDS, a conversion per use - paid: 6.1 ms instead of
// 2.6 for one query).  exact_run takes the grouped kernel when the caller expects several flagged queries
// (ExactRun::grouped: every query flagged by construction, or flags seen in recent searches).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kExGroup = 8;
constexpr int kExGroupCap = 512;

struct ExSmallTopK {
    unsigned long long key[kExGroupCap];
    int id[kExGroupCap];
};

// sort list g of the group by (key, id), keep the k best, tighten its bound.  All threads.
__device__ __forceinline__ void exg_cut(ExSmallTopK& t, int& cnt, unsigned long long& bound, int k) {
    __syncthreads();
    const int n = cnt;
    int n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    for (int i = n + threadIdx.x; i < n_pad; i += kExThreads) {
        t.key[i] = ~0ull;
        t.id[i] = 0x7fffffff;
    }
    for (int size = 2; size <= n_pad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int p = threadIdx.x; p < (n_pad >> 1); p += kExThreads) {
                const int i = ((p / stride) * 2 * stride) + (p % stride), j = i + stride;
                const unsigned long long ka = t.key[i], kb = t.key[j];
                const int ia = t.id[i], ib = t.id[j];
                const bool gt = ka > kb || (ka == kb && ia > ib);
                if (gt == ((i & size) == 0)) {
                    t.key[i] = kb;
                    t.key[j] = ka;
                    t.id[i] = ib;
                    t.id[j] = ia;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt = n < k ? n : k;
        if (n >= k) bound = t.key[k - 1];
    }
    __syncthreads();
}

template <bool F32, int NS>
__global__ __launch_bounds__(kExThreads) void exact_group_kernel(ExactArgs a) {
    extern __shared__ __attribute__((aligned(16))) char ex_smem[];
    // [kExGroup][d] float64 queries | kExGroup small lists (aliased by the merge's big list) | counters
    double* s_qd = reinterpret_cast<double*>(ex_smem);
    char* s_lists = ex_smem + (size_t)kExGroup * a.d * sizeof(double);
    ExSmallTopK* tks = reinterpret_cast<ExSmallTopK*>(s_lists);
    ExTopK& tk_big = *reinterpret_cast<ExTopK*>(s_lists);
    static_assert(sizeof(ExTopK) <= kExGroup * sizeof(ExSmallTopK), "the merge list aliases the group's lists");
    // (counters BEHIND the dynamic region's arrays: a static __shared__ object would sit in front of it and move its base
    //  off the 16-byte alignment the ds_read_b128 of the queries needs - cdna_hip_programming.md Guideline 17)
    unsigned long long* s_bound = reinterpret_cast<unsigned long long*>(s_lists + (size_t)kExGroup * sizeof(ExSmallTopK));
    int* s_cnt = reinterpret_cast<int*>(s_bound + kExGroup);
    int& s_last = s_cnt[kExGroup];
    if (gate_closed(a.gate)) return;
    const uint32_t nf = *a.n_flag;
    if ((uint32_t)a.f0 >= nf) return;
    const int f1 = (int)std::min<uint32_t>(nf, (uint32_t)(a.f0 + a.f_cap));
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int sub = lane & 15, slot = lane >> 4;
    const int d = a.d;
    const int64_t n_tiles = (a.N + 31) / 32;
    for (int fg = a.f0; fg < f1; fg += kExGroup) {
        const int ng = min(kExGroup, f1 - fg);
        __syncthreads();
        for (int i = tid; i < kExGroup * d; i += kExThreads) {
            const int g = i / d, c = i - g * d;
            // (slots past the group's last query repeat the first one: same work, results never written)
            const int b = a.flag_list[fg + (g < ng ? g : 0)];
            s_qd[i] = (double)a.q32[(int64_t)b * d + c];
        }
        if (tid < kExGroup) {
            s_cnt[tid] = 0;
            s_bound[tid] = ~0ull;
        }
        __syncthreads();
        int since = 0;
        // the next tile's row is requested before this tile's is used (one wave holds 8 KB of the stream in flight while
        // its ~700 float64 instructions per tile run: with two waves per SIMD and no prefetch the pass was latency-bound)
        constexpr int MAXS = NS;        // 128-element steps per row: d <= 128 NS
        u32x4 v0[MAXS], v1[F32 ? MAXS : 1], vn0[MAXS], vn1[F32 ? MAXS : 1];
        auto load_row = [&](u32x4 (&r0)[MAXS], u32x4 (&r1)[F32 ? MAXS : 1], int64_t row) __attribute__((always_inline)) {
            const int64_t rc = row < a.N ? row : a.N - 1;          // (past the end: any valid row, never pushed)
#pragma unroll
            for (int it = 0; it < MAXS; ++it) {
                const int e = sub * 8 + it * 128;
                if (e < d) {
                    if constexpr (F32) {
                        const float* p = reinterpret_cast<const float*>(a.rows) + rc * d + e;
                        r0[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
                        r1[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 4));
                    } else {
                        r0[it] = __builtin_nontemporal_load(
                            reinterpret_cast<const u32x4*>(reinterpret_cast<const _Float16*>(a.rows) + rc * d + e));
                    }
                }
            }
        };
        if ((int64_t)blockIdx.x < n_tiles) load_row(v0, v1, (int64_t)blockIdx.x * 32 + w * 4 + slot);
        for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
            const int64_t row = t * 32 + w * 4 + slot;
            const int64_t t_next = t + gridDim.x < n_tiles ? t + gridDim.x : t;
            load_row(vn0, vn1, t_next * 32 + w * 4 + slot);
            double acc[kExGroup];
#pragma unroll
            for (int g = 0; g < kExGroup; ++g) acc[g] = 0.0;
            if (row < a.N) {
#pragma unroll
                for (int it = 0; it < MAXS; ++it) {
                    const int e = sub * 8 + it * 128;
                    if (e < d) {
                        double xd[8];
                        if constexpr (F32) {
                            const f32x4 x0 = __builtin_bit_cast(f32x4, v0[it]);
                            const f32x4 x1 = __builtin_bit_cast(f32x4, v1[it]);
#pragma unroll
                            for (int j = 0; j < 4; ++j) { xd[j] = (double)x0[j]; xd[4 + j] = (double)x1[j]; }
                        } else {
                            const half8 h = __builtin_bit_cast(half8, v0[it]);
#pragma unroll
                            for (int j = 0; j < 8; ++j) xd[j] = (double)(float)h[j];
                        }
                        // element pair u of all eight queries before pair u + 1: eight independent fma chains side by
                        // side (per query the summation order is the single-query kernel's: e, e + 1, ... in turn)
                        typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            d2 qv[kExGroup];
#pragma unroll
                            for (int g = 0; g < kExGroup; ++g)
                                qv[g] = *reinterpret_cast<const d2*>(s_qd + (size_t)g * d + e + 2 * u);
                            if (a.metric_l2) {
#pragma unroll
                                for (int g = 0; g < kExGroup; ++g) {
                                    const double d0 = qv[g][0] - xd[2 * u];
                                    acc[g] = fma(d0, d0, acc[g]);
                                }
#pragma unroll
                                for (int g = 0; g < kExGroup; ++g) {
                                    const double d1 = qv[g][1] - xd[2 * u + 1];
                                    acc[g] = fma(d1, d1, acc[g]);
                                }
                            } else {
#pragma unroll
                                for (int g = 0; g < kExGroup; ++g) acc[g] = fma(qv[g][0], xd[2 * u], acc[g]);
#pragma unroll
                                for (int g = 0; g < kExGroup; ++g) acc[g] = fma(qv[g][1], xd[2 * u + 1], acc[g]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < kExGroup; ++g) {
                const double s = dpp_add16_f64(acc[g]);
                if (sub == 0 && row < a.N && g < ng) {
                    const unsigned long long key = a.metric_l2 ? sortable_u64(s) : ~sortable_u64(s);
                    if (key <= s_bound[g]) {
                        const int sl = atomicAdd(&s_cnt[g], 1);
                        tks[g].key[sl] = key;       // (<= 32 pushes per list and tile; checked every kExCheck tiles)
                        tks[g].id[sl] = (int)row;
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < MAXS; ++it) {
                v0[it] = vn0[it];
                if constexpr (F32) v1[it] = vn1[it];
            }
            if (++since == kExCheck) {
                since = 0;
                __syncthreads();
                for (int g = 0; g < ng; ++g) {        // (uniform: s_cnt is read behind the barrier by every thread)
                    const int c = s_cnt[g];
                    if (c > kExGroupCap - 32 * kExCheck) exg_cut(tks[g], s_cnt[g], s_bound[g], a.k);
                }
                __syncthreads();
            }
        }
        __syncthreads();
        for (int g = 0; g < ng; ++g) {
            exg_cut(tks[g], s_cnt[g], s_bound[g], a.k);
            const int64_t o = ((int64_t)(fg + g - a.f0) * a.n_lists + blockIdx.x) * a.k;
            for (int j = tid; j < a.k; j += kExThreads) {
                const bool ok = j < s_cnt[g];
                a.part_key[o + j] = ok ? tks[g].key[j] : ~0ull;
                a.part_id[o + j] = ok ? tks[g].id[j] : 0x7fffffff;
            }
        }
        // the LAST workgroup to finish a query folds its per-workgroup lists (as in the single-query kernel)
        __threadfence();
        __syncthreads();
        for (int g = 0; g < ng; ++g) {
            const int fs = fg + g - a.f0;
            if (tid == 0) s_last = atomicAdd(a.done + fs, 1u) == gridDim.x - 1 ? 1 : 0;
            __syncthreads();
            if (s_last) {
                __threadfence();
                exact_merge_lists(a, tk_big, fs, a.flag_list[fg + g]);
                if (tid == 0) a.done[fs] = 0u;
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Matrix-core form (round 6): sixteen flagged queries per pass over the rows on v_mfma_f64_16x16x4_f64.
// The grouped kernel above is bound by its LDS query reads (every v_fma_f64 wants a fresh query operand: 512 B of LDS
// per wave instruction against 128 B per clock and CU - 5.8 ms per eight-query pass over 4 M x 640 rows where the
// arithmetic is 0.8 ms).  An MFMA reuses its operands sixteen-fold: a wave owns 16 rows x 16 queries, lane (r, p) holds
// the A operand x[row r][k] and the B operand q[k][query r] for the k of its 16-byte piece p of every 64-byte row
// segment (the dot product does not care in which order k runs, so the coalesced load IS the operand: no transpose), one
// ds_read_b64 and two conversions per MFMA.  The chip sustains 34.7 TFLOP/s on this instruction
// (profiles/r04z_mfma_f64_probe.txt): 16 queries x 4 M x 640 = 84 GFLOP in ~2.4 ms, 0.15 ms per flagged query
// against 0.73 (grouped) and 1.4 (one by one).
//   inner product / cosine: the float64 MFMA sum IS the score (another summation order than the one-query kernel's -
//     the last bit of a float64 may differ, D and I do not unless two float64 scores tie within that bit);
//   squared L2: ||x||^2 - 2 q.x + ||q||^2 in float64 only SELECTS (error <= 2 d 2^-53 (||x||^2 + ||q||^2), a margin of
//     1e-12 (||x||^2 + ||q||^2) is kept); every pair that may enter a list is scored again as the direct sum of
//     (q - x)^2 by 16 lanes in the one-query kernel's own order - bit-identical scores, no cancellation in what is
//     returned (identical rows give exactly 0).
// Queries sit in LDS transposed and padded ([k][16 queries] doubles, 64 B more per 8 k: a wave's ds_read_b64 touches
// every bank twice, the minimum); 16 threshold lists of 256 slots (k <= 64) alias the merge's big list.  d = 32 NS,
// NS in {4, 8, 12, 16, 20, 24} in groups of 16 queries, {32, 48} (rows of 1024 / 1536 elements) in groups of 8 - what
// fits LDS; H row segments are in flight per lane (a register is refilled right behind its use, from the next tile
// when the row is exhausted).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kExMG = 16;
constexpr int kExMCap = 256;
constexpr int kExMRows = 128;     // rows per workgroup step: 8 waves x 16
#ifndef PRAG_EXM_CHAINS
#define PRAG_EXM_CHAINS 4
#endif
constexpr int kExMChains = PRAG_EXM_CHAINS;   // independent accumulator chains per wave
typedef double f64x4 __attribute__((ext_vector_type(4)));

// (G queries per group: 16, or 8 for rows of more than 768 elements - their 16-query block would not fit LDS)
template <int G>
__device__ __forceinline__ int exm_q_off(int e, int col) { return (e * G + col) * 8 + (e >> 3) * 64; }
template <int G>
constexpr int exm_q_off_step = 32 * G * 8 + 4 * 64;     // exm_q_off(e + 32, col) - exm_q_off(e, col)
// the group's lists alias the merge's big list
template <int G>
constexpr size_t exm_list_bytes = ((size_t)G * kExMCap * 12 > sizeof(ExTopK) ? (size_t)G * kExMCap * 12 : sizeof(ExTopK) + 15) / 16 * 16;
constexpr int exm_group(int d) { return d <= 768 ? 16 : 8; }
size_t exact_mfma_lds_bytes(int d) {   // queries | ||q||^2 | spare | lists | bounds, counters (256 B) | pool scratch [512]
    const int G = exm_group(d);
    return (size_t)d * (G * 8 + 8) + 16 * 8 + 8 * 16 * 8 + (G == 16 ? exm_list_bytes<16> : exm_list_bytes<8>) + 256 + 512 * 8;
}

// Sort every list of the group whose bit is set in `need` by (key, id), keep its k best, tighten its bound - all of
// them in the SAME 36 barrier phases (the lists of a group fill at the same pace: one list at a time cost 16 x 36).
// All threads; `need` is uniform.
template <int G>
__device__ __forceinline__ void exm_cut_lists(unsigned long long* l_key, int* l_id, int* s_cnt, unsigned long long* s_bound,
                                              uint32_t need, int k) {
    __syncthreads();
    for (int i = threadIdx.x; i < G * kExMCap; i += kExThreads) {
        const int g = i / kExMCap, e = i - g * kExMCap;
        if (((need >> g) & 1u) && e >= min(s_cnt[g], kExMCap)) {
            l_key[i] = ~0ull;
            l_id[i] = 0x7fffffff;
        }
    }
    for (int size = 2; size <= kExMCap; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int pp = threadIdx.x; pp < G * (kExMCap >> 1); pp += kExThreads) {
                const int g = pp / (kExMCap >> 1), p = pp - g * (kExMCap >> 1);
                if (!((need >> g) & 1u)) continue;
                unsigned long long* key = l_key + g * kExMCap;
                int* id = l_id + g * kExMCap;
                const int i = ((p / stride) * 2 * stride) + (p % stride), j = i + stride;
                const unsigned long long ka = key[i], kb = key[j];
                const int ia = id[i], ib = id[j];
                const bool gt = ka > kb || (ka == kb && ia > ib);
                if (gt == ((i & size) == 0)) {
                    key[i] = kb;
                    key[j] = ka;
                    id[i] = ib;
                    id[j] = ia;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < G && ((need >> threadIdx.x) & 1u)) {
        const int g = threadIdx.x, n = min(s_cnt[g], kExMCap);
        s_cnt[g] = n < k ? n : k;
        if (n >= k && l_key[g * kExMCap + k - 1] < s_bound[g]) s_bound[g] = l_key[g * kExMCap + k - 1];
    }
    __syncthreads();
}

// k-th smallest (kth = 1 ...) of the up to 256 keys a wave holds four per lane (v[c] = key lane + 64 c; absent ones ~0):
// radix select from the top bit down, ballots and popcounts only - no LDS, no barrier.  Uniform result.
__device__ __forceinline__ unsigned long long exm_wave_kth(const unsigned long long (&v)[4], int kth) {
    unsigned long long prefix = 0;
    int rem = kth;
    for (int bit = 63; bit >= 0; --bit) {
        const unsigned long long hi = bit == 63 ? 0ull : ~((2ull << bit) - 1ull);      // the bits above `bit`
        int cnt0 = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool m = ((v[c] ^ prefix) & hi) == 0ull && ((v[c] >> bit) & 1ull) == 0ull;
            cnt0 += __popcll(__ballot(m));
        }
        if (rem > cnt0) {
            prefix |= 1ull << bit;
            rem -= cnt0;
        }
    }
    return prefix;
}

// The fold of the per-workgroup lists of a whole GROUP of queries, by the last workgroup to arrive: one wave per query,
// eight queries at a time.  The lists are sorted, so the k-th smallest of their heads bounds the k-th best entry; a wave
// keeps what is under it (about 2 k of its query's n_lists x k entries) in one of the group's list buffers, ONE batched sort
// (exm_cut_lists) finishes all eight, the outputs follow.  (Query after query through the threshold buffer -
// exact_merge_lists - this cost ~40 us per query, all in that one workgroup.)  A query with more than a buffer holds under
// its bound - rows that tie, mostly - takes the general fold afterwards.
__device__ __forceinline__ void exm_merge_group(const ExactArgs& a, ExTopK& tk_big, unsigned long long* l_key, int* l_id, int* s_cnt,
                                                unsigned long long* s_bound, int fg, int ng) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t total = (int64_t)a.n_lists * a.k;
    for (int g0 = 0; g0 < ng; g0 += 8) {
        __syncthreads();
        const int g = g0 + w;
        int cnt = 0;
        if (g < ng) {
            const int64_t o = (int64_t)(fg + g - a.f0) * total;
            unsigned long long bound = ~0ull;
            if (a.n_lists >= a.k) {
                unsigned long long hv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int l = lane + 64 * c;
                    hv[c] = l < a.n_lists ? a.part_key[o + (int64_t)l * a.k] : ~0ull;
                }
                bound = exm_wave_kth(hv, a.k);
            }
            for (int64_t i0 = 0; i0 < total; i0 += 64) {
                const int64_t i = i0 + lane;
                const unsigned long long key = i < total ? a.part_key[o + i] : ~0ull;
                const int id = i < total ? a.part_id[o + i] : 0x7fffffff;
                const bool keep = id != 0x7fffffff && key <= bound;
                const unsigned long long m = __ballot(keep);
                const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                if (keep && pos < kExMCap) {
                    l_key[w * kExMCap + pos] = key;
                    l_id[w * kExMCap + pos] = id;
                }
                cnt += __popcll(m);
            }
        }
        if (lane == 0) {
            s_cnt[w] = cnt;
            s_bound[w] = ~0ull;
        }
        __syncthreads();
        uint32_t ok = 0, over = 0;                    // (uniform: read behind the barrier)
        for (int q = 0; q < 8; ++q) {
            if (g0 + q < ng) {
                if (s_cnt[q] > kExMCap) over |= 1u << q;
                else ok |= 1u << q;
            }
        }
        if (ok) exm_cut_lists<8>(l_key, l_id, s_cnt, s_bound, ok, a.k);
        for (int idx = tid; idx < 8 * a.k; idx += kExThreads) {
            const int q = idx / a.k, j = idx - q * a.k;
            if (!((ok >> q) & 1u)) continue;
            const int b = a.flag_list[fg + g0 + q];
            const bool have = j < s_cnt[q];
            const unsigned long long key = l_key[q * kExMCap + j];
            const double sc = have ? unsortable_f64(a.metric_l2 ? key : ~key) : 0.0;
            a.D[(int64_t)b * a.k + j] = have ? (float)sc : (a.metric_l2 ? FLT_MAX : -FLT_MAX);
            a.I[(int64_t)b * a.k + j] = have ? tag_id((int64_t)l_id[q * kExMCap + j] + a.id_offset, sc, a.tag_ids) : -1;
        }
        __syncthreads();
        for (int q = 0; q < 8; ++q)                    // (the general fold's buffer aliases the lists: outputs are out)
            if ((over >> q) & 1u) exact_merge_lists(a, tk_big, fg + g0 + q - a.f0, a.flag_list[fg + g0 + q]);
    }
    __syncthreads();
}

template <bool F32, int NS, int H>
__global__ __launch_bounds__(kExThreads) void exact_mfma_kernel(ExactArgs a) {
    static_assert(NS % H == 0 && H <= NS, "the register ring divides the row");
    extern __shared__ __attribute__((aligned(16))) char ex_smem[];
    constexpr int d = NS * 32;
    constexpr int G = exm_group(d);                                         // queries per group
    char* s_q = ex_smem;                                                    // [d][G] doubles, padded (exm_q_off)
    double* s_qn2 = reinterpret_cast<double*>(ex_smem + (size_t)d * (G * 8 + 8));   // [16] ||q||^2 (L2)
    double* s_xn = s_qn2 + 16;                                              // (spare)
    char* s_lists = reinterpret_cast<char*>(s_xn + 8 * 16);
    unsigned long long* l_key = reinterpret_cast<unsigned long long*>(s_lists);            // [G][kExMCap]
    int* l_id = reinterpret_cast<int*>(s_lists + (size_t)G * kExMCap * 8);                 // [G][kExMCap]
    ExTopK& tk_big = *reinterpret_cast<ExTopK*>(s_lists);
    static_assert(sizeof(ExTopK) <= exm_list_bytes<G>, "the merge list aliases the group's lists");
    unsigned long long* s_bound = reinterpret_cast<unsigned long long*>(s_lists + exm_list_bytes<G>);   // (16 slots either way)
    int* s_cnt = reinterpret_cast<int*>(s_bound + kExMG);
    int& s_last = s_cnt[kExMG];
    // this workgroup's best key per query so far (what it publishes for the chip-wide bound): behind the 256 bytes of
    // bounds and counters
    unsigned long long* s_best = reinterpret_cast<unsigned long long*>(s_lists + exm_list_bytes<G> + 256);
    if (gate_closed(a.gate)) return;
    const uint32_t nf = *a.n_flag;
    if ((uint32_t)a.f0 >= nf) return;
    const int f1 = (int)std::min<uint32_t>(nf, (uint32_t)(a.f0 + a.f_cap));
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, p = lane >> 4;
    const int64_t n_tiles = (a.N + kExMRows - 1) / kExMRows;
    const bool l2 = a.metric_l2 != 0;

    auto push = [&](int col, unsigned long long key, int id) __attribute__((always_inline)) {
        const int sl = atomicAdd(&s_cnt[col], 1);
        if (sl < kExMCap) {
            l_key[col * kExMCap + sl] = key;
            l_id[col * kExMCap + sl] = id;
        }
        atomicMin(&s_best[col], key);
    };

    for (int fg = a.f0; fg < f1; fg += G) {
        const int ng = min(G, f1 - fg);
        __syncthreads();
        for (int i = tid; i < G * d; i += kExThreads) {
            const int g = i / d, c = i - g * d;
            // (slots past the group's last query repeat the first one: same work, results never pushed)
            const int b = a.flag_list[fg + (g < ng ? g : 0)];
            *reinterpret_cast<double*>(s_q + exm_q_off<G>(c, g)) = (double)a.q32[(int64_t)b * d + c];
        }
        if (tid < kExMG) {
            s_cnt[tid] = 0;
            s_bound[tid] = ~0ull;
            s_best[tid] = ~0ull;
        }
        __syncthreads();
        {
            for (int gq = w; gq < G; gq += 8) {
                double sq = 0.0;
                for (int c = lane; c < d; c += 64) {
                    const double v = *reinterpret_cast<const double*>(s_q + exm_q_off<G>(c, gq));
                    sq = fma(v, v, sq);
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
                if (lane == 0) s_qn2[gq] = sq;
            }
            __syncthreads();
        }
        // ---- the register ring: H of the row's NS 64-byte segments per lane ---------------------------------------
        u32x4 v0[H], v1[F32 ? H : 1];
        // this lane's piece of its row in tile `tile`: ONE 64-bit base per tile (opaque: left to itself the compiler
        // keeps a precomputed address pair per unrolled step alive across the whole kernel and spills them), the steps
        // are immediate offsets from it
        constexpr int ES = F32 ? 4 : 2;
        auto row_base = [&](int64_t tile) __attribute__((always_inline)) {
            int64_t row = tile * kExMRows + w * 16 + r;
            row = row < a.N ? row : a.N - 1;                  // (past the end: any valid row, never pushed)
            const char* b = reinterpret_cast<const char*>(a.rows) + (row * d + p * 8) * ES;
            asm volatile("" : "+v"(b));
            return b;
        };
        auto load_seg = [&](int slot, const char* base, int it) __attribute__((always_inline)) {
            v0[slot] = exm_load(reinterpret_cast<const u32x4*>(base + it * 32 * ES));
            if constexpr (F32) v1[slot] = exm_load(reinterpret_cast<const u32x4*>(base + it * 32 * ES + 16));
        };
        if ((int64_t)blockIdx.x < n_tiles) {
            const char* b0 = row_base(blockIdx.x);
#pragma unroll
            for (int it = 0; it < H; ++it) load_seg(it, b0, it);
        }
        // inner products: the margin of the selection only needs SOME bound on ||x||^2 - the shard's largest (one word) saves
        // eight float64 fmas per step and lane; squared L2 needs each row's own norm for the selection value itself
        const bool need_xsq = l2 || a.xn_max == nullptr;
        const double xn_shard = need_xsq ? 0.0 : (double)__uint_as_float(*a.xn_max) * (1.0 + 1e-6);
        // byte offset of q[k = 8 p][query r]; a step of 32 k is 32 x 128 + 4 x 64 bytes on
        const int q_lane = exm_q_off<G>(p * 8, r & (G - 1));
        constexpr int q_step = exm_q_off_step<G>, q_k = G * 8;             // bytes per 32 k, per k
        int steps_done = 0;
        // what follows a step's scores: capacity check / cuts, the published best key, the chip-wide bound
        auto step_tail = [&]() __attribute__((always_inline)) {
            __syncthreads();
            uint32_t need = 0;                    // (uniform: the counters are read behind the barrier by every thread)
            for (int g = 0; g < ng; ++g) need |= s_cnt[g] > kExMCap - kExMRows ? 1u << g : 0u;
            // after the FIRST step every list is cut at once: its k-th best of 128 rows lets one pair in four through in
            // the second step, where an empty bound re-scores all 128 x ng of them again (~45 us at 16 queries)
            if (steps_done == 0)
                for (int g = 0; g < ng; ++g) need |= s_cnt[g] > a.k ? 1u << g : 0u;
            if (need) exm_cut_lists<G>(l_key, l_id, s_cnt, s_bound, need, a.k);
            // The chip-wide bound.  Every workgroup publishes its best key per query after 1, 2, 4, 8, ... steps and reads
            // the others' then: the k-th smallest of the workgroups' best keys is the worst of SOME k distinct rows, hence
            // an upper bound on the k-th best of all rows - about the k-th best of n_lists x 128 x steps rows, where a
            // workgroup's own list only knows the k-th best of its own.  Pushes (and with them the exact re-scoring of the
            // pairs that pass the matrix pipe's selection) drop from ~k ln(rows per workgroup / k) per workgroup and
            // query to about that many over the whole chip.  One wave per query (two rounds), no block barrier.
            ++steps_done;
            if ((steps_done & (steps_done - 1)) == 0 && a.n_lists >= a.k) {
                for (int g = w; g < ng; g += 8) {
                    unsigned long long* pool = a.gpool + (int64_t)(fg + g - a.f0) * a.n_lists;
                    if (lane == 0 && s_best[g] != ~0ull)
                        __hip_atomic_store(pool + blockIdx.x, s_best[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned long long hv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int l = lane + 64 * c;
                        hv[c] = l < a.n_lists ? __hip_atomic_load(pool + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
                    }
                    const unsigned long long kth = exm_wave_kth(hv, a.k);
                    if (lane == 0 && kth < s_bound[g]) s_bound[g] = kth;
                }
            }
            __syncthreads();
        };
        {
            // (Round 6 had a plain float64 fma loop for ONE flagged query - 1.49 ms over 4 M x 640 fp16 rows; the four-block
            //  shape below does one query in 1.4, so it went.)
            // One to sixteen flagged queries on the float64 matrix pipe, in one of two shapes (uniform per group):
            //   9..16 queries: one 16 x 16 x 4 tile per k-quad - D[row = p + 4 reg][query r], four registers a lane;
            //   1..8 queries: four 4 x 4 x 4 blocks per instruction (v_mfma_f64_4x4x4_4b) - block b multiplies rows 4 b ..
            //   4 b + 3 by FOUR queries, so a group of <= 4 costs a 3.5th of the tile's time on the pipe and a group of <= 8
            //   (a second instruction for queries 4..7) 1.75 times less (tools/micro/mfma_f64_4x4_probe.hip: the operand
            //   layout - A is the tile's A operand unchanged, B holds q[k = p][query lane & 3], D[i][j] of block b sits in
            //   lane 16 i + 4 b + j - and the rate, 69.6 TFLOP/s against the tile's 78).
            auto mfma_pass = [&](auto small_tag) __attribute__((always_inline)) {
            constexpr bool S = decltype(small_tag)::value;
            const bool hi = ng > 4;                                          // (S: queries 4..7 in use)
            const int q_lane_s = exm_q_off<G>(p * 8, lane & 3);
            for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
                const int64_t t_next = t + gridDim.x < n_tiles ? t + gridDim.x : t;
                const char* cur_b = row_base(t);
                const char* nxt_b = row_base(t_next);
            // (independent accumulator chains per wave: with two waves per SIMD the MFMAs in flight the rate probe ran with)
            f64x4 acc[S ? 1 : kExMChains];
            double a0[4], a1[4];
#pragma unroll
            for (int c = 0; c < (S ? 1 : kExMChains); ++c) acc[c] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < 4; ++c) a0[c] = a1[c] = 0.0;
            double xsq = 0.0;
#pragma unroll
            for (int it = 0; it < NS; ++it) {
                const int slot = it % H;
                double xd[8];
                if constexpr (F32) {
                    const f32x4 x0 = __builtin_bit_cast(f32x4, v0[slot]);
                    const f32x4 x1 = __builtin_bit_cast(f32x4, v1[slot]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xd[j] = (double)x0[j]; xd[4 + j] = (double)x1[j]; }
                } else {
                    const half8 h = __builtin_bit_cast(half8, v0[slot]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xd[j] = (double)(float)h[j];
                }
                // refill the segment just consumed: a later segment of this tile, or the next tile's
                if (it + H < NS) load_seg(slot, cur_b, it + H);
                else load_seg(slot, nxt_b, it + H - NS);
                // (the step's LDS address is formed HERE: hoisted out of the loop, the NS addresses of the unrolled steps -
                //  beyond the 64 KB a ds_read offset reaches - cost a register each and spilled)
                int qo = (S ? q_lane_s : q_lane) + it * q_step;
                asm volatile("" : "+v"(qo));
                const char* qb = s_q + qo;                                   // (+ q_k bytes per k inside the piece)
                if constexpr (S) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const double qv = *reinterpret_cast<const double*>(qb + j * q_k);
                        a0[j % 4] = __builtin_amdgcn_mfma_f64_4x4x4f64(xd[j], qv, a0[j % 4], 0, 0, 0);
                    }
                    if (hi) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const double qv = *reinterpret_cast<const double*>(qb + j * q_k + 32);     // query + 4
                            a1[j % 4] = __builtin_amdgcn_mfma_f64_4x4x4f64(xd[j], qv, a1[j % 4], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const double qv = *reinterpret_cast<const double*>(qb + j * q_k);
                        acc[j % kExMChains] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd[j], qv, acc[j % kExMChains], 0, 0, 0);
                    }
                }
                if (need_xsq) {      // (uniform; inner products take the shard's largest norm for their margin instead)
#pragma unroll
                    for (int j = 0; j < 8; ++j) xsq = fma(xd[j], xd[j], xsq);
                }
                // (one step's reads and conversions at a time: left free, the scheduler hoists the LDS reads and conversions
                //  of all NS steps to the top - 100-500 bytes of scratch per lane)
                __builtin_amdgcn_sched_barrier(0);
            }
            const int64_t base_row = t * kExMRows + w * 16;
            if (need_xsq) {
                xsq += __shfl_xor(xsq, 16, 64);
                xsq += __shfl_xor(xsq, 32, 64);                              // ||x_r||^2 in the row's four lanes
            }
            // One value of the matrix pipe per call: row `row16` of the wave's sixteen, query `col` of the group.  The
            // pipe's value only SELECTS (its float64 sums are not even the same for identical rows in different places of a
            // tile - the last register of an accumulator rounds differently -, so ties between copies of a row would break
            // by position): whatever may enter a list, by a margin above the worst-case error d 2^-53 (||x||^2 + ||q||^2),
            // is scored again here in the one-query kernel's order.
            auto consume = [&](double dotv, int row16, int col) __attribute__((always_inline)) {
                const double xn = need_xsq ? __shfl(xsq, row16, 64) : xn_shard;
                const double qn = s_qn2[col];
                const double margin = 1e-12 * (xn + qn);
                const unsigned long long sel_key = l2 ? sortable_u64(((xn - 2.0 * dotv) + qn) - margin)
                                                      : ~sortable_u64(dotv + margin);
                const bool pass = base_row + row16 < a.N && col < ng && sel_key <= s_bound[col];
                const int mine = row16 | (col << 8);
                unsigned long long m = __ballot(pass);
                while (m) {            // four (row, query) pairs at a time: 16 lanes each, the one-query kernel's sum
                    int src = -1;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int bit = m ? __ffsll((long long)m) - 1 : -1;
                        if (g == p) src = bit;
                        if (m) m &= m - 1;
                    }
                    const int rc = __shfl(mine, src >= 0 ? src : 0, 64);
                    const int64_t rowx = base_row + (rc & 255);
                    const int colx = rc >> 8;
                    double sx = 0.0;
                    for (int e = r * 8; e < d; e += 128) {
                        double xv[8];
                        if constexpr (F32) {
                            const float* src32 = reinterpret_cast<const float*>(a.rows) + rowx * d + e;
                            const f32x4 x0 = *reinterpret_cast<const f32x4*>(src32);
                            const f32x4 x1 = *reinterpret_cast<const f32x4*>(src32 + 4);
#pragma unroll
                            for (int j = 0; j < 4; ++j) { xv[j] = (double)x0[j]; xv[4 + j] = (double)x1[j]; }
                        } else {
                            const half8 h = *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(a.rows) + rowx * d + e);
#pragma unroll
                            for (int j = 0; j < 8; ++j) xv[j] = (double)(float)h[j];
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const double qv = *reinterpret_cast<const double*>(s_q + exm_q_off<G>(e + j, colx));
                            if (l2) {
                                const double df = qv - xv[j];
                                sx = fma(df, df, sx);
                            } else {
                                sx = fma(qv, xv[j], sx);
                            }
                        }
                    }
                    sx = dpp_add16_f64(sx);
                    if (src >= 0 && r == 0) {
                        const unsigned long long key = l2 ? sortable_u64(sx) : ~sortable_u64(sx);
                        if (key <= s_bound[colx]) push(colx, key, (int)rowx);
                    }
                }
            };
            if constexpr (S) {
                const int row16 = 4 * ((lane >> 2) & 3) + p;                 // D[i = p][j = lane & 3] of block (lane >> 2) & 3
                consume((a0[0] + a0[1]) + (a0[2] + a0[3]), row16, lane & 3);
                if (hi) consume((a1[0] + a1[1]) + (a1[2] + a1[3]), row16, 4 + (lane & 3));
            } else {
                f64x4 dot = acc[0];                                          // D[row = p + 4 reg][query r]
#pragma unroll
                for (int c = 1; c < kExMChains; ++c) dot += acc[c];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) consume(dot[reg], p + 4 * reg, r);
            }
                step_tail();
            }
            };
            if constexpr (G == 8) {
                mfma_pass(std::true_type{});
            } else {
                if (ng <= 8) mfma_pass(std::true_type{});
                else mfma_pass(std::false_type{});
            }
        }
        __syncthreads();
        exm_cut_lists<G>(l_key, l_id, s_cnt, s_bound, (1u << ng) - 1u, a.k);
        for (int g = 0; g < ng; ++g) {
            const int64_t o = ((int64_t)(fg + g - a.f0) * a.n_lists + blockIdx.x) * a.k;
            for (int j = tid; j < a.k; j += kExThreads) {
                const bool ok = j < s_cnt[g];
                a.part_key[o + j] = ok ? l_key[g * kExMCap + j] : ~0ull;
                a.part_id[o + j] = ok ? l_id[g * kExMCap + j] : 0x7fffffff;
            }
        }
        // the LAST workgroup to have written its lists of this group folds them for every query of the group (one
        // arrival counter per group - the slot of its first query; the other slots stay zero)
        __threadfence();
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(a.done + (fg - a.f0), 1u) == gridDim.x - 1 ? 1 : 0;
        __syncthreads();
        if (s_last) {
            __threadfence();
            exm_merge_group(a, tk_big, l_key, l_id, s_cnt, s_bound, fg, ng);
            if (tid == 0) a.done[fg - a.f0] = 0u;
            for (int j = tid; j < ng * a.n_lists; j += kExThreads)      // every workgroup is through with the group's pools
                __hip_atomic_store(a.gpool + (int64_t)(fg - a.f0) * a.n_lists + j, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}

bool exact_mfma_supported(int d, int k) {
    return k <= kExMCap / 4 && ((d % 128 == 0 && d >= 128 && d <= 768) || d == 1024 || d == 1536);
}

template <bool F32>
static int launch_exact_mfma(const ExactArgs& a, int grid, hipStream_t st) {
    const size_t lds = exact_mfma_lds_bytes(a.d);
#define PRAG_EXM(NS_, H16_, H32_)                                                                       \
    if (a.d == 32 * NS_) {                                                                              \
        auto kern = exact_mfma_kernel<F32, NS_, F32 ? H32_ : H16_>;                                     \
        static LdsOptIn opt_in;                                                                         \
        const int rc_ = opt_in.ensure(reinterpret_cast<const void*>(kern), 160 * 1024);                 \
        if (rc_ != PRAG_OK) return rc_;                                                                 \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kExThreads), lds, st, a);                             \
        return PRAG_OK;                                                                                 \
    }
    // (row segments in flight per lane: 16 B each on fp16 rows, 32 B on float32 rows - sized so that no form spills)
    PRAG_EXM(4, 4, 4) PRAG_EXM(8, 8, 4) PRAG_EXM(12, 6, 4) PRAG_EXM(16, 8, 4) PRAG_EXM(20, 10, 5) PRAG_EXM(24, 8, 4)
    PRAG_EXM(32, 16, 8) PRAG_EXM(48, 12, 6)
#undef PRAG_EXM
    set_error("internal: exact_mfma_kernel has no d = %d form", a.d);
    return PRAG_EUNSUPPORTED;
}

// (Round 3's batched attempt, for the record: rows loaded once per step, float32 queries in LDS, one threshold list per
// query - four flagged queries cost 10.2 ms batched against 10.4 ms one by one at 8 M x 640 fp16 rows, and ONE flagged
// query 6.1 ms instead of 2.6: 74 KB of LDS left one workgroup per CU.  The grouped kernel above keeps float64 queries
// in LDS and is only taken when several flagged queries are expected.)
size_t exact_part_entries(int f_cap, int grid, int k) { return (size_t)f_cap * grid * k; }

int exact_run(const ExactRun& r, hipStream_t st) {
    PRAG_REQUIRE(r.k >= 1 && r.k <= kExCap / 2 && r.d <= 1536 && r.f_cap >= 1 && r.grid >= 1, PRAG_EUNSUPPORTED,
                 "internal: exact scan called outside its envelope (k=%d d=%d)", r.k, r.d);
    ExactArgs a;
    a.rows = r.rows;
    a.N = r.N;
    a.d = r.d;
    a.metric_l2 = r.metric_l2;
    a.q32 = r.q32;
    a.n_flag = r.n_flag;
    a.flag_list = r.flag_list;
    a.f_cap = r.f_cap;
    a.k = r.k;
    a.part_key = r.part_key;
    a.part_id = r.part_id;
    a.n_lists = r.grid;
    a.id_offset = r.id_offset;
    a.D = r.D;
    a.I = r.I;
    a.done = r.done;
    a.gpool = r.gpool;
    a.xn_max = r.xn_max;
    a.tag_ids = r.tag_ids;
    a.gate = r.gate;
    // several flagged queries expected, and the group's lists hold them: eight queries per pass over the rows
    // (float32 rows of more than 768 elements: the prefetched row is 96 more registers - that instantiation spilled)
    const bool grouped = r.grouped && r.k <= kExGroupCap / 4 && r.d % 16 == 0 && !(r.store_f32 && r.d > 768);
    const size_t g_lds = (size_t)kExGroup * r.d * sizeof(double) + (size_t)kExGroup * sizeof(ExSmallTopK) + kExGroup * 12 + 16;
    // (rows of <= 768 elements keep six 128-element steps in registers, twice: the prefetch; longer rows twelve)
    const bool short_rows = r.d <= 768;
    const void* gk = r.store_f32 ? reinterpret_cast<const void*>(exact_group_kernel<true, 6>)
                                 : (short_rows ? reinterpret_cast<const void*>(exact_group_kernel<false, 6>)
                                               : reinterpret_cast<const void*>(exact_group_kernel<false, 12>));
    if (grouped) {
        static LdsOptIn opt_in[4];
        const int rc_ = opt_in[(r.store_f32 ? 2 : 0) + (short_rows ? 1 : 0)].ensure(gk, 160 * 1024);
        if (rc_ != PRAG_OK) return rc_;
    }
    // ... sixteen per pass on the float64 matrix pipe where the shape allows (exact_mfma_kernel)
    const bool mfma = r.grouped && r.mfma && r.gpool != nullptr && exact_mfma_supported(r.d, r.k);
    for (int f0 = 0; f0 < r.B; f0 += r.f_cap) {
        a.f0 = f0;
        if (mfma) {
            // (one workgroup per CU - 155 KB of LDS -: half the list scan's grid, so no CU runs two in turn)
            ExactArgs am = a;
            am.n_lists = std::max(1, (r.grid + 1) / 2);
            const int rc_m = r.store_f32 ? launch_exact_mfma<true>(am, am.n_lists, st) : launch_exact_mfma<false>(am, am.n_lists, st);
            if (rc_m != PRAG_OK) return rc_m;
        } else if (grouped) {
            if (r.store_f32) {
                hipLaunchKernelGGL((exact_group_kernel<true, 6>), dim3(r.grid), dim3(kExThreads), g_lds, st, a);
            } else {
                if (short_rows) hipLaunchKernelGGL((exact_group_kernel<false, 6>), dim3(r.grid), dim3(kExThreads), g_lds, st, a);
                else hipLaunchKernelGGL((exact_group_kernel<false, 12>), dim3(r.grid), dim3(kExThreads), g_lds, st, a);
            }
        } else if (r.store_f32)
            hipLaunchKernelGGL(exact_scan_kernel<true>, dim3(r.grid), dim3(kExThreads), 0, st, a);
        else
            hipLaunchKernelGGL(exact_scan_kernel<false>, dim3(r.grid), dim3(kExThreads), 0, st, a);
        PRAG_LAUNCH_CHECK();
    }
    return PRAG_OK;
}

}  // namespace prag
