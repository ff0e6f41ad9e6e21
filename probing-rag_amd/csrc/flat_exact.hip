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
    for (int f0 = 0; f0 < r.B; f0 += r.f_cap) {
        a.f0 = f0;
        if (grouped) {
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
