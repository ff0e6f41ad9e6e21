// flat_sharded.hip - the exchange step of the row-sharded flat index (split out of flat_index.hip in
// round 6): the (score, residual, id) merge of per-shard top-k lists and the sharded search as ONE C call - local search
// with tagged ids, ONE ncclAllGather on the caller's stream, merge.  The reference has no counterpart: faiss-cpu, one
// process (exp_rag.py:248, 432; utils.py:378-380).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>

#include "exchange.h"
#include "flat_index_state.h"

namespace prag {

// ---------------------------------------------------------------------------
// cross-shard merge: one wave per query, lane p walks the sorted list of part p
// ---------------------------------------------------------------------------
// k rounds of a 64-lane lexicographic minimum over (score, residual tag, id); the winning lane pops its head and
// prefetches the next entry.  (Round 3 ran one THREAD per query with a head pointer per part in a runtime-indexed
// array: 272 B of scratch per lane and a serial k x n_parts loop - the exchange step of every sharded search.)
__global__ __launch_bounds__(64) void merge_shards_kernel(const float* __restrict__ Dp,
                                                         const int64_t* __restrict__ Ip, int64_t d_stride,
                                                         int64_t i_stride, int n_parts, int B,
                                                         int k, int metric_l2, int tagged, float* __restrict__ D,
                                                         int64_t* __restrict__ I) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t id_mask = tagged ? (int64_t)((1ull << kTagShift) - 1) : ~0ll;
    const bool part = lane < n_parts;
    const float* dp = Dp + (int64_t)lane * d_stride + (int64_t)b * k;
    const int64_t* ip = Ip + (int64_t)lane * i_stride + (int64_t)b * k;
    int head = 0;
    float dv = 0.f;
    int64_t raw = -1;
    if (part) {
        dv = dp[0];
        raw = ip[0];
    }
    for (int j = 0; j < k; ++j) {
        // this lane's candidate as (hi, lo): hi = score key (ascending = better) . residual key, lo = id
        const bool valid = part && head < k && raw >= 0;              // padding (-1) sorts last
        const float dz = dv + 0.0f;                                   // -0 and +0 compare equal, as floats do
        const uint32_t kd = metric_l2 ? sortable_u32(dz) : ~sortable_u32(dz);
        const uint32_t rk = tagged ? (uint32_t)((unsigned long long)raw >> kTagShift) : 0u;
        unsigned long long hi = valid ? (((unsigned long long)kd << 32) | (metric_l2 ? rk : 0xFFFFFFu - rk)) : ~0ull;
        unsigned long long lo = valid ? (unsigned long long)(raw & id_mask) : ~0ull;
        const unsigned long long my_hi = hi, my_lo = lo;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long ohi = __shfl_xor(hi, off, 64), olo = __shfl_xor(lo, off, 64);
            if (ohi < hi || (ohi == hi && olo < lo)) {
                hi = ohi;
                lo = olo;
            }
        }
        const unsigned long long won = __ballot(valid && my_hi == hi && my_lo == lo);
        if (won == 0ull) {   // every part exhausted: faiss padding
            if (lane == 0) {
                D[(int64_t)b * k + j] = metric_l2 ? FLT_MAX : -FLT_MAX;
                I[(int64_t)b * k + j] = -1;
            }
            continue;
        }
        const int wl = __ffsll((long long)won) - 1;                   // equal entries in two parts: the lower part first
        const float out_d = __shfl(dv, wl, 64);
        if (lane == 0) {
            D[(int64_t)b * k + j] = out_d;
            I[(int64_t)b * k + j] = (int64_t)lo;
        }
        if (lane == wl) {
            ++head;
            if (head < k) {
                dv = dp[head];
                raw = ip[head];
            }
        }
    }
}

}  // namespace prag

static int merge_topk_impl(const float* Dp, const int64_t* Ip, int64_t d_stride, int64_t i_stride, int n_parts,
                           int B, int k, int metric, float* D_dev, int64_t* I_dev, void* stream, int tagged = 0) {
    PRAG_REQUIRE(Dp && Ip && D_dev && I_dev, PRAG_EINVAL, "prag_merge_topk: NULL pointer");
    PRAG_REQUIRE(n_parts >= 1 && n_parts <= 64, PRAG_EINVAL, "n_parts=%d outside [1,64]", n_parts);
    PRAG_REQUIRE(B >= 0 && k >= 1, PRAG_EINVAL, "B=%d k=%d", B, k);
    if (B == 0) return PRAG_OK;
    hipLaunchKernelGGL(merge_shards_kernel, dim3(B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                       Dp, Ip, d_stride, i_stride, n_parts, B, k, metric == PRAG_METRIC_L2 ? 1 : 0, tagged, D_dev, I_dev);
    PRAG_LAUNCH_CHECK();
    return PRAG_OK;
}

extern "C" int prag_merge_topk(const float* D_parts_dev, const int64_t* I_parts_dev, int n_parts, int B, int k,
                               int metric, float* D_dev, int64_t* I_dev, void* stream) {
    return merge_topk_impl(D_parts_dev, I_parts_dev, (int64_t)B * k, (int64_t)B * k, n_parts, B, k, metric, D_dev,
                           I_dev, stream);
}

static int merge_packed_impl(const void* parts_dev, int64_t part_stride_bytes, int n_parts, int B, int k, int metric,
                             float* D_dev, int64_t* I_dev, void* stream, int tagged);

extern "C" int prag_merge_topk_packed(const void* parts_dev, int64_t part_stride_bytes, int n_parts, int B, int k,
                                      int metric, float* D_dev, int64_t* I_dev, void* stream) {
    return merge_packed_impl(parts_dev, part_stride_bytes, n_parts, B, k, metric, D_dev, I_dev, stream, 0);
}

extern "C" int prag_merge_topk_packed_tagged(const void* parts_dev, int64_t part_stride_bytes, int n_parts, int B, int k,
                                             int metric, float* D_dev, int64_t* I_dev, void* stream) {
    return merge_packed_impl(parts_dev, part_stride_bytes, n_parts, B, k, metric, D_dev, I_dev, stream, 1);
}

static int merge_packed_impl(const void* parts_dev, int64_t part_stride_bytes, int n_parts, int B, int k, int metric,
                             float* D_dev, int64_t* I_dev, void* stream, int tagged) {
    PRAG_REQUIRE(parts_dev != nullptr, PRAG_EINVAL, "prag_merge_topk_packed: NULL pointer");
    const int64_t i_off = ((int64_t)B * k * 4 + 7) / 8 * 8;  // I block starts 8-byte aligned after the D block
    PRAG_REQUIRE(part_stride_bytes >= i_off + (int64_t)B * k * 8 && part_stride_bytes % 8 == 0, PRAG_EINVAL,
                 "part_stride_bytes=%lld too small or not a multiple of 8", (long long)part_stride_bytes);
    const char* base = reinterpret_cast<const char*>(parts_dev);
    return merge_topk_impl(reinterpret_cast<const float*>(base), reinterpret_cast<const int64_t*>(base + i_off),
                           part_stride_bytes / 4, part_stride_bytes / 8, n_parts, B, k, metric, D_dev, I_dev, stream, tagged);
}

extern "C" int prag_index_set_comm(prag_index_t* ix, void* nccl_comm, int rank, int world) {
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    PRAG_REQUIRE(world >= 1 && rank >= 0 && rank < world, PRAG_EINVAL, "rank %d of %d", rank, world);
    PRAG_REQUIRE(nccl_comm != nullptr || world == 1, PRAG_EINVAL, "world=%d needs a communicator", world);
    ix->comm = nccl_comm;
    ix->comm_rank = rank;
    ix->comm_world = world;
    return PRAG_OK;
}

// The sharded search as ONE call: local search with tagged ids straight into this rank's slot of the packed exchange
// format, ONE all-gather on the caller's stream, the (score, residual, id) merge - what ShardedFlatIndex.search did
// with torch.distributed in between (sharded.py, rounds 1-3).  Device pointers; nothing waits for the stream.
extern "C" int prag_index_search_sharded(prag_index_t* ix, const float* q_dev, int B, int k, int64_t id_offset,
                                         float* D_dev, int64_t* I_dev, void* stream) {
    PRAG_REQUIRE(ix != nullptr, PRAG_EINVAL, "index handle is NULL");
    PRAG_REQUIRE(B >= 0 && k >= 1, PRAG_EINVAL, "B=%d k=%d", B, k);
    if (B == 0) return PRAG_OK;
    PRAG_REQUIRE(q_dev && D_dev && I_dev, PRAG_EINVAL, "prag_index_search_sharded: NULL pointer");
    PRAG_REQUIRE(id_offset >= 0 && id_offset + ix->ntotal < (1ll << kTagShift), PRAG_EUNSUPPORTED,
                 "tagged ids hold %d-bit global row ids", kTagShift);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int world = ix->comm ? ix->comm_world : 1;
    const size_t i_off = ((size_t)B * k * 4 + 7) / 8 * 8;           // the I block starts 8-byte aligned behind the D block
    const size_t stride = (i_off + (size_t)B * k * 8 + 15) / 16 * 16;
    if (stride > ix->xch_send_cap) {
        ix->xch_send_cap = 0;
        const int rc_ws = ws_regrow({{vpp(&ix->xch_send), stride}});
        if (rc_ws != PRAG_OK) return rc_ws;
        ix->xch_send_cap = stride;
    }
    if (ix->comm && stride * world > ix->xch_recv_cap) {
        ix->xch_recv_cap = 0;
        const int rc_ws = ws_regrow({{vpp(&ix->xch_recv), stride * world}});
        if (rc_ws != PRAG_OK) return rc_ws;
        ix->xch_recv_cap = stride * world;
    }
    int rc = index_search_impl(ix, q_dev, B, k, id_offset, reinterpret_cast<float*>(ix->xch_send),
                               reinterpret_cast<int64_t*>(ix->xch_send + i_off), 1, stream, 1);
    if (rc != PRAG_OK) return rc;
    const char* parts = ix->xch_send;
    if (ix->comm) {      // also with one rank: the collective the multi-rank path issues, in its dtype and shape
        ix->prof_xch.begin(st);
        rc = rccl_all_gather_bytes(ix->comm, ix->xch_send, ix->xch_recv, stride, st);
        ix->prof_xch.end(st);
        if (rc != PRAG_OK) return rc;
        parts = ix->xch_recv;
    }
    return prag_merge_topk_packed_tagged(parts, (int64_t)stride, world, B, k, ix->metric, D_dev, I_dev, stream);
}

extern "C" int prag_index_profile_read_exchange(prag_index_t* ix, float* ms, int cap, int* n_out) {
    PRAG_REQUIRE(ix != nullptr && ms != nullptr && cap >= 0, PRAG_EINVAL, "prag_index_profile_read_exchange: bad argument");
    return ix->prof_xch.read(ms, cap, n_out);
}

