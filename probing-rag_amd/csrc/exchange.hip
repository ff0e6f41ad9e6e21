// The exchange step of the row-sharded index in C: RCCL all-gather of the packed local top-k lists on the caller's
// stream (SURVEY.md section 8b sketch: prag_index_set_comm + a sharded search entry point), so that a C / ctypes host
// can shard the index without torch.distributed.  Reference call this scales: utils.py:378-380
// (batch_topk_sim -> index.search); the reference itself is one process with faiss-cpu (exp_rag.py:248, 432).
//
// RCCL is bound at run time (dlopen): libprag.so carries no link-time dependency on it, a single-GPU host never loads
// it, and a process in which PyTorch has already loaded its own librccl uses that copy (RTLD_NOLOAD first).
// Prototypes restated from /opt/rocm/include/rccl/rccl.h (ROCm 7.2): ncclUniqueId is 128 opaque bytes passed BY
// VALUE to ncclCommInitRank; ncclChar = 0; ncclSuccess = 0.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "prag_common.h"
#include "exchange.h"

namespace prag {

namespace {
struct UniqueId {
    char internal[128];
};
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(void**, int, UniqueId, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct Rccl {
    void* lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
    bool tried = false;
};
Rccl g_rccl;              // guarded by g_rccl_mu until `lib` is set; read-only afterwards
std::mutex g_rccl_mu;

int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);     // two host threads creating sharded indexes at once
    if (g_rccl.lib) return PRAG_OK;
    if (g_rccl.tried) {
        set_error("RCCL is not available in this process (librccl.so could not be loaded)");
        return PRAG_EUNSUPPORTED;
    }
    g_rccl.tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names)      // a copy this process already holds (PyTorch's): use that one
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
    if (!h)
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!h) {
        set_error("RCCL is not available: dlopen(librccl.so) failed: %s", dlerror());
        return PRAG_EUNSUPPORTED;
    }
    g_rccl.get_unique_id = reinterpret_cast<fn_get_unique_id>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.comm_init_rank = reinterpret_cast<fn_comm_init_rank>(dlsym(h, "ncclCommInitRank"));
    g_rccl.comm_destroy = reinterpret_cast<fn_comm_destroy>(dlsym(h, "ncclCommDestroy"));
    g_rccl.all_gather = reinterpret_cast<fn_all_gather>(dlsym(h, "ncclAllGather"));
    g_rccl.error_string = reinterpret_cast<fn_error_string>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_gather) {
        set_error("RCCL: a symbol is missing from the loaded librccl.so");
        return PRAG_EUNSUPPORTED;
    }
    g_rccl.lib = h;
    return PRAG_OK;
}

int rccl_check(int rc, const char* what) {
    if (rc == 0) return PRAG_OK;
    set_error("RCCL: %s failed: %s (%d)", what, g_rccl.error_string ? g_rccl.error_string(rc) : "?", rc);
    return PRAG_EHIP;
}
}  // namespace

int rccl_all_gather_bytes(void* comm, const void* send, void* recv, size_t bytes, hipStream_t st) {
    const int rc = rccl_load();
    if (rc != PRAG_OK) return rc;
    return rccl_check(g_rccl.all_gather(send, recv, bytes, /*ncclChar*/ 0, comm, st), "ncclAllGather");
}

}  // namespace prag

using namespace prag;

extern "C" int prag_rccl_unique_id(void* id_out_128) {
    PRAG_REQUIRE(id_out_128 != nullptr, PRAG_EINVAL, "prag_rccl_unique_id: NULL pointer");
    const int rc = rccl_load();
    if (rc != PRAG_OK) return rc;
    UniqueId id;
    memset(&id, 0, sizeof(id));
    const int r2 = rccl_check(g_rccl.get_unique_id(&id), "ncclGetUniqueId");
    if (r2 != PRAG_OK) return r2;
    memcpy(id_out_128, id.internal, sizeof(id.internal));
    return PRAG_OK;
}

extern "C" int prag_rccl_comm_init_rank(void** comm_out, int world, int rank, const void* id_128) {
    PRAG_REQUIRE(comm_out != nullptr && id_128 != nullptr, PRAG_EINVAL, "prag_rccl_comm_init_rank: NULL pointer");
    PRAG_REQUIRE(world >= 1 && rank >= 0 && rank < world, PRAG_EINVAL, "rank %d of %d", rank, world);
    const int rc = rccl_load();
    if (rc != PRAG_OK) return rc;
    UniqueId id;
    memcpy(id.internal, id_128, sizeof(id.internal));
    void* comm = nullptr;
    const int r2 = rccl_check(g_rccl.comm_init_rank(&comm, world, id, rank), "ncclCommInitRank");
    if (r2 != PRAG_OK) return r2;
    *comm_out = comm;
    return PRAG_OK;
}

extern "C" int prag_rccl_all_gather(void* comm, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream) {
    PRAG_REQUIRE(comm && send_dev && recv_dev && bytes_per_rank > 0, PRAG_EINVAL, "prag_rccl_all_gather: NULL argument");
    return rccl_all_gather_bytes(comm, send_dev, recv_dev, bytes_per_rank, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int prag_rccl_comm_destroy(void* comm) {
    if (!comm) return PRAG_OK;
    const int rc = rccl_load();
    if (rc != PRAG_OK) return rc;
    return rccl_check(g_rccl.comm_destroy(comm), "ncclCommDestroy");
}
