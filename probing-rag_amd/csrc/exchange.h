// RCCL, bound at run time (exchange.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

namespace prag {

// ncclAllGather of `bytes` bytes per rank (ncclChar) on `st`; recv holds world x bytes.
int rccl_all_gather_bytes(void* comm, const void* send, void* recv, size_t bytes, hipStream_t st);

}  // namespace prag
