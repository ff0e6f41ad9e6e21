// Shared host-side helpers for libprag.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "prag.h"

namespace prag {

// thread-local last-error message behind prag_last_error()
void set_error(const char* fmt, ...);

#define PRAG_HIP(call)                                                              \
    do {                                                                            \
        hipError_t _e = (call);                                                     \
        if (_e != hipSuccess) {                                                     \
            ::prag::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                              __FILE__, __LINE__);                                  \
            return PRAG_EHIP;                                                       \
        }                                                                           \
    } while (0)

#define PRAG_REQUIRE(cond, code, ...)      \
    do {                                   \
        if (!(cond)) {                     \
            ::prag::set_error(__VA_ARGS__); \
            return (code);                 \
        }                                  \
    } while (0)

// Launch check that does not synchronise.
#define PRAG_LAUNCH_CHECK() PRAG_HIP(hipGetLastError())

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

}  // namespace prag
