// Shared host-side helpers for libprag.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

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

// Optional per-kernel timing: pairs of HIP events recorded on the launch stream
// around the dominant kernel of a call (bench.py's roofline uses these).
struct EventRing {
    std::vector<hipEvent_t> a, b;
    int n = 0;
    bool on = false;
    int enable(int slots) {
        disable();
        a.reserve(slots);
        b.reserve(slots);
        for (int i = 0; i < slots; ++i) {   // only successfully created events are ever tracked
            hipEvent_t ea, eb;
            if (hipEventCreate(&ea) != hipSuccess) { disable(); return PRAG_EHIP; }
            if (hipEventCreate(&eb) != hipSuccess) { (void)hipEventDestroy(ea); disable(); return PRAG_EHIP; }
            a.push_back(ea);
            b.push_back(eb);
        }
        n = 0;
        on = slots > 0;
        return PRAG_OK;
    }
    void disable() {
        for (auto e : a) (void)hipEventDestroy(e);
        for (auto e : b) (void)hipEventDestroy(e);
        a.clear();
        b.clear();
        n = 0;
        on = false;
    }
    void begin(hipStream_t st) {
        if (on && n < (int)a.size()) (void)hipEventRecord(a[n], st);
    }
    void end(hipStream_t st) {
        if (on && n < (int)a.size()) {
            (void)hipEventRecord(b[n], st);
            ++n;
        }
    }
    int read(float* ms, int cap, int* n_out) {
        int m = n < cap ? n : cap;
        for (int i = 0; i < m; ++i) {
            if (hipEventSynchronize(b[i]) != hipSuccess) return PRAG_EHIP;
            if (hipEventElapsedTime(&ms[i], a[i], b[i]) != hipSuccess) return PRAG_EHIP;
        }
        if (n_out) *n_out = m;
        n = 0;
        return PRAG_OK;
    }
};

// Kernels that need more than 64 KiB of dynamic LDS: the attribute is per device, so remember per
// (kernel instantiation, device) - one process may hold handles on several GPUs.
struct LdsOptIn {
    uint64_t done = 0;  // bit i = device i has the attribute
    int ensure(const void* fn, int bytes) {
        int dev = 0;
        PRAG_HIP(hipGetDevice(&dev));
        if (dev < 64 && (done >> dev) & 1) return PRAG_OK;
        PRAG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        if (dev < 64) done |= 1ull << dev;
        return PRAG_OK;
    }
};

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
// raw 16-byte payloads: an ext-vector (HIP's uint4 struct copies defeat SROA -> scratch)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

}  // namespace prag
