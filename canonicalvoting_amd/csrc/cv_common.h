// Shared host-side helpers for libcvhip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "cv_hip.h"

void cv_set_error(const char* fmt, ...);

#define CV_HIP_CHECK(expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            cv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                         __LINE__);                                                         \
            return CV_EHIP;                                                                 \
        }                                                                                   \
    } while (0)

#define CV_LAUNCH_CHECK() CV_HIP_CHECK(hipGetLastError())

#define CV_REQUIRE(cond, code, ...)     \
    do {                                \
        if (!(cond)) {                  \
            cv_set_error(__VA_ARGS__);  \
            return (code);              \
        }                               \
    } while (0)

__host__ __device__ static inline size_t cv_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Carves typed sub-buffers out of a caller-provided device workspace.
struct CvCarver {
    char* base;
    size_t off = 0;
    explicit CvCarver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t count) {
        off = cv_align_up(off, 256);
        T* r = reinterpret_cast<T*>(base + off);
        off += count * sizeof(T);
        return r;
    }
};
