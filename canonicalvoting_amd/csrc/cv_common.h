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

// ---- internal batch launches (one launch for the jobs of a whole scene instead of one per map; used by
//      cv_sp_scene_maps; C++ linkage, not part of the C ABI) ----------------------------------------------------
struct CvMapJob {            // kernel map: out set looked up in an input set's hash table
    const int32_t* out_coords; long long n_out;
    const unsigned long long* keys; const int32_t* vals; long long cap;
    int k, ts;
    int32_t* nbr;
    const int32_t* compose;  // optional: store compose[row] instead of row (a permutation folded into the map)
};
constexpr int CV_MAX_MAP_JOBS = 12;
int cv_sp_kernel_maps_batch(const CvMapJob* jobs, int n_jobs, void* stream);

struct CvUpJob { const int32_t* nbr_down; long long n_coarse; int32_t* up; };
int cv_sp_up_maps_batch(const CvUpJob* jobs, int n_jobs, void* stream);      // the up arrays must be pre-filled with -1

struct CvPermJob { const int32_t* nbr; long long n; int K, groups; int32_t* perm; int with_map; };
constexpr int CV_MAX_PERM_JOBS = 8;
// d_ws: (sum of groups) * 1024 ints, zero-filled by the call
int cv_sp_mask_perms_batch(const CvPermJob* jobs, int n_jobs, void* d_ws, size_t ws_bytes, void* stream);
