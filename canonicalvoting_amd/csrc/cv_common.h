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
    const unsigned* bitmap;  // optional (ts == 1 maps of the set the bitmap was built from): occupancy bits in front of the
    const int32_t* bbox;     // hash probes, see cv_sp_occupancy_bitmap; bbox = the 8 ints of cv_sp_sort_rows' bounds
    int up;                  // 1: transposed k2s2 map instead - out_coords are the FINE rows (tensor stride ts), keys / vals the
                             // table of the next coarser level, nbr[n_out][8] = parent row in the octant column, -1 elsewhere
};
constexpr int CV_MAX_MAP_JOBS = 12;
int cv_sp_kernel_maps_batch(const CvMapJob* jobs, int n_jobs, void* stream);
// Occupancy bitmap of a coordinate set over its bounding box (d_bbox: mm[0..2] = min, mm[3..5] = -max per axis, mm[6] =
// -max batch, as left by cv_sp_sort_rows at the start of its workspace): 87 % of a scene's kernel-map lookups are
// misses, each ~2 random probes into a 2 MB table; a bit test against ~0.5 MB with neighbouring lookups in the same words
// answers them.  d_bits: CV_BITMAP_WORDS words; when the box does not fit (or a batch index is negative) the kernels
// that take the bitmap ignore it (they re-derive the same test from the bounds).  Two launches, asynchronous.
constexpr long long CV_BITMAP_WORDS = 1ll << 20;
// pre_cleared: the words are zero and d_bbox[7] == 1 already (cv_sp_build_levels_zero did it): one launch
int cv_sp_occupancy_bitmap(const int32_t* d_coords, long long n, const int32_t* d_bbox, unsigned* d_bits, void* stream,
                           bool pre_cleared = false);

// cv_sp_build_levels with an extra range of words zeroed by its first launch (saves the caller's fill launches)
int cv_sp_build_levels_zero(int32_t* const* d_coords, unsigned long long* const* d_keys, int32_t* const* d_vals,
                            long long n, long long cap, int num_levels, int32_t* d_counts, int32_t* h_counts, void* d_ws,
                            size_t ws_bytes, int32_t* d_zero, long long n_zero, int32_t* d_set_one, void* stream,
                            uint32_t* d_bits = nullptr);       // d_bits: the occupancy bitmap (inside the zeroed range) is filled too

struct CvUpJob { const int32_t* nbr_down; long long n_coarse; int32_t* up; };
int cv_sp_up_maps_batch(const CvUpJob* jobs, int n_jobs, void* stream);      // the up arrays must be pre-filled with -1

struct CvPermJob { const int32_t* nbr; long long n; int K, groups; int32_t* perm; int with_map; };
constexpr int CV_MAX_PERM_JOBS = 8;
// d_ws: (sum of groups) * 1024 ints, zero-filled by the call (unless pre_zeroed)
int cv_hv_minmax_async_ex(const float* d_points, int64_t n, float* h_minmax6, void* d_ws, size_t ws_bytes, int32_t* d_zero_word,
                          int32_t* d_fill7f, void* stream);                                // hv_vote.hip
// cv_decode_f32 with an event (hipEvent_t, may be NULL) recorded behind its LAST launch, in front of the host's wait for the
// results: the scene call's "decode done" mark is then a device time (hv_decode.hip)
int cv_decode_f32_ev(float* d_grid_obj, const float* d_grid_rot, const float* d_grid_scale, const int dims[3],
                     const float h_corner3[3], float res, const float* d_points, const float* d_xyz, const float* d_prob,
                     const int32_t* d_class, int64_t n, const cv_decode_params* params, int mutate_grid, void* d_ws, size_t ws_bytes,
                     int* h_n_cand, int64_t* h_cand_idx, int32_t* h_verdict, int* h_n_boxes, float* h_boxes, float* h_scores,
                     int32_t* h_classes, int* h_truncated, void* stream, void* ev_done);
int cv_sp_sort_rows_ex(const int32_t* d_coords, long long n, int32_t* d_sorted, int32_t* d_perm, int32_t* d_inv, void* d_ws,
                       size_t ws_bytes, bool single_batch, void* stream, bool bounds_prefilled = false);   // sparse_coords.hip
int cv_sp_scene_plan_ex(const int32_t* d_input, long long n, int32_t* d_perm, int32_t* d_inv, int32_t* const* d_coords,
                        unsigned long long* const* d_keys, int32_t* const* d_vals, long long cap, int32_t* d_counts,
                        int32_t* h_counts, int stem_k, int mask_groups, long long masked_min_rows,
                        int32_t* d_arena, size_t arena_words, cv_scene_maps* offsets, void* d_sort_ws, size_t sort_ws_bytes,
                        void* d_levels_ws, size_t levels_ws_bytes, bool single_batch, void* stream,
                        bool bounds_prefilled = false);                                       // net_exec.cpp
int cv_sp_mask_perms_batch(const CvPermJob* jobs, int n_jobs, void* d_ws, size_t ws_bytes, void* stream,
                           bool pre_zeroed = false);

