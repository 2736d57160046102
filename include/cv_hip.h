/*
 * cv_hip.h -- C ABI of libcvhip.so, the MI355X (gfx950) implementation of the
 * CanonicalVoting hot path.  Plain pointers and sizes only: every `d_` pointer is
 * device memory owned by the caller, every `h_` pointer is host memory, `stream`
 * is a hipStream_t passed as void* (NULL = the legacy default stream).
 *
 * All entry points return 0 on success or a negative CV_E* code; nothing throws
 * across this boundary.  cv_last_error() returns a thread-local message for the
 * last failing call.
 *
 * Each group cites the reference interface it replaces (paths relative to the
 * reference checkout).  The Python binding that presents these under the
 * reference's names (`hv_cuda.forward/backward`, `HoughVoting`, `MinkowskiEngine`
 * facade, `MinkUNet34C`) lives in canonicalvoting_amd/; INTEGRATION.md shows the
 * stub a reference maintainer would add.
 */
#ifndef CV_HIP_H
#define CV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CV_OK 0
#define CV_EINVAL (-22)   /* bad argument (null pointer, negative size, unsupported value) */
#define CV_ENOMEM (-12)   /* workspace too small */
#define CV_EHIP (-5)      /* a HIP runtime call or kernel launch failed */
#define CV_ERANGE (-34)   /* result does not fit the caller's buffers */

int cv_abi_version(void);
const char* cv_last_error(void);

/* ------------------------------------------------------------------------ *
 * Vote op: replaces pybind module `hv_cuda`
 *   forward  <- houghvoting/src/hv_cuda.cpp:30-45  -> hv_cuda_kernel.cu:121-165
 *   backward <- houghvoting/src/hv_cuda.cpp:47-71  -> hv_cuda_kernel.cu:265-302
 * ------------------------------------------------------------------------ */

/* Axis-aligned bounds of d_points[n][3] (torch::min/max(points,0),
 * hv_cuda_kernel.cu:129).  Writes h_min3/h_max3 and synchronises `stream` once.
 * d_ws: >= cv_hv_minmax_workspace_bytes() bytes of device scratch. */
size_t cv_hv_minmax_workspace_bytes(void);
int cv_hv_minmax_f32(const float* d_points, int64_t n, float* h_min3, float* h_max3,
                     void* d_ws, size_t ws_bytes, void* stream);

/* Grid shape from the bounds in the reference's fp32 arithmetic
 * (hv_cuda_kernel.cu:131-134: trunc((max-min)/res) + 1).  Host only. */
int cv_hv_grid_dims_f32(const float h_min3[3], const float h_max3[3], float res, int dims_out[3]);

/* algo: 0 = auto, 1 = direct global fp32 atomics (+memset +normalise pass),
 *       2 = LDS-tiled accumulation with fused normalise (no global atomics). */
size_t cv_hv_forward_workspace_bytes(int64_t n, int num_rots, const int dims[3], int algo);

/* Vote accumulation + per-cell normalisation (hv_cuda_kernel.cu:12-119).
 * d_grid_obj[X][Y][Z], d_grid_rot[X][Y][Z][2], d_grid_scale[X][Y][Z][3] are fully
 * overwritten (no pre-zeroing needed).  h_corner3 = grid origin (min of points, or
 * corners[0] for the 7-argument SUN RGB-D variant, sunrgbd/brnetcanon.py:99).
 * Asynchronous on `stream`. */
int cv_hv_forward_f32(const float* d_points, const float* d_xyz, const float* d_scale,
                      const float* d_obj, int64_t n, float res, int num_rots,
                      const float h_corner3[3], const int dims[3], float* d_grid_obj,
                      float* d_grid_rot, float* d_grid_scale, void* d_ws, size_t ws_bytes,
                      int algo, void* stream);

/* Gradient of sum(grad_obj * grid_obj) wrt xyz/scale/obj (hv_cuda_kernel.cu:168-261),
 * including the reference's missing 1/res factor.  Outputs are overwritten.
 * Asynchronous on `stream`. */
int cv_hv_backward_f32(const float* d_grad_obj, const float* d_points, const float* d_xyz,
                       const float* d_scale, const float* d_obj, int64_t n, float res,
                       int num_rots, const float h_corner3[3], const int dims[3], float* d_dxyz,
                       float* d_dscale, float* d_dobj, void* stream);

/* Number of (point, rotation) votes that pass the bounds test
 * (hv_cuda_kernel.cu:41-44); used for the algorithmic byte count of DESIGN.md.
 * Synchronises `stream`. */
int cv_hv_count_votes_f32(const float* d_points, const float* d_xyz, const float* d_scale, int64_t n,
                          float res, int num_rots, const float h_corner3[3], const int dims[3],
                          int64_t* h_count, void* d_ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------ *
 * Detection decode: replaces the inline loop eval_joint.py:195-263
 * (duplicated at train_joint.py:355-439, train_separate.py:371-431,
 * eval_separate.py:195-264) and the per-class NMS eval_joint.py:75-89,270-280
 * with utils/calc_map.py:6-21 as the IoU.
 * ------------------------------------------------------------------------ */
typedef struct cv_decode_params {
    float thresh_high;  /* eval_joint.py:18  (60)  */
    float thresh_low;   /* eval_joint.py:19  (10)  */
    float valid_ratio;  /* eval_joint.py:20  (0.2) */
    int elimination;    /* eval_joint.py:21  (2)   */
    float prob_thresh;  /* eval_joint.py:245 (0.3) */
    int elim_hi_plus1;  /* 1 = eval_joint.py:211 slice, 0 = eval_separate.py:209 slice */
    int max_iters;      /* capacity of the h_* output arrays (candidates examined) */
    double err_thresh;  /* eval_joint.py:252 (0.3) */
} cv_decode_params;

size_t cv_decode_workspace_bytes(const int dims[3], int64_t n, int max_iters);

/* Greedy peak picking + grid suppression + back-projection check + class vote.
 * Device inputs are read-only unless mutate_grid != 0, in which case d_grid_obj
 * receives the same zeroing the reference applies in place (:211,:243).
 * Host outputs: h_n_cand candidates examined (cell index + verdict 0 accept /
 * 1 too few confident points / 2 LCC error), h_n_boxes accepted boxes in
 * acceptance order (corners [8][3], score, class).  Synchronises `stream` once. */
int cv_decode_f32(float* d_grid_obj, const float* d_grid_rot, const float* d_grid_scale,
                  const int dims[3], const float h_corner3[3], float res, const float* d_points,
                  const float* d_xyz, const float* d_prob, const int32_t* d_class, int64_t n,
                  const cv_decode_params* params, int mutate_grid, void* d_ws, size_t ws_bytes,
                  int* h_n_cand, int64_t* h_cand_idx, int32_t* h_verdict, int* h_n_boxes,
                  float* h_boxes, float* h_scores, int32_t* h_classes, void* stream);

/* utils/calc_map.py:6-21 on two [8][3] corner sets (host). */
double cv_iou_obb(const float* h_box1, const float* h_box2);
/* eval_joint.py:75-89 (host): greedy NMS, returns the pick count, indices in h_pick. */
int cv_nms_obb(const float* h_boxes, const float* h_scores, int n, double thr, int32_t* h_pick);

#ifdef __cplusplus
}
#endif
#endif /* CV_HIP_H */
