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

/* Version of this header.  cv_abi_version() returns the value the LIBRARY was built with: a binding compiled against
 * this header compares the two (csrc/hv_cuda_ext.cpp does at import) and refuses a stale pair.
 * 2 (round 5): neighbour windows (cv_sp_build_windows, cv_conv_desc.win, cv_scene_maps.win, win_levels arguments of the
 *    scene-plan calls, `wins` of cv_net_run_f32); the round 1-3 tile-plan symbols are gone.
 * 3 (round 6): launch sizing per call - cv_scene_desc.conv_split_target / vote_part_records, cv_hv_set_part_records_thread;
 *    cv_sp_copy_unless_flag, cv_sp_pack_weights_h2_batch_f32; the neighbour windows of round 5 (conv_win: exact, measured slower
 *    than the mask-sorted kernels at every level, LABNOTES round 5) are gone - cv_sp_build_windows, cv_conv_desc.win,
 *    cv_scene_maps.win, the win_levels arguments and cv_net_win_levels. */
#define CV_ABI_VERSION 3
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
/* The same reduction without the host wait: h_minmax6 must be PINNED host memory (min xyz, max xyz) and is
 * valid once the work enqueued on `stream` up to this call has completed.  A pipeline can start it before the
 * network forward of a scene, so the vote does not stall on the grid shape (hv_cuda_kernel.cu:129-134 blocks
 * the host twelve times at that point). */
int cv_hv_minmax_async_f32(const float* d_points, int64_t n, float* h_minmax6, void* d_ws, size_t ws_bytes,
                           void* stream);

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

/* Launch sizing of the vote's streaming launch: records of a plane's two y-bins that one workgroup of a (tile, plane) takes; a
 * plane with more is split over up to 8 workgroups per tile whose partial tiles the last arriver adds (integer sums: the grids
 * are the same bits under every setting).  Default 4096 (or CV_HV_PART_RECORDS) - the fastest kernel for ONE scene in flight;
 * a host that keeps several scenes in flight sets a larger value (bench.py: 12288 from four scenes in flight: fewer, longer
 * workgroups and less merge traffic while the other scenes fill the chip; 574 -> 585 scenes/s with seven).  Values below 4096 are raised to it (the workspace
 * bound assumes it); records <= 0 restores the default.  Returns the previous value.  Process-wide, like
 * cv_sp_set_split_target. */
int cv_hv_set_part_records(int records);
/* The same for the CALLING THREAD's launches only (0 = follow the process-wide value); returns the previous thread value.
 * cv_detect_scene_f32 sets it from cv_scene_desc.vote_part_records for the duration of the call. */
int cv_hv_set_part_records_thread(int records);
/* Measurement hook (no reference counterpart): the CALLING THREAD's following cv_hv_forward_f32 calls record the two
 * hipEvent_t handles directly before and after the accumulation kernel of the tile algorithm (hv_fwd_tiles), on the
 * stream of the call - bench.py times exactly the kernel its `roofline` prices, with other scenes in flight, instead
 * of the whole op.  NULL, NULL switches it off.  The events stay the caller's. */
int cv_hv_set_kernel_events(void* ev_start, void* ev_stop);

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
 * acceptance order (corners [8][3], score, class).  *h_truncated (may be NULL) is set to 1 when max_iters candidates
 * were examined and a cell >= thresh_high is still live - the reference's `while True` loop (:204-209) would have gone
 * on, so the caller must re-run with a larger max_iters rather than use the result.  Synchronises `stream` once. */
int cv_decode_f32(float* d_grid_obj, const float* d_grid_rot, const float* d_grid_scale,
                  const int dims[3], const float h_corner3[3], float res, const float* d_points,
                  const float* d_xyz, const float* d_prob, const int32_t* d_class, int64_t n,
                  const cv_decode_params* params, int mutate_grid, void* d_ws, size_t ws_bytes,
                  int* h_n_cand, int64_t* h_cand_idx, int32_t* h_verdict, int* h_n_boxes,
                  float* h_boxes, float* h_scores, int32_t* h_classes, int* h_truncated, void* stream);

/* utils/calc_map.py:6-21 on two [8][3] corner sets (host). */
double cv_iou_obb(const float* h_box1, const float* h_box2);
/* eval_joint.py:75-89 (host): greedy NMS, returns the pick count, indices in h_pick. */
int cv_nms_obb(const float* h_boxes, const float* h_scores, int n, double thr, int32_t* h_pick);

/* ------------------------------------------------------------------------ *
 * Sparse-voxel engine: replaces the MinkowskiEngine v0.5.3 calls of the network
 * (external dependency, README.md:53; call sites utils/minkunet.py:53-180,
 * utils/resnet.py:118-154, train_joint.py:250, eval_joint.py:169-171).
 * Coordinates are int32 [n][4] = (batch, x, y, z), features fp32 row-major with an
 * explicit leading dimension, weights are ME's `kernel` layout [K^3][Cin][Cout]
 * ([Cin][Cout] for 1x1), kernel-offset index with the first spatial axis fastest.
 * ------------------------------------------------------------------------ */

/* Hash-table slots needed for a coordinate set of n rows (power of two >= 2n). */
long long cv_sp_table_capacity(long long n);
size_t cv_sp_levels_workspace_bytes(long long n);

/* Coordinate sets of tensor strides 1,2,4,8,16 (ME coordinate manager: stride-2 sets are
 * unique(floor(c / 2ts) * 2ts), ordered by first appearance) and one hash table per level.
 * d_coords[0] is the caller's input (n rows); d_coords[1..], d_keys[L], d_vals[L] are caller
 * allocated (n rows / `cap` slots each).  h_counts[0..4] = rows per level, h_counts[5] = number
 * of duplicate input coordinates (must be 0), h_counts[6] = rows outside the 16-bit key window (spatial
 * coordinates in [-32704, 32703], batch index < 65536; must be 0).  Synchronises `stream` once, or not at all when
 * h_counts is NULL (the counts then stay in d_counts only). */
int cv_sp_build_levels(int32_t* const* d_coords, unsigned long long* const* d_keys,
                       int32_t* const* d_vals, long long n, long long cap, int num_levels,
                       int32_t* d_counts, int32_t* h_counts, void* d_ws, size_t ws_bytes, void* stream);

/* Spatial row order of a coordinate set (what the fused network runs on; replaces the sort of Morton keys by the
 * caller): stable sort on (batch index, Z-order of the 2^shift cubes), rows of one cube in the caller's order.
 * d_sorted[n][4] = rows in that order, d_perm[n] = original row of each sorted row, d_inv[n] = sorted row of each
 * original row.  d_ws: cv_sp_sort_workspace_bytes(n).  Asynchronous, no host synchronisation. */
size_t cv_sp_sort_workspace_bytes(long long n);
int cv_sp_sort_rows(const int32_t* d_coords, long long n, int32_t* d_sorted, int32_t* d_perm, int32_t* d_inv,
                    void* d_ws, size_t ws_bytes, void* stream);

/* Z-order (Morton) sort keys, batch index in the top bits: d_keys[n] int64.  Asynchronous. */
int cv_sp_morton_keys(const int32_t* d_coords, long long n, long long* d_keys, void* stream);

/* Kernel map of a k^3 kernel (ME "kernel map" / neighbour table): d_nbr[n_out][k^3] = input row of
 * out_coord + offset*ts or -1.  Odd k centred, even k offsets 0..k-1.  Asynchronous. */
int cv_sp_kernel_map(const int32_t* d_out_coords, long long n_out, const unsigned long long* d_keys,
                     const int32_t* d_vals, long long cap, int k, int ts, int32_t* d_nbr, void* stream);

/* Map of MinkowskiConvolutionTranspose(kernel_size=2, stride=2) onto the existing finer set,
 * derived from the matching strided map: d_up[n_fine][8]. */
int cv_sp_up_map(const int32_t* d_nbr_down, long long n_coarse, long long n_fine, int32_t* d_up,
                 void* stream);

/* out[u][:] = relu?( (acc_in[u][:] + sum_{j in [j_begin,j_end)} W_j^T in[nbr[u][j]]) * scale + shift
 *                    + residual[u][:] )
 * = MinkowskiConvolution / ConvolutionTranspose with the eval-mode MinkowskiBatchNorm, bias,
 * BasicBlock residual and MinkowskiReLU folded into the epilogue.  Optional members may be NULL / 0. */
typedef struct cv_conv_desc {
    const float* in;        /* [n_in][in_ld] input features */
    long long n_in;
    int in_ld, cin;
    const float* weight;    /* [K][cin][cout] (ME `kernel`) */
    int K, cout;
    const int32_t* nbr;     /* [n_out][K] kernel map; NULL for K == 1 on the same coordinate set */
    long long n_out;
    const float* scale;     /* [cout] or NULL */
    const float* shift;     /* [cout] or NULL */
    const float* residual;  /* [n_out][res_ld] or NULL */
    int res_ld;
    int relu;
    float* out;             /* [n_out][out_ld] */
    int out_ld;
    int flavour;            /* 0 auto (may split offsets over workgroups through ws), 1 never split */
    void* ws;               /* optional workspace, cv_sp_conv_workspace_bytes */
    size_t ws_bytes;
    const int32_t* row_perm;/* optional [n_out] processing order (rows with equal neighbour masks adjacent) */
    int j_begin, j_end;     /* kernel offsets to accumulate; j_end == 0 means K */
    const float* acc_in;    /* optional [n_out][acc_ld] partial sums from a previous launch */
    int acc_ld;
    int perm_groups;        /* >1: row_perm is [perm_groups][n_out]; the offsets are split into that many
                               contiguous groups, each run in its own order in ONE launch (needs ws) */
    const int32_t* plan_ent;/* reserved, must be NULL (round 1-3: the pair lists of an experimental tile kernel, removed) */
    const int32_t* plan_cnt;/* reserved, must be NULL */
    const float* weight_packed; /* reserved, must be NULL */
    const void* weight_x6;  /* optional: the same weights from cv_sp_pack_weights_x6_f32.  When given (and Cin % 32 == 0)
                               the products run on the bf16 matrix cores as six bf16 x bf16 piece products per fp32
                               product (fp32-level accuracy, 0.375x the matrix time of v_mfma_f32_32x32x2_f32) */
    const float* in2;       /* optional second source on the OUTPUT rows (bf16x6 path only): the accumulator also gets */
    int in2_ld, cin2;       /* in2[u][0:cin2] @ W2 - BasicBlock's 1x1 downsample branch folded into conv2             */
    const void* weight2_x6; /* cv_sp_pack_weights_x6_f32 of W2 [1][cin2][cout]                                         */
    int perm_has_map;       /* with perm_groups > 1: row_perm is followed by the kernel map rows in processing order
                               (cv_sp_mask_perms with_map = 1), which turns the random map reads into coalesced ones */
    int weight_pieces;      /* 1: one bf16 plane (cv_sp_pack_weights_bf16_f32): operands rounded to bf16, one product -
                               the opt-in bf16 compute mode, NOT fp32-level accuracy;
                               0 / 3: weight_x6 (and weight2_x6) hold bf16 triples (cv_sp_pack_weights_x6_f32);
                               2: fp16 pairs (cv_sp_pack_weights_h2_f32): three fp16 x fp16 piece products per fp32
                               product - fp32-level accuracy (operands to 2^-24) while every input magnitude is below
                               65504, half the matrix time of the triples                                            */
    float acc_scale;        /* fp16 pairs: 2^-scale_log2 of the pack call (0 = 1): multiplies the accumulators        */
    int32_t* range_flag;    /* fp16 pairs: optional device-visible word, set to 1 when an input magnitude > 65000 was
                               staged (the result is then invalid: redo the convolution with the bf16 triples)        */
    int in_hl, out_hl, res_hl; /* 1: in (and in2) / out / residual are in the hl format instead of fp32 - the fp16 pair
                               (h, l) of every value stored in place: a row of C channels (C % 32 == 0) keeps its 4*C
                               bytes, each 32-channel chunk = 64 bytes of high pieces then 64 bytes of low pieces
                               (cv_sp_to_hl_f32 / cv_sp_from_hl_f32).  The consumer then loads its matrix-core operand
                               fragments straight from global memory and nothing is split per gather; the producer
                               raises range_flag when an OUTPUT magnitude exceeds 65000.  Needs weight_pieces = 2 for
                               in_hl, channel counts and leading dimensions % 32 == 0, 128-byte aligned rows.        */
    int32_t* split_tickets; /* optional, hl-format input only: CV_SPLIT_TICKETS zero-initialised ints owned by the caller's
                               stream.  A split launch then reduces its partial tiles in the last-arriving workgroup of every
                               output tile (same summation order as the finish launch: bit-identical) instead of a second
                               launch; the library leaves the counters at zero.  NULL: two launches. */
    const float* acc_scale_dev; /* optional, hl-format input only: a device scalar multiplied into acc_scale when the kernel
                               runs (the training backward scales a layer's gradient rows by a power of two chosen on the
                               device, cv_sp_bn_backward_hl_f32, and hands the inverse over here: no host wait). */
} cv_conv_desc;
#define CV_SPLIT_TICKETS 4096

/* fp32 rows -> hl format and back (d_x and d_y may not alias; c % 32 == 0; the hl side 128-byte aligned with a leading
 * dimension % 32 == 0; leading dimensions in 4-byte units on both sides).  range_flag (optional) as in cv_conv_desc. */
int cv_sp_to_hl_f32(const float* d_x, long long n, int c, int x_ld, float* d_y, int y_ld, int32_t* range_flag, void* stream);
int cv_sp_from_hl_f32(const float* d_x, long long n, int c, int x_ld, float* d_y, int y_ld, void* stream);

/* Weights of the bf16x6 path: [K][cin][cout] fp32 split into three bf16 pieces per value (h + m + l == value to half
 * an fp32 ulp), laid out per (offset, 32-channel chunk) as [piece][cout][32]: 3*K*cin*cout 16-bit words, 16-byte aligned;
 * cin % 32 == 0; redo whenever the weights change.  Asynchronous. */
int cv_sp_pack_weights_x6_f32(const float* d_w, int K, int cin, int cout, const float* d_col_scale, void* d_wp6,
                              void* stream);   /* d_col_scale[cout] (optional): weights are multiplied per output
                                                  column before the split (a folded BatchNorm scale) */
/* Weights of the fp16-pair path: (w * d_col_scale * 2^scale_log2) split into h + l fp16 pieces, same layout with two
 * planes: 2*K*cin*cout 16-bit words.  The caller picks scale_log2 (largest scaled magnitude about 2^13; both weight
 * sets of a two-source convolution share it) and passes acc_scale = 2^-scale_log2 in the descriptor. */
int cv_sp_pack_weights_h2_f32(const float* d_w, int K, int cin, int cout, const float* d_col_scale, int scale_log2,
                              void* d_wp, void* stream);
/* cv_sp_pack_weights_h2_f32 / cv_sp_pack_weights_t_f32 (pieces 2) for MANY weight tensors in ONE launch: a training step packs
 * the fp16 pairs of all 63 convolutions - forward and transposed - in front of its forward instead of one launch per layer
 * and direction (117 launches and the stream's idle time in front of them, profiles/r5/train_gaps.txt).  Job i packs
 * w[K][cin][cout] (trans = 0) or, trans = 1, the transposed convolution of a FORWARD kernel w[K][cout][cin] (what
 * cv_sp_pack_weights_t_f32(w, K, rows = cout, cols = cin, 2, ...) makes) into wp: 2*K*cin*cout 16-bit words, 16-byte
 * aligned, no column scale.  h_jobs: host array, copied to d_jobs by hipMemcpyAsync on `stream`: ordinary (pageable) memory is
 * staged before the call returns and may be reused at once; a PINNED array is read when the copy runs on the stream and must stay
 * unchanged until then.  d_jobs: n_jobs * sizeof(cv_pack_job) bytes of device scratch that must stay untouched until the launch
 * has run. */
typedef struct cv_pack_job {
    const float* w;
    void* wp;
    int K, cin, cout;
    int trans;
    int scale_log2;
    int reserved;
} cv_pack_job;
int cv_sp_pack_weights_h2_batch_f32(const cv_pack_job* h_jobs, int n_jobs, void* d_jobs, void* stream);
/* Weights of the opt-in bf16 compute mode (cv_conv_desc.weight_pieces = 1): w * d_col_scale rounded to bf16 (RNE),
 * one plane of the same layout: K*cin*cout 16-bit words.  Not an fp32-parity path (operands carry 8 significant
 * bits); the reference trains and evaluates in fp32 (train_joint.py:218), BASELINE config 3 asks for bf16. */
int cv_sp_pack_weights_bf16_f32(const float* d_w, int K, int cin, int cout, const float* d_col_scale, void* d_wp,
                                void* stream);
/* Packed weights of the TRANSPOSED convolution (the input gradient of a layer: the same kernel on the transposed map with
 * W_j^T) straight from the layer's forward weights d_w[K][rows][cols]: what cv_sp_pack_weights_{x6,h2,bf16}_f32 would make
 * of W'[K][cols][rows], W'[j][a][b] = d_w[j][b][a], without materialising W'.  cols % 32 == 0.  pieces: 3 bf16 triples,
 * 2 fp16 pairs (times 2^scale_log2), 1 one bf16 plane. */
int cv_sp_pack_weights_t_f32(const float* d_w, int K, int rows, int cols, int pieces, int scale_log2, void* d_wp, void* stream);
/* Weights of the matrix-core stem (Cin 3 or 6 -> 32 channels, K <= 128 offsets; cv_conv_desc with weight_pieces = 2 and
 * this buffer in weight_x6): (w * d_col_scale * 2^scale_log2) as fp16 pairs in MFMA B-operand order, 8 * cin * 2048 bytes;
 * pass acc_scale = 2^-scale_log2 and no `scale` in the descriptor. */
int cv_sp_pack_weights_stem_h2_f32(const float* d_w, int K, int cin, const float* d_col_scale, int scale_log2, void* d_wp,
                                   void* stream);

size_t cv_sp_conv_workspace_bytes(long long n_out, int cout, int K);
int cv_sp_conv_f32(const cv_conv_desc* desc, void* stream);

/* Launch sizing of the convolutions whose output tiles alone do not fill the chip (the coarse levels): the number of
 * workgroups a launch is split up to (over the kernel offsets; partial tiles, then a finish pass).  Default 768
 * (or CV_SPLIT_TARGET) - best for ONE scene in flight; a host that keeps several scenes in flight on separate streams
 * lets the other scenes fill the chip and sets a smaller value (bench.py: 256 from four scenes in flight).  Results
 * change in the summation order only (fp32 rounding).  workgroups <= 0 restores the default.  Returns the previous
 * value.  Process-wide; set it before launching, not concurrently with cv_sp_conv_f32 / cv_sp_net_forward callers that
 * need one fixed value.  (No reference counterpart: MinkowskiEngine's launch sizes are internal.) */
int cv_sp_set_split_target(int workgroups);
/* The same for the CALLING THREAD's launches only (0 = follow the process-wide value); returns the previous thread value.
 * cv_detect_scene_f32 uses it to size a scene's launches by how many scenes are in flight when it starts. */
int cv_sp_set_split_target_thread(int workgroups);
/* Measurement hook (no reference counterpart; results are WRONG while a bit is set): timing ablations that can be switched
 * at run time, after a warm-up has filled every buffer with valid values.  bit 0: the finish launches of the split / mask-group
 * convolutions are skipped (what the partial-tile reductions cost with scenes in flight).  Returns the previous bits. */
int cv_sp_set_ablation(int bits);
/* Kernel selection knobs of cv_sp_conv_f32 / cv_net_run_f32 (process-wide, like cv_sp_set_split_target; results are
 * bit-identical under every setting).  "hd_mask": bit NB - 1 sends the hl-format convolutions whose workgroups are
 * NB x 32 columns wide to conv_hd (LDS-DMA operand rings, 256-row workgroups) instead of conv_hl when the launch has at
 * least "hd_min_rows" output rows; "hd_shape": 0 = 8 waves x 3 ring stages (one workgroup per CU), 1 = 4 waves x 2 stages
 * (two per CU), 2 = 8 waves x 2 stages.  *previous (may be NULL) receives the old value.  Environment defaults: CV_HD,
 * CV_HD_MIN_ROWS, CV_HD_SHAPE. */
int cv_sp_set_option(const char* name, long long value, long long* previous);
/* Reads a knob of cv_sp_set_option without touching it (other threads may be launching). */
int cv_sp_get_option(const char* name, long long* value);

/* Every kernel map and processing order the fused MinkUNet forward needs, built by ONE call per scene into one
 * int32 arena (offsets in int32 words; -1 = absent):
 * (level-0 items first - stem, k3[0], mask_perm[0], scratch - then the coarse levels)
 *   stem  [rows0][stem_k^3]  sorted rows <- rows of the caller's order (the sorted set's own map with d_perm, the
 *                            sorted <- original permutation of cv_sp_sort_rows, folded in)
 *   out                      always -1: the caller's rows <- sorted rows map is cv_sp_sort_rows' d_inv
 *   down[i] [rows(i+1)][8]   k2s2 conv level i -> i+1;   k3[i] [rows(i)][27];   up[i] [rows(3-i)][8] level 4-i -> 3-i
 *   mask_perm[i] [groups][rows(i)] for levels with >= masked_min_rows rows;  up_perm[i] [rows(3-i)] octant order
 * d_coords / d_keys / d_vals: the five levels of the (Z-order sorted) coordinate set from cv_sp_build_levels. */
typedef struct cv_scene_maps {
    long long stem, out, down[4], k3[5], up[4], mask_perm[5], up_perm[4], scratch;
    long long bitmap;       /* 2^20 words: occupancy bits of the level-0 set over its bounding box (cv_sp_scene_plan puts
                               them in front of the hash probes of the level-0 maps: 87 % of the lookups are misses) */
} cv_scene_maps;
size_t cv_sp_scene_maps_words(const long long* level_rows, long long n_orig, int stem_k, int mask_groups,
                              long long masked_min_rows, cv_scene_maps* offsets);
int cv_sp_scene_maps(int32_t* const* d_coords, const unsigned long long* const* d_keys, const int32_t* const* d_vals,
                     long long cap, const long long* level_rows, const int32_t* d_perm, long long n_orig, int stem_k,
                     int mask_groups, long long masked_min_rows, int32_t* d_arena, size_t arena_words, void* stream);

/* The whole coordinate plan of a scene in ONE call (replaces cv_sp_sort_rows + cv_sp_build_levels + cv_sp_scene_maps
 * issued by the caller): spatial row sort of d_input[n][4] into d_coords[0], the five levels with their tables, every
 * kernel map and processing order of the fused network into d_arena.  The arena is sized before the coarse row counts are
 * known (cv_sp_scene_plan_words: every level bounded by n); *offsets and h_counts[8] (cv_sp_build_levels' counts) are
 * filled on return.  The level counts are copied to pinned host memory behind the levels and the call waits for that copy
 * only (an event), while `stream` goes on with the level-0 maps queued behind it.  When h_counts[5] (duplicates) or h_counts[6] (rows
 * outside the key window) is non-zero nothing beyond the level-0 maps is built and the caller must reject the input. */
size_t cv_sp_scene_plan_words(long long n, int stem_k, int mask_groups, long long masked_min_rows);
int cv_sp_scene_plan(const int32_t* d_input, long long n, int32_t* d_perm, int32_t* d_inv, int32_t* const* d_coords,
                     unsigned long long* const* d_keys, int32_t* const* d_vals, long long cap, int32_t* d_counts,
                     int32_t* h_counts, int stem_k, int mask_groups, long long masked_min_rows, int32_t* d_arena,
                     size_t arena_words, cv_scene_maps* offsets, void* d_sort_ws, size_t sort_ws_bytes, void* d_levels_ws,
                     size_t levels_ws_bytes, void* stream);

/* Fused eval-mode network as ONE call per scene (host-side executor over cv_sp_conv_f32; replaces the reference's
 * module-by-module MinkUNet34C.forward, utils/minkunet.py:122-180, for inference).  The program is symbolic and built
 * once per model: feature buffers are slots of a per-scene arena, kernel maps / processing orders are slots of
 * per-scene pointer tables.
 *   cv_net_buf: level >= 0: arena buffer of level_rows[level] x channels floats;
 *               level < 0: the caller's tensor - takes the next entry of ext_ptr / ext_ld, rows = level_rows[rows_level].
 *   cv_net_op:  out[:, out_col:out_col+cout] = relu?( conv(in[:, in_col:in_col+cin]) * scale + shift
 *               + res[:, res_col:res_col+cout] ), kernel map maps[map] (map < 0: K == 1), processing order
 *               perms[perm] (perm < 0 or a NULL table entry: natural order; perm_groups > 1: mask-sorted groups). */
typedef struct cv_net_buf {
    int level, channels, rows_level;
    int hl;                    /* 1: the buffer holds the hl format (cv_conv_desc.in_hl): every op reading / writing it runs
                                  with the matching flag; arena buffers only */
} cv_net_buf;
typedef struct cv_net_op {
    int in_buf, in_col, cin;
    int out_buf, out_col, cout;
    int res_buf, res_col;      /* res_buf < 0: no residual */
    int map, K;
    int perm, perm_groups;
    int relu;
    const float* weight;       /* [K][cin][cout] */
    const float* scale;        /* [cout] or NULL */
    const float* shift;        /* [cout] or NULL */
    const void* weight_x6;     /* cv_sp_pack_weights_x6_f32 of weight, or NULL */
    int in2_buf, in2_col, cin2;/* second source (in2_buf < 0: none), see cv_conv_desc.in2 */
    const void* weight2_x6;
    int weight_pieces;         /* see cv_conv_desc.weight_pieces / acc_scale */
    float acc_scale;
} cv_net_op;
size_t cv_net_arena_bytes(const cv_net_buf* bufs, int n_bufs, const long long* level_rows, int n_levels);
int cv_net_run_f32(const cv_net_op* ops, int n_ops, const cv_net_buf* bufs, int n_bufs, const long long* level_rows,
                   int n_levels, void* d_arena, size_t arena_bytes, const void* const* ext_ptr, const int* ext_ld,
                   const int32_t* const* maps, int n_maps, const int32_t* const* perms, int n_perms,
                   void* d_ws, size_t ws_bytes, int32_t* range_flag, void* stream);   /* range_flag: cv_conv_desc.range_flag of every fp16-pair op, or NULL */

/* d_keys[n] (int64) = bit mask of the valid neighbours among offsets [j_begin, j_end) of every row of a
 * kernel map; argsort of it is a row_perm for cv_conv_desc.  Asynchronous. */
int cv_sp_mask_keys(const int32_t* d_nbr, long long n, int K, int j_begin, int j_end, long long* d_keys,
                    void* stream);

/* d_perm[groups][n]: for each contiguous group of the K kernel offsets, the rows ordered by the bit
 * mask of their valid neighbours in that group (counting sort; at most 10 offsets per group).
 * d_ws: groups * 4096 bytes.  with_map = 1: d_perm holds groups*n*(1 + W) + ceil(groups*n/4) words, W = ceil(K/groups); after the
 * orders come the kernel map rows of every group in processing order, [groups][n][W]
 * (cv_conv_desc.perm_has_map).  Asynchronous. */
int cv_sp_mask_perms(const int32_t* d_nbr, long long n, int K, int groups, int32_t* d_perm, void* d_ws,
                     size_t ws_bytes, int with_map, void* stream);

/* Training support (train_joint.py:283 `loss.backward()` through the sparse convolutions).
 * Input gradient = cv_sp_conv_f32 on the transposed map with the transposed weights:
 *   d_nbr_t[n_in][K], nbr_t[i][j] = u iff nbr[u][j] == i.
 * Weight gradient dW[j][ci][co] = sum_u x[nbr[u][j]][ci] * dy[u][co] (fp32 matrix cores, split over rows). */
int cv_sp_transpose_map(const int32_t* d_nbr, long long n_out, int K, long long n_in, int32_t* d_nbr_t, void* stream);
size_t cv_sp_wgrad_workspace_bytes(long long n_out, int cin, int cout, int K);
int cv_sp_conv_wgrad_f32(const float* d_x, int x_ld, int cin, const float* d_dy, int dy_ld, int cout,
                         const int32_t* d_nbr, int K, long long n_out, float* d_dw, void* d_ws, size_t ws_bytes,
                         void* stream);
/* The same with the product precision chosen by the caller: pieces = 0 fp32 matrix cores, 3 six bf16 piece products
 * per fp32 product (fp32-level accuracy; what cv_sp_conv_wgrad_f32 runs), 1 operands rounded to bf16 and ONE
 * bf16 x bf16 product with fp32 accumulation (the opt-in bf16 compute mode, BASELINE configs 3-4). */
int cv_sp_conv_wgrad_px_f32(const float* d_x, int x_ld, int cin, const float* d_dy, int dy_ld, int cout,
                            const int32_t* d_nbr, int K, long long n_out, float* d_dw, void* d_ws, size_t ws_bytes,
                            int pieces, void* stream);
/* out[c] = sum over rows of x[:, c] (bias gradient).  cv_sp_col_sum_f32 joins the row chunks through fp32 atomics: the
 * last bit depends on the order in which workgroups arrive.  cv_sp_col_sum_det_f32 writes the chunk sums to a workspace
 * (cv_sp_col_sum_workspace_bytes) and adds them in chunk order: the same bits on every run (what the training path uses,
 * so that a train_joint.py step is reproducible bit for bit). */
int cv_sp_col_sum_f32(const float* d_x, long long n, int c, int ld, float* d_out, void* stream);
size_t cv_sp_col_sum_workspace_bytes(long long n, int c);
int cv_sp_col_sum_det_f32(const float* d_x, long long n, int c, int ld, float* d_out, void* d_ws, size_t ws_bytes,
                          void* stream);

/* Training-mode MinkowskiBatchNorm (= nn.BatchNorm1d over the feature rows, utils/minkunet.py:56):
 * batch mean / biased variance per channel, running statistics updated in place (NULL to skip), folded
 * scale/shift for cv_sp_affine_f32; and its backward (d_y = output of the ReLU that follows, or NULL;
 * d_dres = optional ReLU-masked gradient for a residual added before that ReLU, resnet_block forward). */
size_t cv_sp_bn_workspace_bytes(int c);
int cv_sp_bn_stats_f32(const float* d_x, long long n, int c, int ld, const float* d_gamma, const float* d_beta,
                       float eps, float momentum, float* d_running_mean, float* d_running_var, float* d_mean,
                       float* d_var, float* d_scale, float* d_shift, void* d_ws, size_t ws_bytes, void* stream);
int cv_sp_bn_backward_f32(const float* d_x, const float* d_dy, const float* d_y, long long n, int c, int ld,
                          const float* d_mean, const float* d_var, float eps, const float* d_gamma, float* d_dgamma,
                          float* d_dbeta, float* d_dx, float* d_dres, void* d_ws, size_t ws_bytes, void* stream);

/* cv_sp_bn_backward_f32 with a second output for the input-gradient convolution of the layer below (the eval path's hl-format
 * kernels on the transposed map): d_dx_hl = d_dx * s as fp16 pairs, s the power of two that put the PREVIOUS call's largest
 * |dx| of this layer into [2^9, 2^10) - a factor of 64 below the fp16 range; *range_flag is raised beyond it.  d_slot =
 * CV_BN_SLOT_WORDS 32-bit words of the layer that live across steps: [0, 4096) receive this call's largest |dx| per workgroup
 * (bits of non-negative floats, plain stores), [4096, 8192) hold the previous step's (the caller copies the first half there
 * and zeroes it between steps; all zero = unknown: s = 1), [8192] receives 1 / s as a float for cv_conv_desc.acc_scale_dev.
 * c % 32 == 0, ld % 32 == 0, 128-byte aligned rows.  d_dx_hl and d_slot may both be NULL (no twin).  d_relu_bits (optional): the
 * ReLU mask as cv_sp_affine_hl_f32 wrote it, read instead of d_y (which may then be NULL). */
#define CV_BN_SLOT_WORDS 8200
int cv_sp_bn_backward_hl_f32(const float* d_x, const float* d_dy, const float* d_y, long long n, int c, int ld,
                             const float* d_mean, const float* d_var, float eps, const float* d_gamma, float* d_dgamma,
                             float* d_dbeta, float* d_dx, float* d_dres, void* d_ws, size_t ws_bytes, float* d_dx_hl,
                             uint32_t* d_slot, int32_t* range_flag, const uint32_t* d_relu_bits, void* stream);

/* Multi-tensor copy with a device-side guard: h_dst[i][0 .. h_bytes[i]) = h_src[i][...] for i < n (device pointers in host
 * arrays, sizes multiples of 4 bytes; ceil(n / 96) launches) UNLESS *flag != 0 when the launch runs (flag: device-visible
 * int32, NULL = always copy).  The training step snapshots the BatchNorm running statistics with it in front of every
 * forward: once a forward has left the fp16 range (cv_conv_desc.range_flag, sticky until the host resets it) the snapshot stays
 * the state before the FIRST flagged step, whatever was queued since - the host restores from it when it notices
 * (canonicalvoting_amd/train.py; the reference's BatchNorm statistics, train_joint.py:250, see every batch exactly once).
 * Asynchronous. */
int cv_sp_copy_unless_flag(const void* const* h_src, void* const* h_dst, const long long* h_bytes, int n, const int32_t* flag,
                           void* stream);

/* y = relu?(x * scale + shift + residual): MinkowskiBatchNorm (eval) / MinkowskiReLU / the residual add of
 * BasicBlock on feature rows; scale, shift and residual may each be NULL. */
int cv_sp_affine_f32(const float* d_x, long long n, int c, int x_ld, const float* d_scale,
                     const float* d_shift, const float* d_residual, int res_ld, int relu, float* d_y, int y_ld,
                     void* stream);
/* The same pass with a second output: d_y_hl receives the values of d_y in the hl format (the fp16 pairs the hl-format
 * convolutions multiply, cv_conv_desc.in_hl; c % 32 == 0, 128-byte aligned rows) and *range_flag is raised when one exceeds
 * 65000.  The training forward feeds the next convolution from d_y_hl and keeps d_y for autograd (BatchNorm backward, the
 * weight gradient, residual adds).  d_relu_bits (optional, with relu): [n][c / 32] words, bit i of word (r, q) = (y[r][32 q + i] > 0)
 * - what the BatchNorm backward reads instead of the rows of y (cv_sp_bn_backward_hl_f32). */
int cv_sp_affine_hl_f32(const float* d_x, long long n, int c, int x_ld, const float* d_scale, const float* d_shift,
                        const float* d_residual, int res_ld, int relu, float* d_y, int y_ld, float* d_y_hl, int y_hl_ld,
                        uint32_t* d_relu_bits, int32_t* range_flag, void* stream);

/* scale = gamma / sqrt(var + eps), shift = beta - mean * scale (+ bias * scale). */
int cv_sp_bn_fold_f32(const float* d_gamma, const float* d_beta, const float* d_mean, const float* d_var,
                      const float* d_bias, float eps, int c, float* d_scale, float* d_shift, void* stream);

/* Per-point head select of the joint model (eval_joint.py:173-190). */
int cv_head_joint_f32(const float* d_feats, long long n, int ld, int nclasses, int log_scale, float* d_xyz,
                      float* d_scale, float* d_prob, int32_t* d_class, void* stream);

/* 8-channel head of a per-category model (eval_separate.py:170-181): xyz, exp(scale), softmax(obj)[1]. */
int cv_head_separate_f32(const float* d_feats, long long n, int ld, int log_scale, float* d_xyz, float* d_scale,
                         float* d_prob, void* stream);

/* ------------------------------------------------------------------------ *
 * One call per scene: eval_joint.py:163-280 (network -> head split -> vote -> decode -> per-class NMS) behind ONE
 * entry point.  Same kernels in the same order as the call-by-call path (cv_sp_scene_plan, cv_net_run_f32,
 * cv_head_joint_f32, cv_hv_forward_f32, cv_decode_f32, cv_nms_obb): bit-identical results.  The host waits twice (level
 * counts of the coordinate plan - the bounds of the points arrive with them - and the decode results); a binding that
 * releases its interpreter lock around foreign calls (ctypes does) keeps it released for the whole scene.
 * ------------------------------------------------------------------------ */
typedef struct cv_scene_desc {
    /* the scene */
    const int32_t* d_coords4;     /* [n][4] (batch, x, y, z), unique rows (eval_joint.py:169).  ONE scene: the spatial row sort does not
                                     sort by the batch column (two launches fewer); rows with different batch indices stay distinct
                                     voxels (every table key carries the index), they only lose their batch-major grouping */
    long long n;
    const float* d_feats;         /* [n][feats_ld] network input features (eval_joint.py:167-168) */
    int feats_ld;
    const float* d_points;        /* [n][3] world points = coords * res (eval_joint.py:193) */
    float res;
    int num_rots;
    /* the network: program of MinkUNet34C.forward (cv_net_run_f32) and the sizes its coordinate plan is built with */
    const cv_net_op* ops; int n_ops;
    const cv_net_buf* bufs; int n_bufs;
    int stem_k, mask_groups;
    long long masked_min_rows;
    int max_channels;             /* widest convolution output (workspace sizing) */
    int use_range_flag;           /* 1: the program runs on fp16 pairs; result.range_flag reports an input beyond the fp16 range */
    float* d_out_feats;           /* [n][out_ld] network output, caller's buffer, caller's row order */
    int out_ld, out_channels;
    int nclasses, log_scale;      /* head split (eval_joint.py:173-190) */
    /* optional: predictions fed to vote + decode instead of the network's (NULL = the network's) */
    const float* d_xyz_in; const float* d_scale_in; const float* d_prob_in; const int32_t* d_class_in;
    int vote_algo;                /* cv_hv_forward_f32 algo (0 auto) */
    cv_decode_params decode;      /* max_iters is taken from max_candidates */
    int max_candidates;           /* capacity of h_cand_idx / h_verdict / h_boxes / h_scores / h_classes / h_pick */
    double nms_threshold;         /* eval_joint.py:273 (0.3) */
    /* scratch: grown by the caller to result.needed_* after a CV_ENOMEM return */
    void* d_ws; size_t ws_bytes;
    float* d_grids; size_t grid_capacity_floats;   /* optional caller buffer for the three grids (6 floats per cell); NULL: in d_ws */
    void* h_pinned; size_t pinned_bytes;           /* >= 256 bytes of page-locked host memory */
    /* host results */
    int64_t* h_cand_idx; int32_t* h_verdict;
    float* h_boxes; float* h_scores; int32_t* h_classes;      /* accepted boxes in acceptance order: [k][8][3], [k], [k] */
    int32_t* h_pick;                                          /* detections after per-class NMS: indices into the box list, class by class */
    int adaptive_split;           /* 1: launch sizing of the coarse-level convolutions by the scenes inside cv_detect_scene_f32 when this
                                     one starts - the one-scene optimum (768 workgroups) below four, 256 from four on (what
                                     bench.py sets by hand for the call-by-call path); 0: the process-wide cv_sp_set_split_target */
    /* optional measurement hook: hipEvent_t handles recorded on `stream` at the scene's start, behind the network, the head
     * split, the vote and the decode (NULL entries are skipped) */
    void* events[5];
    /* launch sizing of THIS call (with masked_min_rows above: the three choices that depend on how many scenes the host keeps
     * in flight - every integer output stays the same bits, the network output moves in fp32 summation order only).  0 = the
     * calling thread's / process-wide value (cv_sp_set_split_target[_thread], cv_hv_set_part_records[_thread]).
     * conv_split_target > 0 overrides adaptive_split. */
    int conv_split_target;        /* workgroups a split coarse-level convolution aims at (library default 768; 256 from four scenes in flight) */
    int vote_part_records;        /* records one workgroup of a hot (tile, plane) takes (library default 4096; 12288 from four in flight) */
} cv_scene_desc;
typedef struct cv_scene_result {
    int n_cand, n_boxes, n_det, truncated, range_flag, duplicates, out_of_window;
    int scenes_in_flight;          /* scenes inside cv_detect_scene_f32 when this one started (itself included) */
    int dims[3];
    float corner[3];
    long long level_rows[5];
    size_t needed_ws_bytes, needed_grid_floats;
    /* device views into d_ws / d_grids, valid until the next call that uses the same scratch */
    float* d_grid_obj; float* d_grid_rot; float* d_grid_scale;
    float* d_xyz; float* d_scale; float* d_prob; int32_t* d_class;      /* the network's own head outputs */
    /* host time the call spent (microseconds): [0] coordinate plan incl. its wait for the level counts, [1] enqueueing the network
     * program (~100 launches, no wait), [2] head + vote enqueue, [3] decode incl. its wait for the results + NMS */
    float host_us[4];
} cv_scene_result;
int cv_detect_scene_f32(const cv_scene_desc* desc, cv_scene_result* result, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CV_HIP_H */
