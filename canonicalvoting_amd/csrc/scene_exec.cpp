// One C call per scene: eval_joint.py:163-280 behind a single C-ABI entry point.
//
//   coordinate plan (cv_sp_scene_plan) -> MinkUNet34C forward (cv_net_run_f32) -> head split (cv_head_joint_f32)
//   -> vote (cv_hv_forward_f32) -> decode (cv_decode_f32) -> per-class NMS (cv_nms_obb)
//
// The Python pipeline issues the same entry points one by one (~40 ctypes calls, tensor allocations and pointer tables
// per scene, every one of them with the GIL held); a host that keeps several scenes in flight from several threads
// (bench.py: eight) then serialises the enqueue work of all of them.  Here a scene thread makes ONE foreign call - the
// GIL is released for its whole duration - and the host waits twice inside it: for the level counts of the coordinate
// plan (the bounds of the points, reduced in front of the plan, arrive with them: no wait of their own for the vote
// grid's shape) and for the decode results.  Same kernels, same launch order, same arguments: results are bit-identical
// to the call-by-call path (tests/test_scene_call_gpu.py).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "cv_common.h"

namespace {

inline size_t up64(size_t v) { return (v + 63) / 64 * 64; }

struct Carver {
    char* base; size_t off = 0, cap;
    Carver(void* p, size_t c) : base(static_cast<char*>(p)), cap(c) {}
    // returns nullptr when the workspace is too small; `need` keeps counting so that the caller learns the full size
    template <typename T> T* take(size_t count) {
        off = cv_align_up(off, 256);
        T* r = (base && off + count * sizeof(T) <= cap) ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return r;
    }
};

// CV_SCENE_SERIALIZE=1 (experiment): the pure-enqueue part of a scene (network program, head, vote: ~110 launches) is issued
// under one process-wide lock, as the interpreter lock did for the call-by-call path
std::atomic<int> g_scenes_inside{0};
struct SceneCount {
    int before;
    bool set_target, set_records;
    int prev_target = 0, prev_records = 0;
    // the launch sizing of THIS call (cv_scene_desc.conv_split_target / vote_part_records / adaptive_split) lives in the calling
    // thread's values for the duration of the call: two hosts with different policies in one process do not see each other
    SceneCount(bool adaptive, int split_target, int part_records)
        : before(g_scenes_inside.fetch_add(1, std::memory_order_relaxed)), set_target(adaptive || split_target > 0),
          set_records(part_records > 0) {
        // adaptive: scenes in flight (this one included) when the scene starts: the other scenes fill the chip from about four on,
        // below that the deeper splits of the one-scene optimum pay (profiles/r3/split_target_8streams.txt)
        if (set_target) prev_target = cv_sp_set_split_target_thread(split_target > 0 ? split_target : before + 1 >= 4 ? 256 : 768);
        if (set_records) prev_records = cv_hv_set_part_records_thread(part_records);
    }
    ~SceneCount() {
        if (set_target) cv_sp_set_split_target_thread(prev_target);
        if (set_records) cv_hv_set_part_records_thread(prev_records);
        g_scenes_inside.fetch_sub(1, std::memory_order_relaxed);
    }
};
std::mutex g_enqueue_mu;
bool serialize_enqueue() {
    static const bool on = getenv("CV_SCENE_SERIALIZE") && atoi(getenv("CV_SCENE_SERIALIZE")) != 0;
    return on;
}

}  // namespace

extern "C" {

int cv_detect_scene_f32(const cv_scene_desc* d, cv_scene_result* r, void* stream) {
    CV_REQUIRE(d && r, CV_EINVAL, "null scene descriptor / result");
    CV_REQUIRE(d->d_coords4 && d->d_feats && d->d_points && d->n > 0 && d->ops && d->bufs && d->n_ops > 0 && d->n_bufs > 0,
               CV_EINVAL, "bad scene descriptor");
    CV_REQUIRE(d->d_out_feats && d->out_ld >= d->out_channels && d->out_channels > 0, CV_EINVAL, "bad network output buffer");
    CV_REQUIRE(d->h_pinned && d->pinned_bytes >= 256 && d->d_ws, CV_EINVAL, "pinned scratch (>= 256 bytes) and a device workspace are required");
    CV_REQUIRE(d->h_boxes && d->h_scores && d->h_classes && d->h_cand_idx && d->h_verdict && d->h_pick && d->max_candidates > 0,
               CV_EINVAL, "null result arrays");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long n = d->n;
    const int NL = 5;
    std::memset(r, 0, sizeof(*r));
    CV_REQUIRE(d->conv_split_target >= 0 && d->vote_part_records >= 0, CV_EINVAL, "negative launch sizing");
    SceneCount in_flight(d->adaptive_split != 0, d->conv_split_target, d->vote_part_records);
    r->scenes_in_flight = in_flight.before + 1;
    Carver cv(d->d_ws, d->ws_bytes);
    auto mark = [&](int i) { return d->events[i] ? hipEventRecord(static_cast<hipEvent_t>(d->events[i]), st) : hipSuccess; };
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](int i) {
        const auto t = std::chrono::steady_clock::now();
        r->host_us[i] += std::chrono::duration<float, std::micro>(t - t_prev).count();
        t_prev = t;
    };
    CV_HIP_CHECK(mark(0));

    // ---- bounds of the points (the vote grid's origin and shape): reduced first, read after the plan's own host wait
    float* h_minmax = static_cast<float*>(d->h_pinned);               // 6 floats; [8] = range flag landing word
    int32_t* h_flag = reinterpret_cast<int32_t*>(static_cast<char*>(d->h_pinned) + 64);
    void* mm_ws = cv.take<char>(cv_hv_minmax_workspace_bytes());
    // ---- coordinate plan buffers (the layout of CoordinateManager.fused_fast)
    const long long cap = cv_sp_table_capacity(n);
    const size_t words = cv_sp_scene_plan_words(n, d->stem_k, d->mask_groups, d->masked_min_rows);
    const size_t o_perm = 0, o_inv = up64((size_t)n);
    size_t o_coords[NL], o_vals[NL];
    for (int i = 0; i < NL; ++i) o_coords[i] = o_inv + up64((size_t)n) + (size_t)i * up64(4 * (size_t)n);
    for (int i = 0; i < NL; ++i) o_vals[i] = o_coords[NL - 1] + up64(4 * (size_t)n) + (size_t)i * up64((size_t)cap);
    const size_t o_counts = o_vals[NL - 1] + up64((size_t)cap), o_arena = o_counts + 64;
    int32_t* ibuf = cv.take<int32_t>(o_arena + words);
    unsigned long long* kbuf = cv.take<unsigned long long>((size_t)NL * (size_t)cap);
    const size_t sws_b = cv_sp_sort_workspace_bytes(n), lws_b = cv_sp_levels_workspace_bytes(n);
    char* sort_ws = cv.take<char>(sws_b);
    char* lev_ws = cv.take<char>(lws_b);
    float* xyz = cv.take<float>((size_t)n * 3);
    float* scale = cv.take<float>((size_t)n * 3);
    float* prob = cv.take<float>((size_t)n);
    int32_t* cls = cv.take<int32_t>((size_t)n);
    int32_t* d_flag = cv.take<int32_t>(64);
    const size_t fixed_end = cv.off;
    if (!mm_ws || !ibuf || !kbuf || !sort_ws || !lev_ws || !xyz || !scale || !prob || !cls || !d_flag) {
        // the part that depends on the level sizes is not known yet: ask for the fixed part plus a generous guess
        r->needed_ws_bytes = fixed_end * 2 + ((size_t)256 << 20);
        CV_REQUIRE(false, CV_ENOMEM, "scene workspace too small (needs at least %zu bytes)", r->needed_ws_bytes);
    }
    // (the one-workgroup final launch of the bounds also zeroes the range flag and fills the eight bound words at the head of the
    // sort's workspace: two fill launches fewer)
    int rc = cv_hv_minmax_async_ex(d->d_points, n, h_minmax, mm_ws, cv_hv_minmax_workspace_bytes(), d_flag,
                                   reinterpret_cast<int32_t*>(sort_ws), stream);
    if (rc != CV_OK) return rc;

    int32_t* c_coords[NL];
    unsigned long long* c_keys[NL];
    int32_t* c_vals[NL];
    for (int i = 0; i < NL; ++i) {
        c_coords[i] = ibuf + o_coords[i];
        c_keys[i] = kbuf + (size_t)cap * i;
        c_vals[i] = ibuf + o_vals[i];
    }
    int32_t counts_h[8] = {0};
    cv_scene_maps off;
    rc = cv_sp_scene_plan_ex(d->d_coords4, n, ibuf + o_perm, ibuf + o_inv, c_coords, c_keys, c_vals, cap, ibuf + o_counts, counts_h,
                          d->stem_k, d->mask_groups, d->masked_min_rows, ibuf + o_arena, words, &off, sort_ws, sws_b, lev_ws, lws_b, /* one scene: the sort skips its batch digit */ true,
                          stream, /* bound words filled above */ true);
    if (rc != CV_OK) return rc;
    r->duplicates = counts_h[5];
    r->out_of_window = counts_h[6];
    CV_REQUIRE(counts_h[5] == 0 && counts_h[6] == 0, CV_EINVAL,
               "duplicate coordinates (%d) or coordinates outside the 16-bit key window (%d)", counts_h[5], counts_h[6]);
    lap(0);
    long long rows[NL];
    for (int i = 0; i < NL; ++i) { rows[i] = counts_h[i]; r->level_rows[i] = counts_h[i]; }
    // (the plan waited for an event recorded behind the bounds reduction: the pinned bounds are valid)
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) { mn[k] = h_minmax[k]; mx[k] = h_minmax[3 + k]; r->corner[k] = mn[k]; }
    int dims[3];
    rc = cv_hv_grid_dims_f32(mn, mx, d->res, dims);
    if (rc != CV_OK) return rc;
    for (int k = 0; k < 3; ++k) r->dims[k] = dims[k];
    const size_t cells = (size_t)dims[0] * dims[1] * dims[2];

    // ---- what depends on the level sizes and the grid shape
    const size_t arena_b = cv_net_arena_bytes(d->bufs, d->n_bufs, rows, NL);
    size_t conv_ws_b = 0;
    for (int i = 0; i < NL; ++i) {
        if (off.mask_perm[i] >= 0) conv_ws_b = std::max(conv_ws_b, (size_t)4 * d->mask_groups * (size_t)rows[i] * d->max_channels + 256);
        conv_ws_b = std::max(conv_ws_b, cv_sp_conv_workspace_bytes(std::min<long long>(rows[i], 128 * 384 - 1), d->max_channels, 27));
    }
    conv_ws_b += 16384 + 256;                                            // + the split-K tickets the executor keeps at the tail
    const size_t vote_ws_b = cv_hv_forward_workspace_bytes(n, d->num_rots, dims, d->vote_algo);
    const size_t dec_ws_b = cv_decode_workspace_bytes(dims, n, d->max_candidates);
    char* arena = cv.take<char>(arena_b);
    char* conv_ws = cv.take<char>(conv_ws_b);
    char* vote_ws = cv.take<char>(std::max<size_t>(vote_ws_b, 256));
    char* dec_ws = cv.take<char>(dec_ws_b);
    float* grids = d->d_grids ? d->d_grids : cv.take<float>(6 * cells);
    r->needed_ws_bytes = cv.off + 4096;
    r->needed_grid_floats = 6 * cells;
    CV_REQUIRE(arena && conv_ws && vote_ws && dec_ws && grids, CV_ENOMEM, "scene workspace too small (needs %zu bytes)", r->needed_ws_bytes);
    CV_REQUIRE(!d->d_grids || d->grid_capacity_floats >= 6 * cells, CV_ENOMEM, "grid buffer too small (needs %zu floats)", 6 * cells);
    float* g_obj = grids;
    float* g_rot = grids + cells;
    float* g_scale = grids + 3 * cells;
    r->d_grid_obj = g_obj; r->d_grid_rot = g_rot; r->d_grid_scale = g_scale;

    // ---- network forward: the program's map / order slots are [stem, down 0-3, k3 0-4, up 0-3, out] and
    //      [mask orders of levels 0-4 (NULL below masked_min_rows), octant orders of the four transposed convs]
    const int32_t* ap = ibuf + o_arena;
    const int32_t* maps[15];
    maps[0] = ap + off.stem;
    for (int i = 0; i < 4; ++i) maps[1 + i] = ap + off.down[i];
    for (int i = 0; i < 5; ++i) maps[5 + i] = ap + off.k3[i];
    for (int i = 0; i < 4; ++i) maps[10 + i] = ap + off.up[i];
    maps[14] = ibuf + o_inv;
    const int32_t* perms[9];
    for (int i = 0; i < 5; ++i) perms[i] = (off.mask_perm[i] >= 0 && rows[i] >= d->masked_min_rows) ? ap + off.mask_perm[i] : nullptr;
    for (int i = 0; i < 4; ++i) perms[5 + i] = ap + off.up_perm[i];
    std::unique_lock<std::mutex> enq(g_enqueue_mu, std::defer_lock);
    if (serialize_enqueue()) enq.lock();
    const void* ext_ptr[2] = {d->d_feats, d->d_out_feats};
    const int ext_ld[2] = {d->feats_ld, d->out_ld};
    rc = cv_net_run_f32(d->ops, d->n_ops, d->bufs, d->n_bufs, rows, NL, arena, arena_b, ext_ptr, ext_ld, maps, 15, perms, 9, conv_ws,
                        conv_ws_b, d->use_range_flag ? d_flag : nullptr, stream);
    if (rc != CV_OK) return rc;
    lap(1);
    CV_HIP_CHECK(mark(1));
    rc = cv_head_joint_f32(d->d_out_feats, n, d->out_ld, d->nclasses, d->log_scale, xyz, scale, prob, cls, stream);
    if (rc != CV_OK) return rc;
    if (d->use_range_flag) CV_HIP_CHECK(hipMemcpyAsync(h_flag, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    r->d_xyz = xyz; r->d_scale = scale; r->d_prob = prob; r->d_class = cls;
    CV_HIP_CHECK(mark(2));

    // ---- vote + decode, on the network's predictions or on the caller's (bench.py --predictions teacher)
    const float* v_xyz = d->d_xyz_in ? d->d_xyz_in : xyz;
    const float* v_scale = d->d_scale_in ? d->d_scale_in : scale;
    const float* v_prob = d->d_prob_in ? d->d_prob_in : prob;
    const int32_t* v_cls = d->d_class_in ? d->d_class_in : cls;
    rc = cv_hv_forward_f32(d->d_points, v_xyz, v_scale, v_prob, n, d->res, d->num_rots, mn, dims, g_obj, g_rot, g_scale, vote_ws,
                           std::max<size_t>(vote_ws_b, 256), d->vote_algo, stream);
    if (rc != CV_OK) return rc;
    CV_HIP_CHECK(mark(3));
    lap(2);
    if (enq.owns_lock()) enq.unlock();
    cv_decode_params prm = d->decode;
    prm.max_iters = d->max_candidates;
    int n_cand = 0, n_boxes = 0, truncated = 0;
    // (events[4] is recorded behind the decode's last launch, in front of its host wait: a device time)
    rc = cv_decode_f32_ev(g_obj, g_rot, g_scale, dims, mn, d->res, d->d_points, v_xyz, v_prob, v_cls, n, &prm, 0, dec_ws, dec_ws_b, &n_cand,
                          d->h_cand_idx, d->h_verdict, &n_boxes, d->h_boxes, d->h_scores, d->h_classes, &truncated, stream, d->events[4]);
    if (rc != CV_OK) return rc;
    r->n_cand = n_cand;
    r->n_boxes = n_boxes;
    r->truncated = truncated;
    // (the decode waited for the stream: the range flag of the network's convolutions has landed)
    r->range_flag = d->use_range_flag ? *h_flag : 0;

    // ---- per-class NMS (eval_joint.py:270-280): kept boxes as indices into the decode's box list, class by class
    int n_det = 0;
    std::vector<float> bc, sc;
    std::vector<int32_t> idx, pick;
    for (int c = 0; c < d->nclasses; ++c) {
        bc.clear(); sc.clear(); idx.clear();
        for (int i = 0; i < n_boxes; ++i)
            if (d->h_classes[i] == c) {
                bc.insert(bc.end(), d->h_boxes + (size_t)i * 24, d->h_boxes + (size_t)i * 24 + 24);
                sc.push_back(d->h_scores[i]);
                idx.push_back(i);
            }
        if (idx.empty()) continue;
        pick.assign(idx.size(), 0);
        const int k = cv_nms_obb(bc.data(), sc.data(), (int)idx.size(), d->nms_threshold, pick.data());
        if (k < 0) return k;
        for (int j = 0; j < k; ++j) d->h_pick[n_det++] = idx[(size_t)pick[j]];
    }
    r->n_det = n_det;
    lap(3);
    return CV_OK;
}

}  // extern "C"
