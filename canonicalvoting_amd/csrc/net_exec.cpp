// Host-side executor for a fused eval-mode sparse network (the 63 convolutions of MinkUNet34C.forward,
// utils/minkunet.py:122-180, with BatchNorm / bias / residual / ReLU folded into the conv epilogues).
// The program is symbolic - feature buffers are (coordinate level, channels) slots of a per-scene arena, kernel
// maps and processing orders are slots of per-scene pointer tables - so it is built once per model and one
// C call per scene issues every launch: the Python layer's per-conv overhead (descriptor marshalling, tensor
// allocation, ~35 us x 63) is what bounded the scene rate once several scenes were in flight.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "cv_common.h"

namespace {

struct Slot { char* ptr; int ld; long long rows; };

}  // namespace

extern "C" {

// ---- every kernel map and processing order of the fused network in ONE call per scene ----------------------
// (the Python coordinate manager issued ~25 calls for them: 0.65 ms of host time per scene on the critical path)
static void scene_maps_layout(const long long* rows, long long n_orig, int stem_k, int mask_groups,
                              long long masked_min_rows, cv_scene_maps* o, size_t* total) {
    size_t off = 0;
    auto take = [&](size_t words) { const size_t at = off; off += cv_align_up(words, 64); return (long long)at; };
    const size_t K5 = (size_t)stem_k * stem_k * stem_k;
    // orders [groups][rows] + map rows in processing order [groups][rows][W] + validity bytes [groups][rows] (1/4 word each)
    const size_t mp_w = mask_groups > 1 ? (size_t)mask_groups * (1 + (27 + mask_groups - 1) / mask_groups) + (mask_groups + 3) / 4 : 0;
    // what depends on the caller's row count only comes first: cv_sp_scene_plan builds it before the coarse counts are known
    o->stem = take((size_t)rows[0] * K5);
    o->k3[0] = take((size_t)rows[0] * 27);
    o->mask_perm[0] = (mask_groups > 1 && rows[0] >= masked_min_rows) ? take(mp_w * rows[0]) : -1;
    // bin counts + running counts of the mask orders (cv_sp_mask_perms_batch: 2 x 1024 words per group): level 0, then <= 4
    // coarse levels and the 4 up-map orders
    o->scratch = take((size_t)(5 * std::max(mask_groups, 1) + 4) * 2048);
    o->bitmap = take((size_t)CV_BITMAP_WORDS);          // occupancy bits of the level-0 set (cv_sp_scene_plan only)
    o->out = -1;                       // (the sort's inverse permutation is the final map)
    (void)n_orig;
    for (int i = 0; i < 4; ++i) o->down[i] = take((size_t)rows[i + 1] * 8);
    for (int i = 1; i < 5; ++i) o->k3[i] = take((size_t)rows[i] * 27);
    for (int i = 0; i < 4; ++i) o->up[i] = take((size_t)rows[3 - i] * 8);            // up[i]: level 4-i -> 3-i
    for (int i = 1; i < 5; ++i) o->mask_perm[i] = (mask_groups > 1 && rows[i] >= masked_min_rows) ? take(mp_w * rows[i]) : -1;
    for (int i = 0; i < 4; ++i) o->up_perm[i] = take((size_t)rows[3 - i]);
    *total = off;
}

size_t cv_sp_scene_maps_words(const long long* level_rows, long long n_orig, int stem_k, int mask_groups,
                              long long masked_min_rows, cv_scene_maps* offsets) {
    if (!level_rows || !offsets || n_orig <= 0 || stem_k < 1) return 0;
    size_t total = 0;
    scene_maps_layout(level_rows, n_orig, stem_k, mask_groups, masked_min_rows, offsets, &total);
    return total;
}

// level-0 part (needs the caller's row count only): the stem map with the sort permutation folded in, the 3x3x3 map of
// the finest level and its mask-sorted orders
static int scene_maps_level0(int32_t* const* d_coords, const unsigned long long* const* d_keys, const int32_t* const* d_vals,
                             long long cap, long long n, const int32_t* d_perm, int stem_k, int mask_groups,
                             const cv_scene_maps& o, int32_t* d_arena, const int32_t* d_bbox, void* stream,
                             bool pre_cleared = false, bool bitmap_filled = false) {
    const unsigned* bits = nullptr;
    if (d_bbox) {       // occupancy bitmap in front of the hash probes (bounds from the sort; cv_sp_scene_plan)
        bits = reinterpret_cast<const unsigned*>(d_arena + o.bitmap);
        if (!bitmap_filled) {          // (cv_sp_scene_plan: the level build's insert pass set the bits already)
            const int rc0 = cv_sp_occupancy_bitmap(d_coords[0], n, d_bbox, reinterpret_cast<unsigned*>(d_arena + o.bitmap), stream,
                                                   pre_cleared);
            if (rc0 != CV_OK) return rc0;
        }
    }
    CvMapJob mj[2];
    // stem: sorted rows <- rows of the ORIGINAL order = the sorted set's own map with the permutation folded in
    // (the caller's set needs no hash table of its own); the final original <- sorted map is the sort's inverse
    mj[0] = {d_coords[0], n, d_keys[0], d_vals[0], cap, stem_k, 1, d_arena + o.stem, d_perm, bits, d_bbox, 0};
    mj[1] = {d_coords[0], n, d_keys[0], d_vals[0], cap, 3, 1, d_arena + o.k3[0], nullptr, bits, d_bbox, 0};
    int rc = cv_sp_kernel_maps_batch(mj, 2, stream);
    if (rc != CV_OK) return rc;
    if (o.mask_perm[0] >= 0) {
        CvPermJob pj = {d_arena + o.k3[0], n, 27, mask_groups, d_arena + o.mask_perm[0], 1};
        rc = cv_sp_mask_perms_batch(&pj, 1, d_arena + o.scratch, sizeof(int) * (size_t)mask_groups * 2048, stream, pre_cleared);
        if (rc != CV_OK) return rc;
    }
    return CV_OK;
}

// the coarse levels: ONE launch for the eight remaining kernel maps and the four transposed maps (by lookup of the
// parent voxel: no pre-fill, no dependency on the strided maps), three launches for the remaining processing orders
static int scene_maps_coarse(int32_t* const* d_coords, const unsigned long long* const* d_keys, const int32_t* const* d_vals,
                             long long cap, const long long* level_rows, int mask_groups, const cv_scene_maps& o,
                             int32_t* d_arena, void* stream, bool pre_zeroed = false) {
    CvMapJob mj[CV_MAX_MAP_JOBS];
    int nm = 0;
    for (int i = 0; i < 4; ++i)
        mj[nm++] = {d_coords[i + 1], level_rows[i + 1], d_keys[i], d_vals[i], cap, 2, 1 << i, d_arena + o.down[i], nullptr, nullptr, nullptr, 0};
    for (int i = 1; i < 5; ++i)
        mj[nm++] = {d_coords[i], level_rows[i], d_keys[i], d_vals[i], cap, 3, 1 << i, d_arena + o.k3[i], nullptr, nullptr, nullptr, 0};
    for (int i = 0; i < 4; ++i) {       // up[i]: level 4 - i -> 3 - i; the fine rows look their parent up in the coarser table
        const int fine = 3 - i;
        mj[nm++] = {d_coords[fine], level_rows[fine], d_keys[fine + 1], d_vals[fine + 1], cap, 2, 1 << fine, d_arena + o.up[i],
                    nullptr, nullptr, nullptr, 1};
    }
    int rc = cv_sp_kernel_maps_batch(mj, nm, stream);
    if (rc != CV_OK) return rc;
    CvPermJob pj[CV_MAX_PERM_JOBS];
    int np = 0, groups = 0;
    for (int i = 1; i < 5; ++i)
        if (o.mask_perm[i] >= 0) {
            pj[np++] = {d_arena + o.k3[i], level_rows[i], 27, mask_groups, d_arena + o.mask_perm[i], 1};
            groups += mask_groups;
        }
    for (int i = 0; i < 4; ++i) {
        pj[np++] = {d_arena + o.up[i], level_rows[3 - i], 8, 1, d_arena + o.up_perm[i], 0};
        groups += 1;
    }
    return cv_sp_mask_perms_batch(pj, np, d_arena + o.scratch + (size_t)std::max(mask_groups, 1) * 2048,
                                  sizeof(int) * (size_t)groups * 2048, stream, pre_zeroed);
}

int cv_sp_scene_maps(int32_t* const* d_coords, const unsigned long long* const* d_keys, const int32_t* const* d_vals,
                     long long cap, const long long* level_rows, const int32_t* d_perm, long long n_orig, int stem_k,
                     int mask_groups, long long masked_min_rows, int32_t* d_arena, size_t arena_words, void* stream) {
    CV_REQUIRE(d_coords && d_keys && d_vals && level_rows && d_perm && d_arena, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n_orig == level_rows[0], CV_EINVAL, "the sorted set must hold the caller's %lld rows", n_orig);
    cv_scene_maps o;
    size_t total = 0;
    scene_maps_layout(level_rows, n_orig, stem_k, mask_groups, masked_min_rows, &o, &total);
    CV_REQUIRE(arena_words >= total, CV_ENOMEM, "scene map arena too small");
    int rc = scene_maps_level0(d_coords, d_keys, d_vals, cap, n_orig, d_perm, stem_k, mask_groups, o, d_arena, nullptr, stream);
    if (rc != CV_OK) return rc;
    return scene_maps_coarse(d_coords, d_keys, d_vals, cap, level_rows, mask_groups, o, d_arena, stream);
}

// ---- the whole coordinate plan of a scene in ONE call: spatial row sort, the five coordinate levels with their hash
// tables, every kernel map and processing order.  The level counts have to reach the host (grid sizes, arena layout):
// they are copied into pinned memory as soon as the levels exist and the host waits for THAT copy (an event), while the
// stream already builds the level-0 maps queued behind it (the 5x5x5 stem map is 3/4 of all lookups) - the host wait,
// which used to leave the GPU idle for ~80 us between cv_sp_build_levels and cv_sp_scene_maps, now overlaps with them.
namespace {
// event + 64-byte pinned landing buffer of one cv_sp_scene_plan call, taken from a per-device pool for the duration of
// the call (a slot keyed by the stream handle was shared by every thread / device that planned on the null stream)
struct PlanSide { hipEvent_t ev; int32_t* h_pinned; int device; };
std::mutex g_plan_mu;
std::unordered_map<int, std::vector<PlanSide>> g_plan_free;
int plan_side_acquire(PlanSide* out) {
    int dev = 0;
    CV_HIP_CHECK(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        auto& pool = g_plan_free[dev];
        if (!pool.empty()) {
            *out = pool.back();
            pool.pop_back();
            return CV_OK;
        }
    }
    PlanSide p{};
    p.device = dev;
    CV_HIP_CHECK(hipEventCreateWithFlags(&p.ev, hipEventDisableTiming));
    CV_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&p.h_pinned), 64, hipHostMallocDefault));
    *out = p;
    return CV_OK;
}
struct PlanSideLease {
    PlanSide ps{};
    bool held = false;
    ~PlanSideLease() {
        if (!held) return;
        (void)hipEventSynchronize(ps.ev);       // an early error return must not hand the buffer on with its copy in flight
        std::lock_guard<std::mutex> lk(g_plan_mu);
        g_plan_free[ps.device].push_back(ps);
    }
};
}  // namespace

size_t cv_sp_scene_plan_words(long long n, int stem_k, int mask_groups, long long masked_min_rows) {
    if (n <= 0 || stem_k < 1) return 0;
    const long long rows[5] = {n, n, n, n, n};          // a coarser level never has more rows than a finer one
    cv_scene_maps o;
    size_t total = 0;
    scene_maps_layout(rows, n, stem_k, mask_groups, masked_min_rows, &o, &total);
    return total;
}

int cv_sp_scene_plan(const int32_t* d_input, long long n, int32_t* d_perm, int32_t* d_inv, int32_t* const* d_coords,
                     unsigned long long* const* d_keys, int32_t* const* d_vals, long long cap, int32_t* d_counts,
                     int32_t* h_counts, int stem_k, int mask_groups, long long masked_min_rows, int32_t* d_arena,
                     size_t arena_words, cv_scene_maps* offsets, void* d_sort_ws, size_t sort_ws_bytes, void* d_levels_ws,
                     size_t levels_ws_bytes, void* stream) {
    return cv_sp_scene_plan_ex(d_input, n, d_perm, d_inv, d_coords, d_keys, d_vals, cap, d_counts, h_counts, stem_k, mask_groups,
                               masked_min_rows, d_arena, arena_words, offsets, d_sort_ws, sort_ws_bytes, d_levels_ws,
                               levels_ws_bytes, false, stream);
}

}  // extern "C"

// (C++ linkage, cv_common.h) single_batch: the rows are one scene (cv_detect_scene_f32): the sort skips its batch digit
int cv_sp_scene_plan_ex(const int32_t* d_input, long long n, int32_t* d_perm, int32_t* d_inv, int32_t* const* d_coords,
                        unsigned long long* const* d_keys, int32_t* const* d_vals, long long cap, int32_t* d_counts,
                        int32_t* h_counts, int stem_k, int mask_groups, long long masked_min_rows,
                        int32_t* d_arena, size_t arena_words, cv_scene_maps* offsets, void* d_sort_ws, size_t sort_ws_bytes,
                        void* d_levels_ws, size_t levels_ws_bytes, bool single_batch, void* stream, bool bounds_prefilled) {
    CV_REQUIRE(d_input && d_perm && d_inv && d_coords && d_keys && d_vals && d_counts && h_counts && d_arena && offsets &&
                   d_sort_ws && d_levels_ws, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(arena_words >= cv_sp_scene_plan_words(n, stem_k, mask_groups, masked_min_rows), CV_ENOMEM,
               "scene map arena too small (cv_sp_scene_plan_words)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = cv_sp_sort_rows_ex(d_input, n, d_coords[0], d_perm, d_inv, d_sort_ws, sort_ws_bytes, single_batch, stream, bounds_prefilled);
    if (rc != CV_OK) return rc;
    // level-0 layout of the arena (depends on n only): the histogram scratch of the mask orders and the occupancy bitmap
    // are neighbours there and are zeroed by the first launch of the level build (no fill launches of their own)
    long long rows[5] = {n, 1, 1, 1, 1};
    cv_scene_maps o;
    size_t total = 0;
    scene_maps_layout(rows, n, stem_k, mask_groups, masked_min_rows, &o, &total);
    static const bool bitmap_on = !(getenv("CV_MAP_BITMAP") && atoi(getenv("CV_MAP_BITMAP")) == 0);
    const long long n_zero = o.bitmap + (bitmap_on ? CV_BITMAP_WORDS : 0) - o.scratch;
    // (the sort leaves its bounds - min, -max per axis, -max batch - in the first 8 ints of its workspace; [7] = 1 marks
    // the bitmap as trusted until bitmap_set finds a row outside them)
    rc = cv_sp_build_levels_zero(d_coords, d_keys, d_vals, n, cap, 5, d_counts, nullptr, d_levels_ws, levels_ws_bytes,
                                 d_arena + o.scratch, n_zero, bitmap_on ? static_cast<int32_t*>(d_sort_ws) + 7 : nullptr, stream,
                                 bitmap_on ? reinterpret_cast<uint32_t*>(d_arena + o.bitmap) : nullptr);
    if (rc != CV_OK) return rc;
    PlanSideLease lease;
    rc = plan_side_acquire(&lease.ps);
    if (rc != CV_OK) return rc;
    lease.held = true;
    const PlanSide& ps = lease.ps;
    CV_HIP_CHECK(hipMemcpyAsync(ps.h_pinned, d_counts, sizeof(int32_t) * 8, hipMemcpyDeviceToHost, st));
    CV_HIP_CHECK(hipEventRecord(ps.ev, st));
    // level-0 maps: their arena offsets depend on n only
    rc = scene_maps_level0(d_coords, d_keys, d_vals, cap, n, d_perm, stem_k, mask_groups, o, d_arena,
                           bitmap_on ? static_cast<const int32_t*>(d_sort_ws) : nullptr, stream, true, bitmap_on);
    if (rc != CV_OK) return rc;
    CV_HIP_CHECK(hipEventSynchronize(ps.ev));          // the counts have landed; the level-0 maps are still being built
    for (int i = 0; i < 8; ++i) h_counts[i] = ps.h_pinned[i];
    if (h_counts[5] != 0 || h_counts[6] != 0) return CV_OK;      // duplicates / out-of-window rows: the caller reports them
    for (int i = 0; i < 5; ++i) rows[i] = h_counts[i];
    CV_REQUIRE(rows[0] == n, CV_EINVAL, "level 0 lost rows (%lld of %lld)", rows[0], n);
    scene_maps_layout(rows, n, stem_k, mask_groups, masked_min_rows, &o, &total);
    CV_REQUIRE(arena_words >= total, CV_ENOMEM, "scene map arena too small");
    *offsets = o;
    return scene_maps_coarse(d_coords, d_keys, d_vals, cap, rows, mask_groups, o, d_arena, stream, true);
}

extern "C" {

size_t cv_net_arena_bytes(const cv_net_buf* bufs, int n_bufs, const long long* level_rows, int n_levels) {
    if (!bufs || !level_rows || n_bufs <= 0) return 0;
    size_t total = 256;
    for (int i = 0; i < n_bufs; ++i) {
        if (bufs[i].level < 0 || bufs[i].level >= n_levels) continue;        // external buffer
        total += cv_align_up(sizeof(float) * (size_t)level_rows[bufs[i].level] * (size_t)bufs[i].channels, 256);
    }
    return total;
}

int cv_net_run_f32(const cv_net_op* ops, int n_ops, const cv_net_buf* bufs, int n_bufs, const long long* level_rows,
                   int n_levels, void* d_arena, size_t arena_bytes, const void* const* ext_ptr, const int* ext_ld,
                   const int32_t* const* maps, int n_maps, const int32_t* const* perms, int n_perms,
                   void* d_ws, size_t ws_bytes, int32_t* range_flag, void* stream) {
    CV_REQUIRE(ops && bufs && level_rows && d_arena && n_ops > 0 && n_bufs > 0 && n_levels > 0, CV_EINVAL,
               "bad network program arguments");
    CV_REQUIRE(arena_bytes >= cv_net_arena_bytes(bufs, n_bufs, level_rows, n_levels), CV_ENOMEM, "arena too small");
    std::vector<Slot> slot((size_t)n_bufs);
    {
        CvCarver cv(d_arena);
        int ext = 0;
        for (int i = 0; i < n_bufs; ++i) {
            const int lv = bufs[i].level;
            if (lv < 0 || lv >= n_levels) {                                   // external: caller's tensor
                CV_REQUIRE(ext_ptr && ext_ld && ext_ptr[ext], CV_EINVAL, "external buffer %d has no pointer", ext);
                slot[i] = {static_cast<char*>(const_cast<void*>(ext_ptr[ext])), ext_ld[ext], bufs[i].rows_level >= 0 &&
                           bufs[i].rows_level < n_levels ? level_rows[bufs[i].rows_level] : 0};
                ++ext;
            } else {
                slot[i] = {reinterpret_cast<char*>(cv.take<float>((size_t)level_rows[lv] * bufs[i].channels)),
                           bufs[i].channels, level_rows[lv]};
            }
        }
    }
    // split-K tickets of the hl-format convolutions (cv_conv_desc.split_tickets): the last 16 KB of the workspace,
    // zeroed once per forward (the launches leave them at zero; the fill keeps a failed forward from poisoning the next)
    int32_t* tickets = nullptr;
    size_t conv_ws_bytes = ws_bytes;
    const size_t ticket_bytes = sizeof(int32_t) * CV_SPLIT_TICKETS;
    // OFF unless CV_HL_FUSE_FINISH=1.  Round 2 published with plain stores + an agent-scope release per workgroup: 2.50 -> 3.79 ms
    // per forward.  Round 4 publishes write-through (sc1 stores, no release fence): the publish is cheap now, but the LAST
    // ARRIVER reads splits x 32 KB alone (~65 GB/s per workgroup): one scene in flight 2.43 -> 3.21 ms per forward (16-32
    // splits), eight in flight 555 -> 545 scenes/s (2-8 splits) - profiles/r4/throughput_ablations.txt.  A finish launch
    // spreads the same reads over the chip.
    static const bool fuse_on = getenv("CV_HL_FUSE_FINISH") && atoi(getenv("CV_HL_FUSE_FINISH")) != 0;
    if (fuse_on && d_ws && ws_bytes > ((size_t)1 << 20) + ticket_bytes) {
        conv_ws_bytes = (ws_bytes - ticket_bytes) & ~(size_t)255;
        tickets = reinterpret_cast<int32_t*>(static_cast<char*>(d_ws) + conv_ws_bytes);
        CV_HIP_CHECK(hipMemsetAsync(tickets, 0, ticket_bytes, static_cast<hipStream_t>(stream)));
    }
    for (int k = 0; k < n_ops; ++k) {
        const cv_net_op& o = ops[k];
        CV_REQUIRE(o.in_buf >= 0 && o.in_buf < n_bufs && o.out_buf >= 0 && o.out_buf < n_bufs &&
                       o.res_buf < n_bufs && o.map < n_maps && o.perm < n_perms, CV_EINVAL, "op %d: bad slot", k);
        const Slot& in = slot[o.in_buf];
        const Slot& out = slot[o.out_buf];
        cv_conv_desc d = {};
        d.in_hl = bufs[o.in_buf].hl;
        d.out_hl = bufs[o.out_buf].hl;
        d.res_hl = o.res_buf >= 0 ? bufs[o.res_buf].hl : 0;
        CV_REQUIRE(o.in2_buf < 0 || bufs[o.in2_buf].hl == d.in_hl, CV_EINVAL, "op %d: both sources must share a format", k);
        d.in = reinterpret_cast<const float*>(in.ptr) + o.in_col;
        d.n_in = in.rows;
        d.in_ld = in.ld;
        d.cin = o.cin;
        d.weight = o.weight;
        d.weight_x6 = o.weight_x6;
        d.weight_pieces = o.weight_pieces;
        d.acc_scale = o.acc_scale;
        d.range_flag = (o.weight_pieces == 2 || d.out_hl) ? range_flag : nullptr;
        if (o.in2_buf >= 0) {
            CV_REQUIRE(o.in2_buf < n_bufs, CV_EINVAL, "op %d: bad second-source slot", k);
            d.in2 = reinterpret_cast<const float*>(slot[o.in2_buf].ptr) + o.in2_col;
            d.in2_ld = slot[o.in2_buf].ld;
            d.cin2 = o.cin2;
            d.weight2_x6 = o.weight2_x6;
        }
        d.K = o.K;
        d.cout = o.cout;
        d.nbr = o.map >= 0 ? maps[o.map] : nullptr;
        d.n_out = out.rows;
        d.scale = o.scale;
        d.shift = o.shift;
        if (o.res_buf >= 0) {
            d.residual = reinterpret_cast<const float*>(slot[o.res_buf].ptr) + o.res_col;
            d.res_ld = slot[o.res_buf].ld;
        }
        d.relu = o.relu;
        d.out = reinterpret_cast<float*>(out.ptr) + o.out_col;
        d.out_ld = out.ld;
        d.ws = d_ws;
        d.ws_bytes = conv_ws_bytes;
        const int32_t* perm = o.perm >= 0 ? perms[o.perm] : nullptr;
        d.split_tickets = d.in_hl ? tickets : nullptr;
        if (perm) {
            d.row_perm = perm;
            d.perm_groups = o.perm_groups;
            d.perm_has_map = o.perm_groups > 1;      /* scene maps build the orders with their map rows */
        }
        const int rc = cv_sp_conv_f32(&d, stream);
        if (rc != CV_OK) return rc;
    }
    return CV_OK;
}

}  // extern "C"
