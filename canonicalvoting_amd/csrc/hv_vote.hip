// Canonical vote accumulation for gfx950 (MI355X).
//
// Replaces houghvoting/src/hv_cuda_kernel.cu (forward :12-97, average :100-119,
// backward :168-261, host shape logic :121-165) behind the C ABI of include/cv_hip.h.
//
// Compiled with -ffp-contract=off: the per-vote geometry (offset, grid position,
// bounds test, floor, trilinear weights) is the strict fp32 operation sequence of
// the reference source, so cell indices and per-vote contributions are bit-identical
// to oracle/hv_oracle.c; only the fp32 summation order differs (as it does run to
// run in the reference, whose accumulation is atomicAdd too).
//
// Two forward algorithms:
//   direct : one lane per (point, rotation), 48 hardware fp32 global atomics per
//            in-bounds vote, hipMemsetAsync before and a normalise pass after.
//            This is the reference's algorithm re-parallelised over N*R lanes
//            instead of N threads (79 blocks at N=80k cannot fill 256 CUs).
//   tiles  : on-chip accumulation.  Every vote of a point has the same y, so points
//            are counting-sorted by their vote's y cell (80k keys, not 9.6M).  One
//            workgroup owns a 32x32 (x,z) tile of ONE y plane in LDS (6 channels,
//            24 KB), pulls the points of y-bins {y-1, y}, culls them with a
//            ring-vs-tile test, wave-compacts survivors (ballot + mbcnt), expands
//            survivors x rotations densely over lanes, wave-compacts the votes that
//            land in the tile, and drains them with LDS float atomics.  The tile is
//            then normalised and stored once: no memset, no global atomics, no
//            second pass over the 63 MB grid.
#include "cv_common.h"

#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

struct F3 { float x, y, z; };
struct I3 { int x, y, z; };

// ---------------------------------------------------------------------------
// rotation table: (cos, sin)(i * (2*3.141592654f / R)), theta in fp32 exactly as
// hv_cuda_kernel.cu:35,37, cos/sin correctly rounded (see oracle/hv_oracle.c).
// Cached per (device, R); built once on the host.
// ---------------------------------------------------------------------------
std::mutex g_tab_mu;
std::unordered_map<uint64_t, float2*> g_tabs;

int get_rot_table(int num_rots, const float2** out) {
    int dev = 0;
    CV_HIP_CHECK(hipGetDevice(&dev));
    const uint64_t key = (uint64_t(uint32_t(dev)) << 32) | uint32_t(num_rots);
    std::lock_guard<std::mutex> lk(g_tab_mu);
    auto it = g_tabs.find(key);
    if (it != g_tabs.end()) { *out = it->second; return CV_OK; }
    std::vector<float2> h(num_rots);
    const float rot_interval = 2 * 3.141592654f / num_rots;
    for (int i = 0; i < num_rots; ++i) {
        const float theta = i * rot_interval;
        h[i].x = (float)std::cos((double)theta);
        h[i].y = (float)std::sin((double)theta);
    }
    float2* d = nullptr;
    CV_HIP_CHECK(hipMalloc(&d, sizeof(float2) * num_rots));
    CV_HIP_CHECK(hipMemcpy(d, h.data(), sizeof(float2) * num_rots, hipMemcpyHostToDevice));
    g_tabs[key] = d;
    *out = d;
    return CV_OK;
}

// ---------------------------------------------------------------------------
// shared per-vote geometry (hv_cuda_kernel.cu:38-47)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float grid_pos(float p, float off, float corner, float res) {
    return ((p + off) - corner) / res;
}

// ---------------------------------------------------------------------------
// min / max of points (hv_cuda_kernel.cu:129)
// ---------------------------------------------------------------------------
constexpr int MM_BLOCKS = 256;

__global__ __launch_bounds__(256) void minmax_partial(const float* __restrict__ pts, int64_t n,
                                                      float* __restrict__ part) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = pts[i * 3 + k];
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
        }
    }
    __shared__ float s[6][256];
#pragma unroll
    for (int k = 0; k < 3; ++k) { s[k][threadIdx.x] = mn[k]; s[3 + k][threadIdx.x] = mx[k]; }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s[k][threadIdx.x] = fminf(s[k][threadIdx.x], s[k][threadIdx.x + st]);
                s[3 + k][threadIdx.x] = fmaxf(s[3 + k][threadIdx.x], s[3 + k][threadIdx.x + st]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = s[threadIdx.x][0];
}

// zero_word / fill7f: words of the scene call's next stages initialised by this one-workgroup launch (the range flag := 0, the
// eight bound words of the row sort := 0x7f7f7f7f) instead of two fill launches of their own
__global__ __launch_bounds__(256) void minmax_final(const float* __restrict__ part, int nblocks,
                                                    float* __restrict__ out, int* __restrict__ zero_word = nullptr,
                                                    int* __restrict__ fill7f = nullptr) {
    if (zero_word && threadIdx.x == 32) *zero_word = 0;
    if (fill7f && threadIdx.x >= 64 && threadIdx.x < 72) fill7f[threadIdx.x - 64] = 0x7f7f7f7f;
    __shared__ float s[6][256];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = threadIdx.x; b < nblocks; b += 256)
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(mn[k], part[b * 6 + k]);
            mx[k] = fmaxf(mx[k], part[b * 6 + 3 + k]);
        }
    for (int k = 0; k < 3; ++k) { s[k][threadIdx.x] = mn[k]; s[3 + k][threadIdx.x] = mx[k]; }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st)
            for (int k = 0; k < 3; ++k) {
                s[k][threadIdx.x] = fminf(s[k][threadIdx.x], s[k][threadIdx.x + st]);
                s[3 + k][threadIdx.x] = fmaxf(s[3 + k][threadIdx.x], s[3 + k][threadIdx.x + st]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) out[threadIdx.x] = s[threadIdx.x][0];
}

// ---------------------------------------------------------------------------
// direct algorithm
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hv_fwd_direct(
    const float* __restrict__ pts, const float* __restrict__ xyz, const float* __restrict__ scl,
    const float* __restrict__ obj, int64_t n, int R, float res, F3 corner, I3 dims,
    const float2* __restrict__ tab, float* __restrict__ g_obj, float* __restrict__ g_rot,
    float* __restrict__ g_scale) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n * R) return;
    const int64_t c = t / R;
    const int i = (int)(t - c * R);
    const float s0 = scl[c * 3 + 0], s1 = scl[c * 3 + 1], s2 = scl[c * 3 + 2];
    const float cx = xyz[c * 3 + 0] * s0, cy = xyz[c * 3 + 1] * s1, cz = xyz[c * 3 + 2] * s2;
    const float2 cs = tab[i];
    const float ox = (-cs.x) * cx + cs.y * cz;
    const float oy = -cy;
    const float oz = (-cs.y) * cx - cs.x * cz;
    const float gx = grid_pos(pts[c * 3 + 0], ox, corner.x, res);
    const float gy = grid_pos(pts[c * 3 + 1], oy, corner.y, res);
    const float gz = grid_pos(pts[c * 3 + 2], oz, corner.z, res);
    if (gx < 0 || gy < 0 || gz < 0 || gx >= (float)(dims.x - 1) || gy >= (float)(dims.y - 1) ||
        gz >= (float)(dims.z - 1))
        return;
    const int fx = (int)gx, fy = (int)gy, fz = (int)gz;
    const float rx = gx - floorf(gx), ry = gy - floorf(gy), rz = gz - floorf(gz);
    const float wx[2] = {1.f - rx, rx}, wy[2] = {1.f - ry, ry}, wz[2] = {1.f - rz, rz};
    const float ob = obj[c];
    const int Y = dims.y, Z = dims.z;
#pragma unroll
    for (int bx = 0; bx < 2; ++bx)
#pragma unroll
        for (int by = 0; by < 2; ++by)
#pragma unroll
            for (int bz = 0; bz < 2; ++bz) {
                const float w = wx[bx] * wy[by] * wz[bz] * ob;
                const int64_t cell = ((int64_t)(fx + bx) * Y + (fy + by)) * Z + (fz + bz);
                unsafeAtomicAdd(&g_obj[cell], w);
                unsafeAtomicAdd(&g_rot[cell * 2 + 0], w * cs.x);
                unsafeAtomicAdd(&g_rot[cell * 2 + 1], w * cs.y);
                unsafeAtomicAdd(&g_scale[cell * 3 + 0], w * s0);
                unsafeAtomicAdd(&g_scale[cell * 3 + 1], w * s1);
                unsafeAtomicAdd(&g_scale[cell * 3 + 2], w * s2);
            }
}

// hv_cuda_kernel.cu:100-119, coalesced along Z (flat cell index).
__global__ __launch_bounds__(256) void hv_normalise(const float* __restrict__ g_obj,
                                                    float* __restrict__ g_rot,
                                                    float* __restrict__ g_scale, int64_t cells) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const double d = (double)g_obj[i] + 1e-7;
    float2 r = reinterpret_cast<float2*>(g_rot)[i];
    r.x = (float)((double)r.x / d);
    r.y = (float)((double)r.y / d);
    reinterpret_cast<float2*>(g_rot)[i] = r;
#pragma unroll
    for (int j = 0; j < 3; ++j) g_scale[i * 3 + j] = (float)((double)g_scale[i * 3 + j] / d);
}

// ---------------------------------------------------------------------------
// tiles algorithm
// ---------------------------------------------------------------------------
// Tile = HV_TX x 32 cells of one y plane, HV_TW waves per workgroup.
// 16 x 32 cells / 8 waves (68 KB of LDS, two workgroups per CU): the fastest shape one scene in flight (0.36 ms) and with
// scenes in flight (profiles/r3/vote_tile_sweep.txt).  Round 3 found it giving WRONG cells - a few dozen cells of one
// (plane, tile), weight moved between neighbouring cells - in about a third of the launches that ran while fp16 / bf16
// matrix-core convolutions of OTHER streams were resident on the same CU (tests/test_concurrency_gpu.py).  The cause is
// not in this kernel: on gfx950 a `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` - what the SLP vectoriser makes of the
// 2-D rotation of the offset below - returns wrong results while another wave of the CU issues
// v_mfma_f32_32x32x16_{f16,bf16} / 16x16x32_f16 (never next to the fp32 or the 32x32x8 f16 MFMA; pinned with synthetic
// co-resident loads and an instruction-class checker: profiles/vote_hammer_probe.py, op_check_probe.py,
// microbench/lds_hammer.hip, r3/vote_concurrency_findings.txt).  The whole library is built without packed fp32
// instructions now (csrc/build.py); with that every tile shape is exact under the same load (0 of 400 / 0 of 4800).
#ifndef HV_TX
#define HV_TX 16
#endif
#ifndef HV_TW
#define HV_TW 8
#endif
#ifndef HV_PART_RECORDS
#define HV_PART_RECORDS 4096
#endif
#ifndef HV_MAX_PARTS
#define HV_MAX_PARTS 8
#endif
constexpr int TX = HV_TX, TZ = 32, TCELLS = TX * TZ;   // TZ = 32 is built into acc_idx
constexpr int TW = HV_TW;    // waves per workgroup
constexpr int PQ = 64;       // surviving points per wave chunk
constexpr int VQ = 128;      // vote queue entries per wave
constexpr int MAX_R_TILES = 256;
#ifndef HV_LIST_PART_ENTRIES
#define HV_LIST_PART_ENTRIES 384
#endif
#ifndef HV_LIST_CHUNK
#define HV_LIST_CHUNK 16
#endif
constexpr int LIST_CHUNK = HV_LIST_CHUNK;                  // list entries a wave takes per hand-out
constexpr int LIST_PART_ENTRIES = HV_LIST_PART_ENTRIES;    // work-list entries one part of a hot (tile, plane) takes

// y cell of every vote of a point (theta-independent: offset.y = -corr.y, :38-39).
// Global atomics on a few hot addresses serialise at ~11 ns each on MI355X (measured: 80k
// atomics over 88 counters = 142 us), so bins are aggregated per workgroup in LDS first.
constexpr int LIST_CHUNK_RECORDS = 1024;     // records of one y-bin per work-list workgroup (hv_list_pass)
constexpr int PREP_MAX_Y = 4096;
constexpr int PREP_THREADS = 1024;

__global__ __launch_bounds__(PREP_THREADS) void hv_prep_count(
    const float* __restrict__ pts, const float* __restrict__ xyz, const float* __restrict__ scl,
    int64_t n, float res, float corner_y, int Y, int* __restrict__ fy_out, int* __restrict__ ycount) {
    __shared__ int lh[PREP_MAX_Y];
    const bool use_lds = Y <= PREP_MAX_Y;
    if (use_lds) {
        for (int i = threadIdx.x; i < Y; i += PREP_THREADS) lh[i] = 0;
        __syncthreads();
    }
    const int64_t c = blockIdx.x * (int64_t)PREP_THREADS + threadIdx.x;
    if (c < n) {
        const float cy = xyz[c * 3 + 1] * scl[c * 3 + 1];
        const float gy = grid_pos(pts[c * 3 + 1], -cy, corner_y, res);
        int fy = -1;
        if (gy >= 0 && gy < (float)(Y - 1)) fy = (int)gy;
        fy_out[c] = fy;
        if (fy >= 0) atomicAdd(use_lds ? &lh[fy] : &ycount[fy], 1);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < Y; i += PREP_THREADS)
            if (lh[i]) atomicAdd(&ycount[i], lh[i]);
    }
}

// exclusive scan of ycount[Y] -> ystart[Y+1], cursor[Y] (single workgroup)
// Hough peaks concentrate votes: the hottest (tile, plane) of an 80k scene receives ~27x the mean and
// would take 0.3 ms on one CU.  Planes whose two y-bins hold many points are therefore split into
// up to MAX_PARTS workgroups per tile (disjoint record chunks), merged by the last arriver.
constexpr int PART_RECORDS = HV_PART_RECORDS;
constexpr int MAX_PARTS = HV_MAX_PARTS;

// (round 3: the three scans used to be Hillis-Steele passes over 1024 LDS slots with two workgroup barriers per step -
// sixty barriers for Y = 88 bins, 7.5 us; they are independent, so three waves run one wave-level scan each: 64 bins per step)
template <class V, class E>
__device__ __forceinline__ int wave_excl_scan(int Y, V value, E emit) {
    const int lane = threadIdx.x & 63;
    int carry = 0;
    for (int base = 0; base < Y; base += 64) {
        const int i = base + lane;
        const int v = i < Y ? value(i) : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (i < Y) emit(i, carry + incl - v, v);
        carry += __shfl(incl, 63);
    }
    return carry;
}
__global__ __launch_bounds__(256) void hv_prep_scan(const int* __restrict__ ycount, int Y,
                                                    int* __restrict__ ystart,
                                                    int* __restrict__ cursor,
                                                    int* __restrict__ part_start, int4* __restrict__ q_info, int max_q,
                                                    int* __restrict__ chunk_start, int* __restrict__ bin_of_chunk,
                                                    int part_records) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        const int total = wave_excl_scan(Y, [&](int i) { return ycount[i]; },
                                         [&](int i, int excl, int) { ystart[i] = excl; cursor[i] = excl; });
        if (lane == 0) ystart[Y] = total;
    } else if (wave == 1) {
        // streaming path (small grids): part_start[y] = exclusive scan of the number of workgroups per tile of plane y, by
        // the records of its two bins; q_info[(plane, part) slot] = (plane, part, parts of the plane, first slot of the plane),
        // nparts = -1 on the unused slots behind the last one: ONE 16-byte load in front of a tile workgroup's work (round 1: a
        // binary search over part_start, seven dependent loads; round 2: three dependent loads part_start[Y] ->
        // plane_of_q[q] -> part_start[y], part_start[y + 1])
        const int total = wave_excl_scan(Y,
            [&](int i) {
                const int n2 = (i >= 1 ? ycount[i - 1] : 0) + (i <= Y - 2 ? ycount[i] : 0);
                return min(max((n2 + part_records - 1) / part_records, 1), MAX_PARTS);
            },
            [&](int i, int excl, int v) {
                part_start[i] = excl;
                for (int p2 = 0; p2 < v; ++p2) q_info[excl + p2] = make_int4(i, p2, v, excl);
            });
        for (int q = total + lane; q < max_q; q += 64) q_info[q] = make_int4(0, 0, -1, 0);
        if (lane == 0) part_start[Y] = total;
    } else if (wave == 2) {
        // chunks of up to LIST_CHUNK_RECORDS records of one bin (the work-list passes run one workgroup per chunk)
        const int total = wave_excl_scan(Y, [&](int i) { return (ycount[i] + LIST_CHUNK_RECORDS - 1) / LIST_CHUNK_RECORDS; },
                                         [&](int i, int excl, int v) {
                                             chunk_start[i] = excl;
                                             for (int p2 = 0; p2 < v; ++p2) bin_of_chunk[excl + p2] = i;
                                         });
        if (lane == 0) chunk_start[Y] = total;
    }
}

// Scatters every point with an in-bounds y into its y-bin and writes a compact SoA record
// (REC_F floats, bin order) so the tile kernel streams coalesced rows instead of gathering
// 10 scalars per point through `order[]`:
//   0 px  1 pz  2 cx  3 cz  4 ry (fractional y)  5 obj  6..8 scale  9 ux  10 uz  11 r  12 a0
// (ux,uz) = ring centre, r = ring radius in grid units and a0 = angle (seen from the ring centre)
// of the vote at rotation 0; used only for culling.
constexpr int REC_F = 13;

__global__ __launch_bounds__(PREP_THREADS) void hv_prep_scatter(
    const float* __restrict__ pts, const float* __restrict__ xyz, const float* __restrict__ scl,
    const float* __restrict__ obj, const int* __restrict__ fy_in, int64_t n, int Y, float res,
    F3 corner, int* __restrict__ cursor, float* __restrict__ rec, int64_t rec_stride) {
    __shared__ int lh[PREP_MAX_Y];     // per-workgroup count, then the workgroup's base in the bin
    const bool use_lds = Y <= PREP_MAX_Y;
    const int64_t c = blockIdx.x * (int64_t)PREP_THREADS + threadIdx.x;
    const int fy = c < n ? fy_in[c] : -1;
    int pos = -1;
    if (!use_lds) {
        if (fy >= 0) pos = atomicAdd(&cursor[fy], 1);
    } else {
        for (int i = threadIdx.x; i < Y; i += PREP_THREADS) lh[i] = 0;
        __syncthreads();
        int rank = 0;
        if (fy >= 0) rank = atomicAdd(&lh[fy], 1);
        __syncthreads();
        for (int i = threadIdx.x; i < Y; i += PREP_THREADS)
            if (lh[i]) lh[i] = atomicAdd(&cursor[i], lh[i]);
        __syncthreads();
        if (fy >= 0) pos = lh[fy] + rank;
    }
    if (pos < 0) return;
    const float px = pts[c * 3 + 0], py = pts[c * 3 + 1], pz = pts[c * 3 + 2];
    const float s0 = scl[c * 3 + 0], s1 = scl[c * 3 + 1], s2 = scl[c * 3 + 2];
    const float cx = xyz[c * 3 + 0] * s0, cy = xyz[c * 3 + 1] * s1, cz = xyz[c * 3 + 2] * s2;
    const float gy = grid_pos(py, -cy, corner.y, res);
    float* r = rec + pos;
    r[0 * rec_stride] = px;
    r[1 * rec_stride] = pz;
    r[2 * rec_stride] = cx;
    r[3 * rec_stride] = cz;
    r[4 * rec_stride] = gy - floorf(gy);
    r[5 * rec_stride] = obj[c];
    r[6 * rec_stride] = s0;
    r[7 * rec_stride] = s1;
    r[8 * rec_stride] = s2;
    r[9 * rec_stride] = (px - corner.x) / res;
    r[10 * rec_stride] = (pz - corner.z) / res;
    r[11 * rec_stride] = sqrtf(cx * cx + cz * cz) / res;
    // vote(theta) = u - r (cos(theta + phi), sin(theta + phi)) with (cx, cz) = r (cos phi, sin phi)
    r[12 * rec_stride] = atan2f(cz, cx) + 3.14159265f;
}


// ---- per-(y-bin, tile) work lists (round 3) ------------------------------------------------------------------------
// Every tile of a plane used to stream ALL records of its two y-bins and cull them (ring vs rectangle, then the arc):
// 66 tiles x 88 planes x ~1800 records at 80k points, 41 % of the kernel's wave time and most of its VALU work
// (profiles/r3/vote_pmc_sq_before.txt).  The test depends on (record, tile) only, so it is done ONCE: one workgroup per
// y-bin walks its records twice (count, then fill) over the tiles in the ring's bounding box and writes, per (bin, tile),
// the list of (record, first rotation, arc length) entries the tile kernel expands.  Lists live in one bump-allocated
// array; when it overflows (rings that cover the whole grid) a flag sends the tile kernel back to the streaming path.
constexpr int LIST_MAX_TILES = 4096;
struct RingArc { bool keep; int a_start, a_len; };
__device__ __forceinline__ bool ring_touches(float ux, float uz, float r, float xlo, float xhi, float zlo, float zhi,
                                             float& dxn, float& dzn) {
    const float slack = 0.05f;
    dxn = fmaxf(0.f, fmaxf(xlo - ux, ux - xhi));
    dzn = fmaxf(0.f, fmaxf(zlo - uz, uz - zhi));
    const float dxf = fmaxf(fabsf(ux - xlo), fabsf(ux - xhi));
    const float dzf = fmaxf(fabsf(uz - zlo), fabsf(uz - zhi));
    const float dmin = sqrtf(dxn * dxn + dzn * dzn), dmax = sqrtf(dxf * dxf + dzf * dzf);
    const float tol = slack + 1e-5f * (r + fabsf(ux) + fabsf(uz));
    return (r >= dmin - tol) && (r <= dmax + tol);
}
// rotations whose vote can fall in the rectangle: it subtends an arc seen from the ring centre unless the centre is
// (nearly) inside it
__device__ __forceinline__ void ring_arc(float ux, float uz, float a0, int R, float xlo, float xhi, float zlo, float zhi,
                                         float dxn, float dzn, int& a_start, int& a_len) {
    a_start = 0;
    a_len = R;
    if (dxn + dzn > 0.5f) {
        const float b0 = atan2f(0.5f * (zlo + zhi) - uz, 0.5f * (xlo + xhi) - ux);
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = atan2f(((k & 2) ? zhi : zlo) - uz, ((k & 1) ? xhi : xlo) - ux) - b0;
            d -= 6.28318531f * rintf(d * 0.159154943f);
            lo = fminf(lo, d);
            hi = fmaxf(hi, d);
        }
        const float inv_step = (float)R * 0.159154943f;          // 1 / rot_interval
        float first = (b0 + lo - a0) * inv_step - 2.f;           // 2 steps of slack per side
        const int len = (int)((hi - lo) * inv_step) + 6;
        first -= (float)R * floorf(first / (float)R);
        a_start = min(max((int)first, 0), R - 1);
        a_len = min(len, R);
    }
}

// Launches over CHUNKS of up to 1024 records of one y-bin (a bin per workgroup left the floor's bin - a quarter of an 80k
// scene - to one workgroup: 200 us):
//   hv_list_pass<false>  per (bin, tile): the summed arc lengths of the rings that reach the tile (tile_w: what the work
//                        queue sizes the parts of a hot tile with) and, when lists are wanted, the entry counts
//                        (atomicAdd on the (bin, tile) total returns the chunk's first slot inside that list)
//   hv_list_scan         lists only, one workgroup: exclusive scan of the entry counts (list starts, overflow flag)
//   hv_list_pass<true>   lists only, the same walk again: entry -> list_start + chunk slot + LDS rank
// chunk_start[Y + 1] / bin_of_chunk[] come from hv_prep_scan.
template <bool FILL>
__global__ __launch_bounds__(1024) void hv_list_pass(const int* __restrict__ ystart, const int* __restrict__ chunk_start,
                                                     const int* __restrict__ bin_of_chunk, int Y,
                                                     const float* __restrict__ rec, int64_t rec_stride, int R, int tiles_x,
                                                     int tiles_z, const int* __restrict__ list_ctl, int* __restrict__ list_cnt,
                                                     const int* __restrict__ list_start, int* __restrict__ chunk_off,
                                                     int2* __restrict__ entries, int* __restrict__ tile_w, int want_lists) {
    __shared__ int cnt[LIST_MAX_TILES];
    __shared__ int wsum[FILL ? 1 : LIST_MAX_TILES];
    const int ntiles = tiles_x * tiles_z;
    const int c = blockIdx.x;
    if (c >= chunk_start[Y]) return;
    if (FILL && list_ctl[1] != 0) return;             // overflow: the tile kernel streams the bins
    const int bin = bin_of_chunk[c];
    const int idx = ystart[bin] + (c - chunk_start[bin]) * LIST_CHUNK_RECORDS + (int)threadIdx.x;
    const int end = ystart[bin + 1];
    for (int t = threadIdx.x; t < ntiles; t += 1024) {
        cnt[t] = FILL ? list_start[(int64_t)bin * ntiles + t] + chunk_off[(int64_t)c * ntiles + t] : 0;
        if (!FILL) wsum[t] = 0;
    }
    __syncthreads();
    if (idx < end) {
        const float ux = rec[9 * rec_stride + idx], uz = rec[10 * rec_stride + idx], r = rec[11 * rec_stride + idx];
        const float a0 = rec[12 * rec_stride + idx];
        // tiles whose rectangle [x0 - 1, x0 + TX] x [z0 - 1, z0 + TZ] can meet the ring's bounding box (+ slack)
        const float ext = r + 2.f + 1e-5f * (r + fabsf(ux) + fabsf(uz));
        const int tx0 = max(0, (int)floorf((ux - ext - (float)TX) / (float)TX));
        const int tx1 = min(tiles_x - 1, (int)floorf((ux + ext + 1.f) / (float)TX));
        const int tz0 = max(0, (int)floorf((uz - ext - (float)TZ) / (float)TZ));
        const int tz1 = min(tiles_z - 1, (int)floorf((uz + ext + 1.f) / (float)TZ));
        for (int tx = tx0; tx <= tx1; ++tx)
            for (int tz = tz0; tz <= tz1; ++tz) {
                const float xlo = (float)(tx * TX - 1), xhi = (float)(tx * TX + TX), zlo = (float)(tz * TZ - 1),
                            zhi = (float)(tz * TZ + TZ);
                float dxn, dzn;
                if (!ring_touches(ux, uz, r, xlo, xhi, zlo, zhi, dxn, dzn)) continue;
                int a_start, a_len;
                ring_arc(ux, uz, a0, R, xlo, xhi, zlo, zhi, dxn, dzn, a_start, a_len);
                const int t = tx * tiles_z + tz;
                if (FILL) {
                    const int p = atomicAdd(&cnt[t], 1);
                    entries[p] = make_int2(idx, a_start | (a_len << 16));
                } else {
                    if (want_lists) atomicAdd(&cnt[t], 1);
                    atomicAdd(&wsum[t], a_len);
                }
            }
    }
    if (FILL) return;
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += 1024) {
        if (want_lists) chunk_off[(int64_t)c * ntiles + t] = cnt[t] ? atomicAdd(&list_cnt[(int64_t)bin * ntiles + t], cnt[t]) : 0;
        if (wsum[t]) atomicAdd(&tile_w[(int64_t)bin * ntiles + t], wsum[t]);
    }
}

// list_ctl[0] = total entries, list_ctl[1] = overflow flag (zeroed by the caller's fill)
__global__ __launch_bounds__(1024) void hv_list_scan(const int* __restrict__ list_cnt, int Y, int ntiles, long long list_cap,
                                                     int* __restrict__ list_ctl, int* __restrict__ list_start) {
    __shared__ int s[1024];
    __shared__ long long carry_s;      // 64 bits: a total past 2^31 must raise the overflow flag, not wrap
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int total_lists = Y * ntiles;
    for (int base = 0; base < total_lists; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < total_lists ? list_cnt[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        const long long incl = (long long)s[threadIdx.x] + carry_s;
        if (i < total_lists) list_start[i] = incl - v < 0x7fffffffll ? (int)(incl - v) : 0x7fffffff;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) { list_ctl[0] = carry_s < 0x7fffffffll ? (int)carry_s : 0x7fffffff; list_ctl[1] = carry_s > list_cap ? 1 : 0; }
}

// ---- work queue of the tile kernel (round 3) -----------------------------------------------------------------------
// The launch used to hold (plane, part) x tile workgroups with the parts of a plane sized by the RECORDS of its two
// bins.  The time of the kernel was the time of its hottest tile: 1 % of the (plane, tile) pairs of an 80k scene
// receive 20 % of the votes (up to 26 x the mean of the non-empty ones) while a plane's record count says nothing about
// where its rings meet.  The queue holds one item per (plane, tile) that receives nothing (it only writes zeros) and
// ceil(weight / PART_VOTES) items for the others, weight = summed arc lengths of the rings that reach the tile in the
// plane's two bins (hv_list_pass<false>); parts merge through `partials` slots handed out by the same scan.
//   item = (plane, tile, part | parts << 8, first partial slot of the (plane, tile))
#ifndef HV_PART_VOTES
#define HV_PART_VOTES 32768
#endif
constexpr int PART_VOTES = HV_PART_VOTES;
constexpr int QUEUE_MAX_PARTS = 32;

__global__ __launch_bounds__(1024) void hv_build_queue(const int* __restrict__ tile_w, int Y, int ntiles, int max_items,
                                                       int max_slots, int* __restrict__ list_ctl /*[2] = items*/,
                                                       int4* __restrict__ items) {
    __shared__ unsigned long long s[1024];
    __shared__ unsigned long long carry_s;
    __shared__ int single_s;
    const int total = Y * ntiles;
    // pass 0 sizes the parts by weight; when the items or the partial slots would not fit, pass 1 gives every tile one part
    for (int pass = 0; pass < 2; ++pass) {
        if (threadIdx.x == 0) { carry_s = 0ull; single_s = pass; }
        __syncthreads();
        for (int base = 0; base < total; base += 1024) {
            const int i = base + threadIdx.x;
            int parts = 0, y = 0, t = 0;
            if (i < total) {
                y = i / ntiles; t = i - y * ntiles;
                const long long w = (long long)(y >= 1 ? tile_w[(int64_t)(y - 1) * ntiles + t] : 0) +
                                    (long long)(y <= Y - 2 ? tile_w[(int64_t)y * ntiles + t] : 0);
                parts = w == 0 ? 0 : (pass ? 1 : (int)min((long long)QUEUE_MAX_PARTS, (w + PART_VOTES - 1) / PART_VOTES));
            }
            // items in the high word, partial slots (tiles with more than one part) in the low word
            const unsigned long long v = i < total ? (((unsigned long long)max(parts, 1) << 32) | (unsigned)(parts > 1 ? parts : 0)) : 0ull;
            s[threadIdx.x] = v;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) {
                const unsigned long long x = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0ull;
                __syncthreads();
                s[threadIdx.x] += x;
                __syncthreads();
            }
            const unsigned long long excl = s[threadIdx.x] + carry_s - v;
            const int item0 = (int)(excl >> 32), slot0 = (int)(excl & 0xffffffffull);
            if (i < total && item0 + max(parts, 1) <= max_items && slot0 + (parts > 1 ? parts : 0) <= max_slots) {
                if (parts == 0) items[item0] = make_int4(y, t, 0, 0);
                for (int p2 = 0; p2 < parts; ++p2) items[item0 + p2] = make_int4(y, t, p2 | (parts << 8), slot0);
            }
            __syncthreads();
            if (threadIdx.x == 1023) carry_s = s[1023] + carry_s;
            __syncthreads();
        }
        const int n_items = (int)(carry_s >> 32), n_slots = (int)(carry_s & 0xffffffffull);
        if (n_items <= max_items && n_slots <= max_slots) {
            if (threadIdx.x == 0) list_ctl[2] = n_items;
            return;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) list_ctl[2] = min(total, max_items);     // (total <= max_items by construction)
}

__device__ __forceinline__ int lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// LDS atomic add rates on gfx950 (lane-ops/clk/CU, measured: profiles/microbench/lds_atomic_rate.hip):
// f32 0.33, f64 3.1, u64 5.6, u32 7.4.  The LDS atomics are 63 % of this kernel's time (ablation: 0.68 ms ->
// 0.25 ms without them).  The objectness channel - the one the decode thresholds and takes argmax of, and the
// one that defines which cells are touched at all - accumulates in f64 (exact sums of the fp32 contributions).
// The five quotient numerators (rot cos/sin, scale xyz) accumulate in 64-bit FIXED POINT with ds_add_u64: every
// fp32 contribution is rounded once to a multiple of 2^-36 (1.5e-11; |sum| < 1.3e8 per cell and channel) and
// integer addition is exact and order independent, so all six sums are reproducible run to run; the rounding is
// 1e-11 relative on cells with weight >= 1 and only matters on cells whose total weight is below ~1e-6 (their
// rot / scale quotients lose digits; the reference's own fp32 atomics lose ~1e-7 relative per add everywhere).
constexpr double FX_SCALE = 68719476736.0;            // 2^36
constexpr double FX_MAGIC = 6755399441055744.0;       // 1.5 * 2^52: (x*2^36 + MAGIC) has round(x*2^36) in its low bits
// round(v * 2^36) to nearest even without a floating-point instruction wider than the input (experiment HV_INT_FX:
// are the f64 conversions what concurrent matrix kernels disturb?); exact for every finite v with |v| < 2^26
__device__ __forceinline__ unsigned long long fx_from_float_int(float v) {
    const unsigned b = __float_as_uint(v);
    const int e = (int)((b >> 23) & 255u);
    if (e == 0) return 0ull;                                  // zero / fp32 denormal: far below one quantum
    unsigned long long m = (unsigned long long)((b & 0x7fffffu) | 0x800000u);
    const int sh = e - 114;                                   // value = m * 2^(e - 150); times 2^36
    unsigned long long q;
    if (sh >= 0) q = sh < 40 ? (m << sh) : (m << 39);
    else if (sh <= -26) q = 0ull;
    else {
        const int r = -sh;
        const unsigned long long half = 1ull << (r - 1), rem = m & ((1ull << r) - 1ull);
        q = m >> r;
        if (rem > half || (rem == half && (q & 1ull))) ++q;
    }
    return (b >> 31) ? (0ull - q) : q;
}
template <bool SMALL>
__device__ __forceinline__ void lds_add(unsigned long long* p, float v) {
    unsigned long long q;
#ifdef HV_INT_FX
    q = fx_from_float_int(SMALL ? v : fminf(fmaxf(v, -6.0e7f), 6.0e7f));
    __hip_atomic_fetch_add(p, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
#endif
    if (SMALL) {                              // the magic-number conversion holds for |v * 2^36| < 2^51
        const double d = __builtin_fma((double)v, FX_SCALE, FX_MAGIC);
        q = (unsigned long long)__double_as_longlong(d) - (unsigned long long)__double_as_longlong(FX_MAGIC);
    } else {                                  // huge contributions (a diverged scale head): exact up to 6e7, clamped beyond
        q = (unsigned long long)__double2ll_rn((double)fminf(fmaxf(v, -6.0e7f), 6.0e7f) * FX_SCALE);
    }
    __hip_atomic_fetch_add(p, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double fx_value(unsigned long long q) { return (double)(long long)q * (1.0 / FX_SCALE); }
// The objectness weight - the channel that is thresholded, arg-maxed and defines the touched-cell set - accumulated in
// f64 until round 3: sums of fp32 contributions spanning more than 53 bits are not exact in f64, so the LDS atomics'
// order reached the last bits and, once in ~1600 runs of one scene, a cell on an fp32 rounding boundary changed the
// greedy walk (tests/test_concurrency_gpu.py found it).  It now uses the same 2^-36 fixed point as the numerators
// (integer addition: exact, order independent, bit-reproducible) with ONE extra rule: a non-zero contribution never rounds
// to zero (it adds one quantum, 1.5e-11), so a cell is non-zero exactly when the reference's fp32 sum is.
template <bool SMALL>
__device__ __forceinline__ void lds_add_obj(unsigned long long* p, float v) {
    unsigned long long q;
#ifdef HV_INT_FX
    q = fx_from_float_int(SMALL ? v : fminf(fmaxf(v, -6.0e7f), 6.0e7f));
    if (q == 0ull && v != 0.f) q = v > 0.f ? 1ull : ~0ull;
    __hip_atomic_fetch_add(p, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
#endif
    if (SMALL) {
        const double d = __builtin_fma((double)v, FX_SCALE, FX_MAGIC);
        q = (unsigned long long)__double_as_longlong(d) - (unsigned long long)__double_as_longlong(FX_MAGIC);
    } else {
        q = (unsigned long long)__double2ll_rn((double)fminf(fmaxf(v, -6.0e7f), 6.0e7f) * FX_SCALE);
    }
    if (q == 0ull && v != 0.f) q = v > 0.f ? 1ull : ~0ull;
    __hip_atomic_fetch_add(p, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// value of accumulator word i of the [6][TCELLS] tile (every channel in 2^-36 fixed point)
__device__ __forceinline__ double acc_value(unsigned long long q, int i) {
    (void)i;
    return fx_value(q);
}

// LDS layout of the tile accumulators: word of (channel ch, cell (cx, cz)) = ch * ACC_CH + cx * ACC_PITCH + cz.
// With a pitch of 32 words the bank of a cell (64 banks x 4 B, a 64-bit word takes two) depended on cz alone, so the
// lanes of one atomic instruction whose votes spread along x - a ring crossing the tile, the cluster around a Hough
// peak - serialised on one bank.  Pitch 40: +1 in z = 2 banks, +1 in x = 80 = 16 banks (mod 64): every cell of a 4x4
// neighbourhood has its own bank.  Vote op 0.512 -> 0.447 ms on the network's predictions (pitches 34 / 36 / 40 / 44:
// 0.457 / 0.445 / 0.447 / 0.443).  Measured and dropped: walking the (corner, channel) slots of a vote in a per-lane
// rotated order so that lanes voting into the SAME cell hit different words in different banks at any one instruction -
// 0.62 ms (same-address lanes of one ds_add are evidently merged more cheaply than 24 rotated selects cost), rotating
// the channel order only - 0.45 / 0.47 ms (no gain).
constexpr int ACC_PITCH = TZ + 8, ACC_CH = TX * ACC_PITCH + 4, ACC_WORDS = 6 * ACC_CH;
__device__ __forceinline__ int acc_idx(int ch, int cell) { return ch * ACC_CH + (cell >> 5) * ACC_PITCH + (cell & 31); }

struct TileShared {
    unsigned long long acc[ACC_WORDS];   // channel 0: objectness weight, 1..5: rot.cos, rot.sin, scale.xyz - all
                                         // in 2^-36 fixed point
    float pq[TW][9][PQ];       // px, pz, cx, cz, wy, obj, s0, s1, s2 of surviving points
    int arc_start[TW][PQ];     // first rotation whose vote can reach the tile
    int arc_cum[TW][PQ];       // inclusive prefix sum of the arc lengths
    uint32_t vq_rec[TW][VQ];   // entry | rot<<6 | (lx+1)<<14 | (lz+1)<<20
    float vq_rx[TW][VQ];
    float vq_rz[TW][VQ];
    float2 tab[MAX_R_TILES];
    int next_chunk[2];         // dynamic hand-out of 64-record chunks of the two y-bins to the waves
#ifdef HV_LDS_PAD
    char lds_pad[HV_LDS_PAD];  // experiment: LDS the workgroup does not use, to control what else fits on its CU
#endif
};

template <bool SMALL>
__device__ __forceinline__ void drain_vote(TileShared& sh, int lx, int lz, float rx, float rz, float wy, float ob,
                                           float s0, float s1, float s2, float2 cs) {
    const float wx[2] = {1.f - rx, rx}, wz[2] = {1.f - rz, rz};
#pragma unroll
    for (int bx = 0; bx < 2; ++bx)
#pragma unroll
        for (int bz = 0; bz < 2; ++bz) {
            const int cxl = lx + bx, czl = lz + bz;
            if (cxl < 0 || cxl >= TX || czl < 0 || czl >= TZ) continue;
            // hv_cuda_kernel.cu:52-59 order: ((wx*wy)*wz)*objness
            const float w = wx[bx] * wy * wz[bz] * ob;
            unsigned long long* a = sh.acc + cxl * ACC_PITCH + czl;
            lds_add_obj<SMALL>(a, w);
            lds_add<SMALL>(a + ACC_CH, w * cs.x);
            lds_add<SMALL>(a + 2 * ACC_CH, w * cs.y);
            lds_add<SMALL>(a + 3 * ACC_CH, w * s0);
            lds_add<SMALL>(a + 4 * ACC_CH, w * s1);
            lds_add<SMALL>(a + 5 * ACC_CH, w * s2);
        }
}

__device__ __forceinline__ void drain64(TileShared& sh, int wave, int slot, bool active) {
    float rx = 0.f, rz = 0.f, wy = 0.f, ob = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
    float2 cs = make_float2(0.f, 0.f);
    int lx = 0, lz = 0;
    // (round 6, VERDICT r5 item 3b, profiles/r6/vote_bank_bucket.txt: dealing the 64 votes of a drain to the two 32-lane halves by
    // their rank among the lanes of the same bank pair - five ballots for the same-bank mask, two for the positions, one
    // ds_permute - took SQ_LDS_BANK_CONFLICT from 3.98e7 to 3.41e7 per launch (45.6 % -> 41.1 % of SQ_LDS_IDX_ACTIVE) and the LDS
    // wait cycles down by a third, and cost 7 % more wave cycles: 0.357 against 0.333 ms, 591 against 598 scenes/s.  Removed.
    // profiles/vote_run_pricing.py has the ceilings: a PERFECT permutation inside the 64 saves 23 % of the bank cycles, merging
    // runs of votes that keep their floor cell 28 % of the atomics (mean run length 1.38).)
    if (active) {
        const uint32_t rec = sh.vq_rec[wave][slot];
        rx = sh.vq_rx[wave][slot]; rz = sh.vq_rz[wave][slot];
        const int e = rec & 63, rot = (rec >> 6) & 255;
        lx = (int)((rec >> 14) & 63) - 1; lz = (int)((rec >> 20) & 63) - 1;
        wy = sh.pq[wave][4][e]; ob = sh.pq[wave][5][e];
        s0 = sh.pq[wave][6][e]; s1 = sh.pq[wave][7][e]; s2 = sh.pq[wave][8][e];
        cs = sh.tab[rot];
    }
    // every contribution of a vote is bounded by |obj| * max(1, |scale|) (the trilinear weights are <= 1); the
    // wave takes the fast fixed-point conversion unless one of its 64 votes could exceed its range
    const bool small = fabsf(ob) * fmaxf(1.f, fmaxf(fabsf(s0), fmaxf(fabsf(s1), fabsf(s2)))) < 16384.f;
    if (__all(small)) {
        if (active) drain_vote<true>(sh, lx, lz, rx, rz, wy, ob, s0, s1, s2, cs);
    } else {
        if (active) drain_vote<false>(sh, lx, lz, rx, rz, wy, ob, s0, s1, s2, cs);
    }
}

__device__ __forceinline__ void wave_sync_lds() {
#ifdef HV_STRONG_WAVE_SYNC
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// atan2 for the arc of rotations a tile can receive from a ring (conservative culling only: every vote is placed by the
// exact fp32 sequence afterwards).  |error| < 2e-4 rad against 2 rotation steps of slack per side (0.1 rad at 120 rotations,
// 0.05 at 256): the libm atan2f calls (five per kept record) were a third of the tile kernel's vector instructions.
__device__ __forceinline__ float arc_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float a = fminf(ax, ay) * __builtin_amdgcn_rcpf(fmaxf(fmaxf(ax, ay), 1e-30f));
    const float q = a * a;
    float r = ((-0.0464964749f * q + 0.15931422f) * q - 0.327622764f) * q * a + a;
    r = ay > ax ? 1.57079637f - r : r;
    r = x < 0.f ? 3.14159274f - r : r;
    return y < 0.f ? -r : r;
}

// VARIANT (ablations for profiling only): 0 full, 1 no LDS atomics, 2 no dense phase, 3 no record streaming
template <int VARIANT, bool QUEUE>
__global__ __launch_bounds__(TW * 64) void hv_fwd_tiles(
    int R, float res, F3 corner, I3 dims, const float2* __restrict__ tab,
    const int* __restrict__ ystart, const int4* __restrict__ items, const float* __restrict__ rec,
    int64_t rec_stride, int tiles_x, int tiles_z, unsigned long long* __restrict__ partials,
    int* __restrict__ arrivals, float* __restrict__ g_obj, float* __restrict__ g_rot,
    float* __restrict__ g_scale, unsigned long long* __restrict__ prof,
    const int* __restrict__ list_ctl, const int* __restrict__ list_start, const int* __restrict__ list_cnt,
    const int2* __restrict__ entries, int list_mode /* 1: stream the bins, 2: work lists */,
    const int4* __restrict__ q_info) {
    __shared__ TileShared sh;
    __shared__ int last_flag;
    // VARIANT 4: shader-clock ticks per phase, summed over waves into prof[0..7], prof[8] = waves
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pt0 = VARIANT == 4 ? __builtin_amdgcn_s_memtime() : 0;
#define HV_TICK(p)                                                        \
    do {                                                                  \
        if (VARIANT == 4) {                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
            pacc[p] += t_ - pt0;                                          \
            pt0 = t_;                                                     \
        }                                                                 \
    } while (0)
    const int X = dims.x, Y = dims.y, Z = dims.z;
    const int ntiles = tiles_x * tiles_z;
    // QUEUE: one work item per workgroup (hv_build_queue; large grids) - ONE 16-byte load says which (plane, tile,
    // part) this is.  !QUEUE (small grids, the round-2 launch): (plane, part) x tile workgroups, parts by the records of
    // the plane's bins (hv_prep_scan)
    int y, tile, part, nparts, slot0, slot_stride;
    if (QUEUE) {
        if ((int)blockIdx.x >= list_ctl[2]) return;
        const int4 item = items[blockIdx.x];
        y = item.x; tile = item.y; part = item.z & 0xff; nparts = item.z >> 8; slot0 = item.w; slot_stride = 1;
    } else {
        tile = blockIdx.x % ntiles;
        const int4 qi = q_info[blockIdx.x / ntiles];     // (plane, part) slot
        if (qi.z < 0) return;
        y = qi.x; part = qi.y; nparts = qi.z;
        slot0 = qi.w * ntiles + tile; slot_stride = ntiles;
    }
    const int x0 = (tile / tiles_z) * TX, z0 = (tile % tiles_z) * TZ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (nparts == 0) {
        // nothing votes into this tile of this plane: zeros straight to the grids (no accumulators, no barriers)
        const int nx0 = min(TX, X - x0), nz0 = min(TZ, Z - z0);
        for (int i = threadIdx.x; i < TCELLS; i += TW * 64) {
            const int lx = i / TZ, lz = i % TZ;
            if (lx < nx0 && lz < nz0) {
                const int64_t cell = ((int64_t)(x0 + lx) * Y + y) * Z + z0 + lz;
                g_obj[cell] = 0.f;
                reinterpret_cast<float2*>(g_rot)[cell] = make_float2(0.f, 0.f);
                g_scale[cell * 3 + 0] = 0.f; g_scale[cell * 3 + 1] = 0.f; g_scale[cell * 3 + 2] = 0.f;
            }
        }
        return;
    }
    // list_mode 2: work lists of this tile in the two y-bins (hv_list_pass; list_ctl[1] != 0: they overflowed, stream the
    // bins instead)
    const bool use_list = QUEUE && VARIANT != 3 && list_mode == 2 && list_ctl[1] == 0;      // (the queue launch is the list launch)
    int lbeg[2] = {0, 0}, llen[2] = {0, 0};
    if (use_list) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int s = y - 1 + k;
            if (s >= 0 && s <= Y - 2) {
                lbeg[k] = list_start[(int64_t)s * ntiles + tile];
                llen[k] = list_cnt[(int64_t)s * ntiles + tile];
            }
        }
    }

#ifdef HV_POISON      // debug: stale-LDS hunt (an uninitialised read shows up as a changed result)
    {
        unsigned* w = reinterpret_cast<unsigned*>(&sh);
        const int lo[5] = {(int)(offsetof(TileShared, pq) / 4), (int)(offsetof(TileShared, arc_start) / 4),
                           (int)(offsetof(TileShared, arc_cum) / 4), (int)(offsetof(TileShared, vq_rec) / 4),
                           (int)(offsetof(TileShared, tab) / 4)};
        const int hi[5] = {lo[1], lo[2], lo[3], lo[4], (int)(offsetof(TileShared, next_chunk) / 4)};
        for (int k = 0; k < 5; ++k)
            if ((HV_POISON >> k) & 1)
                for (int i = lo[k] + threadIdx.x; i < hi[k]; i += TW * 64) w[i] = 0x7fc12345u + i * 2654435761u;
        __syncthreads();
    }
#endif
    for (int i = threadIdx.x; i < ACC_WORDS; i += TW * 64) sh.acc[i] = 0ull;
    for (int i = threadIdx.x; i < R; i += TW * 64) sh.tab[i] = tab[i];
    if (threadIdx.x < 2) sh.next_chunk[threadIdx.x] = 0;
    __syncthreads();
    HV_TICK(0);

    // a vote at grid position g touches cells floor(g), floor(g)+1, so it reaches this
    // tile iff g in [x0-1, x0+TX) x [z0-1, z0+TZ); slack covers fp32 rounding of the test.
    const float slack = 0.05f;
    const float xlo = (float)(x0 - 1), xhi = (float)(x0 + TX), zlo = (float)(z0 - 1),
                zhi = (float)(z0 + TZ);
    int vq_len = 0;   // wave-uniform

    for (int s = y - 1; s <= y; ++s) {
        if (s < 0 || s > Y - 2 || VARIANT == 3) continue;
        const int beg = use_list ? lbeg[s - (y - 1)] : ystart[s];
        const int end = use_list ? beg + llen[s - (y - 1)] : ystart[s + 1];
        // waves take chunks from a shared counter: a wave whose chunk expands into many votes takes fewer chunks
        // (static striding left the waves of a workgroup waiting 22 % of their time for the slowest one)
        for (;;) {
            int c = 0;
            if (lane == 0) c = atomicAdd(&sh.next_chunk[s - (y - 1)], 1);
            c = __builtin_amdgcn_readfirstlane(c);
            // list entries are all kept and each expands into an arc: 16 per hand-out keep the waves of a workgroup even
            const int per = use_list ? LIST_CHUNK : 64;
            const int base = beg + (c * nparts + part) * per;
            if (base >= end) break;
            int idx = base + lane;
            bool keep = false;
            int a_start = 0, a_len = 0;
            if (use_list) {
                if (lane < per && idx < end) {
                    const int2 en = entries[idx];
                    idx = en.x;
                    a_start = en.y & 0xffff;
                    a_len = en.y >> 16;
                    keep = true;
                }
            } else if (idx < end) {
                const float ux = rec[9 * rec_stride + idx], uz = rec[10 * rec_stride + idx],
                            r = rec[11 * rec_stride + idx];
                // conservative ring-vs-rectangle test in grid units
                const float dxn = fmaxf(0.f, fmaxf(xlo - ux, ux - xhi));
                const float dzn = fmaxf(0.f, fmaxf(zlo - uz, uz - zhi));
                const float dxf = fmaxf(fabsf(ux - xlo), fabsf(ux - xhi));
                const float dzf = fmaxf(fabsf(uz - zlo), fabsf(uz - zhi));
                const float dmin = sqrtf(dxn * dxn + dzn * dzn), dmax = sqrtf(dxf * dxf + dzf * dzf);
                const float tol = slack + 1e-5f * (r + fabsf(ux) + fabsf(uz));
                keep = (r >= dmin - tol) && (r <= dmax + tol);
                if (VARIANT == 2) keep = keep && (ux == 1234.5f);
                if (keep) {
                    // rotations whose vote can fall in the rectangle: the rectangle subtends an arc
                    // [b0+lo, b0+hi] seen from the ring centre unless the centre is (nearly) inside it
                    a_len = R;
                    if (dxn + dzn > 0.5f) {
                        const float b0 = arc_atan2(0.5f * (zlo + zhi) - uz, 0.5f * (xlo + xhi) - ux);
                        float lo = 0.f, hi = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float d = arc_atan2(((k & 2) ? zhi : zlo) - uz, ((k & 1) ? xhi : xlo) - ux) - b0;
                            d -= 6.28318531f * rintf(d * 0.159154943f);
                            lo = fminf(lo, d);
                            hi = fmaxf(hi, d);
                        }
                        const float a0 = rec[12 * rec_stride + idx];
                        const float inv_step = (float)R * 0.159154943f;          // 1 / rot_interval
                        float first = (b0 + lo - a0) * inv_step - 2.f;           // 2 steps of slack per side
                        const int len = (int)((hi - lo) * inv_step) + 6;
                        first -= (float)R * floorf(first / (float)R);
                        a_start = min(max((int)first, 0), R - 1);
                        a_len = min(len, R);
                    }
                }
            }
            const uint64_t m = __ballot(keep);
            const int nq = __popcll(m);
            HV_TICK(1);
            if (nq == 0) continue;
            // inclusive scan of the arc lengths in compacted (lane) order
            int cum = keep ? a_len : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(cum, off);
                if (lane >= off) cum += t;
            }
            const int items = __shfl(cum, 63);
            if (keep) {
                const int p = lanes_below(m);
                const float ry = rec[4 * rec_stride + idx];
                sh.pq[wave][0][p] = rec[0 * rec_stride + idx];
                sh.pq[wave][1][p] = rec[1 * rec_stride + idx];
                sh.pq[wave][2][p] = rec[2 * rec_stride + idx];
                sh.pq[wave][3][p] = rec[3 * rec_stride + idx];
                sh.pq[wave][4][p] = (s == y) ? (1.f - ry) : ry;
                sh.pq[wave][5][p] = rec[5 * rec_stride + idx];
                sh.pq[wave][6][p] = rec[6 * rec_stride + idx];
                sh.pq[wave][7][p] = rec[7 * rec_stride + idx];
                sh.pq[wave][8][p] = rec[8 * rec_stride + idx];
                sh.arc_start[wave][p] = a_start;
                sh.arc_cum[wave][p] = cum;
            }
            wave_sync_lds();
            HV_TICK(2);

            // lane l walks items [l*S, (l+1)*S): at any step the 64 lanes sit on 64 different arcs
            // (different cells -> few same-address LDS atomic collisions), and a lane only advances
            // along its arc, so the (entry, step) pair is found by ONE binary search per chunk.
            const int S = (items + 63) >> 6;
            int it0 = lane * S;
            const int it1 = min(it0 + S, items);
            int e = 0;
            {
                int lo = 0, hi = nq - 1;                     // smallest e with cum[e] > it0
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sh.arc_cum[wave][mid] > it0) hi = mid; else lo = mid + 1;
                }
                e = lo;
            }
            int e_end = it0 < items ? sh.arc_cum[wave][e] : 0;          // first item of the next entry
            int rot = 0;
            // the entry's point stays in registers while the lane walks its arc (4 LDS reads per entry instead of
            // per step: the LDS pipe is what the expansion shares with the f64 atomics)
            float epx = 0.f, epz = 0.f, ecx = 0.f, ecz = 0.f;
            if (it0 < items) {
                const int e_beg = e > 0 ? sh.arc_cum[wave][e - 1] : 0;
                rot = sh.arc_start[wave][e] + (it0 - e_beg);
                if (rot >= R) rot -= R;
                epx = sh.pq[wave][0][e]; epz = sh.pq[wave][1][e];
                ecx = sh.pq[wave][2][e]; ecz = sh.pq[wave][3][e];
            }
            HV_TICK(2);
            for (int step = 0; step < S; ++step, ++it0) {
                bool isvote = false;
                uint32_t vrec = 0;
                float rx = 0, rz = 0;
                if (it0 < it1) {
                    if (it0 == e_end) {                      // next arc
                        ++e;
                        e_end = sh.arc_cum[wave][e];
                        rot = sh.arc_start[wave][e];
                        epx = sh.pq[wave][0][e]; epz = sh.pq[wave][1][e];
                        ecx = sh.pq[wave][2][e]; ecz = sh.pq[wave][3][e];
                    }
                    const float2 cs = sh.tab[rot];
                    const float ox = (-cs.x) * ecx + cs.y * ecz;
                    const float oz = (-cs.y) * ecx - cs.x * ecz;
                    const float gx = grid_pos(epx, ox, corner.x, res);
                    const float gz = grid_pos(epz, oz, corner.z, res);
                    if (gx >= 0 && gz >= 0 && gx < (float)(X - 1) && gz < (float)(Z - 1)) {
                        const int lx = (int)gx - x0, lz = (int)gz - z0;
                        if (lx >= -1 && lx < TX && lz >= -1 && lz < TZ) {
                            isvote = true;
                            rx = gx - floorf(gx);
                            rz = gz - floorf(gz);
                            vrec = (uint32_t)e | ((uint32_t)rot << 6) | ((uint32_t)(lx + 1) << 14) |
                                   ((uint32_t)(lz + 1) << 20);
                        }
                    }
                    if (++rot == R) rot = 0;
                }
                const uint64_t mv = __ballot(isvote);
                if (isvote) {
                    const int p = vq_len + lanes_below(mv);
                    sh.vq_rec[wave][p] = vrec;
                    sh.vq_rx[wave][p] = rx;
                    sh.vq_rz[wave][p] = rz;
                }
                vq_len += __popcll(mv);
                wave_sync_lds();
                HV_TICK(4);
                if (vq_len >= 64) {
                    vq_len -= 64;
                    if (VARIANT != 1) drain64(sh, wave, vq_len + lane, true);
                    HV_TICK(5);
                }
            }
            // queued votes index this chunk's pq entries: flush before pq is overwritten
            if (vq_len > 0) {
                if (VARIANT != 1) drain64(sh, wave, lane, lane < vq_len);
                else if (lane < vq_len)
                    sh.acc[ACC_CH + lane] = (unsigned long long)(sh.vq_rx[wave][lane] + sh.vq_rz[wave][lane] + (float)sh.vq_rec[wave][lane]);
                vq_len = 0;
            }
            wave_sync_lds();
            HV_TICK(6);
        }
    }
    HV_TICK(1);
    __syncthreads();
    HV_TICK(3);

    if (nparts > 1) {
        // publish this part's tile as the raw 2^-36 fixed-point words, the last arriver adds the parts as integers: the
        // merged sums are the same bits whichever records landed in whichever part (until round 4 the parts went through
        // fp32, so the last bit of a hot cell depended on the atomic order of the scatter that fills the parts)
        // plain stores -> per-wave vmcnt(0) -> barrier -> one-lane agent release -> ticket.
        unsigned long long* mine = partials + ((int64_t)slot0 + (int64_t)part * slot_stride) * (6 * TCELLS);
        for (int i = threadIdx.x; i < 6 * TCELLS; i += TW * 64) mine[i] = sh.acc[acc_idx(i / TCELLS, i % TCELLS)];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int old = __hip_atomic_fetch_add(&arrivals[y * ntiles + tile], 1, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
            last_flag = old == nparts - 1;
            if (last_flag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (!last_flag) return;
        const unsigned long long* base = partials + (int64_t)slot0 * (6 * TCELLS);
        for (int i = threadIdx.x; i < 6 * TCELLS; i += TW * 64) {
            unsigned long long sum = 0ull;
            for (int p2 = 0; p2 < nparts; ++p2) sum += base[(int64_t)p2 * slot_stride * (6 * TCELLS) + i];
            sh.acc[acc_idx(i / TCELLS, i % TCELLS)] = sum;
        }
        __syncthreads();
    }

    // fused normalise (hv_cuda_kernel.cu:112-117) + single store of the tile.  The weight the
    // reference divides by is the fp32 grid value, so round the double sum to float first.
    const int nx = min(TX, X - x0), nz = min(TZ, Z - z0);
    if (VARIANT == 5) {                       // ablation: no normalise / store
        if (threadIdx.x == 0) g_obj[((int64_t)x0 * Y + y) * Z + z0] = (float)acc_value(sh.acc[0], 0);
        return;
    }
    for (int i = threadIdx.x; i < TCELLS; i += TW * 64) {
        const int lx = i / TZ, lz = i % TZ;
        if (lx < nx && lz < nz)
            g_obj[((int64_t)(x0 + lx) * Y + y) * Z + z0 + lz] = (float)acc_value(sh.acc[acc_idx(0, i)], 0);
    }
    for (int i = threadIdx.x; i < TCELLS * 2; i += TW * 64) {
        const int cell = i >> 1, j = i & 1;
        const int lx = cell / TZ, lz = cell % TZ;
        if (lx < nx && lz < nz) {
            const double d = (double)(float)acc_value(sh.acc[acc_idx(0, cell)], 0) + 1e-7;
            g_rot[(((int64_t)(x0 + lx) * Y + y) * Z + z0 + lz) * 2 + j] =
                (float)((double)(float)fx_value(sh.acc[acc_idx(1 + j, cell)]) / d);
        }
    }
    for (int i = threadIdx.x; i < TCELLS * 3; i += TW * 64) {
        const int cell = i / 3, j = i - cell * 3;
        const int lx = cell / TZ, lz = cell % TZ;
        if (lx < nx && lz < nz) {
            const double d = (double)(float)acc_value(sh.acc[acc_idx(0, cell)], 0) + 1e-7;
            g_scale[(((int64_t)(x0 + lx) * Y + y) * Z + z0 + lz) * 3 + j] =
                (float)((double)(float)fx_value(sh.acc[acc_idx(3 + j, cell)]) / d);
        }
    }
    HV_TICK(7);
    if (VARIANT == 4 && lane == 0) {
        for (int p2 = 0; p2 < 8; ++p2) atomicAdd(&prof[p2], pacc[p2]);
        atomicAdd(&prof[8], 1ull);
    }
#undef HV_TICK
}

// ---------------------------------------------------------------------------
// (Round 3, measured and removed - git 5179aa0 holds the code: a ROLLING tile kernel, one workgroup per tile and RANGE of
// planes with two planes in LDS, every vote expanded once and accumulated into both of its planes, plane s stored after
// bin s.  Exact (all vote / decode / concurrency tests), half the cull and expansion work - and 0.56-0.85 ms against
// 0.37-0.44 ms for hv_fwd_tiles in every tile width / waves / range-count configuration, fastest with the MOST ranges
// (profiles/r3/vote_roll_sweep.txt): the op is bound by how many latency chains run side by side, not by the work in
// them; a serial bin loop per workgroup takes parallelism away.)
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// backward (hv_cuda_kernel.cu:168-261): one wave per point, lanes over rotations,
// butterfly reduction (deterministic; no atomics, like the reference).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hv_bwd(
    const float* __restrict__ grad, const float* __restrict__ pts, const float* __restrict__ xyz,
    const float* __restrict__ scl, const float* __restrict__ obj, int64_t n, int R, float res,
    F3 corner, I3 dims, const float2* __restrict__ tab, float* __restrict__ d_xyz,
    float* __restrict__ d_scl, float* __restrict__ d_obj) {
    const int lane = threadIdx.x & 63;
    const int64_t c = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (c >= n) return;
    const int Y = dims.y, Z = dims.z;
    const float s0 = scl[c * 3 + 0], s1 = scl[c * 3 + 1], s2 = scl[c * 3 + 2];
    const float x0 = xyz[c * 3 + 0], x1 = xyz[c * 3 + 1], x2 = xyz[c * 3 + 2];
    const float cx = x0 * s0, cy = x1 * s1, cz = x2 * s2;
    const float px = pts[c * 3 + 0], py = pts[c * 3 + 1], pz = pts[c * 3 + 2];
    const float ob = obj[c];
    float a_obj = 0, a_x0 = 0, a_x1 = 0, a_x2 = 0, a_s0 = 0, a_s1 = 0, a_s2 = 0;
    for (int i = lane; i < R; i += 64) {
        const float2 cs = tab[i];
        const float ox = (-cs.x) * cx + cs.y * cz;
        const float oy = -cy;
        const float oz = (-cs.y) * cx - cs.x * cz;
        const float gx = grid_pos(px, ox, corner.x, res);
        const float gy = grid_pos(py, oy, corner.y, res);
        const float gz = grid_pos(pz, oz, corner.z, res);
        if (gx < 0 || gy < 0 || gz < 0 || gx >= (float)(dims.x - 1) || gy >= (float)(dims.y - 1) ||
            gz >= (float)(dims.z - 1))
            continue;
        const int fx = (int)gx, fy = (int)gy, fz = (int)gz;
        const float rx = gx - floorf(gx), ry = gy - floorf(gy), rz = gz - floorf(gz);
        const float w0x = 1.f - rx, w0y = 1.f - ry, w0z = 1.f - rz, w1x = rx, w1y = ry, w1z = rz;
        const int64_t b = ((int64_t)fx * Y + fy) * Z + fz;
        const int64_t sx = (int64_t)Y * Z, sy = Z;
        const float lll = grad[b], llh = grad[b + 1], lhl = grad[b + sy], lhh = grad[b + sy + 1];
        const float hll = grad[b + sx], hlh = grad[b + sx + 1], hhl = grad[b + sx + sy],
                    hhh = grad[b + sx + sy + 1];
        float dob = lll * w0x * w0y * w0z;
        dob += llh * w0x * w0y * w1z;
        dob += lhl * w0x * w1y * w0z;
        dob += lhh * w0x * w1y * w1z;
        dob += hll * w1x * w0y * w0z;
        dob += hlh * w1x * w0y * w1z;
        dob += hhl * w1x * w1y * w0z;
        dob += hhh * w1x * w1y * w1z;
        a_obj += dob;
        float dx = -lll * w0y * w0z;
        dx = dx - llh * w0y * w1z; dx = dx - lhl * w1y * w0z; dx = dx - lhh * w1y * w1z;
        dx = dx + hll * w0y * w0z; dx = dx + hlh * w0y * w1z; dx = dx + hhl * w1y * w0z;
        dx = dx + hhh * w1y * w1z;
        float dy = -lll * w0x * w0z;
        dy = dy - llh * w0x * w1z; dy = dy + lhl * w0x * w0z; dy = dy + lhh * w0x * w1z;
        dy = dy - hll * w1x * w0z; dy = dy - hlh * w1x * w1z; dy = dy + hhl * w1x * w0z;
        dy = dy + hhh * w1x * w1z;
        float dz = -lll * w0x * w0y;
        dz = dz + llh * w0x * w0y; dz = dz - lhl * w0x * w1y; dz = dz + lhh * w0x * w1y;
        dz = dz - hll * w1x * w0y; dz = dz + hlh * w1x * w0y; dz = dz - hhl * w1x * w1y;
        dz = dz + hhh * w1x * w1y;
        dx *= ob; dy *= ob; dz *= ob;
        const float dcx = (-cs.x) * dx - cs.y * dz;
        const float dcy = -dy;
        const float dcz = cs.y * dx - cs.x * dz;
        a_x0 += dcx * s0; a_x1 += dcy * s1; a_x2 += dcz * s2;
        a_s0 += dcx * x0; a_s1 += dcy * x1; a_s2 += dcz * x2;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a_obj += __shfl_xor(a_obj, off);
        a_x0 += __shfl_xor(a_x0, off); a_x1 += __shfl_xor(a_x1, off); a_x2 += __shfl_xor(a_x2, off);
        a_s0 += __shfl_xor(a_s0, off); a_s1 += __shfl_xor(a_s1, off); a_s2 += __shfl_xor(a_s2, off);
    }
    if (lane == 0) {
        d_obj[c] = a_obj;
        d_xyz[c * 3 + 0] = a_x0; d_xyz[c * 3 + 1] = a_x1; d_xyz[c * 3 + 2] = a_x2;
        d_scl[c * 3 + 0] = a_s0; d_scl[c * 3 + 1] = a_s1; d_scl[c * 3 + 2] = a_s2;
    }
}

__global__ __launch_bounds__(256) void hv_count_votes(
    const float* __restrict__ pts, const float* __restrict__ xyz, const float* __restrict__ scl,
    int64_t n, int R, float res, F3 corner, I3 dims, const float2* __restrict__ tab,
    unsigned long long* __restrict__ count) {
    unsigned local = 0;
    for (int64_t t = blockIdx.x * 256ll + threadIdx.x; t < n * R; t += (int64_t)gridDim.x * 256) {
        const int64_t c = t / R;
        const int i = (int)(t - c * R);
        const float cx = xyz[c * 3 + 0] * scl[c * 3 + 0], cy = xyz[c * 3 + 1] * scl[c * 3 + 1],
                    cz = xyz[c * 3 + 2] * scl[c * 3 + 2];
        const float2 cs = tab[i];
        const float ox = (-cs.x) * cx + cs.y * cz, oy = -cy, oz = (-cs.y) * cx - cs.x * cz;
        const float gx = grid_pos(pts[c * 3 + 0], ox, corner.x, res);
        const float gy = grid_pos(pts[c * 3 + 1], oy, corner.y, res);
        const float gz = grid_pos(pts[c * 3 + 2], oz, corner.z, res);
        local += !(gx < 0 || gy < 0 || gz < 0 || gx >= (float)(dims.x - 1) ||
                   gy >= (float)(dims.y - 1) || gz >= (float)(dims.z - 1));
    }
    __shared__ unsigned s[256];
    s[threadIdx.x] = local;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0 && s[0]) atomicAdd(count, (unsigned long long)s[0]);
}

int check_common(const void* a, const void* b, const void* c, int64_t n, float res, int num_rots,
                 const float* corner, const int* dims) {
    CV_REQUIRE(a && b && c && corner && dims, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n > 0, CV_EINVAL, "n must be positive (got %lld)", (long long)n);
    CV_REQUIRE(res > 0.f, CV_EINVAL, "res must be positive");
    CV_REQUIRE(num_rots > 0 && num_rots <= 4096, CV_EINVAL, "num_rots out of range (%d)", num_rots);
    CV_REQUIRE(dims[0] > 0 && dims[1] > 0 && dims[2] > 0, CV_EINVAL, "bad grid dims");
    CV_REQUIRE((int64_t)dims[0] * dims[1] * dims[2] < (1ll << 31), CV_EINVAL, "grid too large");
    CV_REQUIRE(n * (int64_t)num_rots < (1ll << 40), CV_EINVAL, "n*num_rots too large");
    return CV_OK;
}

// streaming launch (small grids): sum over planes of parts per tile: sum_y ceil(n2_y / PART_RECORDS) <= Y + 2n / PART_RECORDS
int64_t tiles_q_bound(int64_t n, int Y) { return (int64_t)Y + (2 * n + PART_RECORDS - 1) / PART_RECORDS; }
// the launch shape: work lists + work queue where a plane has many tiles (measured, bench vote stage: 300k-point scene,
// 200 tiles: 2.01 -> 1.32 ms; 80k scene, 66 tiles: 0.37 -> 0.40 ms - the weight pass and the queue build cost more than
// the tile kernel gains there), the streaming launch below that.  CV_HV_LISTS=1 / 2 forces one of them.
#ifndef HV_QUEUE_MIN_TILES
#define HV_QUEUE_MIN_TILES 128      // 16-wide tiles: an 80k-point grid has 66 (streaming launch), a 300k-point grid 190 (work lists)
#endif
bool use_queue(int64_t ntiles) {
    static const int lists_env = getenv("CV_HV_LISTS") ? atoi(getenv("CV_HV_LISTS")) : -1;
    return lists_env >= 1 ? lists_env == 2 : ntiles >= HV_QUEUE_MIN_TILES;
}
// work queue: partial-tile slots for the (plane, tile) pairs with more than one part (sum of arc lengths <= about
// 2 * n * num_rots counting both planes of a vote and the slack steps; twice that again as room) and items = one per
// (plane, tile) + the extra parts; hv_build_queue falls back to one part per tile if either is exceeded
int64_t queue_max_slots(int64_t n, int num_rots) { return 4 * n * (int64_t)num_rots / PART_VOTES + 256; }
int64_t queue_max_items(int64_t n, int num_rots, int64_t Y, int64_t ntiles) { return Y * ntiles + queue_max_slots(n, num_rots); }
// capacity of the work-list array: 40 entries per point on average (a ring of 1.5 m radius crosses ~20 tiles) and never
// more than every (point, tile) pair; beyond it the tile kernel streams the bins as before
int64_t list_capacity(int64_t n, int64_t ntiles) { return std::min<int64_t>(n * ntiles, 40 * n + 65536); }

int pick_algo(int algo, int64_t n, int num_rots, const int* dims) {
    if (algo >= 21 && algo <= 25) return 2;   // profiling ablations of the tiles kernel
    if (algo == 1 || algo == 2) return algo;
    if (num_rots <= MAX_R_TILES && n < (1ll << 31)) return 2;
    return 1;
}

}  // namespace

extern "C" {

size_t cv_hv_minmax_workspace_bytes(void) { return 256 + sizeof(float) * 6 * (MM_BLOCKS + 1) + 256; }

int cv_hv_minmax_f32(const float* d_points, int64_t n, float* h_min3, float* h_max3, void* d_ws,
                     size_t ws_bytes, void* stream) {
    CV_REQUIRE(d_points && h_min3 && h_max3 && d_ws, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n > 0, CV_EINVAL, "n must be positive (got %lld)", (long long)n);
    CV_REQUIRE(ws_bytes >= cv_hv_minmax_workspace_bytes(), CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    CvCarver cv(d_ws);
    float* part = cv.take<float>(6 * MM_BLOCKS);
    float* out = cv.take<float>(6);
    const int blocks = (int)std::min<int64_t>(MM_BLOCKS, (n + 255) / 256);
    minmax_partial<<<blocks, 256, 0, st>>>(d_points, n, part);
    CV_LAUNCH_CHECK();
    minmax_final<<<1, 256, 0, st>>>(part, blocks, out);
    CV_LAUNCH_CHECK();
    float h[6];
    CV_HIP_CHECK(hipMemcpyAsync(h, out, sizeof h, hipMemcpyDeviceToHost, st));
    CV_HIP_CHECK(hipStreamSynchronize(st));
    for (int k = 0; k < 3; ++k) { h_min3[k] = h[k]; h_max3[k] = h[3 + k]; }
    return CV_OK;
}

// Same reduction without the host wait: h_minmax6 (PINNED host memory: min xyz, max xyz) is valid once the
// work enqueued on `stream` up to this call has completed (record an event after it).  Lets a pipeline start
// the bounds reduction of a scene before its network forward instead of stalling in front of the vote.
int cv_hv_minmax_async_f32(const float* d_points, int64_t n, float* h_minmax6, void* d_ws, size_t ws_bytes,
                           void* stream) {
    return cv_hv_minmax_async_ex(d_points, n, h_minmax6, d_ws, ws_bytes, nullptr, nullptr, stream);
}

}  // extern "C"

// (C++ linkage, cv_common.h) d_zero_word / d_fill7f: one word set to 0 and eight words set to 0x7f7f7f7f by the final launch
int cv_hv_minmax_async_ex(const float* d_points, int64_t n, float* h_minmax6, void* d_ws, size_t ws_bytes, int32_t* d_zero_word,
                          int32_t* d_fill7f, void* stream) {
    CV_REQUIRE(d_points && h_minmax6 && d_ws, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n > 0, CV_EINVAL, "n must be positive (got %lld)", (long long)n);
    CV_REQUIRE(ws_bytes >= cv_hv_minmax_workspace_bytes(), CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    CvCarver cv(d_ws);
    float* part = cv.take<float>(6 * MM_BLOCKS);
    float* out = cv.take<float>(6);
    const int blocks = (int)std::min<int64_t>(MM_BLOCKS, (n + 255) / 256);
    minmax_partial<<<blocks, 256, 0, st>>>(d_points, n, part);
    CV_LAUNCH_CHECK();
    minmax_final<<<1, 256, 0, st>>>(part, blocks, out, d_zero_word, d_fill7f);
    CV_LAUNCH_CHECK();
    CV_HIP_CHECK(hipMemcpyAsync(h_minmax6, out, 6 * sizeof(float), hipMemcpyDeviceToHost, st));
    return CV_OK;
}

extern "C" {

int cv_hv_grid_dims_f32(const float h_min3[3], const float h_max3[3], float res, int dims_out[3]) {
    CV_REQUIRE(h_min3 && h_max3 && dims_out, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(res > 0.f, CV_EINVAL, "res must be positive");
    for (int k = 0; k < 3; ++k) {
        volatile float d = h_max3[k] - h_min3[k];   // hv_cuda_kernel.cu:131 (fp32 tensor ops)
        volatile float q = d / res;
        dims_out[k] = (int)q + 1;                   // :132 .item().to<int>() + 1
    }
    return CV_OK;
}

// work queue + work lists (large grids) or the streaming launch: one decision for the workspace size and the launch
static bool tiles_queue_mode(int64_t n, int num_rots, int64_t Y, int64_t ntiles, int algo) {
    return use_queue(ntiles) && ntiles <= LIST_MAX_TILES && list_capacity(n, ntiles) < (1ll << 31) &&
           queue_max_items(n, num_rots, Y, ntiles) < (1ll << 31) && algo != 23;
}

// The workspace follows the launch shape actually chosen: the streaming launch (80k-point scenes) carries neither the
// work lists nor the queue items (they were > 60 MB of every stream's 100 MB vote scratch until round 4).
size_t cv_hv_forward_workspace_bytes(int64_t n, int num_rots, const int dims[3], int algo) {
    if (!dims || n <= 0) return 0;
    if (pick_algo(algo, n, num_rots, dims) == 1) return 256;
    const size_t Y = (size_t)dims[1];
    const size_t ntiles = (size_t)((dims[0] + TX - 1) / TX) * (size_t)((dims[2] + TZ - 1) / TZ);
    const bool queue = tiles_queue_mode(n, num_rots, (int64_t)Y, (int64_t)ntiles, algo);
    const size_t max_chunks = Y + (size_t)((n + LIST_CHUNK_RECORDS - 1) / LIST_CHUNK_RECORDS);
    const size_t slots = queue ? (size_t)queue_max_slots(n, num_rots) : (size_t)tiles_q_bound(n, (int)Y) * ntiles;
    size_t b = 256 * 24 + sizeof(int) * ((size_t)n * (1 + REC_F) + Y * 8 + 16 + 64 + Y * ntiles + max_chunks +
                                         4 * (size_t)tiles_q_bound(n, (int)Y)) +
               sizeof(unsigned long long) * slots * 6 * TCELLS;
    if (queue)
        b += sizeof(int) * (3 * Y * ntiles + max_chunks * ntiles) +
             sizeof(int4) * (size_t)queue_max_items(n, num_rots, (int64_t)Y, (int64_t)ntiles) +
             sizeof(int2) * (size_t)list_capacity(n, (int64_t)ntiles);
    return b;
}

// cv_hv_set_kernel_events: the calling thread's next cv_hv_forward_f32 calls record these events directly before and
// after the accumulation kernel (hv_fwd_tiles), on the stream of the call
static thread_local hipEvent_t t_ev_start = nullptr, t_ev_stop = nullptr;

// records of a plane's two y-bins one workgroup of a hot (tile, plane) takes in the streaming launch: PART_RECORDS (the workspace
// bound assumes at least that many) or more - with several scenes in flight fewer, longer parts pay (less merge traffic, the other
// scenes fill the chip): profiles/r5/vote_parts.txt
static std::atomic<long long> g_part_records{getenv("CV_HV_PART_RECORDS") ? std::max<long long>(PART_RECORDS, atoll(getenv("CV_HV_PART_RECORDS")))
                                                                          : (long long)PART_RECORDS};

static thread_local long long t_part_records = 0;     // cv_hv_set_part_records_thread: this thread's launches (0 = process-wide)

int cv_hv_set_part_records(int records) {
    const long long v = records <= 0 ? (long long)PART_RECORDS : std::max<long long>(PART_RECORDS, records);
    return (int)g_part_records.exchange(v, std::memory_order_relaxed);
}

int cv_hv_set_part_records_thread(int records) {
    const int before = (int)t_part_records;
    t_part_records = records <= 0 ? 0 : std::max<long long>(PART_RECORDS, records);
    return before;
}

int cv_hv_set_kernel_events(void* ev_start, void* ev_stop) {
    t_ev_start = static_cast<hipEvent_t>(ev_start);
    t_ev_stop = static_cast<hipEvent_t>(ev_stop);
    return CV_OK;
}

int cv_hv_forward_f32(const float* d_points, const float* d_xyz, const float* d_scale,
                      const float* d_obj, int64_t n, float res, int num_rots,
                      const float h_corner3[3], const int dims[3], float* d_grid_obj,
                      float* d_grid_rot, float* d_grid_scale, void* d_ws, size_t ws_bytes, int algo,
                      void* stream) {
    int rc = check_common(d_points, d_xyz, d_scale, n, res, num_rots, h_corner3, dims);
    if (rc) return rc;
    CV_REQUIRE(d_obj && d_grid_obj && d_grid_rot && d_grid_scale, CV_EINVAL, "null pointer argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float2* tab = nullptr;
    rc = get_rot_table(num_rots, &tab);
    if (rc) return rc;
    const F3 corner{h_corner3[0], h_corner3[1], h_corner3[2]};
    const I3 d3{dims[0], dims[1], dims[2]};
    const int64_t cells = (int64_t)dims[0] * dims[1] * dims[2];
    const int a = pick_algo(algo, n, num_rots, dims);
    if (a == 2) CV_REQUIRE(num_rots <= MAX_R_TILES, CV_EINVAL, "tiles algorithm needs num_rots <= %d", MAX_R_TILES);
    if (a == 1) {
        CV_HIP_CHECK(hipMemsetAsync(d_grid_obj, 0, sizeof(float) * cells, st));
        CV_HIP_CHECK(hipMemsetAsync(d_grid_rot, 0, sizeof(float) * cells * 2, st));
        CV_HIP_CHECK(hipMemsetAsync(d_grid_scale, 0, sizeof(float) * cells * 3, st));
        const int64_t total = n * num_rots;
        hv_fwd_direct<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
            d_points, d_xyz, d_scale, d_obj, n, num_rots, res, corner, d3, tab, d_grid_obj,
            d_grid_rot, d_grid_scale);
        CV_LAUNCH_CHECK();
        hv_normalise<<<(unsigned)((cells + 255) / 256), 256, 0, st>>>(d_grid_obj, d_grid_rot,
                                                                     d_grid_scale, cells);
        CV_LAUNCH_CHECK();
        return CV_OK;
    }
    // (sized by the SAME algo value that decides the launch shape below: ablation algo 23 takes the streaming launch on
    // grids where algo 2 would take the work queue, and the two carve different workspaces)
    CV_REQUIRE(d_ws && ws_bytes >= cv_hv_forward_workspace_bytes(n, num_rots, dims, algo == 23 ? 23 : 2), CV_ENOMEM,
               "workspace too small for the tiles algorithm");
    const int Y = dims[1];
    CvCarver cv(d_ws);
    int* fy = cv.take<int>(n);
    float* rec = cv.take<float>((size_t)n * REC_F);
    const int tiles_x = (dims[0] + TX - 1) / TX, tiles_z = (dims[2] + TZ - 1) / TZ;
    const int ntiles = tiles_x * tiles_z;
    const int64_t max_items = queue_max_items(n, num_rots, Y, ntiles), max_slots = queue_max_slots(n, num_rots);
    const int64_t max_q = tiles_q_bound(n, Y);
    const int64_t list_cap = list_capacity(n, ntiles);
    const bool queue = tiles_queue_mode(n, num_rots, Y, ntiles, algo);
    // the zero-initialised arrays sit next to each other: one fill launch.  The streaming launch takes none of the list /
    // queue arrays (zero-length carves: nothing of it reads them)
    int* list_ctl = cv.take<int>(64);
    int* list_cnt = cv.take<int>(queue ? (size_t)Y * ntiles : 0);      // (zeroed with the rest: the count pass adds to them)
    int* tile_w = cv.take<int>(queue ? (size_t)Y * ntiles : 0);
    int* ycount = cv.take<int>(Y);
    int* arrivals = cv.take<int>((size_t)Y * ntiles);
    int* ystart = cv.take<int>(Y + 1);
    int* cursor = cv.take<int>(Y);
    int* part_start = cv.take<int>(Y + 1);
    int4* q_info = cv.take<int4>((size_t)max_q);
    unsigned long long* partials = cv.take<unsigned long long>((size_t)(queue ? max_slots : max_q * ntiles) * 6 * TCELLS);
    int4* items = cv.take<int4>(queue ? (size_t)max_items : 0);
    int* list_start = cv.take<int>(queue ? (size_t)Y * ntiles : 0);
    const int64_t max_chunks = (int64_t)Y + (n + LIST_CHUNK_RECORDS - 1) / LIST_CHUNK_RECORDS;
    int* chunk_start = cv.take<int>((size_t)Y + 1);
    int* bin_of_chunk = cv.take<int>((size_t)max_chunks);
    int* chunk_off = cv.take<int>(queue ? (size_t)max_chunks * ntiles : 0);
    int2* entries = cv.take<int2>(queue ? (size_t)list_cap : 0);
    const int list_mode = queue ? 2 : 1;
    // (the streaming launch only needs ycount and the arrival counters zeroed)
    int* zero_from = queue ? list_ctl : ycount;
    CV_HIP_CHECK(hipMemsetAsync(zero_from, 0, (size_t)(reinterpret_cast<char*>(arrivals + (size_t)Y * ntiles) -
                                                       reinterpret_cast<char*>(zero_from)), st));
    hv_prep_count<<<(unsigned)((n + PREP_THREADS - 1) / PREP_THREADS), PREP_THREADS, 0, st>>>(
        d_points, d_xyz, d_scale, n, res, corner.y, Y, fy, ycount);
    CV_LAUNCH_CHECK();
    hv_prep_scan<<<1, 256, 0, st>>>(ycount, Y, ystart, cursor, part_start, q_info, (int)max_q, chunk_start, bin_of_chunk,
                                    (int)(t_part_records > 0 ? t_part_records : g_part_records.load(std::memory_order_relaxed)));
    CV_LAUNCH_CHECK();
    hv_prep_scatter<<<(unsigned)((n + PREP_THREADS - 1) / PREP_THREADS), PREP_THREADS, 0, st>>>(
        d_points, d_xyz, d_scale, d_obj, fy, n, Y, res, corner, cursor, rec, n);
    CV_LAUNCH_CHECK();
    if (queue) {
        hv_list_pass<false><<<(unsigned)max_chunks, 1024, 0, st>>>(ystart, chunk_start, bin_of_chunk, Y, rec, n, num_rots, tiles_x,
                                                               tiles_z, list_ctl, list_cnt, list_start, chunk_off, entries, tile_w, 1);
        CV_LAUNCH_CHECK();
        hv_build_queue<<<1, 1024, 0, st>>>(tile_w, Y, ntiles, (int)max_items, (int)max_slots, list_ctl, items);
        CV_LAUNCH_CHECK();
        hv_list_scan<<<1, 1024, 0, st>>>(list_cnt, Y, ntiles, list_cap, list_ctl, list_start);
        CV_LAUNCH_CHECK();
        hv_list_pass<true><<<(unsigned)max_chunks, 1024, 0, st>>>(ystart, chunk_start, bin_of_chunk, Y, rec, n, num_rots, tiles_x,
                                                              tiles_z, list_ctl, list_cnt, list_start, chunk_off, entries, tile_w, 1);
        CV_LAUNCH_CHECK();
    }
    const int64_t wgs = queue ? max_items : max_q * ntiles;
    CV_REQUIRE(wgs < (1ll << 31), CV_EINVAL, "grid too large");
#define CV_TILES_ARGS num_rots, res, corner, d3, tab, ystart, items, rec, n, tiles_x, tiles_z, partials, arrivals, d_grid_obj, \
                      d_grid_rot, d_grid_scale, prof, list_ctl, list_start, list_cnt, entries, list_mode, q_info
#define CV_TILES_LAUNCH(V)                                                                                   \
    do {                                                                                                     \
        if (t_ev_start) CV_HIP_CHECK(hipEventRecord(t_ev_start, st));                                         \
        if (queue) hv_fwd_tiles<V, true><<<(unsigned)wgs, TW * 64, 0, st>>>(CV_TILES_ARGS);                   \
        else hv_fwd_tiles<V, false><<<(unsigned)wgs, TW * 64, 0, st>>>(CV_TILES_ARGS);                        \
        if (t_ev_stop) CV_HIP_CHECK(hipEventRecord(t_ev_stop, st));                                           \
    } while (0)
    unsigned long long* prof = nullptr;
    if (algo == 24) {
        static unsigned long long* d_prof = nullptr;
        if (!d_prof) CV_HIP_CHECK(hipMalloc(&d_prof, 16 * sizeof(unsigned long long)));
        CV_HIP_CHECK(hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), st));
        prof = d_prof;
        CV_TILES_LAUNCH(4);
        unsigned long long h[16];
        CV_HIP_CHECK(hipMemcpyAsync(h, d_prof, sizeof h, hipMemcpyDeviceToHost, st));
        CV_HIP_CHECK(hipStreamSynchronize(st));
        static const char* names[8] = {"init", "stream+cull", "scan+pq+arc-search", "end-wait", "walk", "drain",
                                       "tail-flush", "merge+store"};
        double tot = 0;
        for (int p2 = 0; p2 < 8; ++p2) tot += (double)h[p2];
        fprintf(stderr, "hv_fwd_tiles waves %llu ticks/wave %.0f:", h[8], tot / (double)std::max(1ull, h[8]));
        for (int p2 = 0; p2 < 8; ++p2) fprintf(stderr, " %s %.1f%%", names[p2], 100.0 * (double)h[p2] / tot);
        fprintf(stderr, "\n");
    } else
    if (algo == 25) CV_TILES_LAUNCH(5);
    else if (algo == 21) CV_TILES_LAUNCH(1);
    else if (algo == 22) CV_TILES_LAUNCH(2);
    else if (algo == 23) CV_TILES_LAUNCH(3);
    else CV_TILES_LAUNCH(0);
#undef CV_TILES_LAUNCH
#undef CV_TILES_ARGS
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_hv_backward_f32(const float* d_grad_obj, const float* d_points, const float* d_xyz,
                       const float* d_scale, const float* d_obj, int64_t n, float res, int num_rots,
                       const float h_corner3[3], const int dims[3], float* d_dxyz, float* d_dscale,
                       float* d_dobj, void* stream) {
    int rc = check_common(d_points, d_xyz, d_scale, n, res, num_rots, h_corner3, dims);
    if (rc) return rc;
    CV_REQUIRE(d_grad_obj && d_obj && d_dxyz && d_dscale && d_dobj, CV_EINVAL, "null pointer argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float2* tab = nullptr;
    rc = get_rot_table(num_rots, &tab);
    if (rc) return rc;
    const F3 corner{h_corner3[0], h_corner3[1], h_corner3[2]};
    const I3 d3{dims[0], dims[1], dims[2]};
    hv_bwd<<<(unsigned)((n + 3) / 4), 256, 0, st>>>(d_grad_obj, d_points, d_xyz, d_scale, d_obj, n,
                                                   num_rots, res, corner, d3, tab, d_dxyz, d_dscale,
                                                   d_dobj);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_hv_count_votes_f32(const float* d_points, const float* d_xyz, const float* d_scale, int64_t n,
                          float res, int num_rots, const float h_corner3[3], const int dims[3],
                          int64_t* h_count, void* d_ws, size_t ws_bytes, void* stream) {
    int rc = check_common(d_points, d_xyz, d_scale, n, res, num_rots, h_corner3, dims);
    if (rc) return rc;
    CV_REQUIRE(h_count && d_ws && ws_bytes >= 256, CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float2* tab = nullptr;
    rc = get_rot_table(num_rots, &tab);
    if (rc) return rc;
    unsigned long long* cnt = static_cast<unsigned long long*>(d_ws);
    CV_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(*cnt), st));
    const F3 corner{h_corner3[0], h_corner3[1], h_corner3[2]};
    const I3 d3{dims[0], dims[1], dims[2]};
    const int64_t total = n * num_rots;
    hv_count_votes<<<(unsigned)std::min<int64_t>((total + 255) / 256, 2048), 256, 0, st>>>(
        d_points, d_xyz, d_scale, n, num_rots, res, corner, d3, tab, cnt);
    CV_LAUNCH_CHECK();
    unsigned long long h = 0;
    CV_HIP_CHECK(hipMemcpyAsync(&h, cnt, sizeof h, hipMemcpyDeviceToHost, st));
    CV_HIP_CHECK(hipStreamSynchronize(st));
    *h_count = (int64_t)h;
    return CV_OK;
}

}  // extern "C"
