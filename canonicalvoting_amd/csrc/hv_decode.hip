// Detection decode for gfx950: greedy peak picking, grid suppression, LCC-aware
// back-projection check, class vote.
//
// Replaces the inline host loop eval_joint.py:195-263 (one full-grid argmax and >= 6
// device->host syncs per candidate) with three launches and ONE sync per scene:
//
//   dec_compact      cells with grid_obj >= thresh_high -> compact list (index, value, cell
//                    coordinates) PLUS the box geometry each cell would have as a candidate
//                    (rotation from the rot grid :213-214, scale :216): the double-precision
//                    atan2 / cos / sin of every listed cell are evaluated here, in parallel,
//                    instead of one at a time on the greedy loop's critical path.
//                    The loop stops when the maximum drops below thresh_high (:208-209)
//                    and only ever writes zeros, so cells below the threshold can never
//                    influence which candidates are examined.
//   dec_greedy       one workgroup walks the list (kept in LDS): ONE pass and ONE barrier per
//                    candidate - the pass applies the suppression of the current candidate (the
//                    +-elimination cube :211 and the cells inside the oriented box :225-229,:243)
//                    to each entry and, for the survivors, accumulates the next argmax (ties ->
//                    lowest flat index, like torch.argmax).  The candidate sequence does not
//                    depend on the back-projection verdicts (the reference `continue`s after zeroing).
//   dec_backproject  all points x all candidates in parallel (:231-250): in-box test,
//                    counts, prob-weighted LCC error, max prob, class histogram; the last
//                    workgroup to finish turns the statistics into verdicts (:246-253), class
//                    mode (:255-256) and box corners (:258), one thread per candidate, and writes
//                    them straight into pinned host memory (no copy launch).
//
// fp32 conventions shared with oracle/decode_oracle.c (see its header): double-rounded
// atan2/cos/sin, k-ordered fmaf chain for the [.,3]@[3,3] products, x*(1/res) where
// torch divides by a python scalar, double accumulation of the masked mean.
// Compiled with -ffp-contract=off.
#include "cv_common.h"

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

#ifndef DEC_PROF
#define DEC_PROF 0      // phase ticks of the greedy walk (thread 0, printed at the end)
#endif
constexpr int NCLS = 64;   // class histogram bins (class ids must be in [0, 64))

struct Cand {
    long long idx;
    int c[3];
    int clo[3], chi[3];
    float cw[3];
    float cs, sn;
    float sc[3];
};

struct Stats {
    unsigned n_in, n_mask, pmax_bits, pad;
    double err;
    unsigned hist[NCLS];
};

struct Geo {
    int X, Y, Z;
    float corner[3];
    float res;
};

// compact list, structure of arrays over list positions
struct List {
    int* idx;        // flat cell index
    float* val;      // grid_obj value (zeroed when suppressed, global fallback only)
    unsigned* xy;    // x | y << 16
    int* z;
    float* geo;      // [5][cap]: cos, sin, scale xyz of the box the cell would propose
    int64_t cap;
};

// workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access
// (vmcnt(0)): with the candidate records stored from inside the greedy loop each barrier then cost a store round
// trip to L2 (~2 us per candidate).  Nothing the loop communicates between threads goes through global memory.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

__global__ __launch_bounds__(256) void dec_compact(const float* __restrict__ g_obj,
                                                   const float* __restrict__ g_rot,
                                                   const float* __restrict__ g_scale, Geo geo,
                                                   int64_t G, float thresh, List L,
                                                   unsigned* __restrict__ list_n) {
    for (int64_t base = blockIdx.x * 256ll; base < G; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + threadIdx.x;
        float v = 0.f;
        bool hit = false;
        if (i < G) { v = g_obj[i]; hit = v >= thresh; }
        const uint64_t m = __ballot(hit);
        if (m == 0) continue;
        unsigned pos = 0;
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((unsigned long long)m) - 1;
        if (lane == leader) pos = atomicAdd(list_n, (unsigned)__popcll(m));
        pos = __shfl(pos, leader);
        if (hit) {
            const unsigned p = pos + lanes_below(m);
            const int id = (int)i;
            L.idx[p] = id;
            L.val[p] = v;
            const int z = id % geo.Z, y = (id / geo.Z) % geo.Y, x = id / (geo.Z * geo.Y);
            L.xy[p] = (unsigned)x | ((unsigned)y << 16);
            L.z[p] = z;
            const float r0 = g_rot[(int64_t)id * 2], r1 = g_rot[(int64_t)id * 2 + 1];
            const float rot = (float)atan2((double)r1, (double)r0);                            // :214
            L.geo[0 * L.cap + p] = (float)cos((double)rot);
            L.geo[1 * L.cap + p] = (float)sin((double)rot);
            for (int k = 0; k < 3; ++k) L.geo[(2 + k) * L.cap + p] = g_scale[(int64_t)id * 3 + k];   // :216
        }
    }
}

// in-box test shared by grid suppression and back-projection:
// ((d @ R) / s) strictly inside (-1,1)^3 with R = [[c,0,-s],[0,1,0],[s,0,c]].
__device__ __forceinline__ bool inv_coords(float d0, float d1, float d2, float c, float s,
                                           const float* sc, float& w0, float& w1, float& w2) {
    w0 = fmaf(d2, s, fmaf(d1, 0.f, d0 * c)) / sc[0];
    w1 = fmaf(d2, 0.f, fmaf(d1, 1.f, d0 * 0.f)) / sc[1];
    w2 = fmaf(d2, c, fmaf(d1, 0.f, d0 * (-s))) / sc[2];
    return -1 < w0 && w0 < 1 && -1 < w1 && w1 < 1 && -1 < w2 && w2 < 1;
}

// the same test as a boolean without the three divisions (the greedy walk only needs "inside or not"): for finite
// operands -1 < RN(t / s) < 1 holds exactly when |t| < |s| - a quotient below one cannot round up to one (t <= |s| - ulp
// gives t / |s| <= 1 - 2^-24, representable), |t| >= |s| gives a quotient >= 1; s = 0, infinities and NaNs fall on the
// same side in both forms.  Bit-identical verdicts, ~30 of the ~60 vector instructions of a kill test gone.
__device__ __forceinline__ bool inside_box(float d0, float d1, float d2, float c, float s, const float* sc) {
    const float t0 = fmaf(d2, s, fmaf(d1, 0.f, d0 * c));
    const float t1 = fmaf(d2, 0.f, fmaf(d1, 1.f, d0 * 0.f));
    const float t2 = fmaf(d2, c, fmaf(d1, 0.f, d0 * (-s)));
    return fabsf(t0) < fabsf(sc[0]) && fabsf(t1) < fabsf(sc[1]) && fabsf(t2) < fabsf(sc[2]);
}

// the eight box corners relative to the centre (:217-219): bb[q][k]
__device__ __forceinline__ void box_corners(float cs, float sn, const float* sc, float* bb) {
    // raw corner signs (:217); literals so that the unrolled loop folds them (no constant-memory loads on the
    // greedy loop's critical path)
    constexpr float kRawX[8] = {1, 1, -1, -1, 1, 1, -1, -1};
    constexpr float kRawY[8] = {1, 1, 1, 1, -1, -1, -1, -1};
    constexpr float kRawZ[8] = {1, -1, -1, 1, 1, -1, -1, 1};
    const float m00 = cs * sc[0], m02 = (-sn) * sc[2], m11 = sc[1], m20 = sn * sc[0], m22 = cs * sc[2];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        bb[q * 3 + 0] = m00 * kRawX[q] + m02 * kRawZ[q];
        bb[q * 3 + 1] = m11 * kRawY[q];
        bb[q * 3 + 2] = m20 * kRawX[q] + m22 * kRawZ[q];
    }
}

struct Best {
    float v;
    int id, pos;
};

__device__ __forceinline__ void take_better(Best& b, float v, int id, int pos) {
    if (v > b.v || (v == b.v && id < b.id)) { b.v = v; b.id = id; b.pos = pos; }
}

#ifndef DEC_SMALL_T
#define DEC_SMALL_T 512     // 256: 0.180 ms, 512: 0.158 ms, 1024: 0.181 ms decode stage at 80k points (profiles/r4/dec_small_t.txt)
#endif
constexpr int GREEDY_T = DEC_SMALL_T;        // threads of the greedy workgroup: two waves per SIMD.  One wave per SIMD (256, rounds
                                     // 2-3) runs the per-candidate part every wave repeats - reduction, candidate set-up -
                                     // once per SIMD, but leaves the LDS round trips of a region's box tests (~1 000 per
                                     // candidate at 80k points, 4 per thread) with nothing to hide behind
constexpr int GREEDY_W = GREEDY_T / 64;
constexpr int GREEDY_CAP = 4096;     // list entries held in LDS (32 bytes each)

// cross-lane moves on the DPP path (one VALU instruction, no LDS round trip): quad swaps, then row rotations leave
// every lane of a 16-lane row with the row's result; the four rows are combined through scalar registers
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int x) {
    return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ float wave_max_f32(float x) {
    x = fmaxf(x, __int_as_float(dpp_mov<0xB1>(__float_as_int(x))));     // quad_perm [1,0,3,2]
    x = fmaxf(x, __int_as_float(dpp_mov<0x4E>(__float_as_int(x))));     // quad_perm [2,3,0,1]
    x = fmaxf(x, __int_as_float(dpp_mov<0x124>(__float_as_int(x))));    // row_ror:4
    x = fmaxf(x, __int_as_float(dpp_mov<0x128>(__float_as_int(x))));    // row_ror:8
    const int xi = __float_as_int(x);
    const float a = __int_as_float(__builtin_amdgcn_readlane(xi, 0)), b = __int_as_float(__builtin_amdgcn_readlane(xi, 16)),
                c = __int_as_float(__builtin_amdgcn_readlane(xi, 32)), d = __int_as_float(__builtin_amdgcn_readlane(xi, 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ int wave_min_i32(int x) {
    x = min(x, dpp_mov<0xB1>(x));
    x = min(x, dpp_mov<0x4E>(x));
    x = min(x, dpp_mov<0x124>(x));
    x = min(x, dpp_mov<0x128>(x));
    return min(min(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)),
               min(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
}

// MODE 1: the thread's entries in registers, the rare by-index reads from an LDS copy (lists <= 16 * T = 4096 at T = 256);
// MODE 2: the same walk with the by-index reads from the global arrays (L2) - no 128 KB LDS copy, so 1024 threads take lists
//         up to 16384 cells (300k-point scenes); MODE 0: everything from the global arrays (any length)
template <int MODE, int T>
__device__ __forceinline__ void dec_greedy(Geo geo, cv_decode_params prm, List L,
                                           const unsigned* __restrict__ list_n, Cand* __restrict__ cands,
                                           Stats* __restrict__ stats, int* __restrict__ n_cand_out) {
    constexpr bool IN_REG = MODE != 0, IN_LDS = MODE == 1;
    constexpr int GREEDY_E = (MODE == 1 ? GREEDY_CAP / T : 16);      // entries per thread of the register-resident walk
    constexpr int CAP = IN_LDS ? GREEDY_E * T : 1;
    constexpr int GREEDY_T = T, GREEDY_W = T / 64;
    __shared__ unsigned l_xy[CAP];
    __shared__ int l_z[CAP];
    __shared__ float l_geo[5][CAP];
    __shared__ float s_val[2][GREEDY_W];
    __shared__ int s_idx[2][GREEDY_W], s_pos[2][GREEDY_W];
    const int n = (int)*list_n;
    float r_val[GREEDY_E];
    unsigned r_xy[GREEDY_E], dead = 0;
    int r_z[GREEDY_E], r_id[GREEDY_E];
    if (IN_REG) {
#pragma unroll
        for (int j = 0; j < GREEDY_E; ++j) {
            const int k = (int)threadIdx.x + j * GREEDY_T;
            const bool ok = k < n;
            r_val[j] = ok ? L.val[k] : 0.f;
            r_xy[j] = ok ? L.xy[k] : 0u;
            r_z[j] = ok ? L.z[k] : 0;
            r_id[j] = ((int)(r_xy[j] & 0xffffu) * geo.Y + (int)(r_xy[j] >> 16)) * geo.Z + r_z[j];
            dead |= ok ? 0u : (1u << j);
            if (ok && IN_LDS) {
                l_xy[k] = r_xy[j];
                l_z[k] = r_z[j];
                for (int q = 0; q < 5; ++q) l_geo[q][k] = L.geo[q * L.cap + k];
            }
        }
        __syncthreads();
    }
    const int wave = threadIdx.x >> 6;
    const float inv_res = 1.0f / geo.res;
    const int e = prm.elimination, hp = e + (prm.elim_hi_plus1 ? 1 : 0);
    // the candidate whose suppression the next pass applies (none before the first pass)
    bool have_cur = false;
    int cx = 0, cy = 0, cz = 0, clo0 = 0, clo1 = 0, clo2 = 0, chi0 = 0, chi1 = 0, chi2 = 0;
    // conservative region of that suppression (cube and box bounds together): one unsigned compare per axis
    // dismisses the entries that are nowhere near it
    int rlo0 = 0, rlo1 = 0, rlo2 = 0;
    unsigned rsp0 = 0, rsp1 = 0, rsp2 = 0;
    float ccs = 0.f, csn = 0.f, csc[3] = {1.f, 1.f, 1.f};
    auto flat_id = [&](int pos) {        // global fallback only
        const unsigned xy = L.xy[pos];
        const int z = L.z[pos];
        return ((int)(xy & 0xffffu) * geo.Y + (int)(xy >> 16)) * geo.Z + z;
    };
    int it = 0;
    bool truncated = false;
    for (;; ++it) {
        // ---- one pass: suppression by the current candidate (:211, :225-229, :243), then the argmax of
        //      what is left (largest value, lowest flat index on ties: eval_joint.py:205)
        float bv = -1.f;
        int bpos = -1, bid = 0x7fffffff;
        if constexpr (IN_REG) {
            // the thread's entries live in registers (r_*; loaded once): a pass is VALU work only.  The first version
            // walked them in LDS - two dependent LDS round trips per entry with one wave per SIMD and nothing to
            // overlap them with: 10k cycles per pass for 13 entries per thread.
            if (have_cur) {
                unsigned near = 0;        // alive entries inside the conservative region
#pragma unroll
                for (int j = 0; j < GREEDY_E; ++j) {
                    const int x = (int)(r_xy[j] & 0xffffu), y = (int)(r_xy[j] >> 16), z = r_z[j];
                    const bool in = (unsigned)(x - rlo0) <= rsp0 && (unsigned)(y - rlo1) <= rsp1 &&
                                    (unsigned)(z - rlo2) <= rsp2;
                    near |= in ? (1u << j) : 0u;
                }
                near &= ~dead;
                while (near) {            // rare: a handful of entries per candidate, taken from LDS by index
                    const int j = __ffs(near) - 1;
                    near &= near - 1;
                    const int k = (int)threadIdx.x + j * GREEDY_T;
                    const unsigned xy = IN_LDS ? l_xy[k] : L.xy[k];
                    const int z = IN_LDS ? l_z[k] : L.z[k];
                    const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
                    bool kill = x >= cx - e && x < cx + hp && y >= cy - e && y < cy + hp && z >= cz - e && z < cz + hp;
                    if (!kill && x >= clo0 && x <= chi0 && y >= clo1 && y <= chi1 && z >= clo2 && z <= chi2) {
                        const float v0 = (float)(x - cx) * geo.res, v1 = (float)(y - cy) * geo.res,
                                    v2 = (float)(z - cz) * geo.res;
                        kill = inside_box(v0, v1, v2, ccs, csn, csc);
                    }
                    if (kill) dead |= 1u << j;
                }
            }
#pragma unroll
            for (int j = 0; j < GREEDY_E; ++j) {
                const bool alive = !((dead >> j) & 1u);
                const float v = r_val[j];
                const int id = r_id[j];
                if (alive && (v > bv || (v == bv && id < bid))) { bv = v; bid = id; bpos = (int)threadIdx.x + j * GREEDY_T; }
            }
        } else {
            for (int k = threadIdx.x; k < n; k += GREEDY_T) {
                const float v = L.val[k];
                if (v == 0.f) continue;
                if (have_cur) {
                    const unsigned xy = L.xy[k];
                    const int z = L.z[k];
                    const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
                    if ((unsigned)(x - rlo0) <= rsp0 && (unsigned)(y - rlo1) <= rsp1 && (unsigned)(z - rlo2) <= rsp2) {
                        bool kill = x >= cx - e && x < cx + hp && y >= cy - e && y < cy + hp && z >= cz - e && z < cz + hp;
                        if (!kill && x >= clo0 && x <= chi0 && y >= clo1 && y <= chi1 && z >= clo2 && z <= chi2) {
                            const float v0 = (float)(x - cx) * geo.res, v1 = (float)(y - cy) * geo.res,
                                        v2 = (float)(z - cz) * geo.res;
                            kill = inside_box(v0, v1, v2, ccs, csn, csc);
                        }
                        if (kill) {
                            L.val[k] = 0.f;
                            continue;
                        }
                    }
                }
                if (v > bv || (v == bv && flat_id(k) < flat_id(bpos))) { bv = v; bpos = k; }
            }
            bid = bpos >= 0 ? flat_id(bpos) : 0x7fffffff;
        }
        const float wv = wave_max_f32(bv);
        const int wid = wave_min_i32(bv == wv ? bid : 0x7fffffff);
        const int par = it & 1;           // double-buffered: the one barrier also frees the other buffer
        if (bv == wv && bid == wid) { s_val[par][wave] = bv; s_idx[par][wave] = bid; s_pos[par][wave] = bpos; }
        lds_barrier();
        Best w{s_val[par][0], s_idx[par][0], s_pos[par][0]};
#pragma unroll
        for (int q = 1; q < GREEDY_W; ++q) take_better(w, s_val[par][q], s_idx[par][q], s_pos[par][q]);
        if (!(w.v >= prm.thresh_high)) break;                                   // :208-209
        if (it >= prm.max_iters) { truncated = true; break; }
        // ---- the candidate (every thread forms the same values; thread 0 records them)
        const int id = w.id;
        {
            const unsigned wxy = IN_LDS ? l_xy[w.pos] : L.xy[w.pos];
            cx = (int)(wxy & 0xffffu); cy = (int)(wxy >> 16);
            cz = IN_LDS ? l_z[w.pos] : L.z[w.pos];
        }
        ccs = IN_LDS ? l_geo[0][w.pos] : L.geo[0 * L.cap + w.pos];
        csn = IN_LDS ? l_geo[1][w.pos] : L.geo[1 * L.cap + w.pos];
        for (int k = 0; k < 3; ++k) csc[k] = IN_LDS ? l_geo[2 + k][w.pos] : L.geo[(2 + k) * L.cap + w.pos];
        // extent of the eight corners (+-m00 +- m02, +-m11, +-m20 +- m22; :217-219): every sign combination occurs and
        // rounding is symmetric, so max = |.| + |.| and min = -max, bit for bit what the min / max over the corners give
        float hi[3];
        {
            const float m00 = ccs * csc[0], m02 = (-csn) * csc[2], m11 = csc[1], m20 = csn * csc[0], m22 = ccs * csc[2];
            hi[0] = fabsf(m00) + fabsf(m02);
            hi[1] = fabsf(m11);
            hi[2] = fabsf(m20) + fabsf(m22);
        }
        const int cc[3] = {cx, cy, cz}, shape[3] = {geo.X, geo.Y, geo.Z};
        int clo[3], chi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {                                             // :220-223
            const int blo = (int)((-hi[k]) * inv_res), bhi = (int)(hi[k] * inv_res);
            clo[k] = min(max(cc[k] + blo, 0), shape[k] - 1);
            chi[k] = min(max(cc[k] + bhi, 0), shape[k] - 1);
        }
        clo0 = clo[0]; clo1 = clo[1]; clo2 = clo[2]; chi0 = chi[0]; chi1 = chi[1]; chi2 = chi[2];
        rlo0 = min(cx - e, clo0); rsp0 = (unsigned)(max(cx + hp - 1, chi0) - rlo0);
        rlo1 = min(cy - e, clo1); rsp1 = (unsigned)(max(cy + hp - 1, chi1) - rlo1);
        rlo2 = min(cz - e, clo2); rsp2 = (unsigned)(max(cz + hp - 1, chi2) - rlo2);
        have_cur = true;
        if (threadIdx.x == 0) {
            Cand cd;
            cd.idx = id;
            for (int k = 0; k < 3; ++k) {
                cd.c[k] = cc[k]; cd.clo[k] = clo[k]; cd.chi[k] = chi[k];
                cd.cw[k] = geo.corner[k] + geo.res * (float)cc[k];                // :206
                cd.sc[k] = csc[k];
            }
            cd.cs = ccs; cd.sn = csn;
            cands[it] = cd;
        }
        // the statistics of this candidate start from zero (the workspace is not cleared by a launch)
        for (int q = threadIdx.x; q < (int)(sizeof(Stats) / 4); q += GREEDY_T)
            reinterpret_cast<unsigned*>(&stats[it])[q] = 0u;
    }
    if (threadIdx.x == 0) { n_cand_out[0] = it; n_cand_out[1] = truncated ? 1 : 0; }
}

// ---- the register-resident walk, second version (round 4).  The first one (dec_greedy<1 / 2>) tests every entry of every
// thread against every candidate and re-scans them for the maximum: ~44 vector instructions per entry and pass, which is
// what a 300k-point scene's walk spent its 1.6 ms on (21 000 listed cells x 180 candidates on ONE CU; the 16-entry
// capacity was also 5 000 cells short, so that scene fell through to the walk over the global arrays).  Here
//   * a wave owns 64 * E CONSECUTIVE list entries (dec_compact appends in rounds of the grid-stride loop: a stretch of the
//     list is a window of x slabs) and every thread keeps the bounding box of its own: a candidate's suppression region is
//     tested against the box first, and a wave whose 64 boxes all miss skips its entries altogether;
//   * the entries are sorted once (value descending, flat index ascending: the order the walk takes them in), so the
//     thread's best live entry is the lowest clear bit of `dead`; value / cell / list position of that entry sit in
//     registers and are re-picked (select chains over the E slots) only when it dies;
//   * the winner's cell travels through LDS from its owner's registers - no LDS copy of the list (the 80k-point walk's 128 KB),
//     no by-index global reads except the five geometry words.
// Same candidates in the same order as the reference loop (eval_joint.py:204-263): the maximum with ties to the lowest flat
// index, suppression by cube and box exactly as before.  Used for the big grids (dec_greedy_dispatch_big): a 300k-point
// scene's decode 1.88 -> 1.28 ms.  What is left there (phase ticks, DEC_PROF): the waves that own the cells inside a
// candidate's region run ~6 kill tests per lane (~100 instructions each with the select trees) in nearly every wave - the four
// waves of a SIMD keep it busy for ~13 000 cycles per candidate - and the five geometry words cost 2 700 cycles (requesting them
// before the barrier, by every wave's best lane, did not shorten that).
// NOT used for the 80k-point lists (3 400 cells, 42 candidates): 0.21 ms against dec_greedy<1>'s 0.18.
template <int N, class Tv>
__device__ __forceinline__ Tv pick_slot(const Tv (&r)[N], int j) {
    // slot j of a register array (a dynamic index has to become selects): a binary tree over the bits of j - N - 1 selects
    // and five bit tests, which two picks of the same j share, instead of N compares + N selects each
    static_assert(N <= 32, "five index bits");
    Tv t[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i] = r[i < N ? i : N - 1];
#pragma unroll
    for (int bit = 0; bit < 5; ++bit) {
        const bool b = (j >> bit) & 1;
#pragma unroll
        for (int i = 0; i < (32 >> (bit + 1)); ++i) t[i] = b ? t[2 * i + 1] : t[2 * i];
    }
    return t[0];
}

template <int T, int E, int G>
__device__ __forceinline__ void dec_greedy_sorted(Geo geo, cv_decode_params prm, List L, int n, Cand* __restrict__ cands,
                                                  Stats* __restrict__ stats, int* __restrict__ n_cand_out) {
    static_assert(E >= 2 && E <= 32 && T % 64 == 0 && G % 64 == 0 && T % G == 0, "entries per thread live in one 32-bit mask");
    constexpr int W = T / 64;
    constexpr unsigned ALL = E == 32 ? 0xffffffffu : ((1u << E) - 1u);
    __shared__ float s_val[2][W];
    __shared__ int s_idx[2][W], s_pos[2][W], s_z[2][W];
    __shared__ unsigned s_xy[2][W];
    // a group of G threads owns G * E consecutive list entries, dealt to its threads round-robin: the cells a candidate
    // suppresses sit next to each other in the list, and a thread that owned them all would walk them alone (E consecutive
    // entries per thread: 0.18 -> 0.30 ms at 80k points; G = 64 at 300k points: the wave that owns the ~500 suppressed cells
    // takes 9 000 cycles per candidate and the other fifteen wait)
    const int base = (int)(threadIdx.x / G) * (G * E) + (int)(threadIdx.x % G);
    float r_val[E];
    unsigned r_xy[E];
    int r_zj[E];                                   // z | j << 16: list position = base + G * j
    int bx0 = 0x7fffffff, bx1 = -1, by0 = 0x7fffffff, by1 = -1, bz0 = 0x7fffffff, bz1 = -1;
    unsigned dead;
    {
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int k = base + G * j;
            const bool ok = k < n;
            r_val[j] = ok ? L.val[k] : 0.f;
            r_xy[j] = ok ? L.xy[k] : 0u;
            const int z = ok ? L.z[k] : 0;
            r_zj[j] = z | (j << 16);
            const int x = (int)(r_xy[j] & 0xffffu), y = (int)(r_xy[j] >> 16);
            if (ok) {
                bx0 = min(bx0, x); bx1 = max(bx1, x);
                by0 = min(by0, y); by1 = max(by1, y);
                bz0 = min(bz0, z); bz1 = max(bz1, z);
                ++cnt;
            }
        }
        // odd-even transposition sort of the thread's entries (static register indices): value descending, flat index
        // ascending (computed only on a tie: no register per entry for it); the slots beyond the list (value 0) end up last
#pragma unroll
        for (int r = 0; r < E; ++r) {
#pragma unroll
            for (int j = r & 1; j + 1 < E; j += 2) {
                const float v0 = r_val[j], v1 = r_val[j + 1];
                const unsigned x0 = r_xy[j], x1 = r_xy[j + 1];
                const int z0 = r_zj[j], z1 = r_zj[j + 1];
                bool sw = v1 > v0;
                if (v1 == v0) {
                    const int i0 = ((int)(x0 & 0xffffu) * geo.Y + (int)(x0 >> 16)) * geo.Z + (z0 & 0xffff);
                    const int i1 = ((int)(x1 & 0xffffu) * geo.Y + (int)(x1 >> 16)) * geo.Z + (z1 & 0xffff);
                    sw = i1 < i0;
                }
                r_val[j] = sw ? v1 : v0; r_val[j + 1] = sw ? v0 : v1;
                r_xy[j] = sw ? x1 : x0;  r_xy[j + 1] = sw ? x0 : x1;
                r_zj[j] = sw ? z1 : z0;  r_zj[j + 1] = sw ? z0 : z1;
            }
        }
        dead = ALL & ~(cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u));
    }
    const int wave = threadIdx.x >> 6;
    const float inv_res = 1.0f / geo.res;
    const int e = prm.elimination, hp = e + (prm.elim_hi_plus1 ? 1 : 0);
    bool have_cur = false;
    int cx = 0, cy = 0, cz = 0, clo0 = 0, clo1 = 0, clo2 = 0, chi0 = 0, chi1 = 0, chi2 = 0;
    int rlo0 = 0, rlo1 = 0, rlo2 = 0, rhi0 = -1, rhi1 = -1, rhi2 = -1;     // conservative region of the current suppression
    float ccs = 0.f, csn = 0.f, csc[3] = {1.f, 1.f, 1.f};
    // the thread's best live entry
    int cur_j = -1, cur_id = 0x7fffffff, cur_pos = -1, cur_z = 0;
    unsigned cur_xy = 0;
    float cur_val = -1.f;
    int it = 0;
    bool truncated = false;
#if DEC_PROF
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_readcyclecounter();
#define DEC_TICK(i) { const unsigned long long t_now = __builtin_readcyclecounter(); pt[i] += t_now - t_prev; t_prev = t_now; }
    DEC_TICK(7)
#else
#define DEC_TICK(i)
#endif
    for (;; ++it) {
        // ---- suppression by the current candidate (:211, :225-229, :243): box against box first
        const bool hit = have_cur && dead != ALL && bx1 >= rlo0 && bx0 <= rhi0 && by1 >= rlo1 && by0 <= rhi1 &&
                         bz1 >= rlo2 && bz0 <= rhi2;
        if (__any(hit)) {
            if (hit) {
                unsigned near = 0;
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const int x = (int)(r_xy[j] & 0xffffu), y = (int)(r_xy[j] >> 16), z = r_zj[j] & 0xffff;
                    const bool in = (unsigned)(x - rlo0) <= (unsigned)(rhi0 - rlo0) && (unsigned)(y - rlo1) <= (unsigned)(rhi1 - rlo1) &&
                                    (unsigned)(z - rlo2) <= (unsigned)(rhi2 - rlo2);
                    near |= in ? (1u << j) : 0u;
                }
                near &= ~dead;
                while (near) {
                    const int j = __ffs(near) - 1;
                    near &= near - 1;
                    const unsigned xy = pick_slot<E>(r_xy, j);
                    const int z = pick_slot<E>(r_zj, j) & 0xffff;
                    const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
                    bool kill = x >= cx - e && x < cx + hp && y >= cy - e && y < cy + hp && z >= cz - e && z < cz + hp;
                    if (!kill && x >= clo0 && x <= chi0 && y >= clo1 && y <= chi1 && z >= clo2 && z <= chi2) {
                        const float v0 = (float)(x - cx) * geo.res, v1 = (float)(y - cy) * geo.res,
                                    v2 = (float)(z - cz) * geo.res;
                        kill = inside_box(v0, v1, v2, ccs, csn, csc);
                    }
                    if (kill) dead |= 1u << j;
                }
            }
        }
        DEC_TICK(0)
        // ---- the thread's best live entry = the first live slot of the sorted order
        float bv = -1.f;
        int bid = 0x7fffffff, bpos = -1;
        {
            const unsigned alive = ALL & ~dead;
            if (alive != 0u) {
                if (cur_j < 0 || ((dead >> cur_j) & 1u)) {
                    cur_j = __ffs(alive) - 1;
                    cur_val = pick_slot<E>(r_val, cur_j);
                    cur_xy = pick_slot<E>(r_xy, cur_j);
                    const int zj = pick_slot<E>(r_zj, cur_j);
                    cur_z = zj & 0xffff;
                    cur_pos = base + G * (zj >> 16);
                    cur_id = ((int)(cur_xy & 0xffffu) * geo.Y + (int)(cur_xy >> 16)) * geo.Z + cur_z;
                }
                bv = cur_val; bid = cur_id; bpos = cur_pos;
            }
        }
        DEC_TICK(1)
        const float wv = wave_max_f32(bv);
        const int wid = wave_min_i32(bv == wv ? bid : 0x7fffffff);
        const int par = it & 1;           // double-buffered: the one barrier also frees the other buffer
        if (bv == wv && bid == wid) {
            s_val[par][wave] = bv; s_idx[par][wave] = bid; s_pos[par][wave] = bpos;
            s_xy[par][wave] = cur_xy; s_z[par][wave] = cur_z;
        }
        DEC_TICK(2)
        lds_barrier();
        DEC_TICK(3)
        Best w{s_val[par][0], s_idx[par][0], s_pos[par][0]};
        int wq = 0;
#pragma unroll
        for (int q = 1; q < W; ++q) {
            const float v = s_val[par][q];
            const int id = s_idx[par][q];
            if (v > w.v || (v == w.v && id < w.id)) { w.v = v; w.id = id; w.pos = s_pos[par][q]; wq = q; }
        }
        if (!(w.v >= prm.thresh_high)) break;                                   // :208-209
        if (it >= prm.max_iters) { truncated = true; break; }
        // ---- the candidate (every thread forms the same values; thread 0 records them)
        const int id = w.id;
        {
            const unsigned wxy = s_xy[par][wq];
            cx = (int)(wxy & 0xffffu); cy = (int)(wxy >> 16);
            cz = s_z[par][wq];
        }
        ccs = L.geo[0 * L.cap + w.pos];
        csn = L.geo[1 * L.cap + w.pos];
        for (int k = 0; k < 3; ++k) csc[k] = L.geo[(2 + k) * L.cap + w.pos];
        DEC_TICK(4)
        float hi[3];
        {
            const float m00 = ccs * csc[0], m02 = (-csn) * csc[2], m11 = csc[1], m20 = csn * csc[0], m22 = ccs * csc[2];
            hi[0] = fabsf(m00) + fabsf(m02);
            hi[1] = fabsf(m11);
            hi[2] = fabsf(m20) + fabsf(m22);
        }
        const int cc[3] = {cx, cy, cz}, shape[3] = {geo.X, geo.Y, geo.Z};
        int clo[3], chi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {                                             // :220-223
            const int blo = (int)((-hi[k]) * inv_res), bhi = (int)(hi[k] * inv_res);
            clo[k] = min(max(cc[k] + blo, 0), shape[k] - 1);
            chi[k] = min(max(cc[k] + bhi, 0), shape[k] - 1);
        }
        clo0 = clo[0]; clo1 = clo[1]; clo2 = clo[2]; chi0 = chi[0]; chi1 = chi[1]; chi2 = chi[2];
        rlo0 = min(cx - e, clo0); rhi0 = max(cx + hp - 1, chi0);
        rlo1 = min(cy - e, clo1); rhi1 = max(cy + hp - 1, chi1);
        rlo2 = min(cz - e, clo2); rhi2 = max(cz + hp - 1, chi2);
        have_cur = true;
        if (threadIdx.x == 0) {
            Cand cd;
            cd.idx = id;
            for (int k = 0; k < 3; ++k) {
                cd.c[k] = cc[k]; cd.clo[k] = clo[k]; cd.chi[k] = chi[k];
                cd.cw[k] = geo.corner[k] + geo.res * (float)cc[k];                // :206
                cd.sc[k] = csc[k];
            }
            cd.cs = ccs; cd.sn = csn;
            cands[it] = cd;
        }
        // the statistics of this candidate start from zero (the workspace is not cleared by a launch)
        for (int q = threadIdx.x; q < (int)(sizeof(Stats) / 4); q += T)
            reinterpret_cast<unsigned*>(&stats[it])[q] = 0u;
        DEC_TICK(5)
    }
#if DEC_PROF
    if ((threadIdx.x & 63) == 0)
        printf("  wave %2d: suppress %llu best %llu reduce+write %llu barrier %llu final+geo %llu math+stores %llu\n", (int)(threadIdx.x >> 6),
               pt[0], pt[1], pt[2], pt[3], pt[4], pt[5]);
    if (threadIdx.x == 0)
        printf("dec_greedy_sorted<%d,%d,%d> n=%d cand=%d ticks: setup %llu | suppress %llu best %llu reduce+write %llu barrier %llu final+geo %llu math+stores %llu\n",
               T, E, G, n, it, pt[7], pt[0], pt[1], pt[2], pt[3], pt[4], pt[5]);
#endif
    if (threadIdx.x == 0) { n_cand_out[0] = it; n_cand_out[1] = truncated ? 1 : 0; }
}

// the list length is only known on the device: lists up to GREEDY_CAP entries are walked in registers + LDS, longer ones by
// the same code over the global arrays (L2-resident)
__global__ __launch_bounds__(GREEDY_T) void dec_greedy_dispatch(Geo geo, cv_decode_params prm, List L,
                                                                const unsigned* __restrict__ list_n,
                                                                Cand* __restrict__ cands, Stats* __restrict__ stats,
                                                                int* __restrict__ n_cand_out) {
    if (*list_n <= (unsigned)GREEDY_CAP) dec_greedy<1, GREEDY_T>(geo, prm, L, list_n, cands, stats, n_cand_out);
    else dec_greedy<0, GREEDY_T>(geo, prm, L, list_n, cands, stats, n_cand_out);
}
// big grids (the host picks this launch from the cell count): 1024 threads, lists up to 24576 cells in registers (a
// 300k-point scene lists ~21 000 cells) on the sorted walk
constexpr int GREEDY_T_BIG = 1024;
constexpr int GREEDY_E_BIG = 24;
constexpr int GREEDY_G_BIG = 64;
__global__ __launch_bounds__(GREEDY_T_BIG) void dec_greedy_dispatch_big(Geo geo, cv_decode_params prm, List L,
                                                                        const unsigned* __restrict__ list_n,
                                                                        Cand* __restrict__ cands, Stats* __restrict__ stats,
                                                                        int* __restrict__ n_cand_out) {
    if (*list_n <= (unsigned)(GREEDY_E_BIG * GREEDY_T_BIG))
        dec_greedy_sorted<GREEDY_T_BIG, GREEDY_E_BIG, GREEDY_G_BIG>(geo, prm, L, (int)*list_n, cands, stats, n_cand_out);
    else dec_greedy<0, GREEDY_T_BIG>(geo, prm, L, list_n, cands, stats, n_cand_out);
}

// packed host result: [0]=n_cand [1]=n_boxes [2]=truncated, then arrays sized by max_iters
struct ResultLayout {
    size_t off_cand, off_verdict, off_boxes, off_scores, off_classes, total;
    __host__ __device__ explicit ResultLayout(int M) {
        size_t o = 16;
        off_cand = o; o += sizeof(long long) * M;
        off_verdict = o; o += sizeof(int) * M;
        off_boxes = o; o += sizeof(float) * 24 * M;
        off_scores = o; o += sizeof(float) * M;
        off_classes = o; o += sizeof(int) * M;
        total = cv_align_up(o, 16);
    }
};

// verdicts (:246-253), class mode (:255-256) and box corners (:258): one thread per candidate, accepted boxes
// written in candidate order (acceptance order of the sequential loop)
__device__ void finalize_block(const Cand* __restrict__ cands, const Stats* __restrict__ stats, int nc,
                               int truncated, cv_decode_params prm, char* result, int M, int* s_scan /*[256]*/) {
    const ResultLayout L(M);
    int* hdr = reinterpret_cast<int*>(result);
    long long* o_cand = reinterpret_cast<long long*>(result + L.off_cand);
    int* o_verdict = reinterpret_cast<int*>(result + L.off_verdict);
    float* o_boxes = reinterpret_cast<float*>(result + L.off_boxes);
    float* o_scores = reinterpret_cast<float*>(result + L.off_scores);
    int* o_classes = reinterpret_cast<int*>(result + L.off_classes);
    int base = 0;
    for (int k0 = 0; k0 < nc; k0 += 256) {
        const int k = k0 + (int)threadIdx.x;
        int verdict = -1, best = 0;
        if (k < nc) {
            const Stats& s = stats[k];
            const float lhs = (float)s.n_mask, rhs = prm.valid_ratio * (float)s.n_in;
            if (lhs < rhs || (float)s.n_in < prm.thresh_low) verdict = 1;               // :246-247
            else {
                const float error = (float)(s.err / (double)s.n_mask);
                if ((double)error > prm.err_thresh) verdict = 2;                         // :252-253
                else {
                    verdict = 0;
                    unsigned best_cnt = s.hist[0];
                    for (int c = 1; c < NCLS; ++c)
                        if (s.hist[c] > best_cnt) { best_cnt = s.hist[c]; best = c; }
                }
            }
            o_cand[k] = cands[k].idx;
            o_verdict[k] = verdict;
        }
        // exclusive scan of the accepted flags over the 256 threads
        s_scan[threadIdx.x] = verdict == 0 ? 1 : 0;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int add = (int)threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0;
            __syncthreads();
            s_scan[threadIdx.x] += add;
            __syncthreads();
        }
        const int incl = s_scan[threadIdx.x], total = s_scan[255];
        if (verdict == 0) {
            const int nb = base + incl - 1;
            const Cand& cd = cands[k];
            float bb[24];
            box_corners(cd.cs, cd.sn, cd.sc, bb);
            for (int q = 0; q < 8; ++q)
                for (int d = 0; d < 3; ++d) o_boxes[(size_t)nb * 24 + q * 3 + d] = bb[q * 3 + d] + cd.cw[d];  // :258
            o_scores[nb] = __uint_as_float(stats[k].pmax_bits);
            o_classes[nb] = best;
        }
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { hdr[0] = nc; hdr[1] = base; hdr[2] = truncated; hdr[3] = 0; }
}

__global__ __launch_bounds__(256) void dec_backproject(
    const float* __restrict__ pts, const float* __restrict__ xyz, const float* __restrict__ prob,
    const int* __restrict__ cls, int64_t n, float prob_thresh, const Cand* __restrict__ cands,
    const int* __restrict__ n_cand, Stats* __restrict__ stats) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    const bool have = i < n;
    float p0 = 0, p1 = 0, p2 = 0, x0 = 0, x1 = 0, x2 = 0, pr = 0;
    int cl = 0;
    if (have) {
        p0 = pts[i * 3]; p1 = pts[i * 3 + 1]; p2 = pts[i * 3 + 2];
        x0 = xyz[i * 3]; x1 = xyz[i * 3 + 1]; x2 = xyz[i * 3 + 2];
        pr = prob[i];
        cl = cls[i];
    }
    const int nc = n_cand[0];
    // A workgroup takes its 256 points through GROUPS of CB candidates (blockIdx.y, strided): with all candidates
    // in one workgroup the launch had 1.2 waves per SIMD, each walking 42 dependent (LDS read -> three IEEE
    // divisions -> LDS atomics) rounds on its own - 46 us for 3.4 M in-box tests.  Candidates are staged through
    // LDS (a dependent L2 load per candidate per wave dominated the first version).
    constexpr int CB = 8;
    __shared__ float c_cw[CB][3], c_sc[CB][3], c_cs[CB], c_sn[CB];
    // per-workgroup statistics in LDS, flushed once per candidate group: thousands of waves adding to
    // the same five global words per candidate serialise at ~11 ns per atomic (93 us for 42 candidates).
    // The lanes inside a box add to them directly: scan order is not spatial, so most waves hold one or two points
    // of most boxes, and a wave-level reduction (18 cross-lane moves per candidate for those one or two lanes) cost
    // several times the handful of LDS atomics it saved
    __shared__ unsigned l_in[CB], l_mask[CB], l_pmax[CB], l_hist[CB][NCLS];
    __shared__ double l_err[CB];
    for (int k0 = blockIdx.y * CB; k0 < nc; k0 += gridDim.y * CB) {
        const int nb = min(CB, nc - k0);
        __syncthreads();
        if (threadIdx.x < nb) {
            const Cand& cd = cands[k0 + threadIdx.x];
            for (int d = 0; d < 3; ++d) { c_cw[threadIdx.x][d] = cd.cw[d]; c_sc[threadIdx.x][d] = cd.sc[d]; }
            c_cs[threadIdx.x] = cd.cs;
            c_sn[threadIdx.x] = cd.sn;
        }
        if (threadIdx.x < CB) { l_in[threadIdx.x] = 0; l_mask[threadIdx.x] = 0; l_pmax[threadIdx.x] = 0; l_err[threadIdx.x] = 0.0; }
        for (int e = threadIdx.x; e < nb * NCLS; e += 256) (&l_hist[0][0])[e] = 0;
        __syncthreads();
        if (have)
            for (int kk = 0; kk < nb; ++kk) {
                float w0, w1, w2;
                if (!inv_coords(p0 - c_cw[kk][0], p1 - c_cw[kk][1], p2 - c_cw[kk][2], c_cs[kk], c_sn[kk], c_sc[kk],
                                w0, w1, w2))                                                 // :231-234
                    continue;
                atomicAdd(&l_in[kk], 1u);
                atomicMax(&l_pmax[kk], __float_as_uint(pr));       // prob >= 0: bit order == value order
                if (pr > prob_thresh) {                                                       // :245
                    const float e0 = x0 - w0, e1 = x1 - w1, e2 = x2 - w2;
                    const float ss = (e0 * e0 + e1 * e1) + e2 * e2;
                    atomicAdd(&l_mask[kk], 1u);
                    __hip_atomic_fetch_add(&l_err[kk], (double)(sqrtf(ss) * pr), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);                     // :250
                    if ((unsigned)cl < (unsigned)NCLS) atomicAdd(&l_hist[kk][cl], 1u);
                }
            }
        __syncthreads();
        if (threadIdx.x < nb && l_in[threadIdx.x]) {
            const int k = k0 + threadIdx.x;
            atomicAdd(&stats[k].n_in, l_in[threadIdx.x]);
            if (l_mask[threadIdx.x]) {
                atomicAdd(&stats[k].n_mask, l_mask[threadIdx.x]);
                atomicAdd(&stats[k].err, l_err[threadIdx.x]);
            }
            atomicMax(&stats[k].pmax_bits, l_pmax[threadIdx.x]);
        }
        for (int e = threadIdx.x; e < nb * NCLS; e += 256) {
            const unsigned v = l_hist[e / NCLS][e % NCLS];
            if (v) atomicAdd(&stats[k0 + e / NCLS].hist[e % NCLS], v);
        }
    }
}

// one launch boundary (~2 us) instead of a release fence in each of the thousands of back-projection workgroups (an
// agent-scope release writes the XCD's L2 back: the fused version ran 4x slower)
__global__ __launch_bounds__(256) void dec_finalize(const Cand* __restrict__ cands, const Stats* __restrict__ stats,
                                                    const int* __restrict__ n_cand, cv_decode_params prm,
                                                    char* result, int M) {
    __shared__ int s_scan[256];
    finalize_block(cands, stats, n_cand[0], n_cand[1], prm, result, M, s_scan);
}

// optional: replay the zeroing on the real grid (the reference mutates grid_obj in place)
__global__ __launch_bounds__(256) void dec_apply(float* __restrict__ g_obj, Geo geo,
                                                 cv_decode_params prm, const Cand* __restrict__ cands,
                                                 const int* __restrict__ n_cand) {
    const int k = blockIdx.x;
    if (k >= *n_cand) return;
    const Cand& cd = cands[k];
    const int e = prm.elimination, hp = e + (prm.elim_hi_plus1 ? 1 : 0);
    {
        const int x0 = max(cd.c[0] - e, 0), y0 = max(cd.c[1] - e, 0), z0 = max(cd.c[2] - e, 0);
        const int x1 = min(cd.c[0] + hp, geo.X), y1 = min(cd.c[1] + hp, geo.Y),
                  z1 = min(cd.c[2] + hp, geo.Z);
        const int nx = max(x1 - x0, 0), ny = max(y1 - y0, 0), nz = max(z1 - z0, 0);
        for (int t = threadIdx.x; t < nx * ny * nz; t += 256) {
            const int z = z0 + t % nz, y = y0 + (t / nz) % ny, x = x0 + t / (nz * ny);
            g_obj[((int64_t)x * geo.Y + y) * geo.Z + z] = 0.f;
        }
    }
    const int nx = cd.chi[0] - cd.clo[0] + 1, ny = cd.chi[1] - cd.clo[1] + 1,
              nz = cd.chi[2] - cd.clo[2] + 1;
    for (int64_t t = threadIdx.x; t < (int64_t)nx * ny * nz; t += 256) {
        const int z = cd.clo[2] + (int)(t % nz), y = cd.clo[1] + (int)((t / nz) % ny),
                  x = cd.clo[0] + (int)(t / ((int64_t)nz * ny));
        float w0, w1, w2;
        if (inv_coords((float)(x - cd.c[0]) * geo.res, (float)(y - cd.c[1]) * geo.res,
                       (float)(z - cd.c[2]) * geo.res, cd.cs, cd.sn, cd.sc, w0, w1, w2))
            g_obj[((int64_t)x * geo.Y + y) * geo.Z + z] = 0.f;
    }
}

struct WsLayout {
    size_t off_list_idx, off_list_val, off_list_xy, off_list_z, off_list_geo, off_counters, off_cands, off_stats,
        off_result, total;
    WsLayout(int64_t G, int M) {
        size_t o = 0;
        off_list_idx = o; o = cv_align_up(o + sizeof(int) * G, 256);
        off_list_val = o; o = cv_align_up(o + sizeof(float) * G, 256);
        off_list_xy = o; o = cv_align_up(o + sizeof(unsigned) * G, 256);
        off_list_z = o; o = cv_align_up(o + sizeof(int) * G, 256);
        off_list_geo = o; o = cv_align_up(o + sizeof(float) * 5 * G, 256);
        off_counters = o; o = cv_align_up(o + 64, 256);
        off_cands = o; o = cv_align_up(o + sizeof(Cand) * M, 256);
        off_stats = o; o = cv_align_up(o + sizeof(Stats) * M, 256);
        off_result = o; o = cv_align_up(o + ResultLayout(M).total, 256);
        total = o;
    }
};

// Pinned, device-visible host buffers the last workgroup writes the results into (no copy launch, no pageable
// staging).  A buffer is owned by one call at a time; concurrent scenes (one host thread + stream each) get their own.
struct PinnedPool {
    std::mutex mu;
    std::vector<std::pair<char*, size_t>> free_list;
    char* take(size_t bytes) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < free_list.size(); ++i)
                if (free_list[i].second >= bytes) {
                    char* p = free_list[i].first;
                    cap_of_last = free_list[i].second;
                    free_list.erase(free_list.begin() + i);
                    return p;
                }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocMapped) != hipSuccess) return nullptr;
        cap_of_last = bytes;
        return static_cast<char*>(p);
    }
    void give(char* p, size_t bytes) {
        std::lock_guard<std::mutex> g(mu);
        if (free_list.size() < 64) free_list.emplace_back(p, bytes);
        else (void)hipHostFree(p);
    }
    static thread_local size_t cap_of_last;
};
thread_local size_t PinnedPool::cap_of_last = 0;
PinnedPool g_pinned;

}  // namespace

int cv_decode_f32_ev(float* d_grid_obj, const float* d_grid_rot, const float* d_grid_scale,
                     const int dims[3], const float h_corner3[3], float res, const float* d_points,
                     const float* d_xyz, const float* d_prob, const int32_t* d_class, int64_t n,
                     const cv_decode_params* params, int mutate_grid, void* d_ws, size_t ws_bytes,
                     int* h_n_cand, int64_t* h_cand_idx, int32_t* h_verdict, int* h_n_boxes,
                     float* h_boxes, float* h_scores, int32_t* h_classes, int* h_truncated, void* stream, void* ev_done);

extern "C" {

size_t cv_decode_workspace_bytes(const int dims[3], int64_t n, int max_iters) {
    if (!dims || max_iters <= 0) return 0;
    (void)n;
    return WsLayout((int64_t)dims[0] * dims[1] * dims[2], max_iters).total;
}

int cv_decode_f32(float* d_grid_obj, const float* d_grid_rot, const float* d_grid_scale,
                  const int dims[3], const float h_corner3[3], float res, const float* d_points,
                  const float* d_xyz, const float* d_prob, const int32_t* d_class, int64_t n,
                  const cv_decode_params* params, int mutate_grid, void* d_ws, size_t ws_bytes,
                  int* h_n_cand, int64_t* h_cand_idx, int32_t* h_verdict, int* h_n_boxes,
                  float* h_boxes, float* h_scores, int32_t* h_classes, int* h_truncated, void* stream) {
    return cv_decode_f32_ev(d_grid_obj, d_grid_rot, d_grid_scale, dims, h_corner3, res, d_points, d_xyz, d_prob, d_class, n, params,
                            mutate_grid, d_ws, ws_bytes, h_n_cand, h_cand_idx, h_verdict, h_n_boxes, h_boxes, h_scores, h_classes,
                            h_truncated, stream, nullptr);
}

}  // extern "C"

int cv_decode_f32_ev(float* d_grid_obj, const float* d_grid_rot, const float* d_grid_scale,
                     const int dims[3], const float h_corner3[3], float res, const float* d_points,
                     const float* d_xyz, const float* d_prob, const int32_t* d_class, int64_t n,
                     const cv_decode_params* params, int mutate_grid, void* d_ws, size_t ws_bytes,
                     int* h_n_cand, int64_t* h_cand_idx, int32_t* h_verdict, int* h_n_boxes,
                     float* h_boxes, float* h_scores, int32_t* h_classes, int* h_truncated, void* stream, void* ev_done) {
    CV_REQUIRE(d_grid_obj && d_grid_rot && d_grid_scale && dims && h_corner3 && d_points && d_xyz &&
                   d_prob && d_class && params && d_ws && h_n_cand && h_cand_idx && h_verdict &&
                   h_n_boxes && h_boxes && h_scores && h_classes,
               CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n > 0, CV_EINVAL, "n must be positive");
    CV_REQUIRE(res > 0.f, CV_EINVAL, "res must be positive");
    CV_REQUIRE(dims[0] > 0 && dims[1] > 0 && dims[2] > 0, CV_EINVAL, "bad grid dims");
    CV_REQUIRE(dims[0] < 65536 && dims[1] < 65536 && dims[2] < 65536, CV_EINVAL, "grid dims must be below 65536");
    const int64_t G = (int64_t)dims[0] * dims[1] * dims[2];
    CV_REQUIRE(G < (1ll << 31), CV_EINVAL, "grid too large");
    const int M = params->max_iters;
    CV_REQUIRE(M > 0 && M <= 65536, CV_EINVAL, "max_iters out of range");
    CV_REQUIRE(params->elimination >= 0, CV_EINVAL, "elimination must be >= 0");
    // the reference loop ends when the grid maximum drops below thresh_high (eval_joint.py:208-209); with a
    // threshold <= 0 it never ends
    CV_REQUIRE(params->thresh_high > 0.f, CV_EINVAL, "thresh_high must be positive");
    const WsLayout W(G, M);
    CV_REQUIRE(ws_bytes >= W.total, CV_ENOMEM, "workspace too small (%zu < %zu)", ws_bytes, W.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(d_ws);
    List L;
    L.idx = reinterpret_cast<int*>(ws + W.off_list_idx);
    L.val = reinterpret_cast<float*>(ws + W.off_list_val);
    L.xy = reinterpret_cast<unsigned*>(ws + W.off_list_xy);
    L.z = reinterpret_cast<int*>(ws + W.off_list_z);
    L.geo = reinterpret_cast<float*>(ws + W.off_list_geo);
    L.cap = G;
    unsigned* list_n = reinterpret_cast<unsigned*>(ws + W.off_counters);
    int* n_cand = reinterpret_cast<int*>(ws + W.off_counters + 16);
    Cand* cands = reinterpret_cast<Cand*>(ws + W.off_cands);
    Stats* stats = reinterpret_cast<Stats*>(ws + W.off_stats);
    const ResultLayout RL(M);
    char* host = g_pinned.take(RL.total);
    CV_REQUIRE(host != nullptr, CV_ENOMEM, "pinned host buffer of %zu bytes", RL.total);
    const size_t host_cap = PinnedPool::cap_of_last;
    struct Giveback {
        char* p; size_t cap;
        ~Giveback() { g_pinned.give(p, cap); }
    } giveback{host, host_cap};

    CV_HIP_CHECK(hipMemsetAsync(ws + W.off_counters, 0, 64, st));
    Geo geo{dims[0], dims[1], dims[2], {h_corner3[0], h_corner3[1], h_corner3[2]}, res};
    const int cblocks = (int)std::min<int64_t>((G + 255) / 256, 2048);
    dec_compact<<<cblocks, 256, 0, st>>>(d_grid_obj, d_grid_rot, d_grid_scale, geo, G, params->thresh_high, L,
                                         list_n);
    CV_LAUNCH_CHECK();
    // the list length is only known on the device: the LDS-resident walker takes lists up to GREEDY_CAP, longer ones
    // the same code over the global arrays - both are launched, the one that does not apply returns at once
    // (grids beyond 4 M cells - 300k-point scenes - list more cells than the 256-thread walker keeps in registers)
    static const long long big_cells = getenv("CV_DEC_BIG_CELLS") ? atoll(getenv("CV_DEC_BIG_CELLS")) : (4ll << 20);
    // (round 6, profiles/r6/decode_in_flight.txt: a walk with the whole list in registers and no 115 KB LDS copy - placeable next
    // to any workgroup - and s_setprio 3 on the walk's waves were built and measured with seven scenes in flight: 0.213 against
    // 0.198 ms in the timed region, 0.163 against 0.147 alone, no change of the scene rate; what round 5 read as "decode takes
    // 4-14 x longer under load" was the host's wake-up behind the stream wait, inside the stage's event pair - the event now sits
    // in front of the wait)
    if (G > big_cells) dec_greedy_dispatch_big<<<1, GREEDY_T_BIG, 0, st>>>(geo, *params, L, list_n, cands, stats, n_cand);
    else dec_greedy_dispatch<<<1, GREEDY_T, 0, st>>>(geo, *params, L, list_n, cands, stats, n_cand);
    CV_LAUNCH_CHECK();
    // candidate groups of 8 over blockIdx.y (the count is only known on the device: groups beyond it return at once)
    const dim3 bgrid((unsigned)((n + 255) / 256), (unsigned)std::min((M + 7) / 8, 8));
    dec_backproject<<<bgrid, 256, 0, st>>>(
        d_points, d_xyz, d_prob, d_class, n, params->prob_thresh, cands, n_cand, stats);
    CV_LAUNCH_CHECK();
    dec_finalize<<<1, 256, 0, st>>>(cands, stats, n_cand, *params, host, M);
    CV_LAUNCH_CHECK();
    if (mutate_grid) {
        dec_apply<<<M, 256, 0, st>>>(d_grid_obj, geo, *params, cands, n_cand);
        CV_LAUNCH_CHECK();
    }
    if (ev_done) CV_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(ev_done), st));
    CV_HIP_CHECK(hipStreamSynchronize(st));
    const int* hdr = reinterpret_cast<const int*>(host);
    const int nc = hdr[0], nb = hdr[1];
    CV_REQUIRE(nc >= 0 && nc <= M && nb >= 0 && nb <= nc, CV_ERANGE, "corrupt decode result");
    *h_n_cand = nc;
    *h_n_boxes = nb;
    if (h_truncated) *h_truncated = hdr[2];
    std::memcpy(h_cand_idx, host + RL.off_cand, sizeof(int64_t) * nc);
    std::memcpy(h_verdict, host + RL.off_verdict, sizeof(int32_t) * nc);
    std::memcpy(h_boxes, host + RL.off_boxes, sizeof(float) * 24 * nb);
    std::memcpy(h_scores, host + RL.off_scores, sizeof(float) * nb);
    std::memcpy(h_classes, host + RL.off_classes, sizeof(int32_t) * nb);
    return CV_OK;
}


