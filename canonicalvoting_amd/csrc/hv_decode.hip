// Detection decode for gfx950: greedy peak picking, grid suppression, LCC-aware
// back-projection check, class vote.
//
// Replaces the inline host loop eval_joint.py:195-263 (one full-grid argmax and >= 6
// device->host syncs per candidate) with four launches and ONE sync per scene:
//
//   dec_compact      cells with grid_obj >= thresh_high -> compact (index, value) list.
//                    The loop stops when the maximum drops below thresh_high (:208-209)
//                    and only ever writes zeros, so cells below the threshold can never
//                    influence which candidates are examined.
//   dec_greedy       one workgroup walks the list: argmax (ties -> lowest flat index,
//                    like torch.argmax), box from the rot/scale grids (:213-223), zero
//                    the +-elimination cube (:211) and the cells inside the oriented box
//                    (:225-229,:243).  The candidate sequence does not depend on the
//                    back-projection verdicts (the reference `continue`s after zeroing).
//   dec_backproject  all points x all candidates in parallel (:231-250): in-box test,
//                    counts, prob-weighted LCC error, max prob, class histogram.
//   dec_finalize     verdicts (:246-253), class mode (:255-256), box corners (:258).
//
// fp32 conventions shared with oracle/decode_oracle.c (see its header): double-rounded
// atan2/cos/sin, k-ordered fmaf chain for the [.,3]@[3,3] products, x*(1/res) where
// torch divides by a python scalar, double accumulation of the masked mean.
// Compiled with -ffp-contract=off.
#include "cv_common.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int NCLS = 64;   // class histogram bins (class ids must be in [0, 64))

struct Cand {
    long long idx;
    int c[3];
    int clo[3], chi[3];
    float cw[3];
    float cs, sn;
    float sc[3];
    float bb[24];
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

__constant__ float kRawX[8] = {1, 1, -1, -1, 1, 1, -1, -1};
__constant__ float kRawY[8] = {1, 1, 1, 1, -1, -1, -1, -1};
__constant__ float kRawZ[8] = {1, -1, -1, 1, 1, -1, -1, 1};

__device__ __forceinline__ int lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

__global__ __launch_bounds__(256) void dec_compact(const float* __restrict__ g_obj, int64_t G,
                                                   float thresh, int* __restrict__ list_idx,
                                                   float* __restrict__ list_val,
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
            list_idx[p] = (int)i;
            list_val[p] = v;
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

__global__ __launch_bounds__(1024) void dec_greedy(const float* __restrict__ g_rot,
                                                   const float* __restrict__ g_scale, Geo geo,
                                                   cv_decode_params prm,
                                                   const int* __restrict__ list_idx,
                                                   float* __restrict__ list_val,
                                                   const unsigned* __restrict__ list_n,
                                                   Cand* __restrict__ cands,
                                                   int* __restrict__ n_cand_out) {
    __shared__ float s_val[16];
    __shared__ int s_idx[16];
    __shared__ Cand cur;
    __shared__ int stop;
    // the list usually holds a few thousand cells: keep it in LDS so the two passes per candidate
    // (argmax, suppression) cost LDS latency instead of an L2 round trip each
    constexpr int LDS_CAP = 12288;
    __shared__ float l_val[LDS_CAP];
    __shared__ int l_idx[LDS_CAP];
    const int n = (int)*list_n;
    const bool in_lds = n <= LDS_CAP;
    if (in_lds) {
        for (int k = threadIdx.x; k < n; k += 1024) { l_val[k] = list_val[k]; l_idx[k] = list_idx[k]; }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_res = 1.0f / geo.res;
    int it = 0;
    for (; it < prm.max_iters; ++it) {
        // ---- argmax: largest value, lowest flat index on ties (eval_joint.py:205)
        float bv = -1.f;
        int bi = 0x7fffffff;
        for (int k = threadIdx.x; k < n; k += 1024) {
            const float v = in_lds ? l_val[k] : list_val[k];
            const int id = in_lds ? l_idx[k] : list_idx[k];
            if (v > bv || (v == bv && id < bi)) { bv = v; bi = id; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float v = s_val[0];
            int id = s_idx[0];
            for (int w = 1; w < 16; ++w)
                if (s_val[w] > v || (s_val[w] == v && s_idx[w] < id)) { v = s_val[w]; id = s_idx[w]; }
            stop = !(v >= prm.thresh_high);   // :208-209
            if (!stop) {
                Cand cd;
                cd.idx = id;
                const int X = geo.X, Y = geo.Y, Z = geo.Z;
                cd.c[2] = id % Z;
                cd.c[1] = (id / Z) % Y;
                cd.c[0] = id / (Z * Y);
                for (int k = 0; k < 3; ++k) cd.cw[k] = geo.corner[k] + geo.res * (float)cd.c[k];  // :206
                const float r0 = g_rot[(int64_t)id * 2], r1 = g_rot[(int64_t)id * 2 + 1];
                const float rot = (float)atan2((double)r1, (double)r0);                            // :214
                cd.cs = (float)cos((double)rot);
                cd.sn = (float)sin((double)rot);
                for (int k = 0; k < 3; ++k) cd.sc[k] = g_scale[(int64_t)id * 3 + k];              // :216
                const float m00 = cd.cs * cd.sc[0], m02 = (-cd.sn) * cd.sc[2], m11 = cd.sc[1],
                            m20 = cd.sn * cd.sc[0], m22 = cd.cs * cd.sc[2];
                float lo[3], hi[3];
                for (int q = 0; q < 8; ++q) {                                                      // :219
                    cd.bb[q * 3 + 0] = m00 * kRawX[q] + m02 * kRawZ[q];
                    cd.bb[q * 3 + 1] = m11 * kRawY[q];
                    cd.bb[q * 3 + 2] = m20 * kRawX[q] + m22 * kRawZ[q];
                    for (int k = 0; k < 3; ++k) {
                        const float v2 = cd.bb[q * 3 + k];
                        if (q == 0 || v2 < lo[k]) lo[k] = v2;
                        if (q == 0 || v2 > hi[k]) hi[k] = v2;
                    }
                }
                const int shape[3] = {X, Y, Z};
                for (int k = 0; k < 3; ++k) {                                                      // :220-223
                    const int blo = (int)(lo[k] * inv_res), bhi = (int)(hi[k] * inv_res);
                    cd.clo[k] = min(max(cd.c[k] + blo, 0), shape[k] - 1);
                    cd.chi[k] = min(max(cd.c[k] + bhi, 0), shape[k] - 1);
                }
                cur = cd;
                cands[it] = cd;
            }
        }
        __syncthreads();
        if (stop) break;
        // ---- suppression on the compact list (:211, :225-229, :243)
        const int e = prm.elimination, hp = e + (prm.elim_hi_plus1 ? 1 : 0);
        const int cx = cur.c[0], cy = cur.c[1], cz = cur.c[2];
        for (int k = threadIdx.x; k < n; k += 1024) {
            if ((in_lds ? l_val[k] : list_val[k]) == 0.f) continue;
            const int id = in_lds ? l_idx[k] : list_idx[k];
            const int z = id % geo.Z, y = (id / geo.Z) % geo.Y, x = id / (geo.Z * geo.Y);
            bool kill = x >= cx - e && x < cx + hp && y >= cy - e && y < cy + hp && z >= cz - e &&
                        z < cz + hp;
            if (!kill && x >= cur.clo[0] && x <= cur.chi[0] && y >= cur.clo[1] && y <= cur.chi[1] &&
                z >= cur.clo[2] && z <= cur.chi[2]) {
                const float v0 = (float)(x - cx) * geo.res, v1 = (float)(y - cy) * geo.res,
                            v2 = (float)(z - cz) * geo.res;
                float w0, w1, w2;
                kill = inv_coords(v0, v1, v2, cur.cs, cur.sn, cur.sc, w0, w1, w2);
            }
            if (kill) { if (in_lds) l_val[k] = 0.f; else list_val[k] = 0.f; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_cand_out = it;
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
    const int lane = threadIdx.x & 63;
    const int nc = *n_cand;
    // candidates are staged through LDS 32 at a time: a dependent L2 load per candidate per wave
    // (42 x ~0.5 us) dominated this kernel
    constexpr int CB = 32;
    __shared__ float c_cw[CB][3], c_sc[CB][3], c_cs[CB], c_sn[CB];
    // per-workgroup statistics in LDS, flushed once per candidate batch: thousands of waves adding to
    // the same five global words per candidate serialise at ~11 ns per atomic (93 us for 42 candidates)
    __shared__ unsigned l_in[CB], l_mask[CB], l_pmax[CB], l_hist[CB][NCLS];
    __shared__ double l_err[CB];
    for (int k0 = 0; k0 < nc; k0 += CB) {
        const int nb = min(CB, nc - k0);
        __syncthreads();
        if (threadIdx.x < nb) {
            const Cand& cd = cands[k0 + threadIdx.x];
            for (int d = 0; d < 3; ++d) { c_cw[threadIdx.x][d] = cd.cw[d]; c_sc[threadIdx.x][d] = cd.sc[d]; }
            c_cs[threadIdx.x] = cd.cs;
            c_sn[threadIdx.x] = cd.sn;
        }
        if (threadIdx.x < CB) { l_in[threadIdx.x] = 0; l_mask[threadIdx.x] = 0; l_pmax[threadIdx.x] = 0; l_err[threadIdx.x] = 0.0; }
        for (int e = threadIdx.x; e < CB * NCLS; e += 256) (&l_hist[0][0])[e] = 0;
        __syncthreads();
        for (int kk = 0; kk < nb; ++kk) {
            float w0, w1, w2;
            const bool in = have && inv_coords(p0 - c_cw[kk][0], p1 - c_cw[kk][1], p2 - c_cw[kk][2], c_cs[kk],
                                               c_sn[kk], c_sc[kk], w0, w1, w2);                      // :231-234
            const uint64_t m_in = __ballot(in);
            if (m_in == 0) continue;
            const bool mk = in && pr > prob_thresh;                                      // :245
            const uint64_t m_mk = __ballot(mk);
            float pm = in ? pr : 0.f;
            double er = 0.0;
            if (mk) {
                const float e0 = x0 - w0, e1 = x1 - w1, e2 = x2 - w2;
                const float ss = (e0 * e0 + e1 * e1) + e2 * e2;
                er = (double)(sqrtf(ss) * pr);                                           // :250
                if ((unsigned)cl < (unsigned)NCLS) atomicAdd(&l_hist[kk][cl], 1u);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                pm = fmaxf(pm, __shfl_xor(pm, off));
                er += __shfl_xor(er, off);
            }
            if (lane == 0) {
                atomicAdd(&l_in[kk], (unsigned)__popcll(m_in));
                if (m_mk) {
                    atomicAdd(&l_mask[kk], (unsigned)__popcll(m_mk));
                    __hip_atomic_fetch_add(&l_err[kk], er, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                atomicMax(&l_pmax[kk], __float_as_uint(pm));   // prob >= 0: bit order == value order
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

// packed host result: [0]=n_cand [1]=n_boxes, then arrays sized by max_iters
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

__global__ void dec_finalize(const Cand* __restrict__ cands, const Stats* __restrict__ stats,
                             const int* __restrict__ n_cand, cv_decode_params prm, char* result,
                             int M) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const ResultLayout L(M);
    int* hdr = reinterpret_cast<int*>(result);
    long long* o_cand = reinterpret_cast<long long*>(result + L.off_cand);
    int* o_verdict = reinterpret_cast<int*>(result + L.off_verdict);
    float* o_boxes = reinterpret_cast<float*>(result + L.off_boxes);
    float* o_scores = reinterpret_cast<float*>(result + L.off_scores);
    int* o_classes = reinterpret_cast<int*>(result + L.off_classes);
    const int nc = *n_cand;
    int nb = 0;
    for (int k = 0; k < nc; ++k) {
        const Stats& s = stats[k];
        o_cand[k] = cands[k].idx;
        const float lhs = (float)s.n_mask, rhs = prm.valid_ratio * (float)s.n_in;
        if (lhs < rhs || (float)s.n_in < prm.thresh_low) { o_verdict[k] = 1; continue; }   // :246-247
        const float error = (float)(s.err / (double)s.n_mask);
        if ((double)error > prm.err_thresh) { o_verdict[k] = 2; continue; }               // :252-253
        int best = 0;
        unsigned best_cnt = 0;
        bool first = true;
        for (int c = 0; c < NCLS; ++c)
            if (first || s.hist[c] > best_cnt) { best_cnt = s.hist[c]; best = c; first = false; }
        for (int q = 0; q < 8; ++q)
            for (int d = 0; d < 3; ++d)
                o_boxes[(size_t)nb * 24 + q * 3 + d] = cands[k].bb[q * 3 + d] + cands[k].cw[d];  // :258
        o_scores[nb] = __uint_as_float(s.pmax_bits);
        o_classes[nb] = best;
        o_verdict[k] = 0;
        ++nb;
    }
    hdr[0] = nc;
    hdr[1] = nb;
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
    size_t off_list_idx, off_list_val, off_counters, off_cands, off_stats, off_result, total;
    WsLayout(int64_t G, int M) {
        size_t o = 0;
        off_list_idx = o; o = cv_align_up(o + sizeof(int) * G, 256);
        off_list_val = o; o = cv_align_up(o + sizeof(float) * G, 256);
        off_counters = o; o = cv_align_up(o + 64, 256);
        off_cands = o; o = cv_align_up(o + sizeof(Cand) * M, 256);
        off_stats = o; o = cv_align_up(o + sizeof(Stats) * M, 256);
        off_result = o; o = cv_align_up(o + ResultLayout(M).total, 256);
        total = o;
    }
};

}  // namespace

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
                  float* h_boxes, float* h_scores, int32_t* h_classes, void* stream) {
    CV_REQUIRE(d_grid_obj && d_grid_rot && d_grid_scale && dims && h_corner3 && d_points && d_xyz &&
                   d_prob && d_class && params && d_ws && h_n_cand && h_cand_idx && h_verdict &&
                   h_n_boxes && h_boxes && h_scores && h_classes,
               CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n > 0, CV_EINVAL, "n must be positive");
    CV_REQUIRE(res > 0.f, CV_EINVAL, "res must be positive");
    CV_REQUIRE(dims[0] > 0 && dims[1] > 0 && dims[2] > 0, CV_EINVAL, "bad grid dims");
    const int64_t G = (int64_t)dims[0] * dims[1] * dims[2];
    CV_REQUIRE(G < (1ll << 31), CV_EINVAL, "grid too large");
    const int M = params->max_iters;
    CV_REQUIRE(M > 0 && M <= 65536, CV_EINVAL, "max_iters out of range");
    CV_REQUIRE(params->elimination >= 0, CV_EINVAL, "elimination must be >= 0");
    const WsLayout W(G, M);
    CV_REQUIRE(ws_bytes >= W.total, CV_ENOMEM, "workspace too small (%zu < %zu)", ws_bytes, W.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(d_ws);
    int* list_idx = reinterpret_cast<int*>(ws + W.off_list_idx);
    float* list_val = reinterpret_cast<float*>(ws + W.off_list_val);
    unsigned* list_n = reinterpret_cast<unsigned*>(ws + W.off_counters);
    int* n_cand = reinterpret_cast<int*>(ws + W.off_counters + 16);
    Cand* cands = reinterpret_cast<Cand*>(ws + W.off_cands);
    Stats* stats = reinterpret_cast<Stats*>(ws + W.off_stats);
    char* result = ws + W.off_result;
    const ResultLayout RL(M);

    CV_HIP_CHECK(hipMemsetAsync(ws + W.off_counters, 0, 64, st));
    CV_HIP_CHECK(hipMemsetAsync(stats, 0, sizeof(Stats) * M, st));
    Geo geo{dims[0], dims[1], dims[2], {h_corner3[0], h_corner3[1], h_corner3[2]}, res};
    const int cblocks = (int)std::min<int64_t>((G + 255) / 256, 2048);
    dec_compact<<<cblocks, 256, 0, st>>>(d_grid_obj, G, params->thresh_high, list_idx, list_val,
                                         list_n);
    CV_LAUNCH_CHECK();
    dec_greedy<<<1, 1024, 0, st>>>(d_grid_rot, d_grid_scale, geo, *params, list_idx, list_val, list_n,
                                   cands, n_cand);
    CV_LAUNCH_CHECK();
    dec_backproject<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(
        d_points, d_xyz, d_prob, d_class, n, params->prob_thresh, cands, n_cand, stats);
    CV_LAUNCH_CHECK();
    dec_finalize<<<1, 64, 0, st>>>(cands, stats, n_cand, *params, result, M);
    CV_LAUNCH_CHECK();
    if (mutate_grid) {
        dec_apply<<<M, 256, 0, st>>>(d_grid_obj, geo, *params, cands, n_cand);
        CV_LAUNCH_CHECK();
    }
    std::vector<char> h(RL.total);
    CV_HIP_CHECK(hipMemcpyAsync(h.data(), result, RL.total, hipMemcpyDeviceToHost, st));
    CV_HIP_CHECK(hipStreamSynchronize(st));
    const int* hdr = reinterpret_cast<const int*>(h.data());
    const int nc = hdr[0], nb = hdr[1];
    CV_REQUIRE(nc >= 0 && nc <= M && nb >= 0 && nb <= nc, CV_ERANGE, "corrupt decode result");
    *h_n_cand = nc;
    *h_n_boxes = nb;
    std::memcpy(h_cand_idx, h.data() + RL.off_cand, sizeof(int64_t) * nc);
    std::memcpy(h_verdict, h.data() + RL.off_verdict, sizeof(int32_t) * nc);
    std::memcpy(h_boxes, h.data() + RL.off_boxes, sizeof(float) * 24 * nb);
    std::memcpy(h_scores, h.data() + RL.off_scores, sizeof(float) * nb);
    std::memcpy(h_classes, h.data() + RL.off_classes, sizeof(int32_t) * nb);
    return CV_OK;
}

}  // extern "C"
