// Sparse-convolution kernels that are NOT on the product's default path: parity-tested experiments kept for A/B runs.
//   conv_wave       (flavour 3) wave-independent: no LDS, no barriers, each wave walks its own live offsets
//   conv_tile       (flavour 4) pair-compacted tiles over a per-map plan (cv_sp_tile_plan) with packed weights
//   conv_rows_prof  instrumented twin of conv_rows (CV_CONV_PROF): shader-clock ticks per phase
// Moved out of sparse_conv.hip in round 3 (VERDICT r2 "weak" 11); arguments, epilogue and the hl format come from
// sparse_conv_common.h, the finish launch from sparse_conv.hip.
#include "sparse_conv_common.h"

using namespace cvsc;

namespace {

// Instrumented twin of conv_rows (CV_CONV_PROF=1): shader-clock ticks per phase, summed over waves into prof[16].
template <int NB, bool VEC>
__global__ __launch_bounds__(THREADS) void conv_rows_prof(ConvArgs a, unsigned long long* prof) {
    unsigned long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#define TICK(p) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[p] += t_ - pt0; pt0 = t_; } while (0)
    __shared__ float A_s[KC][A_LD];
    __shared__ float B_s[KC][NB * 32];
    __shared__ int nbr_s[TM];
    __shared__ int rows_s[TM];
    __shared__ __attribute__((aligned(16))) float ep_s[4][32][EP_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * (NB * 32);

    if (tid < TM) {
        // mask-sorted orders end with the rows that need the most offsets: start those tiles FIRST so the
        // light tiles fill the tail of the launch (longest-processing-time-first)
        const long long tile_id = a.row_perm ? (long long)gridDim.x - 1 - blockIdx.x : blockIdx.x;
        const long long t = tile_id * TM + tid;
        const int* perm = a.row_perm ? a.row_perm + (a.perm_per_split ? (long long)blockIdx.z * a.n_out : 0) : nullptr;
        rows_s[tid] = t < a.n_out ? (perm ? perm[t] : (int)t) : -1;
    }
    __syncthreads();
    TICK(0);

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    auto compute = [&]() {
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) {
            const float av = A_s[kk + (lane >> 5)][wave * 32 + (lane & 31)];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float bv = B_s[kk + (lane >> 5)][nb * 32 + (lane & 31)];
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nb], 0, 0, 0);
            }
        }
    };

    const int nj = a.j_end - a.j_begin;
    if (VEC) {
        // thread -> (row = tid/8 + 32*i, 4 channels at (tid%8)*4) : 8 lanes cover one 128 B row chunk
        const int a_col = (tid & 7) * 4;
        const int a_row = tid >> 3;                      // + 32*i, i = 0..3
        constexpr int B_F4 = KC * NB * 32 / 4;           // float4s in the weight slab
        constexpr int B_PER = (B_F4 + THREADS - 1) / THREADS;
        // work units = (kernel offset, 32-channel chunk); a split owns a contiguous range of units,
        // or a whole offset group when every group has its own row order
        const int nch = a.cin / KC;
        int u_lo, u_hi;
        if (a.perm_per_split) {
            u_lo = (int)((long long)nj * blockIdx.z / a.splits) * nch;
            u_hi = (int)((long long)nj * (blockIdx.z + 1) / a.splits) * nch;
        } else {
            u_lo = (int)((long long)nj * nch * blockIdx.z / a.splits);
            u_hi = (int)((long long)nj * nch * (blockIdx.z + 1) / a.splits);
        }
        const int j_first = a.j_begin + u_lo / nch, j_last = a.j_begin + (u_hi - 1) / nch;
        for (int j = j_first; j <= j_last && u_hi > u_lo; ++j) {
            const int kc_begin = (j == j_first ? u_lo % nch : 0) * KC;
            const int kc_end = (j == j_last ? (u_hi - 1) % nch + 1 : nch) * KC;
            int my = -1;
            if (tid < TM) {
                const int row = rows_s[tid];
                if (row >= 0) {
                    // mask-sorted orders visit the rows at random: the group's map rows were copied in processing
                    // order next to the order itself, so this is a coalesced read (the row-indexed form costs a
                    // 64-byte sector per 4-byte entry: 140 MB per ts1 conv, 35 % of its wave time)
                    if (a.nbr_perm) {
                        const long long tile_id = (long long)gridDim.x - 1 - blockIdx.x;
                        my = a.nbr_perm[((long long)blockIdx.z * a.n_out + tile_id * TM + tid) * a.nbr_perm_w +
                                        (j - (a.j_begin + (int)((long long)nj * blockIdx.z / a.splits)))];
                    } else
                        my = a.nbr ? a.nbr[(long long)row * a.K + j] : row;
                }
                nbr_s[tid] = my;
            }
            const int anyone = __syncthreads_or(my >= 0);
            TICK(1);
            if (!anyone) continue;    // nobody in the tile has this neighbour
            // a wave whose 32 rows all miss this neighbour skips its MFMAs; it still takes part in
            // the staging and the barriers.  Rows are processed in an order that groups equal
            // neighbour masks (row_perm), which is what makes whole waves / tiles skippable.
            const bool wave_live = __any(nbr_s[wave * 32 + (lane & 31)] >= 0);
            float4 ra[4], rb[B_PER];
            auto load = [&](int kc) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int src = nbr_s[a_row + 32 * i];
                    ra[i] = (src >= 0 && !(a.dbg & 2)) ? *reinterpret_cast<const float4*>(a.in + (long long)src * a.in_ld + kc + a_col)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int i = 0; i < B_PER; ++i) {
                    const int f = tid + i * THREADS;
                    if (f < B_F4) {
                        const int kr = f / (NB * 8), c4 = (f % (NB * 8)) * 4;
                        const int col = n0 + c4;
                        const float* wp = a.w + ((long long)j * a.cin + kc + kr) * a.cout + col;
                        if (a.dbg & 4) rb[i] = make_float4(1.f, 1.f, 1.f, 1.f);
                        else if (col + 3 < a.cout) rb[i] = *reinterpret_cast<const float4*>(wp);
                        else {
                            rb[i].x = col < a.cout ? wp[0] : 0.f;
                            rb[i].y = col + 1 < a.cout ? wp[1] : 0.f;
                            rb[i].z = col + 2 < a.cout ? wp[2] : 0.f;
                            rb[i].w = 0.f;
                        }
                    }
                }
            };
            auto stage = [&]() {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = a_row + 32 * i;
                    A_s[a_col + 0][r] = ra[i].x; A_s[a_col + 1][r] = ra[i].y;
                    A_s[a_col + 2][r] = ra[i].z; A_s[a_col + 3][r] = ra[i].w;
                }
#pragma unroll
                for (int i = 0; i < B_PER; ++i) {
                    const int f = tid + i * THREADS;
                    if (f < B_F4) {
                        const int kr = f / (NB * 8), c4 = (f % (NB * 8)) * 4;
                        *reinterpret_cast<float4*>(&B_s[kr][c4]) = rb[i];
                    }
                }
            };
            load(kc_begin);
            TICK(2);
            for (int kc = kc_begin; kc < kc_end; kc += KC) {
                __syncthreads();                 // previous chunk's MFMAs are done with the LDS tiles
                TICK(3);
                stage();
                TICK(4);
                __syncthreads();
                TICK(5);
                if (kc + KC < kc_end) load(kc + KC);  // in flight while the matrix cores run
                TICK(6);
                if (wave_live && !(a.dbg & 1)) compute();
                TICK(7);
            }
            __syncthreads();
            TICK(8);
        }
    } else {
        const int k0 = a.j_begin * a.cin, ktot = a.j_end * a.cin;
        const int nchunks = (ktot - k0 + KC - 1) / KC;
        const int c_lo = (int)((long long)nchunks * blockIdx.z / a.splits);
        const int c_hi = (int)((long long)nchunks * (blockIdx.z + 1) / a.splits);
        for (int kc = k0 + c_lo * KC; kc < k0 + c_hi * KC; kc += KC) {
            __syncthreads();
            for (int e = tid; e < KC * TM; e += THREADS) {
                const int kk = e / TM, r = e % TM;
                const int kf = kc + kk;
                float v = 0.f;
                const int row = rows_s[r];
                if (kf < ktot && row >= 0) {
                    const int j = kf / a.cin, c = kf - j * a.cin;
                    const int src = a.nbr ? a.nbr[(long long)row * a.K + j] : row;
                    if (src >= 0) v = a.in[(long long)src * a.in_ld + c];
                }
                A_s[kk][r] = v;
            }
            for (int e = tid; e < KC * NB * 32; e += THREADS) {
                const int kr = e / (NB * 32), c = e % (NB * 32);
                const int kf = kc + kr, col = n0 + c;
                B_s[kr][c] = (kf < ktot && col < a.cout) ? a.w[(long long)kf * a.cout + col] : 0.f;
            }
            __syncthreads();
            compute();
        }
    }
    if (a.dbg & 8) {
        if (acc[0][0] == 123.456f) a.out[0] = 1.f;
    } else if (a.wide) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            epilogue_store_wide(a, acc[nb], rows_s + wave * 32, n0 + nb * 32, lane, ep_s[wave]);
    } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            epilogue_store(a, acc[nb], rows_s + wave * 32, n0 + nb * 32 + (lane & 31), lane);
    }
    TICK(9);
    if (lane == 0 && prof) {
        for (int p2 = 0; p2 < 10; ++p2) atomicAdd(&prof[p2], pacc[p2]);
        atomicAdd(&prof[10], 1ull);
    }
#undef TICK
}



// ------------------------------------------------------------------ wave-independent flavour
// For the big fine levels.  One wave owns 32 output rows (taken in mask-sorted order) and walks ONLY the
// kernel offsets that at least one of its rows needs: no workgroup barriers, no LDS staging, no
// waiting for neighbours' dead offsets.  MFMA operands come straight from L2 into registers:
//   A: lane l holds row (l&31), channels 8q + 4(l>>5) + {0..3} as one float4 per 8-channel sub-step;
//      sub-step (q, t) multiplies k = 8q + t (lanes 0-31) and k = 8q + 4 + t (lanes 32-63) - the
//      k pairing inside a 32x32x2 MFMA is free as long as B uses the same rows;
//   B: lane l loads NB consecutive floats of weight row k at columns NB*(l&31).. (one dwordxNB load),
//      so MFMA block nb computes the output columns == nb (mod NB); the epilogue un-permutes.
template <int NB>
struct BVec;
template <> struct BVec<1> { typedef float type; };
template <> struct BVec<2> { typedef float2 type; };
template <> struct BVec<3> { typedef float3 type; };
template <> struct BVec<4> { typedef float4 type; };

template <int NB>
__device__ __forceinline__ float bcomp(const typename BVec<NB>::type& v, int i);
template <> __device__ __forceinline__ float bcomp<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float bcomp<2>(const float2& v, int i) { return i ? v.y : v.x; }
template <> __device__ __forceinline__ float bcomp<3>(const float3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
template <> __device__ __forceinline__ float bcomp<4>(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

template <int NB>
__global__ __launch_bounds__(THREADS) void conv_wave(ConvArgs a) {
    typedef typename BVec<NB>::type bvec;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long blk = (long long)blockIdx.x * 4 + wave;            // 32-row block in processing order
    if (blk * 32 >= a.n_out) return;
    const int g = blockIdx.z;
    const int n0 = blockIdx.y * (NB * 32);
    const int* perm = a.row_perm ? a.row_perm + (a.perm_per_split ? (long long)g * a.n_out : 0) : nullptr;
    const long long t = blk * 32 + (lane & 31);
    const int row = t < a.n_out ? (perm ? perm[t] : (int)t) : -1;
    const int nj = a.j_end - a.j_begin;
    const int j_lo = a.j_begin + (int)((long long)nj * g / a.splits);
    const int j_hi = a.j_begin + (int)((long long)nj * (g + 1) / a.splits);
    const int half = lane >> 5;
    const int colb = n0 + NB * (lane & 31);                            // first of this lane's NB columns
    const bool col_ok = colb + NB <= a.cout;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    // software pipeline over the live (offset, chunk) steps: the loads of step s+1 are in flight while
    // the 16*NB MFMAs of step s run (operands are double-buffered in registers)
    struct Operands { float4 av[4]; bvec bv[4][4]; };
    auto next_live = [&](int j, int& src) {          // first offset >= j that some row of the block needs
        for (; j < j_hi; ++j) {
            src = row >= 0 ? (a.nbr ? a.nbr[(long long)row * a.K + j] : row) : -1;
            if (__any(src >= 0)) break;
        }
        return j;
    };
    auto load = [&](Operands& o, int j, int src, int kc) {
        const float* arow = a.in + (long long)(src >= 0 ? src : 0) * a.in_ld + half * 4 + kc;
        const float* wrow = a.w + ((long long)j * a.cin + half * 4 + kc) * a.cout + colb;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            o.av[q] = src >= 0 ? *reinterpret_cast<const float4*>(arow + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                if (col_ok) o.bv[q][tt] = *reinterpret_cast<const bvec*>(wrow + (long long)(8 * q + tt) * a.cout);
                else {
                    float tmp[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int i = 0; i < NB; ++i)
                        if (colb + i < a.cout) tmp[i] = wrow[(long long)(8 * q + tt) * a.cout + i];
                    o.bv[q][tt] = *reinterpret_cast<const bvec*>(tmp);
                }
            }
    };
    int src = -1;
    int j = next_live(j_lo, src);
    int kc = 0;
    Operands cur, nxt;
    if (j < j_hi) load(cur, j, src, 0);
    while (j < j_hi) {
        int j2 = j, src2 = src, kc2 = kc + KC;
        if (kc2 >= a.cin) { kc2 = 0; j2 = next_live(j + 1, src2); }
        if (j2 < j_hi) load(nxt, j2, src2, kc2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float ak[4] = {cur.av[q].x, cur.av[q].y, cur.av[q].z, cur.av[q].w};
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[tt], bcomp<NB>(cur.bv[q][tt], nb), acc[nb], 0, 0, 0);
        }
        cur = nxt;
        j = j2; src = src2; kc = kc2;
    }
    // epilogue: accumulator register r of lane l is output row (r&3) + 8(r>>2) + 4(l>>5) of the block,
    // MFMA block nb / lane column (l&31) is output column colb + nb
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int orow = __shfl(row, rr);
        if (orow < 0) continue;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = colb + nb;
            if (col >= a.cout) continue;
            float v = acc[nb][r];
            if (a.splits > 1) {
                a.partial[((long long)g * a.n_out + orow) * a.cout + col] = v;
            } else {
                if (a.acc_in) v += a.acc_in[(long long)orow * a.acc_ld + col];
                v = v * (a.scale ? a.scale[col] : 1.f) + (a.shift ? a.shift[col] : 0.f);
                if (a.res) v += a.res[(long long)orow * a.res_ld + col];
                if (a.relu) v = fmaxf(v, 0.f);
                a.out[(long long)orow * a.out_ld + col] = v;
            }
        }
    }
}

// ------------------------------------------------------------------ pair-compacted tile flavour
// The matrix cores only ever see (input, output) pairs that exist.  A plan kernel (tile_plan, once per kernel
// map, shared by every convolution on that map) cuts the output rows into tiles of TT = 128 consecutive rows
// (Z-order: neighbours are nearby rows) and compacts, per tile and kernel offset, the rows that HAVE that
// neighbour into a list of (input row, tile row) entries.  One workgroup of CS waves owns a tile x CS*32 output
// channels whose fp32 accumulators live in LDS.  It walks the lists in steps of 32 entries (two 16-row MFMA
// units; a step with <= 16 entries issues half the MFMAs): the 32 gathered input rows of a KW-channel chunk are
// staged row-major in LDS with 16-byte stores (double buffered: the next step's gathers are in flight during
// the MFMAs, entry lists are fetched two steps ahead), wave w multiplies them with columns [32w, 32w+32) of W_j
// on v_mfma_f32_16x16x4_f32, the accumulator tiles being read from and written back to the LDS rows the entries
// belong to - waves own disjoint column slices and the rows of one list are distinct, so no atomics are needed.
// k order: MFMA k-slot q of step s multiplies channel q*KW/4 + s of the chunk, so a lane's A operands of four
// consecutive steps are one ds_read_b128; the weights are pre-packed (pack_weights, once per weight tensor) in
// exactly the per-lane order of the B operand: one fully coalesced dwordx4 load per four steps, kept in registers
// while (offset, chunk) stays the same and prefetched one step before it changes.
// MFMA work = pairs padded to 16 per (tile, offset): 79 % (ts1) - 90 % (coarse levels) useful, against 25 - 55 %
// for output-stationary 32-row blocks.  The epilogue (BatchNorm affine, residual, ReLU) streams the finished
// tile out as full coalesced rows.  Small coordinate sets split the offsets over blockIdx.z into partial tiles
// reduced by conv_finish.
constexpr int TT = 128;
constexpr int T_MAXK = 27;
constexpr int PAD_ENT = -128;          // list padding: negative (no gather) and (e & 255) == TT (scratch row)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per tile: ent[(tile*K + j)*TT + p] = (input row << 8) | tile row of the p-th row of the tile that has
// neighbour j, cnt[tile*32 + j] = number of such rows
__global__ __launch_bounds__(256) void tile_plan(const int* __restrict__ nbr, long long n_out, int K,
                                                 const int* __restrict__ row_perm, int* __restrict__ ent,
                                                 int* __restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const long long tile = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (tile * TT >= n_out) return;
    const long long t0 = tile * TT + lane, t1 = t0 + 64;
    const long long g0 = t0 < n_out ? (row_perm ? row_perm[t0] : t0) : -1;
    const long long g1 = t1 < n_out ? (row_perm ? row_perm[t1] : t1) : -1;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int j = 0; j < K; ++j) {
        const int v0 = g0 >= 0 ? nbr[g0 * K + j] : -1, v1 = g1 >= 0 ? nbr[g1 * K + j] : -1;
        const unsigned long long m0 = __ballot(v0 >= 0), m1 = __ballot(v1 >= 0);
        const int n0v = __popcll(m0);
        int* e = ent + (tile * K + j) * TT;
        if (v0 >= 0) e[__popcll(m0 & below)] = (v0 << 8) | lane;
        if (v1 >= 0) e[n0v + __popcll(m1 & below)] = (v1 << 8) | (lane + 64);
        if (lane == 0) cnt[tile * 32 + j] = n0v + __popcll(m1);
    }
}

// wp float4 index ((((j*NC + c)*NW + w)*NG + g)*2 + t)*64 + lane holds, for lane = 16*q + n,
// W[j][c*KW + q*KW/4 + 4g + {0,1,2,3}][32w + 16t + n]      (NC = cin/KW, NW = cout/32, NG = KW/16)
__global__ __launch_bounds__(256) void pack_weights(const float* __restrict__ w, int K, int cin, int cout, int KW,
                                                    float4* __restrict__ wp) {
    const long long total = (long long)K * cin * cout / 4;
    const int NC = cin / KW, NW = cout / 32, NG = KW / 16;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i;
        const int lane = (int)(r % 64); r /= 64;
        const int t = (int)(r % 2); r /= 2;
        const int g = (int)(r % NG); r /= NG;
        const int ws = (int)(r % NW); r /= NW;
        const int c = (int)(r % NC); r /= NC;
        const int j = (int)r;
        const int q = lane >> 4, n = lane & 15;
        const float* p = w + ((long long)j * cin + c * KW + q * (KW / 4) + 4 * g) * cout + 32 * ws + 16 * t + n;
        wp[i] = make_float4(p[0], p[cout], p[2 * (long long)cout], p[3 * (long long)cout]);
    }
}

template <int CS, int KW>
__global__ __launch_bounds__(64 * CS) void conv_tile(ConvArgs a) {
    constexpr int NT = 64 * CS, CW = 32 * CS;
    constexpr int ROW_F4 = KW / 4;                    // float4s per gathered row chunk
    constexpr int A_F4 = 32 * ROW_F4;
    constexpr int A_PER = (A_F4 + NT - 1) / NT;
    constexpr int NG = KW / 16;                       // groups of four MFMA k-steps
    constexpr int A_LD = KW + 4;                      // floats per staged row (16-byte aligned, odd multiple of 4 banks)
    __shared__ float out_s[TT + 1][CW];               // row TT: scratch row of the list padding
    __shared__ __attribute__((aligned(16))) float A_s[2][32][A_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, l15 = lane & 15, l31 = lane & 31;
    const int n0 = blockIdx.y * CW;
    const int K = a.K;
    const int nj = a.j_end - a.j_begin;
    const int j_lo = a.j_begin + (int)((long long)nj * blockIdx.z / a.splits);
    const int j_hi = a.j_begin + (int)((long long)nj * (blockIdx.z + 1) / a.splits);
    const long long tile = blockIdx.x;
    const int tile_rows = (int)min((long long)TT, a.n_out - tile * TT);
    const int NC = a.cin / KW, NW = a.cout / 32;

    // lane j holds the list length of offset j
    int cntv = 0;
    if (lane >= j_lo && lane < j_hi) cntv = a.plan_cnt ? a.plan_cnt[tile * 32 + lane] : tile_rows;
    const unsigned long long live = __ballot(cntv > 0);
    for (int e = tid; e < TT * CW / 4; e += NT) reinterpret_cast<float4*>(&out_s[0][0])[e] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto first_live = [&](int from) {        // first offset >= from with a non-empty list, 64 if none
        const unsigned long long m = from < 64 ? (live >> from) << from : 0ull;
        return m ? (int)__ffsll((long long)m) - 1 : 64;
    };
    struct Step { int j, c, b, cnt; };       // offset, K chunk, block of 32 list entries, list length
    auto advance = [&](Step s) {             // b runs fastest so the weights stay in registers
        ++s.b;
        if (32 * s.b >= s.cnt) {
            s.b = 0;
            if (++s.c >= NC) {
                s.c = 0;
                s.j = first_live(s.j + 1);
                s.cnt = s.j < 64 ? __builtin_amdgcn_readlane(cntv, s.j) : 0;
            }
        }
        return s;
    };
    // entry of list slot 32*b + l31 (one per lane, lanes 32-63 mirror 0-31)
    auto load_ent = [&](const Step& s) {
        const int p = 32 * s.b + l31;
        if (s.j >= 64 || p >= s.cnt) return PAD_ENT;
        if (a.plan_ent) return a.plan_ent[(tile * K + s.j) * TT + p];
        const long long g = tile * TT + p;                                   // K == 1 on the same coordinate set
        return (int)(((a.row_perm ? a.row_perm[g] : (int)g) << 8) | p);
    };
    // gather mapping of this thread: float4 i covers list slot a_slot[i], channels 4*a_c4[i]..+3 of the chunk
    int a_slot[A_PER], a_c4[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int idx = tid + i * NT;
        a_slot[i] = idx < A_F4 ? idx / ROW_F4 : -1;
        a_c4[i] = idx - (idx / ROW_F4) * ROW_F4;
    }
    auto load_a = [&](int ent, int c, float4* ra) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int e = __shfl(ent, a_slot[i] & 31);
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_slot[i] >= 0 && e >= 0)
                ra[i] = *reinterpret_cast<const float4*>(a.in + (long long)(e >> 8) * a.in_ld + c * KW + 4 * a_c4[i]);
        }
    };
    auto store_a = [&](int buf, const float4* ra) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
            if (a_slot[i] >= 0) *reinterpret_cast<float4*>(&A_s[buf][a_slot[i]][4 * a_c4[i]]) = ra[i];
    };
    const float4* wlane = a.wp + wave * (NG * 2 * 64) + lane;
    auto load_b = [&](const Step& s, float4 (*bv)[2]) {
        const float4* p = wlane + ((long long)(s.j * NC + s.c) * NW + blockIdx.y * CS) * (NG * 2 * 64);
#pragma unroll
        for (int g = 0; g < NG; ++g) { bv[g][0] = p[(g * 2) * 64]; bv[g][1] = p[(g * 2 + 1) * 64]; }
    };

    Step s0, s1, s2;
    s0.j = first_live(j_lo); s0.c = 0; s0.b = 0;
    s0.cnt = s0.j < 64 ? __builtin_amdgcn_readlane(cntv, s0.j) : 0;
    s1 = s0.j < 64 ? advance(s0) : s0;
    s2 = s1.j < 64 ? advance(s1) : s1;
    int e0 = load_ent(s0), e1 = load_ent(s1), e2 = load_ent(s2);
    float4 bc[NG][2], bn[NG][2];
    float4 ra[A_PER];
    int cur = 0;
    if (s0.j < 64) {
        load_a(e0, s0.c, ra);
        load_b(s0, bc);
        store_a(0, ra);
        if (s1.j < 64) load_a(e1, s1.c, ra);
    }
    __syncthreads();
    const int col0 = wave * 32 + l15;
    while (s0.j < 64) {
        // entries of step u+3 and weights of step u+1 go out before the MFMAs of step u; gathers of u+1 are in ra
        const Step s3 = s2.j < 64 ? advance(s2) : s2;
        const int e3 = load_ent(s3);
        const bool more = s1.j < 64;
        const bool new_b = more && (s1.j != s0.j || s1.c != s0.c);
        if (new_b) load_b(s1, bn);
        const bool two = s0.cnt - 32 * s0.b > 16;
        // accumulator tiles come from / go back to the LDS rows of the entries: D row 4*kq + i, col l15
        int r0[4], r1[4];
        f32x4 c00, c01, c10, c11;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r0[i] = __shfl(e0, 4 * kq + i) & 255;
            c00[i] = out_s[r0[i]][col0];
            c01[i] = out_s[r0[i]][col0 + 16];
        }
        if (two) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r1[i] = __shfl(e0, 16 + 4 * kq + i) & 255;
                c10[i] = out_s[r1[i]][col0];
                c11[i] = out_s[r1[i]][col0 + 16];
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const float4 a0 = *reinterpret_cast<const float4*>(&A_s[cur][l15][kq * (KW / 4) + 4 * g]);
                const float4 a1 = *reinterpret_cast<const float4*>(&A_s[cur][16 + l15][kq * (KW / 4) + 4 * g]);
                const float a0v[4] = {a0.x, a0.y, a0.z, a0.w}, a1v[4] = {a1.x, a1.y, a1.z, a1.w};
                const float b0v[4] = {bc[g][0].x, bc[g][0].y, bc[g][0].z, bc[g][0].w};
                const float b1v[4] = {bc[g][1].x, bc[g][1].y, bc[g][1].z, bc[g][1].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v[q], b0v[q], c00, 0, 0, 0);
                    c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v[q], b1v[q], c01, 0, 0, 0);
                    c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v[q], b0v[q], c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v[q], b1v[q], c11, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { out_s[r1[i]][col0] = c10[i]; out_s[r1[i]][col0 + 16] = c11[i]; }
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const float4 a0 = *reinterpret_cast<const float4*>(&A_s[cur][l15][kq * (KW / 4) + 4 * g]);
                const float a0v[4] = {a0.x, a0.y, a0.z, a0.w};
                const float b0v[4] = {bc[g][0].x, bc[g][0].y, bc[g][0].z, bc[g][0].w};
                const float b1v[4] = {bc[g][1].x, bc[g][1].y, bc[g][1].z, bc[g][1].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v[q], b0v[q], c00, 0, 0, 0);
                    c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v[q], b1v[q], c01, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { out_s[r0[i]][col0] = c00[i]; out_s[r0[i]][col0 + 16] = c01[i]; }
        if (more) store_a(cur ^ 1, ra);
        if (new_b) {
#pragma unroll
            for (int g = 0; g < NG; ++g) { bc[g][0] = bn[g][0]; bc[g][1] = bn[g][1]; }
        }
        if (s2.j < 64) load_a(e2, s2.c, ra);      // gathers of step u+2: in flight across the barrier and step u+1
        __syncthreads();
        cur ^= 1;
        s0 = s1; s1 = s2; s2 = s3;
        e0 = e1; e1 = e2; e2 = e3;
    }

    // epilogue: full rows, float4 per lane
    constexpr int C4 = CS * 8;
    for (int e = tid; e < TT * C4; e += NT) {
        const int t = e / C4, q = e - t * C4;
        if (t >= tile_rows) break;
        const long long g = tile * TT + t;
        const int row = a.row_perm ? a.row_perm[g] : (int)g;
        const int col = n0 + 4 * q;
        float4 v = *reinterpret_cast<const float4*>(&out_s[t][4 * q]);
        if (a.splits > 1) {
            *reinterpret_cast<float4*>(a.partial + ((long long)blockIdx.z * a.n_out + row) * a.cout + col) = v;
            continue;
        }
        if (a.acc_in) {
            const float4 p = *reinterpret_cast<const float4*>(a.acc_in + (long long)row * a.acc_ld + col);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if (a.scale) {
            const float4 s = *reinterpret_cast<const float4*>(a.scale + col);
            v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
        }
        if (a.shift) {
            const float4 s = *reinterpret_cast<const float4*>(a.shift + col);
            v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        }
        if (a.res) {
            const float4 p = *reinterpret_cast<const float4*>(a.res + (long long)row * a.res_ld + col);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(a.out + (long long)row * a.out_ld + col) = v;
    }
}


template <int NB>
int launch_wave(const ConvArgs& a, hipStream_t st) {
    dim3 grid((unsigned)((a.n_out + 127) / 128), (unsigned)((a.cout + NB * 32 - 1) / (NB * 32)), (unsigned)a.splits);
    conv_wave<NB><<<grid, THREADS, 0, st>>>(a);
    CV_LAUNCH_CHECK();
    if (a.splits > 1) return launch_finish(a, st);
    return CV_OK;
}


}  // namespace

namespace cvsc {

int launch_rows_prof(const ConvArgs& a, int nb, hipStream_t st) {
    dim3 grid((unsigned)((a.n_out + TM - 1) / TM), (unsigned)((a.cout + nb * 32 - 1) / (nb * 32)), (unsigned)a.splits);
    static unsigned long long* d_prof = nullptr;
    if (!d_prof) CV_HIP_CHECK(hipMalloc(&d_prof, 16 * sizeof(unsigned long long)));
    CV_HIP_CHECK(hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), st));
    static const bool quiet = getenv("CV_CONV_PROF")[0] == 'q';      // ablation timing: no counters, no print
    unsigned long long* prof = quiet ? nullptr : d_prof;
    switch (nb) {
        case 1: conv_rows_prof<1, true><<<grid, THREADS, 0, st>>>(a, prof); break;
        case 2: conv_rows_prof<2, true><<<grid, THREADS, 0, st>>>(a, prof); break;
        case 3: conv_rows_prof<3, true><<<grid, THREADS, 0, st>>>(a, prof); break;
        default: conv_rows_prof<4, true><<<grid, THREADS, 0, st>>>(a, prof); break;
    }
    CV_LAUNCH_CHECK();
    if (!quiet) {
        unsigned long long h[16];
        CV_HIP_CHECK(hipMemcpyAsync(h, d_prof, sizeof h, hipMemcpyDeviceToHost, st));
        CV_HIP_CHECK(hipStreamSynchronize(st));
        static const char* names[10] = {"prologue", "nbr+or-barrier", "first-load-issue", "barrier-A", "stage(+vmcnt)",
                                        "barrier-B", "load-issue", "compute", "end-barrier", "epilogue"};
        fprintf(stderr, "conv_rows<%d> n_out %lld cin %d cout %d K %d splits %d: waves %llu, ticks/wave:", nb, a.n_out,
                a.cin, a.cout, a.K, a.splits, h[10]);
        for (int p2 = 0; p2 < 10; ++p2) fprintf(stderr, " %s %.0f", names[p2], (double)h[p2] / (double)std::max(1ull, h[10]));
        fprintf(stderr, "\n");
    }
    return a.splits > 1 ? launch_finish(a, st) : CV_OK;
}

int launch_wave_nb(const ConvArgs& a, int nb, hipStream_t st) {
    switch (nb) {
        case 1: return launch_wave<1>(a, st);
        case 2: return launch_wave<2>(a, st);
        case 3: return launch_wave<3>(a, st);
        default: return launch_wave<4>(a, st);
    }
}

}  // namespace cvsc

namespace {
// ---- tile flavour dispatch: CS = waves (32-column slices) per workgroup, KW = K chunk width
int tile_cs(int cout) {
    const int s = cout / 32;
    return s % 2 == 0 ? 2 : s % 3 == 0 ? 3 : 1;
}
}  // namespace

namespace cvsc {

int tile_kw(int cin, int cout) {
    if (cin % 32 || cout % 32) return 0;
    const int cap = tile_cs(cout) == 3 ? 96 : 128;      // LDS: tile + double-buffered staging <= 80 KB (2 per CU)
    for (int kw : {128, 96, 64, 32})
        if (kw <= cap && cin % kw == 0) return kw;
    return 0;
}
bool tile_ok(const ConvArgs& a, bool vec) {
    return vec && a.K <= T_MAXK && tile_kw(a.cin, a.cout) > 0 && a.n_in < (1ll << 23) && a.out_ld % 4 == 0 && a.wp &&
           (a.plan_ent || !a.nbr) && (!a.res || a.res_ld % 4 == 0) && (!a.acc_in || a.acc_ld % 4 == 0) &&
           ((reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.res) |
             reinterpret_cast<uintptr_t>(a.acc_in) | reinterpret_cast<uintptr_t>(a.scale) |
             reinterpret_cast<uintptr_t>(a.shift) | reinterpret_cast<uintptr_t>(a.wp)) & 15) == 0;
}
// offsets are split over blockIdx.z only while the launch still fits the chip in one round (2 workgroups per CU)
int tile_splits(long long n_out, int cout, int nj) {
    const long long wgs = ((n_out + TT - 1) / TT) * (cout / (tile_cs(cout) * 32));
    if (wgs >= 256) return 1;
    long long s = std::min<long long>(512 / wgs, nj);
    const long long by_traffic = (32ll << 20) / std::max<long long>(1, n_out * cout * 4);
    s = std::min(s, std::max<long long>(by_traffic, 2));
    return (int)std::max<long long>(s, 1);
}

}  // namespace cvsc

namespace {
template <int CS, int KW>
int launch_tile_k(const ConvArgs& a, hipStream_t st) {
    dim3 grid((unsigned)((a.n_out + TT - 1) / TT), (unsigned)(a.cout / (CS * 32)), (unsigned)a.splits);
    conv_tile<CS, KW><<<grid, 64 * CS, 0, st>>>(a);
    CV_LAUNCH_CHECK();
    if (a.splits > 1) return launch_finish(a, st);
    return CV_OK;
}
template <int CS>
int launch_tile_cs(const ConvArgs& a, hipStream_t st) {
    switch (tile_kw(a.cin, a.cout)) {
        case 128: if (CS < 3) return launch_tile_k<CS < 3 ? CS : 1, 128>(a, st);
        case 96: return launch_tile_k<CS, 96>(a, st);
        case 64: return launch_tile_k<CS, 64>(a, st);
        default: return launch_tile_k<CS, 32>(a, st);
    }
}
}  // namespace

namespace cvsc {
int launch_tile(const ConvArgs& a, hipStream_t st) {
    switch (tile_cs(a.cout)) {
        case 3: return launch_tile_cs<3>(a, st);
        case 2: return launch_tile_cs<2>(a, st);
        default: return launch_tile_cs<1>(a, st);
    }
}

}  // namespace cvsc

extern "C" {

size_t cv_sp_tile_plan_ints(long long n_out, int K, size_t* cnt_offset) {
    if (n_out <= 0 || K <= 0) return 0;
    const size_t tiles = (size_t)((n_out + TT - 1) / TT);
    const size_t ent = cv_align_up(tiles * (size_t)K * TT, 64);
    if (cnt_offset) *cnt_offset = ent;
    return ent + tiles * 32;
}

// Pair lists of the tile flavour for one kernel map (and one processing order): d_plan is
// cv_sp_tile_plan_ints(n_out, K, &cnt_offset) int32 words; plan_ent = d_plan, plan_cnt = d_plan + cnt_offset.
int cv_sp_tile_plan(const int32_t* d_nbr, long long n_out, int K, const int32_t* d_row_perm, int32_t* d_plan,
                    void* stream) {
    CV_REQUIRE(d_nbr && d_plan && n_out > 0 && K > 0 && K <= T_MAXK, CV_EINVAL, "bad tile plan arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    size_t off = 0;
    cv_sp_tile_plan_ints(n_out, K, &off);
    const long long tiles = (n_out + TT - 1) / TT;
    tile_plan<<<(unsigned)((tiles + 3) / 4), 256, 0, st>>>(d_nbr, n_out, K, d_row_perm, d_plan, d_plan + off);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

// K chunk width the tile kernel uses for a Cin x Cout convolution (the packing of its weights depends on it);
// 0 when the tile kernel does not take the shape.
int cv_sp_tile_kw(int cin, int cout) { return tile_kw(cin, cout); }

// d_wp[K*cin*cout] = d_w[K][cin][cout] re-ordered into the per-lane B operand order of the tile kernel.
int cv_sp_pack_weights_f32(const float* d_w, int K, int cin, int cout, float* d_wp, void* stream) {
    CV_REQUIRE(d_w && d_wp && K > 0, CV_EINVAL, "bad pack_weights arguments");
    const int kw = tile_kw(cin, cout);
    CV_REQUIRE(kw > 0, CV_EINVAL, "the tile kernel does not take Cin = %d, Cout = %d", cin, cout);
    CV_REQUIRE((reinterpret_cast<uintptr_t>(d_wp) & 15) == 0, CV_EINVAL, "d_wp must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long total = (long long)K * cin * cout / 4;
    pack_weights<<<(unsigned)std::min<long long>((total + 255) / 256, 4096), 256, 0, st>>>(
        d_w, K, cin, cout, kw, reinterpret_cast<float4*>(d_wp));
    CV_LAUNCH_CHECK();
    return CV_OK;
}

}  // extern "C"
