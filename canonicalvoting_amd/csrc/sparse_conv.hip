// Sparse convolution for gfx950 as an output-stationary implicit GEMM on the fp32 matrix cores.
//
// Supplies the arithmetic the reference gets from MinkowskiEngine [ME-ext]:
//   MinkowskiConvolution / ConvolutionTranspose  (utils/minkunet.py:53-119, utils/resnet.py:128-133)
//   MinkowskiBatchNorm (eval: per-channel affine), MinkowskiReLU, the residual add of
//   BasicBlock (utils/resnet.py:118-154 via ME BasicBlock) and `final`'s bias
//   out[u] = sum_j W_j^T x[nbr[u][j]]  over valid neighbours,  W = `kernel` [K][Cin][Cout]
//
// Precision: the reference network is fp32 (train_joint.py:219-223, no autocast) and the
// parity bar is 1e-4 on its outputs, so the GEMM runs on v_mfma_f32_32x32x2_f32: exact fp32
// products and fp32 accumulation at the fp32 matrix rate (157 TF/s peak; gfx950 has no
// xf32/tf32 path).
//
// Kernel shape (flavour "rows"): one workgroup = 4 waves owns 128 output rows x (NB*32) output
// channels.  Per kernel offset j it pulls the 128 neighbour indices; if no row of the tile has
// that neighbour the whole offset is skipped (tile-level sparsity).  Otherwise, per 32-channel
// K chunk: gathered input rows (A, 128x32) and the weight slab (B, 32 x NB*32) are staged in
// LDS (next chunk's global loads are issued into registers before the MFMAs of the current
// one), each wave runs 16 k-steps x NB MFMAs on its 32 rows.  The epilogue applies the folded
// BatchNorm/bias affine, the residual and ReLU and stores once.
// Small coordinate sets (coarse levels: hundreds of rows x 256 channels) split the kernel offsets over
// blockIdx.z into a workspace that conv_finish reduces (+ epilogue).
// Fine-grained sparsity: only ~23-50 % of (row, offset) pairs exist, but a 32-row MFMA block is live as
// soon as ONE of its rows has the neighbour.  The caller may therefore pass a processing order
// (row_perm) that groups rows with equal neighbour bit masks and run the offsets in two halves, each
// with the order sorted by that half's mask: 32-row blocks then need 29-55 % of the offsets instead of
// 90 %, and dead waves / tiles skip their MFMAs / staging.
#include "cv_common.h"

#include <atomic>
#include <cstring>
#include <type_traits>
#include <utility>

#include "sparse_conv_common.h"

using namespace cvsc;

namespace {

// VEC: Cin % 32 == 0 (float4 gathers inside one offset).  !VEC: flattened K = K*Cin (stem, Cin=3).
template <int NB, bool VEC>
__global__ __launch_bounds__(THREADS) void conv_rows(ConvArgs a) {
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
            if (!__syncthreads_or(my >= 0)) continue;    // nobody in the tile has this neighbour
            // a wave whose 32 rows all miss this neighbour skips its MFMAs; it still takes part in
            // the staging and the barriers.  Rows are processed in an order that groups equal
            // neighbour masks (row_perm), which is what makes whole waves / tiles skippable.
            const bool wave_live = __any(nbr_s[wave * 32 + (lane & 31)] >= 0);
            float4 ra[4], rb[B_PER];
            auto load = [&](int kc) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int src = nbr_s[a_row + 32 * i];
                    ra[i] = src >= 0 ? *reinterpret_cast<const float4*>(a.in + (long long)src * a.in_ld + kc + a_col)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int i = 0; i < B_PER; ++i) {
                    const int f = tid + i * THREADS;
                    if (f < B_F4) {
                        const int kr = f / (NB * 8), c4 = (f % (NB * 8)) * 4;
                        const int col = n0 + c4;
                        const float* wp = a.w + ((long long)j * a.cin + kc + kr) * a.cout + col;
                        if (col + 3 < a.cout) rb[i] = *reinterpret_cast<const float4*>(wp);
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
            for (int kc = kc_begin; kc < kc_end; kc += KC) {
                __syncthreads();                 // previous chunk's MFMAs are done with the LDS tiles
                stage();
                __syncthreads();
                if (kc + KC < kc_end) load(kc + KC);  // in flight while the matrix cores run
                if (wave_live) compute();
            }
            __syncthreads();
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
    if (a.wide) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            epilogue_store_wide(a, acc[nb], rows_s + wave * 32, n0 + nb * 32, lane, ep_s[wave]);
    } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            epilogue_store(a, acc[nb], rows_s + wave * 32, n0 + nb * 32 + (lane & 31), lane);
    }
}

// ------------------------------------------------------------------ fp32 products on the bf16 matrix cores
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate, and wall-time ablations (profiles/conv_ablate.py) show the
// fp32 MFMAs to be half of the conv time (ts1 96->96: 192 us, 95 us without them, gathers / weights / epilogue
// each ~10 us) with the pipe only 60 % busy while they run.  The piece-product kernels compute the SAME fp32 products on
// v_mfma_f32_32x32x16_bf16: every fp32 operand is split exactly into three bf16 pieces x = h + m + l (8 + 8 + 8
// significant bits; the pieces of an operand sum to it to within half an fp32 ulp), and the six piece products whose
// magnitude is >= 2^-16 of the full product (hh, hm, mh, mm, hl, lh) are accumulated in fp32 - bf16 x bf16 products
// are exact in fp32, the dropped ml / lm / ll terms are <= 2^-23 of the product, i.e. fp32 rounding level.
// Six bf16 MFMAs (K = 16, 32 cycles) replace eight fp32 MFMAs (K = 2, 64 cycles) per 16 channels: 0.375x the matrix
// time at fp32 accuracy (network output still within 1e-6 of the fp32 oracle).
// Activations are split when they are staged (5.5 VALU ops per value); weights are split, transposed to [col][k]
// and laid out per (offset, 32-channel chunk) once per weight tensor (cv_sp_pack_weights_x6_f32).
// LDS operand tiles: [plane][row][32 k] bf16, 64-byte rows whose 16-byte chunks are XOR-swizzled with (row >> 2) & 3
// so that the ds_read_b128 of the 16 lanes served together hit 16 different bank groups.
// wp layout of the fp16 pairs (unsigned short): ((((j*nch + c)*2 + plane)*cout + col)*32 + k); values are
// w * col_scale * mult (mult = 2^scale_log2 keeps the low pieces of small weights out of the fp16 subnormals)
__global__ __launch_bounds__(256) void pack_weights_h2(const float* __restrict__ w, int K, int cin, int cout,
                                                       const float* __restrict__ col_scale, float mult,
                                                       unsigned short* __restrict__ wp, int trans = 0) {
    const long long total = (long long)K * cin * cout / 2;
    const int nch = cin / 32;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i;
        const int k2 = (int)(r % 16); r /= 16;
        const int col = (int)(r % cout); r /= cout;
        const int c = (int)(r % nch); r /= nch;
        const int j = (int)r;
        // trans: w is [K][cout][cin] - the weights of the convolution whose transpose (input gradient) is being packed
        const float* p = trans ? w + ((long long)j * cout + col) * cin + c * 32 + 2 * k2
                               : w + ((long long)j * cin + c * 32 + 2 * k2) * cout + col;
        const float sc = (col_scale ? col_scale[col] : 1.f) * mult;
        unsigned h, l;
        split2h(p[0] * sc, p[trans ? 1 : cout] * sc, h, l);
        const long long base = ((long long)(j * nch + c) * 2 * cout + col) * 32 + 2 * k2;
        *reinterpret_cast<unsigned*>(wp + base) = h;
        *reinterpret_cast<unsigned*>(wp + base + (long long)cout * 32) = l;
    }
}

// wp6 layout (unsigned short): ((((j*nch + c)*3 + plane)*cout + col)*32 + k) for channel c*32 + k of offset j
// pack_weights_h2 for a whole network in ONE launch (cv_sp_pack_weights_h2_batch_f32): blockIdx.y = job
__global__ __launch_bounds__(256) void pack_weights_h2_batch(const cv_pack_job* __restrict__ jobs) {
    const cv_pack_job jb = jobs[blockIdx.y];
    const int K = jb.K, cin = jb.cin, cout = jb.cout, trans = jb.trans;
    const float* __restrict__ w = jb.w;
    unsigned short* __restrict__ wp = static_cast<unsigned short*>(jb.wp);
    const float mult = __uint_as_float((unsigned)(127 + jb.scale_log2) << 23);        // 2^scale_log2, |scale_log2| <= 60
    const long long total = (long long)K * cin * cout / 2;
    const int nch = cin / 32;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i;
        const int k2 = (int)(r % 16); r /= 16;
        const int col = (int)(r % cout); r /= cout;
        const int c = (int)(r % nch); r /= nch;
        const int j = (int)r;
        const float* p = trans ? w + ((long long)j * cout + col) * cin + c * 32 + 2 * k2
                               : w + ((long long)j * cin + c * 32 + 2 * k2) * cout + col;
        unsigned h, l;
        split2h(p[0] * mult, p[trans ? 1 : cout] * mult, h, l);
        const long long base = ((long long)(j * nch + c) * 2 * cout + col) * 32 + 2 * k2;
        *reinterpret_cast<unsigned*>(wp + base) = h;
        *reinterpret_cast<unsigned*>(wp + base + (long long)cout * 32) = l;
    }
}

__global__ __launch_bounds__(256) void pack_weights_x6(const float* __restrict__ w, int K, int cin, int cout,
                                                       const float* __restrict__ col_scale,
                                                       unsigned short* __restrict__ wp, int trans = 0) {
    const long long total = (long long)K * cin * cout / 2;                   // pairs of consecutive k
    const int nch = cin / 32;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i;
        const int k2 = (int)(r % 16); r /= 16;                               // k = 2*k2, 2*k2 + 1
        const int col = (int)(r % cout); r /= cout;
        const int c = (int)(r % nch); r /= nch;
        const int j = (int)r;
        const float* p = trans ? w + ((long long)j * cout + col) * cin + c * 32 + 2 * k2
                               : w + ((long long)j * cin + c * 32 + 2 * k2) * cout + col;
        unsigned h, m, l;
        const float sc = col_scale ? col_scale[col] : 1.f;
        split3(p[0] * sc, p[trans ? 1 : cout] * sc, h, m, l);
        const long long base = ((long long)(j * nch + c) * 3 * cout + col) * 32 + 2 * k2;
        *reinterpret_cast<unsigned*>(wp + base) = h;
        *reinterpret_cast<unsigned*>(wp + base + (long long)cout * 32) = m;
        *reinterpret_cast<unsigned*>(wp + base + 2ll * cout * 32) = l;
    }
}

// one bf16 plane (RNE): ((j*nch + c)*cout + col)*32 + k
__global__ __launch_bounds__(256) void pack_weights_b1(const float* __restrict__ w, int K, int cin, int cout,
                                                       const float* __restrict__ col_scale,
                                                       unsigned short* __restrict__ wp, int trans = 0) {
    const long long total = (long long)K * cin * cout / 2;
    const int nch = cin / 32;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i;
        const int k2 = (int)(r % 16); r /= 16;
        const int col = (int)(r % cout); r /= cout;
        const int c = (int)(r % nch); r /= nch;
        const int j = (int)r;
        const float* p = trans ? w + ((long long)j * cout + col) * cin + c * 32 + 2 * k2
                               : w + ((long long)j * cin + c * 32 + 2 * k2) * cout + col;
        const float sc = col_scale ? col_scale[col] : 1.f;
        *reinterpret_cast<unsigned*>(wp + ((long long)(j * nch + c) * cout + col) * 32 + 2 * k2) =
            cvt_pk_bf16(p[0] * sc, p[trans ? 1 : cout] * sc);
    }
}


// Tile of a workgroup.  a.xcd_tiles (CV_XCD_TILES=1, an experiment that is off by default; gridDim.x a multiple of 8): workgroups are dealt to the eight XCDs round-robin by
// their linear id, so blockIdx.x % 8 is the XCD; XCD r takes the r-th eighth of the tiles.  Rows are in spatial order (or
// in mask order INSIDE the same eighths, see group_mask), so the rows a tile gathers were mostly fetched by its
// neighbours on the same XCD: the mask-sorted ts1 conv missed that XCD's L2 on 52 % of its requests and fetched 192 MB
// per launch from the memory side for 31 MB of input when consecutive tiles went to consecutive XCDs.
__device__ __forceinline__ long long xcd_tile(const ConvArgs& a) {
    if (a.xcd_tiles) return (long long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    return a.row_perm ? (long long)gridDim.x - 1 - blockIdx.x : blockIdx.x;
}

// Piece-product convolution on fp32 activations (the training path; the eval network reads the hl format: conv_hl / conv_hd).
// P = 3: bf16 triples, six piece products.  P = 2: fp16 pairs, three piece products (same tiles with two planes).
// P = 1: operands rounded to bf16 (RNE), ONE bf16 x bf16 product with fp32 accumulation - the opt-in bf16 compute
// mode of the training configuration (BASELINE configs 3-4); not an fp32-parity path.
// ONE workgroup barrier per (offset, 32-channel) unit (round 1's conv_rows_x6 - removed in round 5, git 2945191 holds it -
// took two plus two per offset: ~56 barriers per workgroup, ~38 us of the 94 us ts1 96 -> 96 conv,
// profiles/r1/conv_ablate_h2.txt).  Two observations remove most of them:
//  * a wave's 32 rows are its own: it gathers, splits and stages exactly the A rows it multiplies, so the A tile needs
//    wave-level ordering only (LDS operations of one wave execute in order);
//  * the weight tile is shared, so it is double-buffered: the B planes of unit k+1 are written while slow waves may
//    still multiply unit k out of the other buffer, and the single barrier of unit k+1 (B visible) is also the
//    proof that everyone is done with unit k-1's buffer.
// The map entries of all the workgroup's offsets (<= NPRE: 10 in the default instance - the host splits a launch further - and
// 28 in the instance for unsplit 3x3x3 launches; beyond that the fp32 kernel conv_rows) come in with one round of loads, a bit mask of the offsets that exist for the
// tile is reduced once, and dead offsets are skipped without a barrier.
#ifndef CV_WP_ABL
#define CV_WP_ABL 0       // timing ablations of conv_rows_wp (wrong results): 1 no fp16 split, 2 no MFMA, 4 no weight tile, 8 no gathers
#endif
#ifndef CV_WP_NPRE
#define CV_WP_NPRE 10
#endif
constexpr int WP_NPRE = CV_WP_NPRE;          // the traffic cap of pick_splits leaves 9 offsets per workgroup on the training ts8 level
constexpr int WP_NPRE_BIG = 28;              // the second instance: a whole 3x3x3 kernel in one workgroup (flavour 1, wide mask groups)
template <int NB, int P, int NPRE = WP_NPRE>
__global__ __launch_bounds__(THREADS, (NB >= 3 ? 3 : NB == 2 ? 4 : 5)) void conv_rows_wp(ConvArgs a) {
    constexpr int A_BYTES = P * TM * 64, B_BYTES = P * NB * 32 * 64, EP_BYTES = 4 * 32 * EP_LD * 4;
    constexpr int SM_BYTES = A_BYTES + 2 * B_BYTES > EP_BYTES ? A_BYTES + 2 * B_BYTES : EP_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char sm[SM_BYTES];
    __shared__ int rows_s[TM];
    __shared__ int nbr_all[NPRE + 1][TM];            // + the rows themselves: the "map" of the second source
    __shared__ unsigned live_mask;
    unsigned char* const A_h = sm;                    // [plane][row][64 B]
    unsigned char* const B_h = sm + A_BYTES;          // 2 x [plane][col][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * (NB * 32);

    const long long tile_id = xcd_tile(a);
    if (tile_id * TM >= a.n_out) return;             // padding of the XCD-aware grid
    if (tid < TM) {
        const long long t = tile_id * TM + tid;
        const int* perm = a.row_perm ? a.row_perm + (a.perm_per_split ? (long long)blockIdx.z * a.n_out : 0) : nullptr;
        rows_s[tid] = t < a.n_out ? (perm ? perm[t] : (int)t) : -1;
    }
    if (tid == 0) live_mask = 0u;
    __syncthreads();

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    const int half = lane >> 5, l31 = lane & 31;
    auto compute = [&](const unsigned char* Bb) {
        const int arow = wave * 32 + l31;
        const int aswz = (arow >> 2) & 3, bswz = (l31 >> 2) & 3;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = 2 * ks + half;
            bf16x8 av[P];
#pragma unroll
            for (int p = 0; p < P; ++p)
                av[p] = *reinterpret_cast<const bf16x8*>(A_h + (p * TM + arow) * 64 + ((chunk ^ aswz) << 4));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                bf16x8 bv[P];
#pragma unroll
                for (int p = 0; p < P; ++p)
                    bv[p] = *reinterpret_cast<const bf16x8*>(Bb + (p * NB * 32 + nb * 32 + l31) * 64 + ((chunk ^ bswz) << 4));
                if constexpr (P == 1) {
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[0], acc[nb], 0, 0, 0);
                } else if constexpr (P == 3) {
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[2], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2], bv[0], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[1], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[1], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[0], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[0], acc[nb], 0, 0, 0);
                } else {
                    const f16x8 a0 = __builtin_bit_cast(f16x8, av[0]), a1 = __builtin_bit_cast(f16x8, av[P - 1]);
                    const f16x8 b0 = __builtin_bit_cast(f16x8, bv[0]), b1 = __builtin_bit_cast(f16x8, bv[P - 1]);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[nb], 0, 0, 0);
                }
            }
        }
    };

    const int nj = a.j_end - a.j_begin;
    // lane -> (own row = 32*wave + lane/8 + 8*i, 4 channels at (lane%8)*4): 8 lanes cover one 128 B row chunk
    const int a_col = (lane & 7) * 4;
    const int a_row = wave * 32 + (lane >> 3);       // + 8*i, i = 0..3
    constexpr int B_U4 = P * NB * 32 * 4;            // 16-byte pieces of the packed weight slab of one unit
    constexpr int B_PER = (B_U4 + THREADS - 1) / THREADS;
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
    const int njl = u_hi > u_lo ? j_last - j_first + 1 : 0;       // <= WP_NPRE (host)
    auto map_entry = [&](int t, int j) {
        const int row = rows_s[t];
        if (row < 0) return -1;
        if (a.nbr_perm) {
            return a.nbr_perm[((long long)blockIdx.z * a.n_out + tile_id * TM + t) * a.nbr_perm_w +
                              (j - (a.j_begin + (int)((long long)nj * blockIdx.z / a.splits)))];
        }
        return a.nbr ? a.nbr[(long long)row * a.K + j] : row;
    };
    {
        unsigned m = 0u;
        for (int e = tid; e < njl * TM; e += THREADS) {
            const int jj = e / TM, t = e - jj * TM;
            const int v = map_entry(t, j_first + jj);
            nbr_all[jj][t] = v;
            if (v >= 0) m |= 1u << jj;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m |= __shfl_xor(m, off);
        if (lane == 0 && m) atomicOr(&live_mask, m);
        if (a.in2 && tid < TM) nbr_all[njl][tid] = rows_s[tid];
    }
    __syncthreads();
    const unsigned lm = live_mask;

    float4 ra[4];
    uint4 rb[B_PER];
    float in_max = 0.f;                              // fp16 pairs: largest staged input magnitude
    // (offset j, chunk c) walk over the units [u_lo, u_hi) of the offsets that exist for this tile, without divisions
    const int c_first = u_lo - (u_lo / nch) * nch, c_last = u_hi > u_lo ? (u_hi - 1) - ((u_hi - 1) / nch) * nch + 1 : 0;
    auto skip_dead = [&](int& j, int& c) {
#pragma unroll 1
        while (j <= j_last && !((lm >> (j - j_first)) & 1u)) { ++j; c = 0; }
    };
    // after the last offset the units of the second source follow (BasicBlock's 1x1 downsample branch folded into conv2:
    // out += in2 @ W2 on the output rows; its 32-channel chunks are dealt round-robin to the splits / mask groups):
    // j == j_last + 1 marks them, c is then the chunk of in2; j == j_last + 2 ends the walk
    const int nch2 = a.in2 ? a.cin2 / KC : 0;
    const int j_second = j_last + 1, j_done = j_last + 2;
    auto advance = [&](int& j, int& c) {
        if (j <= j_last) {
            if (++c >= (j == j_last ? c_last : nch)) { ++j; c = 0; skip_dead(j, c); }
            if (j > j_last) { j = j_second; c = blockIdx.z; }
        } else {
            c += a.splits;
        }
        if (j == j_second && c >= nch2) j = j_done;
    };
    auto load = [&](int j, int c) {
        const bool second = j == j_second;
        const int kc = c * KC;
        const int* nb_j = nbr_all[j - j_first];          // j_second - j_first == njl: the rows themselves
        const float* src_base = second ? a.in2 : a.in;
        const int src_ld = second ? a.in2_ld : a.in_ld;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int src = nb_j[a_row + 8 * i];
#if (CV_WP_ABL & 8)
            ra[i] = make_float4((float)src, 0.f, 0.f, 0.f);
#else
            ra[i] = src >= 0 ? *reinterpret_cast<const float4*>(src_base + (long long)src * src_ld + kc + a_col)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        }
        const unsigned short* slab = second ? a.wp6_2 + (long long)c * P * a.cout * 32
                                            : a.wp6 + (long long)(j * nch + c) * P * a.cout * 32;
#if !(CV_WP_ABL & 4)
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int f = tid + i * THREADS;
            if (f < B_U4) {
                const int p = f / (NB * 32 * 4), rem = f - p * (NB * 32 * 4);
                const int col = rem >> 2, ch = rem & 3;
                rb[i] = (n0 + col < a.cout)
                            ? *reinterpret_cast<const uint4*>(slab + ((long long)p * a.cout + n0 + col) * 32 + ch * 8)
                            : make_uint4(0u, 0u, 0u, 0u);
            }
        }
#endif
    };
    auto stage = [&](unsigned char* Bb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = a_row + 8 * i;
            unsigned h0 = 0, m0 = 0, l0 = 0, h1 = 0, m1 = 0, l1 = 0;
            unsigned char* dst = A_h + r * 64 + ((((a_col >> 3) ^ ((r >> 2) & 3))) << 4) + ((a_col & 7) << 1);
            if constexpr (P == 1) {
                *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pk_bf16(ra[i].x, ra[i].y), cvt_pk_bf16(ra[i].z, ra[i].w));
            } else if constexpr (P == 3) {
                split3(ra[i].x, ra[i].y, h0, m0, l0);
                split3(ra[i].z, ra[i].w, h1, m1, l1);
                *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(dst + TM * 64) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(dst + 2 * TM * 64) = make_uint2(l0, l1);
            } else {
#if (CV_WP_ABL & 1)
                h0 = __float_as_uint(ra[i].x); l0 = __float_as_uint(ra[i].y); h1 = __float_as_uint(ra[i].z); l1 = __float_as_uint(ra[i].w);
#else
                in_max = fmaxf(fmaxf(in_max, fmaxf(fabsf(ra[i].x), fabsf(ra[i].y))), fmaxf(fabsf(ra[i].z), fabsf(ra[i].w)));
                split2h(ra[i].x, ra[i].y, h0, l0);
                split2h(ra[i].z, ra[i].w, h1, l1);
#endif
                *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(dst + TM * 64) = make_uint2(l0, l1);
            }
        }
#if !(CV_WP_ABL & 4)
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int f = tid + i * THREADS;
            if (f < B_U4) {
                const int p = f / (NB * 32 * 4), rem = f - p * (NB * 32 * 4);
                const int col = rem >> 2, ch = rem & 3;
                *reinterpret_cast<uint4*>(Bb + (p * NB * 32 + col) * 64 + ((ch ^ ((col >> 2) & 3)) << 4)) = rb[i];
            }
        }
#endif
    };
    int j = j_first, c = c_first, buf = 0;
    if (njl > 0) skip_dead(j, c); else j = j_last + 1;
    if (j > j_last) { j = j_second; c = blockIdx.z; if (c >= nch2) j = j_done; }
    if (j < j_done) load(j, c);
#pragma unroll 1
    while (j < j_done) {
        unsigned char* Bb = B_h + buf * B_BYTES;
        stage(Bb);                                   // A rows of this wave, this thread's share of the weight tile
        __syncthreads();                             // weight tile visible; everyone is done with the other buffer's previous use
        const bool wave_live = __any(nbr_all[j - j_first][wave * 32 + l31] >= 0);
        advance(j, c);
        if (j < j_done) load(j, c);                  // in flight while the matrix cores run
#if !(CV_WP_ABL & 2)
        if (wave_live) compute(Bb);
#else
        if (wave_live && in_max == 12345.f) compute(Bb);
#endif
        buf ^= 1;
    }
    __syncthreads();                                 // operand tiles are dead: the epilogue tile reuses their LDS
    if constexpr (P == 2) {
        if (in_max > 65000.f && a.range_flag) *a.range_flag = 1;
        const float k = a.acc_scale;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] *= k;
    }
    float (*ep)[EP_LD] = reinterpret_cast<float (*)[EP_LD]>(sm + wave * 32 * EP_LD * 4);
    if (a.wide) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) epilogue_store_wide(a, acc[nb], rows_s + wave * 32, n0 + nb * 32, lane, ep);
    } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            epilogue_store(a, acc[nb], rows_s + wave * 32, n0 + nb * 32 + (lane & 31), lane);
    }
}

// ------------------------------------------------------------------ fp16-pair convolution on hl-format activations
// conv_rows_wp's tiling (128 rows x NB*32 columns, a wave owns 32 rows, weight tile shared through LDS, one workgroup
// barrier per (offset, 32-channel) unit) with the gathered operand taken straight from global memory: in the hl format
// a lane's MFMA A fragments of a unit are four contiguous 16-byte pieces of its row's 128-byte chunk, so the gather IS
// the fragment load - no fp16 split (it was 2/3 of the VALU work of conv_rows_wp: 17.8 VALU instructions per MFMA,
// profiles/r1/conv_pmc_wp.txt), no LDS staging of A, no in_max scan.  That frees the registers for a three-deep
// software pipeline: the fragments of unit u + 2 are requested before the MFMAs of unit u run (conv_rows_wp: u + 1,
// and the wave-time ablations of profiles/wp_ablate_trace.sh put most of the small layers' 20 us in that dependent
// chain of load round trips).  Same units, same MFMA sequence per accumulator as conv_rows_wp on the same h / l
// pieces: bit-identical accumulators.
#ifndef CV_HL_ABL
#define CV_HL_ABL 0       // timing ablations of conv_hl (wrong results): 1 no gathers, 2 no MFMA, 4 no weight tile, 8 no epilogue, 16 no map reads
#endif
// NS = unit slots (registers for the A fragments and this thread's share of the weight tile, one LDS weight tile each):
// the loads of unit u + NS - 1 are requested while unit u multiplies.  Workgroups of the split coarse levels have no more
// than NS units: all their loads are in flight after the prologue (two dependent round trips - map entries, fragments -
// instead of one per unit).
// NW = waves per workgroup (4: 128 rows; 8: 256 rows, measured slower - profiles/r2/hl_nw8.txt).
// The unit walk is resolved once, before the loop: wave 0 compacts the workgroup's live (offset, chunk) units into a
// list in LDS, the loop is a counted loop over that list with per-thread invariants (weight-tile source / LDS offsets,
// 32-bit row offsets) hoisted - the first version spent ~150 vector and ~250 scalar instructions per unit on the walk
// (advance / skip_dead, 64-bit address arithmetic, spilled scalars) around 18 MFMAs.
#ifndef HL_CB_MIN_NB
#define HL_CB_MIN_NB 3
#endif
#ifndef HL_OCC1
#define HL_OCC1 4
#endif
#ifndef HL_OCC2
#define HL_OCC2 4
#endif
#ifndef HL_OCC3
#define HL_OCC3 4
#endif
// HL_COAL (round-3 experiment, OFF: measured slower; bit NB - 1 switches the NB x 32-column kernel): the row gather
// line-coalesced - 8 lanes fetch the 8 x 16-byte pieces of one row's 128-byte chunk (4 instructions x 8 rows) and a
// wave-private 4 KB LDS tile turns the [32 rows][128 B] image into the MFMA A layout (lane = row, 4 pieces per lane).
// Why it was tried (profiles/r3/gather_rate.txt): with lane = row every 16-byte access of a load instruction is its own
// line look-up in the vector cache and a pure gather tops out at 15 B/clk/CU (9.2 TB/s) even when everything hits the
// L2; line-coalesced it reaches 36 B/clk/CU (22 TB/s).  The piece a lane fetches is XOR-swizzled with (row >> 1) & 7,
// so the b128 stores (lane-contiguous) and the b128 reads (row stride 128 B) are conflict-free in the lane groups of
// ds_read_b128.  Same bytes into the same MFMAs: bit-identical results (82 network tests pass with it).  Measured
// (profiles/r3/hl_coal_ab.txt): net 2.445 -> 2.54 ms, 506 -> 475 scenes/s six in flight; on the 32 / 64-column kernels
// alone (occupancy unchanged) 2.56 ms: the vector cache's look-up rate is not what bounds conv_hl - the LDS round trip
// on every unit's critical path costs more than the faster gather returns.
#ifndef HL_COAL
#define HL_COAL 0
#endif
#ifndef HL_SETPRIO
#define HL_SETPRIO 0      // experiment: s_setprio 1 around a unit's MFMA cluster
#endif
constexpr int HL_MAX_UNITS = 512;      // live units of one workgroup: <= 10 offsets x Cin / 32 chunks + the second source's (host-checked)
// workgroups per CU the register allocation aims at (the second __launch_bounds__ argument; the waves-per-SIMD attribute
// restates it, because an explicit amdgpu_waves_per_eu replaces the bound __launch_bounds__ implies - a (1, 8) range
// silently cost the 96-column kernel a workgroup per CU for most of round 3)
constexpr int hl_blocks(int NB, int NS, int NW) {
    return NW > 4 ? (NS == 2 ? 2 : 1) : NS == 1 ? (NB == 1 ? 7 : 5) : NS == 2 ? (NB == 1 ? HL_OCC1 : NB == 2 ? HL_OCC2 : NB == 3 ? HL_OCC3 : 2) : 3;
}
template <int NB, int NS, int NW>
__global__ __attribute__((amdgpu_waves_per_eu(hl_blocks(NB, NS, NW) * NW / 4 > 8 ? 8 : hl_blocks(NB, NS, NW) * NW / 4, 8)))
__launch_bounds__(NW * 64, hl_blocks(NB, NS, NW)) void conv_hl(ConvArgs a) {
    static_assert(NS == 3 || NS == 2 || NS == 1, "three unit slots, two (the loads of unit k + 2 follow the MFMAs of unit k) or one (no prefetch: more workgroups per CU)");
    constexpr int TMv = NW * 32, THv = NW * 64;
    constexpr int B_BYTES = 2 * NB * 32 * 64, EP_BYTES = NW * 32 * EP_LD * 4;
    constexpr bool COAL = (HL_COAL >> (NB - 1)) & 1;
    constexpr int A_STAGE = COAL ? NW * 4096 : 0;                           // per wave: [32 rows][8 pieces x 16 B], swizzled
    constexpr int NT = NS == 1 ? 2 : NS;                                    // weight tiles in LDS (one register slot still alternates two)
    constexpr int SM_BYTES = NT * B_BYTES + A_STAGE > EP_BYTES ? NT * B_BYTES + A_STAGE : EP_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char sm[SM_BYTES];     // NS x weight tile [plane][col][64 B], the A tiles; then the epilogue tile
    __shared__ int rows_s[TMv];
    __shared__ int nbr_all[WP_NPRE + 1][TMv];
    __shared__ unsigned wave_mask[NW];
    __shared__ unsigned short units_s[HL_MAX_UNITS + 4];      // (jj << 8) | chunk of every live unit, in processing order; [HL_MAX_UNITS] = count
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * (NB * 32);
    const long long tile_id = xcd_tile(a);
    if (tile_id * TMv >= a.n_out) return;             // padding of the XCD-aware grid
    const int half = lane >> 5, l31 = lane & 31;
    const int nj = a.j_end - a.j_begin;
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
    const int njl = u_hi > u_lo ? j_last - j_first + 1 : 0;       // <= WP_NPRE (host)
    const int nch2 = a.in2 ? a.cin2 / KC : 0;

    // order and map entries in one phase (independent unless a single order comes without its map rows), one barrier
    const int* perm = a.row_perm ? a.row_perm + (a.perm_per_split ? (long long)blockIdx.z * a.n_out : 0) : nullptr;
    if (tid < TMv) {
        const long long t = tile_id * TMv + tid;
        const int row = t < a.n_out ? (perm ? perm[t] : (int)t) : -1;
        rows_s[tid] = row;
        if (a.in2) nbr_all[njl][tid] = row;
    }
    {
        unsigned m = 0u;
        const int jg0 = a.j_begin + (int)((long long)nj * blockIdx.z / a.splits);      // first offset of this mask group
        for (int e = tid; e < njl * TMv; e += THv) {
            const int jj = e / TMv, t = e - jj * TMv;
            const long long pos = tile_id * TMv + t;
            int v = -1;
            if (pos < a.n_out) {
                if (a.nbr_perm) {
                    v = a.nbr_perm[((long long)blockIdx.z * a.n_out + pos) * a.nbr_perm_w + (j_first + jj - jg0)];
                } else {
                    const int row = perm ? perm[pos] : (int)pos;
                    v = a.nbr ? a.nbr[(long long)row * a.K + j_first + jj] : row;
                }
                if (CV_HL_ABL & 16) v = (int)pos;
            }
            nbr_all[jj][t] = v;
            if (v >= 0) m |= 1u << jj;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m |= __shfl_xor(m, off);
        if (lane == 0) wave_mask[wave] = m;
    }
    __syncthreads();
    unsigned lm = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w) lm |= wave_mask[w];
    if (wave == 0) {
        // live units of the map in order, then the chunks of the second source dealt to this split
        const int n_first = u_hi - u_lo, n_second = nch2 > (int)blockIdx.z ? (nch2 - (int)blockIdx.z + a.splits - 1) / a.splits : 0;
        int cnt = 0;
        for (int base = 0; base < n_first + n_second; base += 64) {
            const int e = base + lane;
            int code = -1;
            if (e < n_first) {
                const int u = u_lo + e, q = u / nch;
                const int jj = a.j_begin + q - j_first;
                if ((lm >> jj) & 1u) code = (jj << 8) | (u - q * nch);
            } else if (e < n_first + n_second) {
                code = (njl << 8) | ((int)blockIdx.z + (e - n_first) * a.splits);
            }
            const unsigned long long bal = __ballot(code >= 0);
            if (code >= 0) units_s[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)code;
            cnt += __popcll(bal);
        }
        if (lane == 0) units_s[HL_MAX_UNITS] = (unsigned short)cnt;
    }
    __syncthreads();
    const int n_units = __builtin_amdgcn_readfirstlane((int)units_s[HL_MAX_UNITS]);
    if (n_units == 0 && a.gvalid) return;            // no row of the tile has a neighbour in this group: nothing to write (zskip)

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    // per-thread invariants of the weight tile: this thread's 16-byte pieces (source offset inside a unit's slab in 16-bit
    // words, or -1 beyond Cout; LDS offset with the XOR swizzle)
    constexpr int B_U4 = 2 * NB * 32 * 4;
    constexpr int B_PER = (B_U4 + THv - 1) / THv;
    int b_src[B_PER], b_dst[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int f = tid + i * THv;
        const int p = f / (NB * 32 * 4), rem = f - p * (NB * 32 * 4);
        const int col = rem >> 2, ch = rem & 3;
        b_src[i] = (f < B_U4 && n0 + col < a.cout) ? (p * a.cout + n0 + col) * 32 + ch * 8 : -1;
        b_dst[i] = f < B_U4 ? (p * NB * 32 + col) * 64 + ((ch ^ ((col >> 2) & 3)) << 4) : -1;
    }
    const unsigned slab_words = 2u * (unsigned)a.cout * 32u;       // one unit's slab: [plane][cout][32] 16-bit words
    const int my_row = wave * 32 + l31;
    const unsigned in_row_bytes = (unsigned)a.in_ld * 4u, in2_row_bytes = (unsigned)a.in2_ld * 4u;
    const unsigned char* const in_b = reinterpret_cast<const unsigned char*>(a.in);
    const unsigned char* const in2_b = reinterpret_cast<const unsigned char*>(a.in2);

    uint4 ra[NS][4], rb[NS][B_PER];
    bool live[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) live[q] = false;
    auto load = [&](auto S, int k) {                  // unit k of the list -> slot S
        constexpr int sl = decltype(S)::value;
        const int code = __builtin_amdgcn_readfirstlane((int)units_s[k]);
        const int jj = code >> 8, c = code & 255;
        const bool second = jj == njl;
        if constexpr (COAL) {
            // instruction q: rows (lane >> 3) + 8 q of the wave, lane & 7 = LDS slot of the row, slot ^ ((row >> 1) & 7) = piece
            const unsigned rb_ = second ? in2_row_bytes : in_row_bytes;
            const unsigned char* const base = (second ? in2_b : in_b) + (unsigned)(c * 128);
            bool any = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int src = nbr_all[jj][wave * 32 + (lane >> 3) + 8 * q];
                any |= src >= 0;
                if (src >= 0 && !(CV_HL_ABL & 1))
                    ra[sl][q] = *reinterpret_cast<const uint4*>(base + ((unsigned)src * rb_ + (unsigned)((((lane & 7) ^ (((lane >> 4) + 4 * q) & 7))) << 4)));
                else ra[sl][q] = make_uint4(0u, 0u, 0u, 0u);
            }
            live[sl] = __any(any);
        } else {
        const int src = nbr_all[jj][my_row];
        live[sl] = __any(src >= 0);
        if (src >= 0 && !(CV_HL_ABL & 1)) {
            const unsigned off = (unsigned)src * (second ? in2_row_bytes : in_row_bytes) + (unsigned)(c * 128 + half * 16);
            const unsigned char* p = (second ? in2_b : in_b) + off;
            ra[sl][0] = *reinterpret_cast<const uint4*>(p);
            ra[sl][1] = *reinterpret_cast<const uint4*>(p + 32);
            ra[sl][2] = *reinterpret_cast<const uint4*>(p + 64);
            ra[sl][3] = *reinterpret_cast<const uint4*>(p + 96);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) ra[sl][q] = make_uint4(0u, 0u, 0u, 0u);
        }
        }
        const unsigned short* slab = second ? a.wp6_2 + (size_t)c * slab_words
                                            : a.wp6 + (size_t)((j_first + jj) * nch + c) * slab_words;
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            rb[sl][i] = (b_src[i] >= 0 && !(CV_HL_ABL & 4)) ? *reinterpret_cast<const uint4*>(slab + b_src[i])
                                                            : make_uint4(0u, 0u, 0u, 0u);
    };
    auto stage_b = [&](auto S, int toff) {            // toff: byte offset of the LDS weight tile (slot * B_BYTES; NS == 1: (k & 1) * B_BYTES)
        constexpr int sl = decltype(S)::value;
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            if (b_dst[i] >= 0 && !(CV_HL_ABL & 4)) *reinterpret_cast<uint4*>(sm + toff + b_dst[i]) = rb[sl][i];
    };
    const int b_rd = l31 * 64;
    const int bswz = (l31 >> 2) & 3;
    auto compute = [&](auto S, int toff) {
        constexpr int sl = decltype(S)::value;
        const unsigned char* Bb = sm + toff + b_rd;
        uint4 (&fa)[4] = ra[sl];
        if constexpr (COAL) {
            // [32 rows][128 B] image of the wave's gathered chunk -> lane = row, pieces half, 2 + half, 4 + half, 6 + half,
            // back into the slot's own registers.  The tile is the wave's own: program order + the in-order LDS pipe are
            // the only synchronisation needed.
            unsigned char* st = sm + NT * B_BYTES + wave * 4096;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(st + q * 1024 + lane * 16) = ra[sl][q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned char* rd = st + l31 * 128;
            const int g = (l31 >> 1) & 7;
#pragma unroll
            for (int k = 0; k < 4; ++k) fa[k] = *reinterpret_cast<const uint4*>(rd + (((2 * k + half) ^ g) << 4));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (HL_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int piece = ((2 * ks + half) ^ bswz) << 4;
            const f16x8 a0 = __builtin_bit_cast(f16x8, fa[ks]), a1 = __builtin_bit_cast(f16x8, fa[2 + ks]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f16x8 b0 = *reinterpret_cast<const f16x8*>(Bb + nb * 32 * 64 + piece);
                const f16x8 b1 = *reinterpret_cast<const f16x8*>(Bb + (NB * 32 + nb * 32) * 64 + piece);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[nb], 0, 0, 0);
                if constexpr (NB >= HL_CB_MIN_NB && NS == 2) asm volatile("" ::: "memory");   // keep the next fragments' reads behind these MFMAs (registers)
            }
        }
        if (HL_SETPRIO) __builtin_amdgcn_s_setprio(0);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2;
    if constexpr (NS == 1) {
        // one register slot, two LDS tiles: nothing of unit k + 1 is requested before unit k has multiplied - the latency of a
        // unit's loads is covered by the OTHER workgroups of the CU (fewer registers: one more of them fits)
#pragma unroll 1
        for (int k = 0; k < n_units; ++k) {
            const int toff = (k & 1) * B_BYTES;
            load(S0{}, k);
            stage_b(S0{}, toff);
            __syncthreads();         // tile k visible; everyone is past the MFMAs of unit k - 2, whose tile unit k + ... reuses next
            if (live[0] && (!(CV_HL_ABL & 2) || a.acc_scale == 12345.f)) compute(S0{}, toff);
        }
    } else {
    if (n_units > 0) load(S0{}, 0);
    if (n_units > 1) load(S1{}, 1);
    if (n_units > 0) stage_b(S0{}, 0);
    // step: unit k in slot s.  barrier: tile k visible, tile k - 1 consumed by everyone (its slot takes the loads of unit
    // k + 2); the tile of unit k + 1 goes to LDS; MFMAs of unit k
    if constexpr (NS == 3) {
        auto step = [&](auto S, auto SN, auto SP, int k) {
            constexpr int sl = decltype(S)::value;
            __syncthreads();
            if (k + 1 < n_units) stage_b(SN, decltype(SN)::value * B_BYTES);
            if (k + 2 < n_units) load(SP, k + 2);
            if (live[sl] && (!(CV_HL_ABL & 2) || a.acc_scale == 12345.f)) compute(S, sl * B_BYTES);
        };
#pragma unroll 1
        for (int k = 0; k < n_units; k += 3) {
            step(S0{}, S1{}, S2{}, k);
            if (k + 1 >= n_units) break;
            step(S1{}, S2{}, S0{}, k + 1);
            if (k + 2 >= n_units) break;
            step(S2{}, S0{}, S1{}, k + 2);
        }
    } else {
        // two slots: the registers of unit k take the loads of unit k + 2 once its MFMAs are issued (fewer registers:
        // four workgroups per CU instead of three)
        auto step = [&](auto S, auto SN, int k) {
            constexpr int sl = decltype(S)::value;
            __syncthreads();
            if (k + 1 < n_units) stage_b(SN, decltype(SN)::value * B_BYTES);
            if (live[sl] && (!(CV_HL_ABL & 2) || a.acc_scale == 12345.f)) compute(S, sl * B_BYTES);
            if (k + 2 < n_units) load(S, k + 2);
        };
#pragma unroll 1
        for (int k = 0; k < n_units; k += 2) {
            step(S0{}, S1{}, k);
            if (k + 1 >= n_units) break;
            step(S1{}, S0{}, k + 1);
        }
    }
    }
    __syncthreads();                                 // weight tiles are dead: the epilogue tile reuses their LDS
    {
        const float sc = a.acc_scale_dev ? a.acc_scale * *a.acc_scale_dev : a.acc_scale;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] *= sc;
    }
    float (*ep)[EP_LD] = reinterpret_cast<float (*)[EP_LD]>(sm + wave * 32 * EP_LD * 4);
    if ((CV_HL_ABL & 8) && a.acc_scale != 12345.f) return;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) epilogue_store_wide(a, acc[nb], rows_s + wave * 32, n0 + nb * 32, lane, ep);
    if (a.splits > 1 && a.tickets) {
        // split-K without a second launch: every workgroup of an output tile publishes its partial tile, the LAST one to
        // arrive sums the partial tiles in split order and runs the epilogue.  Publish = write-through (sc1) stores in the
        // epilogue above -> per-wave vmcnt(0) -> barrier -> one relaxed agent-scope ticket; the last arriver's agent acquire
        // (an L1 invalidate) -> plain loads.  No release fence: the stores are already past the L2 (round 2 published with
        // plain stores + an agent release per workgroup and ran 50 % slower, profiles/r2/fused_finish.txt).
        // Summation order = conv_finish's (four running sums over the splits k % 4, then ((s0 + s1) + s2) + s3):
        // bit-identical to the two-launch path.  The last arriver leaves the counter at zero for the next launch.
        __shared__ int last_flag;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* ticket = a.tickets + tile_id * gridDim.y + blockIdx.y;
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = old == a.splits - 1;
            if (last_flag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (!last_flag) return;
        if (tid == 0) *ticket = 0;
        const long long plane = a.n_out * (long long)a.cout;       // one split's partial tile set
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = n0 + nb * 32 + (lane & 7) * 4;
            if (col >= a.cout) continue;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = rows_s[wave * 32 + (lane >> 3) + 8 * it];
                if (row < 0) continue;
                const float* p = a.partial + (long long)row * a.cout + col;
                float4 sq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) sq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int k0 = 0; k0 < a.splits; k0 += 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (k0 + q < a.splits) {
                            const float4 v = *reinterpret_cast<const float4*>(p + (long long)(k0 + q) * plane);
                            sq[q].x += v.x; sq[q].y += v.y; sq[q].z += v.z; sq[q].w += v.w;
                        }
                }
                const float4 x = make_float4(sq[0].x + sq[1].x + sq[2].x + sq[3].x, sq[0].y + sq[1].y + sq[2].y + sq[3].y,
                                             sq[0].z + sq[1].z + sq[2].z + sq[3].z, sq[0].w + sq[1].w + sq[2].w + sq[3].w);
                epilogue_apply4(a, row, col, x);
            }
        }
    }
}

// ------------------------------------------------------------------ hl convolution fed by LDS-DMA rings (round 4)
// What the timing ablations of round 4 say binds the throughput with eight scenes in flight (profiles/r4/
// throughput_ablations.txt): conv_hl's row gathers (21 % of the scene rate: the vector cache's address pipe takes one
// lane per clock when every lane of a load sits on its own 128-byte line - 187 M lane-loads per scene = 0.30 ms of every
// CU), its weight tiles (0.21 ms) and the partial-tile machinery - not the matrix pipe (7 %).  conv_hd keeps conv_hl's
// units, MFMA sequence and epilogue (bit-identical accumulators) and changes how the operands travel:
//  * 8 waves x 32 rows = 256 rows per workgroup: a unit's weight tile is fetched once per 256 rows instead of per 128;
//  * both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, asynchronous): the gathered rows
//    line-coalesced - 8 lanes fetch the 8 x 16-byte pieces of one row's 128-byte chunk, 8 rows per instruction - into a
//    wave-private [32 rows][128 B] image whose pieces are XOR-swizzled on the SOURCE side (the DMA writes lane-linear) so
//    that the b128 fragment reads are conflict-free; the weight tile in the layout conv_hl stages by hand;
//  * a ring of three stages (3 x 44 KB at 96 columns): the operands of units k + 1 and k + 2 are in flight while unit k
//    multiplies; one workgroup barrier per unit (weight tile visible + the stage of unit k - 1 free), waits by counted
//    vmcnt (the LDS-DMA instructions a wave issued for the units behind the one it needs stay in flight);
//  * the accumulators are the only long-lived registers: one workgroup of 8 waves per CU (LDS-bound), two waves per SIMD.
// Dead rows of a live wave fetch a row of zeros (one line, all lanes); dead waves issue nothing for the unit.
#ifndef CV_HD_ABL
#define CV_HD_ABL 0
#endif
// LDS of one workgroup: NSTG ring stages (gathered rows of NW waves + the weight tile), the tile's row / map tables, unit list
#ifndef HD_LDS_PAD
#define HD_LDS_PAD 0        // experiment: LDS the workgroup does not use, to control what else fits on its CU (profiles/r5/hd_lds_pad.txt)
#endif
constexpr int hd_lds_bytes(int NB, int NW, int NSTG) {
    return NSTG * (NW * 4096 + 2 * NB * 32 * 64) + NW * 32 * 4 + (WP_NPRE + 1) * NW * 32 * 4 + NW * 4 + (HL_MAX_UNITS + 4) * 2 +
           ((NB == 3 && NW == 8 && NSTG == 2) ? HD_LDS_PAD : 0);
}
constexpr int hd_blocks(int NB, int NW, int NSTG) {          // workgroups per CU (LDS-bound), at most 8 waves per SIMD
    const int b = (160 * 1024) / hd_lds_bytes(NB, NW, NSTG);
    return b * NW > 32 ? 32 / NW : (b < 1 ? 1 : b);
}
// NW waves x 32 rows per workgroup, NSTG ring stages: <8, 3> one workgroup per CU with the requests of two units in flight
// behind the one that multiplies; <4, 2> two workgroups per CU, one unit of prefetch each
// All sixteen fragment reads of a 96-column unit in one go (no wait inside), then two waits that name the registers they
// release: the second k-step's fragments travel while the first one multiplies, and the requests of the next unit are
// issued while the first ones travel (conv_hd, HD_EARLY).
#ifndef HD_EARLY
#define HD_EARLY 0      // measured (profiles/r4/hd2_grid.txt): 562-565 scenes/s with it against 573-575 without
#endif
__device__ __forceinline__ void hd_reads3_issue(const unsigned (&aa)[4], const unsigned (&ab)[2], u32x4v (&A)[4],
                                                u32x4v (&B0)[2][3], u32x4v (&B1)[2][3]) {
    constexpr int P1 = 3 * 32 * 64;
    asm volatile("ds_read_b128 %0, %16\n\tds_read_b128 %1, %17\n\t"
                 "ds_read_b128 %2, %20\n\tds_read_b128 %3, %20 offset:%22\n\t"
                 "ds_read_b128 %4, %20 offset:%23\n\tds_read_b128 %5, %20 offset:%24\n\t"
                 "ds_read_b128 %6, %20 offset:%25\n\tds_read_b128 %7, %20 offset:%26\n\t"
                 "ds_read_b128 %8, %18\n\tds_read_b128 %9, %19\n\t"
                 "ds_read_b128 %10, %21\n\tds_read_b128 %11, %21 offset:%22\n\t"
                 "ds_read_b128 %12, %21 offset:%23\n\tds_read_b128 %13, %21 offset:%24\n\t"
                 "ds_read_b128 %14, %21 offset:%25\n\tds_read_b128 %15, %21 offset:%26"
                 : "=&v"(A[0]), "=&v"(A[2]), "=&v"(B0[0][0]), "=&v"(B1[0][0]), "=&v"(B0[0][1]), "=&v"(B1[0][1]), "=&v"(B0[0][2]),
                   "=&v"(B1[0][2]), "=&v"(A[1]), "=&v"(A[3]), "=&v"(B0[1][0]), "=&v"(B1[1][0]), "=&v"(B0[1][1]), "=&v"(B1[1][1]),
                   "=&v"(B0[1][2]), "=&v"(B1[1][2])
                 : "v"(aa[0]), "v"(aa[2]), "v"(aa[1]), "v"(aa[3]), "v"(ab[0]), "v"(ab[1]), "i"(P1), "i"(2048), "i"(P1 + 2048),
                   "i"(4096), "i"(P1 + 4096)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void hd_reads3_wait(u32x4v& a0, u32x4v& a1, u32x4v (&b0)[3], u32x4v (&b1)[3]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a0), "+v"(a1), "+v"(b0[0]), "+v"(b1[0]), "+v"(b0[1]), "+v"(b1[1]), "+v"(b0[2]), "+v"(b1[2])
                 : "n"(N) : "memory");
}
template <int NB, int NW, int NSTG>
__global__ __launch_bounds__(NW * 64, (hd_blocks(NB, NW, NSTG) * NW + 3) / 4) void conv_hd(ConvArgs a) {
    static_assert(NSTG == 2 || NSTG == 3, "two or three ring stages");
    constexpr int TMv = NW * 32, THv = NW * 64;
    constexpr int A_BYTES = NW * 4096, B_BYTES = 2 * NB * 32 * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int B_INSTR = B_BYTES / 1024;                         // 1 KB (64 lanes x 16 B) per LDS-DMA instruction
    constexpr int B_PER_WAVE = (B_INSTR + NW - 1) / NW;
    constexpr int EP_BYTES = NW * 32 * EP_LD * 4;
    static_assert(EP_BYTES <= NSTG * STAGE, "the epilogue tile aliases the ring");
    constexpr int OFF_ROWS = NSTG * STAGE, OFF_NBR = OFF_ROWS + TMv * 4, OFF_MASK = OFF_NBR + (WP_NPRE + 1) * TMv * 4,
                  OFF_UNITS = OFF_MASK + NW * 4,
                  LDS_TOTAL = OFF_UNITS + (HL_MAX_UNITS + 4) * 2 + ((NB == 3 && NW == 8 && NSTG == 2) ? HD_LDS_PAD : 0);
    static_assert(LDS_TOTAL == hd_lds_bytes(NB, NW, NSTG) && LDS_TOTAL <= 160 * 1024, "LDS budget");
    // ONE __shared__ object (a second one makes hipcc drain vmcnt in front of the LDS reads of an LDS-DMA pipeline)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_TOTAL];
    unsigned char* const sm = lds;
    int* const rows_s = reinterpret_cast<int*>(lds + OFF_ROWS);
    int (*const nbr_all)[TMv] = reinterpret_cast<int (*)[TMv]>(lds + OFF_NBR);
    unsigned* const wave_mask = reinterpret_cast<unsigned*>(lds + OFF_MASK);
    unsigned short* const units_s = reinterpret_cast<unsigned short*>(lds + OFF_UNITS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * (NB * 32);
    const long long tile_id = xcd_tile(a);
    const int half = lane >> 5, l31 = lane & 31;
    const int nj = a.j_end - a.j_begin;
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
    const int njl = u_hi > u_lo ? j_last - j_first + 1 : 0;       // <= WP_NPRE (host)
    const int nch2 = a.in2 ? a.cin2 / KC : 0;

    // ---- tile set-up as conv_hl: processing order, map entries of the workgroup's offsets, live-unit list
    const int* perm = a.row_perm ? a.row_perm + (a.perm_per_split ? (long long)blockIdx.z * a.n_out : 0) : nullptr;
    if (tid < TMv) {
        const long long t = tile_id * TMv + tid;
        const int row = t < a.n_out ? (perm ? perm[t] : (int)t) : -1;
        rows_s[tid] = row;
        if (a.in2) nbr_all[njl][tid] = row;
    }
    {
        unsigned m = 0u;
        const int jg0 = a.j_begin + (int)((long long)nj * blockIdx.z / a.splits);      // first offset of this mask group
        for (int e = tid; e < njl * TMv; e += THv) {
            const int jj = e / TMv, t = e - jj * TMv;
            const long long pos = tile_id * TMv + t;
            int v = -1;
            if (pos < a.n_out) {
                if (a.nbr_perm) {
                    v = a.nbr_perm[((long long)blockIdx.z * a.n_out + pos) * a.nbr_perm_w + (j_first + jj - jg0)];
                } else {
                    const int row = perm ? perm[pos] : (int)pos;
                    v = a.nbr ? a.nbr[(long long)row * a.K + j_first + jj] : row;
                }
            }
            nbr_all[jj][t] = v;
            if (v >= 0) m |= 1u << jj;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m |= __shfl_xor(m, off);
        if (lane == 0) wave_mask[wave] = m;
    }
    __syncthreads();
    unsigned lm = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w) lm |= wave_mask[w];
    if (wave == 0) {
        const int n_first = u_hi - u_lo, n_second = nch2 > (int)blockIdx.z ? (nch2 - (int)blockIdx.z + a.splits - 1) / a.splits : 0;
        int cnt = 0;
        for (int base = 0; base < n_first + n_second; base += 64) {
            const int e = base + lane;
            int code = -1;
            if (e < n_first) {
                const int u = u_lo + e, q = u / nch;
                const int jj = a.j_begin + q - j_first;
                if ((lm >> jj) & 1u) code = (jj << 8) | (u - q * nch);
            } else if (e < n_first + n_second) {
                code = (njl << 8) | ((int)blockIdx.z + (e - n_first) * a.splits);
            }
            const unsigned long long bal = __ballot(code >= 0);
            if (code >= 0) units_s[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)code;
            cnt += __popcll(bal);
        }
        if (lane == 0) units_s[HL_MAX_UNITS] = (unsigned short)cnt;
    }
    __syncthreads();
    const int n_units = (CV_HD_ABL & 8) ? 0 : __builtin_amdgcn_readfirstlane((int)units_s[HL_MAX_UNITS]);      // (8: no unit loop)
    if (n_units == 0 && a.gvalid && !(CV_HD_ABL & 8)) return;     // no row of the tile has a neighbour in this group (zskip)

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    // ---- per-thread invariants of the LDS-DMA requests
    // weight tile: instruction t (1 KB) covers pieces f = t * 64 + lane of the [plane][col][4 slots] image; slot s of column
    // col holds the slab's 16-byte piece s ^ ((col >> 2) & 3) (the swizzle conv_hl applies when it stages by hand)
    int b_src[B_PER_WAVE];                                          // offset inside a unit's slab in 16-bit words
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
        const int t = wave + i * NW;
        const int f = t * 64 + lane;
        const int pl = f / (NB * 32 * 4), rem = f - pl * (NB * 32 * 4);
        const int col = rem >> 2, slot = rem & 3;
        const int gc = min(n0 + col, a.cout - 1);                   // columns beyond Cout are never stored
        b_src[i] = (pl * a.cout + gc) * 32 + ((slot ^ ((col >> 2) & 3)) << 3);
    }
    const unsigned slab_words = 2u * (unsigned)a.cout * 32u;
    const unsigned in_row_bytes = (unsigned)a.in_ld * 4u, in2_row_bytes = (unsigned)a.in2_ld * 4u;
    const unsigned char* const in_b = reinterpret_cast<const unsigned char*>(a.in);
    const unsigned char* const in2_b = reinterpret_cast<const unsigned char*>(a.in2);
    // gathered rows: instruction q covers rows 8 q + (lane >> 3) of the wave, LDS slot lane & 7 of the row receives its
    // piece (lane & 7) ^ ((row >> 1) & 7)
    const int a_row = lane >> 3;
    unsigned a_piece[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a_piece[q] = (unsigned)(((lane & 7) ^ (((8 * q + a_row) >> 1) & 7)) << 4);

    bool live[NSTG];
#pragma unroll
    for (int q = 0; q < NSTG; ++q) live[q] = false;

    // requests of unit k into ring stage S; returns the number of LDS-DMA instructions this wave issued
    auto issue = [&](auto S, int k) -> int {
        constexpr int st = decltype(S)::value;
        const int code = __builtin_amdgcn_readfirstlane((int)units_s[k]);
        const int jj = code >> 8, c = code & 255;
        const bool second = jj == njl;
        int n_issued = 0;
        int src[4];
        bool any = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            src[q] = nbr_all[jj][wave * 32 + 8 * q + a_row];
            any |= src[q] >= 0;
        }
        const bool lv = __any(any);
        live[st] = lv;
        if (lv && !(CV_HD_ABL & 1)) {
            const unsigned rb_ = second ? in2_row_bytes : in_row_bytes;
            const unsigned char* const base = (second ? in2_b : in_b) + (unsigned)(c * 128);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned char* g = src[q] >= 0 ? base + ((unsigned)src[q] * rb_ + a_piece[q]) : g_zero_chunk + a_piece[q];
                lds_dma16(g, sm + st * STAGE + wave * 4096 + q * 1024);
            }
            n_issued += 4;
        }
        const unsigned short* slab = second ? a.wp6_2 + (size_t)c * slab_words
                                            : a.wp6 + (size_t)((j_first + jj) * nch + c) * slab_words;
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; ++i) {
            const int t = wave + i * NW;                            // wave-uniform
            if (t < B_INSTR && !(CV_HD_ABL & 4)) {
                lds_dma16(slab + b_src[i], sm + st * STAGE + A_BYTES + t * 1024);
                ++n_issued;
            }
        }
        return n_issued;
    };
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)lds;
    const int bswz = (l31 >> 2) & 3, a_g = (l31 >> 1) & 7;
    unsigned a_off[4], b_off[2];                                    // byte addresses inside stage 0
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) a_off[kk] = lds0 + (unsigned)(wave * 4096 + l31 * 128 + (((2 * kk + half) ^ a_g) << 4));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_off[ks] = lds0 + (unsigned)(A_BYTES + l31 * 64 + (((2 * ks + half) ^ bswz) << 4));
    auto compute = [&](auto S) {
        constexpr int st = decltype(S)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // pieces of the 128-byte chunk: 0-3 = high halves of channels 0-31, 4-7 = low halves; k-step ks takes
            // channels 16 ks ... 16 ks + 15: high piece 2 ks + half, low piece 4 + 2 ks + half
            const unsigned aa0 = a_off[ks] + st * STAGE, aa1 = a_off[2 + ks] + st * STAGE, ab = b_off[ks] + st * STAGE;
            u32x4v A0, A1, B0[NB], B1[NB];
            hd_read_frags<NB>(aa0, aa1, ab, A0, A1, B0, B1);
            const f16x8 a0 = __builtin_bit_cast(f16x8, A0), a1 = __builtin_bit_cast(f16x8, A1);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f16x8 b0 = __builtin_bit_cast(f16x8, B0[nb]), b1 = __builtin_bit_cast(f16x8, B1[nb]);
                if (!(CV_HD_ABL & 2)) {
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[nb], 0, 0, 0);
                }
            }
        }
    };
    auto reads_early = [&](auto S, u32x4v (&A)[4], u32x4v (&B0)[2][3], u32x4v (&B1)[2][3]) {
        constexpr int st = decltype(S)::value;
        const unsigned aa[4] = {a_off[0] + st * STAGE, a_off[1] + st * STAGE, a_off[2] + st * STAGE, a_off[3] + st * STAGE};
        const unsigned ab[2] = {b_off[0] + st * STAGE, b_off[1] + st * STAGE};
        if constexpr (NB == 3) hd_reads3_issue(aa, ab, A, B0, B1);
    };
    auto mfma_early = [&](u32x4v (&A)[4], u32x4v (&B0)[2][3], u32x4v (&B1)[2][3]) {
        if constexpr (NB == 3) {
            hd_reads3_wait<8>(A[0], A[2], B0[0], B1[0]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 1) hd_reads3_wait<0>(A[1], A[3], B0[1], B1[1]);
                const f16x8 a0 = __builtin_bit_cast(f16x8, A[ks]), a1 = __builtin_bit_cast(f16x8, A[2 + ks]);
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) {
                    const f16x8 b0 = __builtin_bit_cast(f16x8, B0[ks][nb]), b1 = __builtin_bit_cast(f16x8, B1[ks][nb]);
                    if (!(CV_HD_ABL & 2)) {
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[nb], 0, 0, 0);
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[nb], 0, 0, 0);
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[nb], 0, 0, 0);
                    }
                }
            }
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2;
    // step k (stage S = k % 3): this wave's requests of unit k have landed (those of unit k + 1 may stay in flight);
    // barrier = every wave's share of weight tile k is visible AND everyone is past the MFMAs of unit k - 1, whose stage
    // takes the requests of unit k + 2; then the MFMAs of unit k
    if constexpr (NSTG == 3) {
        int pend1 = 0;                                     // LDS-DMA instructions this wave has in flight for the NEXT unit
        if (n_units > 0) issue(S0{}, 0);
        if (n_units > 1) pend1 = issue(S1{}, 1);
        // (pend1 at the wait of step k counts the requests of unit k + 1: it was set by the issue of step k - 1)
        auto step = [&](auto S, auto SP, int k) {
            constexpr int st = decltype(S)::value;
            wait_vmcnt_dyn(__builtin_amdgcn_readfirstlane(k + 1 < n_units ? pend1 : 0));
            __builtin_amdgcn_s_barrier();
            if constexpr (NB == 3 && HD_EARLY) {
                u32x4v A[4], B0[2][3], B1[2][3];
                if (live[st]) reads_early(S, A, B0, B1);
                if (k + 2 < n_units) pend1 = issue(SP, k + 2);
                else pend1 = 0;
                if (live[st]) mfma_early(A, B0, B1);
            } else {
                if (k + 2 < n_units) pend1 = issue(SP, k + 2);
                else pend1 = 0;
                if (live[st]) compute(S);
            }
        };
#pragma unroll 1
        for (int k = 0; k < n_units; k += 3) {
            step(S0{}, S2{}, k);
            if (k + 1 >= n_units) break;
            step(S1{}, S0{}, k + 1);
            if (k + 2 >= n_units) break;
            step(S2{}, S1{}, k + 2);
        }
    } else {
        // two stages: only unit k is in flight at its wait; the requests of unit k + 1 go out behind the barrier
        if (n_units > 0) issue(S0{}, 0);
        auto step = [&](auto S, auto SN, int k) {
            constexpr int st = decltype(S)::value;
            wait_vmcnt_le<0>();
            __builtin_amdgcn_s_barrier();
            if constexpr (NB == 3 && HD_EARLY) {
                u32x4v A[4], B0[2][3], B1[2][3];
                if (live[st]) reads_early(S, A, B0, B1);
                if (k + 1 < n_units) issue(SN, k + 1);
                if (live[st]) mfma_early(A, B0, B1);
            } else {
                if (k + 1 < n_units) issue(SN, k + 1);
                if (live[st]) compute(S);
            }
        };
#pragma unroll 1
        for (int k = 0; k < n_units; k += 2) {
            step(S0{}, S1{}, k);
            if (k + 1 >= n_units) break;
            step(S1{}, S0{}, k + 1);
        }
    }
    wait_vmcnt_le<0>();
    __syncthreads();                                 // the ring is dead: the epilogue tile reuses its LDS
    {
        const float sc = a.acc_scale_dev ? a.acc_scale * *a.acc_scale_dev : a.acc_scale;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] *= sc;
    }
    float (*ep)[EP_LD] = reinterpret_cast<float (*)[EP_LD]>(sm + wave * 32 * EP_LD * 4);
    if ((CV_HD_ABL & 16) && a.acc_scale != 12345.f) return;      // (16: no epilogue)
    ConvArgs ae = a;
    ae.tickets = nullptr;                            // (the in-launch split-K reduction is conv_hl's)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) epilogue_store_wide(ae, acc[nb], rows_s + wave * 32, n0 + nb * 32, lane, ep);
}

// (Round 5, measured and removed - git 2945191 holds the code: `conv_hh`, conv_hd with 16-channel half-chunk ring stages so that two
// 8-wave workgroups share a CU (bit-identical; ts1 96->96 98.5 -> 96.0 us, ts2 61.5 -> 73.9 us per layer, 575 -> 549-559 scenes/s),
// and `group_fused_tail`, the mask groups summed inside the conv_hd launch by the last group's workgroups instead of a finish
// launch (bit-identical; one scene in flight +0.5-1 %, 576 -> 544-566 scenes/s with seven: its prefetches take 246 VGPRs and the
// polling workgroups hold their CU).  LABNOTES "Round 5".)

// ------------------------------------------------------------------ stem on the matrix cores (fp16 pairs)
// conv0p1s1 (utils/minkunet.py:53: 5x5x5, 3 or 6 -> 32 channels): 13 % of the 125 (row, offset) pairs exist.  Rounds 1-5 also
// kept a lane-per-row stem on the vector ALUs (one lane = one output row and all 32 outputs in registers, weights through the
// scalar cache; 93-107 us at 80k rows, removed in round 6: the training forward takes this kernel too, other fp32 callers the
// generic conv_rows): it executed 32*CIN FMAs for every offset ANY of a wave's 64 rows had (~90 % of the 125).  As a GEMM over the gathered operand
// [rows][CIN * 128] (offsets padded to 128, dead entries zero) it is 1.97 GFLOP dense - nothing for the matrix cores -
// and what remains is what the stem really is: 125 map entries and ~16 gathers of CIN floats per row.
// A wave owns 32 rows; per group of 16 offsets a lane (row = lane % 32, half = lane / 32) reads its row's 8 map
// entries, gathers the CIN floats of the existing ones, splits them into fp16 pairs (split2h) and feeds CIN x 3
// v_mfma_f32_32x32x16_f16 (k = the 16 offsets of the group, one MFMA triple per input channel).  The weights
// (cv_sp_pack_weights_stem_h2_f32: BatchNorm scale and a power of two folded in, fp16 pairs in B-operand order,
// 16 KB per input channel) sit in LDS for the whole workgroup.  Same products as the fp32 chain to 2^-22, fp32 accumulation.
template <int CIN>
__global__ __launch_bounds__(THREADS, (CIN <= 3 ? 3 : 1)) void conv_stem_mfma(ConvArgs a) {
    constexpr int G = 8;                              // groups of 16 kernel offsets (K <= 128)
    constexpr int W_BYTES = G * CIN * 2 * 1024;       // [group][channel][plane][col][half][8] fp16
    constexpr int EP_BYTES = 4 * 32 * EP_LD * 4;
    constexpr int SM_BYTES = W_BYTES > EP_BYTES ? W_BYTES : EP_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char w_s[SM_BYTES];    // the weights; then the epilogue tile
    __shared__ int rows_s[TM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const long long r0 = (long long)blockIdx.x * TM;
    for (int f = tid; f < W_BYTES / 16; f += THREADS)
        reinterpret_cast<uint4*>(w_s)[f] = reinterpret_cast<const uint4*>(a.wp6)[f];
    if (tid < TM) rows_s[tid] = r0 + tid < a.n_out ? (int)(r0 + tid) : -1;
    const long long row = r0 + wave * 32 + l31;
    const bool have = row < a.n_out;
    const int K = a.K;
    const int* __restrict__ mrow = a.nbr + (have ? row : 0) * (long long)K;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float in_max = 0.f;
    // every map entry of the lane's row half first (64 independent loads in flight: the map is streamed from HBM / the
    // Infinity Cache once), then the groups with the gathers one group ahead of the MFMAs
    int m[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = g * 16 + half * 8 + i;
            m[g][i] = (have && j < K) ? mrow[j] : -1;
        }
    float x0[8][CIN], x1[8][CIN];
    auto gather = [&](const int* mg, float (*x)[CIN]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (mg[i] >= 0) {
                const float* p = a.in + (long long)mg[i] * a.in_ld;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) x[i][ci] = p[ci];
            } else {
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) x[i][ci] = 0.f;
            }
        }
    };
    gather(m[0], x1);
    __syncthreads();                                  // weights in LDS
#pragma unroll
    for (int g = 0; g < G; ++g) {
        bool any = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) x0[i][ci] = x1[i][ci];
            any |= m[g][i] >= 0;
        }
        if (g + 1 < G) gather(m[g + 1], x1);
        if (__any(any)) {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                unsigned h[4], l[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    in_max = fmaxf(in_max, fmaxf(fabsf(x0[2 * q][ci]), fabsf(x0[2 * q + 1][ci])));
                    split2h(x0[2 * q][ci], x0[2 * q + 1][ci], h[q], l[q]);
                }
                const f16x8 a0 = __builtin_bit_cast(f16x8, make_uint4(h[0], h[1], h[2], h[3]));
                const f16x8 a1 = __builtin_bit_cast(f16x8, make_uint4(l[0], l[1], l[2], l[3]));
                const unsigned char* wb = w_s + ((g * CIN + ci) * 2) * 1024 + (l31 * 2 + half) * 16;
                const f16x8 b0 = *reinterpret_cast<const f16x8*>(wb);
                const f16x8 b1 = *reinterpret_cast<const f16x8*>(wb + 1024);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
            }
        }
    }
    if (in_max > 65000.f && a.range_flag) *a.range_flag = 1;
    {
        const float sc = a.acc_scale_dev ? a.acc_scale * *a.acc_scale_dev : a.acc_scale;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= sc;
    }
    __syncthreads();                                  // the weights are dead: the epilogue tile reuses their LDS
    float (*ep)[EP_LD] = reinterpret_cast<float (*)[EP_LD]>(w_s + wave * 32 * EP_LD * 4);
    epilogue_store_wide(a, acc, rows_s + wave * 32, 0, lane, ep);
}

// weights of conv_stem_mfma: [K][cin][32] fp32 -> per (group of 16 offsets, input channel): two planes (h, l) of
// [col][half][8] fp16 = the B operand of v_mfma_f32_32x32x16_f16 with k = offset inside the group; offsets >= K are zero
__global__ __launch_bounds__(256) void pack_weights_stem_h2(const float* __restrict__ w, int K, int cin,
                                                            const float* __restrict__ col_scale, float mult,
                                                            unsigned short* __restrict__ wp) {
    const int total = 8 * cin * 32 * 2 * 4;           // (group, channel, col, half, pair of offsets)
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        int r = t;
        const int q = r % 4; r /= 4;
        const int hf = r % 2; r /= 2;
        const int col = r % 32; r /= 32;
        const int ci = r % cin; r /= cin;
        const int g = r;
        const int j0 = g * 16 + hf * 8 + 2 * q;
        const float sc = (col_scale ? col_scale[col] : 1.f) * mult;
        const float v0 = j0 < K ? w[((long long)j0 * cin + ci) * 32 + col] * sc : 0.f;
        const float v1 = j0 + 1 < K ? w[((long long)(j0 + 1) * cin + ci) * 32 + col] * sc : 0.f;
        unsigned h, l;
        split2h(v0, v1, h, l);
        const long long base = ((((long long)(g * cin + ci) * 2) * 32 + col) * 2 + hf) * 8 + 2 * q;
        *reinterpret_cast<unsigned*>(wp + base) = h;
        *reinterpret_cast<unsigned*>(wp + base + 512) = l;
    }
}

// ------------------------------------------------------------------ backward: weight gradient
// dW[j][ci][co] = sum_u x[nbr[u][j]][ci] * dy[u][co]   (the reduction runs over the output rows).
// One wave owns (offset j, 32 input channels, NB*32 output channels, a contiguous range of rows) and
// accumulates a 32 x NB*32 tile on the fp32 matrix cores: per MFMA step two rows u0,u1 are consumed -
// A operand lane l = x[nbr[u_{l>>5}][j]][ci0 + (l&31)], B operand lane l = dy[u_{l>>5}][co0 + nb*32 + (l&31)],
// both coalesced 128-byte row segments straight from L2.  Steps whose two rows both miss the neighbour
// are skipped (wave-uniform branch).  Partial tiles go to ws[split][K][cin][cout]; wgrad_reduce sums them.
// Launch plan of one weight-gradient call: offsets in heavy-first order (the centre offset pairs every row, face
// neighbours most, corners few - longest tasks are dispatched first) and a per-offset number of row splits so the
// tasks carry comparable numbers of pairs.  Partial tile of (offset j, split s) = slot first[j] + s.
constexpr int WG_MAX_K = 128;
struct WgradPlan {
    short order[WG_MAX_K];        // rank -> offset
    short nsplit[WG_MAX_K];       // by offset
    int first[WG_MAX_K];          // by offset: first partial slot
    int task_end[WG_MAX_K];       // by rank: cumulative (splits x ci blocks x co blocks)
};

// X6: the 16 rows of a step group are ONE K = 16 bf16 MFMA group: the eight values a lane holds for the steps t = 0..7
// (row 2t + half) are exactly its eight k slots (k = 8 * half + t, the same rows on the A and the B side), so the
// fp32 operands are split into bf16 triples in registers and six piece products replace eight fp32 MFMAs per nb
// (see conv_rows_wp; gradients keep the fp32 exponent range, which fp16 pairs would not).
// NA x NB blocks of 32 x 32 per wave: a wave that owns NA input-channel blocks reads each dy row segment once for all
// of them (and each x segment once for all NB output blocks).  With one input block per wave (the first version) a
// 96 -> 96 convolution moved 1536 bytes per (input, output) pair from L2 for three tiles - the kernel ran at the L2
// bandwidth (6.3 TB/s), not at the matrix rate; 3 x 3 blocks move 768 bytes for nine tiles.
template <int NA, int NB, int PIECES>      // PIECES 0: fp32 MFMA; 3: bf16 triples, six piece products; 1: operands rounded to bf16, one product
__global__ __launch_bounds__(THREADS, (NA * NB >= 8 ? 2 : 1)) void conv_wgrad(const float* __restrict__ x, int x_ld, int cin,
                                                      const float* __restrict__ dy, int dy_ld, int cout,
                                                      const int* __restrict__ nbr, int K, long long n_out,
                                                      WgradPlan plan, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ci_blocks = (cin + NA * 32 - 1) / (NA * 32), co_blocks = (cout + NB * 32 - 1) / (NB * 32);
    int task = blockIdx.x * 4 + wave;
    if (task >= plan.task_end[K - 1]) return;
    int rank = 0;
    while (task >= plan.task_end[rank]) ++rank;                            // wave-uniform, K <= 128 entries
    if (rank) task -= plan.task_end[rank - 1];
    const int j = plan.order[rank], row_splits = plan.nsplit[j];
    const int split = task % row_splits; task /= row_splits;
    const int cob = task % co_blocks;
    const int cib = task / co_blocks;
    const long long r_lo = n_out * split / row_splits, r_hi = n_out * (split + 1) / row_splits;
    const int half = lane >> 5, l31 = lane & 31;
    const int ci0 = cib * NA * 32, co0 = cob * NB * 32;
    f32x16 acc[NA][NB];
#pragma unroll
    for (int na = 0; na < NA; ++na)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[na][nb][r] = 0.f;
    // 64 output rows per batch: every lane fetches one neighbour index (the next batch's is already in
    // flight), the rows that HAVE the neighbour are taken two at a time from the ballot mask (work
    // proportional to the existing pairs), and the operand loads of up to STEPS MFMA steps are issued back
    // to back before the matrix cores consume them.
    constexpr int STEPS = 8;
    auto fetch = [&](long long u0) -> int {
        const long long mu = u0 + lane;
        return mu < r_hi ? (nbr ? nbr[mu * K + j] : (int)mu) : -1;
    };
    int src_next = fetch(r_lo);
    for (long long u0 = r_lo; u0 < r_hi; u0 += 64) {
        const int src_l = src_next;
        src_next = fetch(u0 + 64);
        unsigned long long m = __ballot(src_l >= 0);
        while (m) {
            float av[STEPS][NA], bv[STEPS][NB];
            int nsteps = 0;
#pragma unroll
            for (int t = 0; t < STEPS; ++t) {
#pragma unroll
                for (int na = 0; na < NA; ++na) av[t][na] = 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bv[t][nb] = 0.f;
                if (m) {                                                   // wave-uniform
                    const int ra = __ffsll((unsigned long long)m) - 1;
                    m &= m - 1;
                    int rb = -1;
                    if (m) { rb = __ffsll((unsigned long long)m) - 1; m &= m - 1; }
                    const int r = half ? rb : ra;                          // this half-wave's row of the pair
                    // shuffle with ALL lanes active: ds_bpermute returns 0 for an inactive source lane
                    const int got = __shfl(src_l, r >= 0 ? r : 0);
                    const int src = r >= 0 ? got : -1;
                    if (src >= 0) {
#pragma unroll
                        for (int na = 0; na < NA; ++na) {
                            const int ci = ci0 + na * 32 + l31;
                            if (ci < cin) av[t][na] = x[(long long)src * x_ld + ci];
                        }
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const int col = co0 + nb * 32 + l31;
                            if (col < cout) bv[t][nb] = dy[(u0 + r) * dy_ld + col];
                        }
                    }
                    nsteps = t + 1;
                }
            }
            if constexpr (PIECES == 1) {
                static_assert(STEPS == 8, "one bf16 MFMA group = 8 k slots per lane");
                bf16x8 a1[NA];
#pragma unroll
                for (int na = 0; na < NA; ++na)
                    a1[na] = __builtin_bit_cast(
                        bf16x8, make_uint4(cvt_pk_bf16(av[0][na], av[1][na]), cvt_pk_bf16(av[2][na], av[3][na]),
                                           cvt_pk_bf16(av[4][na], av[5][na]), cvt_pk_bf16(av[6][na], av[7][na])));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bf16x8 b1 = __builtin_bit_cast(
                        bf16x8, make_uint4(cvt_pk_bf16(bv[0][nb], bv[1][nb]), cvt_pk_bf16(bv[2][nb], bv[3][nb]),
                                           cvt_pk_bf16(bv[4][nb], bv[5][nb]), cvt_pk_bf16(bv[6][nb], bv[7][nb])));
#pragma unroll
                    for (int na = 0; na < NA; ++na)
                        acc[na][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[na], b1, acc[na][nb], 0, 0, 0);
                }
            } else if constexpr (PIECES == 3) {
                static_assert(STEPS == 8, "one bf16 MFMA group = 8 k slots per lane");
                bf16x8 a3[NA][3];
#pragma unroll
                for (int na = 0; na < NA; ++na) {
                    unsigned ap[3][4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) split3(av[2 * q][na], av[2 * q + 1][na], ap[0][q], ap[1][q], ap[2][q]);
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc)
                        a3[na][pc] = __builtin_bit_cast(bf16x8, make_uint4(ap[pc][0], ap[pc][1], ap[pc][2], ap[pc][3]));
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    unsigned bp[3][4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) split3(bv[2 * q][nb], bv[2 * q + 1][nb], bp[0][q], bp[1][q], bp[2][q]);
                    bf16x8 b3[3];
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc)
                        b3[pc] = __builtin_bit_cast(bf16x8, make_uint4(bp[pc][0], bp[pc][1], bp[pc][2], bp[pc][3]));
#pragma unroll
                    for (int na = 0; na < NA; ++na) {
                        acc[na][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[na][0], b3[2], acc[na][nb], 0, 0, 0);
                        acc[na][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[na][2], b3[0], acc[na][nb], 0, 0, 0);
                        acc[na][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[na][1], b3[1], acc[na][nb], 0, 0, 0);
                        acc[na][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[na][0], b3[1], acc[na][nb], 0, 0, 0);
                        acc[na][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[na][1], b3[0], acc[na][nb], 0, 0, 0);
                        acc[na][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[na][0], b3[0], acc[na][nb], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < STEPS; ++t)
                    if (t < nsteps)
#pragma unroll
                        for (int na = 0; na < NA; ++na)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[na][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][na], bv[t][nb], acc[na][nb], 0, 0, 0);
            }
        }
    }
    float* p = partial + (long long)(plan.first[j] + split) * cin * cout;
#pragma unroll
    for (int na = 0; na < NA; ++na)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = co0 + nb * 32 + l31;
            if (col >= cout) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + na * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (ci < cin) p[(long long)ci * cout + col] = acc[na][nb][r];
            }
        }
}

// dW[j] = sum of offset j's partial tiles, in slot order (deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ partial, int cc, int K, WgradPlan plan,
                                                    float* __restrict__ dw) {
    const long long e = blockIdx.x * 256ll + threadIdx.x;
    if (e >= (long long)cc * K) return;
    const int j = (int)(e / cc);
    const int w = (int)(e - (long long)j * cc);
    const float* p = partial + (long long)plan.first[j] * cc + w;
    // four interleaved running sums (slots k % 4), combined in a fixed order: four independent load chains in flight
    const int ns = plan.nsplit[j];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < ns; k += 4) {
        s0 += p[(long long)k * cc];
        s1 += p[(long long)(k + 1) * cc];
        s2 += p[(long long)(k + 2) * cc];
        s3 += p[(long long)(k + 3) * cc];
    }
    for (; k < ns; ++k) s0 += p[(long long)k * cc];
    dw[e] = (s0 + s1) + (s2 + s3);
}

// transposed kernel map: nbr_t[i][j] = u with nbr[u][j] == i  (per offset the map is injective)
__global__ __launch_bounds__(256) void transpose_map(const int* __restrict__ nbr, long long n_out, int K,
                                                     int* __restrict__ nbr_t) {
    const long long t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= n_out * K) return;
    const int i = nbr[t];
    if (i >= 0) nbr_t[(long long)i * K + (t % K)] = (int)(t / K);
}

// column sums (bias gradient): blocks over (32 columns, row chunk); partial sums meet through fp32
// atomics on the c output words (out is pre-zeroed)
__global__ __launch_bounds__(256) void col_sum(const float* __restrict__ x, long long n, int c, int ld,
                                               float* __restrict__ out) {
    __shared__ float s[8][32];
    const int col = blockIdx.x * 32 + (threadIdx.x & 31), ry = threadIdx.x >> 5;
    const long long r_lo = n * blockIdx.y / gridDim.y, r_hi = n * (blockIdx.y + 1) / gridDim.y;
    float acc = 0.f;
    if (col < c)
        for (long long r = r_lo + ry; r < r_hi; r += 8) acc += x[r * ld + col];
    s[ry][threadIdx.x & 31] = acc;
    __syncthreads();
    if (ry == 0 && col < c) {
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t += s[k][threadIdx.x & 31];
        unsafeAtomicAdd(&out[col], t);
    }
}

// the same sums without atomics: every (32 columns, row chunk) block writes its partial sums to ws[chunk][c], a second
// launch adds the chunks of a column in chunk order - the result does not depend on the order in which workgroups run
__global__ __launch_bounds__(256) void col_sum_partial(const float* __restrict__ x, long long n, int c, int ld,
                                                       float* __restrict__ ws) {
    __shared__ float s[8][32];
    const int col = blockIdx.x * 32 + (threadIdx.x & 31), ry = threadIdx.x >> 5;
    const long long r_lo = n * blockIdx.y / gridDim.y, r_hi = n * (blockIdx.y + 1) / gridDim.y;
    float acc = 0.f;
    if (col < c)
        for (long long r = r_lo + ry; r < r_hi; r += 8) acc += x[r * ld + col];
    s[ry][threadIdx.x & 31] = acc;
    __syncthreads();
    if (ry == 0 && col < c) {
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t += s[k][threadIdx.x & 31];
        ws[(long long)blockIdx.y * c + col] = t;
    }
}
__global__ __launch_bounds__(256) void col_sum_chunks(const float* __restrict__ ws, int chunks, int c,
                                                      float* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= c) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // four load chains, combined in a fixed order
    int k = 0;
    for (; k + 3 < chunks; k += 4) {
        s0 += ws[(long long)k * c + col];
        s1 += ws[(long long)(k + 1) * c + col];
        s2 += ws[(long long)(k + 2) * c + col];
        s3 += ws[(long long)(k + 3) * c + col];
    }
    for (; k < chunks; ++k) s0 += ws[(long long)k * c + col];
    out[col] = (s0 + s1) + (s2 + s3);
}

// sums the per-split partial tiles and applies the epilogue (scale/shift/residual/relu).
// One thread per 4 consecutive columns (float4 loads, cout % 4 == 0) and per quarter of the splits;
// the four quarter-sums meet through LDS, so small outputs with many splits still have enough loads
// in flight (a one-thread-per-element loop over 64 splits ran at 1.7 TB/s).
__global__ __launch_bounds__(256) void conv_finish(ConvArgs a) {
    __shared__ float4 red[4][64];
    const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
    const long long total4 = a.n_out * (long long)a.cout / 4;
    const long long total = a.n_out * (long long)a.cout;
    for (long long base = blockIdx.x * 64ll; base < total4; base += (long long)gridDim.x * 64) {
        const long long e4 = base + l;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e4 < total4) {
            const float4* p = reinterpret_cast<const float4*>(a.partial) + e4;
#pragma unroll 4
            for (int k = q; k < a.splits; k += 4) {
                const float4 v = partial_load4(p + (long long)k * (total / 4));
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        red[q][l] = s;
        __syncthreads();
        if (q == 0 && e4 < total4) {
            float v[4] = {red[0][l].x + red[1][l].x + red[2][l].x + red[3][l].x,
                          red[0][l].y + red[1][l].y + red[2][l].y + red[3][l].y,
                          red[0][l].z + red[1][l].z + red[2][l].z + red[3][l].z,
                          red[0][l].w + red[1][l].w + red[2][l].w + red[3][l].w};
            const long long e = e4 * 4;
            const long long row = e / a.cout;
            const int col = (int)(e - row * a.cout);
            if (a.wide) {                        // 16-byte aligned operands: one float4 per operand instead of four words
                float4 x = make_float4(v[0], v[1], v[2], v[3]);
                if (a.acc_in) {
                    const float4 p = *reinterpret_cast<const float4*>(a.acc_in + row * a.acc_ld + col);
                    x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
                }
                const float4 sc = a.scale ? *reinterpret_cast<const float4*>(a.scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float4 sh = a.shift ? *reinterpret_cast<const float4*>(a.shift + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                x.x = x.x * sc.x + sh.x; x.y = x.y * sc.y + sh.y; x.z = x.z * sc.z + sh.z; x.w = x.w * sc.w + sh.w;
                if (a.res) {
                    const float4 p = a.res_hl ? hl_load4(a.res + row * a.res_ld, col)
                                              : *reinterpret_cast<const float4*>(a.res + row * a.res_ld + col);
                    x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
                }
                if (a.relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
                if (a.out_hl) {
                    if (a.range_flag && hl_out_of_range(x)) *a.range_flag = 1;
                    hl_store4(a.out + row * a.out_ld, col, x);
                } else {
                    *reinterpret_cast<float4*>(a.out + row * a.out_ld + col) = x;
                }
            } else
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = v[i];
                if (a.acc_in) x += a.acc_in[row * a.acc_ld + col + i];
                x = x * (a.scale ? a.scale[col + i] : 1.f) + (a.shift ? a.shift[col + i] : 0.f);
                if (a.res) x += a.res[row * a.res_ld + col + i];
                if (a.relu) x = fmaxf(x, 0.f);
                a.out[row * a.out_ld + col + i] = x;
            }
        }
        __syncthreads();
    }
}

// up to FINISH_SMALL_MAX partial tiles (the mask groups, the split-K of the middle levels) with 16-byte aligned
// operands: a thread owns a float4 of the output, has all its partial loads in flight at once and runs the epilogue
// itself - no LDS exchange, no barriers, every thread stores (ts1 96 -> 96 conv + finish: 125 -> 104 us).  Summation
// order = conv_finish's: four running sums over the partials k % 4, then ((s0 + s1) + s2) + s3 - bit-identical results.
constexpr int FINISH_SMALL_MAX = 16;
__global__ __launch_bounds__(256) void conv_finish_small(ConvArgs a) {
    const long long total4 = a.n_out * (long long)a.cout / 4;
    const int cq = a.cout >> 2;
    for (long long e4 = blockIdx.x * 256ll + threadIdx.x; e4 < total4; e4 += (long long)gridDim.x * 256) {
        const float4* p = reinterpret_cast<const float4*>(a.partial) + e4;
        const long long row = e4 / cq;
        float4 pv[FINISH_SMALL_MAX];
#pragma unroll
        for (int k = 0; k < FINISH_SMALL_MAX; ++k)          // (a.gvalid: partial tiles of rows without a neighbour in group k were never written)
            pv[k] = (k < a.splits && (!a.gvalid || a.gvalid[(long long)k * a.n_out + row]))
                        ? partial_load4(p + (long long)k * total4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int col = (int)(e4 - row * cq) * 4;
        float4 sq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < FINISH_SMALL_MAX; ++k)
            if (k < a.splits) { sq[k & 3].x += pv[k].x; sq[k & 3].y += pv[k].y; sq[k & 3].z += pv[k].z; sq[k & 3].w += pv[k].w; }
        float4 x = make_float4(sq[0].x + sq[1].x + sq[2].x + sq[3].x, sq[0].y + sq[1].y + sq[2].y + sq[3].y,
                               sq[0].z + sq[1].z + sq[2].z + sq[3].z, sq[0].w + sq[1].w + sq[2].w + sq[3].w);
        if (a.acc_in) {
            const float4 q = *reinterpret_cast<const float4*>(a.acc_in + row * a.acc_ld + col);
            x.x += q.x; x.y += q.y; x.z += q.z; x.w += q.w;
        }
        const float4 sc = a.scale ? *reinterpret_cast<const float4*>(a.scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 sh = a.shift ? *reinterpret_cast<const float4*>(a.shift + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        x.x = x.x * sc.x + sh.x; x.y = x.y * sc.y + sh.y; x.z = x.z * sc.z + sh.z; x.w = x.w * sc.w + sh.w;
        if (a.res) {
            const float4 q = a.res_hl ? hl_load4(a.res + row * a.res_ld, col)
                                      : *reinterpret_cast<const float4*>(a.res + row * a.res_ld + col);
            x.x += q.x; x.y += q.y; x.z += q.z; x.w += q.w;
        }
        if (a.relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
        if (a.out_hl) {
            if (a.range_flag && hl_out_of_range(x)) *a.range_flag = 1;
            hl_store4(a.out + row * a.out_ld, col, x);
        } else {
            *reinterpret_cast<float4*>(a.out + row * a.out_ld + col) = x;
        }
    }
}

// scalar variant for cout % 4 != 0
__global__ __launch_bounds__(256) void conv_finish_scalar(ConvArgs a) {
    const long long total = a.n_out * (long long)a.cout;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long row = t / a.cout;
        const int col = (int)(t - row * a.cout);
        float v = 0.f;
        for (int sidx = 0; sidx < a.splits; ++sidx) v += a.partial[sidx * total + t];
        if (a.acc_in) v += a.acc_in[row * a.acc_ld + col];
        v = v * (a.scale ? a.scale[col] : 1.f) + (a.shift ? a.shift[col] : 0.f);
        if (a.res) v += a.res[row * a.res_ld + col];
        if (a.relu) v = fmaxf(v, 0.f);
        a.out[row * a.out_ld + col] = v;
    }
}

}  // namespace
namespace cvsc {
std::atomic<int> g_ablation{0};
int launch_finish(const ConvArgs& a, hipStream_t st) {
    if (g_ablation.load(std::memory_order_relaxed) & 1) return CV_OK;      // cv_sp_set_ablation: timing only
    const long long total = a.n_out * (long long)a.cout;
    if (a.wide && a.splits <= FINISH_SMALL_MAX)
        conv_finish_small<<<(unsigned)std::min<long long>((total / 4 + 255) / 256, 16384), 256, 0, st>>>(a);
    else if (a.cout % 4 == 0)
        conv_finish<<<(unsigned)std::min<long long>((total / 4 + 63) / 64, 8192), 256, 0, st>>>(a);
    else
        conv_finish_scalar<<<(unsigned)std::min<long long>((total + 255) / 256, 4096), 256, 0, st>>>(a);
    CV_LAUNCH_CHECK();
    return CV_OK;
}
}  // namespace cvsc
namespace {

// sort key of a row = bit mask of its valid neighbours among offsets [j_begin, j_end)
__global__ __launch_bounds__(256) void mask_keys(const int* __restrict__ nbr, long long n, int K, int j_begin,
                                                 int j_end, long long* __restrict__ keys) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    unsigned long long m = 0;
    for (int j = j_begin; j < j_end; ++j)
        if (nbr[i * K + j] >= 0) m |= 1ull << (j - j_begin);
    keys[i] = (long long)m;
}

// ---- row orders per offset group by counting sort on the group's neighbour bit mask (<= 10 bits).
// Three launches for all groups (blockIdx.y = group); LDS-aggregated histograms because global
// atomics on a few hundred hot counters serialise (~11 ns each on MI355X).
constexpr int MP_BINS = 1024;
constexpr int MP_THREADS = 1024;

__device__ __forceinline__ int group_mask(const int* __restrict__ nbr, long long row, long long n, int K, int jb, int je) {
    int m = 0;
    for (int j = jb; j < je; ++j)
        if (nbr[row * K + j] >= 0) m |= 1 << (j - jb);
    // sort key: rows with equal masks adjacent, and (when it fits the 1024 bins) ordered by the NUMBER of
    // offsets they need, so the tiles at the end of the order are the most expensive ones.
    // (Measured and dropped, profiles/r2/xcd_tiles.txt: masks sorted inside the eighths of the spatial row order with
    // tile t run on XCD t / (tiles / 8) - L2 hit rate of the ts1 conv 48 -> 63 %, memory-side fetch 192 -> 120 MB per
    // launch, and the launch 14 % SLOWER: the kernel is bound by its dependent round trips, not by fetch bandwidth,
    // and the cost ordering of the tiles is worth more than the hits.)
    if (je - jb <= 7) m |= __popc(m) << 7;
    (void)n;
    return m;
}

__global__ __launch_bounds__(MP_THREADS) void mp_hist(const int* __restrict__ nbr, long long n, int K, int groups,
                                                      int* __restrict__ hist /*[groups][MP_BINS]*/) {
    __shared__ int lh[MP_BINS];
    const int g = blockIdx.y;
    const int jb = K * g / groups, je = K * (g + 1) / groups;
    for (int i = threadIdx.x; i < MP_BINS; i += MP_THREADS) lh[i] = 0;
    __syncthreads();
    const long long row = blockIdx.x * (long long)MP_THREADS + threadIdx.x;
    if (row < n) atomicAdd(&lh[group_mask(nbr, row, n, K, jb, je)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < MP_BINS; i += MP_THREADS)
        if (lh[i]) atomicAdd(&hist[g * MP_BINS + i], lh[i]);
}

__global__ __launch_bounds__(MP_BINS) void mp_scan(int* __restrict__ hist) {   // exclusive, in place, per group
    __shared__ int s[MP_BINS];
    int* h = hist + blockIdx.x * MP_BINS;
    const int v = h[threadIdx.x];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < MP_BINS; off <<= 1) {
        const int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    h[threadIdx.x] = s[threadIdx.x] - v;
}

__global__ __launch_bounds__(MP_THREADS) void mp_scatter(const int* __restrict__ nbr, long long n, int K, int groups,
                                                         int* __restrict__ cursor, int* __restrict__ perm,
                                                         int* __restrict__ nbrp, int W) {
    // with the map rows (nbrp): behind them, one byte per (group, row) - does the row have a neighbour in the group at all?
    // (27 % of the rows of a scan have none below / above them: their partial sums are never written nor read, zskip)
    unsigned char* gv = nbrp ? reinterpret_cast<unsigned char*>(nbrp + (long long)groups * n * W) : nullptr;
    __shared__ int lh[MP_BINS];
    const int g = blockIdx.y;
    const int jb = K * g / groups, je = K * (g + 1) / groups;
    for (int i = threadIdx.x; i < MP_BINS; i += MP_THREADS) lh[i] = 0;
    __syncthreads();
    const long long row = blockIdx.x * (long long)MP_THREADS + threadIdx.x;
    int key = 0, rank = 0;
    if (row < n) {
        key = group_mask(nbr, row, n, K, jb, je);
        rank = atomicAdd(&lh[key], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MP_BINS; i += MP_THREADS)
        if (lh[i]) lh[i] = atomicAdd(&cursor[g * MP_BINS + i], lh[i]);
    __syncthreads();
    if (row < n) {
        const long long pos = (long long)g * n + lh[key] + rank;
        perm[pos] = (int)row;
        if (nbrp) {
            for (int j = jb; j < je; ++j) nbrp[pos * W + (j - jb)] = nbr[row * K + j];
            gv[(long long)g * n + row] = key != 0;
        }
    }
}

// the same three launches for several maps at once (every processing order of a scene): blockIdx.y = (job, group)
struct PermJobsDev {
    CvPermJob j[CV_MAX_PERM_JOBS];
    int group_begin[CV_MAX_PERM_JOBS + 1];
    int n;
};

__device__ __forceinline__ int perm_job_of(const PermJobsDev& jobs, int gy) {
    int ji = 0;
    while (ji + 1 < jobs.n && gy >= jobs.group_begin[ji + 1]) ++ji;
    return ji;
}

__global__ __launch_bounds__(MP_THREADS) void mp_hist_batch(const PermJobsDev jobs, int* __restrict__ hist) {
    __shared__ int lh[MP_BINS];
    const int ji = perm_job_of(jobs, blockIdx.y);
    const CvPermJob& jb = jobs.j[ji];
    const long long row = blockIdx.x * (long long)MP_THREADS + threadIdx.x;
    if (blockIdx.x * (long long)MP_THREADS >= jb.n) return;
    const int g = blockIdx.y - jobs.group_begin[ji];
    const int jlo = jb.K * g / jb.groups, jhi = jb.K * (g + 1) / jb.groups;
    for (int i = threadIdx.x; i < MP_BINS; i += MP_THREADS) lh[i] = 0;
    __syncthreads();
    if (row < jb.n) atomicAdd(&lh[group_mask(jb.nbr, row, jb.n, jb.K, jlo, jhi)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < MP_BINS; i += MP_THREADS)
        if (lh[i]) atomicAdd(&hist[blockIdx.y * MP_BINS + i], lh[i]);
}

// (round 5: no scan launch between the histogram and the scatter of the batched orders - a scatter workgroup scans its group's
// 1024 bin counts itself (one per thread) and claims its rows' places from a SECOND zeroed array of running counts per bin)
__global__ __launch_bounds__(MP_THREADS) void mp_scatter_batch(const PermJobsDev jobs, const int* __restrict__ hist,
                                                               int* __restrict__ cursor) {
    static_assert(MP_THREADS == MP_BINS, "one bin per thread in the scan");
    __shared__ int lh[MP_BINS];
    __shared__ int pre[MP_BINS];
    __shared__ int wtot[MP_THREADS / 64];
    const int ji = perm_job_of(jobs, blockIdx.y);
    const CvPermJob& jb = jobs.j[ji];
    if (blockIdx.x * (long long)MP_THREADS >= jb.n) return;
    const int g = blockIdx.y - jobs.group_begin[ji];
    const int K = jb.K, jlo = K * g / jb.groups, jhi = K * (g + 1) / jb.groups, W = (K + jb.groups - 1) / jb.groups;
    int* nbrp = jb.with_map ? jb.perm + (long long)jb.groups * jb.n : nullptr;
    for (int i = threadIdx.x; i < MP_BINS; i += MP_THREADS) lh[i] = 0;
    {   // exclusive scan of the group's bin counts: first slot of every bin
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int v = hist[blockIdx.y * MP_BINS + threadIdx.x];
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < MP_THREADS / 64; ++w) base += w < wave ? wtot[w] : 0;
        pre[threadIdx.x] = base + incl - v;
    }
    __syncthreads();
    const long long row = blockIdx.x * (long long)MP_THREADS + threadIdx.x;
    int key = 0, rank = 0;
    if (row < jb.n) {
        key = group_mask(jb.nbr, row, jb.n, K, jlo, jhi);
        rank = atomicAdd(&lh[key], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MP_BINS; i += MP_THREADS)
        if (lh[i]) lh[i] = pre[i] + atomicAdd(&cursor[blockIdx.y * MP_BINS + i], lh[i]);
    __syncthreads();
    if (row < jb.n) {
        const long long pos = (long long)g * jb.n + lh[key] + rank;
        jb.perm[pos] = (int)row;
        if (nbrp) {
            for (int j = jlo; j < jhi; ++j) nbrp[pos * W + (j - jlo)] = jb.nbr[row * K + j];
            reinterpret_cast<unsigned char*>(nbrp + (long long)jb.groups * jb.n * W)[(long long)g * jb.n + row] = key != 0;
        }
    }
}

// ------------------------------------------------------------------ elementwise helpers
// y = x*scale + shift (+relu)   (MinkowskiBatchNorm in eval mode, MinkowskiReLU)
__global__ __launch_bounds__(256) void affine_rows(const float* __restrict__ x, long long n, int c,
                                                   int x_ld, const float* __restrict__ scale,
                                                   const float* __restrict__ shift,
                                                   const float* __restrict__ residual, int res_ld, int relu,
                                                   float* __restrict__ y, int y_ld) {
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < n * c; t += (long long)gridDim.x * 256) {
        const long long r = t / c;
        const int k = (int)(t - r * c);
        float v = x[r * x_ld + k];
        if (scale) v = v * scale[k] + (shift ? shift[k] : 0.f);
        if (residual) v += residual[r * res_ld + k];
        if (relu) v = fmaxf(v, 0.f);
        y[r * y_ld + k] = v;
    }
}

// float4 flavour (c, leading dimensions % 4 == 0, 16-byte aligned bases): one thread = 4 consecutive channels of a row
__global__ __launch_bounds__(256) void affine_rows4(const float* __restrict__ x, long long n, int c,
                                                    int x_ld, const float* __restrict__ scale,
                                                    const float* __restrict__ shift,
                                                    const float* __restrict__ residual, int res_ld, int relu,
                                                    float* __restrict__ y, int y_ld, float* __restrict__ y_hl = nullptr,
                                                    int y_hl_ld = 0, int* __restrict__ range_flag = nullptr,
                                                    unsigned* __restrict__ ybits = nullptr) {
    const int cq = c >> 2;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < n * cq; t += (long long)gridDim.x * 256) {
        const long long r = t / cq;
        const int k = (int)(t - r * cq) * 4;
        float4 v = *reinterpret_cast<const float4*>(x + r * x_ld + k);
        if (scale) {
            const float4 sc = *reinterpret_cast<const float4*>(scale + k);
            const float4 sh = shift ? *reinterpret_cast<const float4*>(shift + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        }
        if (residual) {
            const float4 p = *reinterpret_cast<const float4*>(residual + r * res_ld + k);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(y + r * y_ld + k) = v;
        if (y_hl) {                                  // the same values once more as the fp16 pairs the next convolution multiplies
            if (range_flag && hl_out_of_range(v)) *range_flag = 1;
            hl_store4(y_hl + r * y_hl_ld, k, v);
        }
        if (ybits) {
            // where the ReLU is open, one bit per element: the 8 lanes of a row's 32-channel chunk (c % 32 == 0: they are
            // consecutive lanes of one wave, all active together) OR their nibbles into one word
            unsigned w = ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (k & 31);
            w |= (unsigned)__shfl_xor((int)w, 1);
            w |= (unsigned)__shfl_xor((int)w, 2);
            w |= (unsigned)__shfl_xor((int)w, 4);
            if ((k & 31) == 0) ybits[r * (c >> 5) + (k >> 5)] = w;
        }
    }
}

__host__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// scale = gamma * rsqrt(var + eps), shift = beta - mean*scale (+ bias*scale)
__global__ void bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                        const float* bias, float eps, int c, float* scale, float* shift) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= c) return;
    const float s = gamma[k] / sqrtf(var[k] + eps);
    scale[k] = s;
    shift[k] = beta[k] - mean[k] * s + (bias ? bias[k] * s : 0.f);
}

// fp32 rows <-> hl format (boundary of the format: tests, modular callers)
__global__ __launch_bounds__(256) void to_hl(const float* __restrict__ x, long long n, int c, int x_ld, float* __restrict__ y,
                                             int y_ld, int* __restrict__ range_flag) {
    const int cq = c >> 2;
    const long long total = n * cq;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / cq;
        const int col = (int)(e - row * cq) * 4;
        const float* p = x + row * x_ld + col;
        const float4 v = make_float4(p[0], p[1], p[2], p[3]);
        if (range_flag && hl_out_of_range(v)) *range_flag = 1;
        hl_store4(y + row * y_ld, col, v);
    }
}
__global__ __launch_bounds__(256) void from_hl(const float* __restrict__ x, long long n, int c, int x_ld, float* __restrict__ y,
                                               int y_ld) {
    const int cq = c >> 2;
    const long long total = n * cq;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / cq;
        const int col = (int)(e - row * cq) * 4;
        const float4 v = hl_load4(x + row * x_ld, col);
        float* p = y + row * y_ld + col;
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
}

// ------------------------------------------------------------------ BatchNorm in training mode
// MinkowskiBatchNorm = nn.BatchNorm1d over the [N, C] feature rows (utils/minkunet.py:56): batch mean and
// biased variance per channel.  Two-level column reduction: blocks of (32 channels x 8 row lanes) over row
// chunks accumulate in double, a second tiny kernel combines the chunks.
constexpr int BN_CHUNKS = 1024;

template <int MODE>   // 0: sum x, sum x^2      1: sum dy', sum dy'*xhat   (dy' = dy masked by y > 0 if y given)
__global__ __launch_bounds__(256) void bn_col_reduce(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ y, long long n, int c, int ld,
                                                     const float* __restrict__ mean, const float* __restrict__ var,
                                                     float eps, double* __restrict__ partial) {
    __shared__ double s0[8][32], s1[8][32];
    const int col = blockIdx.x * 32 + (threadIdx.x & 31), ry = threadIdx.x >> 5;
    const long long r_lo = n * blockIdx.y / gridDim.y, r_hi = n * (blockIdx.y + 1) / gridDim.y;
    double a0 = 0.0, a1 = 0.0;
    if (col < c) {
        float mu = 0.f, istd = 0.f;
        if (MODE == 1) { mu = mean[col]; istd = 1.0f / sqrtf(var[col] + eps); }
        for (long long r = r_lo + ry; r < r_hi; r += 8) {
            const float xv = x[r * ld + col];
            if (MODE == 0) { a0 += (double)xv; a1 += (double)xv * (double)xv; }
            else {
                float g = dy[r * ld + col];
                if (y && !(y[r * ld + col] > 0.f)) g = 0.f;
                a0 += (double)g;
                a1 += (double)(g * ((xv - mu) * istd));
            }
        }
    }
    s0[ry][threadIdx.x & 31] = a0; s1[ry][threadIdx.x & 31] = a1;
    __syncthreads();
    if (ry == 0 && col < c) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = 0; k < 8; ++k) { t0 += s0[k][threadIdx.x & 31]; t1 += s1[k][threadIdx.x & 31]; }
        partial[((long long)blockIdx.y * c + col) * 2 + 0] = t0;
        partial[((long long)blockIdx.y * c + col) * 2 + 1] = t1;
    }
}

// float4 flavour of bn_col_reduce (c % 4 == 0, c <= 1024, ld % 4 == 0, 16-byte aligned): a thread owns a quad of
// channels, 256 / (c/4) row lanes per block, four rows of loads in flight per thread; the scalar kernel above kept
// one 4-byte load per thread in flight (1.8 TB/s on the ts1 levels).  Same partial layout, fixed summation order.
// The ReLU mask of a BatchNorm + ReLU output for its backward: either the output rows themselves (y > 0) or one bit per
// element, [row][c / 32] words written by the forward pass (cv_sp_affine_hl_f32) - a 32nd of the bytes of y in the two
// backward passes that only want to know where the ReLU was open.
__device__ __forceinline__ float4 relu_open4(const float* __restrict__ y, const unsigned* __restrict__ ybits, long long r, int k,
                                             int ld, int c) {
    if (ybits) {
        const unsigned w = ybits[r * (c >> 5) + (k >> 5)] >> (k & 31);
        return make_float4((float)(w & 1u), (float)((w >> 1) & 1u), (float)((w >> 2) & 1u), (float)((w >> 3) & 1u));
    }
    return *reinterpret_cast<const float4*>(y + r * ld + k);
}
template <int MODE>
__global__ __launch_bounds__(256) void bn_col_reduce4(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ y, long long n, int c, int ld,
                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                      float eps, double* __restrict__ partial,
                                                      const unsigned* __restrict__ ybits = nullptr) {
    __shared__ double red[256][9];                     // [thread][2 x 4 sums], padded
    const int cq = c >> 2, rl = 256 / cq;              // row lanes
    const int quad = threadIdx.x % cq, lane_r = threadIdx.x / cq;
    const long long r_lo = n * blockIdx.x / gridDim.x, r_hi = n * (blockIdx.x + 1) / gridDim.x;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane_r < rl) {
        const int k = quad * 4;
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu;
        if (MODE == 1) {
            mu = *reinterpret_cast<const float4*>(mean + k);
            const float4 v = *reinterpret_cast<const float4*>(var + k);
            is = make_float4(1.0f / sqrtf(v.x + eps), 1.0f / sqrtf(v.y + eps), 1.0f / sqrtf(v.z + eps), 1.0f / sqrtf(v.w + eps));
        }
        auto take = [&](const float4& xv, float4 g, const float4& yv, bool has_y) {
            if (MODE == 0) {
                a[0] += (double)xv.x; a[1] += (double)xv.y; a[2] += (double)xv.z; a[3] += (double)xv.w;
                a[4] += (double)xv.x * (double)xv.x; a[5] += (double)xv.y * (double)xv.y;
                a[6] += (double)xv.z * (double)xv.z; a[7] += (double)xv.w * (double)xv.w;
            } else {
                if (has_y) {
                    if (!(yv.x > 0.f)) g.x = 0.f;
                    if (!(yv.y > 0.f)) g.y = 0.f;
                    if (!(yv.z > 0.f)) g.z = 0.f;
                    if (!(yv.w > 0.f)) g.w = 0.f;
                }
                a[0] += (double)g.x; a[1] += (double)g.y; a[2] += (double)g.z; a[3] += (double)g.w;
                a[4] += (double)(g.x * ((xv.x - mu.x) * is.x)); a[5] += (double)(g.y * ((xv.y - mu.y) * is.y));
                a[6] += (double)(g.z * ((xv.z - mu.z) * is.z)); a[7] += (double)(g.w * ((xv.w - mu.w) * is.w));
            }
        };
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        long long r = r_lo + lane_r;
        for (; r + 3ll * rl < r_hi; r += 4ll * rl) {
            float4 xv[4], gv[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long o = (r + (long long)u * rl) * ld + k;
                xv[u] = *reinterpret_cast<const float4*>(x + o);
                gv[u] = MODE == 1 ? *reinterpret_cast<const float4*>(dy + o) : z4;
                yv[u] = (MODE == 1 && (y || ybits)) ? relu_open4(y, ybits, r + (long long)u * rl, k, ld, c) : z4;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) take(xv[u], gv[u], yv[u], y != nullptr || ybits != nullptr);
        }
        for (; r < r_hi; r += rl) {
            const long long o = r * ld + k;
            take(*reinterpret_cast<const float4*>(x + o), MODE == 1 ? *reinterpret_cast<const float4*>(dy + o) : z4,
                 (MODE == 1 && (y || ybits)) ? relu_open4(y, ybits, r, k, ld, c) : z4, y != nullptr || ybits != nullptr);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = a[i];
    __syncthreads();
    if (threadIdx.x < cq) {
        double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = 0; q < rl; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] += red[q * cq + threadIdx.x][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            partial[((long long)blockIdx.x * c + threadIdx.x * 4 + i) * 2 + 0] = t[i];
            partial[((long long)blockIdx.x * c + threadIdx.x * 4 + i) * 2 + 1] = t[4 + i];
        }
    }
}

// MODE 0: mean/var from the partials, running statistics update (momentum, unbiased variance), folded
// scale/shift for the apply pass.  MODE 1: out0 = sum dy' (d beta), out1 = sum dy'*xhat (d gamma).
// 16 lanes per channel: each sums every 16th chunk (independent loads in flight instead of one thread walking 256
// dependent ones: 23 -> a few us per call, 124 calls per training step), then a fixed-order butterfly.
template <int MODE>
__global__ __launch_bounds__(256) void bn_col_finish(const double* __restrict__ partial, int chunks, long long n, int c,
                                                     float* out0, float* out1, float* running_mean, float* running_var,
                                                     float momentum, const float* gamma, const float* beta, float eps,
                                                     float* scale, float* shift) {
    const int sub = threadIdx.x & 15;
    const int k = blockIdx.x * 16 + (threadIdx.x >> 4);
    double t0 = 0.0, t1 = 0.0;
    if (k < c)
        for (int q = sub; q < chunks; q += 16) {
            t0 += partial[((long long)q * c + k) * 2];
            t1 += partial[((long long)q * c + k) * 2 + 1];
        }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
        t0 += __shfl_xor(t0, off);
        t1 += __shfl_xor(t1, off);
    }
    if (k >= c || sub != 0) return;
    if (MODE == 0) {
        const double mu = t0 / (double)n;
        double v = t1 / (double)n - mu * mu;
        if (v < 0.0) v = 0.0;
        out0[k] = (float)mu;
        out1[k] = (float)v;
        if (running_mean) {
            const double unb = n > 1 ? v * (double)n / (double)(n - 1) : v;
            running_mean[k] = (1.f - momentum) * running_mean[k] + momentum * (float)mu;
            running_var[k] = (1.f - momentum) * running_var[k] + momentum * (float)unb;
        }
        const float sc = gamma[k] / sqrtf((float)v + eps);
        scale[k] = sc;
        shift[k] = beta[k] - (float)mu * sc;
    } else {
        out0[k] = (float)t0;
        out1[k] = (float)t1;
    }
}

constexpr int BN_SLOT_BLOCKS = 4096;          // = the grid cap of the launch with the hl twin (cv_sp_bn_backward_hl_f32: CV_BN_SLOT_WORDS)
// dx = gamma*istd * (dy' - sum_dy/n - xhat * sum_dy_xhat/n)
__global__ __launch_bounds__(256) void bn_backward_apply(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ y, long long n, int c, int ld,
                                                         const float* __restrict__ mean, const float* __restrict__ var,
                                                         float eps, const float* __restrict__ gamma,
                                                         const float* __restrict__ sum_dy,
                                                         const float* __restrict__ sum_dy_xhat, float* __restrict__ dx,
                                                         float* __restrict__ dres) {
    const float inv_n = 1.0f / (float)n;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < n * c; t += (long long)gridDim.x * 256) {
        const long long r = t / c;
        const int k = (int)(t - r * c);
        const float istd = 1.0f / sqrtf(var[k] + eps);
        const float xh = (x[r * ld + k] - mean[k]) * istd;
        float g = dy[r * ld + k];
        if (y && !(y[r * ld + k] > 0.f)) g = 0.f;
        dx[r * ld + k] = gamma[k] * istd * (g - sum_dy[k] * inv_n - xh * sum_dy_xhat[k] * inv_n);
        if (dres) dres[r * ld + k] = g;
    }
}

// float4 flavour (c, ld % 4 == 0, 16-byte aligned)
__global__ __launch_bounds__(256) void bn_backward_apply4(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ y, long long n, int c, int ld,
                                                          const float* __restrict__ mean, const float* __restrict__ var,
                                                          float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ sum_dy,
                                                          const float* __restrict__ sum_dy_xhat, float* __restrict__ dx,
                                                          float* __restrict__ dres, float* __restrict__ dx_hl = nullptr,
                                                          unsigned* __restrict__ slot = nullptr,
                                                          int* __restrict__ range_flag = nullptr,
                                                          const unsigned* __restrict__ ybits = nullptr) {
    const float inv_n = 1.0f / (float)n;
    const int cq = c >> 2;
    // dx_hl: dx once more as fp16 pairs for the input-gradient convolution of the layer below, times the power of two that put
    // the PREVIOUS step's largest |dx| of this layer into [2^9, 2^10) (a factor of 64 to the fp16 range - beyond it the range
    // flag stops the optimizer step).  slot: BN_SLOT_BLOCKS words, one per workgroup, that receive this step's maxima (bits of
    // non-negative floats; plain stores - atomics on shared words cost ~1 us each across the XCDs: a fixed 28-50 us per launch),
    // BN_SLOT_BLOCKS words with the previous step's, one float that receives the inverse factor for that convolution's epilogue.
    __shared__ unsigned wred[4];
    float hs = 1.f, dmax = 0.f;
    if (dx_hl) {
        unsigned mb = 0u;
        for (int i = threadIdx.x; i < BN_SLOT_BLOCKS; i += 256) mb = max(mb, slot[BN_SLOT_BLOCKS + i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, off));
        if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = mb;
        __syncthreads();
        mb = max(max(wred[0], wred[1]), max(wred[2], wred[3]));
        __syncthreads();
        const unsigned E = (mb >> 23) & 255u;        // m in [2^(E - 127), 2^(E - 126))
        hs = (E >= 10u && E <= 250u) ? __uint_as_float((263u - E) << 23) : 1.f;          // 2^(136 - E)
        if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(slot)[2 * BN_SLOT_BLOCKS] = 1.f / hs;
    }
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < n * cq; t += (long long)gridDim.x * 256) {
        const long long r = t / cq;
        const int k = (int)(t - r * cq) * 4;
        const long long o = r * ld + k;
        const float4 xv = *reinterpret_cast<const float4*>(x + o);
        float4 g = *reinterpret_cast<const float4*>(dy + o);
        if (y || ybits) {
            const float4 yv = relu_open4(y, ybits, r, k, ld, c);
            if (!(yv.x > 0.f)) g.x = 0.f;
            if (!(yv.y > 0.f)) g.y = 0.f;
            if (!(yv.z > 0.f)) g.z = 0.f;
            if (!(yv.w > 0.f)) g.w = 0.f;
        }
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {g.x, g.y, g.z, g.w};
        float d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float istd = 1.0f / sqrtf(var[k + i] + eps);
            const float xh = (xs[i] - mean[k + i]) * istd;
            d[i] = gamma[k + i] * istd * (gs[i] - sum_dy[k + i] * inv_n - xh * sum_dy_xhat[k + i] * inv_n);
        }
        *reinterpret_cast<float4*>(dx + o) = make_float4(d[0], d[1], d[2], d[3]);
        if (dres) *reinterpret_cast<float4*>(dres + o) = g;
        if (dx_hl) {
            dmax = fmaxf(fmaxf(dmax, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
            const float4 v = make_float4(d[0] * hs, d[1] * hs, d[2] * hs, d[3] * hs);
            if (range_flag && hl_out_of_range(v)) *range_flag = 1;
            hl_store4(dx_hl + r * ld, k, v);
        }
    }
    if (dx_hl) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
        if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = __float_as_uint(dmax);
        __syncthreads();
        if (threadIdx.x == 0) slot[blockIdx.x] = max(max(wred[0], wred[1]), max(wred[2], wred[3]));
    }
}

// eval_joint.py:173-190: per point, head select by argmax class (class 9 -> head 0), exp(scale),
// prob = max softmax over the 9 object classes, class = argmax over the 9 object logits.
__global__ __launch_bounds__(256) void head_joint(const float* __restrict__ f, long long n, int ld,
                                                  int ncls, int log_scale, float* __restrict__ xyz,
                                                  float* __restrict__ scale, float* __restrict__ prob,
                                                  int* __restrict__ cls) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    const float* row = f + i * ld;
    const float* logit = row + 6 * ncls;
    float mx = logit[0];
    int am = 0;
    for (int k = 1; k <= ncls; ++k)
        if (logit[k] > mx) { mx = logit[k]; am = k; }
    float mo = logit[0];
    int ao = 0;
    for (int k = 1; k < ncls; ++k)
        if (logit[k] > mo) { mo = logit[k]; ao = k; }
    float den = 0.f;
    for (int k = 0; k <= ncls; ++k) den += expf(logit[k] - mx);
    const int h = am == ncls ? 0 : am;
    for (int d = 0; d < 3; ++d) {
        xyz[i * 3 + d] = row[h * 3 + d];
        const float s = row[3 * ncls + h * 3 + d];
        scale[i * 3 + d] = log_scale ? expf(s) : s;
    }
    prob[i] = expf(mo - mx) / den;
    cls[i] = ao;
}

// eval_separate.py:170-181 / train_separate.py:247-249,362: 8-channel head of a per-category model:
// xyz = f[0:3], scale = exp(f[3:6]), prob = softmax(f[6:8])[1]
__global__ __launch_bounds__(256) void head_separate(const float* __restrict__ f, long long n, int ld,
                                                     int log_scale, float* __restrict__ xyz,
                                                     float* __restrict__ scale, float* __restrict__ prob) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    const float* row = f + i * ld;
    for (int d = 0; d < 3; ++d) {
        xyz[i * 3 + d] = row[d];
        scale[i * 3 + d] = log_scale ? expf(row[3 + d]) : row[3 + d];
    }
    const float mx = fmaxf(row[6], row[7]);
    const float e0 = expf(row[6] - mx), e1 = expf(row[7] - mx);
    prob[i] = e1 / (e0 + e1);
}

// run-time options (cv_sp_set_option): conv_hd switch (bit NB - 1) and its row threshold; defaults from the environment
// defaults (profiles/r4/hd2_grid.txt, hd_shapes.txt): the 96-column fine-level launches (>= 16384 rows) on conv_hd, 8 waves x 2 ring
// stages: 558 -> 573 scenes/s with eight scenes in flight (528 -> 575 under the one-call scene path); 32- and 64-column
// launches and the coarse levels stay on conv_hl (conv_hd measured slower there)
// zskip (round 5, OFF: measured slower, profiles/r5/zskip_ab.txt - 570 against 577 scenes/s seven in flight, 332 against 337 one
// at a time): 27 % of a scan's rows have no neighbour in the first / last mask group; with the switch on their tiles write no
// partial tile and the finish launch reads none for such (group, row) pairs - the three validity bytes a finish thread then
// loads cost more than the zeros it no longer streams
std::atomic<long long> g_opt_zskip{getenv("CV_ZSKIP") ? atoll(getenv("CV_ZSKIP")) : 0};
std::atomic<long long> g_opt_hd_mask{getenv("CV_HD") ? atoll(getenv("CV_HD")) : 4};
std::atomic<long long> g_opt_hd_min_rows{getenv("CV_HD_MIN_ROWS") ? atoll(getenv("CV_HD_MIN_ROWS")) : 16384};
std::atomic<long long> g_opt_hd_shape{getenv("CV_HD_SHAPE") ? atoll(getenv("CV_HD_SHAPE")) : 2};      // 0: 8 waves x 3 stages, 1: 4 x 2, 2: 8 x 2

template <int NB>
int launch_rows(const ConvArgs& a, bool vec, hipStream_t st) {
    dim3 grid((unsigned)((a.n_out + TM - 1) / TM), (unsigned)((a.cout + NB * 32 - 1) / (NB * 32)),
              (unsigned)a.splits);
    int per_wg = 0;                                  // most offsets one workgroup walks (the kernel's own formulas)
    if (vec) {
        const int nj_wp = a.j_end - a.j_begin, nch_wp = a.cin / KC;
        for (int z = 0; z < a.splits; ++z) {
            long long lo, hi;
            if (a.perm_per_split) {
                lo = (long long)nj_wp * z / a.splits * nch_wp;
                hi = (long long)nj_wp * (z + 1) / a.splits * nch_wp;
            } else {
                lo = (long long)nj_wp * nch_wp * z / a.splits;
                hi = (long long)nj_wp * nch_wp * (z + 1) / a.splits;
            }
            if (hi > lo) per_wg = std::max(per_wg, (int)((hi - 1) / nch_wp - lo / nch_wp + 1));
        }
    }
    static const bool xcd_on = getenv("CV_XCD_TILES") && atoi(getenv("CV_XCD_TILES")) != 0;      // experiment, off
    ConvArgs ax = a;
    dim3 gridx = grid;
    if (xcd_on && grid.x >= 64) {                    // a few tiles per XCD at least; below that the input fits every L2
        ax.xcd_tiles = 1;
        gridx.x = (grid.x + 7) / 8 * 8;
    }
    if (a.in_hl) {
        CV_REQUIRE(per_wg * (a.cin / KC) + (a.in2 ? a.cin2 / KC : 0) <= HL_MAX_UNITS, CV_EINVAL,
                   "hl-format convolution: more than %d units per workgroup (Cin too wide)", HL_MAX_UNITS);
        CV_REQUIRE(vec && a.wp6 && a.pieces == 2 && a.wide && per_wg <= WP_NPRE && NB <= 4, CV_EINVAL,
                   "hl-format input needs the fp16-pair weights, Cin %% 32 == 0, 16-byte aligned operands and at most %d "
                   "kernel offsets per workgroup", WP_NPRE);
        // (measured and dropped, profiles/r2/hl_burst.txt: a variant for workgroups of <= 8 units with every weight tile
        // requested at once through global_load_lds and two dependent round trips instead of 3 + units - no faster)
        // (NS = 8 / 6 / 4 slots for NB = 1 / 2 / 3 measured: every layer 25-45 % slower - two workgroups per CU
        // instead of three or four cost more than the deeper prefetch gains, profiles/r2/hl_slots.txt)
        // 256-row workgroups where the launch has plenty of tiles and no split-K: an experiment (CV_HL_NW8=1), off by
        // default - measured slower, profiles/r2/hl_nw8.txt
        // conv_hd (round 4): LDS-DMA operand rings, 256-row workgroups; CV_HD bit NB - 1 switches the NB x 32-column kernel on
        // for launches of at least CV_HD_MIN_ROWS output rows
        const int hd_mask = (int)g_opt_hd_mask.load(std::memory_order_relaxed);
        const long long hd_min_rows = g_opt_hd_min_rows.load(std::memory_order_relaxed);
        const int hd_shape = (int)g_opt_hd_shape.load(std::memory_order_relaxed);
        if constexpr (NB <= 3) {
            if (((hd_mask >> (NB - 1)) & 1) && a.n_out >= hd_min_rows && !ax.xcd_tiles) {
                ConvArgs ah = a;
                ah.tickets = nullptr;
                const int nw = hd_shape == 1 ? 4 : 8;
                dim3 g((unsigned)((a.n_out + nw * 32 - 1) / (nw * 32)), grid.y, grid.z);
                if (hd_shape == 1) conv_hd<NB, 4, 2><<<g, 256, 0, st>>>(ah);
                else if (hd_shape == 2) conv_hd<NB, 8, 2><<<g, 512, 0, st>>>(ah);
                else conv_hd<NB, 8, 3><<<g, 512, 0, st>>>(ah);
                CV_LAUNCH_CHECK();
                if (a.splits > 1) return launch_finish(a, st);
                return CV_OK;
            }
        }
        static const int nw8_mask = getenv("CV_HL_NW8") ? atoi(getenv("CV_HL_NW8")) : 0;        // bit NB - 1
        const bool nw8_on = (nw8_mask >> (NB - 1)) & 1;
        if constexpr (NB <= 4) {
            if (nw8_on && !ax.xcd_tiles && a.n_out >= 16384 && (a.splits == 1 || a.perm_per_split)) {
                dim3 g8((unsigned)((a.n_out + 255) / 256), grid.y, grid.z);
                conv_hl<NB, 2, 8><<<g8, 512, 0, st>>>(ax);
            } else {
                // two unit slots (85 / 113 / 128 VGPRs for 32 / 64 / 96 columns: five / four / four workgroups per CU instead
                // of four / three / three; net 2.53 -> 2.48 ms for the first two, 2.52 -> 2.48 ms for the third once a compiler
                // barrier keeps its fragment reads from being hoisted - its remaining spills are outside the unit loop).
                // CV_HL_NS2: bit nb-1 = two slots for NB = nb
                static const int ns2 = getenv("CV_HL_NS2") ? atoi(getenv("CV_HL_NS2")) : 7;
                // split-K reduced by the last-arriving workgroup (cv_conv_desc.split_tickets): correct and bit-identical
                // (tests), but every workgroup's agent-scope release writes the XCD's L2 back and the forward takes 3.79
                // instead of 2.50 ms (profiles/r2/fused_finish.txt) - the executor only hands the tickets over when
                // CV_HL_FUSE_FINISH=1; direct callers of cv_sp_conv_f32 get what they ask for
                static const bool fuse_on = !(getenv("CV_HL_FUSE_FINISH") && atoi(getenv("CV_HL_FUSE_FINISH")) == 0);
                if (!(fuse_on && ax.splits > 1 && !ax.perm_per_split && !ax.xcd_tiles &&
                      (long long)gridx.x * gridx.y <= CV_SPLIT_TICKETS)) ax.tickets = nullptr;
                // CV_HL_NS1 (round-3 experiment): bit nb-1 = ONE slot for NB = nb
                static const int ns1 = getenv("CV_HL_NS1") ? atoi(getenv("CV_HL_NS1")) : 0;
                if ((ns1 >> (NB - 1)) & 1) conv_hl<NB, 1, 4><<<gridx, THREADS, 0, st>>>(ax);
                else if ((ns2 >> (NB - 1)) & 1) conv_hl<NB, 2, 4><<<gridx, THREADS, 0, st>>>(ax);
                else conv_hl<NB, 3, 4><<<gridx, THREADS, 0, st>>>(ax);
                CV_LAUNCH_CHECK();
                if (a.splits > 1 && !ax.tickets) return launch_finish(a, st);
                return CV_OK;
            }
        }
        CV_LAUNCH_CHECK();
        if (a.splits > 1) return launch_finish(a, st);
        return CV_OK;
    }
    if (vec && a.wp6 && per_wg <= WP_NPRE) {
        if (a.pieces == 2) conv_rows_wp<NB, 2><<<gridx, THREADS, 0, st>>>(ax);
        else if (a.pieces == 1) conv_rows_wp<NB, 1><<<gridx, THREADS, 0, st>>>(ax);
        else conv_rows_wp<NB, 3><<<gridx, THREADS, 0, st>>>(ax);
        CV_LAUNCH_CHECK();
        if (a.splits > 1) return launch_finish(a, st);
        return CV_OK;
    }
    // more offsets per workgroup than the default instance prefetches (flavour 1 with a whole 3x3x3 kernel, mask groups of > 10
    // offsets): the instance that keeps 28 map rows per tile (more LDS, fewer workgroups per CU)
    if (vec && a.wp6 && per_wg <= WP_NPRE_BIG) {
        if (a.pieces == 2) conv_rows_wp<NB, 2, WP_NPRE_BIG><<<gridx, THREADS, 0, st>>>(ax);
        else if (a.pieces == 1) conv_rows_wp<NB, 1, WP_NPRE_BIG><<<gridx, THREADS, 0, st>>>(ax);
        else conv_rows_wp<NB, 3, WP_NPRE_BIG><<<gridx, THREADS, 0, st>>>(ax);
        CV_LAUNCH_CHECK();
        if (a.splits > 1) return launch_finish(a, st);
        return CV_OK;
    }
    CV_REQUIRE(a.pieces != 1 || !(vec && a.wp6), CV_EINVAL, "the bf16 compute mode takes at most %d kernel offsets per workgroup", WP_NPRE_BIG);
    if (vec) conv_rows<NB, true><<<grid, THREADS, 0, st>>>(a);
    else conv_rows<NB, false><<<grid, THREADS, 0, st>>>(a);
    CV_LAUNCH_CHECK();
    if (a.splits > 1) return launch_finish(a, st);
    return CV_OK;
}

// 32-column blocks per workgroup.  Wider than 96 columns is split over blockIdx.y in 64-column workgroups: measured 118 -> 97 us (2349 rows, 256 -> 256), 143 -> 117 us (9929 rows, 128 -> 128),
// 40 -> 33 us (494 rows) against 128-column workgroups - more, lighter workgroups hide the gather latency better
// than the saved operand re-reads are worth; 96 columns stay one workgroup (64 + 32 is uneven: 144 -> 180 us).
}  // namespace
namespace cvsc {
int nb_full(int cout) { return cout <= 32 ? 1 : cout <= 64 ? 2 : cout <= 96 ? 3 : 4; }
}  // namespace cvsc
namespace {
int nb_for(int cout, long long n_out) {
    static const int nb_max = getenv("CV_NB_MAX") ? atoi(getenv("CV_NB_MAX")) : 0;
    if (nb_max > 0) return std::min(nb_max, nb_full(cout));
    if (cout <= 32) return 1;
    if (cout <= 64) return 2;
    if (cout <= 96) return 3;
    static const int nb_wide = getenv("CV_NB_WIDE") ? atoi(getenv("CV_NB_WIDE")) : 2;      // 128 / 256 columns: 64-column workgroups
    // CV_NB_COARSE (round-3 experiment): column blocks per workgroup on the levels below CV_NB_COARSE_ROWS rows - their
    // launches have two workgroups per CU at most, so wider workgroups cost no occupancy and halve the row gathers
    static const int nb_coarse = getenv("CV_NB_COARSE") ? atoi(getenv("CV_NB_COARSE")) : 0;
    static const long long coarse_rows = getenv("CV_NB_COARSE_ROWS") ? atoll(getenv("CV_NB_COARSE_ROWS")) : 16384;
    if (nb_coarse > 0 && n_out < coarse_rows) return std::max(1, std::min(nb_coarse, 4));
    // (until round 5 the levels below 1024 rows took 32-column workgroups - measured with the fp32-row kernels of round 1; on the hl
    // kernels 64 columns are better alone and with scenes in flight: net 2.32 -> 2.29-2.30 ms, 592 -> 599-600 scenes/s,
    // profiles/r5/nb_coarse.txt; 96 / 128 columns: 591 / 583)
    return std::max(1, std::min(nb_wide, 2));
}

// workgroups a split launch aims at: CV_SPLIT_TARGET or 768 until cv_sp_set_split_target changes it
std::atomic<long long> g_split_target{-1};
thread_local long long t_split_target = 0;       // cv_sp_set_split_target_thread: this thread's launches (0 = the process-wide value)
long long process_split_target() {
    long long v = g_split_target.load(std::memory_order_relaxed);
    if (v <= 0) {
        v = getenv("CV_SPLIT_TARGET") ? std::max(1ll, atoll(getenv("CV_SPLIT_TARGET"))) : 768;
        g_split_target.store(v, std::memory_order_relaxed);
    }
    return v;
}
long long split_target() { return t_split_target > 0 ? t_split_target : process_split_target(); }

// Enough workgroups to fill 256 CUs a few times with short dependent chains: split the (offset,
// chunk) units over blockIdx.z; the partial tiles cost 8 bytes of traffic per output element per split.
int pick_splits(long long n_out, int cout, int K, int cin, bool vec) {
    const int nb = nb_for(cout, n_out);
    const long long tiles = ((n_out + TM - 1) / TM) * ((cout + nb * 32 - 1) / (nb * 32));
    const long long units = vec ? (long long)K * (cin / KC) : ((long long)K * cin + KC - 1) / KC;
    if (tiles >= 384 || units <= 1) return 1;
    // workgroups a split launch aims at.  768 since round 5 (the one-scene-at-a-time optimum on the hl kernels: 384 / 512 / 768 /
    // 1024 = 339 / 341-344 / 347.0 / 347.2 scenes/s one in flight, profiles/r5/one_in_flight_defaults.txt; hosts with scenes in flight
    // set their own value).  512 from round 2 on: 384 / 512 / 768 / 1024 = 513 / 508 / 496 / 491 scenes/s six in
    // flight, net 2.58 / 2.51 / 2.47 / 2.47 ms one in flight - the partial tiles of the coarse levels are 0.8 GB of the
    // 1.7 GB a forward writes, and with several scenes in flight the other scenes fill the chip, not the splits
    // (cv_sp_set_split_target: with EIGHT scenes in flight 256 gives 562-566 scenes/s against 550 for 512 and 549-555
    // for 128, profiles/r3/split_target_8streams.txt - the host picks the value by how many scenes it keeps in flight)
    const long long target = split_target();
    long long s = (target + tiles - 1) / tiles;
    // (measured without gain: sizing by the chip's resident workgroup slots, floor(256 x waves-per-SIMD / tiles), so that
    // no second round of workgroups starts: 257 vs 265 scenes/s one scene in flight, 401-409 vs 410-412 with six)
    s = std::min(s, units);
    static const long long traffic_mb = getenv("CV_SPLIT_TRAFFIC_MB") ? atoll(getenv("CV_SPLIT_TRAFFIC_MB")) : 24;
    const long long by_traffic = (traffic_mb << 20) / std::max<long long>(1, n_out * cout * 4);   // <= 24 MB of partials
    s = std::min(s, std::max<long long>(by_traffic, 2));
    s = std::min<long long>(s, 64);
    return (int)std::max<long long>(s, 1);
}

}  // namespace

int cv_sp_set_option(const char* name, long long value, long long* previous) {
    CV_REQUIRE(name, CV_EINVAL, "null option name");
    std::atomic<long long>* o = !strcmp(name, "hd_mask") ? &g_opt_hd_mask : !strcmp(name, "hd_min_rows") ? &g_opt_hd_min_rows :
                                !strcmp(name, "hd_shape") ? &g_opt_hd_shape : !strcmp(name, "zskip") ? &g_opt_zskip : nullptr;
    CV_REQUIRE(o, CV_EINVAL, "unknown option '%s' (hd_mask, hd_min_rows, hd_shape, zskip)", name);
    const long long before = o->exchange(value, std::memory_order_relaxed);
    if (previous) *previous = before;
    return CV_OK;
}

int cv_sp_get_option(const char* name, long long* value) {
    CV_REQUIRE(name && value, CV_EINVAL, "null option name / value");
    const std::atomic<long long>* o = !strcmp(name, "hd_mask") ? &g_opt_hd_mask : !strcmp(name, "hd_min_rows") ? &g_opt_hd_min_rows :
                                      !strcmp(name, "hd_shape") ? &g_opt_hd_shape : !strcmp(name, "zskip") ? &g_opt_zskip : nullptr;
    CV_REQUIRE(o, CV_EINVAL, "unknown option '%s'", name);
    *value = o->load(std::memory_order_relaxed);
    return CV_OK;
}

int cv_sp_set_ablation(int bits) { return cvsc::g_ablation.exchange(bits, std::memory_order_relaxed); }

int cv_sp_set_split_target_thread(int workgroups) {
    const int before = (int)t_split_target;
    t_split_target = workgroups > 0 ? workgroups : 0;
    return before;
}

int cv_sp_set_split_target(int workgroups) {
    const int before = (int)process_split_target();          // (the process-wide value, whatever the calling thread's own is)
    g_split_target.store(workgroups > 0 ? workgroups : -1, std::memory_order_relaxed);
    return before;
}

int cv_sp_mask_perms_batch(const CvPermJob* jobs, int n_jobs, void* d_ws, size_t ws_bytes, void* stream, bool pre_zeroed) {
    CV_REQUIRE(jobs && d_ws && n_jobs > 0 && n_jobs <= CV_MAX_PERM_JOBS, CV_EINVAL, "bad mask perm batch");
    PermJobsDev d;
    d.n = n_jobs;
    int groups = 0;
    long long max_n = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const CvPermJob& j = jobs[i];
        CV_REQUIRE(j.nbr && j.perm && j.n > 0 && j.K > 0 && j.groups >= 1 && j.groups <= j.K, CV_EINVAL,
                   "bad mask perm job %d", i);
        CV_REQUIRE((j.K + j.groups - 1) / j.groups <= 10, CV_EINVAL, "at most 10 kernel offsets per group");
        d.j[i] = j;
        d.group_begin[i] = groups;
        groups += j.groups;
        max_n = std::max(max_n, j.n);
    }
    d.group_begin[n_jobs] = groups;
    // workspace: [groups][1024] bin counts, then [groups][1024] running counts of the scatter (both zero at the start)
    CV_REQUIRE(ws_bytes >= sizeof(int) * (size_t)groups * MP_BINS * 2, CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int* hist = static_cast<int*>(d_ws);
    int* cursor = hist + (size_t)groups * MP_BINS;
    if (!pre_zeroed) CV_HIP_CHECK(hipMemsetAsync(hist, 0, sizeof(int) * (size_t)groups * MP_BINS * 2, st));
    dim3 grid((unsigned)((max_n + MP_THREADS - 1) / MP_THREADS), (unsigned)groups);
    mp_hist_batch<<<grid, MP_THREADS, 0, st>>>(d, hist);
    CV_LAUNCH_CHECK();
    mp_scatter_batch<<<grid, MP_THREADS, 0, st>>>(d, hist, cursor);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

extern "C" {

size_t cv_sp_conv_workspace_bytes(long long n_out, int cout, int K) {
    if (n_out <= 0 || cout <= 0 || K <= 0) return 0;
    const size_t one = sizeof(float) * (size_t)n_out * (size_t)cout;
    return 256 + std::min<size_t>(64 * one, std::max<size_t>((size_t)48 << 20, 3 * one));   // see pick_splits (and the
                                                                                            // 3-way split of conv_hl)
}

int cv_sp_conv_f32(const cv_conv_desc* d, void* stream) {
    CV_REQUIRE(d, CV_EINVAL, "null descriptor");
    CV_REQUIRE(d->in && d->weight && d->out, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(d->n_in > 0 && d->n_out > 0 && d->cin > 0 && d->cout > 0 && d->K > 0, CV_EINVAL, "bad conv sizes");
    CV_REQUIRE(d->nbr || (d->K == 1 && d->n_in == d->n_out), CV_EINVAL, "a kernel map is required unless K == 1");
    CV_REQUIRE(d->in_ld >= d->cin && d->out_ld >= d->cout && (!d->residual || d->res_ld >= d->cout) &&
                   (!d->acc_in || d->acc_ld >= d->cout), CV_EINVAL, "bad leading dimension");
    CV_REQUIRE(d->out != d->in, CV_EINVAL, "conv cannot run in place");
    const int jb = d->j_begin, je = d->j_end > 0 ? d->j_end : d->K;
    CV_REQUIRE(jb >= 0 && jb < je && je <= d->K, CV_EINVAL, "bad kernel offset range");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvArgs a{d->in, d->n_in, d->in_ld, d->cin, d->weight, d->K, d->cout, d->nbr, d->n_out, d->scale,
               d->shift, d->residual, d->res_ld, d->relu, d->out, d->out_ld, 1, nullptr, d->row_perm, 0, jb, je,
               d->acc_in, d->acc_ld, d->plan_ent, d->plan_cnt,
               reinterpret_cast<const float4*>(d->weight_packed), 0, nullptr, 0, 0,
               static_cast<const unsigned short*>(d->weight_x6), d->in2, d->in2_ld, d->cin2,
               static_cast<const unsigned short*>(d->weight2_x6), d->weight_pieces == 2 ? 2 : d->weight_pieces == 1 ? 1 : 3,
               d->acc_scale != 0.f ? d->acc_scale : 1.f, d->range_flag, d->in_hl, d->out_hl, d->res_hl, 0, d->split_tickets};
    if (d->in_hl || d->out_hl || d->res_hl) {
        CV_REQUIRE(!d->in_hl || (d->weight_pieces == 2 && d->cin % 32 == 0 && d->in_ld % 32 == 0 &&
                                 (reinterpret_cast<uintptr_t>(d->in) & 127) == 0 &&
                                 (!d->in2 || (d->cin2 % 32 == 0 && d->in2_ld % 32 == 0 &&
                                              (reinterpret_cast<uintptr_t>(d->in2) & 127) == 0))),
                   CV_EINVAL, "hl-format input: fp16-pair weights, channels and leading dimensions %% 32 == 0, 128-byte aligned rows");
        CV_REQUIRE(!d->out_hl || (d->cout % 32 == 0 && d->out_ld % 32 == 0 && (reinterpret_cast<uintptr_t>(d->out) & 127) == 0),
                   CV_EINVAL, "hl-format output: Cout and leading dimension %% 32 == 0, 128-byte aligned rows");
        CV_REQUIRE(!d->res_hl || !d->residual || (d->res_ld % 32 == 0 && (reinterpret_cast<uintptr_t>(d->residual) & 127) == 0),
                   CV_EINVAL, "hl-format residual: leading dimension %% 32 == 0, 128-byte aligned rows");
        CV_REQUIRE(d->flavour == 0, CV_EINVAL, "the hl format runs with flavour 0 only");
        // conv_hl forms its gather addresses in 32 bits (row * row bytes + chunk offset)
        CV_REQUIRE(!d->in_hl || ((unsigned long long)d->n_in * (unsigned long long)d->in_ld * 4ull + (unsigned long long)d->cin * 4ull < (1ull << 32) &&
                                 (!d->in2 || (unsigned long long)d->n_out * (unsigned long long)d->in2_ld * 4ull + (unsigned long long)d->cin2 * 4ull < (1ull << 32))),
                   CV_EINVAL, "hl-format input larger than 4 GiB (rows x leading dimension x 4): split the batch");
    }
    CV_REQUIRE(d->weight_pieces >= 0 && d->weight_pieces <= 3, CV_EINVAL,
               "weight_pieces is 0/3 (bf16 triples), 2 (fp16 pairs) or 1 (single bf16 product)");
    CV_REQUIRE(d->weight_pieces != 1 || (d->weight_x6 && d->cin % KC == 0), CV_EINVAL,
               "weight_pieces = 1 needs weight_x6 from cv_sp_pack_weights_bf16_f32 and Cin %% 32 == 0");
    const bool stem_h2 = d->weight_pieces == 2 && d->weight_x6 && d->cout == 32 && (d->cin == 3 || d->cin == 6) && d->K <= 128;
    CV_REQUIRE(d->weight_pieces != 2 || stem_h2 || (d->weight_x6 && d->cin % KC == 0), CV_EINVAL,
               "weight_pieces = 2 needs weight_x6 from cv_sp_pack_weights_h2_f32 and Cin %% 32 == 0 (or the stem shape with "
               "cv_sp_pack_weights_stem_h2_f32)");

    if (!d->in_hl) a.tickets = nullptr;       // the in-launch split-K reduction exists in conv_hl only
    CV_REQUIRE(!d->plan_ent && !d->plan_cnt && !d->weight_packed && (d->flavour == 0 || d->flavour == 1), CV_EINVAL,
               "flavours 3 / 4 (the experimental wave / tile kernels of rounds 1-3) are gone: flavour is 0 or 1, plan_ent / plan_cnt / "
               "weight_packed must be NULL");
    a.wide = d->cout % 4 == 0 && d->out_ld % 4 == 0 && (!d->residual || d->res_ld % 4 == 0) &&
             (!d->acc_in || d->acc_ld % 4 == 0) &&
             ((reinterpret_cast<uintptr_t>(d->out) | reinterpret_cast<uintptr_t>(d->residual) |
               reinterpret_cast<uintptr_t>(d->acc_in) | reinterpret_cast<uintptr_t>(d->scale) |
               reinterpret_cast<uintptr_t>(d->shift) | reinterpret_cast<uintptr_t>(d->ws)) & 15) == 0;
    const bool vec = (d->cin % KC == 0) && (d->in_ld % 4 == 0) && (d->cout % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(d->in) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(d->weight) & 15) == 0);
    // packed weights are only read by the vector path: without it the launch would fall back to `weight` - which a caller
    // that packed W^T straight from the forward weights (cv_sp_pack_weights_t_f32) holds in the OTHER layout
    CV_REQUIRE(!d->weight_x6 || vec || stem_h2, CV_EINVAL,
               "weight_x6 needs the vector path: Cin %% 32 == 0, Cout %% 4 == 0, in_ld %% 4 == 0, 16-byte aligned in / weight");
    if (d->in2) {
        CV_REQUIRE(vec && d->weight_x6 && d->weight2_x6 && d->cin2 > 0 && d->cin2 % KC == 0 && d->in2_ld >= d->cin2 &&
                       d->in2_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(d->in2) & 15) == 0,
                   CV_EINVAL, "a second source needs the bf16x6 vector path (weight_x6, weight2_x6, Cin2 %% 32 == 0, "
                              "16-byte aligned rows)");
    }
    if (!vec && d->cout == 32 && (d->cin == 3 || d->cin == 6) && d->nbr && !d->row_perm && d->perm_groups <= 1 &&
        d->flavour == 0) {
        if (d->weight_x6 && d->weight_pieces == 2 && d->K <= 128 && a.wide && !d->acc_in && d->j_begin == 0 &&
            (d->j_end == 0 || d->j_end == d->K)) {
            // matrix-core stem (weights from cv_sp_pack_weights_stem_h2_f32, BatchNorm scale folded in)
            const unsigned gm = (unsigned)((d->n_out + TM - 1) / TM);
            if (d->cin == 3) conv_stem_mfma<3><<<gm, THREADS, 0, st>>>(a);
            else conv_stem_mfma<6><<<gm, THREADS, 0, st>>>(a);
            CV_LAUNCH_CHECK();
            return CV_OK;
        }
    }
    a.acc_scale_dev = d->acc_scale_dev;
    if (d->perm_groups > 1) {
        // offsets split into perm_groups contiguous groups, each processed in its own row order, all in
        // one launch (grid.z); partial tiles are reduced by conv_finish
        const size_t need = sizeof(float) * (size_t)d->perm_groups * (size_t)d->n_out * (size_t)d->cout;
        CV_REQUIRE(d->row_perm && vec && d->ws && d->ws_bytes >= need, CV_EINVAL,
                   "perm_groups needs row_perm[groups][n_out], Cin %% 32 == 0 and a workspace of %zu bytes", need);
        a.splits = d->perm_groups;
        a.perm_per_split = 1;
        a.partial = static_cast<float*>(d->ws);
        if (d->perm_has_map && jb == 0 && je == d->K && d->nbr) {
            a.nbr_perm = d->row_perm + (long long)d->perm_groups * d->n_out;
            a.nbr_perm_w = (d->K + d->perm_groups - 1) / d->perm_groups;
            // zskip: tiles whose rows have no neighbour in their group write no partial tile and the finish launch does not read
            // one (the validity bytes sit behind the map rows).  Not with a second source (its chunks are dealt to the groups:
            // every row gets a sum from them) and only where conv_finish_small finishes.
            if (g_opt_zskip.load(std::memory_order_relaxed) && !d->in2 && a.wide && d->perm_groups <= FINISH_SMALL_MAX && d->in_hl)
                a.gvalid = reinterpret_cast<const unsigned char*>(a.nbr_perm + (long long)d->perm_groups * d->n_out * a.nbr_perm_w);
        }
        // Chain of group launches (CV_GROUP_CHAIN=1, round-3 experiment): group 0 writes its sums to the workspace, every
        // later group adds its own to what it reads back (acc_in, same thread, same element: in place), the last one runs
        // the epilogue - no partial tile per group, no finish launch (w 1 + (r 1 + w 1) (G - 2) + r 1 + w out instead of
        // w G + r G + w out tile sets); the price is G launches of n_out / 128 workgroups instead of one of G times that.
        static const bool chain_on = getenv("CV_GROUP_CHAIN") && atoi(getenv("CV_GROUP_CHAIN")) != 0;
        if (chain_on && d->in_hl && d->flavour == 0 && !d->acc_in && a.wide) {
            const int G = d->perm_groups, nj = je - jb;
            float* tmp = static_cast<float*>(d->ws);
            for (int g = 0; g < G; ++g) {
                ConvArgs ag = a;
                ag.splits = 1;
                ag.perm_per_split = 0;
                ag.partial = nullptr;
                ag.tickets = nullptr;
                ag.gvalid = nullptr;                 // (every launch of the chain writes every row)
                ag.j_begin = jb + (int)((long long)nj * g / G);
                ag.j_end = jb + (int)((long long)nj * (g + 1) / G);
                ag.row_perm = d->row_perm + (long long)g * d->n_out;
                if (a.nbr_perm) ag.nbr_perm = a.nbr_perm + (long long)g * d->n_out * a.nbr_perm_w;
                if (g > 0) { ag.acc_in = tmp; ag.acc_ld = d->cout; ag.in2 = nullptr; ag.cin2 = 0; }
                if (g < G - 1) {
                    ag.out = tmp; ag.out_ld = d->cout; ag.out_hl = 0;
                    ag.scale = nullptr; ag.shift = nullptr; ag.res = nullptr; ag.relu = 0; ag.range_flag = nullptr;
                }
                if (ag.j_end <= ag.j_begin) continue;
                int rc2;
                switch (nb_for(d->cout, d->n_out)) {
                    case 1: rc2 = launch_rows<1>(ag, vec, st); break;
                    case 2: rc2 = launch_rows<2>(ag, vec, st); break;
                    case 3: rc2 = launch_rows<3>(ag, vec, st); break;
                    default: rc2 = launch_rows<4>(ag, vec, st); break;
                }
                if (rc2) return rc2;
            }
            return CV_OK;
        }
    } else if (d->flavour == 0) {
        int sp = pick_splits(d->n_out, d->cout, je - jb, d->cin, vec);
        if (d->in_hl || (vec && a.wp6)) {
            // conv_hl / conv_rows_wp keep the map entries of at most WP_NPRE offsets per workgroup: launches that would not
            // be split (>= 384 tiles below the mask-sorting threshold, e.g. 13k rows x 256 columns) are split over just
            // enough workgroups
            const int nj = je - jb;
            const int per_wg = sp <= 1 ? nj : (nj + sp - 1) / sp + 1;
            if (per_wg > WP_NPRE) sp = std::max(sp, (nj + WP_NPRE - 2) / (WP_NPRE - 1));
        }
        const size_t need = sizeof(float) * (size_t)sp * (size_t)d->n_out * (size_t)d->cout;
        CV_REQUIRE(!d->in_hl || sp <= 1 || (d->ws && d->ws_bytes >= need), CV_ENOMEM,
                   "hl-format convolution needs a workspace of %zu bytes (cv_sp_conv_workspace_bytes)", need);
        if (sp > 1 && d->ws && d->ws_bytes >= need) {
            a.splits = sp;
            a.partial = static_cast<float*>(d->ws);
        }
    }
    switch (nb_for(d->cout, d->n_out)) {
        case 1: return launch_rows<1>(a, vec, st);
        case 2: return launch_rows<2>(a, vec, st);
        case 3: return launch_rows<3>(a, vec, st);
        default: return launch_rows<4>(a, vec, st);
    }
}

// d_keys[n] = bit mask of valid neighbours among kernel offsets [j_begin, j_end) of each output row.
// Sorting rows by this key groups rows that need the same offsets (cv_conv_desc.row_perm).
int cv_sp_mask_keys(const int32_t* d_nbr, long long n, int K, int j_begin, int j_end, long long* d_keys,
                    void* stream) {
    CV_REQUIRE(d_nbr && d_keys && n > 0 && K > 0 && j_begin >= 0 && j_begin < j_end && j_end <= K &&
                   j_end - j_begin <= 63, CV_EINVAL, "bad mask key arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    mask_keys<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(d_nbr, n, K, j_begin, j_end, d_keys);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

// d_perm[groups][n] = rows ordered by the neighbour bit mask of each contiguous group of the K offsets
// (ceil(K/groups) <= 10).  d_ws: groups*1024 ints of scratch.  Asynchronous, three launches.
int cv_sp_mask_perms(const int32_t* d_nbr, long long n, int K, int groups, int32_t* d_perm, void* d_ws,
                     size_t ws_bytes, int with_map, void* stream) {
    CV_REQUIRE(d_nbr && d_perm && d_ws && n > 0 && K > 0 && groups >= 1 && groups <= K, CV_EINVAL,
               "bad mask perm arguments");
    CV_REQUIRE((K + groups - 1) / groups <= 10, CV_EINVAL, "at most 10 kernel offsets per group");
    CV_REQUIRE(ws_bytes >= sizeof(int) * (size_t)groups * MP_BINS, CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int* hist = static_cast<int*>(d_ws);
    CV_HIP_CHECK(hipMemsetAsync(hist, 0, sizeof(int) * (size_t)groups * MP_BINS, st));
    dim3 grid((unsigned)((n + MP_THREADS - 1) / MP_THREADS), (unsigned)groups);
    mp_hist<<<grid, MP_THREADS, 0, st>>>(d_nbr, n, K, groups, hist);
    CV_LAUNCH_CHECK();
    mp_scan<<<groups, MP_BINS, 0, st>>>(hist);
    CV_LAUNCH_CHECK();
    mp_scatter<<<grid, MP_THREADS, 0, st>>>(d_nbr, n, K, groups, hist, d_perm,
                                            with_map ? d_perm + (long long)groups * n : nullptr, (K + groups - 1) / groups);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

// d_wp6[3*K*cin*cout] (16-bit words) = d_w[K][cin][cout] split into three bf16 pieces per value and laid out per
// (offset, 32-channel chunk) as [piece][cout][32 channels] for conv_rows_wp (cin % 32 == 0).  Redo when weights change.
int cv_sp_pack_weights_x6_f32(const float* d_w, int K, int cin, int cout, const float* d_col_scale, void* d_wp6,
                              void* stream) {
    CV_REQUIRE(d_w && d_wp6 && K > 0 && cin > 0 && cout > 0, CV_EINVAL, "bad pack_weights_x6 arguments");
    CV_REQUIRE(cin % 32 == 0, CV_EINVAL, "pack_weights_x6 needs Cin %% 32 == 0 (got %d)", cin);
    CV_REQUIRE((reinterpret_cast<uintptr_t>(d_wp6) & 15) == 0, CV_EINVAL, "d_wp6 must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long total = (long long)K * cin * cout / 2;
    pack_weights_x6<<<(unsigned)std::min<long long>((total + 255) / 256, 8192), 256, 0, st>>>(
        d_w, K, cin, cout, d_col_scale, static_cast<unsigned short*>(d_wp6));
    CV_LAUNCH_CHECK();
    return CV_OK;
}

// d_wp[K*cin*cout] (16-bit words) = d_w * d_col_scale rounded to bf16 (RNE), laid out per (offset, 32-channel
// chunk) as [cout][32 channels] (cv_conv_desc.weight_pieces = 1: the opt-in bf16 compute mode).
int cv_sp_pack_weights_bf16_f32(const float* d_w, int K, int cin, int cout, const float* d_col_scale, void* d_wp,
                                void* stream) {
    CV_REQUIRE(d_w && d_wp && K > 0 && cin > 0 && cout > 0, CV_EINVAL, "bad pack_weights_bf16 arguments");
    CV_REQUIRE(cin % 32 == 0, CV_EINVAL, "pack_weights_bf16 needs Cin %% 32 == 0 (got %d)", cin);
    CV_REQUIRE((reinterpret_cast<uintptr_t>(d_wp) & 15) == 0, CV_EINVAL, "d_wp must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long total = (long long)K * cin * cout / 2;
    pack_weights_b1<<<(unsigned)std::min<long long>((total + 255) / 256, 8192), 256, 0, st>>>(
        d_w, K, cin, cout, d_col_scale, static_cast<unsigned short*>(d_wp));
    CV_LAUNCH_CHECK();
    return CV_OK;
}

// d_wp[2*K*cin*cout] (16-bit words) = d_w * d_col_scale * 2^scale_log2 split into two fp16 pieces per value, laid
// out per (offset, 32-channel chunk) as [piece][cout][32 channels] (cv_conv_desc.weight_pieces = 2; pass
// acc_scale = 2^-scale_log2).  Choose scale_log2 so that the largest scaled magnitude is about 2^13: the low pieces
// of weights down to 2^-14 of the largest one then stay normal fp16 numbers.
int cv_sp_pack_weights_h2_f32(const float* d_w, int K, int cin, int cout, const float* d_col_scale, int scale_log2,
                              void* d_wp, void* stream) {
    CV_REQUIRE(d_w && d_wp && K > 0 && cin > 0 && cout > 0, CV_EINVAL, "bad pack_weights_h2 arguments");
    CV_REQUIRE(cin % 32 == 0, CV_EINVAL, "pack_weights_h2 needs Cin %% 32 == 0 (got %d)", cin);
    CV_REQUIRE(scale_log2 >= -60 && scale_log2 <= 60, CV_EINVAL, "scale_log2 out of range");
    CV_REQUIRE((reinterpret_cast<uintptr_t>(d_wp) & 15) == 0, CV_EINVAL, "d_wp must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long total = (long long)K * cin * cout / 2;
    pack_weights_h2<<<(unsigned)std::min<long long>((total + 255) / 256, 8192), 256, 0, st>>>(
        d_w, K, cin, cout, d_col_scale, ldexpf(1.f, scale_log2), static_cast<unsigned short*>(d_wp));
    CV_LAUNCH_CHECK();
    return CV_OK;
}

// Packed weights of the TRANSPOSED convolution (the input gradient: dX = conv over the transposed map with W_j^T) straight
// from the forward weights d_w[K][rows][cols]: the packed tensor is that of W'[K][cin = cols][cout = rows] with
// W'[j][a][b] = d_w[j][b][a] - no transposed copy of the weights per layer and step.  pieces 3: bf16 triples, 2: fp16 pairs
// (times 2^scale_log2), 1: one bf16 plane.
int cv_sp_pack_weights_t_f32(const float* d_w, int K, int rows, int cols, int pieces, int scale_log2, void* d_wp, void* stream) {
    CV_REQUIRE(d_w && d_wp && K > 0 && rows > 0 && cols > 0, CV_EINVAL, "bad pack_weights_t arguments");
    const int cin = cols, cout = rows;
    CV_REQUIRE(cin % 32 == 0, CV_EINVAL, "pack_weights_t needs %d (the input channels of the transposed convolution) %% 32 == 0", cin);
    CV_REQUIRE(pieces >= 1 && pieces <= 3 && scale_log2 >= -60 && scale_log2 <= 60, CV_EINVAL, "pieces is 1, 2 or 3");
    CV_REQUIRE((reinterpret_cast<uintptr_t>(d_wp) & 15) == 0, CV_EINVAL, "d_wp must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long total = (long long)K * cin * cout / 2;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 8192);
    unsigned short* wp = static_cast<unsigned short*>(d_wp);
    if (pieces == 3) pack_weights_x6<<<grid, 256, 0, st>>>(d_w, K, cin, cout, nullptr, wp, 1);
    else if (pieces == 2) pack_weights_h2<<<grid, 256, 0, st>>>(d_w, K, cin, cout, nullptr, ldexpf(1.f, scale_log2), wp, 1);
    else pack_weights_b1<<<grid, 256, 0, st>>>(d_w, K, cin, cout, nullptr, wp, 1);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_sp_pack_weights_h2_batch_f32(const cv_pack_job* h_jobs, int n_jobs, void* d_jobs, void* stream) {
    CV_REQUIRE(n_jobs >= 0 && (n_jobs == 0 || (h_jobs && d_jobs)), CV_EINVAL, "bad pack batch");
    if (n_jobs == 0) return CV_OK;
    for (int i = 0; i < n_jobs; ++i) {
        const cv_pack_job& j = h_jobs[i];
        CV_REQUIRE(j.w && j.wp && j.K > 0 && j.cin > 0 && j.cout > 0 && j.cin % 32 == 0, CV_EINVAL,
                   "pack job %d: null pointer or Cin %% 32 != 0 (K %d, Cin %d, Cout %d)", i, j.K, j.cin, j.cout);
        CV_REQUIRE(j.scale_log2 >= -60 && j.scale_log2 <= 60 && (j.trans == 0 || j.trans == 1), CV_EINVAL, "pack job %d: scale_log2 / trans out of range", i);
        CV_REQUIRE((reinterpret_cast<uintptr_t>(j.wp) & 15) == 0, CV_EINVAL, "pack job %d: wp must be 16-byte aligned", i);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    CV_HIP_CHECK(hipMemcpyAsync(d_jobs, h_jobs, sizeof(cv_pack_job) * (size_t)n_jobs, hipMemcpyHostToDevice, st));
    pack_weights_h2_batch<<<dim3(64, (unsigned)n_jobs), 256, 0, st>>>(static_cast<const cv_pack_job*>(d_jobs));
    CV_LAUNCH_CHECK();
    return CV_OK;
}

constexpr long long WGRAD_TARGET_TASKS = 8192;

// ---- training support -------------------------------------------------------------------------
int cv_sp_transpose_map(const int32_t* d_nbr, long long n_out, int K, long long n_in, int32_t* d_nbr_t, void* stream) {
    CV_REQUIRE(d_nbr && d_nbr_t && n_out > 0 && n_in > 0 && K > 0, CV_EINVAL, "bad transpose_map arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    CV_HIP_CHECK(hipMemsetAsync(d_nbr_t, 0xff, sizeof(int) * (size_t)n_in * K, st));
    transpose_map<<<(unsigned)((n_out * K + 255) / 256), 256, 0, st>>>(d_nbr, n_out, K, d_nbr_t);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

// Row splits of the weight-gradient reduction: enough (offset, ci block, co block, split) wave tasks to fill
// 256 CUs x 4 SIMDs a few times over even on the coarse levels (a few thousand rows), at least 128 rows each;
// the centre offset of an odd cubic kernel and its face neighbours hold the most pairs: they go first and get
// twice the splits (measured: profiles/wgrad_micro.py).
// input-channel blocks per wave (see conv_wgrad): as many as divide Cin / 32 while NA x NB accumulators fit
static int wgrad_na(int cin, int cout) {
    static const int na_max = getenv("CV_WGRAD_NA") ? atoi(getenv("CV_WGRAD_NA")) : 4;
    const int nb = nb_full(cout);
    int na = 1;
    if (cin % 32 == 0 && cin >= 64) {
        const int b = cin / 32;
        if (nb == 4) na = b % 2 == 0 ? 2 : 1;
        else if (nb == 3) na = b % 3 == 0 ? 3 : (b % 2 == 0 ? 2 : 1);
        else na = b % 4 == 0 ? 4 : (b % 2 == 0 ? 2 : (b % 3 == 0 ? 3 : 1));
    }
    return std::max(1, std::min(na, na_max));
}

static long long wgrad_plan(long long n_out, int cin, int cout, int K, WgradPlan* plan) {
    const int nb = nb_full(cout), na = wgrad_na(cin, cout);
    const int tiles = ((cin + na * 32 - 1) / (na * 32)) * ((cout + nb * 32 - 1) / (nb * 32));
    int ks = 1;
    while (ks * ks * ks < K) ++ks;
    const bool cubic = ks * ks * ks == K && (ks & 1) && ks > 1;
    int mult[WG_MAX_K], cls[WG_MAX_K], msum = 0;
    for (int j = 0; j < K; ++j) {
        int d1 = 3;
        if (cubic) {
            const int h = ks / 2;
            d1 = std::abs(j % ks - h) + std::abs(j / ks % ks - h) + std::abs(j / (ks * ks) - h);
        }
        // splits per offset class (centre / face / edge+corner of an odd cubic kernel); CV_WGRAD_MULT="c,f,o" overrides (experiment)
        static int m_c = 2, m_f = 2, m_o = 1;
        static const bool m_env = [] { const char* e = getenv("CV_WGRAD_MULT"); if (e) sscanf(e, "%d,%d,%d", &m_c, &m_f, &m_o); return true; }();
        (void)m_env;
        mult[j] = !cubic ? 1 : d1 == 0 ? m_c : d1 == 1 ? m_f : m_o;
        cls[j] = !cubic ? 2 : d1 == 0 ? 0 : d1 == 1 ? 1 : 2;
        msum += mult[j];
    }
    static const long long target_tasks = getenv("CV_WGRAD_TASKS") ? atoll(getenv("CV_WGRAD_TASKS")) : WGRAD_TARGET_TASKS;
    static const long long cap_env = getenv("CV_WGRAD_CAP") ? atoll(getenv("CV_WGRAD_CAP")) : 64;
    const long long want = (target_tasks + (long long)msum * tiles - 1) / ((long long)msum * tiles);
    // partial tiles <= 96 MB (they are written once and read once by wgrad_reduce); few-offset kernels (1x1, 2x2x2)
    // have few (offset, block) tasks and get their parallelism from the rows instead: up to 1024 row splits
    const long long ws_cap = (96ll << 20) / ((long long)msum * cin * cout * 4);
    const long long cap = K <= 8 ? 1024 : cap_env;
    const long long base = std::max<long long>(1, std::min({cap, want, n_out / (K <= 8 ? 128 : 256), ws_cap}));
    long long slots = 0;
    int rank = 0, tasks = 0;
    for (int pass = 0; pass < 3; ++pass)
        for (int j = 0; j < K; ++j) {
            if (cls[j] != pass) continue;
            const int sp = (int)std::max<long long>(1, std::min<long long>(base * mult[j], n_out / 128));
            if (plan) {
                plan->order[rank] = (short)j;
                plan->nsplit[j] = (short)sp;
                plan->first[j] = (int)slots;
                tasks += sp * tiles;
                plan->task_end[rank] = tasks;
            }
            slots += sp;
            ++rank;
        }
    return slots;
}

size_t cv_sp_wgrad_workspace_bytes(long long n_out, int cin, int cout, int K) {
    if (n_out <= 0 || cin <= 0 || cout <= 0 || K <= 0 || K > WG_MAX_K) return 0;
    return 256 + sizeof(float) * (size_t)wgrad_plan(n_out, cin, cout, K, nullptr) * (size_t)cin * cout;
}

int cv_sp_conv_wgrad_px_f32(const float* d_x, int x_ld, int cin, const float* d_dy, int dy_ld, int cout,
                            const int32_t* d_nbr, int K, long long n_out, float* d_dw, void* d_ws, size_t ws_bytes,
                            int pieces, void* stream) {
    CV_REQUIRE(d_x && d_dy && d_dw && d_ws, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n_out > 0 && cin > 0 && cout > 0 && K > 0 && x_ld >= cin && dy_ld >= cout, CV_EINVAL, "bad wgrad sizes");
    CV_REQUIRE(K <= WG_MAX_K, CV_EINVAL, "kernel volume above 128 is not supported");
    CV_REQUIRE(d_nbr || K == 1, CV_EINVAL, "a kernel map is required unless K == 1");
    CV_REQUIRE(ws_bytes >= cv_sp_wgrad_workspace_bytes(n_out, cin, cout, K), CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    WgradPlan plan = {};
    wgrad_plan(n_out, cin, cout, K, &plan);
    float* partial = static_cast<float*>(d_ws);
    // any Cin: channels beyond Cin are zero lanes of the 32-wide A operand (stem: Cin = 3)
    const unsigned grid = (unsigned)((plan.task_end[K - 1] + 3) / 4);
    CV_REQUIRE(pieces == 0 || pieces == 1 || pieces == 3, CV_EINVAL,
               "pieces is 0 (fp32 MFMA), 3 (six bf16 piece products per fp32 product) or 1 (one bf16 product)");
#define CV_WGRAD_LAUNCH(NAV, NBV)                                                                                    \
    do {                                                                                                             \
        if (pieces == 3) conv_wgrad<NAV, NBV, 3><<<grid, THREADS, 0, st>>>(d_x, x_ld, cin, d_dy, dy_ld, cout, d_nbr, K, n_out, plan, partial); \
        else if (pieces == 1) conv_wgrad<NAV, NBV, 1><<<grid, THREADS, 0, st>>>(d_x, x_ld, cin, d_dy, dy_ld, cout, d_nbr, K, n_out, plan, partial); \
        else conv_wgrad<NAV, NBV, 0><<<grid, THREADS, 0, st>>>(d_x, x_ld, cin, d_dy, dy_ld, cout, d_nbr, K, n_out, plan, partial);   \
    } while (0)
    const int na = wgrad_na(cin, cout);
    switch (nb_full(cout) * 10 + na) {
        case 11: CV_WGRAD_LAUNCH(1, 1); break;
        case 12: CV_WGRAD_LAUNCH(2, 1); break;
        case 13: CV_WGRAD_LAUNCH(3, 1); break;
        case 14: CV_WGRAD_LAUNCH(4, 1); break;
        case 21: CV_WGRAD_LAUNCH(1, 2); break;
        case 22: CV_WGRAD_LAUNCH(2, 2); break;
        case 23: CV_WGRAD_LAUNCH(3, 2); break;
        case 24: CV_WGRAD_LAUNCH(4, 2); break;
        case 31: CV_WGRAD_LAUNCH(1, 3); break;
        case 32: CV_WGRAD_LAUNCH(2, 3); break;
        case 33: CV_WGRAD_LAUNCH(3, 3); break;
        case 41: CV_WGRAD_LAUNCH(1, 4); break;
        case 42: CV_WGRAD_LAUNCH(2, 4); break;
        default: CV_REQUIRE(false, CV_EINVAL, "no weight-gradient kernel for this block shape");
    }
#undef CV_WGRAD_LAUNCH
    CV_LAUNCH_CHECK();
    const long long per = (long long)K * cin * cout;
    wgrad_reduce<<<(unsigned)((per + 255) / 256), 256, 0, st>>>(partial, cin * cout, K, plan, d_dw);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_sp_conv_wgrad_f32(const float* d_x, int x_ld, int cin, const float* d_dy, int dy_ld, int cout,
                         const int32_t* d_nbr, int K, long long n_out, float* d_dw, void* d_ws, size_t ws_bytes,
                         void* stream) {
    // CV_WGRAD_X6=0: exact fp32 products on v_mfma_f32_32x32x2_f32 instead of six bf16 piece products
    static const bool x6 = !(getenv("CV_WGRAD_X6") && atoi(getenv("CV_WGRAD_X6")) == 0);
    return cv_sp_conv_wgrad_px_f32(d_x, x_ld, cin, d_dy, dy_ld, cout, d_nbr, K, n_out, d_dw, d_ws, ws_bytes,
                                   x6 ? 3 : 0, stream);
}

int cv_sp_col_sum_f32(const float* d_x, long long n, int c, int ld, float* d_out, void* stream) {
    CV_REQUIRE(d_x && d_out && n > 0 && c > 0 && ld >= c, CV_EINVAL, "bad col_sum arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    CV_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(float) * c, st));
    dim3 grid((unsigned)((c + 31) / 32), (unsigned)std::min<long long>(256, (n + 1023) / 1024));
    col_sum<<<grid, 256, 0, st>>>(d_x, n, c, ld, d_out);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

static unsigned col_sum_chunks_for(long long n) { return (unsigned)std::min<long long>(256, (n + 1023) / 1024); }

size_t cv_sp_col_sum_workspace_bytes(long long n, int c) {
    if (n <= 0 || c <= 0) return 0;
    return sizeof(float) * (size_t)col_sum_chunks_for(n) * (size_t)c;
}

int cv_sp_col_sum_det_f32(const float* d_x, long long n, int c, int ld, float* d_out, void* d_ws, size_t ws_bytes,
                          void* stream) {
    CV_REQUIRE(d_x && d_out && d_ws && n > 0 && c > 0 && ld >= c, CV_EINVAL, "bad col_sum arguments");
    CV_REQUIRE(ws_bytes >= cv_sp_col_sum_workspace_bytes(n, c), CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned chunks = col_sum_chunks_for(n);
    float* ws = static_cast<float*>(d_ws);
    col_sum_partial<<<dim3((unsigned)((c + 31) / 32), chunks), 256, 0, st>>>(d_x, n, c, ld, ws);
    CV_LAUNCH_CHECK();
    col_sum_chunks<<<(unsigned)((c + 255) / 256), 256, 0, st>>>(ws, (int)chunks, c, d_out);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_sp_affine_f32(const float* d_x, long long n, int c, int x_ld, const float* d_scale,
                     const float* d_shift, const float* d_residual, int res_ld, int relu, float* d_y, int y_ld,
                     void* stream) {
    CV_REQUIRE(d_x && d_y && n > 0 && c > 0 && x_ld >= c && y_ld >= c, CV_EINVAL, "bad affine arguments");
    CV_REQUIRE(!d_residual || res_ld >= c, CV_EINVAL, "bad residual stride");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c % 4 == 0 && x_ld % 4 == 0 && y_ld % 4 == 0 && (!d_residual || res_ld % 4 == 0) && aligned16(d_x) &&
        aligned16(d_y) && aligned16(d_residual) && aligned16(d_scale) && aligned16(d_shift))
        affine_rows4<<<(unsigned)std::min<long long>((n * (c / 4) + 255) / 256, 16384), 256, 0, st>>>(
            d_x, n, c, x_ld, d_scale, d_shift, d_residual, res_ld, relu, d_y, y_ld);
    else
        affine_rows<<<(unsigned)std::min<long long>((n * c + 255) / 256, 8192), 256, 0, st>>>(
            d_x, n, c, x_ld, d_scale, d_shift, d_residual, res_ld, relu, d_y, y_ld);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_sp_affine_hl_f32(const float* d_x, long long n, int c, int x_ld, const float* d_scale, const float* d_shift,
                        const float* d_residual, int res_ld, int relu, float* d_y, int y_ld, float* d_y_hl, int y_hl_ld,
                        uint32_t* d_relu_bits, int32_t* range_flag, void* stream) {
    CV_REQUIRE(d_x && d_y && d_y_hl && n > 0 && c > 0 && x_ld >= c && y_ld >= c && y_hl_ld >= c, CV_EINVAL, "bad affine arguments");
    CV_REQUIRE(!d_residual || res_ld >= c, CV_EINVAL, "bad residual stride");
    CV_REQUIRE(c % 32 == 0 && y_hl_ld % 32 == 0 && (reinterpret_cast<uintptr_t>(d_y_hl) & 127) == 0, CV_EINVAL,
               "hl-format output: channels and leading dimension %% 32 == 0, 128-byte aligned rows");
    CV_REQUIRE(x_ld % 4 == 0 && y_ld % 4 == 0 && (!d_residual || res_ld % 4 == 0) && aligned16(d_x) && aligned16(d_y) &&
                   aligned16(d_residual) && aligned16(d_scale) && aligned16(d_shift), CV_EINVAL,
               "16-byte aligned operands with leading dimensions %% 4 == 0");
    affine_rows4<<<(unsigned)std::min<long long>((n * (c / 4) + 255) / 256, 16384), 256, 0, static_cast<hipStream_t>(stream)>>>(
        d_x, n, c, x_ld, d_scale, d_shift, d_residual, res_ld, relu, d_y, y_ld, d_y_hl, y_hl_ld, range_flag, d_relu_bits);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_sp_pack_weights_stem_h2_f32(const float* d_w, int K, int cin, const float* d_col_scale, int scale_log2, void* d_wp,
                                   void* stream) {
    CV_REQUIRE(d_w && d_wp && K > 0 && K <= 128 && (cin == 3 || cin == 6), CV_EINVAL,
               "stem weights: K <= 128, Cin 3 or 6, 32 output channels");
    CV_REQUIRE(scale_log2 >= -100 && scale_log2 <= 100, CV_EINVAL, "bad scale exponent");
    pack_weights_stem_h2<<<64, 256, 0, static_cast<hipStream_t>(stream)>>>(d_w, K, cin, d_col_scale, ldexpf(1.f, scale_log2),
                                                                           static_cast<unsigned short*>(d_wp));
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_sp_to_hl_f32(const float* d_x, long long n, int c, int x_ld, float* d_y, int y_ld, int32_t* range_flag, void* stream) {
    CV_REQUIRE(d_x && d_y && n > 0 && c > 0 && c % 32 == 0 && x_ld >= c && y_ld >= c && y_ld % 32 == 0 &&
                   (reinterpret_cast<uintptr_t>(d_y) & 127) == 0, CV_EINVAL,
               "hl format: channels and leading dimension %% 32 == 0, 128-byte aligned rows");
    to_hl<<<(unsigned)std::min<long long>((n * (c / 4) + 255) / 256, 8192), 256, 0, static_cast<hipStream_t>(stream)>>>(
        d_x, n, c, x_ld, d_y, y_ld, range_flag);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_sp_from_hl_f32(const float* d_x, long long n, int c, int x_ld, float* d_y, int y_ld, void* stream) {
    CV_REQUIRE(d_x && d_y && n > 0 && c > 0 && c % 32 == 0 && x_ld >= c && y_ld >= c && x_ld % 32 == 0 &&
                   (reinterpret_cast<uintptr_t>(d_x) & 127) == 0, CV_EINVAL,
               "hl format: channels and leading dimension %% 32 == 0, 128-byte aligned rows");
    from_hl<<<(unsigned)std::min<long long>((n * (c / 4) + 255) / 256, 8192), 256, 0, static_cast<hipStream_t>(stream)>>>(
        d_x, n, c, x_ld, d_y, y_ld);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_sp_bn_fold_f32(const float* d_gamma, const float* d_beta, const float* d_mean, const float* d_var,
                      const float* d_bias, float eps, int c, float* d_scale, float* d_shift, void* stream) {
    CV_REQUIRE(d_gamma && d_beta && d_mean && d_var && d_scale && d_shift && c > 0, CV_EINVAL, "bad bn_fold arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    bn_fold<<<(c + 127) / 128, 128, 0, st>>>(d_gamma, d_beta, d_mean, d_var, d_bias, eps, c, d_scale, d_shift);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

// dst_i = src_i for up to COPY_BATCH small tensors in ONE launch, unless *flag != 0 (cv_sp_copy_unless_flag)
constexpr int COPY_BATCH = 96;
struct CopyJobs {
    const uint32_t* src[COPY_BATCH];
    uint32_t* dst[COPY_BATCH];
    int words[COPY_BATCH];
};
static __global__ __launch_bounds__(256) void copy_unless_flag(CopyJobs jobs, const int32_t* __restrict__ flag) {
    if (flag && __builtin_nontemporal_load(flag) != 0) return;
    const uint32_t* s = jobs.src[blockIdx.x];
    uint32_t* d = jobs.dst[blockIdx.x];
    for (int i = threadIdx.x; i < jobs.words[blockIdx.x]; i += 256) d[i] = s[i];
}

int cv_sp_copy_unless_flag(const void* const* h_src, void* const* h_dst, const long long* h_bytes, int n, const int32_t* flag,
                           void* stream) {
    CV_REQUIRE(n >= 0 && (n == 0 || (h_src && h_dst && h_bytes)), CV_EINVAL, "bad copy batch");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int base = 0; base < n; base += COPY_BATCH) {
        CopyJobs jobs;
        const int m = std::min(COPY_BATCH, n - base);
        for (int i = 0; i < m; ++i) {
            CV_REQUIRE(h_src[base + i] && h_dst[base + i] && h_bytes[base + i] >= 0 && h_bytes[base + i] % 4 == 0 &&
                           h_bytes[base + i] < (1ll << 33), CV_EINVAL, "copy %d: null pointer or a size that is not a multiple of 4 bytes", base + i);
            jobs.src[i] = static_cast<const uint32_t*>(h_src[base + i]);
            jobs.dst[i] = static_cast<uint32_t*>(h_dst[base + i]);
            jobs.words[i] = (int)(h_bytes[base + i] / 4);
        }
        copy_unless_flag<<<m, 256, 0, st>>>(jobs, flag);
        CV_LAUNCH_CHECK();
    }
    return CV_OK;
}

size_t cv_sp_bn_workspace_bytes(int c) { return c > 0 ? 256 + sizeof(double) * 2 * (size_t)BN_CHUNKS * c : 0; }

static int bn_chunks(long long n) { return (int)std::min<long long>(BN_CHUNKS, std::max<long long>(1, n / 256)); }

// Training-mode BatchNorm statistics of x[n][c]: d_mean, d_var (biased), running statistics updated in place
// (may be NULL), and the folded d_scale/d_shift for cv_sp_affine_f32 (y = x*scale + shift).
int cv_sp_bn_stats_f32(const float* d_x, long long n, int c, int ld, const float* d_gamma, const float* d_beta,
                       float eps, float momentum, float* d_running_mean, float* d_running_var, float* d_mean,
                       float* d_var, float* d_scale, float* d_shift, void* d_ws, size_t ws_bytes, void* stream) {
    CV_REQUIRE(d_x && d_gamma && d_beta && d_mean && d_var && d_scale && d_shift && d_ws, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n > 0 && c > 0 && ld >= c, CV_EINVAL, "bad bn sizes");
    CV_REQUIRE(ws_bytes >= cv_sp_bn_workspace_bytes(c), CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* partial = static_cast<double*>(d_ws);
    const int chunks = bn_chunks(n);
    dim3 grid((unsigned)((c + 31) / 32), (unsigned)chunks);
    if (c % 4 == 0 && c <= 1024 && ld % 4 == 0 && aligned16(d_x))
        bn_col_reduce4<0><<<chunks, 256, 0, st>>>(d_x, nullptr, nullptr, n, c, ld, nullptr, nullptr, eps, partial);
    else
        bn_col_reduce<0><<<grid, 256, 0, st>>>(d_x, nullptr, nullptr, n, c, ld, nullptr, nullptr, eps, partial);
    CV_LAUNCH_CHECK();
    bn_col_finish<0><<<(c + 15) / 16, 256, 0, st>>>(partial, chunks, n, c, d_mean, d_var, d_running_mean,
                                                     d_running_var, momentum, d_gamma, d_beta, eps, d_scale, d_shift);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

// Backward of training-mode BatchNorm (optionally with the ReLU that follows it: pass its output d_y, else NULL):
// d_dgamma, d_dbeta, d_dx, and optionally d_dres = the ReLU-masked incoming gradient (gradient of a residual
// that was added between the normalisation and the ReLU).
static int bn_backward_impl(const float* d_x, const float* d_dy, const float* d_y, long long n, int c, int ld,
                            const float* d_mean, const float* d_var, float eps, const float* d_gamma, float* d_dgamma,
                            float* d_dbeta, float* d_dx, float* d_dres, void* d_ws, size_t ws_bytes, float* d_dx_hl,
                            unsigned* d_slot, int32_t* range_flag, const unsigned* d_relu_bits, void* stream) {
    CV_REQUIRE(d_x && d_dy && d_mean && d_var && d_gamma && d_dgamma && d_dbeta && d_dx && d_ws, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n > 0 && c > 0 && ld >= c, CV_EINVAL, "bad bn sizes");
    CV_REQUIRE(ws_bytes >= cv_sp_bn_workspace_bytes(c), CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* partial = static_cast<double*>(d_ws);
    const int chunks = bn_chunks(n);
    dim3 grid((unsigned)((c + 31) / 32), (unsigned)chunks);
    const bool v4 = c % 4 == 0 && c <= 1024 && ld % 4 == 0 && aligned16(d_x) && aligned16(d_dy) && aligned16(d_y) &&
                    aligned16(d_dx) && aligned16(d_dres) && aligned16(d_mean) && aligned16(d_var);
    CV_REQUIRE(!d_relu_bits || (v4 && c % 32 == 0), CV_EINVAL, "ReLU bits: c %% 32 == 0 and 16-byte aligned operands");
    if (v4)
        bn_col_reduce4<1><<<chunks, 256, 0, st>>>(d_x, d_dy, d_y, n, c, ld, d_mean, d_var, eps, partial, d_relu_bits);
    else
        bn_col_reduce<1><<<grid, 256, 0, st>>>(d_x, d_dy, d_y, n, c, ld, d_mean, d_var, eps, partial);
    CV_LAUNCH_CHECK();
    bn_col_finish<1><<<(c + 15) / 16, 256, 0, st>>>(partial, chunks, n, c, d_dbeta, d_dgamma, nullptr, nullptr, 0.f,
                                                     nullptr, nullptr, eps, nullptr, nullptr);
    CV_LAUNCH_CHECK();
    CV_REQUIRE(!d_dx_hl || (v4 && d_slot && c % 32 == 0 && ld % 32 == 0 && (reinterpret_cast<uintptr_t>(d_dx_hl) & 127) == 0), CV_EINVAL,
               "hl-format gradient: channels and leading dimension %% 32 == 0, 128-byte aligned rows, a scale slot");
    if (v4)
        bn_backward_apply4<<<(unsigned)std::min<long long>((n * (c / 4) + 255) / 256, d_dx_hl ? BN_SLOT_BLOCKS : 16384), 256, 0, st>>>(
            d_x, d_dy, d_y, n, c, ld, d_mean, d_var, eps, d_gamma, d_dbeta, d_dgamma, d_dx, d_dres, d_dx_hl, d_slot, range_flag, d_relu_bits);
    else
        bn_backward_apply<<<(unsigned)std::min<long long>((n * c + 255) / 256, 8192), 256, 0, st>>>(
            d_x, d_dy, d_y, n, c, ld, d_mean, d_var, eps, d_gamma, d_dbeta, d_dgamma, d_dx, d_dres);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_sp_bn_backward_f32(const float* d_x, const float* d_dy, const float* d_y, long long n, int c, int ld,
                          const float* d_mean, const float* d_var, float eps, const float* d_gamma, float* d_dgamma,
                          float* d_dbeta, float* d_dx, float* d_dres, void* d_ws, size_t ws_bytes, void* stream) {
    return bn_backward_impl(d_x, d_dy, d_y, n, c, ld, d_mean, d_var, eps, d_gamma, d_dgamma, d_dbeta, d_dx, d_dres, d_ws, ws_bytes,
                            nullptr, nullptr, nullptr, nullptr, stream);
}

int cv_sp_bn_backward_hl_f32(const float* d_x, const float* d_dy, const float* d_y, long long n, int c, int ld,
                             const float* d_mean, const float* d_var, float eps, const float* d_gamma, float* d_dgamma,
                             float* d_dbeta, float* d_dx, float* d_dres, void* d_ws, size_t ws_bytes, float* d_dx_hl,
                             uint32_t* d_slot, int32_t* range_flag, const uint32_t* d_relu_bits, void* stream) {
    CV_REQUIRE((d_dx_hl != nullptr) == (d_slot != nullptr), CV_EINVAL, "the hl twin of dx and its scale slot come together");
    return bn_backward_impl(d_x, d_dy, d_y, n, c, ld, d_mean, d_var, eps, d_gamma, d_dgamma, d_dbeta, d_dx, d_dres, d_ws, ws_bytes,
                            d_dx_hl, d_slot, range_flag, d_relu_bits, stream);
}

int cv_head_joint_f32(const float* d_feats, long long n, int ld, int nclasses, int log_scale, float* d_xyz,
                      float* d_scale, float* d_prob, int32_t* d_class, void* stream) {
    CV_REQUIRE(d_feats && d_xyz && d_scale && d_prob && d_class, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n > 0 && nclasses > 0 && nclasses < 64 && ld >= 7 * nclasses + 1, CV_EINVAL, "bad head arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    head_joint<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(d_feats, n, ld, nclasses, log_scale, d_xyz,
                                                           d_scale, d_prob, d_class);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_head_separate_f32(const float* d_feats, long long n, int ld, int log_scale, float* d_xyz, float* d_scale,
                         float* d_prob, void* stream) {
    CV_REQUIRE(d_feats && d_xyz && d_scale && d_prob && n > 0 && ld >= 8, CV_EINVAL, "bad head arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    head_separate<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(d_feats, n, ld, log_scale, d_xyz, d_scale, d_prob);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

}  // extern "C"
