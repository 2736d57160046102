// Neighbour-window convolution for the fine levels (round 5): the 3x3x3 convolutions of a level whose rows are in spatial
// (Z-order) order, as a tiled GEMM whose gathered operand is fetched ONCE per tile.
//
// Supplies the same arithmetic as conv_hl / conv_hd (sparse_conv.hip) for MinkowskiConvolution(kernel_size=3) with the
// folded BatchNorm / residual / ReLU epilogue (utils/minkunet.py:122-180, utils/resnet.py:118-154 through ME BasicBlock):
//   out[u] = sum_j W_j^T x[nbr[u][j]]
//
// Why: with the rows mask-sorted (conv_hl / conv_hd) a ts1 96 -> 96 layer gathers 6.2 input rows per output row (~200 MB
// of 128-byte line gathers from the L2s / Infinity Cache), writes three mask-group partial tiles per output row (92 MB)
// and reads them back in a finish launch (92 MB + 31 MB): the matrix pipe is 15 % busy (profiles/r4/conv_pmc_hd.txt).
// A tile of 256 CONSECUTIVE rows of the Z-order touches only ~1.33 x 256 distinct input rows (1.5 x at tensor stride 2):
// its *window*.  conv_win lands the window in LDS once per 32-channel chunk (coalesced LDS-DMA, 41 MB per layer instead
// of 200), multiplies all 27 offsets out of LDS with the sums kept in the accumulators - no mask groups, no partial
// tiles, no finish launch - and pays for it on the matrix cores: a 32-row block of spatial neighbours needs 94 % of
// the 27 offsets (24 % with the mask orders), i.e. ~3 x the MFMAs.  The pipe had the room.
//
// Plan side (build_windows, once per level and scene, shared by every layer of the level): per tile the sorted list of
// distinct input rows (win_rows[tile][WIN_CAP]) and the kernel map rewritten to 16-bit window slots
// (lmap[tile][256][28]: 0xFFFF = no neighbour, 0xFFFE = neighbour outside the first WIN_CAP window rows - those rare
// pairs are gathered from global memory by the slow path, so any input is computed correctly).
//
// Kernel (conv_win<NB>): 8 waves x 32 rows, Cout = NB x 32 columns (all of them: nothing is gathered twice).
//   LDS: two window buffers [WIN_CAP + 1 rows][128 B] (chunk c multiplies while chunk c + 1 lands; the extra row is
//   zeros: lanes without a neighbour read it), a ring of three weight tiles (the tile of unit s + 2 is requested behind the
//   barrier of unit s), the epilogue tile aliased over the first window buffer.  Units run chunk-major: for each
//   32-channel chunk the 27 offsets, unrolled (the lane's 27 window slots live in 14 registers).  Operands by LDS-DMA
//   with counted vmcnt waits and asm fragment reads exactly as conv_hd; window pieces XOR-swizzled by the window slot on
//   the source side so that the b128 fragment reads of 16 arbitrary rows spread over the banks.
//   A BasicBlock's 1x1 downsample branch (second source on the output rows) runs as nch2 extra units with the A
//   fragments straight from global memory.
#include "cv_common.h"

#include <atomic>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "sparse_conv_common.h"

using namespace cvsc;

namespace {

constexpr int WIN_T = 256;                 // rows per tile
constexpr int WIN_CAP = CV_WIN_CAP;        // window rows held in LDS (multiple of 64: 8 rows per LDS-DMA instruction x 8 waves)
// a window slot as the kernel map stores it: byte offset of the slot's 64-byte row in a window buffer, with the row's
// read swizzle (slot >> 2) & 3 in bits 4-5: the fragment address of piece p is (entry ^ (p << 4)) + buffer.  The two rows
// behind the window are zeros: WIN_NONE (no neighbour) and WIN_OUT (neighbour beyond the window: added by the extra units).
constexpr unsigned WIN_NONE = (unsigned)CV_WIN_CAP * 64u, WIN_OUT = WIN_NONE + 64u;
__host__ __device__ constexpr unsigned win_entry(unsigned slot) { return slot * 64u + (((slot >> 2) & 3u) << 4); }
static_assert(WIN_CAP % 128 == 0 && (WIN_CAP + 2) * 64 <= 0xFFFF, "window capacity");

// ------------------------------------------------------------------ plan: windows of a 27-offset kernel map
struct WinJobsDev {
    int n;
    int tile_begin[CV_MAX_WIN_JOBS + 1];
    const int* nbr[CV_MAX_WIN_JOBS];
    long long rows[CV_MAX_WIN_JOBS];
    int* win[CV_MAX_WIN_JOBS];
};

// One workgroup per tile, thread = row.  The distinct neighbour rows of the tile are found with a bitmap over the row
// range the tile touches (LDS), ranked by prefix popcount: window slot = rank in ascending row order (deterministic).
__global__ __launch_bounds__(256) void build_windows(const WinJobsDev jobs, int cap_words) {
    extern __shared__ unsigned win_lds[];
    __shared__ int red[16];
    int job = 0;
    while (job + 1 < jobs.n && (int)blockIdx.x >= jobs.tile_begin[job + 1]) ++job;
    const int tile = (int)blockIdx.x - jobs.tile_begin[job];
    const long long n = jobs.rows[job];
    const int* __restrict__ nbr = jobs.nbr[job];
    const long long ntiles = (n + WIN_T - 1) / WIN_T;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int* __restrict__ wrows = jobs.win[job] + (long long)tile * WIN_CAP;
    unsigned short* __restrict__ lm_out = reinterpret_cast<unsigned short*>(jobs.win[job] + ntiles * WIN_CAP) +
                                          (long long)tile * (27 * WIN_T) + t;            // [tile][27][256]
    unsigned* bits = win_lds;
    unsigned short* wpre = reinterpret_cast<unsigned short*>(win_lds + cap_words);
    const long long row = (long long)tile * WIN_T + t;
    int e[27];
    int mn = INT_MAX, mx = -1;
#pragma unroll
    for (int j = 0; j < 27; ++j) {
        e[j] = row < n ? nbr[row * 27 + j] : -1;
        if (e[j] >= 0) { mn = min(mn, e[j]); mx = max(mx, e[j]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_xor(mn, off)); mx = max(mx, __shfl_xor(mx, off)); }
    if (lane == 0) { red[wave] = mn; red[4 + wave] = mx; }
    __syncthreads();
    mn = min(min(red[0], red[1]), min(red[2], red[3]));
    mx = max(max(red[4], red[5]), max(red[6], red[7]));
    const int lo_w = mx >= 0 ? mn >> 5 : 0;
    const int nwr = mx >= 0 ? (mx >> 5) - lo_w + 1 : 0;           // bitmap words of the touched row range (<= cap_words)
    for (int w = t; w < nwr; w += 256) bits[w] = 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 27; ++j)
        if (e[j] >= 0) atomicOr(&bits[(e[j] >> 5) - lo_w], 1u << (e[j] & 31));
    __syncthreads();
    // rank of every set bit: words [w0, w1) of this thread, exclusive scan of the per-thread counts over the workgroup
    const int per = (nwr + 255) / 256, w0 = min(nwr, t * per), w1 = min(nwr, w0 + per);
    int cnt = 0;
    for (int w = w0; w < w1; ++w) cnt += __popc(bits[w]);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) red[8 + wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) base += red[8 + w]; total += red[8 + w]; }
    int run = base + incl - cnt;
    for (int w = w0; w < w1; ++w) {
        wpre[w] = (unsigned short)min(run, 0xFFFF);
        unsigned b = bits[w];
        while (b) {
            const int bit = __ffs(b) - 1;
            b &= b - 1;
            if (run < WIN_CAP) wrows[run] = ((lo_w + w) << 5) + bit;
            ++run;
        }
    }
    for (int r = total + t; r < WIN_CAP; r += 256) wrows[r] = -1;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 27; ++j) {
        unsigned v = WIN_NONE;
        if (e[j] >= 0) {
            const int w = (e[j] >> 5) - lo_w;
            const unsigned rank = (unsigned)wpre[w] + (unsigned)__popc(bits[w] & ((1u << (e[j] & 31)) - 1u));
            v = rank < (unsigned)WIN_CAP ? win_entry(rank) : WIN_OUT;
        }
        lm_out[j * WIN_T] = (unsigned short)v;
    }
}

// ------------------------------------------------------------------ the convolution
#ifndef CV_WIN_ABL
#define CV_WIN_ABL 0      // timing ablations (wrong results): 1 no window DMA, 2 no MFMA, 4 no weight DMA, 8 no fragment reads, 16 no step barriers, 32 no liveness tests, 64 no counted waits
#endif

// MFMA operand fragments of one unit: per offset g of the unit the lane's row (high piece, low piece), per (g, plane, nb)
// the weight pieces.  One asm statement reads them all and waits (see hd_read_frags for why it is asm and a plain function).
template <int NB, int G>
struct WinFrags {
    u32x4v a[2 * G];            // [g][h | l]
    u32x4v b[2 * G * NB];       // [(g * 2 + plane) * NB + nb]
};
// request every fragment of a unit (no wait: the MFMAs of the previous unit run while they travel); OFF = the unit's ring
// tile relative to the base register ...
template <int OFF>
__device__ __forceinline__ void win_read_issue(WinFrags<3, 1>& f, const unsigned (&aa)[2], unsigned ab) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\t"
                 "ds_read_b128 %2, %10 offset:%11\n\tds_read_b128 %3, %10 offset:%12\n\tds_read_b128 %4, %10 offset:%13\n\t"
                 "ds_read_b128 %5, %10 offset:%14\n\tds_read_b128 %6, %10 offset:%15\n\tds_read_b128 %7, %10 offset:%16"
                 : "=&v"(f.a[0]), "=&v"(f.a[1]), "=&v"(f.b[0]), "=&v"(f.b[1]), "=&v"(f.b[2]), "=&v"(f.b[3]), "=&v"(f.b[4]), "=&v"(f.b[5])
                 : "v"(aa[0]), "v"(aa[1]), "v"(ab), "i"(OFF), "i"(OFF + 1024), "i"(OFF + 2048), "i"(OFF + 3072), "i"(OFF + 4096),
                   "i"(OFF + 5120) : "memory");
}
template <int OFF>
__device__ __forceinline__ void win_read_issue(WinFrags<2, 1>& f, const unsigned (&aa)[2], unsigned ab) {
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\t"
                 "ds_read_b128 %2, %8 offset:%9\n\tds_read_b128 %3, %8 offset:%10\n\tds_read_b128 %4, %8 offset:%11\n\t"
                 "ds_read_b128 %5, %8 offset:%12"
                 : "=&v"(f.a[0]), "=&v"(f.a[1]), "=&v"(f.b[0]), "=&v"(f.b[1]), "=&v"(f.b[2]), "=&v"(f.b[3])
                 : "v"(aa[0]), "v"(aa[1]), "v"(ab), "i"(OFF), "i"(OFF + 1024), "i"(OFF + 2048), "i"(OFF + 3072) : "memory");
}
template <int OFF>
__device__ __forceinline__ void win_read_issue(WinFrags<1, 3>& f, const unsigned (&aa)[6], unsigned ab) {
    asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %13\n\tds_read_b128 %2, %14\n\tds_read_b128 %3, %15\n\t"
                 "ds_read_b128 %4, %16\n\tds_read_b128 %5, %17\n\t"
                 "ds_read_b128 %6, %18 offset:%19\n\tds_read_b128 %7, %18 offset:%20\n\tds_read_b128 %8, %18 offset:%21\n\t"
                 "ds_read_b128 %9, %18 offset:%22\n\tds_read_b128 %10, %18 offset:%23\n\tds_read_b128 %11, %18 offset:%24"
                 : "=&v"(f.a[0]), "=&v"(f.a[1]), "=&v"(f.a[2]), "=&v"(f.a[3]), "=&v"(f.a[4]), "=&v"(f.a[5]),
                   "=&v"(f.b[0]), "=&v"(f.b[1]), "=&v"(f.b[2]), "=&v"(f.b[3]), "=&v"(f.b[4]), "=&v"(f.b[5])
                 : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(aa[4]), "v"(aa[5]), "v"(ab), "i"(OFF), "i"(OFF + 1024),
                   "i"(OFF + 2048), "i"(OFF + 3072), "i"(OFF + 4096), "i"(OFF + 5120) : "memory");
}
// ... and the wait that hands them (and the prefetched window slots) over: the operands tie the registers to the wait, nothing
// that uses them may be scheduled in front of it
__device__ __forceinline__ void win_read_wait(WinFrags<3, 1>& f, unsigned (&sl)[1]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3]), "+v"(f.b[4]), "+v"(f.b[5]), "+v"(sl[0])
                 :: "memory");
}
__device__ __forceinline__ void win_read_wait(WinFrags<2, 1>& f, unsigned (&sl)[1]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3]), "+v"(sl[0]) :: "memory");
}
__device__ __forceinline__ void win_read_wait(WinFrags<1, 3>& f, unsigned (&sl)[3]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.a[4]), "+v"(f.a[5]), "+v"(f.b[0]), "+v"(f.b[1]),
                   "+v"(f.b[2]), "+v"(f.b[3]), "+v"(f.b[4]), "+v"(f.b[5]), "+v"(sl[0]), "+v"(sl[1]), "+v"(sl[2])
                 :: "memory");
}
// window slots of a unit's G offsets (16-bit LDS reads at an immediate offset from the lane's slot column, no wait)
template <int G, int OFF>
__device__ __forceinline__ void win_slots_issue(unsigned (&sl)[G], unsigned addr) {
    if constexpr (G == 1) {
        asm volatile("ds_read_u16 %0, %1 offset:%2" : "=&v"(sl[0]) : "v"(addr), "i"(OFF) : "memory");
    } else {
        static_assert(G == 3, "one or three offsets per unit");
        asm volatile("ds_read_u16 %0, %3 offset:%4\n\tds_read_u16 %1, %3 offset:%5\n\tds_read_u16 %2, %3 offset:%6"
                     : "=&v"(sl[0]), "=&v"(sl[1]), "=&v"(sl[2]) : "v"(addr), "i"(OFF), "i"(OFF + 512), "i"(OFF + 1024) : "memory");
    }
}
// the weight pieces of a one-offset unit alone (extra units: the A fragments come from global memory)
template <int NB>
__device__ __forceinline__ void win_read_b(u32x4v (&b)[2 * NB], unsigned ab) {
    if constexpr (NB == 1) {
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(b[0]), "=&v"(b[1]) : "v"(ab) : "memory");
    } else if constexpr (NB == 2) {
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                     "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]) : "v"(ab) : "memory");
    } else {
        asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %6 offset:2048\n\t"
                     "ds_read_b128 %3, %6 offset:3072\n\tds_read_b128 %4, %6 offset:4096\n\tds_read_b128 %5, %6 offset:5120\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]) : "v"(ab) : "memory");
    }
}

// Units.  A unit = G kernel offsets x one 16-channel k-step x NB column blocks = 3 NB G MFMAs per wave and NB G 2 KB of
// weights (6 KB for the two shapes the network has: <3, 1> 96 columns, <1, 3> 32 columns).  They run k-step-major: for each
// HALF chunk h = 2 c + ks (16 channels: 32 B of high and 32 B of low pieces = a 64-byte window row) the 27 / G units of
// its offsets; the window of half chunk h + 1 lands in the other buffer meanwhile.
//
// What the first versions measured (profiles/r5/win_v*_*.txt): a barrier per unit with "read fragments -> wait -> multiply":
// 2100 cycles per 32-channel unit where the MFMAs need 1152; ping-pong between the two waves of a SIMD, a software pipeline
// inside every wave, steps of three units behind one barrier: 1340 cycles per 16-channel unit - of which 480 remain with
// the DMA, the fragment reads AND the MFMAs compiled out.  The run-time bookkeeping of a unit (ring stage, slab pointer, slot
// row, liveness, ~60 scalar instructions and a dozen branches through the CU's one scalar unit) was a third of the kernel and
// added to the rest.  Hence this version:
//  * the 54 / G units of a 32-channel chunk are unrolled: unit, k-step, window buffer, ring tile and slot row are
//    immediates of the instructions; what remains at run time is the chunk loop, one slab pointer and the live-unit mask;
//  * steps of THREE units (27 MFMAs per wave) behind one barrier and one counted wait; their weight tiles are requested
//    three steps ahead into a ring of R = 12 unit tiles (a tile has two steps = six units of MFMA time to land; the ring
//    position of a chunk's first unit alternates between tile 0 and tile 6: two base registers per chunk);
//  * inside a step the wave's own software pipeline: request the fragments of unit s + 1 (second register set) and the
//    window slots of unit s + 2, multiply unit s, wait.  The barrier of step T therefore guarantees the tiles of step T + 1;
//  * the plan stores a slot as its row's byte offset with the read swizzle folded in (win_entry): a fragment address is one
//    xor-add.
template <int NB, int G>
__global__ __launch_bounds__(512, 2) void conv_win(ConvArgs a, int xcd_per) {
    static_assert((NB == 3 && G == 1) || (NB == 2 && G == 1) || (NB == 1 && G == 3), "unit shapes");
    constexpr int NW = 8, R = 12, BU = 3;                           // ring tiles, units per step (barrier)
    constexpr int UPH = 27 / G;                                     // units per half chunk
    constexpr int WBUF = (WIN_CAP + 2) * 64;                        // one window buffer: WIN_CAP rows of 64 B + two rows of zeros
    constexpr int B_UNIT = NB * G * 2048;                           // weight tile of a unit: [g][plane][col][32 B]
    constexpr int B_INSTR = B_UNIT / 1024;                          // <= 6: wave t requests KB t of the tile
    constexpr int WIN_PER_WAVE = WIN_CAP / 16 / NW;                 // LDS-DMA instructions (16 rows each) per wave and half chunk
    constexpr int OFF_B = 2 * WBUF, OFF_LM = OFF_B + R * B_UNIT, OFF_ROWS = OFF_LM + 27 * WIN_T * 2,
                  OFF_OM = OFF_ROWS + WIN_T * 4, LDS_TOTAL = OFF_OM + NW * 4;
    constexpr int EP_BYTES = NW * 32 * EP_LD * 4;
    static_assert(EP_BYTES <= 2 * WBUF, "the epilogue tile aliases the window buffers");
    static_assert(LDS_TOTAL <= 160 * 1024 && B_INSTR <= NW && UPH % BU == 0 && R == 4 * BU, "LDS budget / ring");
    static_assert(WIN_PER_WAVE * NW * 16 == WIN_CAP, "window instructions are dealt evenly to the waves");
    // ONE __shared__ object (a second one makes hipcc drain vmcnt in front of the LDS reads of an LDS-DMA pipeline)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_TOTAL];
    int* const rows_s = reinterpret_cast<int*>(lds + OFF_ROWS);
    const unsigned short* const lm_s = reinterpret_cast<const unsigned short*>(lds + OFF_LM);      // [27][256] window slots
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const long long ntiles = (a.n_out + WIN_T - 1) / WIN_T;
    // XCD-aware numbering: the workgroups an XCD receives (blockIdx.x % 8) walk a contiguous range of tiles, so the windows of
    // neighbouring tiles - which overlap - meet in ONE L2
    const long long tile = xcd_per > 0 ? (long long)(blockIdx.x & 7) * xcd_per + (blockIdx.x >> 3) : (long long)blockIdx.x;
    if (tile >= ntiles) return;
    const long long row0 = tile * WIN_T;
    const int nch = a.cin / KC, nch2 = a.in2 ? a.cin2 / KC : 0;
    const int NH = 2 * nch, U = NH * UPH;                           // half chunks, main units

    // ---- tile set-up: the tile's window slots -> LDS, this wave's share of the window rows
    {
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(a.win + ntiles * WIN_CAP) +
                                                          tile * (27 * WIN_T));
        for (int f = tid; f < 27 * WIN_T * 2 / 16; f += NW * 64) reinterpret_cast<uint4*>(lds + OFF_LM)[f] = src[f];
    }
    const unsigned in_row_bytes = (unsigned)a.in_ld * 4u, in2_row_bytes = (unsigned)a.in2_ld * 4u;
    const unsigned char* const in_b = reinterpret_cast<const unsigned char*>(a.in);
    const unsigned char* const in2_b = reinterpret_cast<const unsigned char*>(a.in2);
    // window instruction q = wave + 8 i covers window rows 16 q + (lane >> 2); LDS slot lane & 3 of row w receives the row's
    // piece (lane & 3) ^ ((w >> 2) & 3) of the half chunk's four [h lo-half, h hi-half, l lo-half, l hi-half] - and
    // (w >> 2) & 3 = (4 q + (lane >> 4)) & 3 = (lane >> 4) & 3
    const unsigned w_sp = (unsigned)((lane & 3) ^ ((lane >> 4) & 3));
    const unsigned w_piece = (w_sp >> 1) * 64u + (w_sp & 1u) * 16u;            // byte offset inside the 128-byte chunk, k-step 0
    unsigned woff[WIN_PER_WAVE];
    {
        const int* wr = a.win + tile * WIN_CAP;
#pragma unroll
        for (int i = 0; i < WIN_PER_WAVE; ++i) {
            const int r = wr[16 * (wave + NW * i) + (lane >> 2)];
            woff[i] = r >= 0 ? (unsigned)r * in_row_bytes : 0xFFFFFFFFu;
        }
    }
    if (tid < WIN_T) rows_s[tid] = row0 + tid < a.n_out ? (int)(row0 + tid) : -1;
    if (tid < 16) *reinterpret_cast<uint4*>(lds + (tid >> 3) * WBUF + WIN_CAP * 64 + (tid & 7) * 16) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const int my_t = wave * 32 + l31;
    unsigned jmask = 0u, omask = 0u;                                // offsets this wave has a neighbour at / an outside-window pair at
#pragma unroll
    for (int j = 0; j < 27; ++j) {
        const unsigned v = lm_s[j * WIN_T + my_t];
        if (__any(v != WIN_NONE)) jmask |= 1u << j;          // (an outside-window pair makes the offset live too: harmless)
        if (__any(v == WIN_OUT)) omask |= 1u << j;
    }
    if (lane == 0) reinterpret_cast<unsigned*>(lds + OFF_OM)[wave] = omask;
    __syncthreads();
    unsigned omask_wg = 0u;                                         // offsets with an outside-window pair in ANY wave of the tile
#pragma unroll
    for (int w = 0; w < NW; ++w) omask_wg |= reinterpret_cast<const unsigned*>(lds + OFF_OM)[w];
    omask_wg = __builtin_amdgcn_readfirstlane(omask_wg);
    jmask = __builtin_amdgcn_readfirstlane(jmask);
    unsigned umask = 0u;                                            // units with a live offset
#pragma unroll
    for (int u = 0; u < UPH; ++u)
        if ((jmask >> (u * G)) & ((1u << G) - 1u)) umask |= 1u << u;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    // ---- weight tile requests: wave t < B_INSTR requests KB t = (g, plane, nb) of every unit's tile: lane f -> column f >> 1,
    // LDS slot f & 1 of the column's 32 bytes, which receives the 16-byte piece (f & 1) ^ ((col >> 4) & 1) (conflict-free reads)
    const int bt_g = wave / (2 * NB), bt_pl = (wave / NB) & 1, bt_nb = wave % NB;
    const unsigned b_src = (unsigned)((bt_pl * a.cout + bt_nb * 32 + (lane >> 1)) * 32 + (((lane & 1) ^ ((lane >> 5) & 1)) << 3));
    const unsigned slab_words = 2u * (unsigned)a.cout * 32u;
    const bool has_w = wave < B_INSTR;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)lds;
    const unsigned b_rd = lds0 + (unsigned)(OFF_B + l31 * 32) + (unsigned)((half ^ ((l31 >> 4) & 1)) << 4);
    const long long my_row = row0 + my_t;

    constexpr int UPC = 2 * UPH;                                     // units per 32-channel chunk: k-step 0, then k-step 1
    static_assert(UPC % (2 * BU) == 0 && (UPC % R == 0 || UPC % R == R / 2), "chunk = whole steps; ring position alternates 0 / 6");
    WinFrags<NB, G> F[2];                                           // fragments of the unit that multiplies / of the next one
    unsigned SL[G];                                                 // window-slot entries of the lane's row for the next unit's offsets
    const unsigned lm_rd = lds0 + (unsigned)(OFF_LM + my_t * 2);
    const unsigned k_h = (unsigned)half << 4, k_l = (2u + (unsigned)half) << 4;          // piece selectors of the lane: high, low
    unsigned b_lo = b_rd, b_hi = b_rd + (unsigned)(R / 2 * B_UNIT);  // ring tiles 0-5 / 6-11 of the CHUNK's numbering (see below)
    // fragments of unit w (k-step KS, ring tile TL of the chunk's numbering) whose slot entries are in SL -> F[PAR]
    auto read_issue = [&](auto PAR, auto KS, auto TL) {
        constexpr int par = decltype(PAR)::value, ks = decltype(KS)::value, tl = decltype(TL)::value;
        unsigned aa[2 * G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            aa[2 * g] = (SL[g] ^ k_h) + (lds0 + (unsigned)(ks * WBUF));
            aa[2 * g + 1] = (SL[g] ^ k_l) + (lds0 + (unsigned)(ks * WBUF));
        }
        if (!(CV_WIN_ABL & 8)) win_read_issue<(tl % (R / 2)) * B_UNIT>(F[par], aa, tl < R / 2 ? b_lo : b_hi);
    };
    auto mfma_frags = [&](const u32x4v& Ah, const u32x4v& Al, const u32x4v* Bh, const u32x4v* Bl) {
        const f16x8 a0 = __builtin_bit_cast(f16x8, Ah), a1 = __builtin_bit_cast(f16x8, Al);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f16x8 b0 = __builtin_bit_cast(f16x8, Bh[nb]), b1 = __builtin_bit_cast(f16x8, Bl[nb]);
            if (!(CV_WIN_ABL & 2)) {
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[nb], 0, 0, 0);
            }
        }
    };
    auto mfma_unit = [&](auto PAR) {
        constexpr int par = decltype(PAR)::value;
#pragma unroll
        for (int g = 0; g < G; ++g)
            mfma_frags(F[par].a[2 * g], F[par].a[2 * g + 1], &F[par].b[(2 * g) * NB], &F[par].b[(2 * g + 1) * NB]);
    };

    // Extra units behind the main ones (ONE code instance, a run-time loop): one offset x one k-step each.  e < 2 nch2: the
    // second source (A = the output row itself, chunk e / 2); then, for every offset at which ANY wave of the tile has a
    // pair outside its window, the offset's 2 nch half chunks with A = the outside neighbours only.  Their A fragments
    // come straight from global memory, the weight tiles (KB (plane, nb) of a G = 1 tile) through the ring.
    const int n_fix = __popc(omask_wg);
    const int E = 2 * nch2 + n_fix * NH;
    const bool has_we = wave < 2 * NB;
    const unsigned be_src = (unsigned)(((wave / NB) * a.cout + (wave % NB) * 32 + (lane >> 1)) * 32 + (((lane & 1) ^ ((lane >> 5) & 1)) << 3));
    auto extra_decode = [&](int e, int& j, int& h) -> bool {        // true: second-source unit
        if (e < 2 * nch2) { j = 0; h = e; return true; }
        const int f = e - 2 * nch2, idx = f / NH;
        h = f - idx * NH;
        unsigned m = omask_wg;
        for (int q = 0; q < idx; ++q) m &= m - 1u;
        j = __ffs(m) - 1;
        return false;
    };
    auto issue_extra = [&](int e) {
        int j, h;
        const bool second = extra_decode(e, j, h);
        const unsigned short* slab = second ? a.wp6_2 + (size_t)(unsigned)(h >> 1) * slab_words
                                            : a.wp6 + (size_t)(unsigned)(j * nch + (h >> 1)) * slab_words;
        if (has_we && !(CV_WIN_ABL & 4)) lds_dma16(slab + be_src + (h & 1) * 16, lds + OFF_B + ((U + e) % R) * B_UNIT + wave * 1024);
    };
    // (a wave that requests main tiles but no extra tiles - G = 3 - or the reverse would break the counted waits: the extra
    // loop below drains the queue instead of counting)

    // ---- weight tile requests.  q_ptr: this wave's source of the next main unit to request (uniform part; b_src is the
    // lane's); it moves by one of three constants per unit.  q_lo / q_hi: LDS destination of ring tiles 0-5 / 6-11 of the
    // chunk's numbering (with the wave's KB).
    const ptrdiff_t w_step = (ptrdiff_t)(G * nch) * (ptrdiff_t)slab_words;      // 16-bit words: a unit's slab -> the next unit's
    const ptrdiff_t w_to_ks1 = 16 - (UPH - 1) * w_step;             // last unit of k-step 0 -> first of k-step 1
    const ptrdiff_t w_to_next_c = (ptrdiff_t)slab_words - 16 - (UPH - 1) * w_step;        // last unit of a chunk -> next chunk
    const unsigned short* q_ptr = a.wp6 + (size_t)(bt_g * nch) * slab_words;
    unsigned q_lo = (unsigned)(OFF_B + wave * 1024), q_hi = q_lo + (unsigned)(R / 2 * B_UNIT);
    // request unit x of the current chunk's numbering (x >= UPC: a unit of the next chunk, when there is one)
    auto issue_unit = [&](auto X, bool exists) {
        constexpr int x = decltype(X)::value, tl = x % R;
        if (exists) {
            if (has_w && !(CV_WIN_ABL & 4)) lds_dma16(q_ptr + b_src, lds + ((tl < R / 2 ? q_lo : q_hi) + (unsigned)((tl % (R / 2)) * B_UNIT)));
            q_ptr += (x + 1) % UPC == 0 ? w_to_next_c : (x + 1) % UPC == UPH ? w_to_ks1 : w_step;
        }
    };
    // the window of the half chunk that follows k-step KS of chunk c, this wave's WIN_PER_WAVE instructions.  ALWAYS that many
    // (rows beyond the window - and a whole half chunk beyond the last - fetch the line of zeros): the counted waits below
    // know the queue at compile time
    auto issue_window = [&](auto KS, int c) {
        constexpr int ks = decltype(KS)::value;                    // the window goes to the buffer the CURRENT k-step does not use
        const bool dummy = (CV_WIN_ABL & 1) || (ks == 1 && c + 1 >= nch);
        const size_t off = ks == 0 ? (size_t)(c * 128 + 32) : (size_t)((c + 1) * 128);
#pragma unroll
        for (int i = 0; i < WIN_PER_WAVE; ++i) {
            const unsigned char* g = (!dummy && woff[i] != 0xFFFFFFFFu) ? in_b + ((size_t)woff[i] + off + w_piece) : g_zero_chunk + w_piece;
            lds_dma16(g, lds + (1 - ks) * WBUF + (wave + NW * i) * 1024);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    // ---- prologue: the window of half chunk 0, the weight tiles of the first three steps
    {
        const bool dummy = CV_WIN_ABL & 1;
#pragma unroll
        for (int i = 0; i < WIN_PER_WAVE; ++i) {
            const unsigned char* g = (!dummy && woff[i] != 0xFFFFFFFFu) ? in_b + ((size_t)woff[i] + w_piece) : g_zero_chunk + w_piece;
            lds_dma16(g, lds + (wave + NW * i) * 1024);
        }
    }
    static_for<3 * BU>([&](auto X) { issue_unit(X, true); });       // (a chunk has at least 18 units)
    wait_vmcnt_le<0>();
    __builtin_amdgcn_s_barrier();                                   // the first three steps' tiles and the window are there
    win_slots_issue<G, 0>(SL, lm_rd);
    win_read_wait(F[1], SL);                                        // (the register set is a dummy here: only the slots travel)
    read_issue(I0{}, I0{}, I0{});
    win_slots_issue<G, G * WIN_T * 2>(SL, lm_rd);
    win_read_wait(F[0], SL);                                        // fragments of unit 0 in F[0], slot entries of unit 1 in SL
    // Step T = units 3 T ... 3 T + 2 of chunk c (unit w: k-step w / UPH, unit u = w % UPH of its half chunk):
    // [wait: everything this wave requested up to step T - 2 has landed - the tiles of step T + 1 among it; what it requested at
    //  step T - 1 (three tiles, and the window instructions when that was the first step of a half chunk) stays in flight]
    // -> barrier (everybody's have landed; everybody is past step T - 1) -> requests: at the first step of a half chunk the
    // window of the next one (its buffer was last read in the half chunk before), the tiles of step T + 3 (the ring slots of
    // step T - 1) -> three times: request the fragments of the next unit and the slot entries of the one behind it, multiply,
    // wait.  Ring tiles are numbered from the chunk's first unit; UPC % 12 == 6: the chunk's tile 0 is ring tile 0 or 6.
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        const bool more = c + 1 < nch;
        static_for<UPC>([&](auto W) {
            constexpr int w = decltype(W)::value, ks = w / UPH, u = w % UPH;
            constexpr int wn = (w + 1) % UPC, ksn = wn / UPH, un = wn % UPH;              // unit s + 1
            constexpr int un2 = ((w + 2) % UPC) % UPH;                                   // unit s + 2: its slot entries only
            typedef std::integral_constant<int, w & 1> PAR;
            typedef std::integral_constant<int, (w + 1) & 1> PARN;
            if constexpr (w % BU == 0) {
                constexpr int T = w / BU;
                // what step T - 1 requested: tiles unless it was beyond the last chunk's units; the window at T - 1 == 0 or UPH / BU
                constexpr bool prev_win = T - 1 == 0 || T - 1 == UPH / BU;
                constexpr bool prev_next_chunk = T >= 1 && (T - 1) * BU + 3 * BU >= UPC;  // step T - 1 requested units of chunk c + 1
                const bool prev_tiles = T == 0 ? true : (prev_next_chunk ? more : true);    // (T == 0: the previous chunk's last step requested this chunk's)
                if (!(CV_WIN_ABL & 64)) {
                    if (!prev_tiles) wait_vmcnt_le<prev_win ? WIN_PER_WAVE : 0>();
                    else if (has_w) wait_vmcnt_le<BU + (prev_win ? WIN_PER_WAVE : 0)>();
                    else wait_vmcnt_le<prev_win ? WIN_PER_WAVE : 0>();
                }
                if (!(CV_WIN_ABL & 16)) __builtin_amdgcn_s_barrier();
                if constexpr (u == 0) issue_window(std::integral_constant<int, ks>{}, c);
                static_for<BU>([&](auto I) {
                    constexpr int x = w + 3 * BU + decltype(I)::value;
                    issue_unit(std::integral_constant<int, x>{}, x < UPC ? true : more);
                });
            }
            const bool live_next = (CV_WIN_ABL & 32) ? (wn != 0 || more) : wn == 0 ? (more && (umask & 1u)) : ((umask >> un) & 1u);
            if (live_next) read_issue(PARN{}, std::integral_constant<int, ksn>{}, std::integral_constant<int, (wn == 0 ? UPC : wn) % R>{});
            win_slots_issue<G, un2 * G * WIN_T * 2>(SL, lm_rd);
            if ((CV_WIN_ABL & 32) || ((umask >> u) & 1u)) mfma_unit(PAR{});
            win_read_wait(F[(w + 1) & 1], SL);
        });
        if (UPC % R != 0) {                                         // the next chunk's tile 0 is half a ring further
            const unsigned t0 = b_lo; b_lo = b_hi; b_hi = t0;
            const unsigned t1 = q_lo; q_lo = q_hi; q_hi = t1;
        }
    }
    // extra units: the main units are consumed (every ring slot is free behind a barrier): up to PE tiles ahead
    constexpr int PE = 6;
    int q_e = 0;
    if (E > 0) {
        wait_vmcnt_le<0>();
        __builtin_amdgcn_s_barrier();
        for (; q_e < E && q_e < PE; ++q_e) issue_extra(q_e);
    }
#pragma unroll 1
    for (int e = 0; e < E; ++e) {
        wait_vmcnt_dyn(__builtin_amdgcn_readfirstlane(has_we ? q_e - 1 - e : 0));       // the tiles behind unit e's stay in flight
        __builtin_amdgcn_s_barrier();
        if (q_e < E) issue_extra(q_e++);                            // (its ring slot held extra unit q_e - 12 or a main unit)
        int j, h;
        const bool second = extra_decode(e, j, h);
        long long src = -1;
        if (my_row < a.n_out) {
            if (second) src = my_row;
            else if (lm_s[j * WIN_T + my_t] == WIN_OUT) src = a.nbr[my_row * 27 + j];
        }
        if (!__any(src >= 0)) continue;
        uint4 fa[2];
        if (src >= 0) {
            const unsigned char* p = (second ? in2_b + (size_t)src * in2_row_bytes : in_b + (size_t)src * in_row_bytes) +
                                     (size_t)((h >> 1) * 128 + (h & 1) * 32 + half * 16);
            fa[0] = *reinterpret_cast<const uint4*>(p);
            fa[1] = *reinterpret_cast<const uint4*>(p + 64);
        } else {
            fa[0] = fa[1] = make_uint4(0u, 0u, 0u, 0u);
        }
        u32x4v B[2 * NB];
        win_read_b<NB>(B, b_rd + (unsigned)(((U + e) % R) * B_UNIT));
        mfma_frags(__builtin_bit_cast(u32x4v, fa[0]), __builtin_bit_cast(u32x4v, fa[1]), &B[0], &B[NB]);
    }
    wait_vmcnt_le<0>();
    __syncthreads();                                                // the window buffers are dead: the epilogue tile reuses them
    {
        const float sc = a.acc_scale_dev ? a.acc_scale * *a.acc_scale_dev : a.acc_scale;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] *= sc;
    }
    float (*ep)[EP_LD] = reinterpret_cast<float (*)[EP_LD]>(lds + wave * 32 * EP_LD * 4);
    ConvArgs ae = a;
    ae.splits = 1;
    ae.tickets = nullptr;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) epilogue_store_wide(ae, acc[nb], rows_s + wave * 32, nb * 32, lane, ep);
}

// OFF by default: measured slower than the mask-sorted kernels on MI355X (profiles/r5/win_*.txt):
// 80 us against 63 us per 96 -> 96 layer at tensor stride 2, 161 us against 93 us at stride 1 (313 tiles on 256 CUs: two rounds)
std::atomic<long long> g_win_on{getenv("CV_WIN") ? atoll(getenv("CV_WIN")) : 0};
std::atomic<long long> g_win_xcd{getenv("CV_WIN_XCD") ? atoll(getenv("CV_WIN_XCD")) : 1};
std::atomic<long long> g_win_levels{getenv("CV_WIN_LEVELS") ? atoll(getenv("CV_WIN_LEVELS")) : 31};      // levels that may take windows

size_t win_lds_bytes(long long n) { return (size_t)((n + 31) / 32) * 6 + 64; }

}  // namespace

namespace cvsc {

bool win_option(const char* name, long long value, long long* previous) {
    std::atomic<long long>* o = !strcmp(name, "win") ? &g_win_on : !strcmp(name, "win_xcd") ? &g_win_xcd :
                                !strcmp(name, "win_levels") ? &g_win_levels : nullptr;
    if (!o) return false;
    const long long before = o->exchange(value, std::memory_order_relaxed);
    if (previous) *previous = before;
    return true;
}
bool win_option_get(const char* name, long long* value) {
    const std::atomic<long long>* o = !strcmp(name, "win") ? &g_win_on : !strcmp(name, "win_xcd") ? &g_win_xcd :
                                      !strcmp(name, "win_levels") ? &g_win_levels : nullptr;
    if (!o) return false;
    *value = o->load(std::memory_order_relaxed);
    return true;
}
bool win_enabled() { return g_win_on.load(std::memory_order_relaxed) != 0; }
int win_level_mask() { return (int)g_win_levels.load(std::memory_order_relaxed); }

bool win_eligible(const ConvArgs& a) {
    return a.win && a.in_hl && a.K == 27 && a.j_begin == 0 && a.j_end == 27 && a.nbr && a.n_in == a.n_out && a.wp6 &&
           a.pieces == 2 && a.wide && (a.cout == 32 || a.cout == 64 || a.cout == 96) && a.cin % KC == 0 && a.cin >= KC &&
           !a.acc_in && (!a.in2 || (a.wp6_2 && a.cin2 % KC == 0));
}

int launch_win(const ConvArgs& a, hipStream_t st) {
    const long long ntiles = (a.n_out + WIN_T - 1) / WIN_T;
    int per = 0;
    unsigned grid = (unsigned)ntiles;
    if (g_win_xcd.load(std::memory_order_relaxed) && ntiles >= 16) {
        per = (int)((ntiles + 7) / 8);
        grid = (unsigned)per * 8u;
    }
    if (a.cout == 32) conv_win<1, 3><<<grid, 512, 0, st>>>(a, per);
    else if (a.cout == 64) conv_win<2, 1><<<grid, 512, 0, st>>>(a, per);
    else conv_win<3, 1><<<grid, 512, 0, st>>>(a, per);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

}  // namespace cvsc

int cv_sp_windows_batch(const CvWinJob* jobs, int n_jobs, void* stream) {
    CV_REQUIRE(jobs && n_jobs > 0 && n_jobs <= CV_MAX_WIN_JOBS, CV_EINVAL, "bad window batch");
    WinJobsDev d;
    d.n = n_jobs;
    int tiles = 0;
    long long max_n = 0;
    for (int i = 0; i < n_jobs; ++i) {
        CV_REQUIRE(jobs[i].nbr && jobs[i].win && jobs[i].n > 0 && cv_sp_windows_supported(jobs[i].n), CV_EINVAL, "bad window job %d", i);
        d.tile_begin[i] = tiles;
        d.nbr[i] = jobs[i].nbr;
        d.rows[i] = jobs[i].n;
        d.win[i] = jobs[i].win;
        tiles += (int)((jobs[i].n + WIN_T - 1) / WIN_T);
        max_n = std::max(max_n, jobs[i].n);
    }
    d.tile_begin[n_jobs] = tiles;
    const int cap_words = (int)((max_n + 31) / 32);
    const size_t lds = win_lds_bytes(max_n);
    if (lds > 64 * 1024)
        CV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(build_windows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    build_windows<<<(unsigned)tiles, 256, lds, static_cast<hipStream_t>(stream)>>>(d, cap_words);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

extern "C" {

int cv_sp_windows_supported(long long n) { return n > 0 && win_lds_bytes(n) <= (size_t)150 * 1024 ? 1 : 0; }

size_t cv_sp_windows_words(long long n) {
    if (n <= 0) return 0;
    const size_t ntiles = (size_t)((n + WIN_T - 1) / WIN_T);
    return cv_align_up(ntiles * (size_t)(WIN_CAP + 27 * WIN_T / 2), 64);
}

int cv_sp_build_windows(const int32_t* d_nbr, long long n, int32_t* d_win, void* stream) {
    CV_REQUIRE(d_nbr && d_win && n > 0, CV_EINVAL, "bad window arguments");
    CV_REQUIRE(cv_sp_windows_supported(n), CV_EINVAL, "coordinate set too large for the window plan (%lld rows)", n);
    CvWinJob j = {d_nbr, n, d_win};
    return cv_sp_windows_batch(&j, 1, stream);
}

}  // extern "C"
