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
constexpr int WIN_LM = 28;                 // 16-bit map entries per row (27 offsets + 1 pad: 56 bytes, 8-byte aligned)
constexpr unsigned WIN_NONE = 0xFFFFu, WIN_OUT = 0xFFFEu;
static_assert(WIN_CAP % 64 == 0 && WIN_CAP < 0xFFFE, "window capacity");

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
    unsigned* __restrict__ lm_out = reinterpret_cast<unsigned*>(jobs.win[job] + ntiles * WIN_CAP) +
                                    ((long long)tile * WIN_T + t) * (WIN_LM / 2);
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
    unsigned out[WIN_LM / 2];
#pragma unroll
    for (int q = 0; q < WIN_LM / 2; ++q) out[q] = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 27; ++j) {
        unsigned v = WIN_NONE;
        if (e[j] >= 0) {
            const int w = (e[j] >> 5) - lo_w;
            const unsigned rank = (unsigned)wpre[w] + (unsigned)__popc(bits[w] & ((1u << (e[j] & 31)) - 1u));
            v = rank < (unsigned)WIN_CAP ? rank : WIN_OUT;
        }
        out[j >> 1] = (j & 1) ? ((out[j >> 1] & 0x0000FFFFu) | (v << 16)) : ((out[j >> 1] & 0xFFFF0000u) | v);
    }
#pragma unroll
    for (int q = 0; q < WIN_LM / 2; ++q) lm_out[q] = out[q];
}

// ------------------------------------------------------------------ the convolution
#ifndef CV_WIN_ABL
#define CV_WIN_ABL 0      // timing ablations (wrong results): 1 no window DMA, 2 no MFMA, 4 no weight DMA, 8 no fragment reads
#endif

template <int NB>
__global__ __launch_bounds__(512, 2) void conv_win(ConvArgs a, int xcd_per) {
    constexpr int NW = 8, NSTG = 3;
    constexpr int WBUF = (WIN_CAP + 1) * 128;                       // one window buffer: WIN_CAP rows + the row of zeros
    constexpr int B_BYTES = 2 * NB * 32 * 64;                       // weight tile of a unit: [plane][col][64 B]
    constexpr int B_INSTR = B_BYTES / 1024, B_PER_WAVE = (B_INSTR + NW - 1) / NW;
    constexpr int WIN_INSTR = WIN_CAP / 8, WIN_PER_WAVE = WIN_INSTR / NW;          // LDS-DMA instructions of one window chunk
    constexpr int OFF_B = 2 * WBUF, OFF_ROWS = OFF_B + NSTG * B_BYTES, OFF_OM = OFF_ROWS + WIN_T * 4, LDS_TOTAL = OFF_OM + NW * 4;
    constexpr int EP_BYTES = NW * 32 * EP_LD * 4;
    static_assert(EP_BYTES <= WBUF, "the epilogue tile aliases the first window buffer");
    static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
    static_assert(WIN_PER_WAVE * NW * 8 == WIN_CAP, "window instructions are dealt evenly to the waves");
    // ONE __shared__ object (a second one makes hipcc drain vmcnt in front of the LDS reads of an LDS-DMA pipeline)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_TOTAL];
    int* const rows_s = reinterpret_cast<int*>(lds + OFF_ROWS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const long long ntiles = (a.n_out + WIN_T - 1) / WIN_T;
    // XCD-aware numbering: the workgroups an XCD receives (blockIdx.x % 8) walk a contiguous range of tiles, so the windows of
    // neighbouring tiles - which overlap - meet in ONE L2
    const long long tile = xcd_per > 0 ? (long long)(blockIdx.x & 7) * xcd_per + (blockIdx.x >> 3) : (long long)blockIdx.x;
    if (tile >= ntiles) return;
    const long long row0 = tile * WIN_T;
    const int nch = a.cin / KC, nch2 = a.in2 ? a.cin2 / KC : 0;
    const int U = 27 * nch;

    // ---- tile set-up: the lane's 27 window slots, this wave's share of the window rows
    unsigned lm[WIN_LM / 2];
    const unsigned short* const lm16 = reinterpret_cast<const unsigned short*>(a.win + ntiles * WIN_CAP) +
                                       (row0 + wave * 32 + l31) * WIN_LM;        // the same 27 slots in memory (extra units)
    {
        const uint2* p = reinterpret_cast<const uint2*>(lm16);
#pragma unroll
        for (int q = 0; q < WIN_LM / 4; ++q) { const uint2 v = p[q]; lm[2 * q] = v.x; lm[2 * q + 1] = v.y; }
    }
    const unsigned in_row_bytes = (unsigned)a.in_ld * 4u, in2_row_bytes = (unsigned)a.in2_ld * 4u;
    const unsigned char* const in_b = reinterpret_cast<const unsigned char*>(a.in);
    const unsigned char* const in2_b = reinterpret_cast<const unsigned char*>(a.in2);
    // window instruction q = wave + 8 i covers window rows 8 q + (lane >> 3); LDS slot lane & 7 of row w receives the
    // row's piece (lane & 7) ^ ((w >> 1) & 7) - and (w >> 1) & 7 = (4 q + (lane >> 4)) & 7 = (4 (wave & 1) + (lane >> 4)) & 7
    const int a_row = lane >> 3;
    const unsigned w_piece = (unsigned)(((lane & 7) ^ ((4 * (wave & 1) + (a_row >> 1)) & 7)) << 4);
    unsigned woff[WIN_PER_WAVE];
    {
        const int* wr = a.win + tile * WIN_CAP;
#pragma unroll
        for (int i = 0; i < WIN_PER_WAVE; ++i) {
            const int r = wr[8 * (wave + NW * i) + a_row];
            woff[i] = r >= 0 ? (unsigned)r * in_row_bytes : 0xFFFFFFFFu;
        }
    }
    if (tid < WIN_T) rows_s[tid] = row0 + tid < a.n_out ? (int)(row0 + tid) : -1;
    if (tid < 16) *reinterpret_cast<uint4*>(lds + (tid >> 3) * WBUF + WIN_CAP * 128 + (tid & 7) * 16) = make_uint4(0u, 0u, 0u, 0u);
    unsigned jmask = 0u, omask = 0u;                                // offsets this wave has a neighbour at / an outside-window pair at
    static_for<27>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const unsigned v = (lm[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        if (__any(v != WIN_NONE)) jmask |= 1u << j;
        if (__any(v == WIN_OUT)) omask |= 1u << j;
    });
    if (lane == 0) reinterpret_cast<unsigned*>(lds + OFF_OM)[wave] = omask;
    __syncthreads();
    unsigned omask_wg = 0u;                                         // offsets with an outside-window pair in ANY wave of the tile
#pragma unroll
    for (int w = 0; w < NW; ++w) omask_wg |= reinterpret_cast<const unsigned*>(lds + OFF_OM)[w];
    omask_wg = __builtin_amdgcn_readfirstlane(omask_wg);

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    // ---- per-thread invariants of the weight tile requests (conv_hd's layout: slot s of column col holds the slab's
    // 16-byte piece s ^ ((col >> 2) & 3))
    int b_src[B_PER_WAVE];
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
        const int f = (wave + i * NW) * 64 + lane;
        const int pl = f / (NB * 32 * 4), rem = f - pl * (NB * 32 * 4);
        const int col = rem >> 2, slot = rem & 3;
        b_src[i] = (pl * a.cout + min(col, a.cout - 1)) * 32 + ((slot ^ ((col >> 2) & 3)) << 3);
    }
    const unsigned slab_words = 2u * (unsigned)a.cout * 32u;
    // weight tile of a unit -> ring stage: main unit (offset j, chunk c) has stage j % 3 (27 % 3 == 0), second-source unit k
    // stage k % 3 (the main units are a multiple of three)
    auto issue_slab = [&](const unsigned short* slab, int stage) {
        unsigned char* dst = lds + OFF_B + stage * B_BYTES;
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; ++i) {
            const int t = wave + i * NW;
            if (t < B_INSTR && !(CV_WIN_ABL & 4)) lds_dma16(slab + b_src[i], dst + t * 1024);
        }
    };
    auto issue_w = [&](int j, int c) {                              // unit (j, c); c == nch: second-source unit j
        int nch_l = nch;
        asm volatile("" : "+s"(nch_l));                             // (keeps 27 hoisted slab offsets out of the scalar registers)
        if (c < nch_l) issue_slab(a.wp6 + (size_t)(unsigned)(j * nch_l + c) * slab_words, j % NSTG);
        else issue_slab(a.wp6_2 + (size_t)(unsigned)j * slab_words, j % NSTG);
    };
    // instruction i of this wave's share of window chunk c.  ALWAYS one instruction (rows beyond the window - and a whole
    // chunk beyond the last, `dummy` - fetch the line of zeros): the counted waits below then know the queue at compile time
    auto issue_win = [&](int c, int i, bool dummy) {
        if (CV_WIN_ABL & 1) dummy = true;
        const unsigned char* g = (!dummy && woff[i] != 0xFFFFFFFFu) ? in_b + ((size_t)woff[i] + (size_t)(c * 128) + w_piece)
                                                                    : g_zero_chunk + w_piece;
        lds_dma16(g, lds + (c & 1) * WBUF + (wave + NW * i) * 1024);
    };
    // vmcnt wait that leaves `base` + this wave's weight instructions of one unit in flight (nWw is 1 or 2 / 0 or 1 by wave half)
    auto wait_keep = [&](auto BASE, bool plus_w) {
        constexpr int base = decltype(BASE)::value;
        if (!plus_w) wait_vmcnt_le<base>();
        else if (B_INSTR % NW == 0 || wave < B_INSTR % NW) wait_vmcnt_le<base + B_PER_WAVE>();
        else wait_vmcnt_le<base + B_PER_WAVE - 1>();
    };

    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)lds;
    const unsigned bswz = (unsigned)((l31 >> 2) & 3);
    const unsigned b_rd[2] = {lds0 + (unsigned)(OFF_B + l31 * 64) + (((0u + half) ^ bswz) << 4),
                              lds0 + (unsigned)(OFF_B + l31 * 64) + (((2u + half) ^ bswz) << 4)};
    const long long my_row = row0 + wave * 32 + l31;

    auto mfma_step = [&](const u32x4v& A0, const u32x4v& A1, const u32x4v (&B0)[NB], const u32x4v (&B1)[NB]) {
        const f16x8 a0 = __builtin_bit_cast(f16x8, A0), a1 = __builtin_bit_cast(f16x8, A1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f16x8 b0 = __builtin_bit_cast(f16x8, B0[nb]), b1 = __builtin_bit_cast(f16x8, B1[nb]);
            if (!(CV_WIN_ABL & 2)) {
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[nb], 0, 0, 0);
            }
        }
    };
    // main units: window slot -> fragment addresses -> asm fragment reads -> MFMAs.  A lane without a neighbour - or whose
    // neighbour is outside the window (those pairs are added by the extra units below) - reads the row of zeros.
    auto compute_fast = [&](unsigned v, int c, int stage) {
        const unsigned li = v >= WIN_OUT ? (unsigned)WIN_CAP : v;
        const unsigned base = lds0 + (unsigned)((c & 1) * WBUF) + li * 128u, sw = (li >> 1) & 7u;
        unsigned aa[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) aa[k] = base + (((unsigned)(2 * k + half) ^ sw) << 4);
        if (CV_WIN_ABL & 8) return;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4v A0, A1, B0[NB], B1[NB];
            hd_read_frags<NB>(aa[ks], aa[2 + ks], b_rd[ks] + (unsigned)(stage * B_BYTES), A0, A1, B0, B1);
            mfma_step(A0, A1, B0, B1);
        }
    };
    // Extra units, behind the main ones (ONE code instance, a run-time loop - a second path inside the unrolled steps made
    // hipcc shuttle the accumulators between two register sets): e < nch2: chunk e of the second source (A = the output row
    // itself); then, for every offset at which ANY wave of the tile has a pair outside its window, the offset's nch chunks
    // with A = the outside neighbours only.  Their A fragments come straight from global memory, the weight tiles through
    // the ring like every unit's.
    const int n_fix = __popc(omask_wg);
    const int E = nch2 + n_fix * nch, S = U + E;
    auto extra_decode = [&](int e, int& j, int& c) -> bool {         // true: second-source unit
        if (e < nch2) { j = 0; c = e; return true; }
        const int f = e - nch2, idx = f / nch;
        c = f - idx * nch;
        unsigned m = omask_wg;
        for (int q = 0; q < idx; ++q) m &= m - 1u;
        j = __ffs(m) - 1;
        return false;
    };
    auto issue_extra = [&](int e) {
        int j, c;
        if (extra_decode(e, j, c)) issue_slab(a.wp6_2 + (size_t)(unsigned)c * slab_words, e % NSTG);
        else issue_slab(a.wp6 + (size_t)(unsigned)(j * nch + c) * slab_words, e % NSTG);
    };

    // ---- prologue: window chunk 0, weight tiles of units 0 and 1
#pragma unroll
    for (int i = 0; i < WIN_PER_WAVE; ++i) issue_win(0, i, false);
    issue_w(0, 0);
    issue_w(1, 0);
    // Step s: this wave's requests up to the weight tile of unit s have landed (behind it in the queue, allowed to stay in
    // flight: the window instruction issued at step s - 1 and the weight tile of unit s + 1); barrier = everybody's have, and
    // everybody is past the MFMAs of unit s - 1, whose ring stage takes unit s + 2.  The window of chunk c + 1 is requested
    // one instruction per step behind the barriers of chunk c's first steps (its buffer was last read by chunk c - 1).
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        static_for<27>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int stage = j % NSTG, j2 = (j + 2) % 27;
            // in flight behind the weight tile of this unit: the window instruction of step j - 1 (steps 0 ... WIN_PER_WAVE - 1
            // issue one each) and the weight tile of the next unit (none behind the very last unit)
            wait_keep(std::integral_constant<int, (j >= 1 && j - 1 < WIN_PER_WAVE) ? 1 : 0>{}, j < 26 || c * 27 + j + 1 < S);
            __builtin_amdgcn_s_barrier();
            if constexpr (j < WIN_PER_WAVE) issue_win(c + 1, j, c + 1 >= nch);
            if constexpr (j + 2 < 27) {
                issue_w(j2, c);
            } else {
                if (c + 1 < nch) issue_w(j2, c + 1);
                else if (j2 < E) issue_extra(j2);
            }
            if ((jmask >> j) & 1u) {
                unsigned v = (lm[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                asm volatile("" : "+v"(v));                         // (the slot's LDS addresses are formed here, not hoisted for all 27 offsets)
                compute_fast(v, c, stage);
            }
        });
    }
#pragma unroll 1
    for (int e = 0; e < E; ++e) {
        wait_keep(std::integral_constant<int, 0>{}, e + 1 < E);
        __builtin_amdgcn_s_barrier();
        if (e + 2 < E) issue_extra(e + 2);
        int j, c;
        const bool second = extra_decode(e, j, c);
        long long src = -1;
        if (my_row < a.n_out) {
            if (second) src = my_row;
            else if (lm16[j] == WIN_OUT) src = a.nbr[my_row * 27 + j];
        }
        if (!__any(src >= 0)) continue;
        uint4 fa[4];
        if (src >= 0) {
            const unsigned char* p = (second ? in2_b + (size_t)src * in2_row_bytes : in_b + (size_t)src * in_row_bytes) +
                                     (size_t)(c * 128 + half * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) fa[k] = *reinterpret_cast<const uint4*>(p + 32 * k);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) fa[k] = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4v B0[NB], B1[NB];
            hd_read_b<NB>(b_rd[ks] + (unsigned)((e % NSTG) * B_BYTES), B0, B1);
            mfma_step(__builtin_bit_cast(u32x4v, fa[ks]), __builtin_bit_cast(u32x4v, fa[2 + ks]), B0, B1);
        }
    }
    wait_vmcnt_le<0>();
    __syncthreads();                                                // the window buffers are dead: the epilogue tile reuses the first
    {
        const float sc = a.acc_scale;
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

std::atomic<long long> g_win_on{getenv("CV_WIN") ? atoll(getenv("CV_WIN")) : 1};
std::atomic<long long> g_win_xcd{getenv("CV_WIN_XCD") ? atoll(getenv("CV_WIN_XCD")) : 1};

size_t win_lds_bytes(long long n) { return (size_t)((n + 31) / 32) * 6 + 64; }

}  // namespace

namespace cvsc {

bool win_option(const char* name, long long value, long long* previous) {
    std::atomic<long long>* o = !strcmp(name, "win") ? &g_win_on : !strcmp(name, "win_xcd") ? &g_win_xcd : nullptr;
    if (!o) return false;
    const long long before = o->exchange(value, std::memory_order_relaxed);
    if (previous) *previous = before;
    return true;
}
bool win_enabled() { return g_win_on.load(std::memory_order_relaxed) != 0; }

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
    if (a.cout == 32) conv_win<1><<<grid, 512, 0, st>>>(a, per);
    else if (a.cout == 64) conv_win<2><<<grid, 512, 0, st>>>(a, per);
    else conv_win<3><<<grid, 512, 0, st>>>(a, per);
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
    return cv_align_up(ntiles * (size_t)(WIN_CAP + WIN_T * WIN_LM / 2), 64);
}

int cv_sp_build_windows(const int32_t* d_nbr, long long n, int32_t* d_win, void* stream) {
    CV_REQUIRE(d_nbr && d_win && n > 0, CV_EINVAL, "bad window arguments");
    CV_REQUIRE(cv_sp_windows_supported(n), CV_EINVAL, "coordinate set too large for the window plan (%lld rows)", n);
    CvWinJob j = {d_nbr, n, d_win};
    return cv_sp_windows_batch(&j, 1, stream);
}

}  // extern "C"
