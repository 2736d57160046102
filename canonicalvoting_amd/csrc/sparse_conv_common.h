// Launch arguments, the hl format and the fused epilogue of the convolution kernels in sparse_conv.hip.  (Rounds 1-3 kept
// parity-tested experiments - a wave-independent flavour, a pair-compacted tile flavour, an instrumented twin - in a
// second translation unit over this header; they were removed in round 4, git history and LABNOTES.md hold them.)
#pragma once
#include "cv_common.h"

#include <type_traits>
#include <utility>

namespace cvsc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 32;            // K chunk (channels per staging step)
constexpr int TM = 128;           // rows per workgroup
constexpr int A_LD = TM + 2;      // A staged k-major: A_s[k][row]; +2 makes the 4-row-strided staging stores 2-way (free) instead of 4-way conflicted
constexpr int THREADS = 256;

struct ConvArgs {
    const float* in; long long n_in; int in_ld; int cin;
    const float* w; int K; int cout;
    const int* nbr; long long n_out;
    const float* scale; const float* shift;
    const float* res; int res_ld;
    int relu;
    float* out; int out_ld;
    int splits;            // >1: blockIdx.z handles a contiguous sub-range of the offsets and writes raw
    float* partial;        //     partial sums to partial[split][n_out][cout] (finished by conv_finish)
    const int* row_perm;   // optional processing order: tile row t works on output row row_perm[t]
    int perm_per_split;    // 1: row_perm is [splits][n_out], one order per offset group (blockIdx.z)
    int j_begin, j_end;    // kernel offsets handled by this launch
    const float* acc_in;   // optional [n_out][acc_ld] added to the accumulator before the epilogue
    int acc_ld;
    const int* plan_ent;   // tile flavour: compacted (input row, tile row) lists per (tile, offset), see tile_plan
    const int* plan_cnt;
    const float4* wp;      // tile flavour: weights in MFMA operand order, see pack_weights
    int wide;              // epilogue operands are 16-byte aligned with leading dimensions % 4 == 0: float4 row stores
    const int* nbr_perm;   // mask-sorted groups: [splits][n_out][nbr_perm_w] kernel map rows in processing order
    int nbr_perm_w;
    int reserved0;         // (was the switch of round 1's instrumented kernel)
    const unsigned short* wp6;   // weights split into bf16 pieces, see pack_weights_x6
    const float* in2;            // second source on the output rows (out += in2 @ W2), or NULL
    int in2_ld, cin2;
    const unsigned short* wp6_2;
    int pieces;                  // 3: wp6 holds bf16 triples (six piece products), 2: fp16 pairs (three piece products)
    float acc_scale;             // fp16 pairs: the packed weights carry a power-of-two factor; accumulators *= acc_scale
    int* range_flag;             // fp16 pairs: set to 1 when a staged input magnitude does not fit fp16
    int in_hl, out_hl, res_hl;   // 1: the operand is in the hl format (fp16 pairs in place, see below) instead of fp32
    int xcd_tiles;               // conv_rows_wp / conv_hl: XCD-aware tile numbering (xcd_tile)
    int* tickets;                // conv_hl split-K: arrival counters per output tile (zero; the last arriver reduces, see there)
    const float* acc_scale_dev;  // optional device scalar multiplied into acc_scale (the input gradient's hl operand carries a
                                 // per-layer power of two chosen on the device)
    const unsigned char* gvalid; // mask groups: [splits][n_out] 1 = the row has a neighbour in the group; tiles without any write no
                                 // partial tile and conv_finish_small reads none for such (group, row) pairs (NULL: off)
};

// ---- hl format: activations stored as the fp16 pairs the matrix cores multiply --------------------------------------
// A row of C channels (C % 32 == 0) keeps its 4*C bytes: 32-channel chunk q occupies bytes [128 q, 128 q + 128) =
// 32 fp16 high pieces h = RNE16(x), then the 32 low pieces l = RNE16(x - h) (split2h below).  A convolution that reads
// the format loads its MFMA operand fragments straight from global memory (no split, no LDS staging of the gathered
// rows: a lane's 16-byte pieces are contiguous), the producing epilogue splits every value ONCE instead of once per
// gather (27 x for a 3x3x3 kernel).  Column windows that start at a multiple of 32 channels keep the plain pointer
// arithmetic (32 channels = 32 floats = 128 bytes).  h + l reproduces x to 2^-24 relative (|x| < 65504).
__device__ __forceinline__ void hl_split2(float x0, float x1, unsigned& h, unsigned& l);
__device__ __forceinline__ float4 hl_load4(const float* row, int col) {          // col % 4 == 0
    const unsigned char* p = reinterpret_cast<const unsigned char*>(row) + (col >> 5) * 128 + (col & 31) * 2;
    const uint2 h = *reinterpret_cast<const uint2*>(p), l = *reinterpret_cast<const uint2*>(p + 64);
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h0 = __builtin_bit_cast(h2, h.x), h1 = __builtin_bit_cast(h2, h.y), l0 = __builtin_bit_cast(h2, l.x),
             l1 = __builtin_bit_cast(h2, l.y);
    return make_float4((float)h0[0] + (float)l0[0], (float)h0[1] + (float)l0[1], (float)h1[0] + (float)l1[0],
                       (float)h1[1] + (float)l1[1]);
}
__device__ __forceinline__ void hl_store4(float* row, int col, float4 v) {
    unsigned h0, l0, h1, l1;
    hl_split2(v.x, v.y, h0, l0);
    hl_split2(v.z, v.w, h1, l1);
    unsigned char* p = reinterpret_cast<unsigned char*>(row) + (col >> 5) * 128 + (col & 31) * 2;
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + 64) = make_uint2(l0, l1);
}
__device__ __forceinline__ bool hl_out_of_range(float4 v) {
    return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) > 65000.f;
}

// Partial tiles (split-K / mask-group sums) are written once and read once by the finish launch: streamed past the
// caches with the nontemporal policy (CV_NT_PARTIAL bit 0: the stores, bit 1: the finish launch's loads) so that they do
// not push the activations and weight slabs, which ARE re-read, out of the 4 MB L2s.  Measured (profiles/r3/nt_partial_ab.txt):
// net 2.42 -> 2.375 ms one scene in flight, 483 -> 495 scenes/s six in flight with both; the XCD-aware tile numbering on top of
// either: 2.61-2.64 ms (still slower - the cost ordering of the tiles is worth more than the L2 hits).
#ifndef CV_NT_PARTIAL
#define CV_NT_PARTIAL 3
#endif
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void partial_store4(float* p, float4 v) {
    if (CV_NT_PARTIAL & 1) {
        f32x4v t; t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4v*>(p));
    } else {
        *reinterpret_cast<float4*>(p) = v;
    }
}
// split-K partial tiles that the LAST-ARRIVING workgroup of the output tile reduces inside the same launch
// (ConvArgs.tickets): written through to memory with sc1 stores - visible to every XCD once the wave's vmcnt has drained,
// no agent-scope release (whose buffer_wbl2 writes the whole XCD's dirty L2 back: round 2's version of this path ran the
// forward 50 % slower for it).  The statement ends with s_nop 1: hipcc does not know the store still reads its registers.
__device__ __forceinline__ void partial_store4_wt(float* p, float4 v) {
    f32x4v t; t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ float4 partial_load4(const float4* p) {
    if (CV_NT_PARTIAL & 2) {
        const f32x4v t = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p));
        return make_float4(t[0], t[1], t[2], t[3]);
    }
    return *p;
}

__device__ __forceinline__ void epilogue_store(const ConvArgs& a, const f32x16& acc, const int* rows,
                                               int col, int lane) {
    if (col >= a.cout) return;
    if (a.splits > 1) {
        float* p = a.partial + (long long)blockIdx.z * a.n_out * a.cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rows[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
            if (row >= 0) p[(long long)row * a.cout + col] = acc[r];
        }
        return;
    }
    const float sc = a.scale ? a.scale[col] : 1.f;
    const float sh = a.shift ? a.shift[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rows[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
        if (row < 0) continue;
        float v = acc[r];
        if (a.acc_in) v += a.acc_in[(long long)row * a.acc_ld + col];
        v = v * sc + sh;
        if (a.res) v += a.res[(long long)row * a.res_ld + col];
        if (a.relu) v = fmaxf(v, 0.f);
        a.out[(long long)row * a.out_ld + col] = v;
    }
}

// the fused epilogue on four consecutive columns of one output row (16-byte aligned operands): partial-sum input, folded
// BatchNorm affine / bias, residual (fp32 or hl), ReLU, store (fp32 or hl + range flag)
__device__ __forceinline__ void epilogue_apply4(const ConvArgs& a, long long row, int col, float4 v) {
    if (a.acc_in) {
        const float4 p = *reinterpret_cast<const float4*>(a.acc_in + row * a.acc_ld + col);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    const float4 sc = a.scale ? *reinterpret_cast<const float4*>(a.scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = a.shift ? *reinterpret_cast<const float4*>(a.shift + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    if (a.res) {
        const float4 p = a.res_hl ? hl_load4(a.res + row * a.res_ld, col)
                                  : *reinterpret_cast<const float4*>(a.res + row * a.res_ld + col);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (a.out_hl) {
        if (a.range_flag && hl_out_of_range(v)) *a.range_flag = 1;
        hl_store4(a.out + row * a.out_ld, col, v);
    } else {
        *reinterpret_cast<float4*>(a.out + row * a.out_ld + col) = v;
    }
}

// The same epilogue with 16-byte stores: the 32x32 accumulator tile goes through a wave-private LDS tile so that
// 8 lanes write 128 contiguous bytes of one output row (4 dwordx4 stores per lane instead of 16 dword stores).
// Measured with the instrumented twin (profiles/conv_phases.py): the dword epilogue was 63 % of the wave time
// of the split ts16 convs and 20 % of the mask-sorted ts1 convs - it is store-ISSUE bound, not bandwidth bound.
constexpr int EP_LD = 36;
__device__ __forceinline__ void epilogue_store_wide(const ConvArgs& a, const f32x16& acc, const int* rows, int colbase,
                                                    int lane, float (*T)[EP_LD]) {
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) T[(r & 3) + 8 * (r >> 2) + 4 * h][c] = acc[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int col = colbase + (lane & 7) * 4;
    if (col < a.cout) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rl = (lane >> 3) + 8 * it;
            const int row = rows[rl];
            if (row < 0) continue;
            float4 v = *reinterpret_cast<const float4*>(&T[rl][(lane & 7) * 4]);
            if (a.splits > 1) {
                float* pp = a.partial + ((long long)blockIdx.z * a.n_out + row) * a.cout + col;
                if (a.tickets) partial_store4_wt(pp, v);
                else partial_store4(pp, v);
                continue;
            }
            epilogue_apply4(a, row, col, v);
        }
    }
    __builtin_amdgcn_wave_barrier();       // the tile is rewritten by the next column block
}

typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {      // RNE, lo in bits 0-15
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// two fp32 values -> their h / m / l bf16 pieces, packed (first value in the low half)
__device__ __forceinline__ void split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

// fp16 pairs: x = h + l with h = RNE16(x), l = RNE16(x - h): 11 + 11 significant bits and l's own sign, i.e. x to
// 2^-24 relative as long as neither piece leaves the fp16 range (|x| < 65504; below 2^-14 the absolute error floor is
// 2^-25).  Three piece products (hh, hl, lh; the dropped ll is <= 2^-24 of the product) instead of six.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h(float x0, float x1, unsigned& h, unsigned& l) {
    const f16x2 hv = {(_Float16)x0, (_Float16)x1};                          // v_cvt_pk_f16_f32 (RNE)
    h = __builtin_bit_cast(unsigned, hv);
    // x - h with the fp16 half read in place: v_fma_mix_f32 (fma(h, -1, x) is the exactly rounded difference, the value
    // v_sub_f32 gives) instead of v_cvt_f16_f32 + v_cvt_f32_f16 + v_sub_f32 on a second, scalar conversion of x:
    // 4 -> 2 VALU instructions per value in the staging loop (64 -> 32 per unit and lane)
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
    const f16x2 lv = {(_Float16)r0, (_Float16)r1};
    l = __builtin_bit_cast(unsigned, lv);
}

__device__ __forceinline__ void hl_split2(float x0, float x1, unsigned& h, unsigned& l) { split2h(x0, x1, h, l); }


// ---- shared by the LDS-DMA kernels (conv_hd in sparse_conv.hip) ------------------------------------------------------------
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

static __device__ __attribute__((aligned(128))) unsigned char g_zero_chunk[128];
template <int N>
__device__ __forceinline__ void wait_vmcnt_le() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {      // n is wave-uniform, 0 ... 8
    switch (n) {
        case 0: wait_vmcnt_le<0>(); break;
        case 1: wait_vmcnt_le<1>(); break;
        case 2: wait_vmcnt_le<2>(); break;
        case 3: wait_vmcnt_le<3>(); break;
        case 4: wait_vmcnt_le<4>(); break;
        case 5: wait_vmcnt_le<5>(); break;
        case 6: wait_vmcnt_le<6>(); break;
        case 7: wait_vmcnt_le<7>(); break;
        default: wait_vmcnt_le<8>(); break;
    }
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// one LDS-DMA request: every lane's 16 bytes at g land at l + 16 * lane (l wave-uniform).  A plain device function: inside a
// generic lambda the builtin keeps hipcc's host pass from emitting the kernel's launch stub.
__device__ __forceinline__ void lds_dma16(const void* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)l, 16, 0, 0);
}
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
// MFMA fragments of one k-step out of a ring stage: A high / low piece of the lane's row, NB x (high, low) weight pieces.
// Inline asm (and a plain device function, not a lambda: the host pass must not meet the register constraints): hipcc
// orders every LDS read that may alias an LDS-DMA destination behind vmcnt(0) - it would drain the requests of units k + 1
// and k + 2 in front of unit k's MFMAs.  The waits that matter are conv_hd's counted vmcnt and its workgroup barrier.
template <int NB>
__device__ __forceinline__ void hd_read_frags(unsigned aa0, unsigned aa1, unsigned ab, u32x4v& A0, u32x4v& A1,
                                              u32x4v (&B0)[NB], u32x4v (&B1)[NB]) {
    constexpr int P1 = NB * 32 * 64;                            // low-piece plane of the weight tile
    if constexpr (NB == 1) {
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:%7\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(A0), "=&v"(A1), "=&v"(B0[0]), "=&v"(B1[0])
                     : "v"(aa0), "v"(aa1), "v"(ab), "i"(P1) : "memory");
    } else if constexpr (NB == 2) {
        asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %8 offset:%9\n\t"
                     "ds_read_b128 %4, %8 offset:%10\n\tds_read_b128 %5, %8 offset:%11\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(A0), "=&v"(A1), "=&v"(B0[0]), "=&v"(B1[0]), "=&v"(B0[1]), "=&v"(B1[1])
                     : "v"(aa0), "v"(aa1), "v"(ab), "i"(P1), "i"(2048), "i"(P1 + 2048) : "memory");
    } else {
        static_assert(NB == 3, "conv_hd: 32, 64 or 96 columns per workgroup");
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %10 offset:%11\n\t"
                     "ds_read_b128 %4, %10 offset:%12\n\tds_read_b128 %5, %10 offset:%13\n\t"
                     "ds_read_b128 %6, %10 offset:%14\n\tds_read_b128 %7, %10 offset:%15\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(A0), "=&v"(A1), "=&v"(B0[0]), "=&v"(B1[0]), "=&v"(B0[1]), "=&v"(B1[1]), "=&v"(B0[2]), "=&v"(B1[2])
                     : "v"(aa0), "v"(aa1), "v"(ab), "i"(P1), "i"(2048), "i"(P1 + 2048), "i"(4096), "i"(P1 + 4096) : "memory");
    }
}

// ---- across the two translation units
int launch_finish(const ConvArgs& a, hipStream_t st);                  // sparse_conv.hip: reduce the partial tiles + epilogue
int nb_full(int cout);                                                 // sparse_conv.hip
}  // namespace cvsc
