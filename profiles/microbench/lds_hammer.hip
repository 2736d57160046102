// Co-resident load generators for the scenes-in-flight finding of the vote tile kernel (profiles/r3/vote_concurrency_findings.txt):
// which ingredient of the fp16 / bf16 convolution kernels disturbs a 512-thread tile workgroup on the same CU?
// Workgroups of 256 threads with 26 KB of LDS (what conv_hl<1/2> allocates), a few thousand iterations each.
//   mode 1: ds_write_b128 + ds_read_b128 traffic        mode 2: ds_write_b32 + ds_read_b32 traffic
//   mode 3: v_mfma_f32_32x32x16_f16 only (no LDS use)    mode 4: v_mfma_f32_32x32x2_f32 only
//   mode 5: 1 + 3 (b128 LDS traffic and 16-bit MFMAs)    mode 6: ds_add_u64 atomics on its own LDS
//   mode 7: global loads / stores only (streams a buffer through the L1)
//   mode 8: v_mfma_f32_32x32x16_bf16    mode 9: v_mfma_f32_16x16x32_f16    mode 10: v_mfma_f32_32x32x8_f16 (the gfx90a-era shape)
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC lds_hammer.hip -o liblds_hammer.so
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void hammer(int mode, int iters, float* sink, const float* src, long long n_src) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[26 * 1024];
    const int tid = threadIdx.x;
    uint4 v = make_uint4(tid, tid * 3, tid * 5, tid * 7);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (tid + k)); b[k] = (_Float16)(0.002f * (tid - k)); }
    float fa = 0.001f * tid, fb = 0.5f;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    f16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    unsigned long long* lq = reinterpret_cast<unsigned long long*>(lds);
    float g = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (mode == 1 || mode == 5) {
            uint4* p = reinterpret_cast<uint4*>(lds) + ((tid * 5 + it * 7) & 1535);
            *p = v;
            const uint4 q = reinterpret_cast<uint4*>(lds)[(tid * 3 + it) & 1535];
            v.x ^= q.x; v.y += q.y; v.z ^= q.z; v.w += q.w;
        }
        if (mode == 2) {
            unsigned* p = reinterpret_cast<unsigned*>(lds) + ((tid * 5 + it * 7) & 6143);
            *p = v.x;
            v.x ^= reinterpret_cast<unsigned*>(lds)[(tid * 3 + it) & 6143];
        }
        if (mode == 3 || mode == 5) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
        }
        if (mode == 4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        if (mode == 8) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, a), __builtin_bit_cast(bf16x8v, b), acc, 0, 0, 0);
        if (mode == 9) { acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4, 0, 0, 0); }
        if (mode == 10) acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc, 0, 0, 0);
        if (mode == 6) __hip_atomic_fetch_add(&lq[(tid * 7 + it * 13) & 3071], (unsigned long long)v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (mode == 7) g += src[((long long)blockIdx.x * 256 + tid + (long long)it * 65536) % n_src];
    }
    float s = g + acc4[0] + acc4[1] + acc4[2] + acc4[3];
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 1234.5f || (v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) sink[blockIdx.x] = s;
    if (mode == 6 && tid == 0 && lq[0] == 0x123456789ull) sink[0] = 1.f;
}

extern "C" int lds_hammer_launch(int mode, int blocks, int iters, float* sink, const float* src, long long n_src, void* stream) {
    hammer<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(mode, iters, sink, src, n_src);
    return (int)hipGetLastError();
}

// ---- which instruction class goes wrong next to the 16-bit MFMAs?  Every thread folds the results of a few thousand
// deterministic evaluations of each class into a checksum; a run with nothing else on the GPU writes the reference sums,
// a run next to the hammer compares.  One counter per class = threads whose checksum changed.
constexpr int N_CLASS = 24;
__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ float uf(unsigned& s, float lo, float hi) { return lo + (hi - lo) * (float)(lcg(s) >> 8) * (1.0f / 16777216.0f); }
__global__ __launch_bounds__(512) void op_check(int iters, unsigned long long* ref, int write_ref, unsigned* mism) {
    __shared__ unsigned long long acc[512];
    __shared__ float tabf[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const long long gid = (long long)blockIdx.x * 512 + tid;
    acc[tid] = 0ull;
    if (tid < 256) tabf[tid] = 0.37f * tid;
    __syncthreads();
    unsigned s = (unsigned)gid * 2654435761u + 12345u;
    unsigned long long c[N_CLASS];
    for (int k = 0; k < N_CLASS; ++k) c[k] = 0ull;
    for (int it = 0; it < iters; ++it) {
        const float a = uf(s, -3.f, 7.f), b = uf(s, 0.01f, 0.05f), p = uf(s, 0.f, 6.f);
        // 0: IEEE fp32 division (v_div_scale / v_rcp / v_div_fmas / v_div_fixup)
        const float q = ((p + a) - 1.5f) / b;
        c[0] += __float_as_uint(q);
        // 1: v_rcp_f32      2: v_sqrt_f32     3: floor + float -> int
        c[1] += __float_as_uint(__builtin_amdgcn_rcpf(b));
        c[2] += __float_as_uint(sqrtf(a * a + p * p));
        c[3] += (unsigned)(int)q + __float_as_uint(q - floorf(q));
        // 4: fp32 multiply-add chain (plain VALU)
        const float m = (-a) * p + b * a;
        c[4] += __float_as_uint(m * b - p);
        // 5: f32 -> f64, f64 fma (magic-number fixed point)     6: f64 division
        const double d = __builtin_fma((double)m, 68719476736.0, 6755399441055744.0);
        c[5] += (unsigned long long)__double_as_longlong(d);
        c[6] += (unsigned long long)__double_as_longlong((double)a / ((double)b + 1e-7));
        // 7: ds_bpermute (__shfl_up)      8: ballot + mbcnt      9: readlane
        c[7] += (unsigned)__shfl_up((int)__float_as_uint(a), 1 + (it & 31));
        const unsigned long long bal = __ballot(a > p);
        c[8] += bal + (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
        c[9] += (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(p), 17);
        // 10: 64-bit integer multiply / add      11: LDS read by a computed index
        c[10] += (unsigned long long)(long long)((long long)(int)q * 4099 + it) * 88 + 175;
        c[11] += __float_as_uint(tabf[lcg(s) >> 24]);
        // 12: LDS 64-bit atomic add (own slot, then read back at the end)     13: polynomial atan (fma chain + rcp)
        __hip_atomic_fetch_add(&acc[tid], (unsigned long long)__float_as_uint(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const float r = fminf(fabsf(a), p) * __builtin_amdgcn_rcpf(fmaxf(fmaxf(fabsf(a), p), 1e-30f));
        c[13] += __float_as_uint(((-0.0464964749f * r * r + 0.15931422f) * r * r - 0.327622764f) * r * r * r + r);
        // 14-16: the packed fp32 instructions, one at a time; 17: the same arithmetic on v_mul_f32 / v_add_f32
        {
            const f32x2 x = {a, p}, y = {b, m};
            f32x2 r0, r1, r2;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(x), "v"(y));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r1) : "v"(x), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r2) : "v"(x), "v"(y), "v"(x));
            c[14] += __float_as_uint(r0[0]) + 3u * __float_as_uint(r0[1]);
            c[15] += __float_as_uint(r1[0]) + 3u * __float_as_uint(r1[1]);
            c[16] += __float_as_uint(r2[0]) + 3u * __float_as_uint(r2[1]);
            float s0, s1, t0, t1;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(p), "v"(m));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(t0) : "v"(a), "v"(b));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(t1) : "v"(p), "v"(m));
            c[17] += __float_as_uint(s0) + 3u * __float_as_uint(s1) + 5u * __float_as_uint(t0) + 7u * __float_as_uint(t1);
            // 18-23: packed fp32 instructions with the operand modifiers the compiler attaches (SLP-vectorised scalar code)
            f32x2 q0, q1, q2, q3, q4, q5;
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(q0) : "v"(x), "v"(y));
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(q1) : "v"(x));
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(q2) : "v"(x), "v"(y));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(q3) : "v"(x), "v"(y));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(q4) : "v"(x), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(q5) : "v"(x), "v"(y), "v"(x));
            c[18] += __float_as_uint(q0[0]) + 3u * __float_as_uint(q0[1]);
            c[19] += __float_as_uint(q1[0]) + 3u * __float_as_uint(q1[1]);
            c[20] += __float_as_uint(q2[0]) + 3u * __float_as_uint(q2[1]);
            c[21] += __float_as_uint(q3[0]) + 3u * __float_as_uint(q3[1]);
            c[22] += __float_as_uint(q4[0]) + 3u * __float_as_uint(q4[1]);
            c[23] += __float_as_uint(q5[0]) + 3u * __float_as_uint(q5[1]);
        }
    }
    __syncthreads();
    c[12] = acc[tid];
    for (int k = 0; k < N_CLASS; ++k) {
        unsigned long long* slot = ref + (long long)k * gridDim.x * 512 + gid;
        if (write_ref) *slot = c[k];
        else if (*slot != c[k]) atomicAdd(&mism[k], 1u);
    }
    (void)lane;
}
extern "C" int op_check_launch(int blocks, int iters, unsigned long long* ref, int write_ref, unsigned* mism, void* stream) {
    op_check<<<blocks, 512, 0, static_cast<hipStream_t>(stream)>>>(iters, ref, write_ref, mism);
    return (int)hipGetLastError();
}
