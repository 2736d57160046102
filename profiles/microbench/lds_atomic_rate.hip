// Microbenchmark: LDS fp32 atomic-add throughput on gfx950 by address pattern.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_rate.hip -o lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float acc[6144];
    for (int i = threadIdx.x; i < 6144; i += 256) acc[i] = 0.f;
    __syncthreads();
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int it = 0; it < iters; ++it) {
        int idx;
        if (MODE == 0) { h = h * 1664525u + 1013904223u; idx = (h >> 8) % 6144; }          // random
        else if (MODE == 1) idx = (it * 7) % 6144;                                            // all lanes same address
        else if (MODE == 2) idx = (threadIdx.x + it * 256) % 6144;                            // conflict-free
        else if (MODE == 3) { h = h * 1664525u + 1013904223u; idx = ((h >> 8) % 96) * 64 + (threadIdx.x & 63) ; }  // distinct banks, random rows
        else { h = h * 1664525u + 1013904223u; idx = ((h >> 8) % 768) * 8 + ((threadIdx.x >> 3) & 7); }  // 8 lanes share an address
        __hip_atomic_fetch_add(&acc[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    float s = 0;
    for (int i = threadIdx.x; i < 6144; i += 256) s += acc[i];
    if (s == 12345.f) out[blockIdx.x] = s;
}

template <int MODE>
__global__ __launch_bounds__(256) void kint(float* out, int iters) {
    __shared__ unsigned acc[6144];
    for (int i = threadIdx.x; i < 6144; i += 256) acc[i] = 0;
    __syncthreads();
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        int idx = (h >> 8) % 6144;
        atomicAdd(&acc[idx], 1u);
    }
    __syncthreads();
    unsigned s = 0;
    for (int i = threadIdx.x; i < 6144; i += 256) s += acc[i];
    if (s == 12345u) out[blockIdx.x] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void k64(float* out, int iters) {
    __shared__ T acc[6144];
    for (int i = threadIdx.x; i < 6144; i += 256) acc[i] = 0;
    __syncthreads();
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        int idx = (h >> 8) % 6144;
        __hip_atomic_fetch_add(&acc[idx], (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    T s = 0;
    for (int i = threadIdx.x; i < 6144; i += 256) s += acc[i];
    if (s == (T)12345) out[blockIdx.x] = (float)s;
}

// address sharing: SHARE lanes of a wave hit the same (random) address with lane-dependent values
template <typename T, int SHARE>
__global__ __launch_bounds__(256) void kshare(float* out, int iters) {
    __shared__ T acc[6144];
    for (int i = threadIdx.x; i < 6144; i += 256) acc[i] = 0;
    __syncthreads();
    uint32_t h = (threadIdx.x / SHARE) * 2654435761u + blockIdx.x * 40503u;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        int idx = (h >> 8) % 6144;
        __hip_atomic_fetch_add(&acc[idx], (T)(1 + (threadIdx.x & 3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    T s = 0;
    for (int i = threadIdx.x; i < 6144; i += 256) s += acc[i];
    if (s == (T)12345) out[blockIdx.x] = (float)s;
}

template <typename F>
void run(const char* name, F launch, int iters, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    double n = (double)blocks * 256 * iters;
    printf("%-34s %8.3f ms  %8.1f G atomics/s  %6.2f lane-atomics/clk/CU(@2.4GHz,256CU)\n", name, ms, n / ms / 1e6,
           n / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    float* out; hipMalloc(&out, 1 << 20);
    const int iters = 4096, blocks = 256 * 6;
    run("f32 random", [&] { k<0><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("f32 same address (wave)", [&] { k<1><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("f32 conflict-free", [&] { k<2><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("f32 distinct banks random rows", [&] { k<3><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("f32 8 lanes share address", [&] { k<4><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("u32 random", [&] { kint<0><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("u64 random", [&] { k64<unsigned long long><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("f64 random", [&] { k64<double><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("i32 random", [&] { k64<int><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("f64 4 lanes share address", [&] { kshare<double, 4><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("u64 4 lanes share address", [&] { kshare<unsigned long long, 4><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("f64 16 lanes share address", [&] { kshare<double, 16><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("u64 16 lanes share address", [&] { kshare<unsigned long long, 16><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("f64 64 lanes share address", [&] { kshare<double, 64><<<blocks, 256>>>(out, iters); }, iters, blocks);
    run("u64 64 lanes share address", [&] { kshare<unsigned long long, 64><<<blocks, 256>>>(out, iters); }, iters, blocks);
    return 0;
}
