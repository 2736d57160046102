// Which bits of a hipExtStreamCreateWithCUMask mask belong to which XCD?  Launches a kernel that histograms
// HW_REG_XCC_ID over its workgroups under (a) bits i with i % 8 == k, (b) bits [32 k, 32 k + 32).
// hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip && ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void where(unsigned* hist, unsigned* cu_seen) {
    if (threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID, bits [3:0]
        atomicAdd(&hist[xcc & 15], 1u);
    }
    // keep the workgroup alive for a while so that the grid spreads over every enabled CU
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 20000) {}
    (void)cu_seen;
}
int main() {
    unsigned* d; hipMalloc(&d, 64); 
    for (int mode = 0; mode < 2; ++mode)
        for (int k = 0; k < 8; k += 3) {
            std::vector<uint32_t> mask(8, 0);
            for (int i = 0; i < 256; ++i) {
                bool on = mode == 0 ? (i % 8 == k) : (i / 32 == k);
                if (on) mask[i / 32] |= 1u << (i % 32);
            }
            hipStream_t st;
            if (hipExtStreamCreateWithCUMask(&st, 8, mask.data()) != hipSuccess) { printf("create failed\n"); return 1; }
            hipMemsetAsync(d, 0, 64, st);
            where<<<2048, 64, 0, st>>>(d, nullptr);
            unsigned h[16]; hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
            printf("%s k=%d : XCC histogram", mode == 0 ? "bits i%%8==k " : "bits i/32==k", k);
            for (int x = 0; x < 8; ++x) printf(" %u", h[x]);
            printf("\n");
            hipStreamDestroy(st);
        }
    return 0;
}
