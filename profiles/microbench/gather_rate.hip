// Microbenchmark: rate of the sparse-conv A-operand row gather on gfx950 by lane -> address mapping.
// A wave needs, per (offset, 32-channel) unit, the 128-byte hl chunk of 32 gathered rows (4 KB).  conv_hl loads it
// straight into the MFMA A layout (lane = row, 4 x 16 B per lane at +0/+32/+64/+96 (+16 for the upper half-wave)): every
// lane of a load instruction sits on a different 128-byte line.  Question: is that mapping bound by the vector cache's
// line look-ups, and what does a line-coalesced mapping (8 lanes per 128-byte chunk) reach on the same rows?
// build: hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0: MFMA 32x32x16 A layout (lane&31 = row, lane>>5 = 16-byte half, pieces +0/+32/+64/+96)
// MODE 1: line-coalesced (8 lanes x 16 B = one row's chunk; 4 instructions x 8 rows)
// MODE 2: MFMA 16x16x32 A layout (lane&15 = row, lane>>4 = 16-byte quarter; rows r / r+16, h half then l half)
// MODE 3: as 0, two lanes per row adjacent (lane>>1 = row, lane&1 = half): same bytes, pairs of lanes share a line
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256, 4) void gather(const unsigned char* __restrict__ buf, const int* __restrict__ idx,
                                                  int units, int row_bytes, int chunks, unsigned* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int* my = idx + (size_t)wave * units * 32;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 r[DEPTH][4];
    auto issue = [&](int s, int u) {
        const int c = u % chunks;
        if (MODE == 0) {
            const int row = my[u * 32 + (lane & 31)];
            const unsigned char* p = buf + (size_t)row * row_bytes + c * 128 + (lane >> 5) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) r[s][q] = *reinterpret_cast<const uint4*>(p + 32 * q);
        } else if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = my[u * 32 + (lane >> 3) + 8 * q];
                r[s][q] = *reinterpret_cast<const uint4*>(buf + (size_t)row * row_bytes + c * 128 + (lane & 7) * 16);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = my[u * 32 + (lane & 15) + 16 * (q & 1)];
                r[s][q] = *reinterpret_cast<const uint4*>(buf + (size_t)row * row_bytes + c * 128 + (q >> 1) * 64 + (lane >> 4) * 16);
            }
        } else {
            const int row = my[u * 32 + (lane >> 1)];
            const unsigned char* p = buf + (size_t)row * row_bytes + c * 128 + (lane & 1) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) r[s][q] = *reinterpret_cast<const uint4*>(p + 32 * q);
        }
    };
    auto eat = [&](int s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc.x ^= r[s][q].x; acc.y ^= r[s][q].y; acc.z ^= r[s][q].z; acc.w ^= r[s][q].w; }
    };
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) issue(s, s);
#pragma unroll 1
    for (int u = 0; u < units; u += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const int nxt = u + s + DEPTH - 1;
            if (nxt < units) issue((s + DEPTH - 1) % DEPTH, nxt);
            if (u + s < units) eat(s);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[wave] = acc.x;
}

template <int MODE, int DEPTH>
static int run(const char* name, const unsigned char* buf, const int* idx, int units, int row_bytes, int chunks, unsigned* out,
               int wgs, double clock_ghz, int cus) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) gather<MODE, DEPTH><<<wgs, 256>>>(buf, idx, units, row_bytes, chunks, out);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) gather<MODE, DEPTH><<<wgs, 256>>>(buf, idx, units, row_bytes, chunks, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)wgs * 4 * units * 4096.0;
    printf("  %-34s depth %d: %8.1f us  %7.2f TB/s  %6.2f B/clk/CU\n", name, DEPTH, ms * 1e3, bytes / ms / 1e9,
           bytes / (ms * 1e-3) / (clock_ghz * 1e9) / cus);
    return 0;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("%s: %d CUs, %.2f GHz\n", prop.name, cus, ghz);
    const int row_bytes = 384, chunks = 3, units = 96;
    const int wgs = cus * 4, waves = wgs * 4;
    const int max_rows = 80000;
    unsigned char* buf; int* idx; unsigned* out;
    CK(hipMalloc(&buf, (size_t)max_rows * row_bytes)); CK(hipMemset(buf, 1, (size_t)max_rows * row_bytes));
    CK(hipMalloc(&idx, (size_t)waves * units * 32 * 4)); CK(hipMalloc(&out, waves * 4));
    std::vector<int> h((size_t)waves * units * 32);
    // working set = rows x 384 B, every wave's units draw their 32 rows at random from it; "per XCD": workgroup w (XCD
    // w % 8 under the round-robin dispatch) draws from its own eighth of the buffer
    struct Case { int rows; int per_xcd; const char* what; };
    const Case cases[] = {{1024, 0, "0.4 MB working set (fits the 32 KB L1s poorly, the L2 well)"},
                          {4096, 0, "1.5 MB working set (every XCD's L2 holds all of it)"},
                          {10000, 0, "3.75 MB working set (just inside a 4 MB L2)"},
                          {80000, 0, "30 MB working set (Infinity Cache)"},
                          {80000, 1, "30 MB, each XCD draws from its own eighth (3.75 MB per L2)"},
                          {40000, 1, "15 MB, each XCD draws from its own eighth (1.9 MB per L2)"}};
    for (const Case& cs : cases) {
        uint32_t s = 12345u;
        for (int w = 0; w < waves; ++w) {
            const int wg = w / 4, xcd = wg % 8;
            const int lo = cs.per_xcd ? xcd * (cs.rows / 8) : 0, span = cs.per_xcd ? cs.rows / 8 : cs.rows;
            for (int u = 0; u < units; ++u)
                for (int r = 0; r < 32; ++r) {
                    s = s * 1664525u + 1013904223u;
                    const int row = lo + (int)((s >> 8) % (unsigned)span);
                    // the three chunks of the same rows in consecutive units, as the conv walks them
                    h[((size_t)w * units + u) * 32 + r] = (u % chunks) ? h[((size_t)w * units + u - (u % chunks)) * 32 + r] : row;
                }
        }
        CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("%s\n", cs.what);
        if (run<0, 2>("MFMA 32x32 layout (lane = row)", buf, idx, units, row_bytes, chunks, out, wgs, ghz, cus)) return 1;
        if (run<0, 3>("MFMA 32x32 layout (lane = row)", buf, idx, units, row_bytes, chunks, out, wgs, ghz, cus)) return 1;
        if (run<2, 2>("MFMA 16x16 layout (4 lanes / 64 B)", buf, idx, units, row_bytes, chunks, out, wgs, ghz, cus)) return 1;
        if (run<1, 2>("line-coalesced (8 lanes per chunk)", buf, idx, units, row_bytes, chunks, out, wgs, ghz, cus)) return 1;
        if (run<1, 3>("line-coalesced (8 lanes per chunk)", buf, idx, units, row_bytes, chunks, out, wgs, ghz, cus)) return 1;
    }
    return 0;
}
