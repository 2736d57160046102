// Sparse-voxel coordinate manager for gfx950: coordinate hash, stride-2 coordinate sets and
// kernel maps (neighbour tables).
//
// Supplies what the reference gets from MinkowskiEngine's coordinate manager [ME-ext]
// (README.md:53; call sites train_joint.py:250 `ME.SparseTensor`, utils/minkunet.py:53-107
// conv / strided conv / transposed conv) for exactly the maps one MinkUNet34C forward needs
// (SURVEY.md 3.4): k5@ts1, k3@ts{1,2,4,8,16}, k2s2@{1->2,2->4,4->8,8->16}.
//
// Design: one open-addressing hash (64-bit packed key -> row index, linear probing, table
// >= 2x entries, L2-resident at these sizes) per coordinate set.  Coarse sets are ordered by
// first appearance of a child (deterministic, like a sequential insert).  All five levels are
// built back to back with the level sizes kept ON THE DEVICE; the host reads the four counts
// with a single copy at the end (one sync per scene).
#include "cv_common.h"

namespace {

constexpr unsigned long long EMPTY_KEY = ~0ull;

__host__ __device__ __forceinline__ unsigned long long pack_key(int b, int x, int y, int z) {
    return ((unsigned long long)(unsigned)(b & 0xffff) << 48) |
           ((unsigned long long)(unsigned)((x + 32768) & 0xffff) << 32) |
           ((unsigned long long)(unsigned)((y + 32768) & 0xffff) << 16) |
           (unsigned long long)(unsigned)((z + 32768) & 0xffff);
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

__device__ __forceinline__ int floor_div(int a, int b) {   // b > 0
    const int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

// insert key -> min(row) ; returns the slot
__device__ __forceinline__ long long table_insert_min(unsigned long long* keys, int* vals,
                                                      long long mask, unsigned long long key, int row) {
    long long slot = (long long)(mix64(key) & (unsigned long long)mask);
    while (true) {
        const unsigned long long prev = atomicCAS(&keys[slot], EMPTY_KEY, key);
        if (prev == EMPTY_KEY || prev == key) {
            atomicMin(&vals[slot], row);
            return slot;
        }
        slot = (slot + 1) & mask;
    }
}

__device__ __forceinline__ long long table_find(const unsigned long long* keys, long long mask,
                                                unsigned long long key) {
    long long slot = (long long)(mix64(key) & (unsigned long long)mask);
    while (true) {
        const unsigned long long k = keys[slot];
        if (k == key) return slot;
        if (k == EMPTY_KEY) return -1;
        slot = (slot + 1) & mask;
    }
}

__global__ __launch_bounds__(256) void table_clear(unsigned long long* keys, int* vals, long long cap) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < cap; i += (long long)gridDim.x * 256) {
        keys[i] = EMPTY_KEY;
        vals[i] = 0x7fffffff;
    }
}

// level 1: key(coord i) -> i ; duplicates are counted (the reference feeds unique coordinates,
// utils/dataloader.py:197-204)
__global__ __launch_bounds__(256) void insert_rows(const int* __restrict__ coords, const int* n_ptr,
                                                   unsigned long long* keys, int* vals, long long mask,
                                                   int* dup_count) {
    const int n = *n_ptr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        // pack_key keeps 16 bits per field: a coordinate whose neighbour lookups (up to +-2 * 16 at the coarsest
        // level, rounded up to 64) leave the window, or a batch index beyond 16 bits, would alias another voxel
        const int lo = -32768 + 64, hi = 32767 - 64;
        if ((unsigned)c.x > 0xffffu || c.y < lo || c.y > hi || c.z < lo || c.z > hi || c.w < lo || c.w > hi)
            atomicAdd(dup_count + 1, 1);
        const long long slot = table_insert_min(keys, vals, mask, pack_key(c.x, c.y, c.z, c.w), i);
        (void)slot;
    }
}

__global__ __launch_bounds__(256) void count_dups(const int* __restrict__ coords, const int* n_ptr,
                                                  const unsigned long long* keys, const int* vals,
                                                  long long mask, int* dup_count) {
    const int n = *n_ptr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        const long long slot = table_find(keys, mask, pack_key(c.x, c.y, c.z, c.w));
        if (slot < 0 || vals[slot] != i) atomicAdd(dup_count, 1);
    }
}

// coarse key of every fine row inserted with min(fine row)
__global__ __launch_bounds__(256) void insert_coarse(const int* __restrict__ coords, const int* n_ptr,
                                                     int stride2, unsigned long long* keys, int* vals,
                                                     long long mask, long long* slot_of_row) {
    const int n = *n_ptr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        const int x = floor_div(c.y, stride2) * stride2, y = floor_div(c.z, stride2) * stride2,
                  z = floor_div(c.w, stride2) * stride2;
        slot_of_row[i] = table_insert_min(keys, vals, mask, pack_key(c.x, x, y, z), i);
    }
}

__global__ __launch_bounds__(256) void flag_first(const int* n_ptr, const int* vals,
                                                  const long long* slot_of_row, int* flag) {
    const int n = *n_ptr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        flag[i] = vals[slot_of_row[i]] == i ? 1 : 0;
}

// exclusive scan of the 0/1 flags in three short launches (a single-workgroup scan of 80k flags took
// 37 us): per-block sums -> scan of the block sums -> per-block scan + offset.
constexpr int SCAN_PER = 8, SCAN_BLOCK = 1024 * SCAN_PER;

__device__ __forceinline__ int block_exclusive_scan(int v, int* s /*[1024]*/) {
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    return s[threadIdx.x] - v;
}

__global__ __launch_bounds__(1024) void scan_block_sums(const int* __restrict__ flag, const int* n_ptr,
                                                        int* __restrict__ bsum) {
    __shared__ int s[1024];
    const int n = *n_ptr;
    const int b0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER;
    int sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER; ++j) sum += (b0 + j < n) ? flag[b0 + j] : 0;
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(1024) void scan_block_offsets(int* __restrict__ bsum, int nblocks, int* total_out) {
    __shared__ int s[1024];
    const int v = threadIdx.x < nblocks ? bsum[threadIdx.x] : 0;
    const int ex = block_exclusive_scan(v, s);
    if (threadIdx.x < nblocks) bsum[threadIdx.x] = ex;
    if (threadIdx.x == 1023) *total_out = ex + v;
}

__global__ __launch_bounds__(1024) void scan_apply(const int* __restrict__ flag, const int* n_ptr,
                                                   const int* __restrict__ boff, int* __restrict__ rank) {
    __shared__ int s[1024];
    const int n = *n_ptr;
    const int b0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER;
    int v[SCAN_PER], sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER; ++j) { v[j] = (b0 + j < n) ? flag[b0 + j] : 0; sum += v[j]; }
    int run = block_exclusive_scan(sum, s) + boff[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_PER; ++j) {
        if (b0 + j < n) rank[b0 + j] = run;
        run += v[j];
    }
}

__global__ __launch_bounds__(256) void emit_coarse(const int* __restrict__ coords, const int* n_ptr,
                                                   int stride2, const int* flag, const int* rank,
                                                   const long long* slot_of_row, int* vals,
                                                   int* __restrict__ out_coords) {
    const int n = *n_ptr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (!flag[i]) continue;
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        int4 o;
        o.x = c.x;
        o.y = floor_div(c.y, stride2) * stride2;
        o.z = floor_div(c.z, stride2) * stride2;
        o.w = floor_div(c.w, stride2) * stride2;
        const int r = rank[i];
        reinterpret_cast<int4*>(out_coords)[r] = o;
        vals[slot_of_row[i]] = r;     // table now maps coarse key -> compact coarse row
    }
}

// nbr[u][j] = row of (out_coord[u] + offset_j * ts) in the input set, or -1.  Offset index j runs
// with the first spatial axis fastest (oracle/sparse_oracle.py kernel_offsets); odd kernels are
// centred, even kernels start at 0.
__global__ __launch_bounds__(256) void build_kernel_map(const int* __restrict__ out_coords,
                                                        long long n_out,
                                                        const unsigned long long* __restrict__ keys,
                                                        const int* __restrict__ vals, long long mask,
                                                        int k, int ts, int* __restrict__ nbr) {
    const int K = k * k * k;
    const int lo = (k & 1) ? -(k / 2) : 0;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < n_out * K; t += (long long)gridDim.x * 256) {
        const long long u = t / K;
        const int j = (int)(t - u * K);
        const int ox = lo + j % k, oy = lo + (j / k) % k, oz = lo + j / (k * k);
        const int4 c = reinterpret_cast<const int4*>(out_coords)[u];
        const long long slot = table_find(keys, mask, pack_key(c.x, c.y + ox * ts, c.z + oy * ts, c.w + oz * ts));
        nbr[t] = slot >= 0 ? vals[slot] : -1;
    }
}

// transposed k2s2 map from the strided map: up[f] = {coarse row, octant}
__global__ __launch_bounds__(256) void build_up_map(const int* __restrict__ nbr_down,
                                                    long long n_coarse, int* __restrict__ up_nbr) {
    const long long n = n_coarse * 8;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        const int f = nbr_down[t];
        if (f >= 0) up_nbr[(long long)f * 8 + (t & 7)] = (int)(t >> 3);
    }
}

__global__ void set_int(int* p, int v) { *p = v; }

__device__ __forceinline__ unsigned long long spread3(unsigned long long v) {   // 16 bits -> every 3rd bit
    v &= 0xffffull;
    v = (v | (v << 32)) & 0x00ff00000000ffffull;     // not needed for 16 bits, kept general to 21
    v = (v | (v << 16)) & 0x00ff0000ff0000ffull;
    v = (v | (v << 8)) & 0xf00f00f00f00f00full;
    v = (v | (v << 4)) & 0x30c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x9249249249249249ull;
    return v;
}

// Z-order key (batch in the top bits) so that consecutive rows are spatially compact
__global__ __launch_bounds__(256) void morton_keys(const int* __restrict__ coords, long long n,
                                                   long long* __restrict__ keys) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const unsigned long long m = spread3((unsigned)(c.y + 32768)) | (spread3((unsigned)(c.z + 32768)) << 1) |
                                 (spread3((unsigned)(c.w + 32768)) << 2);
    keys[i] = (long long)(((unsigned long long)(c.x & 0x7fff) << 48) | m);
}

int grid_for(long long n) { return (int)std::min<long long>((n + 255) / 256, 4096); }

}  // namespace

extern "C" {

long long cv_sp_table_capacity(long long n) {
    long long cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}

size_t cv_sp_levels_workspace_bytes(long long n) {
    // slot_of_row (8n) + flag (4n) + rank (4n) + counters
    return cv_align_up((size_t)n * 8, 256) + 2 * cv_align_up((size_t)n * 4, 256) + 4096 + 1024;
}

// Builds the coordinate sets of tensor strides 1,2,4,8,16 and their hash tables.
//   d_coords[L]   : int32 [cap_rows][4] (L = 0 is the caller's input set, rows n)
//   d_keys/vals[L]: hash tables of cv_sp_table_capacity(n) slots each
//   d_counts      : int32[8] device; [L] = rows at level L, [5] = duplicate count at level 0, [6] = rows whose
//                   coordinates are outside the 16-bit key window (|c| <= 32703, batch < 65536)
// h_counts receives the same 8 ints (one synchronisation).
int cv_sp_build_levels(int32_t* const* d_coords, unsigned long long* const* d_keys,
                       int32_t* const* d_vals, long long n, long long cap, int num_levels,
                       int32_t* d_counts, int32_t* h_counts, void* d_ws, size_t ws_bytes, void* stream) {
    CV_REQUIRE(d_coords && d_keys && d_vals && d_counts && d_ws, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n > 0 && n < (1ll << 30), CV_EINVAL, "bad row count %lld", n);
    CV_REQUIRE(num_levels >= 1 && num_levels <= 5, CV_EINVAL, "num_levels must be 1..5");
    CV_REQUIRE(cap >= 2 * n && (cap & (cap - 1)) == 0, CV_EINVAL, "table capacity must be a power of two >= 2n");
    CV_REQUIRE(ws_bytes >= cv_sp_levels_workspace_bytes(n), CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    CvCarver cv(d_ws);
    long long* slot_of_row = cv.take<long long>(n);
    int* flag = cv.take<int>(n);
    int* rank = cv.take<int>(n);
    int* bsum = cv.take<int>(1024);
    const int nsb = (int)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    CV_REQUIRE(nsb <= 1024, CV_EINVAL, "coordinate set too large for the scan (%lld rows)", n);
    const int g = grid_for(n);
    CV_HIP_CHECK(hipMemsetAsync(d_counts, 0, sizeof(int) * 8, st));
    set_int<<<1, 1, 0, st>>>(d_counts, (int)n);
    CV_LAUNCH_CHECK();
    table_clear<<<grid_for(cap), 256, 0, st>>>(d_keys[0], d_vals[0], cap);
    CV_LAUNCH_CHECK();
    insert_rows<<<g, 256, 0, st>>>(d_coords[0], d_counts, d_keys[0], d_vals[0], cap - 1, d_counts + 5);
    CV_LAUNCH_CHECK();
    count_dups<<<g, 256, 0, st>>>(d_coords[0], d_counts, d_keys[0], d_vals[0], cap - 1, d_counts + 5);
    CV_LAUNCH_CHECK();
    for (int L = 1; L < num_levels; ++L) {
        const int stride2 = 1 << L;
        table_clear<<<grid_for(cap), 256, 0, st>>>(d_keys[L], d_vals[L], cap);
        CV_LAUNCH_CHECK();
        insert_coarse<<<g, 256, 0, st>>>(d_coords[L - 1], d_counts + L - 1, stride2, d_keys[L], d_vals[L],
                                         cap - 1, slot_of_row);
        CV_LAUNCH_CHECK();
        flag_first<<<g, 256, 0, st>>>(d_counts + L - 1, d_vals[L], slot_of_row, flag);
        CV_LAUNCH_CHECK();
        scan_block_sums<<<nsb, 1024, 0, st>>>(flag, d_counts + L - 1, bsum);
        CV_LAUNCH_CHECK();
        scan_block_offsets<<<1, 1024, 0, st>>>(bsum, nsb, d_counts + L);
        CV_LAUNCH_CHECK();
        scan_apply<<<nsb, 1024, 0, st>>>(flag, d_counts + L - 1, bsum, rank);
        CV_LAUNCH_CHECK();
        emit_coarse<<<g, 256, 0, st>>>(d_coords[L - 1], d_counts + L - 1, stride2, flag, rank,
                                       slot_of_row, d_vals[L], d_coords[L]);
        CV_LAUNCH_CHECK();
    }
    if (h_counts) {
        CV_HIP_CHECK(hipMemcpyAsync(h_counts, d_counts, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
        CV_HIP_CHECK(hipStreamSynchronize(st));
    }
    return CV_OK;
}

// Z-order sort keys of a coordinate set (the fused network runs on spatially sorted rows so that
// 32-row wave tiles are compact and whole kernel offsets can be skipped).  Asynchronous.
int cv_sp_morton_keys(const int32_t* d_coords, long long n, long long* d_keys, void* stream) {
    CV_REQUIRE(d_coords && d_keys && n > 0, CV_EINVAL, "bad morton arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    morton_keys<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(d_coords, n, d_keys);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

// Kernel map of a k^3 kernel: out set (rows n_out) looked up in the input set's table.
// d_nbr: int32 [n_out][k^3].  ts = tensor stride of the INPUT set (offset unit).
int cv_sp_kernel_map(const int32_t* d_out_coords, long long n_out, const unsigned long long* d_keys,
                     const int32_t* d_vals, long long cap, int k, int ts, int32_t* d_nbr, void* stream) {
    CV_REQUIRE(d_out_coords && d_keys && d_vals && d_nbr, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n_out > 0 && k >= 1 && k <= 7 && ts >= 1, CV_EINVAL, "bad kernel map arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    build_kernel_map<<<grid_for(n_out * k * k * k), 256, 0, st>>>(d_out_coords, n_out, d_keys, d_vals,
                                                                cap - 1, k, ts, d_nbr);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

// Transposed k2s2 map: d_up[n_fine][8] = coarse row in the octant column that generates the
// fine row, -1 elsewhere (so the generic conv kernel evaluates out[f] = W_oct^T x[parent]).
int cv_sp_up_map(const int32_t* d_nbr_down, long long n_coarse, long long n_fine, int32_t* d_up,
                 void* stream) {
    CV_REQUIRE(d_nbr_down && d_up, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n_coarse > 0 && n_fine > 0, CV_EINVAL, "bad sizes");
    hipStream_t st = static_cast<hipStream_t>(stream);
    CV_HIP_CHECK(hipMemsetAsync(d_up, 0xff, sizeof(int) * n_fine * 8, st));
    build_up_map<<<grid_for(n_coarse * 8), 256, 0, st>>>(d_nbr_down, n_coarse, d_up);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

}  // extern "C"
