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

// ---- all coordinate levels from the level-0 rows in five launches (round 3; the level-by-level build took 19) -------
// A coarse set is ordered by the first appearance of a child in the next finer set.  By induction that is the order of
// the SMALLEST LEVEL-0 ROW among a voxel's descendants (the finer set is itself in that order, so the first child is the
// one holding the smallest descendant), so every level can be built from the level-0 rows directly:
//   insert_all       key_L(row i) -> min(i) into the table of every level L (one pass over the rows; a lane whose
//                    level-L key equals its left neighbour's cannot hold the minimum and skips the atomics - the rows
//                    of the fused network arrive spatially sorted, so most lanes do)
//   flag_levels      row i is the first descendant of its level-L voxel iff vals_L[slot] == i (+ the duplicate check
//                    of level 0 as one more job of the same launch)
//   emit_levels      rank of a flagged row = its coarse row: coordinates out, table value := coarse row
struct LevelsDev { unsigned long long* keys[5]; int* vals[5]; int* coords[5]; int n_levels; };

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

// ---- occupancy bitmap over the bounding box (cv_sp_occupancy_bitmap) ----
struct BitBox { int mn[3], d[3], nb; bool ok; };
__device__ __forceinline__ BitBox bitbox(const int* __restrict__ mm) {
    BitBox b;
    long long cells = 1;
    bool sane = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // extents in 64 bits: coordinates far outside the key window must not wrap the product into the accepted range
        const long long d = -(long long)mm[3 + k] - (long long)mm[k] + 1;
        sane = sane && d > 0 && d <= 65536;
        b.mn[k] = mm[k];
        b.d[k] = sane ? (int)d : 0;
        cells *= sane ? d : 0;
    }
    const long long nb = -(long long)mm[6] + 1;
    sane = sane && nb > 0 && nb <= 65536;
    b.nb = sane ? (int)nb : 0;
    cells = sane ? cells * nb : 0;
    b.ok = sane && cells > 0 && cells <= CV_BITMAP_WORDS * 32 && mm[7] == 1;      // mm[7] == 1: no negative batch index seen
    return b;
}
// bit of (batch, x, y, z), or -1 outside the box
__device__ __forceinline__ long long bitbox_index(const BitBox& b, int bi, int x, int y, int z) {
    const int ux = x - b.mn[0], uy = y - b.mn[1], uz = z - b.mn[2];
    if ((unsigned)ux >= (unsigned)b.d[0] || (unsigned)uy >= (unsigned)b.d[1] || (unsigned)uz >= (unsigned)b.d[2] ||
        (unsigned)bi >= (unsigned)b.nb) return -1;
    return (((long long)bi * b.d[0] + ux) * b.d[1] + uy) * b.d[2] + uz;
}

__global__ __launch_bounds__(256) void insert_all(const int* __restrict__ coords, int n, const LevelsDev t, long long mask,
                                                  int* __restrict__ slots /*[n_levels - 1][n]*/, int* dup_count,
                                                  int* __restrict__ mm = nullptr, unsigned* __restrict__ bits = nullptr) {
    const int lane = threadIdx.x & 63;
    const int n_pad = (n + 255) / 256 * 256;           // whole waves stay in the loop (the shuffles need them)
    // bits (cv_sp_scene_plan): this pass also sets the occupancy bits of the level-0 rows over the sort's bounding box (what
    // bitmap_set did in a launch of its own; the launch in front cleared the words and set mm[7] = 1 = "may be trusted")
    BitBox bb;
    bb.ok = false;
    if (bits) bb = bitbox(mm);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_pad; i += gridDim.x * 256) {
        const bool have = i < n;
        int4 c = make_int4(0, 0, 0, 0);
        if (have) {
            c = reinterpret_cast<const int4*>(coords)[i];
            // pack_key keeps 16 bits per field: a coordinate whose neighbour lookups (up to +-2 * 16 at the coarsest
            // level, rounded up to 64) leave the window, or a batch index beyond 16 bits, would alias another voxel
            const int lo = -32768 + 64, hi = 32767 - 64;
            if ((unsigned)c.x > 0xffffu || c.y < lo || c.y > hi || c.z < lo || c.z > hi || c.w < lo || c.w > hi)
                atomicAdd(dup_count + 1, 1);
            table_insert_min(t.keys[0], t.vals[0], mask, pack_key(c.x, c.y, c.z, c.w), i);
            if (bb.ok) {
                const long long bit = bitbox_index(bb, c.x, c.y, c.z, c.w);
                if (bit >= 0) atomicOr(&bits[bit >> 5], 1u << (bit & 31));
                else mm[7] = 0;                // a row outside its own bounds (negative batch index): do not trust the bitmap
            }
        }
        for (int L = 1; L < t.n_levels; ++L) {
            const int m = ~((1 << L) - 1);             // floor(c / 2^L) * 2^L in two's complement
            const unsigned long long key = have ? pack_key(c.x, c.y & m, c.z & m, c.w & m) : EMPTY_KEY;
            const unsigned long long left = __shfl_up(key, 1);
            int slot = -1;
            if (have && (lane == 0 || left != key)) slot = (int)table_insert_min(t.keys[L], t.vals[L], mask, key, i);
            if (have) slots[(long long)(L - 1) * n + i] = slot;
        }
    }
}

// blockIdx.y < n_levels - 1: flags of level y + 1 (kept as slot >= 0) and their block sums; blockIdx.y == n_levels - 1:
// every level-0 row must find ITSELF in the level-0 table (duplicates are counted: the reference feeds unique
// coordinates, utils/dataloader.py:197-204)
template <int SCAN_PER>
__global__ __launch_bounds__(1024) void flag_levels(const int* __restrict__ coords, int n, const LevelsDev t, long long mask,
                                                    int* __restrict__ slots, int* __restrict__ bsum /*[4][1024]*/,
                                                    int* dup_count) {
    __shared__ int s[1024];
    const int b0 = (blockIdx.x * 1024 + threadIdx.x) * SCAN_PER;
    if ((int)blockIdx.y == t.n_levels - 1) {
#pragma unroll
        for (int j = 0; j < SCAN_PER; ++j)
            if (b0 + j < n) {
                const int4 c = reinterpret_cast<const int4*>(coords)[b0 + j];
                const long long slot = table_find(t.keys[0], mask, pack_key(c.x, c.y, c.z, c.w));
                if (slot < 0 || t.vals[0][slot] != b0 + j) atomicAdd(dup_count, 1);
            }
        return;
    }
    const int L = blockIdx.y + 1;
    int* sl = slots + (long long)blockIdx.y * n;
    int sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER; ++j)
        if (b0 + j < n) {
            const int slot = sl[b0 + j];
            if (slot >= 0) {
                if (t.vals[L][slot] == b0 + j) sum += 1;
                else sl[b0 + j] = -1;
            }
        }
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.y * 1024 + blockIdx.x] = s[0];
}

// (round 5: the one-block-per-level scan launch between flag_levels and emit_levels is gone - a block adds up the sums of the
// blocks before it itself (at most 1024 words, one per thread), the last block of a level also writes the level's row count)
template <int SCAN_PER>
__global__ __launch_bounds__(1024) void emit_levels(const int* __restrict__ coords, int n, const LevelsDev t,
                                                    const int* __restrict__ slots, const int* __restrict__ bsum,
                                                    int* __restrict__ counts) {
    __shared__ int s[1024];
    __shared__ int wsum[16];
    const int L = blockIdx.y + 1;
    const int* sl = slots + (long long)blockIdx.y * n;
    const int b0 = (blockIdx.x * 1024 + threadIdx.x) * SCAN_PER;
    int v[SCAN_PER], sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER; ++j) { v[j] = (b0 + j < n) ? sl[b0 + j] : -1; sum += v[j] >= 0; }
    int before = (int)threadIdx.x < (int)blockIdx.x ? bsum[blockIdx.y * 1024 + threadIdx.x] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = before;
    __syncthreads();
    int boff = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) boff += wsum[w];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) counts[L] = boff + bsum[blockIdx.y * 1024 + blockIdx.x];
    int run = block_exclusive_scan(sum, s) + boff;
    const int m = ~((1 << L) - 1);
#pragma unroll
    for (int j = 0; j < SCAN_PER; ++j) {
        if (v[j] >= 0) {
            const int4 c = reinterpret_cast<const int4*>(coords)[b0 + j];
            reinterpret_cast<int4*>(t.coords[L])[run] = make_int4(c.x, c.y & m, c.z & m, c.w & m);
            t.vals[L][v[j]] = run;          // table now maps coarse key -> compact coarse row
            ++run;
        }
    }
}

struct TablesDev { unsigned long long* keys[5]; int* vals[5]; int n; };

// every level's table cleared by one launch; also the counters: counts[0] = n, the rest 0; and an optional extra range
// of words set to zero (cv_sp_scene_plan: the histogram scratch of the mask orders, so that they need no fill launches)
__global__ __launch_bounds__(256) void table_clear_all(const TablesDev t, long long cap, int* counts, int n_rows,
                                                       int* __restrict__ zero_words, long long n_zero, int* set_one) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < cap * t.n; i += (long long)gridDim.x * 256) {
        const int L = (int)(i / cap);
        const long long k = i - (long long)L * cap;
        t.keys[L][k] = EMPTY_KEY;
        t.vals[L][k] = 0x7fffffff;
    }
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n_zero; i += (long long)gridDim.x * 256) zero_words[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 8) counts[threadIdx.x] = threadIdx.x == 0 ? n_rows : 0;
    if (set_one && blockIdx.x == 0 && threadIdx.x == 0) *set_one = 1;
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

// ---- occupancy bitmap over the bounding box (cv_sp_occupancy_bitmap; BitBox is defined in front of insert_all) ----
__global__ __launch_bounds__(256) void bitmap_clear(const int* __restrict__ coords, long long n, int* __restrict__ mm,
                                                    unsigned* __restrict__ bits) {
    // mm[7] = "the bitmap may be trusted": set here, taken back by bitmap_set when a row falls outside the bounds (a negative
    // batch index: the sort tracks the largest batch index only and such inputs fail the key-window check anyway)
    if (blockIdx.x == 0 && threadIdx.x == 0) mm[7] = 1;
    long long cells = 1;
    for (int k = 0; k < 3; ++k) {
        const long long d = -(long long)mm[3 + k] - (long long)mm[k] + 1;
        cells = (d > 0 && d <= 65536 && cells > 0) ? cells * d : 0;
    }
    const long long nb = -(long long)mm[6] + 1;
    cells = (nb > 0 && nb <= 65536) ? cells * nb : 0;
    if (cells <= 0 || cells > CV_BITMAP_WORDS * 32) return;
    const long long words = (cells + 31) / 32;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < words; i += (long long)gridDim.x * 256) bits[i] = 0u;
    (void)coords; (void)n;
}
__global__ __launch_bounds__(256) void bitmap_set(const int* __restrict__ coords, long long n, int* __restrict__ mm,
                                                  unsigned* __restrict__ bits) {
    const BitBox b = bitbox(mm);
    if (!b.ok) return;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        const long long bit = bitbox_index(b, c.x, c.y, c.z, c.w);
        if (bit >= 0) atomicOr(&bits[bit >> 5], 1u << (bit & 31));
        else mm[7] = 0;                    // a row outside its own bounds (negative batch index): do not trust the bitmap
    }
}

// all kernel maps of a scene in one launch: blockIdx.x ranges per job
struct MapJobsDev {
    CvMapJob j[CV_MAX_MAP_JOBS];
    int block_begin[CV_MAX_MAP_JOBS + 1];
    int n;
};

// one job's lookups; KC = k^3 as a compile-time constant (the index split t -> (row, offset) is a constant division and the
// 32-bit form is used whenever the map has fewer than 2^31 entries: the generic 64-bit division by a run-time K was most of
// what a lookup cost once the bitmap answers the misses)
template <int KK>
__device__ __forceinline__ void map_job(const CvMapJob& jb, int nblk, int blk) {
    const int k = KK > 0 ? (KK == 125 ? 5 : KK == 27 ? 3 : 2) : jb.k;
    const int K = KK > 0 ? KK : k * k * k, ts = jb.ts;
    const int lo = (k & 1) ? -(k / 2) : 0;
    const long long mask = jb.cap - 1;
    BitBox bb;
    bb.ok = false;
    if (jb.bitmap && ts == 1) bb = bitbox(jb.bbox);
    const long long total = jb.n_out * K;
    const bool small = total < (1ll << 31);
    for (long long t = blk * 256ll + threadIdx.x; t < total; t += (long long)nblk * 256) {
        long long u;
        int j;
        if (small) { const unsigned tu = (unsigned)t, uu = tu / (unsigned)K; u = uu; j = (int)(tu - uu * (unsigned)K); }
        else { u = t / K; j = (int)(t - u * K); }
        const int ox = lo + j % k, oy = lo + (j / k) % k, oz = lo + j / (k * k);
        const int4 c = reinterpret_cast<const int4*>(jb.out_coords)[u];
        int r = -1;
        bool probe = true;
        if (bb.ok) {                       // most lookups are misses: one bit answers them
            const long long bit = bitbox_index(bb, c.x, c.y + ox, c.z + oy, c.w + oz);
            probe = bit >= 0 && ((jb.bitmap[bit >> 5] >> (bit & 31)) & 1u);
        }
        if (probe) {
            const long long slot = table_find(jb.keys, mask, pack_key(c.x, c.y + ox * ts, c.z + oy * ts, c.w + oz * ts));
            r = slot >= 0 ? jb.vals[slot] : -1;
            if (jb.compose && r >= 0) r = jb.compose[r];
        }
        jb.nbr[t] = r;
    }
}

// transposed k2s2 map by lookup: fine row f -> its parent's row in the octant column (no pre-fill, no down map needed)
__device__ __forceinline__ void up_job(const CvMapJob& jb, int nblk, int blk) {
    const long long mask = jb.cap - 1;
    const int ts = jb.ts, m = ~(2 * ts - 1);
    for (long long f = blk * 256ll + threadIdx.x; f < jb.n_out; f += (long long)nblk * 256) {
        const int4 c = reinterpret_cast<const int4*>(jb.out_coords)[f];
        const int px = c.y & m, py = c.z & m, pz = c.w & m;
        const long long slot = table_find(jb.keys, mask, pack_key(c.x, px, py, pz));
        const int parent = slot >= 0 ? jb.vals[slot] : -1;
        const int oct = ((c.y - px) / ts) + 2 * ((c.z - py) / ts) + 4 * ((c.w - pz) / ts);
        int4 lo = make_int4(-1, -1, -1, -1), hi = lo;
        switch (oct) {
            case 0: lo.x = parent; break; case 1: lo.y = parent; break; case 2: lo.z = parent; break; case 3: lo.w = parent; break;
            case 4: hi.x = parent; break; case 5: hi.y = parent; break; case 6: hi.z = parent; break; default: hi.w = parent; break;
        }
        reinterpret_cast<int4*>(jb.nbr)[2 * f] = lo;
        reinterpret_cast<int4*>(jb.nbr)[2 * f + 1] = hi;
    }
}

__global__ __launch_bounds__(256) void build_kernel_maps(const MapJobsDev jobs) {
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.block_begin[ji + 1]) ++ji;
    const CvMapJob& jb = jobs.j[ji];
    const int nblk = jobs.block_begin[ji + 1] - jobs.block_begin[ji], blk = blockIdx.x - jobs.block_begin[ji];
    if (jb.up) { up_job(jb, nblk, blk); return; }
    switch (jb.k) {
        case 5: map_job<125>(jb, nblk, blk); break;
        case 3: map_job<27>(jb, nblk, blk); break;
        case 2: map_job<8>(jb, nblk, blk); break;
        default: map_job<0>(jb, nblk, blk); break;
    }
}

struct UpJobsDev {
    CvUpJob j[4];
    int block_begin[5];
    int n;
};

__global__ __launch_bounds__(256) void build_up_maps(const UpJobsDev jobs) {
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.block_begin[ji + 1]) ++ji;
    const CvUpJob& jb = jobs.j[ji];
    const int nblk = jobs.block_begin[ji + 1] - jobs.block_begin[ji], blk = blockIdx.x - jobs.block_begin[ji];
    const long long n = jb.n_coarse * 8;
    for (long long t = blk * 256ll + threadIdx.x; t < n; t += (long long)nblk * 256) {
        const int f = jb.nbr_down[t];
        if (f >= 0) jb.up[(long long)f * 8 + (t & 7)] = (int)(t >> 3);
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

// ---------------------------------------------------------------------------------------------------------------
// Spatial row order of a scene: stable LSD radix sort (three 9-bit digits) on
//   key = batch (9 bits) | Morton code of ((c - min) >> shift) (6 bits per axis),
// shift = the smallest that brings every axis' extent under 64.  Rows of one 2^shift cube stay in the caller's order,
// cubes run in Z-order, scenes of a batch one after the other: what the gathers and the 32-row MFMA tiles need (and
// deterministic: the sort is stable).  Replaces a 64-bit device-wide sort + gathers (13 launches, ~170 us per scene).
constexpr int SORT_BITS = 9, SORT_BINS = 1 << SORT_BITS, SORT_ROWS = 2048, SORT_T = 256;   // (1024 rows per block measured: scatter 28 us per pass instead of 23 - every block sums a longer table)

// mm[0..2] = min c, mm[3..5] = min(-c) (i.e. -max), mm[6] = min(-batch); initialised to 0x7f7f7f7f by a fill
__global__ __launch_bounds__(256) void sort_minmax(const int* __restrict__ coords, long long n, int* __restrict__ mm) {
    __shared__ int s[7];
    if (threadIdx.x < 7) s[threadIdx.x] = 0x7f7f7f7f;
    __syncthreads();
    int v[7] = {0x7f7f7f7f, 0x7f7f7f7f, 0x7f7f7f7f, 0x7f7f7f7f, 0x7f7f7f7f, 0x7f7f7f7f, 0x7f7f7f7f};
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        v[0] = min(v[0], c.y); v[1] = min(v[1], c.z); v[2] = min(v[2], c.w);
        v[3] = min(v[3], -c.y); v[4] = min(v[4], -c.z); v[5] = min(v[5], -c.w);
        v[6] = min(v[6], -c.x);
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        int x = v[k];
        for (int off = 32; off > 0; off >>= 1) x = min(x, __shfl_xor(x, off));
        if ((threadIdx.x & 63) == 0) atomicMin(&s[k], x);
    }
    __syncthreads();
    if (threadIdx.x < 7) atomicMin(&mm[threadIdx.x], s[threadIdx.x]);
}

__device__ __forceinline__ unsigned sort_key_of(int4 c, const int* __restrict__ mm) {
    const int ex = max(max(-mm[3] - mm[0], -mm[4] - mm[1]), -mm[5] - mm[2]);      // largest extent
    int shift = 0;
    while ((ex >> shift) >= 64) ++shift;
    const unsigned x = (unsigned)((c.y - mm[0]) >> shift) & 63u, y = (unsigned)((c.z - mm[1]) >> shift) & 63u,
                   z = (unsigned)((c.w - mm[2]) >> shift) & 63u;
    const unsigned m = (unsigned)(spread3(x) | (spread3(y) << 1) | (spread3(z) << 2));    // 18 bits
    const unsigned b = (unsigned)min(max(c.x, 0), SORT_BINS - 1);                        // batch index, clamped
    return (b << 18) | m;
}

// pass 0: keys + the per-block histogram of digit 0; later passes: histogram of digit `pass` of keys_in
__global__ __launch_bounds__(SORT_T) void sort_hist(const int* __restrict__ coords, const int* __restrict__ mm,
                                                    unsigned* __restrict__ keys, long long n, int pass,
                                                    int* __restrict__ hist /*[nblk][SORT_BINS]*/) {
    __shared__ int lh[SORT_BINS];
    if (pass == 2 && mm[6] == 0) return;              // one scene (batch index 0 everywhere): the third digit is constant
    for (int i = threadIdx.x; i < SORT_BINS; i += SORT_T) lh[i] = 0;
    __syncthreads();
    const long long base = blockIdx.x * (long long)SORT_ROWS;
    for (int r = 0; r < SORT_ROWS / SORT_T; ++r) {
        const long long i = base + r * SORT_T + threadIdx.x;
        if (i < n) {
            unsigned k;
            if (coords) { k = sort_key_of(reinterpret_cast<const int4*>(coords)[i], mm); keys[i] = k; }
            else k = keys[i];
            atomicAdd(&lh[(k >> (SORT_BITS * pass)) & (SORT_BINS - 1)], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SORT_BINS; i += SORT_T) hist[(long long)blockIdx.x * SORT_BINS + i] = lh[i];
}

// stable scatter of one digit.  vals_in == nullptr: the value of row i is i (first pass).
// Last pass (coords != nullptr): instead of keys_out / vals_out it writes perm[pos] = original row,
// inv[original row] = pos and sorted[pos] = coords[original row].
// A wave owns SORT_ROWS / 4 consecutive rows of the block (8 rounds of 64): it ranks them against its OWN running
// counters in LDS (leader lane per digit value, no workgroup barrier between the rounds), the four waves' counters are
// then chained per bin.  Three workgroup barriers after the clear (the first version took 48: a turn per wave and round, and a
// 256-wide scan with two barriers per step).
__global__ __launch_bounds__(SORT_T) void sort_scatter(const unsigned* __restrict__ keys_in, const int* __restrict__ vals_in,
                                                       long long n, int pass, const int* __restrict__ hist, int nblk,
                                                       unsigned* __restrict__ keys_out, int* __restrict__ vals_out,
                                                       const int* __restrict__ coords_in, int* __restrict__ perm,
                                                       int* __restrict__ inv, int* __restrict__ sorted,
                                                       const int* __restrict__ mm, int force_single = 0) {
    constexpr int NWAVE = SORT_T / 64, ROUNDS = SORT_ROWS / SORT_T, PER_T = SORT_BINS / SORT_T;
    __shared__ int wcnt[NWAVE][SORT_BINS];          // rows of (wave, bin); then the first output slot of (wave, bin)
    __shared__ int wave_tot[NWAVE];
    // a single scene (largest batch index 0) is sorted after two digits: pass 1 then writes the final outputs and
    // pass 2 has nothing to do (two launches that exit at once instead of 30 us of histogram + scatter)
    // (force_single: the caller says so - cv_detect_scene_f32's input is one scene - and does not even queue pass 2; rows with
    // another batch index would only lose their batch-major grouping, every table key carries the batch index)
    const bool single = force_single || mm[6] == 0;
    if (pass == 2 && single) return;
    const int* coords = (pass == 2 || (pass == 1 && single)) ? coords_in : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NWAVE * SORT_BINS; i += SORT_T) (&wcnt[0][0])[i] = 0;
    // totals and this block's prefix per bin (PER_T bins per thread, coalesced over the block-major histogram): requested
    // now, consumed after the ranking
    int tot[PER_T], pre[PER_T];
#pragma unroll
    for (int q = 0; q < PER_T; ++q) { tot[q] = 0; pre[q] = 0; }
    for (int b = 0; b < nblk; ++b) {
#pragma unroll
        for (int q = 0; q < PER_T; ++q) {
            const int h = hist[(long long)b * SORT_BINS + threadIdx.x * PER_T + q];
            tot[q] += h;
            if (b < (int)blockIdx.x) pre[q] += h;
        }
    }
    __syncthreads();                                 // counters cleared
    // ---- ranking: rows base + wave * 512 + r * 64 + lane
    const long long base = blockIdx.x * (long long)SORT_ROWS + wave * (SORT_ROWS / NWAVE);
    unsigned k[ROUNDS];
    int d[ROUNDS], local[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const long long i = base + r * 64 + lane;
        k[r] = i < n ? keys_in[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const long long i = base + r * 64 + lane;
        const bool have = i < n;
        d[r] = (int)((k[r] >> (SORT_BITS * pass)) & (SORT_BINS - 1));
        uint64_t peers = __ballot(have);             // lanes of this wave with the same digit
#pragma unroll
        for (int bit = 0; bit < SORT_BITS; ++bit) {
            const uint64_t bm = __ballot(have && ((d[r] >> bit) & 1));
            peers &= ((d[r] >> bit) & 1) ? bm : ~bm;
        }
        const int rank_w = __popcll(peers & ((1ull << lane) - 1ull)), cnt_w = __popcll(peers);
        const int leader = have ? (int)__ffsll((unsigned long long)peers) - 1 : lane;
        int off = 0;
        if (have && lane == leader) { off = wcnt[wave][d[r]]; wcnt[wave][d[r]] = off + cnt_w; }
        local[r] = __shfl(off, leader) + rank_w;     // rank among the wave's rows with this digit
    }
    // ---- exclusive scan of the bin totals (bins threadIdx.x * PER_T + q, in bin order): per wave, then over the waves
    int mine = 0;
#pragma unroll
    for (int q = 0; q < PER_T; ++q) mine += tot[q];
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();                                 // wave totals; every wave's counters are final
    int run = incl - mine;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) run += w < wave ? wave_tot[w] : 0;
    // first output slot of (wave, bin) = bin start + rows of the blocks before + rows of the waves before
    int c[PER_T][NWAVE];
#pragma unroll
    for (int q = 0; q < PER_T; ++q)
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) c[q][w] = wcnt[w][threadIdx.x * PER_T + q];
    // (a bin's counters are read and rewritten by the thread that owns the bin: no barrier in between)
#pragma unroll
    for (int q = 0; q < PER_T; ++q) {
        int slot = run + pre[q];
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) { wcnt[w][threadIdx.x * PER_T + q] = slot; slot += c[q][w]; }
        run += tot[q];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const long long i = base + r * 64 + lane;
        if (i < n) {
            const long long pos = wcnt[wave][d[r]] + local[r];
            const int v = vals_in ? vals_in[i] : (int)i;
            if (coords) {
                perm[pos] = v;
                inv[v] = (int)pos;
                reinterpret_cast<int4*>(sorted)[pos] = reinterpret_cast<const int4*>(coords)[v];
            } else {
                keys_out[pos] = k[r];
                vals_out[pos] = v;
            }
        }
    }
}

int grid_for(long long n) { return (int)std::min<long long>((n + 255) / 256, 4096); }

}  // namespace

int cv_sp_kernel_maps_batch(const CvMapJob* jobs, int n_jobs, void* stream) {
    CV_REQUIRE(jobs && n_jobs > 0 && n_jobs <= CV_MAX_MAP_JOBS, CV_EINVAL, "bad kernel map batch");
    MapJobsDev d;
    d.n = n_jobs;
    int total = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const CvMapJob& j = jobs[i];
        CV_REQUIRE(j.out_coords && j.keys && j.vals && j.nbr && j.n_out > 0 && j.k >= 1 && j.k <= 7 && j.ts >= 1,
                   CV_EINVAL, "bad kernel map job %d", i);
        d.j[i] = j;
        d.block_begin[i] = total;
        total += j.up ? grid_for(j.n_out) : grid_for(j.n_out * j.k * j.k * j.k);
    }
    d.block_begin[n_jobs] = total;
    build_kernel_maps<<<total, 256, 0, static_cast<hipStream_t>(stream)>>>(d);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_sp_occupancy_bitmap(const int32_t* d_coords, long long n, const int32_t* d_bbox, unsigned* d_bits, void* stream,
                           bool pre_cleared) {
    CV_REQUIRE(d_coords && d_bbox && d_bits && n > 0, CV_EINVAL, "bad bitmap arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int* mm = const_cast<int*>(d_bbox);
    if (!pre_cleared) {
        bitmap_clear<<<256, 256, 0, st>>>(d_coords, n, mm, d_bits);
        CV_LAUNCH_CHECK();
    }
    bitmap_set<<<grid_for(n), 256, 0, st>>>(d_coords, n, mm, d_bits);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

int cv_sp_up_maps_batch(const CvUpJob* jobs, int n_jobs, void* stream) {
    CV_REQUIRE(jobs && n_jobs > 0 && n_jobs <= 4, CV_EINVAL, "bad up map batch");
    UpJobsDev d;
    d.n = n_jobs;
    int total = 0;
    for (int i = 0; i < n_jobs; ++i) {
        CV_REQUIRE(jobs[i].nbr_down && jobs[i].up && jobs[i].n_coarse > 0, CV_EINVAL, "bad up map job %d", i);
        d.j[i] = jobs[i];
        d.block_begin[i] = total;
        total += grid_for(jobs[i].n_coarse * 8);
    }
    d.block_begin[n_jobs] = total;
    build_up_maps<<<total, 256, 0, static_cast<hipStream_t>(stream)>>>(d);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

extern "C" {

long long cv_sp_table_capacity(long long n) {
    long long cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}

size_t cv_sp_levels_workspace_bytes(long long n) {
    // table slot of every (coarse level, level-0 row) + block sums of the four flag scans
    return 4 * cv_align_up((size_t)n * 4, 256) + 4 * 4096 + 1024;
}

// Builds the coordinate sets of tensor strides 1,2,4,8,16 and their hash tables.
//   d_coords[L]   : int32 [cap_rows][4] (L = 0 is the caller's input set, rows n)
//   d_keys/vals[L]: hash tables of cv_sp_table_capacity(n) slots each
//   d_counts      : int32[8] device; [L] = rows at level L, [5] = duplicate count at level 0, [6] = rows whose
//                   coordinates are outside the 16-bit key window (|c| <= 32703, batch < 65536)
// h_counts receives the same 8 ints (one synchronisation).
}  // extern "C"

// (C++ linkage, cv_common.h) d_zero / n_zero: an optional range of words the first launch also clears
int cv_sp_build_levels_zero(int32_t* const* d_coords, unsigned long long* const* d_keys, int32_t* const* d_vals,
                            long long n, long long cap, int num_levels, int32_t* d_counts, int32_t* h_counts, void* d_ws,
                            size_t ws_bytes, int32_t* d_zero, long long n_zero, int32_t* d_set_one, void* stream,
                            uint32_t* d_bits) {
    CV_REQUIRE(d_coords && d_keys && d_vals && d_counts && d_ws, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n > 0 && n < (1ll << 30), CV_EINVAL, "bad row count %lld", n);
    CV_REQUIRE(num_levels >= 1 && num_levels <= 5, CV_EINVAL, "num_levels must be 1..5");
    CV_REQUIRE(cap >= 2 * n && (cap & (cap - 1)) == 0, CV_EINVAL, "table capacity must be a power of two >= 2n");
    CV_REQUIRE(ws_bytes >= cv_sp_levels_workspace_bytes(n), CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    CvCarver cv(d_ws);
    int* slots = cv.take<int>((size_t)4 * n);
    int* bsum = cv.take<int>(4 * 1024);
    const int scan_per = n <= (1ll << 20) ? 1 : 8;
    const int nsb = (int)((n + 1024ll * scan_per - 1) / (1024ll * scan_per));
    CV_REQUIRE(nsb <= 1024, CV_EINVAL, "coordinate set too large for the scan (%lld rows)", n);
    const int g = grid_for(n);
    TablesDev tabs;
    LevelsDev lv;
    tabs.n = lv.n_levels = num_levels;
    for (int L = 0; L < 5; ++L) {
        tabs.keys[L] = lv.keys[L] = L < num_levels ? d_keys[L] : nullptr;
        tabs.vals[L] = lv.vals[L] = L < num_levels ? d_vals[L] : nullptr;
        lv.coords[L] = L < num_levels ? d_coords[L] : nullptr;
    }
    table_clear_all<<<grid_for(cap * num_levels), 256, 0, st>>>(tabs, cap, d_counts, (int)n, d_zero, d_zero ? n_zero : 0, d_set_one);
    CV_LAUNCH_CHECK();
    // d_bits (with d_set_one = the eighth word of the sort's bounds): the occupancy bitmap is filled by the same pass
    insert_all<<<g, 256, 0, st>>>(d_coords[0], (int)n, lv, cap - 1, slots, d_counts + 5,
                                  (d_bits && d_set_one) ? d_set_one - 7 : nullptr, (d_bits && d_set_one) ? d_bits : nullptr);
    CV_LAUNCH_CHECK();
    if (scan_per == 1) flag_levels<1><<<dim3(nsb, num_levels), 1024, 0, st>>>(d_coords[0], (int)n, lv, cap - 1, slots, bsum, d_counts + 5);
    else flag_levels<8><<<dim3(nsb, num_levels), 1024, 0, st>>>(d_coords[0], (int)n, lv, cap - 1, slots, bsum, d_counts + 5);
    CV_LAUNCH_CHECK();
    if (num_levels > 1) {
        if (scan_per == 1) emit_levels<1><<<dim3(nsb, num_levels - 1), 1024, 0, st>>>(d_coords[0], (int)n, lv, slots, bsum, d_counts);
        else emit_levels<8><<<dim3(nsb, num_levels - 1), 1024, 0, st>>>(d_coords[0], (int)n, lv, slots, bsum, d_counts);
        CV_LAUNCH_CHECK();
    }
    if (h_counts) {
        CV_HIP_CHECK(hipMemcpyAsync(h_counts, d_counts, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
        CV_HIP_CHECK(hipStreamSynchronize(st));
    }
    return CV_OK;
}

extern "C" {

int cv_sp_build_levels(int32_t* const* d_coords, unsigned long long* const* d_keys,
                       int32_t* const* d_vals, long long n, long long cap, int num_levels,
                       int32_t* d_counts, int32_t* h_counts, void* d_ws, size_t ws_bytes, void* stream) {
    return cv_sp_build_levels_zero(d_coords, d_keys, d_vals, n, cap, num_levels, d_counts, h_counts, d_ws, ws_bytes, nullptr,
                                   0, nullptr, stream);
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

size_t cv_sp_sort_workspace_bytes(long long n) {
    if (n <= 0) return 0;
    const size_t nblk = (size_t)((n + SORT_ROWS - 1) / SORT_ROWS);
    return 256 + 4 * cv_align_up(sizeof(int) * (size_t)n, 256) + cv_align_up(sizeof(int) * nblk * SORT_BINS, 256);
}

// Spatial (batch, Z-order of coarse cubes, caller order inside a cube) row order of a coordinate set, stable.
//   d_sorted [n][4] = the rows in that order, d_perm[n] = original row of sorted row, d_inv[n] = sorted row of
//   original row.  Asynchronous, 8 launches, no host synchronisation.
int cv_sp_sort_rows(const int32_t* d_coords, long long n, int32_t* d_sorted, int32_t* d_perm, int32_t* d_inv,
                    void* d_ws, size_t ws_bytes, void* stream) {
    return cv_sp_sort_rows_ex(d_coords, n, d_sorted, d_perm, d_inv, d_ws, ws_bytes, false, stream);
}

}  // extern "C"

// (C++ linkage, cv_common.h) single_batch: every row is of one scene - the third digit (the batch index) is not sorted:
// five launches instead of seven
int cv_sp_sort_rows_ex(const int32_t* d_coords, long long n, int32_t* d_sorted, int32_t* d_perm, int32_t* d_inv,
                       void* d_ws, size_t ws_bytes, bool single_batch, void* stream, bool bounds_prefilled) {
    CV_REQUIRE(d_coords && d_sorted && d_perm && d_inv && d_ws, CV_EINVAL, "null pointer argument");
    CV_REQUIRE(n > 0 && n < (1ll << 30), CV_EINVAL, "bad row count %lld", n);
    CV_REQUIRE(ws_bytes >= cv_sp_sort_workspace_bytes(n), CV_ENOMEM, "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    CvCarver cv(d_ws);
    int* mm = cv.take<int>(8);
    unsigned* keys_a = cv.take<unsigned>(n);
    unsigned* keys_b = cv.take<unsigned>(n);
    int* vals_a = cv.take<int>(n);
    int* vals_b = cv.take<int>(n);
    const int nblk = (int)((n + SORT_ROWS - 1) / SORT_ROWS);
    int* hist = cv.take<int>((size_t)nblk * SORT_BINS);
    if (!bounds_prefilled) CV_HIP_CHECK(hipMemsetAsync(mm, 0x7f, sizeof(int) * 8, st));      // (the scene call's first stage fills them)
    sort_minmax<<<(unsigned)std::min<long long>((n + 255) / 256, 256), 256, 0, st>>>(d_coords, n, mm);
    CV_LAUNCH_CHECK();
    sort_hist<<<nblk, SORT_T, 0, st>>>(d_coords, mm, keys_a, n, 0, hist);
    CV_LAUNCH_CHECK();
    sort_scatter<<<nblk, SORT_T, 0, st>>>(keys_a, nullptr, n, 0, hist, nblk, keys_b, vals_b, nullptr, nullptr, nullptr,
                                          nullptr, mm);
    CV_LAUNCH_CHECK();
    sort_hist<<<nblk, SORT_T, 0, st>>>(nullptr, mm, keys_b, n, 1, hist);
    CV_LAUNCH_CHECK();
    sort_scatter<<<nblk, SORT_T, 0, st>>>(keys_b, vals_b, n, 1, hist, nblk, keys_a, vals_a, d_coords, d_perm, d_inv,
                                          d_sorted, mm, single_batch ? 1 : 0);
    CV_LAUNCH_CHECK();
    if (single_batch) return CV_OK;
    sort_hist<<<nblk, SORT_T, 0, st>>>(nullptr, mm, keys_a, n, 2, hist);
    CV_LAUNCH_CHECK();
    sort_scatter<<<nblk, SORT_T, 0, st>>>(keys_a, vals_a, n, 2, hist, nblk, nullptr, nullptr, d_coords, d_perm, d_inv,
                                          d_sorted, mm);
    CV_LAUNCH_CHECK();
    return CV_OK;
}

extern "C" {

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
