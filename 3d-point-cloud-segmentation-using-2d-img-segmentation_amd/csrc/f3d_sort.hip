// Cell sort of a point cloud: order the points by coarse grid cell (HBM-bound, key-index radix sort, all kernels our own).
//
// Why it exists: the fused kernel assigns 128 consecutive points to a wavefront.  When consecutive
// points are spatial neighbours, a wave is (almost always) entirely inside or entirely outside a
// view's frustum, so the expensive projection runs only for waves that have visible points.  A
// cloud in arbitrary order wastes ~60 % of the projection work on masked-off lanes.  Sorting changes
// nothing in the results: every point's label depends on its own xyz only, and the kernel writes it
// back to the caller's index through `perm`.
//
//   k_bbox_partial : finite bounding box of a strided sample of the cloud, <= 64 partial boxes (the cell grid -- 16 key bits dealt to
//                    the axes -- is derived from them by the first wave of every k_rs_keys block)
//   k_rs_keys      : ONE read of xyz: key[i] = 16-bit Morton code of point i's cell (2 B/point written) and, per 8192-point
//                    tile, the histogram of the keys' low bytes                                                  (streaming)
//   k_rs_scan      : per digit value, the exclusive scan of its counts over the tiles (one block per digit value)
//   k_rs_scatter<1>: LSD pass 1 (low byte): one record (high byte << 24 | index) per key in low-byte order, 4 B/point written
//                    (8 B beyond 2^24 points)
//   k_rs_hist2     : histogram of the high bytes per tile of that order
//   k_rs_scatter<2>: LSD pass 2 (high byte, stable): perm
//   k_gather_xyz   : sorted[j] = xyz[perm[j]]     (only for the "prepared layout" entry point; the
//                    in-step sort lets the fused kernel read xyz through perm instead)
// A tile's ranks come from per-wave digit counts in LDS and 8 ballots per 64 keys (match on the digit's bits): no global atomic,
// no look-back chain, the same perm in every run (equal keys keep their index order: both passes are stable).
// History: a first version used one returning global atomic per point (counting sort): ~0.5 ms per kernel at 10M points.  r1/r2
// used rocprim::radix_sort_pairs on (uint16 key, uint32 index): 0.27 ms for the whole sort at 10M points, of which 2 x 81 us in
// its onesweep passes (decoupled look-back); the passes here move fewer bytes (the index is implicit in pass 1, the key is
// 1 byte in pass 2) and need no look-back.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include "f3d.h"
#include "f3d_kernels.h"

namespace {

constexpr int SB = 256;
#define F3D_SORT_KEY_BITS 16                 // two 8-bit radix passes
typedef uint16_t sort_key_t;

struct bbox6 { double lo[3], hi[3]; };

template <typename T>
__global__ __launch_bounds__(SB) void k_bbox_partial(const T* __restrict__ xyz, int64_t n, int64_t stride, bbox6* __restrict__ partial) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    // a strided sample is enough: points outside the sampled box are clamped into the border cells, which can only cost
    // a little coherence, never correctness
#pragma unroll 4
    for (int64_t i = ((int64_t)blockIdx.x * SB + threadIdx.x) * stride; i < n; i += (int64_t)gridDim.x * SB * stride) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double x = (double)xyz[3 * i + c];
            if (fabs(x) < 1e300) { lo[c] = fmin(lo[c], x); hi[c] = fmax(hi[c], x); }
        }
    }
    __shared__ double sl[3][SB / 64], sh[3][SB / 64];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a = lo[c], b = hi[c];
        for (int off = 32; off >= 1; off >>= 1) { a = fmin(a, __shfl_xor(a, off, 64)); b = fmax(b, __shfl_xor(b, off, 64)); }
        if ((threadIdx.x & 63) == 0) { sl[c][threadIdx.x >> 6] = a; sh[c][threadIdx.x >> 6] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double a = sl[threadIdx.x][0], b = sh[threadIdx.x][0];
        for (int w = 1; w < SB / 64; ++w) { a = fmin(a, sl[threadIdx.x][w]); b = fmax(b, sh[threadIdx.x][w]); }
        partial[blockIdx.x].lo[threadIdx.x] = a; partial[blockIdx.x].hi[threadIdx.x] = b;
    }
}

// the cell grid from the partial boxes (at most 64 of them: one lane each).  Called by the first wave of every k_rs_keys block --
// 64 x 48 bytes out of L2 and ~100 scalar-ish instructions per block, instead of a one-block kernel of its own between two launches.
__device__ __forceinline__ void grid_from_partials(const bbox6* __restrict__ partial, int nparts, f3d_cellgrid* out /*LDS*/) {
    const int lane = threadIdx.x & 63;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    if (lane < nparts)
        for (int c = 0; c < 3; ++c) { lo[c] = partial[lane].lo[c]; hi[c] = partial[lane].hi[c]; }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        for (int off = 32; off >= 1; off >>= 1) { lo[c] = fmin(lo[c], __shfl_xor(lo[c], off, 64)); hi[c] = fmax(hi[c], __shfl_xor(hi[c], off, 64)); }
    if (lane == 0) {
        f3d_cellgrid g;
        double ext[3];
        for (int c = 0; c < 3; ++c) {
            g.lo[c] = lo[c] <= hi[c] ? lo[c] : 0.0;
            ext[c] = lo[c] <= hi[c] ? (hi[c] - lo[c]) : 0.0;
            if (!(ext[c] > 1e-12)) ext[c] = 1e-12;
        }
        // F3D_SORT_KEY_BITS key bits are dealt to the axes one at a time, always to the axis whose cells are currently the
        // longest, so the cells come out as cubic as the box allows; the key interleaves the axes' bits from the most
        // significant level down (a Morton code with per-axis bit counts), so a contiguous run of sorted points is a
        // compact 3-D block (small footprint in every view's mask, good for gather coalescing and the per-XCD L2).
        int bits[3] = {0, 0, 0};
        for (int k = 0; k < F3D_SORT_KEY_BITS; ++k) {
            int best = 0;
            double bl = -1.0;
            for (int c = 0; c < 3; ++c) {
                const double len = ext[c] / (double)(1 << bits[c]);
                if (len > bl) { bl = len; best = c; }
            }
            ++bits[best];
        }
        for (int c = 0; c < 3; ++c) {
            g.dim[c] = 1 << bits[c];
            g.bits[c] = bits[c];
            g.inv_cell[c] = (double)g.dim[c] / (ext[c] * 1.0000001);
        }
        g.ncells = 1 << F3D_SORT_KEY_BITS;                     // key space; the last key also collects non-finite points
        *out = g;
    }
}

// the bits of axis c's cell index `i` at their places in the key (a Morton code with per-axis bit counts: from the most significant
// level down, every axis that still has a bit at that level contributes it)
__device__ __forceinline__ uint32_t spread_axis(uint32_t i, int c, const int bits[3]) {
    uint32_t key = 0;
    int pos = F3D_SORT_KEY_BITS;                                    // next output position, counted down
    for (int level = F3D_SORT_KEY_BITS - 1; level >= 0; --level)
        for (int a = 0; a < 3; ++a)
            if (bits[a] > level) { --pos; if (a == c) key |= ((i >> level) & 1u) << pos; }
    return key;
}

constexpr int RS_LUT = 1024;                 // per-axis spread tables in LDS when no axis has more cells than this (the usual case: 64 x 64 x 16)

// float32 is enough for the cell of a point: the ORDER of the points never changes a result, only how compact a wave's 128 points are
template <typename T>
__device__ __forceinline__ void cell_index(const T* __restrict__ p, const float lo[3], const float inv[3], const int dim[3], int idx[3], bool& ok) {
    ok = true;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float x = (float)p[c];
        ok = ok && (fabsf(x) < 1e30f);
        int k = (int)((x - lo[c]) * inv[c]);
        k = k < 0 ? 0 : (k >= dim[c] ? dim[c] - 1 : k);
        idx[c] = k;
    }
}

template <typename T>
__device__ __forceinline__ uint32_t cell_of(const T* __restrict__ p, const f3d_cellgrid& g) {
    int idx[3];
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double x = (double)p[c];
        ok = ok && (fabs(x) < 1e300);
        int k = (int)((x - g.lo[c]) * g.inv_cell[c]);
        k = k < 0 ? 0 : (k >= g.dim[c] ? g.dim[c] - 1 : k);
        idx[c] = k;
    }
    if (!ok) return (1u << F3D_SORT_KEY_BITS) - 1u;
    uint32_t key = 0;
    for (int level = F3D_SORT_KEY_BITS - 1; level >= 0; --level)     // at most 3 bits appended per level
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (g.bits[c] > level) key = (key << 1) | (((uint32_t)idx[c] >> level) & 1u);
    return key;
}

// ---- the radix passes --------------------------------------------------------------------------------------------
#ifndef F3D_RS_THREADS
#define F3D_RS_THREADS 512
#endif
#ifndef F3D_RS_ROUNDS
#define F3D_RS_ROUNDS 16
#endif
constexpr int RS_THREADS = F3D_RS_THREADS;   // 8 waves per block
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ROUNDS = F3D_RS_ROUNDS;     // a wave walks 16 x 64 consecutive keys
constexpr int RS_TILE = RS_THREADS * RS_ROUNDS;   // 8192 keys per block

// item (wave w, round r, lane l) of tile b: consecutive lanes = consecutive keys (coalesced), consecutive rounds and waves
// = ascending index (what makes the ranks below stable)
__device__ __forceinline__ int64_t rs_item(int64_t tile, int wave, int round, int lane) {
    return tile * RS_TILE + (int64_t)wave * (RS_ROUNDS * 64) + round * 64 + lane;
}

// keys of one tile + histogram of their low bytes.  blkhist is digit-major: [256][ntiles].  The block shape is this kernel's own
// (F3D_RK_THREADS threads walk the tile's 8192 points): measured 1024 / 512 / 256 / 128 / 64 threads = 76 / 64 / 57 / 74 / 114 us at 10M points --
// 1221 blocks of 512 threads are 1.2 "waves" of resident blocks at 4 per CU (the last fifth runs on an empty chip), 256 threads all fit at once.
#ifndef F3D_RK_THREADS
#define F3D_RK_THREADS 256
#endif
// A small cloud (fewer tiles than two per CU) takes 512 threads per tile: the chip is not full either way and the rounds of a thread halve
// (1.25M points: sort 71 -> 64 us).
constexpr int RK_SMALL_THREADS = 512;
template <typename T, int RK_THREADS>
__global__ __launch_bounds__(RK_THREADS) void k_rs_keys(const T* __restrict__ xyz, int64_t n, const bbox6* __restrict__ partial, int nparts,
                                                         sort_key_t* __restrict__ keys, uint32_t* __restrict__ blkhist, int ntiles) {
    __shared__ uint32_t hist[256];
    __shared__ f3d_cellgrid sg;
    __shared__ uint16_t lut[3][RS_LUT];
    constexpr int RK_ROUNDS = RS_TILE / RK_THREADS;
    static_assert(RK_THREADS >= 64 && RK_THREADS * RK_ROUNDS == RS_TILE, "the key kernel's block must cover a tile");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) grid_from_partials(partial, nparts, &sg);
    for (int d = threadIdx.x; d < 256; d += RK_THREADS) hist[d] = 0u;
    __syncthreads();
    f3d_cellgrid g = sg;
#pragma unroll
    for (int c = 0; c < 3; ++c) {                                   // block-uniform: keep the integers in SGPRs (scalar branches in cell_of)
        g.dim[c] = __builtin_amdgcn_readfirstlane(g.dim[c]);
        g.bits[c] = __builtin_amdgcn_readfirstlane(g.bits[c]);
    }
    const bool tables = g.dim[0] <= RS_LUT && g.dim[1] <= RS_LUT && g.dim[2] <= RS_LUT;
    if (tables) {                                                   // key = lut[0][ix] | lut[1][iy] | lut[2][iz]: three LDS reads instead of a 48-step bit loop
        for (int c = 0; c < 3; ++c)
            for (int i = threadIdx.x; i < g.dim[c]; i += RK_THREADS) lut[c][i] = (uint16_t)spread_axis((uint32_t)i, c, g.bits);
        __syncthreads();
    }
    const float lo[3] = {(float)g.lo[0], (float)g.lo[1], (float)g.lo[2]};
    const float inv[3] = {(float)g.inv_cell[0], (float)g.inv_cell[1], (float)g.inv_cell[2]};
#pragma unroll 8
    for (int r = 0; r < RK_ROUNDS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * (RK_ROUNDS * 64) + r * 64 + lane;
        if (i < n) {
            uint32_t key;
            if (tables) {
                int idx[3]; bool ok;
                cell_index(xyz + 3 * i, lo, inv, g.dim, idx, ok);
                key = ok ? ((uint32_t)lut[0][idx[0]] | lut[1][idx[1]] | lut[2][idx[2]]) : ((1u << F3D_SORT_KEY_BITS) - 1u);
            } else {
                key = cell_of(xyz + 3 * i, g);
            }
            keys[i] = (sort_key_t)key;
            atomicAdd(&hist[key & 0xFFu], 1u);
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < 256; d += RK_THREADS) blkhist[(size_t)d * ntiles + blockIdx.x] = hist[d];
}

// Pass 1 hands pass 2 ONE record per key: (high byte << 24) | index in 32 bits while the indices fit 24 bits (n <= 2^24: 4 B/point),
// (high byte << 32) | index in 64 bits beyond.  (Separate 1-byte and 4-byte arrays cost two store instructions per key; the byte
// stores alone made pass 1 twice as slow as pass 2.)
template <typename R> struct rs_rec;
template <> struct rs_rec<uint32_t> { static constexpr int shift = 24; static constexpr uint32_t imask = 0xFFFFFFu; };
template <> struct rs_rec<uint64_t> { static constexpr int shift = 32; static constexpr uint64_t imask = 0xFFFFFFFFull; };

// histogram of the high bytes per tile of the pass-1 order
template <typename R>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist2(const R* __restrict__ recs, int64_t n, uint32_t* __restrict__ blkhist, int ntiles) {
    __shared__ uint32_t hist[256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = rs_item(blockIdx.x, wave, r, lane);
        if (i < n) atomicAdd(&hist[(uint32_t)(recs[i] >> rs_rec<R>::shift) & 0xFFu], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256) blkhist[(size_t)threadIdx.x * ntiles + blockIdx.x] = hist[threadIdx.x];
}

// block d: exclusive scan of row d of blkhist (the tiles' counts of digit value d) in place; tot[d] = the row's sum
__global__ __launch_bounds__(256) void k_rs_scan(uint32_t* __restrict__ blkhist, int ntiles, uint32_t* __restrict__ tot) {
    __shared__ uint32_t part[256];
    uint32_t* row = blkhist + (size_t)blockIdx.x * ntiles;
    const int per = (ntiles + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(ntiles, lo + per);
    uint32_t sum = 0;
    for (int k = lo; k < hi; ++k) sum += row[k];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {                        // Hillis-Steele inclusive scan over the 256 partial sums
        const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;                          // exclusive prefix of this thread's piece
    for (int k = lo; k < hi; ++k) { const uint32_t c = row[k]; row[k] = run; run += c; }
    if (threadIdx.x == 255) tot[blockIdx.x] = part[255];
}

// exclusive prefix of `v` over threads 0..255 (threads >= 256 pass 0 and get garbage); every thread of the block must call it
__device__ __forceinline__ uint32_t rs_excl_scan256(uint32_t v, uint32_t* wsum /*[4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
    __syncthreads();                                                  // (wsum may still be read by an earlier call)
    if (lane == 63 && wave < 4) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = 0u;
    for (int w = 0; w < 4; ++w) before += (w < wave) ? wsum[w] : 0u;
    return before + inc - v;
}

// One LSD pass over a tile.  PASS 1: digit = low byte of keys[i]; writes the record (high byte, i) to the key's place in low-byte
// order.  PASS 2: digit = the record's high byte (pass-1 order, stable); writes the record's index to perm.  offs = the scanned
// blkhist, tot = the digit totals.
// The tile is first ordered in LDS (per-wave digit counts -> positions; ranks inside a round from 8 ballots), then written out by
// consecutive threads: the keys of one digit value leave as one contiguous run (~128 B of records at 8192 keys / 256 values)
// instead of 64 scattered stores per instruction (measured at 10M keys: pass 1 144 -> 95 us with the run-wise output).
template <int PASS, typename R>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const sort_key_t* __restrict__ keys, const R* __restrict__ recs_in, int64_t n,
                                                            const uint32_t* __restrict__ offs, const uint32_t* __restrict__ tot, int ntiles,
                                                            R* __restrict__ recs_out, uint32_t* __restrict__ perm_out) {
    __shared__ uint32_t wcount[RS_WAVES][256];                       // per wave: count of each digit value, then its running position in the tile
    __shared__ uint32_t gdelta[256];                                 // global position of the tile's first key of a digit value - its position in the tile
    __shared__ uint32_t wsum[4];
    extern __shared__ unsigned long long rs_dyn[];                   // the tile in digit order: RS_TILE records (+ pass 1: their low bytes)
    R* srec = reinterpret_cast<R*>(rs_dyn);
    uint8_t* sdig = reinterpret_cast<uint8_t*>(srec + RS_TILE);      // pass 1 only: the record carries the HIGH byte, the run is found by the low one
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    volatile uint32_t* mine = wcount[wave];
    // requested first: the two table entries of this thread's digit value (latency hidden behind the key loads and the counting)
    const int d_own = threadIdx.x & 255;
    const bool own = threadIdx.x < 256;
    const uint32_t t_d = own ? tot[d_own] : 0u;
    const uint32_t o_d = own ? offs[(size_t)d_own * ntiles + blockIdx.x] : 0u;
    for (int k = threadIdx.x; k < RS_WAVES * 256; k += RS_THREADS) (&wcount[0][0])[k] = 0u;
    __syncthreads();
    uint32_t dig[RS_ROUNDS];
    R val[RS_ROUNDS];                                                // the record that leaves (pass 1: built here; pass 2: the one that came in)
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = rs_item(blockIdx.x, wave, r, lane);
        dig[r] = 0xFFFFFFFFu; val[r] = 0;
        if (i < n) {
            if (PASS == 1) { const uint32_t k = keys[i]; dig[r] = k & 0xFFu; val[r] = ((R)(k >> 8) << rs_rec<R>::shift) | (R)i; }
            else { val[r] = recs_in[i]; dig[r] = (uint32_t)(val[r] >> rs_rec<R>::shift) & 0xFFu; }
            atomicAdd(const_cast<uint32_t*>(&mine[dig[r]]), 1u);
        }
    }
    __syncthreads();
    {
        uint32_t c_d = 0u;
        if (own) for (int w = 0; w < RS_WAVES; ++w) c_d += wcount[w][d_own];
        const uint32_t gstart = rs_excl_scan256(t_d, wsum);          // keys of smaller digit values in the whole array
        const uint32_t lstart = rs_excl_scan256(c_d, wsum);          // ... in this tile
        if (own) {
            gdelta[d_own] = gstart + o_d - lstart;
            uint32_t run = lstart;
            for (int w = 0; w < RS_WAVES; ++w) { const uint32_t c = wcount[w][d_own]; wcount[w][d_own] = run; run += c; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const bool live = dig[r] != 0xFFFFFFFFu;
        unsigned long long same = __ballot(live);                    // lanes of this round with my digit value: 8 ballots, one per bit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dig[r] >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            same &= bit ? m : ~m;
        }
        if (live) {                                                  // (the wave's lanes run in lockstep: every read below precedes the leader's write)
            const unsigned rank = __popcll(same & ((1ull << lane) - 1ull));
            const uint32_t base = mine[dig[r]];
            if (rank == 0u) mine[dig[r]] = base + (uint32_t)__popcll(same);
            srec[base + rank] = val[r];
            if (PASS == 1) sdig[base + rank] = (uint8_t)dig[r];
        }
    }
    __syncthreads();
    const int64_t left = n - (int64_t)blockIdx.x * RS_TILE;
    const int count = (int)(left < RS_TILE ? left : RS_TILE);
#pragma unroll 4
    for (int j = threadIdx.x; j < count; j += RS_THREADS) {
        const R rec = srec[j];
        if (PASS == 1) recs_out[(uint32_t)j + gdelta[sdig[j]]] = rec;
        else perm_out[(uint32_t)j + gdelta[(uint32_t)(rec >> rs_rec<R>::shift) & 0xFFu]] = (uint32_t)(rec & rs_rec<R>::imask);
    }
}

template <typename T>
__global__ __launch_bounds__(SB) void k_gather_xyz(const T* __restrict__ xyz, int64_t n, const int32_t* __restrict__ perm,
                                                    T* __restrict__ sorted) {
    for (int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x; i < n; i += (int64_t)gridDim.x * SB) {
        const size_t src = (size_t)perm[i];
        sorted[3 * i] = xyz[3 * src]; sorted[3 * i + 1] = xyz[3 * src + 1]; sorted[3 * i + 2] = xyz[3 * src + 2];
    }
}

struct sort_layout {
    size_t partial, keys, recs, hist, tot, total;
    int ntiles;
    bool wide;                                                      // 64-bit records (more than 2^24 points)
};

sort_layout layout_for(int64_t n) {
    sort_layout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    L.ntiles = (int)((n + RS_TILE - 1) / RS_TILE);
    L.wide = n > (1 << 24);
    L.partial = take(64 * sizeof(bbox6));
    L.keys = take((size_t)n * sizeof(sort_key_t));
    L.recs = take((size_t)n * (L.wide ? 8 : 4));
    L.hist = take((size_t)256 * L.ntiles * 4);
    L.tot = take(256 * 4);
    L.total = off;
    return L;
}

template <typename R>
hipError_t run_passes(const sort_layout& L, char* base, int64_t n, int32_t* perm, hipStream_t s) {
    const sort_key_t* keys = reinterpret_cast<const sort_key_t*>(base + L.keys);
    R* recs = reinterpret_cast<R*>(base + L.recs);
    uint32_t* hist = reinterpret_cast<uint32_t*>(base + L.hist);
    uint32_t* tot = reinterpret_cast<uint32_t*>(base + L.tot);
    const dim3 gt(L.ntiles), bt(RS_THREADS);
    const size_t lds1 = (size_t)RS_TILE * sizeof(R) + RS_TILE, lds2 = (size_t)RS_TILE * sizeof(R);
    if (lds1 > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_rs_scatter<1, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_rs_scatter<2, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_rs_scan, dim3(256), dim3(256), 0, s, hist, L.ntiles, tot);
    hipLaunchKernelGGL((k_rs_scatter<1, R>), gt, bt, lds1, s, keys, (const R*)nullptr, n, hist, tot, L.ntiles, recs, (uint32_t*)nullptr);
    hipLaunchKernelGGL(k_rs_hist2<R>, gt, bt, 0, s, recs, n, hist, L.ntiles);
    hipLaunchKernelGGL(k_rs_scan, dim3(256), dim3(256), 0, s, hist, L.ntiles, tot);
    hipLaunchKernelGGL((k_rs_scatter<2, R>), gt, bt, lds2, s, (const sort_key_t*)nullptr, recs, n, hist, tot, L.ntiles, (R*)nullptr, reinterpret_cast<uint32_t*>(perm));
    return hipGetLastError();
}

}  // namespace

size_t f3d_sort_scratch_bytes(int64_t n) { return layout_for(n < 1 ? 1 : n).total; }

hipError_t f3d_launch_cell_sort(const void* xyz, int dtype, int64_t n, void* sorted_xyz, int32_t* perm, void* scratch, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n > 0x7fffffffLL) return hipErrorInvalidValue;
    const sort_layout L = layout_for(n);
    char* base = reinterpret_cast<char*>(scratch);
    bbox6* partial = reinterpret_cast<bbox6*>(base + L.partial);
    sort_key_t* keys = reinterpret_cast<sort_key_t*>(base + L.keys);
    uint32_t* hist = reinterpret_cast<uint32_t*>(base + L.hist);
    const int64_t gb = (n + SB - 1) / SB;
    const int64_t stride = n > (1 << 16) ? n >> 16 : 1;          // inspect <= ~65k points for the bounding box (4 strided loads per thread of 64 blocks)
    const int64_t sb = ((n + stride - 1) / stride + SB - 1) / SB;
    const int nparts = (int)(sb < 64 ? sb : 64);                 // (one lane of k_rs_keys' first wave per partial box)
    const int gstream = (int)(gb < 8192 ? gb : 8192);
    const dim3 gt(L.ntiles);
    const bool small = L.ntiles < 512;                          // fewer tiles than two per CU
    if (dtype == F3D_F64) {
        hipLaunchKernelGGL(k_bbox_partial<double>, dim3(nparts), dim3(SB), 0, s, (const double*)xyz, n, stride, partial);
        if (small) hipLaunchKernelGGL((k_rs_keys<double, RK_SMALL_THREADS>), gt, dim3(RK_SMALL_THREADS), 0, s, (const double*)xyz, n, partial, nparts, keys, hist, L.ntiles);
        else hipLaunchKernelGGL((k_rs_keys<double, F3D_RK_THREADS>), gt, dim3(F3D_RK_THREADS), 0, s, (const double*)xyz, n, partial, nparts, keys, hist, L.ntiles);
    } else {
        hipLaunchKernelGGL(k_bbox_partial<float>, dim3(nparts), dim3(SB), 0, s, (const float*)xyz, n, stride, partial);
        if (small) hipLaunchKernelGGL((k_rs_keys<float, RK_SMALL_THREADS>), gt, dim3(RK_SMALL_THREADS), 0, s, (const float*)xyz, n, partial, nparts, keys, hist, L.ntiles);
        else hipLaunchKernelGGL((k_rs_keys<float, F3D_RK_THREADS>), gt, dim3(F3D_RK_THREADS), 0, s, (const float*)xyz, n, partial, nparts, keys, hist, L.ntiles);
    }
    hipError_t e = L.wide ? run_passes<uint64_t>(L, base, n, perm, s) : run_passes<uint32_t>(L, base, n, perm, s);
    if (e != hipSuccess) return e;
    if (sorted_xyz) {
        if (dtype == F3D_F64) hipLaunchKernelGGL(k_gather_xyz<double>, dim3(gstream), dim3(SB), 0, s, (const double*)xyz, n, perm, (double*)sorted_xyz);
        else hipLaunchKernelGGL(k_gather_xyz<float>, dim3(gstream), dim3(SB), 0, s, (const float*)xyz, n, perm, (float*)sorted_xyz);
    }
    return hipGetLastError();
}
