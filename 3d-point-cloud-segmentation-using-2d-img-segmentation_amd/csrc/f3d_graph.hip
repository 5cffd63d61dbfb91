// (f)#1 adjacency: sklearn.neighbors.KDTree(points).query_radius(points, r=2*ds_radius) of fusion.py:374-375 as a CSR graph.
//
// A uniform grid of cell edge >= r replaces the tree: every neighbour of a point lies in the 27 cells around its own.
//   k_graph_bbox      : finite bounding box of the whole cloud + count of non-finite coordinates (sklearn rejects those)
//   k_graph_keys      : key[i] = linear cell id, idx[i] = i
//   rocprim radix sort: (key, idx) -> cell order (stable: indices ascend inside a cell)
//   k_graph_cells     : [first, last) position of every non-empty cell in the sorted order
//   k_graph_gather    : sorted float64 copy of the cloud (candidate loops read it contiguously)
//   k_graph_scan<0>   : neighbours per point -> counts[orig] ; rocprim exclusive scan -> offsets[n + 1]
//   k_graph_scan<1>   : same loops, writes the neighbours' caller-order indices at offsets[orig]
// The distance test is the tree's leaf test, operation for operation (sklearn/metrics/_dist_metrics: euclidean_rdist
// accumulates tmp * tmp over the 3 coordinates left to right; query_radius compares it with r * r, inclusive).  The
// order inside a row is (cell, index) instead of the tree's traversal order, which sklearn does not specify either.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "f3d.h"
#include "f3d_kernels.h"

namespace {

constexpr int GB = 256;

struct gbox { double lo[3], hi[3]; unsigned long long bad; };

template <typename T>
__global__ __launch_bounds__(GB) void k_graph_bbox(const T* __restrict__ xyz, int64_t n, gbox* __restrict__ partial) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    unsigned long long bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x; i < n; i += (int64_t)gridDim.x * GB) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double x = (double)xyz[3 * i + c];
            if (fabs(x) <= 1.7976931348623157e308) { lo[c] = fmin(lo[c], x); hi[c] = fmax(hi[c], x); } else ++bad;
        }
    }
    __shared__ gbox sh[GB / 64];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a = lo[c], b = hi[c];
        for (int off = 32; off >= 1; off >>= 1) { a = fmin(a, __shfl_xor(a, off, 64)); b = fmax(b, __shfl_xor(b, off, 64)); }
        if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6].lo[c] = a; sh[threadIdx.x >> 6].hi[c] = b; }
    }
    for (int off = 32; off >= 1; off >>= 1) bad += __shfl_xor(bad, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6].bad = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        gbox o = sh[0];
        for (int w = 1; w < GB / 64; ++w) {
            for (int c = 0; c < 3; ++c) { o.lo[c] = fmin(o.lo[c], sh[w].lo[c]); o.hi[c] = fmax(o.hi[c], sh[w].hi[c]); }
            o.bad += sh[w].bad;
        }
        partial[blockIdx.x] = o;
    }
}

__device__ __forceinline__ void cell_of(const f3d_graphgrid& g, double x, double y, double z, int& cx, int& cy, int& cz) {
    // clamped: rounding at the upper faces of the box must not leave the grid
    cx = min(g.dim[0] - 1, max(0, (int)floor((x - g.lo[0]) * g.inv_cell)));
    cy = min(g.dim[1] - 1, max(0, (int)floor((y - g.lo[1]) * g.inv_cell)));
    cz = min(g.dim[2] - 1, max(0, (int)floor((z - g.lo[2]) * g.inv_cell)));
}

template <typename T>
__global__ __launch_bounds__(GB) void k_graph_keys(const T* __restrict__ xyz, int64_t n, f3d_graphgrid g, uint32_t* __restrict__ keys,
                                                    uint32_t* __restrict__ idx) {
    for (int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x; i < n; i += (int64_t)gridDim.x * GB) {
        int cx, cy, cz;
        cell_of(g, (double)xyz[3 * i], (double)xyz[3 * i + 1], (double)xyz[3 * i + 2], cx, cy, cz);
        keys[i] = (uint32_t)((cz * g.dim[1] + cy) * g.dim[0] + cx);
        idx[i] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(GB) void k_graph_cells(const uint32_t* __restrict__ keys, int64_t n, int2* __restrict__ cells) {
    for (int64_t j = (int64_t)blockIdx.x * GB + threadIdx.x; j < n; j += (int64_t)gridDim.x * GB) {
        const uint32_t k = keys[j];
        if (j == 0 || keys[j - 1] != k) cells[k].x = (int)j;
        if (j == n - 1 || keys[j + 1] != k) cells[k].y = (int)j + 1;
    }
}

template <typename T>
__global__ __launch_bounds__(GB) void k_graph_gather(const T* __restrict__ xyz, int64_t n, const uint32_t* __restrict__ perm,
                                                      double* __restrict__ sorted) {
    for (int64_t j = (int64_t)blockIdx.x * GB + threadIdx.x; j < n; j += (int64_t)gridDim.x * GB) {
        const int64_t i = perm[j];
        sorted[3 * j] = (double)xyz[3 * i]; sorted[3 * j + 1] = (double)xyz[3 * i + 1]; sorted[3 * j + 2] = (double)xyz[3 * i + 2];
    }
}

// one thread per point, in cell order (a wave's threads walk nearly the same candidate ranges)
template <bool FILL>
__global__ __launch_bounds__(GB) void k_graph_scan(const double* __restrict__ sorted, int64_t n, const uint32_t* __restrict__ perm,
                                                    f3d_graphgrid g, const int2* __restrict__ cells, double r2,
                                                    int64_t* __restrict__ offsets, int32_t* __restrict__ nbrs) {
    for (int64_t j = (int64_t)blockIdx.x * GB + threadIdx.x; j < n; j += (int64_t)gridDim.x * GB) {
        const double px = sorted[3 * j], py = sorted[3 * j + 1], pz = sorted[3 * j + 2];
        int cx, cy, cz;
        cell_of(g, px, py, pz, cx, cy, cz);
        const int64_t orig = perm[j];
        int64_t out = FILL ? offsets[orig] : 0;
        for (int dz = -1; dz <= 1; ++dz) {
            const int z = cz + dz;
            if (z < 0 || z >= g.dim[2]) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int y = cy + dy;
                if (y < 0 || y >= g.dim[1]) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    const int x = cx + dx;
                    if (x < 0 || x >= g.dim[0]) continue;
                    const int2 range = cells[(z * g.dim[1] + y) * g.dim[0] + x];
                    for (int k = range.x; k < range.y; ++k) {
                        const double t0 = px - sorted[3 * (int64_t)k], t1 = py - sorted[3 * (int64_t)k + 1], t2 = pz - sorted[3 * (int64_t)k + 2];
                        const double d = (t0 * t0 + t1 * t1) + t2 * t2;            // euclidean_rdist, left to right
                        if (d <= r2) {
                            if (FILL) nbrs[out] = (int32_t)perm[k];
                            ++out;
                        }
                    }
                }
            }
        }
        if (!FILL) offsets[orig] = out;
    }
}

inline int grid_blocks(int64_t n) { int64_t b = (n + GB - 1) / GB; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

struct graph_layout { size_t keys_a, keys_b, idx_a, perm, sorted, cells, bbox, temp, total; };

graph_layout layout_for(int64_t n, int64_t ncells, size_t temp_bytes) {
    graph_layout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    L.keys_a = take((size_t)n * 4); L.keys_b = take((size_t)n * 4); L.idx_a = take((size_t)n * 4); L.perm = take((size_t)n * 4);
    L.sorted = take((size_t)n * 24); L.cells = take((size_t)ncells * 8); L.bbox = take(sizeof(gbox) * 1024); L.temp = take(temp_bytes);
    L.total = o;
    return L;
}

size_t temp_bytes_for(int64_t n) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0u, 32u);
    (void)rocprim::exclusive_scan(nullptr, b, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>());
    return (a > b ? a : b) + 256;
}

}  // namespace

size_t f3d_graph_bbox_bytes(void) { return sizeof(gbox) * 1024; }

// stage 1 of the count pass: bounding box partials (the host reduces <= 1024 of them and chooses the grid)
hipError_t f3d_launch_graph_bbox(const void* xyz, int dtype, int64_t n, void* partial, int* nblocks, hipStream_t s) {
    int b = grid_blocks(n); if (b > 1024) b = 1024;
    *nblocks = b;
    if (dtype == F3D_F64) hipLaunchKernelGGL(k_graph_bbox<double>, dim3(b), dim3(GB), 0, s, (const double*)xyz, n, (gbox*)partial);
    else hipLaunchKernelGGL(k_graph_bbox<float>, dim3(b), dim3(GB), 0, s, (const float*)xyz, n, (gbox*)partial);
    return hipGetLastError();
}

int f3d_graph_reduce_bbox(const void* partial_host, int nblocks, double lo[3], double hi[3]) {
    const gbox* p = (const gbox*)partial_host;
    unsigned long long bad = 0;
    for (int c = 0; c < 3; ++c) { lo[c] = INFINITY; hi[c] = -INFINITY; }
    for (int b = 0; b < nblocks; ++b) {
        for (int c = 0; c < 3; ++c) { lo[c] = fmin(lo[c], p[b].lo[c]); hi[c] = fmax(hi[c], p[b].hi[c]); }
        bad += p[b].bad;
    }
    return bad ? 1 : 0;
}

size_t f3d_graph_scratch_bytes(int64_t n, int64_t ncells) { return layout_for(n, ncells, temp_bytes_for(n)).total; }

// count pass after the grid is known: sort by cell, cell table, sorted copy, neighbour counts, exclusive scan into offsets[n + 1]
hipError_t f3d_launch_graph_count(const void* xyz, int dtype, int64_t n, const f3d_graphgrid& g, double r2, void* scratch,
                                  int64_t* offsets, hipStream_t s) {
    const int64_t ncells = (int64_t)g.dim[0] * g.dim[1] * g.dim[2];
    const size_t tb = temp_bytes_for(n);
    const graph_layout L = layout_for(n, ncells, tb);
    char* base = (char*)scratch;
    uint32_t *ka = (uint32_t*)(base + L.keys_a), *kb = (uint32_t*)(base + L.keys_b), *ia = (uint32_t*)(base + L.idx_a), *perm = (uint32_t*)(base + L.perm);
    double* sorted = (double*)(base + L.sorted);
    int2* cells = (int2*)(base + L.cells);
    const dim3 gr(grid_blocks(n)), b(GB);
    if (dtype == F3D_F64) hipLaunchKernelGGL(k_graph_keys<double>, gr, b, 0, s, (const double*)xyz, n, g, ka, ia);
    else hipLaunchKernelGGL(k_graph_keys<float>, gr, b, 0, s, (const float*)xyz, n, g, ka, ia);
    unsigned bits = 1; while (bits < 32 && ((int64_t)1 << bits) < ncells) ++bits;
    size_t t = tb;
    hipError_t e = rocprim::radix_sort_pairs(base + L.temp, t, ka, kb, ia, perm, (size_t)n, 0u, bits, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(cells, 0, (size_t)ncells * 8, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_graph_cells, gr, b, 0, s, kb, n, cells);
    if (dtype == F3D_F64) hipLaunchKernelGGL(k_graph_gather<double>, gr, b, 0, s, (const double*)xyz, n, perm, sorted);
    else hipLaunchKernelGGL(k_graph_gather<float>, gr, b, 0, s, (const float*)xyz, n, perm, sorted);
    hipLaunchKernelGGL(k_graph_scan<false>, gr, b, 0, s, sorted, n, perm, g, cells, r2, offsets, (int32_t*)nullptr);
    e = hipMemsetAsync(offsets + n, 0, 8, s);
    if (e != hipSuccess) return e;
    t = tb;
    e = rocprim::exclusive_scan(base + L.temp, t, offsets, offsets, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), s);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// fill pass: the scratch still holds the grid of the count pass
hipError_t f3d_launch_graph_fill(int64_t n, const f3d_graphgrid& g, double r2, const void* scratch, const int64_t* offsets,
                                 int32_t* nbrs, hipStream_t s) {
    const int64_t ncells = (int64_t)g.dim[0] * g.dim[1] * g.dim[2];
    const graph_layout L = layout_for(n, ncells, temp_bytes_for(n));
    const char* base = (const char*)scratch;
    hipLaunchKernelGGL(k_graph_scan<true>, dim3(grid_blocks(n)), dim3(GB), 0, s, (const double*)(base + L.sorted), n,
                       (const uint32_t*)(base + L.perm), g, (const int2*)(base + L.cells), r2, const_cast<int64_t*>(offsets), nbrs);
    return hipGetLastError();
}
