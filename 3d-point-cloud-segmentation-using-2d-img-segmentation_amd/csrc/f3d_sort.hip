// Cell sort of a point cloud: order the points by coarse grid cell (HBM-bound, key-index radix sort).
//
// Why it exists: the fused kernel assigns 64 consecutive points to a wavefront.  When consecutive
// points are spatial neighbours, a wave is (almost always) entirely inside or entirely outside a
// view's frustum, so the expensive projection runs only for waves that have visible points.  A
// cloud in arbitrary order wastes ~60 % of the projection work on masked-off lanes.  Sorting changes
// nothing in the results: every point's label depends on its own xyz only, and the kernel writes it
// back to the caller's index through `perm`.
//
//   k_bbox_partial / k_bbox_final : finite bounding box of a strided sample of the cloud -> cubic cell grid, <= 32 per axis
//   k_cell_keys    : key[i] = 15-bit Morton code of point i's cell, idx[i] = i  (streaming)
//   rocprim::radix_sort_pairs on the 15/16 key bits (2 x 8-bit onesweep passes over 8 B/point)
//   k_gather_xyz   : sorted[j] = xyz[perm[j]]     (only for the "prepared layout" entry point; the
//                    in-step sort lets the fused kernel read xyz through perm instead)
//
// A first version used one returning global atomic per point (counting sort): 64 lanes hitting 64
// random counters made both its count and its scatter kernel ~0.5 ms each at 10M points, 4x the
// whole radix sort.  The order of equal keys is the radix sort's (stable), so perm is deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "f3d.h"
#include "f3d_kernels.h"

namespace {

constexpr int SB = 256;
#ifndef F3D_SORT_KEY_BITS
#define F3D_SORT_KEY_BITS 16                 // total key bits = radix passes x 8
#endif
#if F3D_SORT_KEY_BITS <= 16
typedef uint16_t sort_key_t;                 // 2-byte keys: the radix passes move 6 B per point instead of 8
#else
typedef uint32_t sort_key_t;
#endif

struct bbox6 { double lo[3], hi[3]; };

template <typename T>
__global__ __launch_bounds__(SB) void k_bbox_partial(const T* __restrict__ xyz, int64_t n, int64_t stride, bbox6* __restrict__ partial) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    // a strided sample is enough: points outside the sampled box are clamped into the border cells, which can only cost
    // a little coherence, never correctness
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

// one block: reduce the partial boxes, derive the cell grid, zero the cell counters
__global__ __launch_bounds__(SB) void k_bbox_final(const bbox6* __restrict__ partial, int nparts, f3d_cellgrid* __restrict__ grid,
                                                    int max_cells) {
    __shared__ double sl[3][SB], sh[3][SB];
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int k = threadIdx.x; k < nparts; k += SB)
        for (int c = 0; c < 3; ++c) { lo[c] = fmin(lo[c], partial[k].lo[c]); hi[c] = fmax(hi[c], partial[k].hi[c]); }
    for (int c = 0; c < 3; ++c) { sl[c][threadIdx.x] = lo[c]; sh[c][threadIdx.x] = hi[c]; }
    __syncthreads();
    for (int s = SB / 2; s >= 1; s >>= 1) {
        if (threadIdx.x < s)
            for (int c = 0; c < 3; ++c) {
                sl[c][threadIdx.x] = fmin(sl[c][threadIdx.x], sl[c][threadIdx.x + s]);
                sh[c][threadIdx.x] = fmax(sh[c][threadIdx.x], sh[c][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        f3d_cellgrid g;
        double ext[3], vol = 1.0;
        for (int c = 0; c < 3; ++c) {
            g.lo[c] = sl[c][0] <= sh[c][0] ? sl[c][0] : 0.0;
            ext[c] = sl[c][0] <= sh[c][0] ? (sh[c][0] - sl[c][0]) : 0.0;
            if (!(ext[c] > 1e-12)) ext[c] = 1e-12;
            vol *= ext[c];
        }
        // F3D_SORT_KEY_BITS key bits are dealt to the axes one at a time, always to the axis whose cells are currently the
        // longest, so the cells come out as cubic as the box allows; the key interleaves the axes' bits from the most
        // significant level down (a Morton code with per-axis bit counts), so a contiguous run of sorted points is a
        // compact 3-D block (small footprint in every view's mask, good for gather coalescing and the per-XCD L2).
        (void)vol; (void)max_cells;
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
        *grid = g;
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

template <typename T>
__global__ __launch_bounds__(SB) void k_cell_keys(const T* __restrict__ xyz, int64_t n, const f3d_cellgrid* __restrict__ grid,
                                                   sort_key_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const f3d_cellgrid g = *grid;
    for (int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x; i < n; i += (int64_t)gridDim.x * SB) {
        keys[i] = (sort_key_t)cell_of(xyz + 3 * i, g);
        idx[i] = (uint32_t)i;
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
    size_t grid, partial, keys_in, keys_out, idx_in, temp, total, temp_bytes;
};

sort_layout layout_for(int64_t n) {
    sort_layout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    L.grid = take(sizeof(f3d_cellgrid));
    L.partial = take(1024 * sizeof(bbox6));
    L.keys_in = take((size_t)n * sizeof(sort_key_t));
    L.keys_out = take((size_t)n * sizeof(sort_key_t));
    L.idx_in = take((size_t)n * 4);
    size_t tb = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tb, (sort_key_t*)nullptr, (sort_key_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    (size_t)n, 0u, (unsigned)F3D_SORT_KEY_BITS, (hipStream_t)0);
    L.temp_bytes = tb;
    L.temp = take(tb);
    L.total = off;
    return L;
}

}  // namespace

size_t f3d_sort_scratch_bytes(int64_t n) { return layout_for(n < 1 ? 1 : n).total; }

hipError_t f3d_launch_cell_sort(const void* xyz, int dtype, int64_t n, void* sorted_xyz, int32_t* perm, void* scratch, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n > 0x7fffffffLL) return hipErrorInvalidValue;
    const sort_layout L = layout_for(n);
    char* base = reinterpret_cast<char*>(scratch);
    f3d_cellgrid* grid = reinterpret_cast<f3d_cellgrid*>(base + L.grid);
    bbox6* partial = reinterpret_cast<bbox6*>(base + L.partial);
    sort_key_t* keys_in = reinterpret_cast<sort_key_t*>(base + L.keys_in);
    sort_key_t* keys_out = reinterpret_cast<sort_key_t*>(base + L.keys_out);
    uint32_t* idx_in = reinterpret_cast<uint32_t*>(base + L.idx_in);
    const int64_t gb = (n + SB - 1) / SB;
    const int64_t stride = n > (1 << 18) ? n >> 18 : 1;          // inspect <= ~262k points for the bounding box
    const int64_t sb = ((n + stride - 1) / stride + SB - 1) / SB;
    const int nparts = (int)(sb < 1024 ? sb : 1024);
    const int gstream = (int)(gb < 8192 ? gb : 8192);
    if (dtype == F3D_F64) {
        hipLaunchKernelGGL(k_bbox_partial<double>, dim3(nparts), dim3(SB), 0, s, (const double*)xyz, n, stride, partial);
        hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(SB), 0, s, partial, nparts, grid, F3D_SORT_MAX_CELLS);
        hipLaunchKernelGGL(k_cell_keys<double>, dim3(gstream), dim3(SB), 0, s, (const double*)xyz, n, grid, keys_in, idx_in);
    } else {
        hipLaunchKernelGGL(k_bbox_partial<float>, dim3(nparts), dim3(SB), 0, s, (const float*)xyz, n, stride, partial);
        hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(SB), 0, s, partial, nparts, grid, F3D_SORT_MAX_CELLS);
        hipLaunchKernelGGL(k_cell_keys<float>, dim3(gstream), dim3(SB), 0, s, (const float*)xyz, n, grid, keys_in, idx_in);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t tb = L.temp_bytes;
    e = rocprim::radix_sort_pairs(base + L.temp, tb, keys_in, keys_out, idx_in, reinterpret_cast<uint32_t*>(perm), (size_t)n, 0u, (unsigned)F3D_SORT_KEY_BITS, s);
    if (e != hipSuccess) return e;
    if (sorted_xyz) {
        if (dtype == F3D_F64) hipLaunchKernelGGL(k_gather_xyz<double>, dim3(gstream), dim3(SB), 0, s, (const double*)xyz, n, perm, (double*)sorted_xyz);
        else hipLaunchKernelGGL(k_gather_xyz<float>, dim3(gstream), dim3(SB), 0, s, (const float*)xyz, n, perm, (float*)sorted_xyz);
    }
    return hipGetLastError();
}
