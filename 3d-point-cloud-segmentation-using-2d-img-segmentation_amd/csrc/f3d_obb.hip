// Support kernels of merge_bb's oriented-box fits (merge_intersecting_bb.py:72-76,86,122-128; get3DSeg.py:434) for gfx950.
//
// The reference fits a box on `points[ids == id]` for every instance (Open3D: convex hull -> PCA of the hull vertices).  At
// C5 scale (50M points, 4096 instances) 90 % of merge_bb's time was host work: grouping the points by instance id (a 50M
// argsort) and 4096 convex hulls of ~12k points each.  Here
//   f3d_group_by_id*     : the grouping -- key = id, stable radix sort of (key, index): `order` lists every instance's
//                          members in ascending point index, exactly the array `np.nonzero(ids == id)[0]`;
//   f3d_obb_extremes*    : per instance the members that are extreme along 26 fixed directions (+-x, +-y, +-z, the face and
//                          body diagonals);
//   f3d_obb_hull_filter* : drops every member that lies strictly inside the convex hull of those <= 26 points (facet
//                          equations from the host: a tiny Qhull per instance).  Such a point is interior to the hull of all
//                          members, so the hull -- and the box -- of the survivors (a few hundred points) is the same.
// Streaming, HBM-bound passes over the cloud; no MFMA (nothing here is a dense contraction).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "f3d.h"
#include "f3d_kernels.h"

namespace {

constexpr int OB = 256;

template <typename T>
__device__ __forceinline__ void load3(const T* __restrict__ xyz, int64_t i, double& x, double& y, double& z) {
    x = (double)xyz[3 * i]; y = (double)xyz[3 * i + 1]; z = (double)xyz[3 * i + 2];
}

__global__ __launch_bounds__(OB) void k_id_keys(const int64_t* __restrict__ ids, int64_t n, int64_t nids, uint32_t* __restrict__ keys,
                                                 uint32_t* __restrict__ idx) {
    for (int64_t i = (int64_t)blockIdx.x * OB + threadIdx.x; i < n; i += (int64_t)gridDim.x * OB) {
        const int64_t v = ids[i];
        keys[i] = (v >= 0 && v < nids) ? (uint32_t)v : (uint32_t)nids;     // ids outside [0, nids) share one bucket behind the others
        idx[i] = (uint32_t)i;
    }
}

// starts[k] = first position of key k in the sorted keys (lower bound), k = 0 .. nids + 1; starts[nids + 1] = n
__global__ __launch_bounds__(OB) void k_seg_starts(const uint32_t* __restrict__ keys, int64_t n, int64_t nids, int64_t* __restrict__ starts) {
    const int64_t k = (int64_t)blockIdx.x * OB + threadIdx.x;
    if (k > nids + 1) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    starts[k] = lo;
}

// orderable bits of a float (larger float <-> larger unsigned)
__device__ __forceinline__ uint32_t fbits(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// the 13 axes; direction 2k = +axis k, 2k + 1 = -axis k
__device__ __forceinline__ void dir_dots(float x, float y, float z, float d[13]) {
    d[0] = x; d[1] = y; d[2] = z;
    d[3] = x + y; d[4] = x - y; d[5] = x + z; d[6] = x - z; d[7] = y + z; d[8] = y - z;
    d[9] = (x + y) + z; d[10] = (x + y) - z; d[11] = (x - y) + z; d[12] = (x - y) - z;
}

// table[seg][dir] = max over the segment's members of (orderable dot value << 32 | position in `order`).  Any member works as an
// "extreme" (the filter only needs actual members), so float32 dots are enough.  A wave whose 64 positions belong to one segment
// reduces first and issues one atomic per direction; a wave that straddles a segment border falls back to per-lane atomics.
template <typename T>
__global__ __launch_bounds__(OB) void k_obb_extremes(const T* __restrict__ xyz, int64_t n, const int32_t* __restrict__ order,
                                                      const uint32_t* __restrict__ keys, int64_t nseg, unsigned long long* __restrict__ table) {
    const int64_t nwork = (n + 63) & ~(int64_t)63;                       // whole waves: every lane takes part in the shuffles
    for (int64_t i = (int64_t)blockIdx.x * OB + threadIdx.x; i < nwork; i += (int64_t)gridDim.x * OB) {
        const bool live = i < n;
        const uint32_t seg = live ? keys[i] : 0xFFFFFFFFu;
        float d[13];
        if (live) {
            double x, y, z;
            load3(xyz, (int64_t)order[i], x, y, z);
            dir_dots((float)x, (float)y, (float)z, d);
        } else {
            for (int k = 0; k < 13; ++k) d[k] = 0.f;
        }
        const uint32_t seg0 = __shfl(seg, 0, 64);
        const bool uniform = __all(seg == seg0);
        if (uniform) {
            if (seg0 >= (uint32_t)nseg) continue;                        // dead tail or the out-of-range bucket
#pragma unroll
            for (int k = 0; k < 13; ++k) {
#pragma unroll
                for (int sgn = 0; sgn < 2; ++sgn) {
                    unsigned long long key = ((unsigned long long)fbits(sgn ? -d[k] : d[k]) << 32) | (unsigned long long)(uint32_t)i;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) {
                        const unsigned long long o = __shfl_xor(key, off, 64);
                        key = o > key ? o : key;
                    }
                    if ((threadIdx.x & 63) == 0) atomicMax(&table[(size_t)seg0 * F3D_OBB_NDIR + 2 * k + sgn], key);
                }
            }
        } else if (live && seg < (uint32_t)nseg) {
#pragma unroll
            for (int k = 0; k < 13; ++k) {
                atomicMax(&table[(size_t)seg * F3D_OBB_NDIR + 2 * k], ((unsigned long long)fbits(d[k]) << 32) | (unsigned long long)(uint32_t)i);
                atomicMax(&table[(size_t)seg * F3D_OBB_NDIR + 2 * k + 1], ((unsigned long long)fbits(-d[k]) << 32) | (unsigned long long)(uint32_t)i);
            }
        }
    }
}

// extremes[seg][dir] = caller-order index of the extreme member, -1 for an empty segment
__global__ __launch_bounds__(OB) void k_obb_extremes_out(const unsigned long long* __restrict__ table, const int32_t* __restrict__ order,
                                                          int64_t count, int32_t* __restrict__ extremes) {
    const int64_t k = (int64_t)blockIdx.x * OB + threadIdx.x;
    if (k >= count) return;
    const unsigned long long t = table[k];
    extremes[k] = t ? order[(uint32_t)t] : -1;
}

// A member survives unless it lies strictly inside its segment's polytope: n_f . p + o_f < -margin[seg] for every facet f of
// facets[fstart[seg] .. fstart[seg + 1]).  A segment without facets keeps everything.  cand[starts[seg] + slot] = caller-order
// index, slots handed out by an atomic (the host sorts each short list).
template <typename T>
__global__ __launch_bounds__(OB) void k_obb_hull_filter(const T* __restrict__ xyz, int64_t n, const int32_t* __restrict__ order,
                                                         const uint32_t* __restrict__ keys, const int64_t* __restrict__ starts, int64_t nseg,
                                                         const int32_t* __restrict__ fstart, const double* __restrict__ facets,
                                                         const double* __restrict__ margin, int32_t* __restrict__ cand,
                                                         int32_t* __restrict__ cand_count) {
    for (int64_t i = (int64_t)blockIdx.x * OB + threadIdx.x; i < n; i += (int64_t)gridDim.x * OB) {
        const uint32_t seg = keys[i];
        if (seg >= (uint32_t)nseg) continue;
        const int32_t o = order[i];
        const int f0 = fstart[seg], f1 = fstart[seg + 1];
        bool inside = f1 > f0;
        if (inside) {
            double x, y, z;
            load3(xyz, (int64_t)o, x, y, z);
            const double mg = -margin[seg];
            for (int f = f0; f < f1 && inside; ++f) {
                const double* e = facets + 4 * (size_t)f;
                inside = (__builtin_fma(e[0], x, __builtin_fma(e[1], y, __builtin_fma(e[2], z, e[3]))) < mg);
            }
        }
        if (!inside) cand[starts[seg] + atomicAdd(&cand_count[seg], 1)] = o;
    }
}

// ---- the all-device candidate pipeline: survivors of the inner-hull test as a bit per position of `order`, then a stable compaction
// (ascending point index inside every instance: the order the reference's pcd_points[ids == id] has)
// facets: [nseg][F3D_OBB_SMALL_FACETS][4] with nfacets[seg] of them in use (0: the instance keeps every member)
template <typename T>
__global__ __launch_bounds__(OB) void k_obb_filter_mask(const T* __restrict__ xyz, int64_t n, const int32_t* __restrict__ order,
                                                         const uint32_t* __restrict__ keys, int64_t nseg, const int32_t* __restrict__ nfacets,
                                                         const double* __restrict__ facets, const double* __restrict__ margin,
                                                         unsigned long long* __restrict__ maskw) {
    const int64_t nwork = (n + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * OB + threadIdx.x; i < nwork; i += (int64_t)gridDim.x * OB) {
        bool keep = false;
        if (i < n) {
            const uint32_t seg = keys[i];
            if (seg < (uint32_t)nseg) {
                const int nf = nfacets[seg];
                bool inside = nf > 0;
                if (inside) {
                    double x, y, z;
                    load3(xyz, (int64_t)order[i], x, y, z);
                    const double mg = -margin[seg];
                    const double* e = facets + (size_t)seg * F3D_OBB_SMALL_FACETS * 4;
                    for (int f = 0; f < nf && inside; ++f, e += 4)
                        inside = (__builtin_fma(e[0], x, __builtin_fma(e[1], y, __builtin_fma(e[2], z, e[3]))) < mg);
                }
                keep = !inside;
            }
        }
        const unsigned long long m = __ballot(keep);
        if ((threadIdx.x & 63) == 0) maskw[i >> 6] = m;
    }
}

constexpr int SCAN_WORDS = 4096;             // mask words per block of the prefix pass (256 threads x 16)
// wordoff[w] = survivors in the words before w INSIDE its block of SCAN_WORDS words; blocksum[b] = the block's total
__global__ __launch_bounds__(OB) void k_mask_scan_local(const unsigned long long* __restrict__ maskw, int64_t nwords, uint32_t* __restrict__ wordoff,
                                                         uint32_t* __restrict__ blocksum) {
    __shared__ uint32_t part[OB];
    const int64_t w0 = (int64_t)blockIdx.x * SCAN_WORDS + (int64_t)threadIdx.x * 16;
    uint32_t c[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { c[k] = (w0 + k < nwords) ? (uint32_t)__popcll(maskw[w0 + k]) : 0u; sum += c[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < OB; off <<= 1) {
        const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < 16; ++k) { if (w0 + k < nwords) wordoff[w0 + k] = run; run += c[k]; }
    if (threadIdx.x == OB - 1) blocksum[blockIdx.x] = part[OB - 1];
}

// one block: blocksum -> exclusive prefix in place; total[0] = the number of survivors
__global__ __launch_bounds__(OB) void k_mask_scan_blocks(uint32_t* __restrict__ blocksum, int nblocks, int64_t* __restrict__ total) {
    __shared__ uint32_t part[OB];
    const int per = (nblocks + OB - 1) / OB;
    const int lo = threadIdx.x * per, hi = min(nblocks, lo + per);
    uint32_t sum = 0;
    for (int k = lo; k < hi; ++k) sum += blocksum[k];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < OB; off <<= 1) {
        const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (int k = lo; k < hi; ++k) { const uint32_t c = blocksum[k]; blocksum[k] = run; run += c; }
    if (threadIdx.x == OB - 1) *total = (int64_t)part[OB - 1];
}

__device__ __forceinline__ int64_t survivors_before(int64_t pos, int64_t n, const unsigned long long* __restrict__ maskw, const uint32_t* __restrict__ wordoff,
                                                    const uint32_t* __restrict__ blockoff, int64_t total) {
    if (pos >= n) return total;
    const int64_t w = pos >> 6;
    return (int64_t)blockoff[w / SCAN_WORDS] + wordoff[w] + __popcll(maskw[w] & ((1ull << (pos & 63)) - 1ull));
}

__global__ __launch_bounds__(OB) void k_obb_compact(int64_t n, const int32_t* __restrict__ order, const unsigned long long* __restrict__ maskw,
                                                     const uint32_t* __restrict__ wordoff, const uint32_t* __restrict__ blockoff, int32_t* __restrict__ cand) {
    for (int64_t i = (int64_t)blockIdx.x * OB + threadIdx.x; i < n; i += (int64_t)gridDim.x * OB) {
        const int64_t w = i >> 6;
        const unsigned long long m = maskw[w];
        if ((m >> (i & 63)) & 1ull) cand[(int64_t)blockoff[w / SCAN_WORDS] + wordoff[w] + __popcll(m & ((1ull << (i & 63)) - 1ull))] = order[i];
    }
}

__global__ __launch_bounds__(OB) void k_obb_cand_starts(int64_t n, const int64_t* __restrict__ starts, int64_t nseg, const unsigned long long* __restrict__ maskw,
                                                         const uint32_t* __restrict__ wordoff, const uint32_t* __restrict__ blockoff,
                                                         const int64_t* __restrict__ total, int64_t* __restrict__ cand_start) {
    const int64_t k = (int64_t)blockIdx.x * OB + threadIdx.x;
    if (k > nseg) return;
    cand_start[k] = survivors_before(starts[k], n, maskw, wordoff, blockoff, *total);     // starts[nseg] = first position of the out-of-range bucket
}

template <typename T>
__global__ __launch_bounds__(OB) void k_gather_points(const T* __restrict__ xyz, const int32_t* __restrict__ idx, int64_t count, double* __restrict__ out) {
    for (int64_t j = (int64_t)blockIdx.x * OB + threadIdx.x; j < count; j += (int64_t)gridDim.x * OB) {
        const int64_t i = idx[j];
        out[3 * j] = (double)xyz[3 * i]; out[3 * j + 1] = (double)xyz[3 * i + 1]; out[3 * j + 2] = (double)xyz[3 * i + 2];
    }
}

inline int grid_for(int64_t n, int cap) {
    int64_t g = (n + OB - 1) / OB;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

struct group_layout { size_t keys_in, idx_in, temp, temp_bytes, total; };
group_layout group_layout_for(int64_t n, unsigned bits) {
    group_layout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    L.keys_in = take((size_t)n * 4);
    L.idx_in = take((size_t)n * 4);
    size_t tb = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0u, bits,
                                    (hipStream_t)0);
    L.temp_bytes = tb;
    L.temp = take(tb);
    L.total = off;
    return L;
}

unsigned key_bits(int64_t nids) {
    unsigned b = 1;
    while (((int64_t)1 << b) <= nids) ++b;                               // keys 0 .. nids
    return b;
}

}  // namespace

size_t f3d_group_scratch_bytes(int64_t n, int64_t nids) { return group_layout_for(n < 1 ? 1 : n, key_bits(nids)).total; }

hipError_t f3d_launch_group_by_id(const int64_t* ids, int64_t n, int64_t nids, int32_t* order, uint32_t* sorted_keys, int64_t* starts,
                                  void* scratch, hipStream_t s) {
    if (n <= 0) {
        hipLaunchKernelGGL(k_seg_starts, dim3(grid_for(nids + 2, 65536)), dim3(OB), 0, s, sorted_keys, (int64_t)0, nids, starts);
        return hipGetLastError();
    }
    if (n > 0x7fffffffLL || nids < 0 || nids >= 0x7fffffffLL) return hipErrorInvalidValue;
    const unsigned bits = key_bits(nids);
    const group_layout L = group_layout_for(n, bits);
    char* base = reinterpret_cast<char*>(scratch);
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(base + L.keys_in);
    uint32_t* idx_in = reinterpret_cast<uint32_t*>(base + L.idx_in);
    hipLaunchKernelGGL(k_id_keys, dim3(grid_for(n, 8192)), dim3(OB), 0, s, ids, n, nids, keys_in, idx_in);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t tb = L.temp_bytes;
    e = rocprim::radix_sort_pairs(base + L.temp, tb, keys_in, sorted_keys, idx_in, reinterpret_cast<uint32_t*>(order), (size_t)n, 0u, bits, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_seg_starts, dim3(grid_for(nids + 2, 65536)), dim3(OB), 0, s, sorted_keys, n, nids, starts);
    return hipGetLastError();
}

hipError_t f3d_launch_obb_extremes(const void* xyz, int dtype, int64_t n, const int32_t* order, const uint32_t* sorted_keys, int64_t nseg,
                                   unsigned long long* table, int32_t* extremes, hipStream_t s) {
    if (nseg <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(table, 0, (size_t)nseg * F3D_OBB_NDIR * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    if (n > 0) {
        const dim3 g(grid_for(n, 8192)), b(OB);
        if (dtype == F3D_F64) hipLaunchKernelGGL(k_obb_extremes<double>, g, b, 0, s, (const double*)xyz, n, order, sorted_keys, nseg, table);
        else hipLaunchKernelGGL(k_obb_extremes<float>, g, b, 0, s, (const float*)xyz, n, order, sorted_keys, nseg, table);
    }
    hipLaunchKernelGGL(k_obb_extremes_out, dim3(grid_for(nseg * F3D_OBB_NDIR, 65536)), dim3(OB), 0, s, table, order, nseg * F3D_OBB_NDIR, extremes);
    return hipGetLastError();
}

hipError_t f3d_launch_obb_hull_filter(const void* xyz, int dtype, int64_t n, const int32_t* order, const uint32_t* sorted_keys,
                                      const int64_t* starts, int64_t nseg, const int32_t* fstart, const double* facets, const double* margin,
                                      int32_t* cand, int32_t* cand_count, hipStream_t s) {
    if (nseg <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(cand_count, 0, (size_t)nseg * sizeof(int32_t), s);
    if (e != hipSuccess || n <= 0) return e;
    const dim3 g(grid_for(n, 8192)), b(OB);
    if (dtype == F3D_F64)
        hipLaunchKernelGGL(k_obb_hull_filter<double>, g, b, 0, s, (const double*)xyz, n, order, sorted_keys, starts, nseg, fstart, facets, margin, cand, cand_count);
    else
        hipLaunchKernelGGL(k_obb_hull_filter<float>, g, b, 0, s, (const float*)xyz, n, order, sorted_keys, starts, nseg, fstart, facets, margin, cand, cand_count);
    return hipGetLastError();
}

// ---- candidates of every instance without a host round trip -----------------------------------------------------------------------
namespace {
struct cand_layout { size_t extremes, gathered, isvert, facets, nfacets, margin, maskw, wordoff, blocksum, total, bytes; int64_t nwords; int nblocks; };
cand_layout cand_layout_for(int64_t n, int64_t nids) {
    cand_layout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    L.nwords = (n + 63) / 64;
    L.nblocks = (int)((L.nwords + SCAN_WORDS - 1) / SCAN_WORDS);
    L.extremes = take((size_t)nids * F3D_OBB_NDIR * 4);
    L.gathered = take((size_t)nids * F3D_OBB_NDIR * 24);
    L.isvert = take((size_t)nids * F3D_OBB_NDIR);
    L.facets = take((size_t)nids * F3D_OBB_SMALL_FACETS * 32);
    L.nfacets = take((size_t)nids * 4);
    L.margin = take((size_t)nids * 8);
    L.maskw = take((size_t)L.nwords * 8);
    L.wordoff = take((size_t)L.nwords * 4);
    L.blocksum = take((size_t)(L.nblocks + 1) * 4);
    L.total = take(8);
    L.bytes = off;
    return L;
}
}  // namespace

size_t f3d_obb_candidates_scratch_bytes(int64_t n, int64_t nids) { return cand_layout_for(n < 1 ? 1 : n, nids < 1 ? 1 : nids).bytes; }

hipError_t f3d_launch_obb_candidates(const void* xyz, int dtype, int64_t n, const int32_t* order, const uint32_t* sorted_keys, const int64_t* starts,
                                     int64_t nids, int min_members, unsigned long long* table, void* scratch, int32_t* cand, int64_t* cand_start,
                                     hipStream_t s) {
    if (nids <= 0) return hipSuccess;
    if (n <= 0) return hipMemsetAsync(cand_start, 0, (size_t)(nids + 1) * 8, s);
    const cand_layout L = cand_layout_for(n, nids);
    char* base = reinterpret_cast<char*>(scratch);
    int32_t* extremes = reinterpret_cast<int32_t*>(base + L.extremes);
    double* gathered = reinterpret_cast<double*>(base + L.gathered);
    uint8_t* isvert = reinterpret_cast<uint8_t*>(base + L.isvert);
    double* facets = reinterpret_cast<double*>(base + L.facets);
    int32_t* nfacets = reinterpret_cast<int32_t*>(base + L.nfacets);
    double* margin = reinterpret_cast<double*>(base + L.margin);
    unsigned long long* maskw = reinterpret_cast<unsigned long long*>(base + L.maskw);
    uint32_t* wordoff = reinterpret_cast<uint32_t*>(base + L.wordoff);
    uint32_t* blocksum = reinterpret_cast<uint32_t*>(base + L.blocksum);
    int64_t* total = reinterpret_cast<int64_t*>(base + L.total);
    hipError_t e = f3d_launch_obb_extremes(xyz, dtype, n, order, sorted_keys, nids, table, extremes, s);
    if (e != hipSuccess) return e;
    if ((e = f3d_launch_obb_small_hulls(xyz, dtype, extremes, starts, nids, min_members, gathered, isvert, facets, nfacets, margin, s)) != hipSuccess) return e;
    const dim3 g(grid_for(n, 8192)), b(OB);
    if (dtype == F3D_F64) hipLaunchKernelGGL(k_obb_filter_mask<double>, g, b, 0, s, (const double*)xyz, n, order, sorted_keys, nids, nfacets, facets, margin, maskw);
    else hipLaunchKernelGGL(k_obb_filter_mask<float>, g, b, 0, s, (const float*)xyz, n, order, sorted_keys, nids, nfacets, facets, margin, maskw);
    hipLaunchKernelGGL(k_mask_scan_local, dim3(L.nblocks), b, 0, s, maskw, L.nwords, wordoff, blocksum);
    hipLaunchKernelGGL(k_mask_scan_blocks, dim3(1), b, 0, s, blocksum, L.nblocks, total);
    hipLaunchKernelGGL(k_obb_compact, g, b, 0, s, n, order, maskw, wordoff, blocksum, cand);
    hipLaunchKernelGGL(k_obb_cand_starts, dim3(grid_for(nids + 1, 65536)), b, 0, s, n, starts, nids, maskw, wordoff, blocksum, total, cand_start);
    return hipGetLastError();
}

hipError_t f3d_launch_gather_points(const void* xyz, int dtype, const int32_t* idx, int64_t count, double* out, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    const dim3 g(grid_for(count, 8192)), b(OB);
    if (dtype == F3D_F64) hipLaunchKernelGGL(k_gather_points<double>, g, b, 0, s, (const double*)xyz, idx, count, out);
    else hipLaunchKernelGGL(k_gather_points<float>, g, b, 0, s, (const float*)xyz, idx, count, out);
    return hipGetLastError();
}
