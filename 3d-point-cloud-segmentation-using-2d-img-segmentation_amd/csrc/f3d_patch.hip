// Row a5: the patch matching of Fusion.fuse (Fusion3DSeg/fusion.py:269-298) as a data-parallel ownership problem.
//
// The reference walks the in-frustum points ("seeds") of the fused cloud in index order; seed k looks at the depth
// pixels of the (2*half+1)^2 window around its projection that are still free, keeps those within `radius` and
// within the normal angle of ITS OWN position/normal as they were before this frame, and takes them.  A seed's test
// never depends on what other seeds did, so the outcome is: a free pixel belongs to the FIRST seed (lowest k) whose
// window covers it and whose test accepts it.  One thread per pixel scans the seeds bucketed by projected pixel.
//
//   k_patch_count / scan / k_patch_fill : seeds whose projection is inside the image -> per-pixel buckets (any order inside a
//                                         bucket: the owner is a minimum); the others go to a short "odd" list
//   k_patch_owner                       : owner[p] = min k over covering, accepting seeds, -1 if none
//
// The test is NumPy's, operation for operation, as evaluated by the build this repository pins (tests/golden/fuse.npz):
//   np.linalg.norm(points - ds[None, :], axis=-1) < radius   ->  sqrt((t0*t0 + t1*t1) + t2*t2) < radius
//   np.einsum('ij, j -> i', normals, ds_normal) > min_cosine ->  (n0*s0 + n2*s2) + n1*s1 > min_cosine
// A seed projecting outside the image keeps the reference's window arithmetic, Python slice semantics included
// (start = max(0, c - half), stop = c + half + 1, a negative stop counts from the end).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include "f3d.h"
#include "f3d_kernels.h"

namespace {

constexpr int PB = 256;

struct patch_args { int h, w, half; double radius, min_cosine; };

__device__ __forceinline__ bool accepts(const double* __restrict__ spt, const double* __restrict__ snr, int64_t k,
                                        double qx, double qy, double qz, double nx, double ny, double nz, const patch_args& a) {
    const double t0 = qx - spt[3 * k], t1 = qy - spt[3 * k + 1], t2 = qz - spt[3 * k + 2];
    const double dist = sqrt((t0 * t0 + t1 * t1) + t2 * t2);
    const double cs = (nx * snr[3 * k] + nz * snr[3 * k + 2]) + ny * snr[3 * k + 1];
    return (dist < a.radius) & (cs > a.min_cosine);
}

__global__ __launch_bounds__(PB) void k_patch_count(const int32_t* __restrict__ uv, int64_t m, patch_args a, int32_t* __restrict__ count,
                                                     int32_t* __restrict__ odd, int32_t* __restrict__ nodd) {
    for (int64_t k = (int64_t)blockIdx.x * PB + threadIdx.x; k < m; k += (int64_t)gridDim.x * PB) {
        const int u = uv[k], v = uv[m + k];
        if (u >= 0 && u < a.w && v >= 0 && v < a.h) atomicAdd(&count[v * a.w + u], 1);
        else odd[atomicAdd(nodd, 1)] = (int32_t)k;
    }
}

__global__ __launch_bounds__(PB) void k_patch_fill(const int32_t* __restrict__ uv, int64_t m, patch_args a, const int32_t* __restrict__ start,
                                                    int32_t* __restrict__ cursor, int32_t* __restrict__ bucket) {
    for (int64_t k = (int64_t)blockIdx.x * PB + threadIdx.x; k < m; k += (int64_t)gridDim.x * PB) {
        const int u = uv[k], v = uv[m + k];
        if (u >= 0 && u < a.w && v >= 0 && v < a.h) {
            const int pix = v * a.w + u;
            bucket[start[pix] + atomicAdd(&cursor[pix], 1)] = (int32_t)k;
        }
    }
}

// [start, stop) of the reference's slice max(0, c - half) : c + half + 1 over an axis of length n
__device__ __forceinline__ void py_slice(int c, int half, int n, int& lo, int& hi) {
    long long s = (long long)c - half; if (s < 0) s = 0;
    long long e = (long long)c + half + 1;
    if (e < 0) { e += n; if (e < 0) e = 0; }
    if (s > n) s = n;
    if (e > n) e = n;
    lo = (int)s; hi = (int)e;
}

__global__ __launch_bounds__(PB) void k_patch_owner(const int32_t* __restrict__ uv, int64_t m, patch_args a, const int32_t* __restrict__ start,
                                                     const int32_t* __restrict__ bucket, const int32_t* __restrict__ odd,
                                                     const int32_t* __restrict__ nodd, const double* __restrict__ spt,
                                                     const double* __restrict__ snr, const double* __restrict__ qpt,
                                                     const double* __restrict__ qnr, const uint8_t* __restrict__ free_px,
                                                     int32_t* __restrict__ owner) {
    const int64_t npx = (int64_t)a.h * a.w;
    const int n_odd = *nodd;
    for (int64_t p = (int64_t)blockIdx.x * PB + threadIdx.x; p < npx; p += (int64_t)gridDim.x * PB) {
        int best = 0x7fffffff;
        if (free_px[p]) {
            const int v = (int)(p / a.w), u = (int)(p - (int64_t)v * a.w);
            const double qx = qpt[3 * p], qy = qpt[3 * p + 1], qz = qpt[3 * p + 2];
            const double nx = qnr[3 * p], ny = qnr[3 * p + 1], nz = qnr[3 * p + 2];
            // seeds projecting at (su, sv) cover this pixel when |su - u| <= half and |sv - v| <= half
            const int v0 = max(0, v - a.half), v1 = min(a.h - 1, v + a.half), u0 = max(0, u - a.half), u1 = min(a.w - 1, u + a.half);
            for (int sv = v0; sv <= v1; ++sv) {
                const int row = sv * a.w;
                for (int s = start[row + u0]; s < start[row + u1 + 1]; ++s) {      // the row's buckets are contiguous
                    const int k = bucket[s];
                    if (k < best && accepts(spt, snr, k, qx, qy, qz, nx, ny, nz, a)) best = k;
                }
            }
            for (int j = 0; j < n_odd; ++j) {
                const int k = odd[j];
                if (k >= best) continue;
                int r0, r1, c0, c1;
                py_slice(uv[m + k], a.half, a.h, r0, r1);
                py_slice(uv[k], a.half, a.w, c0, c1);
                if (v >= r0 && v < r1 && u >= c0 && u < c1 && accepts(spt, snr, k, qx, qy, qz, nx, ny, nz, a)) best = k;
            }
        }
        owner[p] = best == 0x7fffffff ? -1 : best;
    }
}

// ---- patch_downsample (fusion.py:134-210): which pixels become seeds.  Pixel p (visited at position prio[p] of the shuffled
// order) is a seed iff it is still free then, i.e. iff no EARLIER seed whose window covers it accepts it.  Resolved in rounds:
// an undecided pixel becomes "claimed" as soon as one earlier accepting neighbour is known to be a seed, and "seed" once all
// of them are known not to be; the earliest undecided pixel always resolves, typical depth is a handful of rounds.
enum { PD_UNKNOWN = 0, PD_SEED = 1, PD_CLAIMED = 2, PD_NOT_FREE = 3 };

__device__ __forceinline__ bool accepts_px(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t seed, int64_t cand,
                                           const patch_args& a) {
    // criterion(ds = seed, points = candidates): candidate minus seed, candidate normal . seed normal
    return accepts(pts, nrm, seed, pts[3 * cand], pts[3 * cand + 1], pts[3 * cand + 2], nrm[3 * cand], nrm[3 * cand + 1], nrm[3 * cand + 2], a);
}

__global__ __launch_bounds__(PB) void k_pd_init(const uint8_t* __restrict__ free_px, int64_t npx, int32_t* __restrict__ status) {
    for (int64_t p = (int64_t)blockIdx.x * PB + threadIdx.x; p < npx; p += (int64_t)gridDim.x * PB) status[p] = free_px[p] ? PD_UNKNOWN : PD_NOT_FREE;
}

__global__ __launch_bounds__(PB) void k_pd_round(const double* __restrict__ pts, const double* __restrict__ nrm, const int32_t* __restrict__ prio,
                                                  patch_args a, int32_t* status, int32_t* __restrict__ unknown_left) {
    const int64_t npx = (int64_t)a.h * a.w;
    for (int64_t p = (int64_t)blockIdx.x * PB + threadIdx.x; p < npx; p += (int64_t)gridDim.x * PB) {
        if (__atomic_load_n(&status[p], __ATOMIC_RELAXED) != PD_UNKNOWN) continue;
        const int v = (int)(p / a.w), u = (int)(p - (int64_t)v * a.w), mine = prio[p];
        const int v0 = max(0, v - a.half), v1 = min(a.h - 1, v + a.half), u0 = max(0, u - a.half), u1 = min(a.w - 1, u + a.half);
        bool claimed = false, waiting = false;
        for (int sv = v0; sv <= v1 && !claimed; ++sv)
            for (int su = u0; su <= u1; ++su) {
                const int64_t q = (int64_t)sv * a.w + su;
                if (prio[q] >= mine) continue;                                   // only earlier pixels can have taken p
                const int st = __atomic_load_n(&status[q], __ATOMIC_RELAXED);
                if (st == PD_CLAIMED || st == PD_NOT_FREE) continue;             // never a seed
                if (!accepts_px(pts, nrm, q, p, a)) continue;
                if (st == PD_SEED) { claimed = true; break; }
                waiting = true;                                                  // q undecided: it may still turn out to be a seed
            }
        if (claimed) __atomic_store_n(&status[p], PD_CLAIMED, __ATOMIC_RELAXED);
        else if (!waiting) __atomic_store_n(&status[p], PD_SEED, __ATOMIC_RELAXED);
        else atomicAdd(unknown_left, 1);
    }
}

// owner[p]: the seed (pixel index) that takes pixel p = the earliest seed that covers and accepts it; a seed takes itself when it
// accepts itself; -1 for pixels nobody takes
__global__ __launch_bounds__(PB) void k_pd_owner(const double* __restrict__ pts, const double* __restrict__ nrm, const int32_t* __restrict__ prio,
                                                  patch_args a, const int32_t* __restrict__ status, int32_t* __restrict__ owner) {
    const int64_t npx = (int64_t)a.h * a.w;
    for (int64_t p = (int64_t)blockIdx.x * PB + threadIdx.x; p < npx; p += (int64_t)gridDim.x * PB) {
        int best = -1, best_prio = 0x7fffffff;
        const int st = status[p];
        if (st == PD_SEED || st == PD_CLAIMED) {
            const int v = (int)(p / a.w), u = (int)(p - (int64_t)v * a.w);
            const int v0 = max(0, v - a.half), v1 = min(a.h - 1, v + a.half), u0 = max(0, u - a.half), u1 = min(a.w - 1, u + a.half);
            for (int sv = v0; sv <= v1; ++sv)
                for (int su = u0; su <= u1; ++su) {
                    const int64_t q = (int64_t)sv * a.w + su;
                    if (status[q] != PD_SEED || prio[q] >= best_prio) continue;
                    if (accepts_px(pts, nrm, q, p, a)) { best = (int)q; best_prio = prio[q]; }
                }
        }
        owner[p] = best;
    }
}

// ---- ordered per-seed sums (fusion.py:195-201, 289-298).  The reference stacks a seed's accepted pixels in window order
// (row-major = ascending pixel index) and takes np.mean, i.e. adds the rows one after the other: ((r0 + r1) + r2) + ...  One
// thread per seed walks its window in that order and adds the rows of the pixels it owns -- the same additions in the same
// order, so the means come out bit for bit.  FUSE: seed k sits at its projection uv[:, k] (Python slice window); otherwise a
// seed is a pixel that owns itself (patch_downsample) and the window is clamped to the image.
template <bool FUSE>
__global__ __launch_bounds__(PB) void k_patch_sums(const int32_t* __restrict__ owner, const int32_t* __restrict__ uv, int64_t m, patch_args a,
                                                    const double* __restrict__ ra, const double* __restrict__ rb, const double* __restrict__ rc,
                                                    double* __restrict__ sums, int32_t* __restrict__ counts) {
    for (int64_t k = (int64_t)blockIdx.x * PB + threadIdx.x; k < m; k += (int64_t)gridDim.x * PB) {
        int r0, r1, c0, c1;
        if (FUSE) {
            py_slice(uv[m + k], a.half, a.h, r0, r1);
            py_slice(uv[k], a.half, a.w, c0, c1);
        } else {
            if (owner[k] != (int32_t)k) { counts[k] = 0; continue; }
            const int v = (int)(k / a.w), u = (int)(k - (int64_t)v * a.w);
            r0 = max(0, v - a.half); r1 = min(a.h, v + a.half + 1); c0 = max(0, u - a.half); c1 = min(a.w, u + a.half + 1);
        }
        double acc[9];
        int n = 0;
        for (int y = r0; y < r1; ++y)
            for (int x = c0; x < c1; ++x) {
                const int64_t p = (int64_t)y * a.w + x;
                if (owner[p] != (int32_t)k) continue;
                if (n == 0) {
                    for (int c = 0; c < 3; ++c) { acc[c] = ra ? ra[3 * p + c] : 0.0; acc[3 + c] = rb ? rb[3 * p + c] : 0.0; acc[6 + c] = rc ? rc[3 * p + c] : 0.0; }
                } else {
                    for (int c = 0; c < 3; ++c) {
                        if (ra) acc[c] += ra[3 * p + c];
                        if (rb) acc[3 + c] += rb[3 * p + c];
                        if (rc) acc[6 + c] += rc[3 * p + c];
                    }
                }
                ++n;
            }
        counts[k] = n;
        if (n) for (int c = 0; c < 9; ++c) sums[9 * k + c] = acc[c];
    }
}

inline int blocks_for(int64_t n) { int64_t b = (n + PB - 1) / PB; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

struct patch_layout { size_t count, start, cursor, bucket, odd, nodd, temp, total; };

size_t scan_temp(int64_t npx) {
    size_t b = 0;
    (void)rocprim::exclusive_scan(nullptr, b, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, (size_t)npx + 1, rocprim::plus<int32_t>());
    return b + 256;
}

patch_layout layout_for(int64_t npx, int64_t m) {
    patch_layout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    L.count = take((size_t)(npx + 1) * 4); L.start = take((size_t)(npx + 1) * 4); L.cursor = take((size_t)(npx + 1) * 4);
    L.bucket = take((size_t)m * 4 + 4); L.odd = take((size_t)m * 4 + 4); L.nodd = take(256); L.temp = take(scan_temp(npx));
    L.total = o;
    return L;
}

}  // namespace

size_t f3d_patch_scratch_bytes(int h, int w, int64_t m) { return layout_for((int64_t)h * w, m).total; }

hipError_t f3d_launch_patch_owner(const int32_t* uv, int64_t m, int h, int w, int half, double radius, double min_cosine,
                                  const double* seed_pts, const double* seed_nrm, const double* q_pts, const double* q_nrm,
                                  const uint8_t* free_px, int32_t* owner, void* scratch, hipStream_t s) {
    const int64_t npx = (int64_t)h * w;
    if (npx <= 0) return hipSuccess;
    const patch_layout L = layout_for(npx, m);
    char* base = (char*)scratch;
    int32_t *count = (int32_t*)(base + L.count), *start = (int32_t*)(base + L.start), *cursor = (int32_t*)(base + L.cursor),
            *bucket = (int32_t*)(base + L.bucket), *odd = (int32_t*)(base + L.odd), *nodd = (int32_t*)(base + L.nodd);
    patch_args a; a.h = h; a.w = w; a.half = half; a.radius = radius; a.min_cosine = min_cosine;
    hipError_t e = hipMemsetAsync(base + L.count, 0, (size_t)(npx + 1) * 4, s);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(base + L.cursor, 0, (size_t)(npx + 1) * 4, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(nodd, 0, 4, s)) != hipSuccess) return e;
    if (m > 0) hipLaunchKernelGGL(k_patch_count, dim3(blocks_for(m)), dim3(PB), 0, s, uv, m, a, count, odd, nodd);
    size_t t = scan_temp(npx);
    e = rocprim::exclusive_scan(base + L.temp, t, count, start, (int32_t)0, (size_t)npx + 1, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) return e;
    if (m > 0) hipLaunchKernelGGL(k_patch_fill, dim3(blocks_for(m)), dim3(PB), 0, s, uv, m, a, start, cursor, bucket);
    hipLaunchKernelGGL(k_patch_owner, dim3(blocks_for(npx)), dim3(PB), 0, s, uv, m, a, start, bucket, odd, nodd, seed_pts, seed_nrm, q_pts,
                       q_nrm, free_px, owner);
    return hipGetLastError();
}

// patch_downsample: status (seed / claimed / not free) by rounds, then the owners.  `status` and `counter` are device scratch;
// returns through *rounds the number of passes it took (diagnostic).
hipError_t f3d_launch_patch_seeds(const double* pts, const double* nrm, const int32_t* prio, const uint8_t* free_px, int h, int w, int half,
                                  double radius, double min_cosine, int32_t* status, int32_t* owner, int32_t* counter, int* rounds,
                                  hipStream_t s) {
    const int64_t npx = (int64_t)h * w;
    *rounds = 0;
    if (npx <= 0) return hipSuccess;
    patch_args a; a.h = h; a.w = w; a.half = half; a.radius = radius; a.min_cosine = min_cosine;
    const dim3 g(blocks_for(npx)), b(PB);
    hipLaunchKernelGGL(k_pd_init, g, b, 0, s, free_px, npx, status);
    for (int r = 0; r < (int)(npx < 1000000 ? npx + 2 : 1000002); ++r) {        // every pass resolves at least the earliest undecided pixel
        hipError_t e = hipMemsetAsync(counter, 0, 4, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_pd_round, g, b, 0, s, pts, nrm, prio, a, status, counter);
        int32_t left = 0;
        if ((e = hipMemcpyAsync(&left, counter, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        *rounds = r + 1;
        if (left == 0) break;
    }
    hipLaunchKernelGGL(k_pd_owner, g, b, 0, s, pts, nrm, prio, a, status, owner);
    return hipGetLastError();
}

hipError_t f3d_launch_patch_sums(const int32_t* owner, const int32_t* uv, int64_t m, int h, int w, int half, const double* rows_a,
                                 const double* rows_b, const double* rows_c, double* sums, int32_t* counts, hipStream_t s) {
    if (m <= 0) return hipSuccess;
    patch_args a; a.h = h; a.w = w; a.half = half; a.radius = 0; a.min_cosine = 0;
    if (uv) hipLaunchKernelGGL(k_patch_sums<true>, dim3(blocks_for(m)), dim3(PB), 0, s, owner, uv, m, a, rows_a, rows_b, rows_c, sums, counts);
    else hipLaunchKernelGGL(k_patch_sums<false>, dim3(blocks_for(m)), dim3(PB), 0, s, owner, uv, m, a, rows_a, rows_b, rows_c, sums, counts);
    return hipGetLastError();
}
