// a12 / (f)#4: the remaining vectorised primitives of the reference's Fusion3DSeg/intersections.py (:6-143, 167-204),
// one thread per output element.  None of them is called by the reference's entry points; they are provided so that the
// module is complete.  fp64, products and sums rounded separately (-ffp-contract=off), 3-term dots left to right except
// the 'nmc,mc->mn' einsum of point_inside_polygon, which keeps the (d0 e0 + d2 e2) + d1 e1 order of point_inside_polyhedra.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "f3d.h"
#include "f3d_kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int GB = 256;
struct v3 { double x, y, z; };
__device__ __forceinline__ v3 ldv(const double* p) { v3 r; r.x = p[0]; r.y = p[1]; r.z = p[2]; return r; }
__device__ __forceinline__ v3 sub(v3 a, v3 b) { v3 r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; return r; }
__device__ __forceinline__ double dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) { v3 r; r.x = a.y * b.z - a.z * b.y; r.y = a.z * b.x - a.x * b.z; r.z = a.x * b.y - a.y * b.x; return r; }
__device__ __forceinline__ double norm(v3 a) { return sqrt((a.x * a.x + a.y * a.y) + a.z * a.z); }
__device__ __forceinline__ v3 axpy(v3 o, v3 d, double t) { v3 r; r.x = o.x + t * d.x; r.y = o.y + t * d.y; r.z = o.z + t * d.z; return r; }

struct vec3_arg { double v[3]; };

// ray_x_lines, intersections.py:6-38
__global__ __launch_bounds__(GB) void k_ray_x_lines(vec3_arg origin, vec3_arg direction, const double* __restrict__ starts,
                                                    const double* __restrict__ ends, int64_t n, double* __restrict__ pts,
                                                    uint8_t* __restrict__ within) {
    const v3 o = ldv(origin.v), d = ldv(direction.v);
    for (int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x; i < n; i += (int64_t)gridDim.x * GB) {
        const v3 s = ldv(starts + 3 * i), e = ldv(ends + 3 * i);
        const v3 ld_ = sub(e, s), rl = sub(s, o);
        const v3 perp = cross(d, ld_), rlxl = cross(rl, ld_);
        const double t = dot(rlxl, perp) / dot(perp, perp);
        const v3 x = axpy(o, d, t);                                  // origin + t*direction (:32)
        pts[3 * i] = x.x; pts[3 * i + 1] = x.y; pts[3 * i + 2] = x.z;
        const double both = norm(sub(x, s)) + norm(sub(x, e));
        const double len = norm(sub(e, s)) + 1e-6;
        within[i] = (both < len) & (t > 0.0);
    }
}

// rays_x_plane, intersections.py:41-63
__global__ __launch_bounds__(GB) void k_rays_x_plane(vec3_arg pp_, vec3_arg pn_, const double* __restrict__ origins,
                                                     const double* __restrict__ dirs, int64_t n, double* __restrict__ pts,
                                                     uint8_t* __restrict__ valid) {
    const v3 pp = ldv(pp_.v), pn = ldv(pn_.v);
    for (int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x; i < n; i += (int64_t)gridDim.x * GB) {
        const v3 o = ldv(origins + 3 * i), d = ldv(dirs + 3 * i);
        const double denom = dot(pn, d);
        const bool ok = denom < -1e-6;
        const double t = ok ? dot(sub(pp, o), pn) / denom : 0.0;
        v3 x; x.x = o.x + d.x * t; x.y = o.y + d.y * t; x.z = o.z + d.z * t;
        pts[3 * i] = x.x; pts[3 * i + 1] = x.y; pts[3 * i + 2] = x.z;
        valid[i] = ok;
    }
}

// lines_x_planes, intersections.py:66-94.  `bmode` reproduces the reference's [N,M,3] - [N,3] broadcast at :89-90:
// 0: N == 1 (line 0), 1: N == M (the segment test uses line m, not line n).
__global__ __launch_bounds__(GB) void k_lines_x_planes(const double* __restrict__ lo, const double* __restrict__ le, int64_t n,
                                                       const double* __restrict__ pps, const double* __restrict__ pns, int m,
                                                       int bmode, double* __restrict__ pts, uint8_t* __restrict__ valid) {
    const int64_t total = n * m;
    for (int64_t k = (int64_t)blockIdx.x * GB + threadIdx.x; k < total; k += (int64_t)gridDim.x * GB) {
        const int64_t i = k / m; const int j = (int)(k - i * m);
        const v3 o = ldv(lo + 3 * i), e = ldv(le + 3 * i);
        v3 d = sub(e, o);
        const double dn = norm(d);
        d.x /= dn; d.y /= dn; d.z /= dn;
        const v3 pp = ldv(pps + 3 * j), pn = ldv(pns + 3 * j);
        const double denom = dot(d, pn);
        const bool ok = (denom < -1e-6) | (denom > 1e-6);
        const double t = ok ? dot(sub(pp, o), pn) / denom : 0.0;
        v3 x; x.x = o.x + d.x * t; x.y = o.y + d.y * t; x.z = o.z + d.z * t;
        pts[3 * k] = x.x; pts[3 * k + 1] = x.y; pts[3 * k + 2] = x.z;
        const int64_t b = bmode ? j : 0;
        const double both = norm(sub(x, ldv(lo + 3 * b))) + norm(sub(x, ldv(le + 3 * b)));
        const double len = dn + 1e-6;
        valid[k] = (both < len) & ok;
    }
}

// point_inside_polygon, intersections.py:97-119: within [m,n], inside [n]
__global__ __launch_bounds__(GB) void k_point_inside_polygon(const double* __restrict__ points, int64_t n,
                                                             const double* __restrict__ verts, int m, uint8_t* __restrict__ inside,
                                                             uint8_t* __restrict__ within) {
    for (int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x; i < n; i += (int64_t)gridDim.x * GB) {
        const v3 p = ldv(points + 3 * i);
        int count = 0;
        for (int j = 0; j < m; ++j) {
            const v3 v = ldv(verts + 3 * j), w = ldv(verts + 3 * ((j + 1 == m) ? 0 : j + 1));
            const v3 e = sub(w, v), d = sub(p, v);
            const double dp = (d.x * e.x + d.z * e.z) + d.y * e.y;
            const bool in = dp >= 0.0;
            within[(size_t)j * n + i] = in;
            count += in;
        }
        inside[i] = (count == 0) | (count == m);
    }
}

// points_plane_projection, intersections.py:167-180 (also the first half of lines_plane_projection :183-204)
__global__ __launch_bounds__(GB) void k_points_plane_projection(const double* __restrict__ points, int64_t n, vec3_arg pp_, vec3_arg nr_,
                                                                double* __restrict__ out) {
    const v3 pp = ldv(pp_.v), nr = ldv(nr_.v);
    const double c = dot(pp, nr);
    for (int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x; i < n; i += (int64_t)gridDim.x * GB) {
        const v3 p = ldv(points + 3 * i);
        const double t = c - dot(nr, p);
        out[3 * i] = p.x + t * nr.x; out[3 * i + 1] = p.y + t * nr.y; out[3 * i + 2] = p.z + t * nr.z;
    }
}

// unit(end_projection - start_projection), intersections.py:200-202
__global__ __launch_bounds__(GB) void k_unit_difference(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                                        double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x; i < n; i += (int64_t)gridDim.x * GB) {
        v3 d = sub(ldv(b + 3 * i), ldv(a + 3 * i));
        const double dn = norm(d);
        out[3 * i] = d.x / dn; out[3 * i + 1] = d.y / dn; out[3 * i + 2] = d.z / dn;
    }
}

inline dim3 grid(int64_t n) { int64_t g = (n + GB - 1) / GB; if (g < 1) g = 1; if (g > 8192) g = 8192; return dim3((unsigned)g); }
inline vec3_arg va(const double* p) { vec3_arg r; r.v[0] = p[0]; r.v[1] = p[1]; r.v[2] = p[2]; return r; }

}  // namespace

hipError_t f3d_launch_ray_x_lines(const double o[3], const double d[3], const double* starts, const double* ends, int64_t n, double* pts,
                                  uint8_t* within, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_ray_x_lines, grid(n), dim3(GB), 0, s, va(o), va(d), starts, ends, n, pts, within);
    return hipGetLastError();
}
hipError_t f3d_launch_rays_x_plane(const double pp[3], const double pn[3], const double* origins, const double* dirs, int64_t n, double* pts,
                                   uint8_t* valid, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_rays_x_plane, grid(n), dim3(GB), 0, s, va(pp), va(pn), origins, dirs, n, pts, valid);
    return hipGetLastError();
}
hipError_t f3d_launch_lines_x_planes(const double* lo, const double* le, int64_t n, const double* pps, const double* pns, int m, int bmode,
                                     double* pts, uint8_t* valid, hipStream_t s) {
    if (n > 0 && m > 0) hipLaunchKernelGGL(k_lines_x_planes, grid(n * m), dim3(GB), 0, s, lo, le, n, pps, pns, m, bmode, pts, valid);
    return hipGetLastError();
}
hipError_t f3d_launch_point_inside_polygon(const double* points, int64_t n, const double* verts, int m, uint8_t* inside, uint8_t* within,
                                           hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_point_inside_polygon, grid(n), dim3(GB), 0, s, points, n, verts, m, inside, within);
    return hipGetLastError();
}
hipError_t f3d_launch_points_plane_projection(const double* points, int64_t n, const double pp[3], const double nr[3], double* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_points_plane_projection, grid(n), dim3(GB), 0, s, points, n, va(pp), va(nr), out);
    return hipGetLastError();
}
hipError_t f3d_launch_unit_difference(const double* a, const double* b, int64_t n, double* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_unit_difference, grid(n), dim3(GB), 0, s, a, b, n, out);
    return hipGetLastError();
}
