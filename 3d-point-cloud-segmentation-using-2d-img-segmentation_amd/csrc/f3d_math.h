// Canonical fp64 arithmetic of the projection path, shared by host (view building) and device
// (kernels) code of libf3d_hip.so.  Compiled with -ffp-contract=off: every product and sum is
// rounded on its own, in exactly the order written here, which is the order the oracle
// (oracle/np_ref.py, oracle/f3d_oracle.c) fixes.  Explicit f3d_fma() is used only where a
// result is NOT part of the reference's arithmetic (the conservative pre-cull).
#pragma once
#include <math.h>
#include <stdint.h>
#include "f3d.h"

#pragma clang fp contract(off)

#if defined(__HIPCC__)
#define F3D_HD __host__ __device__ __forceinline__
#else
#define F3D_HD inline
#endif

struct f3d_p3 { double x, y, z; };

// (a0*b0 + a1*b1) + a2*b2  -- the left-to-right order of the oracle
F3D_HD double f3d_dot3(double a0, double a1, double a2, double b0, double b1, double b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

// SpatQuadranion.rotate, RTAB_utils/spatQuad.py:16-27:  q p conj(q), q = (rq, v) un-normalised
F3D_HD f3d_p3 f3d_rotate(const double q[4], f3d_p3 p) {
    const double rq = q[0], v0 = q[1], v1 = q[2], v2 = q[3];
    const double m0 = -v0, m1 = -v1, m2 = -v2;                       // vq_           (:18)
    const double rqp = -f3d_dot3(p.x, p.y, p.z, v0, v1, v2);         // -dot(p, vq)   (:22)
    const double c0 = v1 * p.z - v2 * p.y;                           // cross(vq, p)  (:23)
    const double c1 = v2 * p.x - v0 * p.z;
    const double c2 = v0 * p.y - v1 * p.x;
    const double a0 = rq * p.x + c0, a1 = rq * p.y + c1, a2 = rq * p.z + c2;    // vqp
    const double d0 = a1 * m2 - a2 * m1;                             // cross(vqp, vq_) (:27)
    const double d1 = a2 * m0 - a0 * m2;
    const double d2 = a0 * m1 - a1 * m0;
    f3d_p3 o;
    o.x = (rqp * m0 + rq * a0) + d0;
    o.y = (rqp * m1 + rq * a1) + d1;
    o.z = (rqp * m2 + rq * a2) + d2;
    return o;
}

// float stage of points2pixel (camera_utils.py:21-24): homogeneous pixel (h0, h1, h2)
F3D_HD f3d_p3 f3d_project_h(const double K[9], const double qinv[4], const double t[3], f3d_p3 p) {
    f3d_p3 d;
    d.x = p.x - t[0]; d.y = p.y - t[1]; d.z = p.z - t[2];            // :21
    const f3d_p3 c = f3d_rotate(qinv, d);                            // :22
    f3d_p3 h;
    h.x = (K[0] * c.x + K[1] * c.y) + K[2] * c.z;                    // :23
    h.y = (K[3] * c.x + K[4] * c.y) + K[5] * c.z;
    h.z = (K[6] * c.x + K[7] * c.y) + K[8] * c.z;
    return h;
}

// floor(x).astype(int32) with the x86-64 C-cast result for NaN / out-of-range (INT32_MIN)
F3D_HD int32_t f3d_floor_to_i32(double x) {
    const double f = floor(x);
    return (f >= -2147483648.0 && f <= 2147483647.0) ? (int32_t)f : INT32_MIN;
}

// one plane of point_inside_polyhedra (intersections.py:157-160).  The three products are
// summed as (d0*n0 + d2*n2) + d1*n1: the order np.einsum('nmc,mc->mn') shows in the build
// container, pinned by the near-plane golden vectors.
F3D_HD double f3d_plane_dp(const double pp[3], const double n[3], f3d_p3 p) {
    const double d0 = p.x - pp[0], d1 = p.y - pp[1], d2 = p.z - pp[2];
    return (d0 * n[0] + d2 * n[2]) + d1 * n[1];
}

F3D_HD bool f3d_inside_view(const f3d_view& v, f3d_p3 p) {
    bool in = true;
#pragma unroll
    for (int m = 0; m < F3D_NPLANES; ++m) in = in & (f3d_plane_dp(v.plane_pt[m], v.plane_n[m], p) >= 0.0);
    return in;
}
