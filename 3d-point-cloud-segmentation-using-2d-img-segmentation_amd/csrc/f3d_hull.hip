// Oriented-box fit of many instances on the GPU (gfx950): the recipe behind Open3D's OrientedBoundingBox.create_from_points at the
// reference's call sites (merge_intersecting_bb.py:75,86,126; get3DSeg.py:434-435):
//     convex hull of the instance's points -> mean and covariance of the HULL VERTICES -> symmetric eigen-decomposition, axes by
//     descending eigenvalue, third axis = first x second -> extents of the hull vertices in that frame.
//
// One wavefront (64 lanes) per instance, wave-uniform control flow, lanes over the instance's points:
//   * hull vertex set by gift wrapping.  A hull facet (u, v, w) is known; the facet across its edge u->v is (v, u, p) with p the point
//     that "wins" against every other point q:  S(v, u, p, q) = ((u - v) x (p - v)) . (q - v) > 0  =>  q replaces p.  Around the edge the
//     points span less than a half turn (they all lie behind the known facet), so the comparison is a total order and the
//     tournament runs lane-parallel: a sequential pass over the lane's share, then a six-step butterfly.  The frontier (directed
//     edges whose neighbour facet is still unknown) lives in LDS; a new facet cancels frontier edges it closes and adds the ones it
//     opens; the hull is complete when the frontier is empty.  Cost O(facets x points) -- a few hundred x a few hundred after the
//     inner-hull prefilter (f3d_obb.hip), all instances of a cloud in one launch.
//   * every sign is CERTIFIED: S is evaluated in float64 together with Shewchuk's static bound (7 + 56 eps) eps * permanent; a sign
//     inside the bound (coplanar or duplicate points, a facet with more than three vertices) is never guessed -- the instance is
//     marked F3D_OBB_DEFERRED and the caller fits it on the host (Qhull), exactly the "prove or defer" rule of the fused kernel.
//     Certificates checked at the end: Euler's formula V - E + F = 2 with E = 3F / 2.
//   * mean / covariance over the vertices in ASCENDING candidate order with a fixed lane-strided partial sum + tree (same bits in
//     every run), cyclic Jacobi for the 3 x 3 eigen-problem (orthogonal to rounding, no loss for close eigenvalues unlike the
//     closed-form trigonometric solution), the eigenvector signs normalised (largest component positive; the box as a point set
//     does not depend on them, the corner order does -- LAPACK's / Eigen's signs are not specified either).
// The same wave code builds the tiny hulls of the <= 26 directional extremes per instance (facet equations for the prefilter).
// No MFMA: nothing here is a dense contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "f3d.h"
#include "f3d_kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int HW = 64;                        // one wave per instance
constexpr int FRONT_CAP = 1024;               // frontier edges kept in LDS per instance (a hull of ~150 vertices peaks near 60)
constexpr double O3D_ERR_A = (7.0 + 56.0 * 1.1102230246251565e-16) * 1.1102230246251565e-16;   // Shewchuk, orient3d stage A

struct p3 { double x, y, z; };

__device__ __forceinline__ p3 ldp(const double* __restrict__ pts, int64_t i) { return p3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}; }

// S(a, b, c, d) = ((b - a) x (c - a)) . (d - a): > 0 when d is on the side of plane (a, b, c) its counter-clockwise normal points to.
// Returns +1 / -1 when the sign is certain, 0 when |S| is within the rounding bound (the caller defers the instance).
__device__ __forceinline__ int side(const p3& a, const p3& b, const p3& c, const p3& d) {
    const double adx = a.x - d.x, ady = a.y - d.y, adz = a.z - d.z;
    const double bdx = b.x - d.x, bdy = b.y - d.y, bdz = b.z - d.z;
    const double cdx = c.x - d.x, cdy = c.y - d.y, cdz = c.z - d.z;
    const double bdxcdy = bdx * cdy, cdxbdy = cdx * bdy;
    const double cdxady = cdx * ady, adxcdy = adx * cdy;
    const double adxbdy = adx * bdy, bdxady = bdx * ady;
    const double det = adz * (bdxcdy - cdxbdy) + bdz * (cdxady - adxcdy) + cdz * (adxbdy - bdxady);      // Shewchuk's orient3d = -S
    const double perm = (fabs(bdxcdy) + fabs(cdxbdy)) * fabs(adz) + (fabs(cdxady) + fabs(adxcdy)) * fabs(bdz) + (fabs(adxbdy) + fabs(bdxady)) * fabs(cdz);
    const double bound = O3D_ERR_A * perm;
    if (det > bound) return -1;
    if (-det > bound) return 1;
    return 0;                                                        // (NaN lands here too)
}

// wave-wide: the index (into the instance's points) that wins the tournament around the directed edge v -> u, starting from `start`
// (a point every other point beats: the third vertex of the known facet, or a virtual point given by coordinates).
// `unsure` is raised when a comparison that decided something was not certified.
__device__ __forceinline__ int wrap_edge(const double* __restrict__ pts, int m, const p3& pv, const p3& pu, int iv, int iu, int istart, const p3& pstart,
                                         bool& unsure) {
    const int lane = threadIdx.x & (HW - 1);
    int best = istart;
    p3 pb = pstart;
    for (int j = lane; j < m; j += HW) {
        if (j == iv || j == iu || j == best) continue;
        const p3 q = ldp(pts, j);
        const int s = side(pv, pu, pb, q);
        if (s == 0) unsure = true;
        if (s > 0) { best = j; pb = q; }
    }
#pragma unroll
    for (int off = 1; off < HW; off <<= 1) {
        const int ob = __shfl_xor(best, off, HW);
        p3 q;
        q.x = __shfl_xor(pb.x, off, HW); q.y = __shfl_xor(pb.y, off, HW); q.z = __shfl_xor(pb.z, off, HW);
        if (ob != best && ob != istart) {
            if (best == istart) { best = ob; pb = q; }
            else {
                const int s = side(pv, pu, pb, q);
                if (s == 0) unsure = true;
                if (s > 0) { best = ob; pb = q; }
            }
        }
    }
    unsure = __any(unsure);
    return __builtin_amdgcn_readfirstlane(best);
}

// cyclic Jacobi, symmetric 3 x 3: A = V diag(w) V^T.  a: a00 a01 a02 a11 a12 a22.
__device__ __forceinline__ void jacobi3(double a00, double a01, double a02, double a11, double a12, double a22, double w[3], double V[9]) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 16; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (!(off > 1e-300) || off <= 1e-18 * diag) break;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int p = k == 2 ? 1 : 0, q = k == 0 ? 1 : 2;
            const double apq = A[p][q];
            if (fabs(apq) <= 1e-300) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            const double app = A[p][p], aqq = A[q][q];
            A[p][p] = app - t * apq; A[q][q] = aqq + t * apq; A[p][q] = A[q][p] = 0.0;
            const int r = 3 - p - q;
            const double arp = A[r][p], arq = A[r][q];
            A[r][p] = A[p][r] = c * arp - s * arq;
            A[r][q] = A[q][r] = s * arp + c * arq;
#pragma unroll
            for (int i = 0; i < 3; ++i) { const double vip = Q[i][p], viq = Q[i][q]; Q[i][p] = c * vip - s * viq; Q[i][q] = s * vip + c * viq; }
        }
    }
    for (int i = 0; i < 3; ++i) { w[i] = A[i][i]; for (int j = 0; j < 3; ++j) V[3 * i + j] = Q[i][j]; }
}

// MODE_FIT: box of every instance.  pts = the instances' candidate points, compact: instance k owns pts[start[k] .. start[k + 1]).
// MODE_FACETS: hull of <= 26 gathered points per instance; writes outward facet equations (unit normal, offset) and the margin.
enum { MODE_FIT = 0, MODE_FACETS = 1 };

struct hull_lds {
    int fu[FRONT_CAP], fv[FRONT_CAP], fw[FRONT_CAP];                  // frontier: directed edge u -> v of a known facet whose third vertex is w
};

// the hull of points pts[0 .. m) by one wave; marks vertices in isvert[0 .. m) (global, zeroed here); for MODE_FACETS also writes the
// facets.  Returns F3D_OBB_OK / F3D_OBB_FEW / F3D_OBB_DEFERRED.  nvert / nfacet by reference.
template <int MODE>
__device__ int wave_hull(const double* __restrict__ pts, int m, uint8_t* __restrict__ isvert, hull_lds& L, int& nvert, int& nfacet,
                         double* __restrict__ facets, int facet_cap) {
    const int lane = threadIdx.x & (HW - 1);
    nvert = 0; nfacet = 0;
    for (int j = lane; j < m; j += HW) isvert[j] = 0;
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    if (m < 4) return F3D_OBB_FEW;
    bool unsure = false;
    // ---- first vertex: the lexicographic minimum (x, then y, then z); a duplicate of it is found by the predicates later
    int ia = 0;
    {
        double bx = INFINITY, by = INFINITY, bz = INFINITY; int bi = 0x7fffffff;
        for (int j = lane; j < m; j += HW) {
            const p3 q = ldp(pts, j);
            if (!(fabs(q.x) < 1e300 && fabs(q.y) < 1e300 && fabs(q.z) < 1e300)) unsure = true;      // non-finite coordinates: the host decides (Qhull raises)
            const bool less = q.x < bx || (q.x == bx && (q.y < by || (q.y == by && (q.z < bz || (q.z == bz && j < bi)))));
            if (less) { bx = q.x; by = q.y; bz = q.z; bi = j; }
        }
#pragma unroll
        for (int off = 1; off < HW; off <<= 1) {
            const double ox = __shfl_xor(bx, off, HW), oy = __shfl_xor(by, off, HW), oz = __shfl_xor(bz, off, HW);
            const int oi = __shfl_xor(bi, off, HW);
            const bool less = ox < bx || (ox == bx && (oy < by || (oy == by && (oz < bz || (oz == bz && oi < bi)))));
            if (less) { bx = ox; by = oy; bz = oz; bi = oi; }
        }
        ia = __builtin_amdgcn_readfirstlane(bi);
        if (__any(unsure) || ia == 0x7fffffff) return F3D_OBB_DEFERRED;
    }
    const p3 pa = ldp(pts, ia);
    // ---- first edge a - b: the supporting plane x = a.x contains the virtual points a' = a + (0, L, 0) and a'' = a + (0, 0, -L);
    // (a, a', a'') is counter-clockwise seen from -x, i.e. every point is behind it, so a'' is a valid start around the axis a -> a'
    double ext = 1.0;
    {
        double mx = 0.0;
        for (int j = lane; j < m; j += HW) { const p3 q = ldp(pts, j); mx = fmax(mx, fmax(fabs(q.x - pa.x), fmax(fabs(q.y - pa.y), fabs(q.z - pa.z)))); }
#pragma unroll
        for (int off = 1; off < HW; off <<= 1) mx = fmax(mx, __shfl_xor(mx, off, HW));
        ext = mx > 0.0 ? mx : 1.0;
    }
    const p3 pa1 = p3{pa.x, pa.y + ext, pa.z}, pa2 = p3{pa.x, pa.y, pa.z - ext};
    // around the directed axis a1 -> a (so that the known "facet" is (a, a1, a2) = u -> v with u = a, v = a1, w = a2)
    const int ib = wrap_edge(pts, m, pa1, pa, -1, ia, -2, pa2, unsure);
    if (unsure || ib < 0) return F3D_OBB_DEFERRED;
    const p3 pb = ldp(pts, ib);
    // the plane (a1, a, b) supports the hull and contains a, b: facet "(a1, a, b)" plays the known facet of edge a -> b (u = a, v = b?)
    // Known facet orientation: (v', u', p) = (a1, a, b) is outward, its directed edges are a1 -> a, a -> b, b -> a1.  The hull facet
    // across the directed edge a -> b contains b -> a: wrap around v = b, u = a starting from w = a1.
    const int ic = wrap_edge(pts, m, pb, pa, ib, ia, -2, pa1, unsure);
    if (unsure || ic < 0) return F3D_OBB_DEFERRED;
    // first hull facet: (b, a, c), outward by construction
    int nf = 0, nfront = 0, nv = 0, nedges2 = 0;
    auto mark = [&](int i) { if (lane == 0) isvert[i] = 1; };
    auto emit_facet = [&](int i0, int i1, int i2) {                   // outward facet (i0, i1, i2)
        if (MODE == MODE_FACETS) {
            if (nf < facet_cap && lane == 0) {
                const p3 A = ldp(pts, i0), B = ldp(pts, i1), C = ldp(pts, i2);
                const double ux = B.x - A.x, uy = B.y - A.y, uz = B.z - A.z, vx = C.x - A.x, vy = C.y - A.y, vz = C.z - A.z;
                double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
                const double nn = sqrt((nx * nx + ny * ny) + nz * nz);
                nx /= nn; ny /= nn; nz /= nn;
                double* e = facets + 4 * (size_t)nf;
                e[0] = nx; e[1] = ny; e[2] = nz; e[3] = -((nx * A.x + ny * A.y) + nz * A.z);
            }
        }
        ++nf;
    };
    // frontier operations (wave-uniform state in LDS, lane-parallel search)
    auto close_or_open = [&](int x, int y, int third) -> bool {       // the new facet has the directed edge x -> y
        int found = -1;
        for (int k = lane; k < nfront; k += HW) if (L.fu[k] == y && L.fv[k] == x) found = k;
        const unsigned long long hit = __ballot(found >= 0);
        if (hit) {                                                   // its reverse was waiting: both are matched now
            const int src = __builtin_ctzll(hit);
            const int k = __shfl(found, src, HW);
            --nfront;
            if (lane == 0 && k != nfront) { L.fu[k] = L.fu[nfront]; L.fv[k] = L.fv[nfront]; L.fw[k] = L.fw[nfront]; }
        } else {
            if (nfront >= FRONT_CAP) return false;
            if (lane == 0) { L.fu[nfront] = x; L.fv[nfront] = y; L.fw[nfront] = third; }
            ++nfront;
        }
        __builtin_amdgcn_wave_barrier();
        return true;
    };
    emit_facet(ib, ia, ic);
    mark(ia); mark(ib); mark(ic);
    if (lane == 0) { L.fu[0] = ib; L.fv[0] = ia; L.fw[0] = ic; L.fu[1] = ia; L.fv[1] = ic; L.fw[1] = ib; L.fu[2] = ic; L.fv[2] = ib; L.fw[2] = ia; }
    nfront = 3;
    __builtin_amdgcn_wave_barrier();
    const int max_facets = 2 * m;                                    // a triangulated hull of h <= m vertices has 2h - 4 facets
    while (nfront > 0) {
        if (nf > max_facets) return F3D_OBB_DEFERRED;                 // cannot happen with consistent signs
        --nfront;
        const int u = L.fu[nfront], v = L.fv[nfront], w = L.fw[nfront];       // known facet (u, v, w); wanted: the facet with edge v -> u
        __builtin_amdgcn_wave_barrier();
        const p3 pu = ldp(pts, u), pv = ldp(pts, v), pw = ldp(pts, w);
        const int p = wrap_edge(pts, m, pv, pu, v, u, w, pw, unsure);
        if (unsure || p == w) return F3D_OBB_DEFERRED;                // p == w: every other point is coplanar with the known facet (flat input)
        emit_facet(v, u, p);
        mark(p);
        // edges of (v, u, p): v -> u closes the popped one; u -> p and p -> v close or open
        if (!close_or_open(u, p, v) || !close_or_open(p, v, u)) return F3D_OBB_DEFERRED;
    }
    __threadfence_block();                                            // lane 0's marks, read by every lane below (and by the caller)
    __builtin_amdgcn_wave_barrier();
    // certificates: count the vertices, Euler V - E + F = 2 with 2E = 3F
    int cnt = 0;
    for (int j = lane; j < m; j += HW) cnt += isvert[j] ? 1 : 0;
#pragma unroll
    for (int off = 1; off < HW; off <<= 1) cnt += __shfl_xor(cnt, off, HW);
    nv = cnt; nedges2 = 3 * nf;
    if ((nedges2 & 1) || nv - nedges2 / 2 + nf != 2) return F3D_OBB_DEFERRED;
    if (MODE == MODE_FACETS && nf > facet_cap) return F3D_OBB_DEFERRED;
    nvert = nv; nfacet = nf;
    return F3D_OBB_OK;
}

__global__ __launch_bounds__(HW) void k_obb_fit(const double* __restrict__ pts_all, const int64_t* __restrict__ start, int nfit,
                                                 double* __restrict__ boxes, int32_t* __restrict__ status, uint8_t* __restrict__ isvert_all,
                                                 int32_t* __restrict__ vlist_all, int32_t* __restrict__ nvert_out) {
    __shared__ hull_lds L;
    const int lane = threadIdx.x;
    for (int k = blockIdx.x; k < nfit; k += gridDim.x) {
        const int64_t s0 = start[k], s1 = start[k + 1];
        const double* pts = pts_all + 3 * s0;
        uint8_t* isvert = isvert_all + s0;
        double* box = boxes + 15 * (size_t)k;
        int nv = 0, nf = 0, st;
        if (s1 - s0 > 0x7fffffff) st = F3D_OBB_DEFERRED;
        else st = wave_hull<MODE_FIT>(pts, (int)(s1 - s0), isvert, L, nv, nf, nullptr, 0);
        const int m = (int)(s1 - s0);
        if (st == F3D_OBB_OK) {
            // the vertices in ascending order as a list: the sums below then depend on the vertex set alone, not on which other
            // candidates came along (a box fitted on all members and one fitted on the prefilter's survivors are the same bits)
            int32_t* vlist = vlist_all + s0;
            int run = 0;
            for (int base = 0; base < m; base += HW) {
                const bool v = base + lane < m && isvert[base + lane];
                const unsigned long long mk = __ballot(v);
                if (v) vlist[run + __popcll(mk & ((1ull << lane) - 1ull))] = base + lane;
                run += __popcll(mk);
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            // mean and covariance of the vertices: lane-strided partial sums over the list, fixed tree
            double sx = 0, sy = 0, sz = 0;
            for (int r = lane; r < nv; r += HW) { const p3 q = ldp(pts, vlist[r]); sx += q.x; sy += q.y; sz += q.z; }
#pragma unroll
            for (int off = 1; off < HW; off <<= 1) { sx += __shfl_xor(sx, off, HW); sy += __shfl_xor(sy, off, HW); sz += __shfl_xor(sz, off, HW); }
            const double mx = sx / nv, my = sy / nv, mz = sz / nv;
            double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
            for (int r = lane; r < nv; r += HW) {
                const p3 q = ldp(pts, vlist[r]);
                const double dx = q.x - mx, dy = q.y - my, dz = q.z - mz;
                c00 += dx * dx; c01 += dx * dy; c02 += dx * dz; c11 += dy * dy; c12 += dy * dz; c22 += dz * dz;
            }
#pragma unroll
            for (int off = 1; off < HW; off <<= 1) {
                c00 += __shfl_xor(c00, off, HW); c01 += __shfl_xor(c01, off, HW); c02 += __shfl_xor(c02, off, HW);
                c11 += __shfl_xor(c11, off, HW); c12 += __shfl_xor(c12, off, HW); c22 += __shfl_xor(c22, off, HW);
            }
            double w[3], V[9];
            jacobi3(c00 / nv, c01 / nv, c02 / nv, c11 / nv, c12 / nv, c22 / nv, w, V);
            // axes by descending eigenvalue (ties: the later column first, like argsort(-w, stable) on ascending LAPACK values would not
            // distinguish either); signs: largest component positive; third axis = first x second
            int o0 = 0, o1 = 1, o2 = 2;
            if (w[o1] > w[o0]) { const int t = o0; o0 = o1; o1 = t; }
            if (w[o2] > w[o0]) { const int t = o0; o0 = o2; o2 = t; }
            if (w[o2] > w[o1]) { const int t = o1; o1 = o2; o2 = t; }
            double R[9];                                              // row-major, columns = axes
            const int ord[2] = {o0, o1};
            for (int a = 0; a < 2; ++a) {
                double x = V[ord[a]], y = V[3 + ord[a]], z = V[6 + ord[a]];
                const double ax = fabs(x), ay = fabs(y), az = fabs(z);
                const double big = (ax >= ay && ax >= az) ? x : (ay >= az ? y : z);
                if (big < 0.0) { x = -x; y = -y; z = -z; }
                R[a] = x; R[3 + a] = y; R[6 + a] = z;
            }
            R[2] = R[3] * R[7] - R[6] * R[4];
            R[5] = R[6] * R[1] - R[0] * R[7];
            R[8] = R[0] * R[4] - R[3] * R[1];
            // extents of the vertices in the frame
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int r = lane; r < nv; r += HW) {
                const p3 q = ldp(pts, vlist[r]);
                const double dx = q.x - mx, dy = q.y - my, dz = q.z - mz;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double t = (dx * R[a] + dy * R[3 + a]) + dz * R[6 + a];
                    lo[a] = fmin(lo[a], t); hi[a] = fmax(hi[a], t);
                }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int off = 1; off < HW; off <<= 1) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], off, HW)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], off, HW)); }
            if (lane == 0) {
                const double h0 = (lo[0] + hi[0]) / 2, h1 = (lo[1] + hi[1]) / 2, h2 = (lo[2] + hi[2]) / 2;
                box[0] = mx + ((R[0] * h0 + R[1] * h1) + R[2] * h2);
                box[1] = my + ((R[3] * h0 + R[4] * h1) + R[5] * h2);
                box[2] = mz + ((R[6] * h0 + R[7] * h1) + R[8] * h2);
                for (int a = 0; a < 9; ++a) box[3 + a] = R[a];
                for (int a = 0; a < 3; ++a) box[12 + a] = hi[a] - lo[a];
            }
        } else if (lane < 15) box[lane] = 0.0;
        if (lane == 0) { status[k] = st; if (nvert_out) nvert_out[k] = nv; }
        __builtin_amdgcn_wave_barrier();
    }
}

// hulls of the directional extremes: instance k's <= 26 extreme members (caller-order indices, -1 = none; duplicates allowed) are
// de-duplicated, gathered into LDS-free scratch (global `gathered` [nids][26][3]) and wrapped; facets [nids][F3D_OBB_SMALL_FACETS][4],
// nfacets[k] = 0 when the instance keeps all of its members (fewer than min_members members, fewer than 4 distinct extremes, or a
// hull that could not be certified -- flat, duplicate coordinates ...).  margin[k] = 1e-9 (max |coordinate| + 1).
template <typename T>
__global__ __launch_bounds__(HW) void k_obb_small_hulls(const T* __restrict__ xyz, const int32_t* __restrict__ extremes, const int64_t* __restrict__ starts,
                                                         int64_t nids, int min_members, double* __restrict__ gathered, uint8_t* __restrict__ isvert_all,
                                                         double* __restrict__ facets, int32_t* __restrict__ nfacets, double* __restrict__ margin) {
    __shared__ hull_lds L;
    __shared__ int uniq[F3D_OBB_NDIR];
    const int lane = threadIdx.x;
    for (int64_t k = blockIdx.x; k < nids; k += gridDim.x) {
        double* pts = gathered + (size_t)k * F3D_OBB_NDIR * 3;
        int m = 0;
        if (starts[k + 1] - starts[k] >= min_members) {
            // distinct indices in ascending order (the order only has to be the same in every run)
            const int mine = lane < F3D_OBB_NDIR ? extremes[k * F3D_OBB_NDIR + lane] : -1;
            int rank = 0; bool dup = mine < 0;
            for (int j = 0; j < F3D_OBB_NDIR; ++j) {
                const int o = __shfl(mine, j, HW);
                if (o >= 0 && o < mine) ++rank;
                if (o == mine && j < lane) dup = true;
            }
            // rank counts smaller entries with multiplicity; compact through LDS instead
            const unsigned long long keep = __ballot(!dup);
            m = __popcll(keep);
            if (!dup) uniq[__popcll(keep & ((1ull << lane) - 1ull))] = mine;
            __builtin_amdgcn_wave_barrier();
            (void)rank;
            if (lane < m) {
                const int64_t i = uniq[lane];
                pts[3 * lane] = (double)xyz[3 * i]; pts[3 * lane + 1] = (double)xyz[3 * i + 1]; pts[3 * lane + 2] = (double)xyz[3 * i + 2];
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
        }
        int nv = 0, nf = 0;
        int st = F3D_OBB_FEW;
        if (m >= 4) st = wave_hull<MODE_FACETS>(pts, m, isvert_all + (size_t)k * F3D_OBB_NDIR, L, nv, nf, facets + (size_t)k * F3D_OBB_SMALL_FACETS * 4, F3D_OBB_SMALL_FACETS);
        double amax = 0.0;
        if (lane < m) amax = fmax(fabs(pts[3 * lane]), fmax(fabs(pts[3 * lane + 1]), fabs(pts[3 * lane + 2])));
#pragma unroll
        for (int off = 1; off < HW; off <<= 1) amax = fmax(amax, __shfl_xor(amax, off, HW));
        if (lane == 0) { nfacets[k] = st == F3D_OBB_OK ? nf : 0; margin[k] = 1e-9 * (amax + 1.0); }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

hipError_t f3d_launch_obb_fit(const double* pts, const int64_t* start, int nfit, double* boxes, int32_t* status, uint8_t* isvert, int32_t* vlist,
                              int32_t* nvert, hipStream_t s) {
    if (nfit <= 0) return hipSuccess;
    const int grid = nfit < 65536 ? nfit : 65536;
    hipLaunchKernelGGL(k_obb_fit, dim3(grid), dim3(HW), 0, s, pts, start, nfit, boxes, status, isvert, vlist, nvert);
    return hipGetLastError();
}

hipError_t f3d_launch_obb_small_hulls(const void* xyz, int dtype, const int32_t* extremes, const int64_t* starts, int64_t nids, int min_members,
                                      double* gathered, uint8_t* isvert, double* facets, int32_t* nfacets, double* margin, hipStream_t s) {
    if (nids <= 0) return hipSuccess;
    const int grid = (int)(nids < 65536 ? nids : 65536);
    if (dtype == F3D_F64) hipLaunchKernelGGL(k_obb_small_hulls<double>, dim3(grid), dim3(HW), 0, s, (const double*)xyz, extremes, starts, nids, min_members, gathered, isvert, facets, nfacets, margin);
    else hipLaunchKernelGGL(k_obb_small_hulls<float>, dim3(grid), dim3(HW), 0, s, (const float*)xyz, extremes, starts, nids, min_members, gathered, isvert, facets, nfacets, margin);
    return hipGetLastError();
}
