/*
 * f3d_oracle.c -- plain-C restatement of the Fusion3DSeg hot path.  TEST INFRASTRUCTURE, NOT PRODUCT:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (libf3d_hip.so and the Python package) never links or calls anything in oracle/.
 *
 * Same canonical arithmetic as oracle/np_ref.py (see its header): every product and sum rounded on
 * its own (compile with -ffp-contract=off), 3-term dots left to right, the plane test as
 * (d0*n0 + d2*n2) + d1*n1, IEEE division, floor, int32 cast of NaN/out-of-range = INT32_MIN.
 * Pinned by tests/test_oracle_golden.py against the tests/golden npz files (vectors produced by running the
 * reference) both directly and through bit-for-bit agreement with np_ref.
 *
 * Build: make -C oracle   ->  oracle/_build/libf3d_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_INDEX (-3)
#define ORC_ERR_ZERO_QUAT (-4)

static double dot3(double a0, double a1, double a2, double b0, double b1, double b2) { return (a0 * b0 + a1 * b1) + a2 * b2; }

/* pyquaternion Quaternion(q).inverse.elements (camera_utils.py:22): conj / sum of squares, not normalised */
int orc_quat_inverse(const double q[4], double o[4]) {
    const double ss = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
    if (ss == 0.0) return ORC_ERR_ZERO_QUAT;
    o[0] = q[0] / ss; o[1] = -q[1] / ss; o[2] = -q[2] / ss; o[3] = -q[3] / ss;
    return ORC_OK;
}

/* SpatQuadranion.rotate, RTAB_utils/spatQuad.py:7-28 */
static void rotate1(const double q[4], const double p[3], double o[3]) {
    const double rq = q[0], v0 = q[1], v1 = q[2], v2 = q[3];
    const double m0 = -v0, m1 = -v1, m2 = -v2;
    const double rqp = -dot3(p[0], p[1], p[2], v0, v1, v2);                      /* :22 */
    const double c0 = v1 * p[2] - v2 * p[1], c1 = v2 * p[0] - v0 * p[2], c2 = v0 * p[1] - v1 * p[0];
    const double a0 = rq * p[0] + c0, a1 = rq * p[1] + c1, a2 = rq * p[2] + c2;  /* :23 */
    const double d0 = a1 * m2 - a2 * m1, d1 = a2 * m0 - a0 * m2, d2 = a0 * m1 - a1 * m0;
    o[0] = (rqp * m0 + rq * a0) + d0;                                            /* :27 */
    o[1] = (rqp * m1 + rq * a1) + d1;
    o[2] = (rqp * m2 + rq * a2) + d2;
}

void orc_rotate(const double q[4], const double* xyz, int64_t n, double* out) {
    for (int64_t i = 0; i < n; ++i) rotate1(q, xyz + 3 * i, out + 3 * i);
}

static int32_t floor_i32(double x) {
    const double f = floor(x);
    return (f >= -2147483648.0 && f <= 2147483647.0) ? (int32_t)f : INT32_MIN;
}

/* camera_utils.py:21-24: homogeneous pixel of one point */
static void project_h(const double K[9], const double qinv[4], const double t[3], const double p[3], double h[3]) {
    const double d[3] = {p[0] - t[0], p[1] - t[1], p[2] - t[2]};
    double c[3];
    rotate1(qinv, d, c);
    h[0] = (K[0] * c[0] + K[1] * c[1]) + K[2] * c[2];
    h[1] = (K[3] * c[0] + K[4] * c[1]) + K[5] * c[2];
    h[2] = (K[6] * c[0] + K[7] * c[1]) + K[8] * c[2];
}

/* points2pixel, camera_utils.py:9-26 -> uv[2*n], row 0 = u */
int orc_points2pixel(const double* xyz, int64_t n, const double K[9], const double q[4], const double t[3], int32_t* uv) {
    double qi[4];
    if (orc_quat_inverse(q, qi)) return ORC_ERR_ZERO_QUAT;
    for (int64_t i = 0; i < n; ++i) {
        double h[3];
        project_h(K, qi, t, xyz + 3 * i, h);
        uv[i] = floor_i32(h[0] / h[2]);
        uv[n + i] = floor_i32(h[1] / h[2]);
    }
    return ORC_OK;
}

static void inv3(const double K[9], double o[9]) {
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const double A = e * i - f * h, B = c * h - b * i, C = b * f - c * e;
    const double D = f * g - d * i, E = a * i - c * g, F = c * d - a * f;
    const double G = d * h - e * g, H = b * g - a * h, I = a * e - b * d;
    const double det = (a * A + b * D) + c * G;
    o[0] = A / det; o[1] = B / det; o[2] = C / det; o[3] = D / det; o[4] = E / det; o[5] = F / det;
    o[6] = G / det; o[7] = H / det; o[8] = I / det;
}

/* Fusion._get_frustum_data (fusion.py:119-132) + the 5 planes of Fusion.fuse (fusion.py:254-258).
 * plane_pts / plane_nrm: [V,5,3]; eyes / lookats: [V,3] (any may be NULL). */
void orc_frustum_planes(const double K[9], double w, double h, const double* q, const double* t, int nviews, double max_depth,
                        double* plane_pts, double* plane_nrm, double* eyes, double* lookats) {
    double Ki[9];
    inv3(K, Ki);
    const double pix[6][3] = {{0, 0, 0}, {0, 0, 1}, {w, 0, 1}, {w, h, 1}, {0, h, 1}, {w / 2, h / 2, 1}};
    for (int v = 0; v < nviews; ++v) {
        double world[6][3];
        for (int k = 0; k < 6; ++k) {
            double c[3], r[3];
            for (int a = 0; a < 3; ++a) c[a] = ((Ki[3 * a] * pix[k][0] + Ki[3 * a + 1] * pix[k][1]) + Ki[3 * a + 2] * pix[k][2]) / 1;
            rotate1(q + 4 * v, c, r);
            for (int a = 0; a < 3; ++a) world[k][a] = r[a] + t[3 * v + a];
        }
        double look[3];
        {
            const double v0 = world[5][0] - world[0][0], v1 = world[5][1] - world[0][1], v2 = world[5][2] - world[0][2];
            const double nn = sqrt((v0 * v0 + v1 * v1) + v2 * v2);
            look[0] = v0 / nn; look[1] = v1 / nn; look[2] = v2 / nn;
        }
        if (eyes) memcpy(eyes + 3 * v, world[0], 24);
        if (lookats) memcpy(lookats + 3 * v, look, 24);
        for (int k = 0; k < 4; ++k) {
            const double* ca = world[1 + k];
            const double* cb = world[1 + (k + 1) % 4];
            const double a0 = ca[0] - world[0][0], a1 = ca[1] - world[0][1], a2 = ca[2] - world[0][2];
            const double b0 = cb[0] - world[0][0], b1 = cb[1] - world[0][1], b2 = cb[2] - world[0][2];
            const double n0 = a1 * b2 - a2 * b1, n1 = a2 * b0 - a0 * b2, n2 = a0 * b1 - a1 * b0;
            const double nn = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
            if (plane_pts) memcpy(plane_pts + 15 * v + 3 * k, world[0], 24);
            if (plane_nrm) { plane_nrm[15 * v + 3 * k] = n0 / nn; plane_nrm[15 * v + 3 * k + 1] = n1 / nn; plane_nrm[15 * v + 3 * k + 2] = n2 / nn; }
        }
        for (int a = 0; a < 3; ++a) {
            if (plane_pts) plane_pts[15 * v + 12 + a] = world[0][a] + max_depth * look[a];
            if (plane_nrm) plane_nrm[15 * v + 12 + a] = -look[a];
        }
    }
}

/* point_inside_polyhedra, intersections.py:146-164 */
void orc_inside_polyhedra(const double* xyz, int64_t n, const double* pp, const double* nr, int m, uint8_t* inside) {
    for (int64_t i = 0; i < n; ++i) {
        int in = 1;
        for (int k = 0; k < m; ++k) {
            const double d0 = xyz[3 * i] - pp[3 * k], d1 = xyz[3 * i + 1] - pp[3 * k + 1], d2 = xyz[3 * i + 2] - pp[3 * k + 2];
            const double dp = (d0 * nr[3 * k] + d2 * nr[3 * k + 2]) + d1 * nr[3 * k + 1];
            in &= (dp >= 0.0);
        }
        inside[i] = (uint8_t)in;
    }
}

/* one frame of VotingSegmentation.vote (voting.py:94-98) with quirk Q1: each distinct (point,label) pair adds 1 */
int orc_vote_frame(double* votes, int64_t npts, int ncols, const int32_t* uv2pt, const uint8_t* mask, int64_t hw) {
    for (int64_t i = 0; i < hw; ++i) {                       /* NumPy validates every index before writing */
        const int64_t p = uv2pt[i];
        if (p == -1) continue;
        if (p >= npts || p < -npts || mask[i] >= ncols) return ORC_ERR_INDEX;
    }
    uint8_t* seen = (uint8_t*)calloc((size_t)npts * ncols, 1);
    if (!seen) return -5;
    for (int64_t i = 0; i < hw; ++i) {
        int64_t p = uv2pt[i];
        if (p == -1) continue;
        if (p < 0) p += npts;
        const size_t k = (size_t)p * ncols + mask[i];
        if (!seen[k]) { seen[k] = 1; votes[k] += 1.0; }
    }
    free(seen);
    return ORC_OK;
}

/* VotingSegmentation.segment, voting.py:106-137.  filter == NULL / nfilter == 0: no filter. */
void orc_segment(const double* votes, int64_t npts, int ncols, int nclasses, double threshold, const int32_t* filter, int nfilter,
                 int64_t* classes) {
    for (int64_t i = 0; i < npts; ++i) {
        const double* r = votes + (size_t)i * ncols;
        double total = 0.0;
        for (int c = 0; c < ncols; ++c) total += r[c];               /* :120 (all columns) */
        double best;
        int64_t cls = 0;
        if (nfilter > 0) {
            best = r[filter[0] < 0 ? filter[0] + ncols : filter[0]];
            for (int k = 1; k < nfilter; ++k) {
                const double x = r[filter[k] < 0 ? filter[k] + ncols : filter[k]];
                if (x > best) { best = x; cls = k; }                   /* first maximum wins (:124) */
            }
        } else {
            best = r[0];
            for (int c = 1; c < ncols; ++c) if (r[c] > best) { best = r[c]; cls = c; }
        }
        if (!(total > 0.0)) cls = nclasses;                           /* :126 */
        else if (best / total < threshold) cls = nclasses;            /* :128-130 */
        if (best == 0.0) cls = nclasses;                              /* :131 */
        for (int k = 0; k < nfilter; ++k) if (cls == k) cls = filter[k];   /* sequential, aliasing (Q3) :133-135 */
        classes[i] = cls;
    }
}

/* composed forward path: per view inside -> uv -> bounds -> label -> vote; then segment.
 * votes (optional): double [n, nclasses+1], zeroed by the caller.  Returns ORC_ERR_INDEX for a label > nclasses. */
int orc_project_vote_argmax(const double* xyz, int64_t n, const double K[9], double w, double h, const double* q, const double* t,
                            int nviews, double max_depth, const uint8_t* masks, int H, int W, int nclasses, double threshold,
                            const int32_t* filter, int nfilter, int64_t* classes, double* votes_out) {
    const int ncols = nclasses + 1;
    double* pp = (double*)malloc(sizeof(double) * 15 * (size_t)(nviews > 0 ? nviews : 1));
    double* pn = (double*)malloc(sizeof(double) * 15 * (size_t)(nviews > 0 ? nviews : 1));
    double* qi = (double*)malloc(sizeof(double) * 4 * (size_t)(nviews > 0 ? nviews : 1));
    double* row = (double*)malloc(sizeof(double) * (size_t)ncols);
    int rc = ORC_OK;
    orc_frustum_planes(K, w, h, q, t, nviews, max_depth, pp, pn, NULL, NULL);
    for (int v = 0; v < nviews && !rc; ++v) rc = orc_quat_inverse(q + 4 * v, qi + 4 * v);
    for (int64_t i = 0; i < n && !rc; ++i) {
        memset(row, 0, sizeof(double) * (size_t)ncols);
        for (int v = 0; v < nviews; ++v) {
            uint8_t in;
            orc_inside_polyhedra(xyz + 3 * i, 1, pp + 15 * v, pn + 15 * v, 5, &in);
            if (!in) continue;
            double hh[3];
            project_h(K, qi + 4 * v, t + 3 * v, xyz + 3 * i, hh);
            const int32_t u = floor_i32(hh[0] / hh[2]), vv = floor_i32(hh[1] / hh[2]);
            if (u < 0 || u >= W || vv < 0 || vv >= H) continue;
            const int label = masks[(size_t)v * H * W + (size_t)vv * W + u];
            if (label > nclasses) { rc = ORC_ERR_INDEX; break; }
            row[label] += 1.0;
        }
        if (rc) break;
        orc_segment(row, 1, ncols, nclasses, threshold, filter, nfilter, classes + i);
        if (votes_out) memcpy(votes_out + (size_t)i * ncols, row, sizeof(double) * (size_t)ncols);
    }
    free(pp); free(pn); free(qi); free(row);
    return rc;
}

/* open3d get_point_indices_within_bounding_box as used at merge_intersecting_bb.py:76,87 (restated, unpinned) */
void orc_points_in_obb(const double* xyz, int64_t n, const double center[3], const double R[9], const double extent[3], uint8_t* inside) {
    for (int64_t i = 0; i < n; ++i) {
        const double d0 = xyz[3 * i] - center[0], d1 = xyz[3 * i + 1] - center[1], d2 = xyz[3 * i + 2] - center[2];
        int in = 1;
        for (int a = 0; a < 3; ++a) {
            const double pr = (d0 * R[a] + d1 * R[3 + a]) + d2 * R[6 + a];
            in &= (fabs(pr) <= extent[a] / 2);
        }
        inside[i] = (uint8_t)in;
    }
}
