// C-ABI layer of libf3d_hip.so (see include/f3d.h): context, scratch arena, host-pointer
// wrappers, and the tiny per-view host geometry.  No CPU fallback: every compute entry point
// needs a HIP device.
#include <hip/hip_runtime.h>
#include <vector>
#include <cstddef>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "f3d.h"
#include "f3d_kernels.h"
#include "f3d_math.h"

static_assert(sizeof(f3d_view) == 704, "f3d_view is 88 doubles");
static_assert(offsetof(f3d_view, t) == 72 && offsetof(f3d_view, mnorm) == 96, "the kernels stage M, t, mnorm as the record's first 15 doubles");
static_assert(offsetof(f3d_view, img_h) == offsetof(f3d_view, cull_n32) + 23 * 4, "cull planes, margins and image size: 24 consecutive floats");

#pragma clang fp contract(off)

namespace {

enum { SLOT_XYZ = 0, SLOT_OUT0, SLOT_OUT1, SLOT_VIEWS, SLOT_MASKS, SLOT_AUX0, SLOT_AUX1, SLOT_SORT_PERM, SLOT_SORT_SCRATCH,
       SLOT_TILED_MASKS, SLOT_TODO, SLOT_GRAPH, SLOT_GRAPH_BBOX, SLOT_PATCH,
       SLOT_GRP_ORDER, SLOT_GRP_KEYS, SLOT_GRP_STARTS, SLOT_GRP_SCRATCH, SLOT_OBB_TABLE, SLOT_OBB_FACETS, SLOT_OBB_CAND, SLOT_FUSE_TABLES, SLOT_FUSE_CARRY, SLOT_FUSE_XYZ, SLOT_COUNT };

thread_local char g_create_err[512] = "";

}  // namespace

struct f3d_ctx {
    int device;
    hipStream_t stream;
    char err[512];
    void* slot[SLOT_COUNT];
    size_t cap[SLOT_COUNT];
    int* dev_err;                       // sticky device error word: one bit per operation (F3D_DEVERR_*)
    int strict;                         // 1: a scratch buffer that would have to grow is F3D_ERR_NOMEM (allocation-free _dev calls)
    long long allocs;                   // device allocations made by this context so far (f3d_ctx_alloc_count)
    unsigned long long* table;          // open-addressing set of the uv2pt vote
    size_t table_slots;
    unsigned vote_gen;                  // generation stamp of the batched vote's set entries (the set is cleared when it wraps)
    bool table_stamped;                 // the table holds generation-stamped entries only (else: clear before a batched call)
    int* first_bad;                     // device int: first frame of the current batched call with an out-of-range index
    int* filter_dev;                    // filter_classes of the call being enqueued (device copy)
    int32_t filter_host[F3D_MAX_FILTER];  // its staging copy: must outlive the asynchronous upload
    unsigned long long* count_dev;
    f3d_codebook* codebook;             // vote-bin code book of the fused path (device)
    // radius graph: the grid of the last count pass (the fill pass must follow it for the same cloud)
    f3d_graphgrid graph_grid;
    int64_t graph_n;
    double graph_r2;
    const void* graph_xyz;
    // instance grouping of the last f3d_group_by_id call (host-pointer sequence group -> extremes -> hull filter)
    int64_t grp_n, grp_nids;
    int grp_dtype;
    // view-chunked fused call in progress (f3d_fuse_chunked_begin_dev .. the chunk with v_end == nviews)
    struct { int active, next, nviews, h, w, nclasses, gather; int64_t n; const int32_t* perm; const void* xyz; } chunk;
};

namespace {

int fail(f3d_ctx* ctx, int code, const char* fmt, ...) {
    char* dst = ctx ? ctx->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define F3D_HIP(ctx, call)                                                                               \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return fail(ctx, e_ == hipErrorOutOfMemory ? F3D_ERR_NOMEM : F3D_ERR_HIP,  \
                                          "%s: %s", #call, hipGetErrorString(e_));                       \
    } while (0)

int ensure(f3d_ctx* ctx, int s, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (ctx->cap[s] < bytes) {
        if (ctx->strict)
            return fail(ctx, F3D_ERR_NOMEM, "strict context: scratch slot %d holds %zu bytes, %zu needed (f3d_ctx_reserve first)", s, ctx->cap[s], bytes);
        if (ctx->slot[s]) { F3D_HIP(ctx, hipFree(ctx->slot[s])); ctx->slot[s] = nullptr; ctx->cap[s] = 0; }
        size_t want = bytes + bytes / 8;
        F3D_HIP(ctx, hipMalloc(&ctx->slot[s], want));
        ctx->cap[s] = want;
        ++ctx->allocs;
    }
    *out = ctx->slot[s];
    return F3D_OK;
}

hipStream_t pick(f3d_ctx* ctx, void* stream) { return stream ? (hipStream_t)stream : ctx->stream; }

int enter(f3d_ctx* ctx) {
    if (!ctx) return fail(nullptr, F3D_ERR_INVALID, "null context");
    ctx->err[0] = 0;
    F3D_HIP(ctx, hipSetDevice(ctx->device));
    return F3D_OK;
}

size_t xyz_bytes(f3d_dtype dt, int64_t n) { return (size_t)n * 3 * (dt == F3D_F64 ? 8 : 4); }

int make_filter(f3d_ctx* ctx, const int32_t* filter, int nfilter, int ncols, bool cols_must_exist, hipStream_t s,
                f3d_filter_args* fa) {
    fa->nfilter = 0; fa->cls_dev = nullptr; fa->cls_host = nullptr;
    for (int k = 0; k < 8; ++k) fa->cls[k] = -1;
    if (nfilter < 0 || nfilter > F3D_MAX_FILTER) return fail(ctx, F3D_ERR_INVALID, "nfilter %d out of range", nfilter);
    if (nfilter == 0 || !filter) {
        if (nfilter != 0) return fail(ctx, F3D_ERR_INVALID, "filter is NULL");
        return F3D_OK;
    }
    for (int k = 0; k < nfilter; ++k) {
        // votes[:, filter_classes] raises IndexError for a column that does not exist (voting.py:121)
        if (cols_must_exist && (filter[k] >= ncols || filter[k] < -ncols))
            return fail(ctx, F3D_ERR_INDEX, "filter class %d is out of bounds for %d vote columns", filter[k], ncols);
    }
    fa->nfilter = nfilter;
    int32_t tmp[F3D_MAX_FILTER];
    for (int k = 0; k < nfilter; ++k) tmp[k] = filter[k] < 0 ? filter[k] + ncols : filter[k];   // NumPy negative index
    if (nfilter <= 8) for (int k = 0; k < nfilter; ++k) fa->cls[k] = tmp[k];
    // the device copy always exists (the fused kernel reads the list from memory, whatever its length)
    memcpy(ctx->filter_host, tmp, sizeof(int32_t) * nfilter);
    F3D_HIP(ctx, hipMemcpyAsync(ctx->filter_dev, ctx->filter_host, sizeof(int32_t) * nfilter, hipMemcpyHostToDevice, s));
    fa->cls_dev = ctx->filter_dev;
    fa->cls_host = ctx->filter_host;
    return F3D_OK;
}

// Reads the sticky device error word and consumes the bits in `mask` (each operation owns one bit, so that an error
// recorded by one operation is never blamed on, or silently skips, another one that shares the context).
int take_error(f3d_ctx* ctx, hipStream_t s, int mask = F3D_DEVERR_ALL) {
    int e = 0;
    F3D_HIP(ctx, hipMemcpyAsync(&e, ctx->dev_err, sizeof(int), hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    e &= mask;
    if (e) {
        F3D_HIP(ctx, f3d_launch_clear_error_bits(ctx->dev_err, e, s));
        F3D_HIP(ctx, hipStreamSynchronize(s));
        if (e & F3D_DEVERR_FUSE)
            return fail(ctx, F3D_ERR_INDEX, "project_vote_argmax: a sampled mask label exceeds nclasses (the reference raises IndexError at voting.py:98)");
        if (e & F3D_DEVERR_VOTE)
            return fail(ctx, F3D_ERR_INDEX, "vote_uv2pt: point index or mask label out of bounds (the reference raises IndexError at voting.py:98)");
        if (e & F3D_DEVERR_CC)
            return fail(ctx, F3D_ERR_INDEX, "components_same_class: neighbour index out of bounds");
    }
    return F3D_OK;
}

// ---- host geometry, mirrored operation for operation by oracle/np_ref.py::frustum_data ------
void inv3(const double K[9], double o[9]) {
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const double A = e * i - f * h, B = c * h - b * i, C = b * f - c * e;
    const double D = f * g - d * i, E = a * i - c * g, F = c * d - a * f;
    const double G = d * h - e * g, H = b * g - a * h, I = a * e - b * d;
    const double det = (a * A + b * D) + c * G;
    o[0] = A / det; o[1] = B / det; o[2] = C / det;
    o[3] = D / det; o[4] = E / det; o[5] = F / det;
    o[6] = G / det; o[7] = H / det; o[8] = I / det;
}

struct frustum { double eye[3], lookat[3], normal[4][3]; };

void frustum_of(const double K[9], double w, double h, const double q[4], const double t[3], frustum* fr) {
    double Ki[9];
    inv3(K, Ki);
    const double pix[6][3] = {{0, 0, 0}, {0, 0, 1}, {w, 0, 1}, {w, h, 1}, {0, h, 1}, {w / 2, h / 2, 1}};
    double world[6][3];
    for (int k = 0; k < 6; ++k) {
        f3d_p3 c;
        c.x = (Ki[0] * pix[k][0] + Ki[1] * pix[k][1]) + Ki[2] * pix[k][2];      // camera_utils.py:86
        c.y = (Ki[3] * pix[k][0] + Ki[4] * pix[k][1]) + Ki[5] * pix[k][2];
        c.z = (Ki[6] * pix[k][0] + Ki[7] * pix[k][1]) + Ki[8] * pix[k][2];
        c.x = c.x / 1; c.y = c.y / 1; c.z = c.z / 1;                           // rescale = 1 (:111)
        const f3d_p3 r = f3d_rotate(q, c);                                      // :128-129 (forward rotation)
        world[k][0] = r.x + t[0]; world[k][1] = r.y + t[1]; world[k][2] = r.z + t[2];
    }
    for (int c = 0; c < 3; ++c) fr->eye[c] = world[0][c];
    {                                                                           // lookat = unit(centre - eye) (:148-150)
        const double v0 = world[5][0] - world[0][0], v1 = world[5][1] - world[0][1], v2 = world[5][2] - world[0][2];
        const double nn = sqrt((v0 * v0 + v1 * v1) + v2 * v2);
        fr->lookat[0] = v0 / nn; fr->lookat[1] = v1 / nn; fr->lookat[2] = v2 / nn;
    }
    for (int k = 0; k < 4; ++k) {                                               // camera_utils.py:163-170
        const double* ca = world[1 + k];
        const double* cb = world[1 + (k + 1) % 4];
        const double a0 = ca[0] - world[0][0], a1 = ca[1] - world[0][1], a2 = ca[2] - world[0][2];
        const double b0 = cb[0] - world[0][0], b1 = cb[1] - world[0][1], b2 = cb[2] - world[0][2];
        const double n0 = a1 * b2 - a2 * b1, n1 = a2 * b0 - a0 * b2, n2 = a0 * b1 - a1 * b0;
        const double nn = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
        fr->normal[k][0] = n0 / nn; fr->normal[k][1] = n1 / nn; fr->normal[k][2] = n2 / nn;
    }
}

int quat_inverse(const double q[4], double o[4]) {
    const double ss = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
    if (ss == 0.0) return F3D_ERR_ZERO_QUAT;
    o[0] = q[0] / ss; o[1] = -q[1] / ss; o[2] = -q[2] / ss; o[3] = -q[3] / ss;
    return F3D_OK;
}

}  // namespace

static int ensure_table(f3d_ctx* ctx, int64_t hw);

extern "C" {

int f3d_version(void) { return F3D_VERSION; }

f3d_ctx* f3d_ctx_create(int device) {
    g_create_err[0] = 0;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        fail(nullptr, F3D_ERR_HIP, "no HIP device available (%s); libf3d_hip has no CPU fallback",
             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= count) { fail(nullptr, F3D_ERR_INVALID, "device %d out of range [0,%d)", device, count); return nullptr; }
    f3d_ctx* ctx = (f3d_ctx*)calloc(1, sizeof(f3d_ctx));
    if (ctx) { ctx->graph_n = -1; ctx->grp_n = -1; }
    if (!ctx) { fail(nullptr, F3D_ERR_NOMEM, "out of host memory"); return nullptr; }
    ctx->device = device;
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void**)&ctx->dev_err, sizeof(int)) == hipSuccess &&
              hipMalloc((void**)&ctx->filter_dev, sizeof(int32_t) * F3D_MAX_FILTER) == hipSuccess &&
              hipMalloc((void**)&ctx->count_dev, sizeof(unsigned long long)) == hipSuccess &&
              hipMalloc((void**)&ctx->codebook, sizeof(f3d_codebook)) == hipSuccess &&
              hipMalloc((void**)&ctx->first_bad, sizeof(int)) == hipSuccess &&
              hipMemset(ctx->dev_err, 0, sizeof(int)) == hipSuccess &&
              hipMemset(ctx->codebook, 0, sizeof(f3d_codebook)) == hipSuccess;       // (the presence set is kept zero between calls)
    if (!ok) {
        fail(nullptr, F3D_ERR_HIP, "context setup failed: %s", hipGetErrorString(hipGetLastError()));
        f3d_ctx_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

void f3d_ctx_destroy(f3d_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int s = 0; s < SLOT_COUNT; ++s) if (ctx->slot[s]) (void)hipFree(ctx->slot[s]);
    if (ctx->dev_err) (void)hipFree(ctx->dev_err);
    if (ctx->table) (void)hipFree(ctx->table);
    if (ctx->filter_dev) (void)hipFree(ctx->filter_dev);
    if (ctx->count_dev) (void)hipFree(ctx->count_dev);
    if (ctx->codebook) (void)hipFree(ctx->codebook);
    if (ctx->first_bad) (void)hipFree(ctx->first_bad);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    free(ctx);
}

const char* f3d_last_error(const f3d_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

int f3d_ctx_synchronize(f3d_ctx* ctx) {
    int rc = enter(ctx); if (rc) return rc;
    F3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F3D_OK;
}

void* f3d_ctx_stream(f3d_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int f3d_ctx_reserve(f3d_ctx* ctx, int64_t n, int nviews, int h, int w) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || n > 0x7fffffffLL || nviews < 0 || h < 0 || w < 0) return fail(ctx, F3D_ERR_INVALID, "ctx_reserve: bad arguments");
    const int strict = ctx->strict;
    ctx->strict = 0;
    void* p;
    rc = ensure(ctx, SLOT_TODO, f3d_fuse_todo_bytes(n, nviews, F3D_CODE_MAX_NCLASSES), &p);   // (any number of classes)
    if (!rc && n > 0) rc = ensure(ctx, SLOT_SORT_PERM, (size_t)n * 4, &p);
    if (!rc && n > 0) rc = ensure(ctx, SLOT_SORT_SCRATCH, f3d_sort_scratch_bytes(n), &p);
    if (!rc && nviews > 0 && h > 0 && w > 0) rc = ensure(ctx, SLOT_TILED_MASKS, f3d_coded_masks_bytes(nviews, h, w), &p);
    if (!rc && h > 0 && w > 0) rc = ensure_table(ctx, (int64_t)h * w);
    if (!rc) rc = ensure(ctx, SLOT_FUSE_TABLES, f3d_fuse_tables_bytes(nviews > 0 ? nviews : 1), &p);
    ctx->strict = strict;
    return rc;
}

int f3d_ctx_set_strict(f3d_ctx* ctx, int strict) {
    if (!ctx) return F3D_ERR_INVALID;
    ctx->strict = strict ? 1 : 0;
    return F3D_OK;
}

long long f3d_ctx_alloc_count(const f3d_ctx* ctx) { return ctx ? ctx->allocs : -1; }

// ---------------------------------------------------------------------------------------------
// host geometry
// ---------------------------------------------------------------------------------------------
int f3d_quat_inverse(const double q[4], double o[4]) {
    if (!q || !o) return F3D_ERR_INVALID;
    return quat_inverse(q, o);
}

int f3d_frustum_data(const double K[9], double w, double h, const double* q, const double* t, int nviews,
                     double* eyes, double* lookats, double* face_normals) {
    if (!K || !q || !t || nviews < 0) return F3D_ERR_INVALID;
    for (int v = 0; v < nviews; ++v) {
        frustum fr;
        frustum_of(K, w, h, q + 4 * v, t + 3 * v, &fr);
        if (eyes) memcpy(eyes + 3 * v, fr.eye, sizeof fr.eye);
        if (lookats) memcpy(lookats + 3 * v, fr.lookat, sizeof fr.lookat);
        if (face_normals) memcpy(face_normals + 12 * v, fr.normal, sizeof fr.normal);
    }
    return F3D_OK;
}

int f3d_views_build(const double K[9], double w, double h, const double* q, const double* t, int nviews, double max_depth,
                    f3d_view* out) {
    if (!K || !q || !t || !out || nviews < 0) return F3D_ERR_INVALID;
    for (int v = 0; v < nviews; ++v) {
        f3d_view* vw = out + v;
        memset(vw, 0, sizeof *vw);
        memcpy(vw->K, K, sizeof vw->K);
        const int rc = quat_inverse(q + 4 * v, vw->qinv);
        if (rc) return rc;
        memcpy(vw->t, t + 3 * v, sizeof vw->t);
        frustum fr;
        frustum_of(K, w, h, q + 4 * v, t + 3 * v, &fr);
        for (int m = 0; m < 4; ++m) {                                           // fusion.py:254 (spoke origins = eye)
            memcpy(vw->plane_pt[m], fr.eye, sizeof fr.eye);
            memcpy(vw->plane_n[m], fr.normal[m], sizeof fr.normal[m]);
        }
        for (int c = 0; c < 3; ++c) {                                           // fusion.py:255-256
            vw->plane_pt[4][c] = fr.eye[c] + max_depth * fr.lookat[c];
            vw->plane_n[4][c] = -fr.lookat[c];
        }
        double l1max = 0.0, nmax = 1.0;
        for (int m = 0; m < F3D_NPLANES; ++m) {
            const double nl1 = fabs(vw->plane_n[m][0]) + fabs(vw->plane_n[m][1]) + fabs(vw->plane_n[m][2]);
            const double l1 = (fabs(vw->plane_pt[m][0]) + fabs(vw->plane_pt[m][1]) + fabs(vw->plane_pt[m][2])) * (nl1 > 1 ? nl1 : 1);
            if (l1 > l1max) l1max = l1;
            for (int c = 0; c < 3; ++c) if (fabs(vw->plane_n[m][c]) > nmax) nmax = fabs(vw->plane_n[m][c]);
        }
        // float32 pre-cull a = n32.p32 - off32: inputs rounded to f32 (2^-24 relative each) + 3 f32 FMAs, against the
        // exact plane value that itself carries ~12 eps64 of rounding: 32 * 2^-24 * (|p|_1 + |pp|_1) covers both, 2x margin
        const double eps32 = 32.0 * 5.9604644775390625e-08;
        for (int m = 0; m < F3D_NPLANES; ++m) {
            for (int c = 0; c < 3; ++c) vw->cull_n32[m][c] = (float)vw->plane_n[m][c];
            const double off = fma(vw->plane_n[m][0], vw->plane_pt[m][0],
                               fma(vw->plane_n[m][1], vw->plane_pt[m][1], vw->plane_n[m][2] * vw->plane_pt[m][2]));
            vw->cull_off32[m] = (float)off;
            vw->plane_off[m] = off;
        }
        // float64 refinement: FMA value and exact value are both within ~12 eps64 * (|p|_1 + |pp|_1) of the real number
        vw->cull_rel64 = 64.0 * 2.220446049250313e-16 * nmax;
        vw->cull_abs64 = 64.0 * 2.220446049250313e-16 * l1max + 1e-300;
        vw->img_w = (float)w; vw->img_h = (float)h;
        vw->cull_rel32 = (float)(eps32 * nmax * 1.0000002);
        vw->cull_abs32 = (float)(eps32 * l1max * 1.0000002 + 1e-30);
        // fast projection operator M = K * Rot(qinv), Rot = the matrix of x -> q x q* for the un-normalised q
        {
            const long double w_ = vw->qinv[0], x = vw->qinv[1], y = vw->qinv[2], z = vw->qinv[3];
            const long double R[9] = {w_ * w_ + x * x - y * y - z * z, 2 * (x * y - w_ * z), 2 * (x * z + w_ * y),
                                      2 * (x * y + w_ * z), w_ * w_ - x * x + y * y - z * z, 2 * (y * z - w_ * x),
                                      2 * (x * z - w_ * y), 2 * (y * z + w_ * x), w_ * w_ - x * x - y * y + z * z};
            const long double q2 = w_ * w_ + x * x + y * y + z * z;
            for (int r = 0; r < 3; ++r) {
                long double l1 = 0;
                for (int c = 0; c < 3; ++c) {
                    long double acc = 0;
                    for (int k = 0; k < 3; ++k) acc += (long double)K[3 * r + k] * R[3 * k + c];
                    vw->M[3 * r + c] = (double)acc;
                    vw->M32[3 * r + c] = (float)vw->M[3 * r + c];
                    l1 += fabsl((long double)K[3 * r + c]);
                }
                vw->mnorm[r] = (double)(l1 * q2 * 1.000000001L);
            }
        }
    }
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// a1 rotate
// ---------------------------------------------------------------------------------------------
int f3d_rotate_f64(f3d_ctx* ctx, const double* xyz, int64_t n, const double q[4], double* out) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || (n > 0 && (!xyz || !out)) || !q) return fail(ctx, F3D_ERR_INVALID, "rotate: bad arguments");
    if (n == 0) return F3D_OK;
    void *din, *dout;
    if ((rc = ensure(ctx, SLOT_XYZ, (size_t)n * 24, &din))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT0, (size_t)n * 24, &dout))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(din, xyz, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    F3D_HIP(ctx, f3d_launch_rotate((const double*)din, n, q, (double*)dout, ctx->stream));
    F3D_HIP(ctx, hipMemcpyAsync(out, dout, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
    F3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F3D_OK;
}

int f3d_rotate_f64_dev(f3d_ctx* ctx, const double* xyz, int64_t n, const double q[4], double* out, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || (n > 0 && (!xyz || !out)) || !q) return fail(ctx, F3D_ERR_INVALID, "rotate: bad arguments");
    if (n == 0) return F3D_OK;
    F3D_HIP(ctx, f3d_launch_rotate(xyz, n, q, out, pick(ctx, stream)));
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// (f)#3 depth frame -> world points
// ---------------------------------------------------------------------------------------------
static size_t depth_bytes(int depth_type, int64_t n) { return (size_t)n * (depth_type == F3D_DEPTH_U16 ? 2 : depth_type == F3D_DEPTH_F32 ? 4 : 8); }

int f3d_unproject_depth_dev(f3d_ctx* ctx, const void* depth, int depth_type, int h, int w, const double K[9], double depth_scale,
                            const double q_wxyz[4], const double t[3], double* xyz, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (h < 0 || w < 0 || !K || !q_wxyz || !t || (depth_type != F3D_DEPTH_U16 && depth_type != F3D_DEPTH_F32 && depth_type != F3D_DEPTH_F64) ||
        ((int64_t)h * w > 0 && (!depth || !xyz)))
        return fail(ctx, F3D_ERR_INVALID, "unproject_depth: bad arguments");
    F3D_HIP(ctx, f3d_launch_unproject_depth(depth, depth_type, h, w, K, depth_scale, q_wxyz, t, xyz, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_unproject_depth_batch_dev(f3d_ctx* ctx, const void* depth, int depth_type, int nframes, int h, int w, const double K[9], double depth_scale,
                                  const double* q_wxyz, const double* t, double* xyz, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (nframes < 0 || h < 0 || w < 0 || !K || (depth_type != F3D_DEPTH_U16 && depth_type != F3D_DEPTH_F32 && depth_type != F3D_DEPTH_F64) ||
        (nframes > 0 && (!q_wxyz || !t)) || ((int64_t)nframes * h * w > 0 && (!depth || !xyz)))
        return fail(ctx, F3D_ERR_INVALID, "unproject_depth_batch: bad arguments");
    if (nframes == 0) return F3D_OK;
    F3D_HIP(ctx, f3d_launch_unproject_depth_batch(depth, depth_type, nframes, h, w, K, depth_scale, q_wxyz, t, xyz, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_unproject_depth(f3d_ctx* ctx, const void* depth, int depth_type, int h, int w, const double K[9], double depth_scale,
                        const double q_wxyz[4], const double t[3], double* xyz) {
    int rc = enter(ctx); if (rc) return rc;
    const int64_t n = (int64_t)h * w;
    if (h < 0 || w < 0 || (n > 0 && (!depth || !xyz))) return fail(ctx, F3D_ERR_INVALID, "unproject_depth: bad arguments");
    if (depth_type != F3D_DEPTH_U16 && depth_type != F3D_DEPTH_F32 && depth_type != F3D_DEPTH_F64)
        return fail(ctx, F3D_ERR_INVALID, "unproject_depth: unknown depth type %d", depth_type);
    if (n == 0) return F3D_OK;
    void *din, *dout;
    if ((rc = ensure(ctx, SLOT_AUX0, depth_bytes(depth_type, n), &din))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT0, (size_t)n * 24, &dout))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(din, depth, depth_bytes(depth_type, n), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = f3d_unproject_depth_dev(ctx, din, depth_type, h, w, K, depth_scale, q_wxyz, t, (double*)dout, ctx->stream))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(xyz, dout, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
    F3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// a2 / a4 / single-view fused
// ---------------------------------------------------------------------------------------------
int f3d_project_view_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_view* view, int32_t* uv,
                         uint8_t* inside, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || !view || (!uv && !inside) || (n > 0 && !xyz)) return fail(ctx, F3D_ERR_INVALID, "project_view: bad arguments");
    F3D_HIP(ctx, f3d_launch_project_view(xyz, dtype, n, *view, uv, inside, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_project_view_f64(f3d_ctx* ctx, const double* xyz, int64_t n, const f3d_view* view, int32_t* uv, uint8_t* inside) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || !view || (!uv && !inside) || (n > 0 && !xyz)) return fail(ctx, F3D_ERR_INVALID, "project_view: bad arguments");
    if (n == 0) return F3D_OK;
    void *din, *duv = nullptr, *din_s = nullptr;
    if ((rc = ensure(ctx, SLOT_XYZ, (size_t)n * 24, &din))) return rc;
    if (uv && (rc = ensure(ctx, SLOT_OUT0, (size_t)n * 8, &duv))) return rc;
    if (inside && (rc = ensure(ctx, SLOT_OUT1, (size_t)n, &din_s))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(din, xyz, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    F3D_HIP(ctx, f3d_launch_project_view(din, F3D_F64, n, *view, (int32_t*)duv, (uint8_t*)din_s, ctx->stream));
    if (uv) F3D_HIP(ctx, hipMemcpyAsync(uv, duv, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (inside) F3D_HIP(ctx, hipMemcpyAsync(inside, din_s, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    F3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F3D_OK;
}

static int pose_view(f3d_ctx* ctx, const double K[9], const double q[4], const double t[3], f3d_view* vw) {
    if (!K || !q || !t) return fail(ctx, F3D_ERR_INVALID, "points2pixel: NULL camera");
    memset(vw, 0, sizeof *vw);
    memcpy(vw->K, K, sizeof vw->K);
    memcpy(vw->t, t, sizeof vw->t);
    if (quat_inverse(q, vw->qinv)) return fail(ctx, F3D_ERR_ZERO_QUAT, "a zero quaternion cannot be inverted");
    return F3D_OK;
}

int f3d_points2pixel_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const double K[9], const double q[4],
                         const double t[3], int32_t* uv, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    f3d_view vw;
    if ((rc = pose_view(ctx, K, q, t, &vw))) return rc;
    if (n < 0 || !uv || (n > 0 && !xyz)) return fail(ctx, F3D_ERR_INVALID, "points2pixel: bad arguments");
    F3D_HIP(ctx, f3d_launch_project_view(xyz, dtype, n, vw, uv, nullptr, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_points2pixel_f64(f3d_ctx* ctx, const double* xyz, int64_t n, const double K[9], const double q[4], const double t[3],
                         int32_t* uv) {
    int rc = enter(ctx); if (rc) return rc;
    f3d_view vw;
    if ((rc = pose_view(ctx, K, q, t, &vw))) return rc;
    return f3d_project_view_f64(ctx, xyz, n, &vw, uv, nullptr);
}

int f3d_inside_polyhedra_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const double* plane_pts,
                             const double* normals, int m, uint8_t* inside, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || m < 0 || !inside || (n > 0 && !xyz) || (m > 0 && (!plane_pts || !normals)))
        return fail(ctx, F3D_ERR_INVALID, "inside_polyhedra: bad arguments");
    hipStream_t s = pick(ctx, stream);
    int done = 0;
    do {                                                    // m == 0: every point is inside (signsum == 0 == len)
        f3d_plane_args pa;
        pa.m = (m - done) < F3D_PLANES_PER_LAUNCH ? (m - done) : F3D_PLANES_PER_LAUNCH;
        pa.accumulate = done > 0;
        memcpy(pa.pt, plane_pts + 3 * done, sizeof(double) * 3 * pa.m);
        memcpy(pa.n, normals + 3 * done, sizeof(double) * 3 * pa.m);
        F3D_HIP(ctx, f3d_launch_inside_polyhedra(xyz, dtype, n, pa, inside, s));
        done += pa.m;
    } while (done < m);
    return F3D_OK;
}

int f3d_inside_polyhedra_f64(f3d_ctx* ctx, const double* xyz, int64_t n, const double* plane_pts, const double* normals, int m,
                             uint8_t* inside) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || !inside || (n > 0 && !xyz)) return fail(ctx, F3D_ERR_INVALID, "inside_polyhedra: bad arguments");
    if (n == 0) return F3D_OK;
    void *din, *dout;
    if ((rc = ensure(ctx, SLOT_XYZ, (size_t)n * 24, &din))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT1, (size_t)n, &dout))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(din, xyz, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = f3d_inside_polyhedra_dev(ctx, din, F3D_F64, n, plane_pts, normals, m, (uint8_t*)dout, ctx->stream))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(inside, dout, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    F3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// fused multi-view path
// ---------------------------------------------------------------------------------------------
int f3d_cloud_sort_cells_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, void* sorted_xyz, int32_t* perm,
                             void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || n > 0x7fffffffLL || (n > 0 && (!xyz || !perm)))
        return fail(ctx, F3D_ERR_INVALID, "cloud_sort_cells: bad arguments (n < 2^31)");
    if (n == 0) return F3D_OK;
    void* scratch;
    if ((rc = ensure(ctx, SLOT_SORT_SCRATCH, f3d_sort_scratch_bytes(n), &scratch))) return rc;   // grows on first use only
    F3D_HIP(ctx, f3d_launch_cell_sort(xyz, dtype, n, sorted_xyz, perm, scratch, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_project_vote_argmax_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_view* views_dev, int nviews,
                                const uint8_t* masks, int h, int w, int nclasses, const int32_t* filter, int nfilter,
                                double threshold, int64_t* classes, uint16_t* votes_u16, unsigned flags, const int32_t* perm,
                                void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || nviews < 0 || h <= 0 || w <= 0 || nclasses < 0 || nclasses > 65534 || !classes ||
        (n > 0 && !xyz) || (nviews > 0 && (!views_dev || !masks)))
        return fail(ctx, F3D_ERR_INVALID, "project_vote_argmax: bad arguments");
    if (nviews > 65535) return fail(ctx, F3D_ERR_INVALID, "project_vote_argmax: at most 65535 views");
    if ((int64_t)h * (int64_t)w >= (int64_t)1 << 31) return fail(ctx, F3D_ERR_INVALID, "project_vote_argmax: mask of %d x %d pixels is too large", h, w);
    if ((flags & F3D_FUSE_SORT) && perm) return fail(ctx, F3D_ERR_INVALID, "project_vote_argmax: F3D_FUSE_SORT and perm are exclusive");
    hipStream_t s = pick(ctx, stream);
    f3d_filter_args fa;
    if ((rc = make_filter(ctx, filter, nfilter, nclasses + 1, true, s, &fa))) return rc;
    bool gather = (flags & F3D_FUSE_GATHER) && perm;
    const bool sort = (flags & F3D_FUSE_SORT) && n > 512 && n <= 0x7fffffffLL;
    // the accelerated kernels address the coded masks with 32-bit offsets; beyond 4 GiB of them the exact kernel labels every point
    const bool coded = nviews > 0 && nclasses <= F3D_CODE_MAX_NCLASSES && f3d_coded_masks_bytes(nviews, h, w) < ((size_t)1 << 32);
    // (Measured and dropped: coding the masks on a second stream, forked from and joined into `stream` with events, while the cloud is
    // sorted -- the two event dependencies cost more than the ~50 us of overlap they buy: 1.32 ms per C3 step instead of 1.26.)
    if (sort) {
        void *sperm, *scratch;                                                                  // grow on first use only
        if ((rc = ensure(ctx, SLOT_SORT_PERM, (size_t)n * 4, &sperm))) return rc;
        if ((rc = ensure(ctx, SLOT_SORT_SCRATCH, f3d_sort_scratch_bytes(n), &scratch))) return rc;
        F3D_HIP(ctx, f3d_launch_cell_sort(xyz, dtype, n, nullptr, (int32_t*)sperm, scratch, s));
        perm = (const int32_t*)sperm; gather = true;                                            // the kernel reads xyz[perm[i]]
    }
    const uint8_t* cmasks = nullptr;                                                            // coded, tiled copy for the fast kernel
    void *todo, *tables;                                                                        // grow on first use only
    if ((rc = ensure(ctx, SLOT_TODO, f3d_fuse_todo_bytes(n, nviews, nclasses), &todo))) return rc;
    if ((rc = ensure(ctx, SLOT_FUSE_TABLES, f3d_fuse_tables_bytes(nviews > 0 ? nviews : 1), &tables))) return rc;
    if (coded) {
        void* tm;                                                                               // grows on first use only
        if ((rc = ensure(ctx, SLOT_TILED_MASKS, f3d_coded_masks_bytes(nviews, h, w), &tm))) return rc;
        F3D_HIP(ctx, f3d_launch_code_masks_with_setup(masks, (uint8_t*)tm, nviews, h, w, nclasses, fa, votes_u16 != nullptr, ctx->codebook, views_dev, tables,
                                                      threshold, (unsigned int*)todo, s));
        cmasks = (const uint8_t*)tm;
    }
    F3D_HIP(ctx, f3d_launch_fuse(xyz, dtype, n, views_dev, nviews, masks, cmasks, h, w, nclasses, fa, threshold, classes, votes_u16,
                                 ctx->dev_err, perm, gather, (unsigned int*)todo, (int32_t*)((char*)todo + 16), ctx->codebook, tables,
                                 0, nviews, nullptr, nullptr, s));
    return F3D_OK;
}

// ---- the same path with the views arriving in chunks (multi-GPU: the masks of chunk c+1 are still in flight while chunk c votes)
int f3d_mask_presence_dev(f3d_ctx* ctx, const uint8_t* masks, int nviews, int h, int w, uint8_t* present256, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (nviews < 0 || h <= 0 || w <= 0 || !present256 || (nviews > 0 && !masks)) return fail(ctx, F3D_ERR_INVALID, "mask_presence: bad arguments");
    hipStream_t s = pick(ctx, stream);
    F3D_HIP(ctx, f3d_launch_mask_presence(masks, (int64_t)nviews * h * w, ctx->codebook, s));
    F3D_HIP(ctx, f3d_launch_presence_bytes(ctx->codebook, present256, true, s));
    return F3D_OK;
}

int f3d_fuse_chunked_begin_dev(f3d_ctx* ctx, const uint8_t* present256, int64_t n, int nviews, int h, int w, int nclasses,
                               const int32_t* filter, int nfilter, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    ctx->chunk.active = 0;
    if (n < 0 || n > 0x7ffff000LL || nviews <= 0 || nviews > 255 || h <= 0 || w <= 0 || nclasses < 0 || nclasses > F3D_CODE_MAX_NCLASSES)
        return fail(ctx, F3D_ERR_INVALID, "fuse_chunked_begin: bad arguments (1..255 views, nclasses <= %d)", F3D_CODE_MAX_NCLASSES);
    if (f3d_coded_masks_bytes(nviews, h, w) >= ((size_t)1 << 32))
        return fail(ctx, F3D_ERR_INVALID, "fuse_chunked_begin: %d coded masks of %d x %d exceed 4 GiB; use f3d_project_vote_argmax_dev", nviews, h, w);
    hipStream_t s = pick(ctx, stream);
    f3d_filter_args fa;
    if ((rc = make_filter(ctx, filter, nfilter, nclasses + 1, true, s, &fa))) return rc;
    void* p;                                                   // every scratch buffer of the chunk calls: they allocate nothing
    if ((rc = ensure(ctx, SLOT_TILED_MASKS, f3d_coded_masks_bytes(nviews, h, w), &p))) return rc;
    if ((rc = ensure(ctx, SLOT_TODO, f3d_fuse_todo_bytes(n, nviews, nclasses), &p))) return rc;
    if ((rc = ensure(ctx, SLOT_FUSE_TABLES, f3d_fuse_tables_bytes(nviews), &p))) return rc;
    if ((rc = ensure(ctx, SLOT_FUSE_CARRY, f3d_fuse_carry_bytes(n, nclasses), &p))) return rc;
    if (n > 0) {                                               // F3D_FUSE_SORT / a gathered cloud: permutation, sort scratch, cell-order copy
        if ((rc = ensure(ctx, SLOT_SORT_PERM, (size_t)n * 4, &p))) return rc;
        if ((rc = ensure(ctx, SLOT_SORT_SCRATCH, f3d_sort_scratch_bytes(n), &p))) return rc;
        if ((rc = ensure(ctx, SLOT_FUSE_XYZ, (size_t)n * 24, &p))) return rc;
    }
    if (present256) F3D_HIP(ctx, f3d_launch_presence_bytes(ctx->codebook, const_cast<uint8_t*>(present256), false, s));
    else F3D_HIP(ctx, hipMemsetAsync(ctx->codebook->presence, 0xFF, sizeof ctx->codebook->presence, s));     // every label gets a bin
    F3D_HIP(ctx, f3d_launch_code_book(ctx->codebook, nclasses, fa, false, s));
    ctx->chunk.active = 1; ctx->chunk.next = 0; ctx->chunk.nviews = nviews; ctx->chunk.h = h; ctx->chunk.w = w;
    ctx->chunk.nclasses = nclasses; ctx->chunk.n = n; ctx->chunk.perm = nullptr; ctx->chunk.gather = 0; ctx->chunk.xyz = nullptr;
    return F3D_OK;
}

int f3d_code_planes_dev(f3d_ctx* ctx, const uint8_t* masks, int nplanes, int h, int w, uint8_t* coded, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (!ctx->chunk.active || ctx->chunk.h != h || ctx->chunk.w != w)
        return fail(ctx, F3D_ERR_INVALID, "code_planes: no view-chunked call with masks of %d x %d is in progress (f3d_fuse_chunked_begin_dev first)", h, w);
    if (nplanes < 0 || (nplanes > 0 && (!masks || !coded)) || ((uintptr_t)coded & 7)) return fail(ctx, F3D_ERR_INVALID, "code_planes: bad arguments (coded: 8-byte aligned)");
    F3D_HIP(ctx, f3d_launch_code_planes(masks, coded, nplanes, h, w, ctx->codebook, pick(ctx, stream)));
    return F3D_OK;
}

static int fuse_chunk_impl(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_view* views_dev, int nviews, int v_begin, int v_end,
                           const uint8_t* masks, const uint8_t* coded, int h, int w, int nclasses, const int32_t* filter, int nfilter, double threshold,
                           int64_t* classes, unsigned flags, const int32_t* perm, void* stream);

int f3d_fuse_chunk_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_view* views_dev, int nviews, int v_begin, int v_end,
                       const uint8_t* masks, int h, int w, int nclasses, const int32_t* filter, int nfilter, double threshold,
                       int64_t* classes, unsigned flags, const int32_t* perm, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (!masks) return fail(ctx, F3D_ERR_INVALID, "fuse_chunk: bad arguments");
    return fuse_chunk_impl(ctx, xyz, dtype, n, views_dev, nviews, v_begin, v_end, masks, nullptr, h, w, nclasses, filter, nfilter, threshold, classes, flags, perm, stream);
}

int f3d_fuse_chunk_coded_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_view* views_dev, int nviews, int v_begin, int v_end,
                             const uint8_t* coded, int h, int w, int nclasses, const int32_t* filter, int nfilter, double threshold,
                             int64_t* classes, unsigned flags, const int32_t* perm, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (!coded || ((uintptr_t)coded & 7)) return fail(ctx, F3D_ERR_INVALID, "fuse_chunk_coded: bad arguments (coded: 8-byte aligned)");
    return fuse_chunk_impl(ctx, xyz, dtype, n, views_dev, nviews, v_begin, v_end, nullptr, coded, h, w, nclasses, filter, nfilter, threshold, classes, flags, perm, stream);
}

// masks: raw planes [nviews, H, W], coded here into context scratch -- or coded: planes some rank coded already (f3d_code_planes_dev with the
// same book), [nviews] x f3d_coded_plane_bytes, used where they lie
static int fuse_chunk_impl(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_view* views_dev, int nviews, int v_begin, int v_end,
                           const uint8_t* masks, const uint8_t* coded, int h, int w, int nclasses, const int32_t* filter, int nfilter, double threshold,
                           int64_t* classes, unsigned flags, const int32_t* perm, void* stream) {
    int rc;
    if (!ctx->chunk.active || ctx->chunk.next != v_begin || ctx->chunk.nviews != nviews || ctx->chunk.h != h || ctx->chunk.w != w ||
        ctx->chunk.nclasses != nclasses || ctx->chunk.n != n || v_end <= v_begin || v_end > nviews)
        return fail(ctx, F3D_ERR_INVALID, "fuse_chunk: views [%d, %d) do not continue the call begun with f3d_fuse_chunked_begin_dev "
                    "(next view %d of %d, same n / h / w / nclasses required)", v_begin, v_end, ctx->chunk.active ? ctx->chunk.next : -1, ctx->chunk.nviews);
    if (!classes || (n > 0 && !xyz) || !views_dev) return fail(ctx, F3D_ERR_INVALID, "fuse_chunk: bad arguments");
    if ((flags & F3D_FUSE_SORT) && perm) return fail(ctx, F3D_ERR_INVALID, "fuse_chunk: F3D_FUSE_SORT and perm are exclusive");
    hipStream_t s = pick(ctx, stream);
    f3d_filter_args fa;
    if ((rc = make_filter(ctx, filter, nfilter, nclasses + 1, true, s, &fa))) return rc;
    if (v_begin == 0) {                                        // the point order is fixed by the first chunk
        ctx->chunk.perm = perm; ctx->chunk.gather = ((flags & F3D_FUSE_GATHER) && perm) ? 1 : 0;
        if ((flags & F3D_FUSE_SORT) && n > 512) {
            void *sperm, *scratch;
            if ((rc = ensure(ctx, SLOT_SORT_PERM, (size_t)n * 4, &sperm))) return rc;
            if ((rc = ensure(ctx, SLOT_SORT_SCRATCH, f3d_sort_scratch_bytes(n), &scratch))) return rc;
            F3D_HIP(ctx, f3d_launch_cell_sort(xyz, dtype, n, nullptr, (int32_t*)sperm, scratch, s));
            ctx->chunk.perm = (const int32_t*)sperm; ctx->chunk.gather = 1;
        }
        ctx->chunk.xyz = xyz;
    }
    // a cloud read through a permutation and more chunks to come: the first chunk leaves it behind in cell order (context scratch),
    // the later chunks stream that copy instead of gathering 24-byte points again
    void* keep = nullptr;
    if (v_begin == 0 && v_end < nviews && ctx->chunk.gather && n > 0) {
        if ((rc = ensure(ctx, SLOT_FUSE_XYZ, (size_t)n * 3 * (dtype == F3D_F64 ? 8 : 4), &keep))) return rc;
    }
    const void* cxyz = ctx->chunk.xyz;
    const bool cgather = ctx->chunk.gather != 0;
    void *tm, *todo, *tables, *carry;
    if ((rc = ensure(ctx, SLOT_TILED_MASKS, f3d_coded_masks_bytes(nviews, h, w), &tm))) return rc;
    if ((rc = ensure(ctx, SLOT_TODO, f3d_fuse_todo_bytes(n, nviews, nclasses), &todo))) return rc;
    if ((rc = ensure(ctx, SLOT_FUSE_TABLES, f3d_fuse_tables_bytes(nviews), &tables))) return rc;
    if ((rc = ensure(ctx, SLOT_FUSE_CARRY, f3d_fuse_carry_bytes(n, nclasses), &carry))) return rc;
    const size_t plane = f3d_coded_masks_bytes(1, h, w);
    if (coded) tm = const_cast<uint8_t*>(coded);                // the exchange delivered coded planes: no coding, no raw masks, the exact tier reads codes
    else F3D_HIP(ctx, f3d_launch_code_planes(masks + (size_t)v_begin * h * w, (uint8_t*)tm + (size_t)v_begin * plane, v_end - v_begin, h, w, ctx->codebook, s));
    F3D_HIP(ctx, f3d_launch_fuse_setup(views_dev, v_begin, v_end, tables, ctx->codebook, threshold, v_begin == 0 ? (unsigned int*)todo : nullptr, s));
    F3D_HIP(ctx, f3d_launch_fuse(cxyz, dtype, n, views_dev, nviews, masks, (const uint8_t*)tm, h, w, nclasses, fa, threshold, classes, nullptr,
                                 ctx->dev_err, ctx->chunk.perm, cgather, (unsigned int*)todo, (int32_t*)((char*)todo + 16),
                                 ctx->codebook, tables, v_begin, v_end, (uint32_t*)carry, keep, s));
    if (keep) { ctx->chunk.xyz = keep; ctx->chunk.gather = 0; }
    ctx->chunk.next = v_end;
    if (v_end == nviews) ctx->chunk.active = 0;
    return F3D_OK;
}

int f3d_debug_fastpath_audit(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_view* views, int nviews, int w, int h,
                             uint64_t stats[4]) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || n > 0x7fffffffLL || nviews < 0 || w <= 0 || h <= 0 || !stats || (n > 0 && !xyz) || (nviews > 0 && !views))
        return fail(ctx, F3D_ERR_INVALID, "fastpath_audit: bad arguments");
    void *dxyz, *dsorted, *dperm, *dviews, *dstats, *scratch;
    if ((rc = ensure(ctx, SLOT_XYZ, xyz_bytes(dtype, n), &dxyz))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT0, xyz_bytes(dtype, n), &dsorted))) return rc;
    if ((rc = ensure(ctx, SLOT_SORT_PERM, (size_t)n * 4, &dperm))) return rc;
    if ((rc = ensure(ctx, SLOT_SORT_SCRATCH, f3d_sort_scratch_bytes(n), &scratch))) return rc;
    if ((rc = ensure(ctx, SLOT_VIEWS, sizeof(f3d_view) * (size_t)nviews, &dviews))) return rc;
    if ((rc = ensure(ctx, SLOT_AUX0, 64, &dstats))) return rc;
    hipStream_t s = ctx->stream;
    if (n) F3D_HIP(ctx, hipMemcpyAsync(dxyz, xyz, xyz_bytes(dtype, n), hipMemcpyHostToDevice, s));
    if (nviews) F3D_HIP(ctx, hipMemcpyAsync(dviews, views, sizeof(f3d_view) * (size_t)nviews, hipMemcpyHostToDevice, s));
    // waves of 64 consecutive points must be spatial neighbours, as in the fused call: audit the cell-sorted copy
    if (n) F3D_HIP(ctx, f3d_launch_cell_sort(dxyz, dtype, n, dsorted, (int32_t*)dperm, scratch, s));
    F3D_HIP(ctx, f3d_launch_fastpath_audit(dsorted, dtype, n, (const f3d_view*)dviews, nviews, w, h, (unsigned long long*)dstats, s));
    F3D_HIP(ctx, hipMemcpyAsync(stats, dstats, 32, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

int f3d_debug_fuse_deferred(f3d_ctx* ctx, void* stream, uint32_t counts[2]) {
    int rc = enter(ctx); if (rc) return rc;
    if (!counts || !ctx->slot[SLOT_TODO]) return fail(ctx, F3D_ERR_INVALID, "fuse_deferred: no fused call has run in this context");
    hipStream_t s = pick(ctx, stream);
    F3D_HIP(ctx, hipMemcpyAsync(counts, ctx->slot[SLOT_TODO], 8, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

int f3d_take_device_error(f3d_ctx* ctx, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    return take_error(ctx, pick(ctx, stream));
}

int f3d_project_vote_argmax(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_view* views, int nviews,
                            const uint8_t* masks, int h, int w, int nclasses, const int32_t* filter, int nfilter,
                            double threshold, int64_t* classes, uint16_t* votes_u16) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || nviews < 0 || h <= 0 || w <= 0 || nclasses < 0 || !classes || (n > 0 && !xyz) || (nviews > 0 && (!views || !masks)))
        return fail(ctx, F3D_ERR_INVALID, "project_vote_argmax: bad arguments");
    if (n == 0) return F3D_OK;
    const size_t mbytes = (size_t)nviews * h * w, ncols = (size_t)nclasses + 1;
    void *dxyz, *dviews, *dmasks, *dcls, *dvotes = nullptr;
    if ((rc = ensure(ctx, SLOT_XYZ, xyz_bytes(dtype, n), &dxyz))) return rc;
    if ((rc = ensure(ctx, SLOT_VIEWS, sizeof(f3d_view) * (size_t)nviews, &dviews))) return rc;
    if ((rc = ensure(ctx, SLOT_MASKS, mbytes, &dmasks))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT0, (size_t)n * 8, &dcls))) return rc;
    if (votes_u16 && (rc = ensure(ctx, SLOT_OUT1, (size_t)n * ncols * 2, &dvotes))) return rc;
    hipStream_t s = ctx->stream;
    F3D_HIP(ctx, hipMemcpyAsync(dxyz, xyz, xyz_bytes(dtype, n), hipMemcpyHostToDevice, s));
    if (nviews) {
        F3D_HIP(ctx, hipMemcpyAsync(dviews, views, sizeof(f3d_view) * (size_t)nviews, hipMemcpyHostToDevice, s));
        F3D_HIP(ctx, hipMemcpyAsync(dmasks, masks, mbytes, hipMemcpyHostToDevice, s));
    }
    // NumPy callers hand over clouds in arbitrary order: cell-sort large ones (results are order-independent)
    const unsigned flags = n >= 65536 ? F3D_FUSE_SORT : 0u;
    if ((rc = f3d_project_vote_argmax_dev(ctx, dxyz, dtype, n, (const f3d_view*)dviews, nviews, (const uint8_t*)dmasks, h, w,
                                          nclasses, filter, nfilter, threshold, (int64_t*)dcls, (uint16_t*)dvotes, flags,
                                          nullptr, s)))
        return rc;
    if ((rc = take_error(ctx, s, F3D_DEVERR_FUSE))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(classes, dcls, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    if (votes_u16) F3D_HIP(ctx, hipMemcpyAsync(votes_u16, dvotes, (size_t)n * ncols * 2, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// a7 vote, a8 segment
// ---------------------------------------------------------------------------------------------
static int ensure_table(f3d_ctx* ctx, int64_t hw) {
    size_t want = 1024;
    while (want < (size_t)hw * 2) want <<= 1;
    if (ctx->table_slots < want) {
        if (ctx->strict) return fail(ctx, F3D_ERR_NOMEM, "strict context: the vote table holds %zu slots, %zu needed (f3d_ctx_reserve first)", ctx->table_slots, want);
        if (ctx->table) { F3D_HIP(ctx, hipFree(ctx->table)); ctx->table = nullptr; ctx->table_slots = 0; }
        F3D_HIP(ctx, hipMalloc((void**)&ctx->table, want * sizeof(unsigned long long)));
        ctx->table_slots = want;
        ctx->table_stamped = false;
        ++ctx->allocs;
    }
    return F3D_OK;
}

int f3d_vote_uv2pt_dev(f3d_ctx* ctx, const int32_t* uv2pt, const uint8_t* mask, int64_t hw, double* votes, int64_t npts,
                       int ncols, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (hw < 0 || npts < 0 || ncols <= 0 || (hw > 0 && (!uv2pt || !mask || !votes)))
        return fail(ctx, F3D_ERR_INVALID, "vote_uv2pt: bad arguments");
    if (hw == 0) return F3D_OK;
    if ((rc = ensure_table(ctx, hw))) return rc;              // grows only when a larger frame arrives
    size_t slots = 1024;
    while (slots < (size_t)hw * 2) slots <<= 1;
    F3D_HIP(ctx, f3d_launch_vote_uv2pt(uv2pt, mask, hw, votes, npts, ncols, ctx->table, slots, ctx->dev_err, pick(ctx, stream)));
    ctx->table_stamped = false;                               // raw keys in the table: a batched call clears it first
    return F3D_OK;
}

int f3d_vote_uv2pt_batch_dev(f3d_ctx* ctx, const int32_t* luts, const uint8_t* masks, int64_t nframes, int h, int w, double* votes,
                             int64_t npts, int ncols, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    const int64_t hw = (int64_t)h * w;
    if (nframes < 0 || h < 0 || w < 0 || npts < 0 || npts >= ((int64_t)1 << 31) || ncols <= 0 || ncols > 256 ||
        (nframes > 0 && hw > 0 && (!luts || !masks || !votes)))
        return fail(ctx, F3D_ERR_INVALID, "vote_uv2pt_batch: bad arguments (npts < 2^31, ncols <= 256)");
    if (nframes == 0 || hw == 0) return F3D_OK;
    hipStream_t s = pick(ctx, stream);
    // frames per launch: at most 1023 (10 bits of the key) and at most 2^25 lookups (a 512 MiB set at load <= 1/2)
    int64_t per = ((int64_t)1 << 25) / hw;
    if (per < 1) per = 1;
    if (per > 1023) per = 1023;
    const int64_t first = nframes < per ? nframes : per;
    if ((rc = ensure_table(ctx, first * hw))) return rc;
    F3D_HIP(ctx, hipMemsetAsync(ctx->first_bad, 0x7f, sizeof(int), s));                        // 0x7f7f7f7f: no bad frame
    for (int64_t f0 = 0; f0 < nframes; f0 += per) {
        const int nf = (int)(nframes - f0 < per ? nframes - f0 : per);
        if (!ctx->table_stamped || ctx->vote_gen >= 16382u) {
            F3D_HIP(ctx, hipMemsetAsync(ctx->table, 0xFF, ctx->table_slots * sizeof(unsigned long long), s));   // generation 0x3FFF = never current
            ctx->table_stamped = true; ctx->vote_gen = 0;
        }
        const unsigned gen = ctx->vote_gen++;
        size_t slots = 1024;
        while (slots < (size_t)nf * hw * 2) slots <<= 1;      // the share of the table this launch hashes into (<= table_slots)
        F3D_HIP(ctx, f3d_launch_vote_uv2pt_batch(luts + f0 * hw, masks + f0 * hw, nf, h, w, votes, npts, ncols, ctx->table, slots, gen, (int)f0,
                                                 ctx->first_bad, ctx->dev_err, s));
    }
    return F3D_OK;
}

int f3d_vote_uv2pt_batch(f3d_ctx* ctx, const int32_t* luts, const uint8_t* masks, int64_t nframes, int h, int w, double* votes,
                         int64_t npts, int ncols) {
    int rc = enter(ctx); if (rc) return rc;
    const int64_t hw = (int64_t)h * w;
    if (nframes < 0 || h < 0 || w < 0 || npts < 0 || ncols <= 0 || (nframes > 0 && hw > 0 && (!luts || !masks || !votes)))
        return fail(ctx, F3D_ERR_INVALID, "vote_uv2pt_batch: bad arguments");
    if (nframes == 0 || hw == 0) return F3D_OK;
    const size_t vbytes = (size_t)npts * ncols * 8, lb = (size_t)nframes * hw * 4, mb = (size_t)nframes * hw;
    void *dlut, *dmask, *dvotes;
    if ((rc = ensure(ctx, SLOT_AUX0, lb, &dlut)) || (rc = ensure(ctx, SLOT_AUX1, mb, &dmask)) || (rc = ensure(ctx, SLOT_OUT1, vbytes, &dvotes))) return rc;
    hipStream_t s = ctx->stream;
    F3D_HIP(ctx, hipMemcpyAsync(dlut, luts, lb, hipMemcpyHostToDevice, s));                    // every frame's lookup in ONE copy
    F3D_HIP(ctx, hipMemcpyAsync(dmask, masks, mb, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dvotes, votes, vbytes, hipMemcpyHostToDevice, s));             // the matrix travels once per BATCH, not per frame
    if ((rc = f3d_vote_uv2pt_batch_dev(ctx, (const int32_t*)dlut, (const uint8_t*)dmask, nframes, h, w, (double*)dvotes, npts, ncols, s))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(votes, dvotes, vbytes, hipMemcpyDeviceToHost, s));             // frames before a bad one stay applied, like NumPy
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return take_error(ctx, s, F3D_DEVERR_VOTE);
}

int f3d_vote_uv2pt(f3d_ctx* ctx, const int32_t* uv2pt, const uint8_t* mask, int64_t hw, double* votes, int64_t npts, int ncols) {
    int rc = enter(ctx); if (rc) return rc;
    if (hw < 0 || npts < 0 || ncols <= 0 || (hw > 0 && (!uv2pt || !mask || !votes)))
        return fail(ctx, F3D_ERR_INVALID, "vote_uv2pt: bad arguments");
    if (hw == 0) return F3D_OK;
    const size_t vbytes = (size_t)npts * ncols * 8;
    void *dlut, *dmask, *dvotes;
    if ((rc = ensure(ctx, SLOT_AUX0, (size_t)hw * 4, &dlut))) return rc;
    if ((rc = ensure(ctx, SLOT_AUX1, (size_t)hw, &dmask))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT1, vbytes, &dvotes))) return rc;
    hipStream_t s = ctx->stream;
    F3D_HIP(ctx, hipMemcpyAsync(dlut, uv2pt, (size_t)hw * 4, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dmask, mask, (size_t)hw, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dvotes, votes, vbytes, hipMemcpyHostToDevice, s));
    if ((rc = f3d_vote_uv2pt_dev(ctx, (const int32_t*)dlut, (const uint8_t*)dmask, hw, (double*)dvotes, npts, ncols, s))) return rc;
    if ((rc = take_error(ctx, s, F3D_DEVERR_VOTE))) return rc;   // nothing was written in that case
    F3D_HIP(ctx, hipMemcpyAsync(votes, dvotes, vbytes, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

int f3d_segment_votes_dev(f3d_ctx* ctx, const double* votes, int64_t npts, int ncols, int nclasses, double threshold,
                          const int32_t* filter, int nfilter, int64_t* classes, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (npts < 0 || ncols <= 0 || (npts > 0 && (!votes || !classes))) return fail(ctx, F3D_ERR_INVALID, "segment_votes: bad arguments");
    hipStream_t s = pick(ctx, stream);
    f3d_filter_args fa;
    if ((rc = make_filter(ctx, filter, nfilter, ncols, true, s, &fa))) return rc;
    F3D_HIP(ctx, f3d_launch_segment_votes(votes, npts, ncols, nclasses, threshold, fa, classes, s));
    return F3D_OK;
}

int f3d_segment_votes(f3d_ctx* ctx, const double* votes, int64_t npts, int ncols, int nclasses, double threshold,
                      const int32_t* filter, int nfilter, int64_t* classes) {
    int rc = enter(ctx); if (rc) return rc;
    if (npts < 0 || ncols <= 0 || (npts > 0 && (!votes || !classes))) return fail(ctx, F3D_ERR_INVALID, "segment_votes: bad arguments");
    if (npts == 0) return F3D_OK;
    const size_t vbytes = (size_t)npts * ncols * 8;
    void *dvotes, *dcls;
    if ((rc = ensure(ctx, SLOT_OUT1, vbytes, &dvotes))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT0, (size_t)npts * 8, &dcls))) return rc;
    hipStream_t s = ctx->stream;
    F3D_HIP(ctx, hipMemcpyAsync(dvotes, votes, vbytes, hipMemcpyHostToDevice, s));
    if ((rc = f3d_segment_votes_dev(ctx, (const double*)dvotes, npts, ncols, nclasses, threshold, filter, nfilter, (int64_t*)dcls, s)))
        return rc;
    F3D_HIP(ctx, hipMemcpyAsync(classes, dcls, (size_t)npts * 8, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// a9 mask post-processing
// ---------------------------------------------------------------------------------------------
int f3d_sem_logits_to_mask_dev(f3d_ctx* ctx, const float* sem, int c, int64_t hw, float conf, int low_label, uint8_t* mask,
                               void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (c <= 0 || c > 256 || hw < 0 || low_label < 0 || low_label > 255 || (hw > 0 && (!sem || !mask)))
        return fail(ctx, F3D_ERR_INVALID, "sem_logits_to_mask: bad arguments (1 <= c <= 256)");
    F3D_HIP(ctx, f3d_launch_sem_to_mask(sem, 1, c, hw, conf, low_label, mask, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_sem_logits_to_masks_dev(f3d_ctx* ctx, const float* sem, int nimg, int c, int64_t hw, float conf, int low_label, uint8_t* masks,
                                void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (nimg < 0 || nimg > 65535 || c <= 0 || c > 256 || hw < 0 || low_label < 0 || low_label > 255 || (nimg > 0 && hw > 0 && (!sem || !masks)))
        return fail(ctx, F3D_ERR_INVALID, "sem_logits_to_masks: bad arguments (1 <= c <= 256, at most 65535 images per call)");
    F3D_HIP(ctx, f3d_launch_sem_to_mask(sem, nimg, c, hw, conf, low_label, masks, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_sem_logits_to_mask(f3d_ctx* ctx, const float* sem, int c, int64_t hw, float conf, int low_label, uint8_t* mask) {
    int rc = enter(ctx); if (rc) return rc;
    if (c <= 0 || hw < 0 || (hw > 0 && (!sem || !mask))) return fail(ctx, F3D_ERR_INVALID, "sem_logits_to_mask: bad arguments");
    if (hw == 0) return F3D_OK;
    void *dsem, *dmask;
    if ((rc = ensure(ctx, SLOT_MASKS, (size_t)c * hw * 4, &dsem))) return rc;
    if ((rc = ensure(ctx, SLOT_AUX1, (size_t)hw, &dmask))) return rc;
    hipStream_t s = ctx->stream;
    F3D_HIP(ctx, hipMemcpyAsync(dsem, sem, (size_t)c * hw * 4, hipMemcpyHostToDevice, s));
    if ((rc = f3d_sem_logits_to_mask_dev(ctx, (const float*)dsem, c, hw, conf, low_label, (uint8_t*)dmask, s))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(mask, dmask, (size_t)hw, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// a10/a11 oriented boxes
// ---------------------------------------------------------------------------------------------
int f3d_points_in_obb_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_obb* boxes, int b,
                          uint32_t* inside_bits, uint8_t* cooc, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || b < 0 || b > F3D_OBB_MAX_BOXES || (b > 0 && !boxes) || (n > 0 && !xyz) || (!inside_bits && !cooc))
        return fail(ctx, F3D_ERR_INVALID, "points_in_obb: bad arguments (at most %d boxes per call)", F3D_OBB_MAX_BOXES);
    if (b == 0) return F3D_OK;
    hipStream_t s = pick(ctx, stream);
    void* dboxes;                                              // the cell table (8-byte aligned), the boxes, then their float32 bounds
    const size_t cells = f3d_obb_cells_bytes();
    if ((rc = ensure(ctx, SLOT_VIEWS, cells + (sizeof(f3d_obb) + 6 * sizeof(float)) * (size_t)b, &dboxes))) return rc;
    char* base = (char*)dboxes + cells;
    F3D_HIP(ctx, hipMemcpyAsync(base, boxes, sizeof(f3d_obb) * (size_t)b, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, f3d_launch_points_in_obb(xyz, dtype, n, (const f3d_obb*)base, b, (float*)(base + sizeof(f3d_obb) * (size_t)b), dboxes, inside_bits, cooc, s));
    return F3D_OK;
}

int f3d_points_in_obb(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_obb* boxes, int b,
                      uint32_t* inside_bits, uint8_t* cooc) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || b < 0 || (n > 0 && !xyz) || (!inside_bits && !cooc)) return fail(ctx, F3D_ERR_INVALID, "points_in_obb: bad arguments");
    if (b == 0) return F3D_OK;
    const size_t words = (size_t)(b + 31) / 32;
    void *dxyz, *dbits = nullptr, *dcooc = nullptr;
    if ((rc = ensure(ctx, SLOT_XYZ, xyz_bytes(dtype, n), &dxyz))) return rc;
    if (inside_bits && (rc = ensure(ctx, SLOT_OUT1, (size_t)n * words * 4, &dbits))) return rc;
    if (cooc && (rc = ensure(ctx, SLOT_AUX0, (size_t)b * b, &dcooc))) return rc;
    hipStream_t s = ctx->stream;
    if (n) F3D_HIP(ctx, hipMemcpyAsync(dxyz, xyz, xyz_bytes(dtype, n), hipMemcpyHostToDevice, s));
    if (cooc && n == 0) F3D_HIP(ctx, hipMemsetAsync(dcooc, 0, (size_t)b * b, s));
    if ((rc = f3d_points_in_obb_dev(ctx, dxyz, dtype, n, boxes, b, (uint32_t*)dbits, (uint8_t*)dcooc, s))) return rc;
    if (inside_bits && n) F3D_HIP(ctx, hipMemcpyAsync(inside_bits, dbits, (size_t)n * words * 4, hipMemcpyDeviceToHost, s));
    if (cooc) F3D_HIP(ctx, hipMemcpyAsync(cooc, dcooc, (size_t)b * b, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

int f3d_relabel_dev(f3d_ctx* ctx, int64_t* ids, int64_t n, int64_t from, int64_t to, int64_t* count_dev, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || (n > 0 && !ids)) return fail(ctx, F3D_ERR_INVALID, "relabel: bad arguments");
    hipStream_t s = pick(ctx, stream);
    if (count_dev) F3D_HIP(ctx, hipMemsetAsync(count_dev, 0, 8, s));
    F3D_HIP(ctx, f3d_launch_relabel(ids, n, from, to, (unsigned long long*)count_dev, s));
    return F3D_OK;
}

int f3d_relabel(f3d_ctx* ctx, int64_t* ids, int64_t n, int64_t from, int64_t to, int64_t* count) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || (n > 0 && !ids)) return fail(ctx, F3D_ERR_INVALID, "relabel: bad arguments");
    if (count) *count = 0;
    if (n == 0) return F3D_OK;
    void* dids;
    if ((rc = ensure(ctx, SLOT_OUT0, (size_t)n * 8, &dids))) return rc;
    hipStream_t s = ctx->stream;
    F3D_HIP(ctx, hipMemcpyAsync(dids, ids, (size_t)n * 8, hipMemcpyHostToDevice, s));
    if ((rc = f3d_relabel_dev(ctx, (int64_t*)dids, n, from, to, (int64_t*)ctx->count_dev, s))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(ids, dids, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    if (count) F3D_HIP(ctx, hipMemcpyAsync(count, ctx->count_dev, 8, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// a12 remaining intersections.py primitives (host pointers)
// ---------------------------------------------------------------------------------------------
namespace {
struct dbuf { void* p; };
int up(f3d_ctx* ctx, int slot, const void* src, size_t bytes, void** dst) {
    int rc = ensure(ctx, slot, bytes, dst); if (rc) return rc;
    if (bytes) F3D_HIP(ctx, hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return F3D_OK;
}
int down(f3d_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes) F3D_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return F3D_OK;
}
}  // namespace

int f3d_ray_x_lines(f3d_ctx* ctx, const double origin[3], const double direction[3], const double* starts, const double* ends, int64_t n,
                    double* points, uint8_t* within) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || !origin || !direction || (n > 0 && (!starts || !ends || !points || !within))) return fail(ctx, F3D_ERR_INVALID, "ray_x_lines: bad arguments");
    if (n == 0) return F3D_OK;
    void *ds, *de, *dp, *dw;
    if ((rc = up(ctx, SLOT_XYZ, starts, (size_t)n * 24, &ds)) || (rc = up(ctx, SLOT_OUT1, ends, (size_t)n * 24, &de))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT0, (size_t)n * 24, &dp)) || (rc = ensure(ctx, SLOT_AUX1, (size_t)n, &dw))) return rc;
    F3D_HIP(ctx, f3d_launch_ray_x_lines(origin, direction, (const double*)ds, (const double*)de, n, (double*)dp, (uint8_t*)dw, ctx->stream));
    if ((rc = down(ctx, points, dp, (size_t)n * 24)) || (rc = down(ctx, within, dw, (size_t)n))) return rc;
    F3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F3D_OK;
}

int f3d_rays_x_plane(f3d_ctx* ctx, const double pp[3], const double pn[3], const double* origins, const double* dirs, int64_t n, double* points,
                     uint8_t* valid) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || !pp || !pn || (n > 0 && (!origins || !dirs || !points || !valid))) return fail(ctx, F3D_ERR_INVALID, "rays_x_plane: bad arguments");
    if (n == 0) return F3D_OK;
    void *d_o, *dd, *dp, *dv;
    if ((rc = up(ctx, SLOT_XYZ, origins, (size_t)n * 24, &d_o)) || (rc = up(ctx, SLOT_OUT1, dirs, (size_t)n * 24, &dd))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT0, (size_t)n * 24, &dp)) || (rc = ensure(ctx, SLOT_AUX1, (size_t)n, &dv))) return rc;
    F3D_HIP(ctx, f3d_launch_rays_x_plane(pp, pn, (const double*)d_o, (const double*)dd, n, (double*)dp, (uint8_t*)dv, ctx->stream));
    if ((rc = down(ctx, points, dp, (size_t)n * 24)) || (rc = down(ctx, valid, dv, (size_t)n))) return rc;
    F3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F3D_OK;
}

int f3d_lines_x_planes(f3d_ctx* ctx, const double* lo, const double* le, int64_t n, const double* pps, const double* pns, int m, double* points,
                       uint8_t* valid) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || m < 0 || (n > 0 && (!lo || !le)) || (m > 0 && (!pps || !pns)) || (n > 0 && m > 0 && (!points || !valid)))
        return fail(ctx, F3D_ERR_INVALID, "lines_x_planes: bad arguments");
    if (n != 1 && n != m) return fail(ctx, F3D_ERR_INVALID, "operands could not be broadcast together with shapes (%lld,%d,3) (%lld,3)", (long long)n, m, (long long)n);
    if (n == 0 || m == 0) return F3D_OK;
    void *d_o, *de, *dpp, *dpn, *dp, *dv;
    if ((rc = up(ctx, SLOT_XYZ, lo, (size_t)n * 24, &d_o)) || (rc = up(ctx, SLOT_OUT1, le, (size_t)n * 24, &de))) return rc;
    if ((rc = up(ctx, SLOT_VIEWS, pps, (size_t)m * 24, &dpp)) || (rc = up(ctx, SLOT_AUX0, pns, (size_t)m * 24, &dpn))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT0, (size_t)n * m * 24, &dp)) || (rc = ensure(ctx, SLOT_AUX1, (size_t)n * m, &dv))) return rc;
    F3D_HIP(ctx, f3d_launch_lines_x_planes((const double*)d_o, (const double*)de, n, (const double*)dpp, (const double*)dpn, m, n == 1 ? 0 : 1,
                                           (double*)dp, (uint8_t*)dv, ctx->stream));
    if ((rc = down(ctx, points, dp, (size_t)n * m * 24)) || (rc = down(ctx, valid, dv, (size_t)n * m))) return rc;
    F3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F3D_OK;
}

int f3d_point_inside_polygon(f3d_ctx* ctx, const double* points, int64_t n, const double* verts, int m, uint8_t* inside, uint8_t* within) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || m < 1 || !verts || (n > 0 && (!points || !inside || !within))) return fail(ctx, F3D_ERR_INVALID, "point_inside_polygon: bad arguments");
    if (n == 0) return F3D_OK;
    void *dp, *dv, *di, *dw;
    if ((rc = up(ctx, SLOT_XYZ, points, (size_t)n * 24, &dp)) || (rc = up(ctx, SLOT_VIEWS, verts, (size_t)m * 24, &dv))) return rc;
    if ((rc = ensure(ctx, SLOT_AUX1, (size_t)n, &di)) || (rc = ensure(ctx, SLOT_OUT0, (size_t)n * m, &dw))) return rc;
    F3D_HIP(ctx, f3d_launch_point_inside_polygon((const double*)dp, n, (const double*)dv, m, (uint8_t*)di, (uint8_t*)dw, ctx->stream));
    if ((rc = down(ctx, inside, di, (size_t)n)) || (rc = down(ctx, within, dw, (size_t)n * m))) return rc;
    F3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F3D_OK;
}

int f3d_points_plane_projection(f3d_ctx* ctx, const double* points, int64_t n, const double pp[3], const double nr[3], double* out) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || !pp || !nr || (n > 0 && (!points || !out))) return fail(ctx, F3D_ERR_INVALID, "points_plane_projection: bad arguments");
    if (n == 0) return F3D_OK;
    void *dp, *d_o;
    if ((rc = up(ctx, SLOT_XYZ, points, (size_t)n * 24, &dp)) || (rc = ensure(ctx, SLOT_OUT0, (size_t)n * 24, &d_o))) return rc;
    F3D_HIP(ctx, f3d_launch_points_plane_projection((const double*)dp, n, pp, nr, (double*)d_o, ctx->stream));
    if ((rc = down(ctx, out, d_o, (size_t)n * 24))) return rc;
    F3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F3D_OK;
}

int f3d_lines_plane_projection(f3d_ctx* ctx, const double* starts, const double* ends, int64_t n, const double pp[3], const double nr[3],
                               double* sp, double* ep, double* dirs) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || !pp || !nr || (n > 0 && (!starts || !ends || !sp || !ep || !dirs))) return fail(ctx, F3D_ERR_INVALID, "lines_plane_projection: bad arguments");
    if (n == 0) return F3D_OK;
    void *ds, *de, *dsp, *dep, *dd;
    if ((rc = up(ctx, SLOT_XYZ, starts, (size_t)n * 24, &ds)) || (rc = up(ctx, SLOT_OUT1, ends, (size_t)n * 24, &de))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT0, (size_t)n * 24, &dsp)) || (rc = ensure(ctx, SLOT_MASKS, (size_t)n * 24, &dep)) ||
        (rc = ensure(ctx, SLOT_AUX0, (size_t)n * 24, &dd))) return rc;
    F3D_HIP(ctx, f3d_launch_points_plane_projection((const double*)ds, n, pp, nr, (double*)dsp, ctx->stream));
    F3D_HIP(ctx, f3d_launch_points_plane_projection((const double*)de, n, pp, nr, (double*)dep, ctx->stream));
    F3D_HIP(ctx, f3d_launch_unit_difference((const double*)dsp, (const double*)dep, n, (double*)dd, ctx->stream));
    if ((rc = down(ctx, sp, dsp, (size_t)n * 24)) || (rc = down(ctx, ep, dep, (size_t)n * 24)) || (rc = down(ctx, dirs, dd, (size_t)n * 24))) return rc;
    F3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// (f)#1 same-class connected components
// ---------------------------------------------------------------------------------------------
int f3d_components_same_class_dev(f3d_ctx* ctx, const int64_t* classes, int64_t n, const int64_t* offsets, const int32_t* nbrs,
                                  int32_t* parent, int64_t* root, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || n > 0x7fffffffLL || (n > 0 && (!classes || !offsets || !parent || !root)))
        return fail(ctx, F3D_ERR_INVALID, "components_same_class: bad arguments (n < 2^31)");
    F3D_HIP(ctx, f3d_launch_components(classes, n, offsets, nbrs, parent, root, ctx->dev_err, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_components_same_class(f3d_ctx* ctx, const int64_t* classes, int64_t n, const int64_t* offsets, const int32_t* nbrs,
                              int64_t* root) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || (n > 0 && (!classes || !offsets || !root))) return fail(ctx, F3D_ERR_INVALID, "components_same_class: bad arguments");
    if (n == 0) return F3D_OK;
    const int64_t e = offsets[n];
    if (e < 0 || (e > 0 && !nbrs)) return fail(ctx, F3D_ERR_INVALID, "components_same_class: bad adjacency");
    void *dcls, *doffs, *dnb, *dpar, *droot;
    if ((rc = ensure(ctx, SLOT_XYZ, (size_t)n * 8, &dcls))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT1, (size_t)(n + 1) * 8, &doffs))) return rc;
    if ((rc = ensure(ctx, SLOT_MASKS, (size_t)e * 4, &dnb))) return rc;
    if ((rc = ensure(ctx, SLOT_AUX0, (size_t)n * 4, &dpar))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT0, (size_t)n * 8, &droot))) return rc;
    hipStream_t s = ctx->stream;
    F3D_HIP(ctx, hipMemcpyAsync(dcls, classes, (size_t)n * 8, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(doffs, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, s));
    if (e) F3D_HIP(ctx, hipMemcpyAsync(dnb, nbrs, (size_t)e * 4, hipMemcpyHostToDevice, s));
    if ((rc = f3d_components_same_class_dev(ctx, (const int64_t*)dcls, n, (const int64_t*)doffs, (const int32_t*)dnb, (int32_t*)dpar,
                                            (int64_t*)droot, s))) return rc;
    if ((rc = take_error(ctx, s, F3D_DEVERR_CC))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(root, droot, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// a5 patch matching of Fusion.fuse
// ---------------------------------------------------------------------------------------------
int f3d_patch_owner_dev(f3d_ctx* ctx, const int32_t* uv, int64_t m, int h, int w, int half, double radius, double min_cosine,
                        const double* seed_pts, const double* seed_nrm, const double* q_pts, const double* q_nrm,
                        const uint8_t* free_px, int32_t* owner, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    const int64_t npx = (int64_t)h * w;
    if (h < 0 || w < 0 || m < 0 || half < 0 || npx > 0x7fffffffLL || m > 0x7fffffffLL || (m > 0 && (!uv || !seed_pts || !seed_nrm)) ||
        (npx > 0 && (!q_pts || !q_nrm || !free_px || !owner)))
        return fail(ctx, F3D_ERR_INVALID, "patch_owner: bad arguments");
    if (npx == 0) return F3D_OK;
    void* scratch;
    if ((rc = ensure(ctx, SLOT_PATCH, f3d_patch_scratch_bytes(h, w, m), &scratch))) return rc;
    F3D_HIP(ctx, f3d_launch_patch_owner(uv, m, h, w, half, radius, min_cosine, seed_pts, seed_nrm, q_pts, q_nrm, free_px, owner, scratch,
                                        pick(ctx, stream)));
    return F3D_OK;
}

int f3d_patch_owner(f3d_ctx* ctx, const int32_t* uv, int64_t m, int h, int w, int half, double radius, double min_cosine,
                    const double* seed_pts, const double* seed_nrm, const double* q_pts, const double* q_nrm,
                    const uint8_t* free_px, int32_t* owner) {
    int rc = enter(ctx); if (rc) return rc;
    const int64_t npx = (int64_t)h * w;
    if (h < 0 || w < 0 || m < 0 || (m > 0 && (!uv || !seed_pts || !seed_nrm)) || (npx > 0 && (!q_pts || !q_nrm || !free_px || !owner)))
        return fail(ctx, F3D_ERR_INVALID, "patch_owner: bad arguments");
    if (npx == 0) return F3D_OK;
    void *duv, *dsp, *dsn, *dqp, *dqn, *dfree, *down;
    if ((rc = ensure(ctx, SLOT_AUX0, (size_t)m * 8 + 8, &duv)) || (rc = ensure(ctx, SLOT_XYZ, (size_t)m * 24 + 8, &dsp)) ||
        (rc = ensure(ctx, SLOT_OUT1, (size_t)m * 24 + 8, &dsn)) || (rc = ensure(ctx, SLOT_MASKS, (size_t)npx * 24, &dqp)) ||
        (rc = ensure(ctx, SLOT_VIEWS, (size_t)npx * 24, &dqn)) || (rc = ensure(ctx, SLOT_AUX1, (size_t)npx, &dfree)) ||
        (rc = ensure(ctx, SLOT_OUT0, (size_t)npx * 4, &down)))
        return rc;
    hipStream_t s = ctx->stream;
    if (m) {
        F3D_HIP(ctx, hipMemcpyAsync(duv, uv, (size_t)m * 8, hipMemcpyHostToDevice, s));
        F3D_HIP(ctx, hipMemcpyAsync(dsp, seed_pts, (size_t)m * 24, hipMemcpyHostToDevice, s));
        F3D_HIP(ctx, hipMemcpyAsync(dsn, seed_nrm, (size_t)m * 24, hipMemcpyHostToDevice, s));
    }
    F3D_HIP(ctx, hipMemcpyAsync(dqp, q_pts, (size_t)npx * 24, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dqn, q_nrm, (size_t)npx * 24, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dfree, free_px, (size_t)npx, hipMemcpyHostToDevice, s));
    if ((rc = f3d_patch_owner_dev(ctx, (const int32_t*)duv, m, h, w, half, radius, min_cosine, (const double*)dsp, (const double*)dsn,
                                  (const double*)dqp, (const double*)dqn, (const uint8_t*)dfree, (int32_t*)down, s))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(owner, down, (size_t)npx * 4, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

int f3d_patch_seeds(f3d_ctx* ctx, const double* pts, const double* nrm, const int32_t* prio, const uint8_t* free_px, int h, int w,
                    int half, double radius, double min_cosine, int32_t* owner, int32_t* rounds) {
    int rc = enter(ctx); if (rc) return rc;
    const int64_t npx = (int64_t)h * w;
    if (h < 0 || w < 0 || half < 0 || npx > 0x7fffffffLL || (npx > 0 && (!pts || !nrm || !prio || !free_px || !owner)))
        return fail(ctx, F3D_ERR_INVALID, "patch_seeds: bad arguments");
    if (rounds) *rounds = 0;
    if (npx == 0) return F3D_OK;
    void *dp, *dn, *dprio, *dfree, *dstat, *down;
    if ((rc = ensure(ctx, SLOT_MASKS, (size_t)npx * 24, &dp)) || (rc = ensure(ctx, SLOT_VIEWS, (size_t)npx * 24, &dn)) ||
        (rc = ensure(ctx, SLOT_AUX0, (size_t)npx * 4, &dprio)) || (rc = ensure(ctx, SLOT_AUX1, (size_t)npx, &dfree)) ||
        (rc = ensure(ctx, SLOT_PATCH, (size_t)npx * 4 + 256, &dstat)) || (rc = ensure(ctx, SLOT_OUT0, (size_t)npx * 4, &down)))
        return rc;
    hipStream_t s = ctx->stream;
    F3D_HIP(ctx, hipMemcpyAsync(dp, pts, (size_t)npx * 24, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dn, nrm, (size_t)npx * 24, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dprio, prio, (size_t)npx * 4, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dfree, free_px, (size_t)npx, hipMemcpyHostToDevice, s));
    int r = 0;
    int32_t* counter = (int32_t*)((char*)dstat + (((size_t)npx * 4 + 63) & ~(size_t)63));
    F3D_HIP(ctx, f3d_launch_patch_seeds((const double*)dp, (const double*)dn, (const int32_t*)dprio, (const uint8_t*)dfree, h, w, half, radius,
                                        min_cosine, (int32_t*)dstat, (int32_t*)down, counter, &r, s));
    F3D_HIP(ctx, hipMemcpyAsync(owner, down, (size_t)npx * 4, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    if (rounds) *rounds = r;
    return F3D_OK;
}

// The matching of one frame of Fusion.fuse with the ordered sums of what every seed takes: the frame is uploaded once, the owner and
// the sums kernels run back to back on the resident arrays (colours optional).
int f3d_patch_match(f3d_ctx* ctx, const int32_t* uv, int64_t m, int h, int w, int half, double radius, double min_cosine,
                    const double* seed_pts, const double* seed_nrm, const double* q_pts, const double* q_nrm, const double* q_clr,
                    const uint8_t* free_px, int32_t* owner, double* sums, int32_t* counts) {
    int rc = enter(ctx); if (rc) return rc;
    const int64_t npx = (int64_t)h * w;
    if (h < 0 || w < 0 || m < 0 || half < 0 || npx > 0x7fffffffLL || m > 0x7fffffffLL || (m > 0 && (!uv || !seed_pts || !seed_nrm || !sums || !counts)) ||
        (npx > 0 && (!q_pts || !q_nrm || !free_px || !owner)))
        return fail(ctx, F3D_ERR_INVALID, "patch_match: bad arguments");
    if (npx == 0) return F3D_OK;
    void *duv, *dsp, *dsn, *dqp, *dqn, *dqc, *dfree, *down, *dsum, *scratch;
    if ((rc = ensure(ctx, SLOT_AUX0, (size_t)m * 8 + 8, &duv)) || (rc = ensure(ctx, SLOT_XYZ, (size_t)m * 24 + 8, &dsp)) ||
        (rc = ensure(ctx, SLOT_OUT1, (size_t)m * 24 + 8, &dsn)) || (rc = ensure(ctx, SLOT_MASKS, (size_t)npx * 24, &dqp)) ||
        (rc = ensure(ctx, SLOT_VIEWS, (size_t)npx * 24, &dqn)) || (rc = ensure(ctx, SLOT_TILED_MASKS, (size_t)npx * 24, &dqc)) ||
        (rc = ensure(ctx, SLOT_AUX1, (size_t)npx, &dfree)) || (rc = ensure(ctx, SLOT_OUT0, (size_t)npx * 4, &down)) ||
        (rc = ensure(ctx, SLOT_GRAPH, (size_t)m * 76 + 16, &dsum)) || (rc = ensure(ctx, SLOT_PATCH, f3d_patch_scratch_bytes(h, w, m), &scratch)))
        return rc;
    hipStream_t s = ctx->stream;
    if (m) {
        F3D_HIP(ctx, hipMemcpyAsync(duv, uv, (size_t)m * 8, hipMemcpyHostToDevice, s));
        F3D_HIP(ctx, hipMemcpyAsync(dsp, seed_pts, (size_t)m * 24, hipMemcpyHostToDevice, s));
        F3D_HIP(ctx, hipMemcpyAsync(dsn, seed_nrm, (size_t)m * 24, hipMemcpyHostToDevice, s));
    }
    F3D_HIP(ctx, hipMemcpyAsync(dqp, q_pts, (size_t)npx * 24, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dqn, q_nrm, (size_t)npx * 24, hipMemcpyHostToDevice, s));
    if (q_clr) F3D_HIP(ctx, hipMemcpyAsync(dqc, q_clr, (size_t)npx * 24, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dfree, free_px, (size_t)npx, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, f3d_launch_patch_owner((const int32_t*)duv, m, h, w, half, radius, min_cosine, (const double*)dsp, (const double*)dsn, (const double*)dqp,
                                        (const double*)dqn, (const uint8_t*)dfree, (int32_t*)down, scratch, s));
    double* dsums = (double*)dsum;
    int32_t* dcnt = (int32_t*)((char*)dsum + (((size_t)m * 72 + 15) & ~(size_t)15));
    F3D_HIP(ctx, f3d_launch_patch_sums((const int32_t*)down, (const int32_t*)duv, m, h, w, half, (const double*)dqp, (const double*)dqn,
                                       q_clr ? (const double*)dqc : nullptr, dsums, dcnt, s));
    F3D_HIP(ctx, hipMemcpyAsync(owner, down, (size_t)npx * 4, hipMemcpyDeviceToHost, s));
    if (m) {
        F3D_HIP(ctx, hipMemcpyAsync(sums, dsums, (size_t)m * 72, hipMemcpyDeviceToHost, s));
        F3D_HIP(ctx, hipMemcpyAsync(counts, dcnt, (size_t)m * 4, hipMemcpyDeviceToHost, s));
    }
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

// patch_downsample with the ordered sums of every seed's members (sums [h*w, 9], counts [h*w]; only the self-owning pixels carry values)
int f3d_patch_seeds_sums(f3d_ctx* ctx, const double* pts, const double* nrm, const double* clr, const int32_t* prio, const uint8_t* free_px,
                         int h, int w, int half, double radius, double min_cosine, int32_t* owner, double* sums, int32_t* counts,
                         int32_t* rounds) {
    int rc = enter(ctx); if (rc) return rc;
    const int64_t npx = (int64_t)h * w;
    if (h < 0 || w < 0 || half < 0 || npx > 0x7fffffffLL || (npx > 0 && (!pts || !nrm || !prio || !free_px || !owner || !sums || !counts)))
        return fail(ctx, F3D_ERR_INVALID, "patch_seeds_sums: bad arguments");
    if (rounds) *rounds = 0;
    if (npx == 0) return F3D_OK;
    void *dp, *dn, *dc, *dprio, *dfree, *dstat, *down, *dsum;
    if ((rc = ensure(ctx, SLOT_MASKS, (size_t)npx * 24, &dp)) || (rc = ensure(ctx, SLOT_VIEWS, (size_t)npx * 24, &dn)) ||
        (rc = ensure(ctx, SLOT_TILED_MASKS, (size_t)npx * 24, &dc)) || (rc = ensure(ctx, SLOT_AUX0, (size_t)npx * 4, &dprio)) ||
        (rc = ensure(ctx, SLOT_AUX1, (size_t)npx, &dfree)) || (rc = ensure(ctx, SLOT_PATCH, (size_t)npx * 4 + 256, &dstat)) ||
        (rc = ensure(ctx, SLOT_OUT0, (size_t)npx * 4, &down)) || (rc = ensure(ctx, SLOT_GRAPH, (size_t)npx * 76 + 16, &dsum)))
        return rc;
    hipStream_t s = ctx->stream;
    F3D_HIP(ctx, hipMemcpyAsync(dp, pts, (size_t)npx * 24, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dn, nrm, (size_t)npx * 24, hipMemcpyHostToDevice, s));
    if (clr) F3D_HIP(ctx, hipMemcpyAsync(dc, clr, (size_t)npx * 24, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dprio, prio, (size_t)npx * 4, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dfree, free_px, (size_t)npx, hipMemcpyHostToDevice, s));
    int r = 0;
    int32_t* counter = (int32_t*)((char*)dstat + (((size_t)npx * 4 + 63) & ~(size_t)63));
    F3D_HIP(ctx, f3d_launch_patch_seeds((const double*)dp, (const double*)dn, (const int32_t*)dprio, (const uint8_t*)dfree, h, w, half, radius,
                                        min_cosine, (int32_t*)dstat, (int32_t*)down, counter, &r, s));
    double* dsums = (double*)dsum;
    int32_t* dcnt = (int32_t*)((char*)dsum + (((size_t)npx * 72 + 15) & ~(size_t)15));
    F3D_HIP(ctx, f3d_launch_patch_sums((const int32_t*)down, nullptr, npx, h, w, half, (const double*)dp, (const double*)dn, clr ? (const double*)dc : nullptr,
                                       dsums, dcnt, s));
    F3D_HIP(ctx, hipMemcpyAsync(owner, down, (size_t)npx * 4, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipMemcpyAsync(sums, dsums, (size_t)npx * 72, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipMemcpyAsync(counts, dcnt, (size_t)npx * 4, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    if (rounds) *rounds = r;
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// (f)#1 adjacency: KDTree(points).query_radius(points, r) (fusion.py:374-375) as CSR
// ---------------------------------------------------------------------------------------------
int f3d_radius_graph_count_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, double radius, int64_t* offsets,
                               int64_t* nnz, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || n > 0x7fffffffLL || !nnz || (n > 0 && (!xyz || !offsets)) || !(radius >= 0.0) || !(radius < 1e300))
        return fail(ctx, F3D_ERR_INVALID, "radius_graph: bad arguments (n < 2^31, finite radius >= 0)");
    *nnz = 0; ctx->graph_n = -1;
    if (n == 0) return F3D_OK;
    hipStream_t s = pick(ctx, stream);
    void* dbox;
    if ((rc = ensure(ctx, SLOT_GRAPH_BBOX, f3d_graph_bbox_bytes(), &dbox))) return rc;
    int nb = 0;
    F3D_HIP(ctx, f3d_launch_graph_bbox(xyz, dtype, n, dbox, &nb, s));
    std::vector<char> hbox(f3d_graph_bbox_bytes());
    F3D_HIP(ctx, hipMemcpyAsync(hbox.data(), dbox, hbox.size(), hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    double lo[3], hi[3];
    if (f3d_graph_reduce_bbox(hbox.data(), nb, lo, hi))
        return fail(ctx, F3D_ERR_INVALID, "radius_graph: the cloud contains NaN or infinity (sklearn's KDTree raises ValueError)");
    // cell edge: a hair above the radius (two points within r are then provably in adjacent cells whatever the rounding of
    // the cell index), grown until every axis has <= 1024 cells and the table <= 2^24 cells
    f3d_graphgrid g;
    double cell = radius * 1.000001 + 1e-300;
    double ext[3];
    for (int c = 0; c < 3; ++c) { ext[c] = hi[c] - lo[c]; if (!(ext[c] < 1e300)) return fail(ctx, F3D_ERR_INVALID, "radius_graph: extent overflow"); }
    for (;;) {
        double cells = 1.0; bool ok = true;
        for (int c = 0; c < 3; ++c) { const double d = floor(ext[c] / cell) + 1.0; if (!(d <= 1024.0)) ok = false; cells *= d; }
        if (ok && cells <= 16777216.0) break;
        cell *= 1.25;
    }
    for (int c = 0; c < 3; ++c) { g.lo[c] = lo[c]; g.dim[c] = (int)(floor(ext[c] / cell) + 1.0); }
    g.inv_cell = 1.0 / cell; g.pad = 0;
    const int64_t ncells = (int64_t)g.dim[0] * g.dim[1] * g.dim[2];
    void* scratch;
    if ((rc = ensure(ctx, SLOT_GRAPH, f3d_graph_scratch_bytes(n, ncells), &scratch))) return rc;
    const double r2 = radius * radius;                                     // sklearn: reduced radius = r ** 2
    F3D_HIP(ctx, f3d_launch_graph_count(xyz, dtype, n, g, r2, scratch, offsets, s));
    F3D_HIP(ctx, hipMemcpyAsync(nnz, offsets + n, 8, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    ctx->graph_grid = g; ctx->graph_n = n; ctx->graph_r2 = r2; ctx->graph_xyz = xyz;
    return F3D_OK;
}

int f3d_radius_graph_fill_dev(f3d_ctx* ctx, int64_t n, const int64_t* offsets, int32_t* nbrs, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n != ctx->graph_n || n < 0) return fail(ctx, F3D_ERR_INVALID, "radius_graph_fill: call f3d_radius_graph_count for this cloud first");
    if (n == 0) return F3D_OK;
    if (!offsets || !nbrs) return fail(ctx, F3D_ERR_INVALID, "radius_graph_fill: bad arguments");
    F3D_HIP(ctx, f3d_launch_graph_fill(n, ctx->graph_grid, ctx->graph_r2, ctx->slot[SLOT_GRAPH], offsets, nbrs, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_radius_graph_count(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, double radius, int64_t* offsets, int64_t* nnz) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || !nnz || (n > 0 && (!xyz || !offsets))) return fail(ctx, F3D_ERR_INVALID, "radius_graph: bad arguments");
    *nnz = 0;
    if (n == 0) { ctx->graph_n = 0; return F3D_OK; }
    void *dxyz, *doffs;
    if ((rc = ensure(ctx, SLOT_XYZ, xyz_bytes(dtype, n), &dxyz))) return rc;
    if ((rc = ensure(ctx, SLOT_OUT1, (size_t)(n + 1) * 8, &doffs))) return rc;
    hipStream_t s = ctx->stream;
    F3D_HIP(ctx, hipMemcpyAsync(dxyz, xyz, xyz_bytes(dtype, n), hipMemcpyHostToDevice, s));
    if ((rc = f3d_radius_graph_count_dev(ctx, dxyz, dtype, n, radius, (int64_t*)doffs, nnz, s))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(offsets, doffs, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

int f3d_radius_graph_fill(f3d_ctx* ctx, int64_t n, int32_t* nbrs) {
    int rc = enter(ctx); if (rc) return rc;
    if (n != ctx->graph_n || n < 0) return fail(ctx, F3D_ERR_INVALID, "radius_graph_fill: call f3d_radius_graph_count for this cloud first");
    if (n == 0) return F3D_OK;
    hipStream_t s = ctx->stream;
    int64_t nnz = 0;
    const int64_t* doffs = (const int64_t*)ctx->slot[SLOT_OUT1];           // left there by f3d_radius_graph_count
    F3D_HIP(ctx, hipMemcpyAsync(&nnz, doffs + n, 8, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    if (nnz == 0) return F3D_OK;
    if (!nbrs) return fail(ctx, F3D_ERR_INVALID, "radius_graph_fill: nbrs is NULL");
    void* dnb;
    if ((rc = ensure(ctx, SLOT_MASKS, (size_t)nnz * 4, &dnb))) return rc;
    if ((rc = f3d_radius_graph_fill_dev(ctx, n, doffs, (int32_t*)dnb, s))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(nbrs, dnb, (size_t)nnz * 4, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

// ---------------------------------------------------------------------------------------------
// merge_bb support: grouping by instance id, hull candidates of every instance
// ---------------------------------------------------------------------------------------------
int f3d_group_by_id_dev(f3d_ctx* ctx, const int64_t* ids, int64_t n, int64_t nids, int32_t* order, uint32_t* sorted_ids, int64_t* starts,
                        void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || n > 0x7fffffffLL || nids < 0 || nids >= 0x7fffffffLL || !starts || (n > 0 && (!ids || !order || !sorted_ids)))
        return fail(ctx, F3D_ERR_INVALID, "group_by_id: bad arguments (n, nids < 2^31)");
    void* scratch;
    if ((rc = ensure(ctx, SLOT_GRP_SCRATCH, f3d_group_scratch_bytes(n, nids), &scratch))) return rc;
    F3D_HIP(ctx, f3d_launch_group_by_id(ids, n, nids, order, sorted_ids, starts, scratch, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_obb_extremes_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const int32_t* order, const uint32_t* sorted_ids,
                         int64_t nids, int32_t* extremes, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || nids < 0 || (nids > 0 && !extremes) || (n > 0 && (!xyz || !order || !sorted_ids)))
        return fail(ctx, F3D_ERR_INVALID, "obb_extremes: bad arguments");
    void* table;
    if ((rc = ensure(ctx, SLOT_OBB_TABLE, (size_t)nids * F3D_OBB_NDIR * 8, &table))) return rc;
    F3D_HIP(ctx, f3d_launch_obb_extremes(xyz, dtype, n, order, sorted_ids, nids, (unsigned long long*)table, extremes, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_obb_hull_filter_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const int32_t* order, const uint32_t* sorted_ids,
                            const int64_t* starts, int64_t nids, const int32_t* facet_start, const double* facets, const double* margin,
                            int32_t* cand, int32_t* cand_count, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || nids < 0 || (nids > 0 && (!facet_start || !margin || !cand_count || !starts)) || (n > 0 && (!xyz || !order || !sorted_ids || !cand)))
        return fail(ctx, F3D_ERR_INVALID, "obb_hull_filter: bad arguments");
    F3D_HIP(ctx, f3d_launch_obb_hull_filter(xyz, dtype, n, order, sorted_ids, starts, nids, facet_start, facets, margin, cand, cand_count,
                                            pick(ctx, stream)));
    return F3D_OK;
}

int f3d_obb_candidates_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const int32_t* order, const uint32_t* sorted_ids,
                           const int64_t* starts, int64_t nids, int min_members, int32_t* cand, int64_t* cand_start, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || n > 0x7fffffffLL || nids < 0 || nids >= 0x7fffffffLL || (nids > 0 && (!starts || !cand_start)) ||
        (n > 0 && (!xyz || !order || !sorted_ids || !cand)))
        return fail(ctx, F3D_ERR_INVALID, "obb_candidates: bad arguments");
    void *table, *scratch;
    if ((rc = ensure(ctx, SLOT_OBB_TABLE, (size_t)nids * F3D_OBB_NDIR * 8, &table))) return rc;
    if ((rc = ensure(ctx, SLOT_OBB_FACETS, f3d_obb_candidates_scratch_bytes(n, nids), &scratch))) return rc;
    F3D_HIP(ctx, f3d_launch_obb_candidates(xyz, dtype, n, order, sorted_ids, starts, nids, min_members, (unsigned long long*)table, scratch, cand,
                                           cand_start, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_gather_points_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, const int32_t* idx, int64_t count, double* out, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (count < 0 || (count > 0 && (!xyz || !idx || !out))) return fail(ctx, F3D_ERR_INVALID, "gather_points: bad arguments");
    F3D_HIP(ctx, f3d_launch_gather_points(xyz, dtype, idx, count, out, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_obb_fit_dev(f3d_ctx* ctx, const double* pts, const int64_t* start, int nfit, int64_t total, double* boxes, int32_t* status, uint8_t* isvert,
                    int32_t* nvert, void* stream) {
    int rc = enter(ctx); if (rc) return rc;
    if (nfit < 0 || total < 0 || (nfit > 0 && (!start || !boxes || !status || !isvert))) return fail(ctx, F3D_ERR_INVALID, "obb_fit: bad arguments");
    void* vlist;                                               // the vertices of every instance as a list (grows on first use only)
    if ((rc = ensure(ctx, SLOT_OBB_CAND, (size_t)total * 4, &vlist))) return rc;
    F3D_HIP(ctx, f3d_launch_obb_fit(pts, start, nfit, boxes, status, isvert, (int32_t*)vlist, nvert, pick(ctx, stream)));
    return F3D_OK;
}

int f3d_obb_fit(f3d_ctx* ctx, const double* pts, const int64_t* start, int nfit, double* boxes, int32_t* status, uint8_t* isvert, int32_t* nvert) {
    int rc = enter(ctx); if (rc) return rc;
    if (nfit < 0 || (nfit > 0 && (!start || !boxes || !status))) return fail(ctx, F3D_ERR_INVALID, "obb_fit: bad arguments");
    if (nfit == 0) return F3D_OK;
    const int64_t total = start[nfit];
    if (total < 0 || start[0] != 0 || (total > 0 && !pts)) return fail(ctx, F3D_ERR_INVALID, "obb_fit: start must run from 0 to the number of points");
    for (int k = 0; k < nfit; ++k) if (start[k + 1] < start[k]) return fail(ctx, F3D_ERR_INVALID, "obb_fit: start must be non-decreasing");
    void *dpts, *dstart, *dboxes, *dstatus, *dvert, *dnv;
    if ((rc = ensure(ctx, SLOT_XYZ, (size_t)total * 24, &dpts)) || (rc = ensure(ctx, SLOT_AUX0, (size_t)(nfit + 1) * 8, &dstart)) ||
        (rc = ensure(ctx, SLOT_OUT0, (size_t)nfit * sizeof(f3d_obb), &dboxes)) || (rc = ensure(ctx, SLOT_AUX1, (size_t)nfit * 8, &dstatus)) ||
        (rc = ensure(ctx, SLOT_OUT1, (size_t)total, &dvert)))
        return rc;
    dnv = (char*)dstatus + (size_t)nfit * 4;
    hipStream_t s = ctx->stream;
    if (total) F3D_HIP(ctx, hipMemcpyAsync(dpts, pts, (size_t)total * 24, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dstart, start, (size_t)(nfit + 1) * 8, hipMemcpyHostToDevice, s));
    if ((rc = f3d_obb_fit_dev(ctx, (const double*)dpts, (const int64_t*)dstart, nfit, total, (double*)dboxes, (int32_t*)dstatus, (uint8_t*)dvert, (int32_t*)dnv, s))) return rc;
    F3D_HIP(ctx, hipMemcpyAsync(boxes, dboxes, (size_t)nfit * sizeof(f3d_obb), hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipMemcpyAsync(status, dstatus, (size_t)nfit * 4, hipMemcpyDeviceToHost, s));
    if (isvert && total) F3D_HIP(ctx, hipMemcpyAsync(isvert, dvert, (size_t)total, hipMemcpyDeviceToHost, s));
    if (nvert) F3D_HIP(ctx, hipMemcpyAsync(nvert, dnv, (size_t)nfit * 4, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

// host-pointer sequence; the grouping (and, from the extremes call on, the cloud) stays in the context between the calls
int f3d_group_by_id(f3d_ctx* ctx, const int64_t* ids, int64_t n, int64_t nids, int32_t* order, int64_t* starts) {
    int rc = enter(ctx); if (rc) return rc;
    if (n < 0 || nids < 0 || !starts || (n > 0 && (!ids || !order))) return fail(ctx, F3D_ERR_INVALID, "group_by_id: bad arguments");
    ctx->grp_n = -1;
    void *dids, *dorder, *dkeys, *dstarts;
    if ((rc = ensure(ctx, SLOT_OUT0, (size_t)n * 8, &dids)) || (rc = ensure(ctx, SLOT_GRP_ORDER, (size_t)n * 4, &dorder)) ||
        (rc = ensure(ctx, SLOT_GRP_KEYS, (size_t)n * 4, &dkeys)) || (rc = ensure(ctx, SLOT_GRP_STARTS, (size_t)(nids + 2) * 8, &dstarts)))
        return rc;
    hipStream_t s = ctx->stream;
    if (n) F3D_HIP(ctx, hipMemcpyAsync(dids, ids, (size_t)n * 8, hipMemcpyHostToDevice, s));
    if ((rc = f3d_group_by_id_dev(ctx, (const int64_t*)dids, n, nids, (int32_t*)dorder, (uint32_t*)dkeys, (int64_t*)dstarts, s))) return rc;
    if (n) F3D_HIP(ctx, hipMemcpyAsync(order, dorder, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipMemcpyAsync(starts, dstarts, (size_t)(nids + 2) * 8, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    ctx->grp_n = n; ctx->grp_nids = nids;
    return F3D_OK;
}

int f3d_obb_extremes(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, int32_t* extremes) {
    int rc = enter(ctx); if (rc) return rc;
    if (n != ctx->grp_n || n < 0) return fail(ctx, F3D_ERR_INVALID, "obb_extremes: call f3d_group_by_id for this cloud first");
    const int64_t nids = ctx->grp_nids;
    if ((n > 0 && !xyz) || (nids > 0 && !extremes)) return fail(ctx, F3D_ERR_INVALID, "obb_extremes: bad arguments");
    void *dxyz, *dext;
    if ((rc = ensure(ctx, SLOT_XYZ, xyz_bytes(dtype, n), &dxyz)) || (rc = ensure(ctx, SLOT_OBB_CAND, (size_t)(n > nids * F3D_OBB_NDIR ? n : nids * F3D_OBB_NDIR) * 4 + (size_t)nids * 4, &dext)))
        return rc;
    hipStream_t s = ctx->stream;
    if (n) F3D_HIP(ctx, hipMemcpyAsync(dxyz, xyz, xyz_bytes(dtype, n), hipMemcpyHostToDevice, s));
    if ((rc = f3d_obb_extremes_dev(ctx, dxyz, dtype, n, (const int32_t*)ctx->slot[SLOT_GRP_ORDER], (const uint32_t*)ctx->slot[SLOT_GRP_KEYS], nids,
                                   (int32_t*)dext, s))) return rc;
    if (nids) F3D_HIP(ctx, hipMemcpyAsync(extremes, dext, (size_t)nids * F3D_OBB_NDIR * 4, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    ctx->grp_dtype = dtype;
    return F3D_OK;
}

int f3d_obb_hull_filter(f3d_ctx* ctx, int64_t n, const int32_t* facet_start, const double* facets, const double* margin, int32_t* cand,
                        int32_t* cand_count) {
    int rc = enter(ctx); if (rc) return rc;
    if (n != ctx->grp_n || n < 0) return fail(ctx, F3D_ERR_INVALID, "obb_hull_filter: call f3d_group_by_id and f3d_obb_extremes for this cloud first");
    const int64_t nids = ctx->grp_nids;
    if (nids > 0 && (!facet_start || !margin || !cand_count)) return fail(ctx, F3D_ERR_INVALID, "obb_hull_filter: bad arguments");
    if (nids == 0) return F3D_OK;
    const int64_t nf = facet_start[nids];
    if (nf < 0 || (nf > 0 && !facets) || (n > 0 && !cand)) return fail(ctx, F3D_ERR_INVALID, "obb_hull_filter: bad facet table");
    void *dfac, *dcand;
    const size_t fbytes = (size_t)(nids + 1) * 4, ebytes = (size_t)nf * 32, mbytes = (size_t)nids * 8;
    if ((rc = ensure(ctx, SLOT_OBB_FACETS, ((fbytes + 7) & ~(size_t)7) + ebytes + mbytes + 64, &dfac)) ||
        (rc = ensure(ctx, SLOT_OBB_CAND, (size_t)(n > nids * F3D_OBB_NDIR ? n : nids * F3D_OBB_NDIR) * 4 + (size_t)nids * 4, &dcand)))
        return rc;
    char* base = (char*)dfac;
    int32_t* dfs = (int32_t*)base;
    double* deq = (double*)(base + ((fbytes + 7) & ~(size_t)7));
    double* dmg = deq + 4 * (size_t)nf;
    int32_t* dcnt = (int32_t*)((char*)dcand + (size_t)(n > nids * F3D_OBB_NDIR ? n : nids * F3D_OBB_NDIR) * 4);
    hipStream_t s = ctx->stream;
    F3D_HIP(ctx, hipMemcpyAsync(dfs, facet_start, fbytes, hipMemcpyHostToDevice, s));
    if (nf) F3D_HIP(ctx, hipMemcpyAsync(deq, facets, ebytes, hipMemcpyHostToDevice, s));
    F3D_HIP(ctx, hipMemcpyAsync(dmg, margin, mbytes, hipMemcpyHostToDevice, s));
    if ((rc = f3d_obb_hull_filter_dev(ctx, ctx->slot[SLOT_XYZ], (f3d_dtype)ctx->grp_dtype, n, (const int32_t*)ctx->slot[SLOT_GRP_ORDER],
                                      (const uint32_t*)ctx->slot[SLOT_GRP_KEYS], (const int64_t*)ctx->slot[SLOT_GRP_STARTS], nids, dfs, deq, dmg,
                                      (int32_t*)dcand, dcnt, s))) return rc;
    if (n) F3D_HIP(ctx, hipMemcpyAsync(cand, dcand, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipMemcpyAsync(cand_count, dcnt, (size_t)nids * 4, hipMemcpyDeviceToHost, s));
    F3D_HIP(ctx, hipStreamSynchronize(s));
    return F3D_OK;
}

}  // extern "C"
