/*
 * f3d.h -- C-ABI of libf3d_hip.so: the MI355X (gfx950) implementation of the Fusion3DSeg hot
 * path (per-point pinhole projection + frustum visibility, multi-view mask sampling and label
 * voting/argmax, oriented-box membership/merge support).
 *
 * The reference is pure Python/NumPy and has no FFI; the boundary is its Python call surface
 * (SURVEY.md 8(b)).  Each entry point below names the reference function (file:line, relative
 * to the reference repository) whose arithmetic it replaces; INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add at each call site.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / numpy types.
 *   - every function returns F3D_OK (0) or a negative f3d_status; the text of the last error
 *     of a context is available from f3d_last_error().
 *   - functions without a "_dev" suffix take HOST pointers: inputs are copied to the device
 *     through the context's scratch arena, the kernels run on the context's stream, outputs are
 *     copied back, and the call returns after the stream has drained (NumPy drop-in).
 *   - "_dev" functions take DEVICE pointers and a hipStream_t (as void*; NULL = the context's
 *     own stream).  They only enqueue work and never synchronise.  Scratch (sort keys, coded masks,
 *     the deferred-point list, the vote table) lives in the context and GROWS ON FIRST USE of a larger
 *     problem (hipMalloc, not capturable); size it beforehand with f3d_ctx_reserve() and a _dev call
 *     of that or a smaller size performs no allocation at all (hipGraph-safe).  With
 *     f3d_ctx_set_strict(ctx, 1) a call that would have to grow scratch fails with F3D_ERR_NOMEM instead.
 *   - the caller owns every buffer; the library keeps no caller pointer after a call returns
 *     (host variants) / after the enqueued work has completed (_dev variants).
 *   - a context is not thread-safe; use one per thread.  There is no global state.
 *   - quaternions are (w, x, y, z) float64 and are NOT assumed to be normalised (quirk Q4).
 *   - the library has no CPU fallback: without a HIP device every compute entry point fails
 *     with F3D_ERR_HIP.
 */
#ifndef F3D_H
#define F3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F3D_VERSION 100           /* 0.1.0 */
#define F3D_MAX_FILTER 512        /* longest filter_classes list accepted */
#define F3D_NPLANES 5             /* 4 frustum side planes + far plane (fusion.py:254-258) */

typedef enum f3d_status {
    F3D_OK = 0,
    F3D_ERR_INVALID = -1,         /* bad argument (NULL, negative size, unsupported combination)  */
    F3D_ERR_HIP = -2,             /* HIP runtime error / no device / extension not usable         */
    F3D_ERR_INDEX = -3,           /* reference raises IndexError (voting.py:98): label or point id
                                     out of range                                                 */
    F3D_ERR_ZERO_QUAT = -4,       /* reference raises ZeroDivisionError (pyquaternion inverse)    */
    F3D_ERR_NOMEM = -5
} f3d_status;

typedef enum f3d_dtype {
    F3D_F64 = 0,                  /* xyz stored as float64 [N,3] row-major (the reference layout) */
    F3D_F32 = 1                   /* xyz stored as float32 [N,3]; widened to f64 in registers     */
} f3d_dtype;

/* One camera view, in the form the kernels consume (8-byte aligned, 704 bytes = 88 doubles).
 * Built on the host by f3d_views_build(); the fused kernel reads it through scalar loads and
 * stages the float32 cull planes in LDS.
 *   exact data (the reference's arithmetic uses exactly these):  K, qinv, t, plane_pt, plane_n
 *   accelerators (never decide a result on their own, see DESIGN.md "fast paths"):
 *     M, mnorm      : M = K * Rot(qinv) rounded once from extended precision; mnorm[k] bounds the
 *                     row-k operand magnitudes.  The fast projection h = M (p - t) is accepted only
 *                     when its floor provably equals the canonical path's; otherwise the canonical
 *                     arithmetic (camera_utils.py:21-25 order) is evaluated.
 *     cull_*32      : float32 copy of the planes, a = n32 . p32 - off32; a point (or a whole tile's
 *                     bounding box) is accepted/rejected without the exact plane test only when |a|
 *                     exceeds rel32 * (|x|+|y|+|z|) + abs32.
 *     M32           : (float)M, the operator of the float32 "centre + offset" projection: one lane per view
 *                     projects the centre c of a wavefront's bounding box with M in float64, every lane then
 *                     only needs the float32 offset term M32 (p - c); accepted when farther than a rigorous
 *                     bound from a pixel border, otherwise the point goes to the exact kernel. */
typedef struct f3d_view {
    /* hot (128 B = two scalar-cache lines): everything the fast projection reads */
    double M[9];                  /* K * Rot(qinv), row-major                                     */
    double t[3];                  /* camera translation       (camera_utils.py:21)                */
    double mnorm[3];              /* ||K row k||_1 * |qinv|^2, rounded up                          */
    double pad0;
    /* warm (128 B): float32 planes of the pre-culls */
    float  cull_n32[F3D_NPLANES][3];
    float  cull_off32[F3D_NPLANES];
    float  cull_rel32;
    float  cull_abs32;
    float  img_w;                 /* the image the frustum planes were built for (f3d_views_build's w, h)         */
    float  img_h;
    double cull_rel64;            /* float64 refinement of the point cull (middle tier):                          */
    double cull_abs64;            /* a = n . p - plane_off in FMAs; |a| <= rel64 * |p|_1 + abs64 -> exact kernel   */
    float  pad1[4];
    float  M32[9];                /* (float)M[k]                                                                  */
    float  pad2[7];
    /* exact data: what the reference's arithmetic uses */
    double K[9];                  /* intrinsics, row-major (camera_utils.py:23)                   */
    double qinv[4];               /* conj(q)/|q|^2 (w,x,y,z)  (camera_utils.py:22)                */
    double plane_pt[F3D_NPLANES][3];   /* fusion.py:254-257                                       */
    double plane_n[F3D_NPLANES][3];    /* inward normals, fusion.py:256-258                       */
    double plane_off[F3D_NPLANES];     /* n . plane_pt (only for the float64 cull refinement)      */
} f3d_view;

/* An oriented box as open3d's OrientedBoundingBox exposes it (center, R columns = axes, extent);
 * merge_intersecting_bb.py:75-76. */
typedef struct f3d_obb {
    double center[3];
    double R[9];                  /* row-major 3x3, column i = axis i                             */
    double extent[3];
} f3d_obb;

typedef struct f3d_ctx f3d_ctx;

/* ---- context ---------------------------------------------------------------------------- */
int         f3d_version(void);
f3d_ctx*    f3d_ctx_create(int device);            /* NULL on failure (no device, HIP error)      */
void        f3d_ctx_destroy(f3d_ctx* ctx);
const char* f3d_last_error(const f3d_ctx* ctx);    /* ctx may be NULL: last creation error        */
int         f3d_ctx_synchronize(f3d_ctx* ctx);
void*       f3d_ctx_stream(f3d_ctx* ctx);          /* the context's hipStream_t                   */
/* Sizes the context's scratch for f3d_project_vote_argmax_dev / f3d_cloud_sort_cells_dev on up to n points, nviews
 * masks of h x w pixels, and for f3d_vote_uv2pt_dev frames of h*w pixels (any argument may be 0 to skip its part). */
int         f3d_ctx_reserve(f3d_ctx* ctx, int64_t n, int nviews, int h, int w);
/* strict = 1: scratch never grows inside a call; a too-small buffer is F3D_ERR_NOMEM (reserve first). */
int         f3d_ctx_set_strict(f3d_ctx* ctx, int strict);
/* Number of device allocations this context has made so far (diagnostic: unchanged across an allocation-free call). */
long long   f3d_ctx_alloc_count(const f3d_ctx* ctx);

/* ---- host-side geometry (tiny, per view; no device needed) ------------------------------ */
/* pyquaternion Quaternion(q).inverse.elements at camera_utils.py:22 */
int f3d_quat_inverse(const double q_wxyz[4], double qinv_wxyz[4]);
/* Fusion._get_frustum_data (fusion.py:119-132; camera_utils.py:60-171), frame_ids = all.
 * eyes [V,3], lookats [V,3], face_normals [V,4,3]; any output may be NULL. */
int f3d_frustum_data(const double K[9], double w, double h, const double* q_wxyz /*[V,4]*/,
                     const double* t /*[V,3]*/, int nviews,
                     double* eyes, double* lookats, double* face_normals);
/* The packed per-view records: K, inverse pose, and the 5 planes Fusion.fuse builds per frame
 * (fusion.py:254-258) with far plane at eye + max_depth * lookat. */
int f3d_views_build(const double K[9], double w, double h, const double* q_wxyz, const double* t,
                    int nviews, double max_depth, f3d_view* out /*[V]*/);

/* ---- a1: SpatQuadranion.rotate (RTAB_utils/spatQuad.py:7-28) ----------------------------- */
int f3d_rotate_f64(f3d_ctx* ctx, const double* xyz, int64_t n, const double q_wxyz[4], double* out);
/* device pointers (out may not alias xyz), enqueue only; the quaternion is read on the host at call time */
int f3d_rotate_f64_dev(f3d_ctx* ctx, const double* xyz, int64_t n, const double q_wxyz[4], double* out, void* stream);

/* ---- a2: points2pixel (Fusion3DSeg/camera_utils.py:9-26) -> int32 uv[2*n], row 0 = u ------ */
int f3d_points2pixel_f64(f3d_ctx* ctx, const double* xyz, int64_t n, const double K[9],
                         const double q_wxyz[4], const double t[3], int32_t* uv);
int f3d_points2pixel_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n,
                         const double K[9], const double q_wxyz[4], const double t[3],
                         int32_t* uv, void* stream);

/* ---- a4: point_inside_polyhedra (Fusion3DSeg/intersections.py:146-164) -> uint8 inside[n] - */
int f3d_inside_polyhedra_f64(f3d_ctx* ctx, const double* xyz, int64_t n, const double* plane_pts,
                             const double* normals, int m, uint8_t* inside);
int f3d_inside_polyhedra_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n,
                             const double* plane_pts /*host [m,3]*/, const double* normals /*host*/,
                             int m, uint8_t* inside, void* stream);

/* ---- a2+a4 fused for one view: the per-frame body of Fusion.fuse (fusion.py:254-266) ------ */
/* uv is written for every point (like points2pixel), inside as a4. */
int f3d_project_view_f64(f3d_ctx* ctx, const double* xyz, int64_t n, const f3d_view* view /*host*/,
                         int32_t* uv, uint8_t* inside);
int f3d_project_view_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n,
                         const f3d_view* view /*host*/, int32_t* uv, uint8_t* inside, void* stream);

/* ---- fused multi-view path: project -> sample -> vote -> segment ------------------------- */
/* For every point and view: inside (a4) -> uv (a2) -> drop u not in [0,W) or v not in [0,H) ->
 * label = masks[v][row][col] -> one vote; then VotingSegmentation.segment (voting.py:106-137)
 * with `nclasses`, `threshold`, optional `filter` (NULL / 0 = none).
 * classes: int64 [n].  votes_u16 (optional, may be NULL): uint16 [n, nclasses+1] vote counts,
 * equal to the reference's float64 votes matrix on this path.
 * Labels > nclasses make the reference raise IndexError -> F3D_ERR_INDEX (host variant; the
 * _dev variant records it, fetch with f3d_take_device_error after synchronising). */
int f3d_project_vote_argmax(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n,
                            const f3d_view* views, int nviews,
                            const uint8_t* masks /*[V,H,W]*/, int h, int w,
                            int nclasses, const int32_t* filter, int nfilter, double threshold,
                            int64_t* classes, uint16_t* votes_u16);
/* flags for the device variant (results never depend on them):
 *   F3D_FUSE_SORT     the cloud is in arbitrary order: cell-sort it inside the call (into context
 *                     scratch, ~4 HBM passes over xyz) so that a wavefront's 64 points are spatial
 *                     neighbours and culled waves skip the projection; labels are written back in the
 *                     caller's order.  `perm` must be NULL.
 *   F3D_FUSE_GATHER   with `perm`: xyz is still the caller-order cloud and the kernel reads point
 *                     perm[i] (what F3D_FUSE_SORT does internally; lets a caller time / reuse the sort).
 * Every call first finds the labels that occur in the masks and copies the masks into context scratch as 8x8-pixel
 * tiles (one cache line each) of vote-bin codes -- two passes over V*H*W bytes; the 1-byte gathers of neighbouring
 * points then share lines in both image directions, and a thread's vote histogram only has bins for labels that
 * exist.  With nclasses > 253 (no byte left for the "no sample" and "rejected label" codes) the accelerated kernel
 * is skipped and the reference-arithmetic kernel labels every point.
 * perm (device, may be NULL): perm[i] is the caller-order index of the i-th point in cell order
 * (from f3d_cloud_sort_cells_dev); without F3D_FUSE_GATHER xyz must be the sorted copy.
 * classes/votes are always written at caller-order indices. */
#define F3D_FUSE_SORT    2u
#define F3D_FUSE_GATHER  4u
int f3d_project_vote_argmax_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n,
                                const f3d_view* views_dev /*device [V]*/, int nviews,
                                const uint8_t* masks, int h, int w,
                                int nclasses, const int32_t* filter /*host*/, int nfilter,
                                double threshold, int64_t* classes, uint16_t* votes_u16,
                                unsigned flags, const int32_t* perm, void* stream);
/* The same path with the views arriving in CHUNKS (with the masks sharded by view over the ranks, the
 * all-gather of view chunk c+1 is in flight while chunk c votes).  Same results as one f3d_project_vote_argmax_dev call
 * over all views (votes are a sum over views: order-free), at most 255 views in total, no vote output.
 *   f3d_mask_presence_dev      labels that occur in `nviews` masks -> present256 (device, 256 bytes of 0 / 1).  Ranks
 *                              combine theirs with an all-reduce MAX (the vote-bin code book must be the same on every
 *                              rank and for every chunk, before any mask of another rank has arrived).
 *   f3d_fuse_chunked_begin_dev builds the code book from the combined presence (NULL: every label 0..nclasses gets a
 *                              bin -- no exchange, larger histograms) or from `filter`, and sizes ALL scratch of the
 *                              chunk calls (coded masks of nviews views, the deferred lists, and the per-point vote
 *                              state between chunks: f3d "carry", 4 * ceil((nclasses + 3) / 4) bytes per point, 8-bit
 *                              bins in context scratch in HBM).
 *   f3d_fuse_chunk_dev         views [v_begin, v_end): codes those planes of `masks` and lets every point vote on them.
 *                              Chunks must follow each other without gaps from 0 to nviews, with the same xyz / n /
 *                              views_dev / masks base pointer / h / w / nclasses / filter / threshold; flags and perm
 *                              are taken from the first chunk.  masks[v] must be complete for the chunk's views when the
 *                              call is enqueued on `stream` and for ALL views at the last chunk (the float64 tier and the
 *                              reference-arithmetic kernel run there, over every view).  classes is written by the last
 *                              chunk only.  F3D_ERR_INVALID for a chunk out of sequence. */
int f3d_mask_presence_dev(f3d_ctx* ctx, const uint8_t* masks, int nviews, int h, int w, uint8_t* present256, void* stream);
int f3d_fuse_chunked_begin_dev(f3d_ctx* ctx, const uint8_t* present256, int64_t n, int nviews, int h, int w, int nclasses,
                               const int32_t* filter /*host*/, int nfilter, void* stream);
int f3d_fuse_chunk_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_view* views_dev, int nviews,
                       int v_begin, int v_end, const uint8_t* masks /*[nviews,H,W]*/, int h, int w, int nclasses,
                       const int32_t* filter /*host*/, int nfilter, double threshold, int64_t* classes,
                       unsigned flags, const int32_t* perm, void* stream);
/* The same with the mask CODING sharded as well: a rank codes only the masks it produced -- with the book of
 * f3d_fuse_chunked_begin_dev, identical on every rank -- and the ranks all-gather CODED planes (f3d_coded_plane_bytes(h, w) each:
 * 8 x 8-pixel tiles of vote-bin codes with a one-tile border; ~3 % more bytes than the raw plane).
 *   f3d_code_planes_dev        `nplanes` raw masks -> `coded` (nplanes x f3d_coded_plane_bytes, 8-byte aligned), enqueue only.
 *   f3d_fuse_chunk_coded_dev   like f3d_fuse_chunk_dev, but `coded` = the gather buffer of coded planes, [nviews] planes in the
 *                              order of views_dev, read where they lie.  No raw mask ever reaches this rank, so the last
 *                              tier runs the reference's arithmetic on the coded planes. */
size_t f3d_coded_plane_bytes(int h, int w);
int f3d_code_planes_dev(f3d_ctx* ctx, const uint8_t* masks, int nplanes, int h, int w, uint8_t* coded, void* stream);
int f3d_fuse_chunk_coded_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const f3d_view* views_dev, int nviews,
                             int v_begin, int v_end, const uint8_t* coded /*[nviews] coded planes*/, int h, int w,
                             int nclasses, const int32_t* filter /*host*/, int nfilter, double threshold, int64_t* classes,
                             unsigned flags, const int32_t* perm, void* stream);
/* Test hook: cell-sorts the cloud and evaluates, for every (point, view) pair, the accelerated decisions of the
 * fused kernel (wave-box and per-point float32 culls, centre + offset projection for a w x h image) next to the exact
 * arithmetic.  stats[0] = pairs inside the frustum, stats[1] = pairs the offset projection leaves to the exact
 * kernel, stats[2] = decided pairs whose pixel differs from the canonical path (must be 0), stats[3] = float32 cull
 * decisions the exact plane test contradicts (must be 0).  Host pointers. */
int f3d_debug_fastpath_audit(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n,
                             const f3d_view* views, int nviews, int w, int h, uint64_t stats[4]);
/* Diagnostic: how many points of the last fused call of this context the float32 kernel handed to the float64 middle
 * tier (counts[0]) and how many of those needed the reference's own arithmetic for at least one view (counts[1]).
 * Synchronises `stream`. */
int f3d_debug_fuse_deferred(f3d_ctx* ctx, void* stream, uint32_t counts[2]);
/* Sort of the cloud by coarse grid cell: perm (int32 [n], caller-order index of sorted point i)
 * and, unless NULL, sorted_xyz (same dtype/size as xyz); device buffers owned by the caller. */
int f3d_cloud_sort_cells_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n,
                             void* sorted_xyz, int32_t* perm, void* stream);
/* Returns and clears the sticky error recorded by _dev kernels of this context (F3D_OK or F3D_ERR_INDEX; the
 * message names the operation that recorded it: each operation owns one bit of the word, and the host-pointer
 * variants consume only their own).  Synchronises `stream`. */
int f3d_take_device_error(f3d_ctx* ctx, void* stream);

/* ---- a7: one frame of VotingSegmentation.vote (voting.py:94-98) --------------------------- */
/* votes[uv2pt[i], mask[i]] += 1 for uv2pt[i] != -1, each distinct (point,label) pair of the
 * frame counted once (quirk Q1).  votes: float64 [npts, ncols], updated in place. */
int f3d_vote_uv2pt(f3d_ctx* ctx, const int32_t* uv2pt, const uint8_t* mask, int64_t hw,
                   double* votes, int64_t npts, int ncols);
int f3d_vote_uv2pt_dev(f3d_ctx* ctx, const int32_t* uv2pt, const uint8_t* mask, int64_t hw,
                       double* votes, int64_t npts, int ncols, void* stream);

/* The whole loop of VotingSegmentation.vote (voting.py:88-98) for F frames of h x w pixels in one call: luts int32 [F, h*w],
 * masks uint8 [F, h*w] (already at the lookup's resolution).  Same result as F calls of f3d_vote_uv2pt in frame order, incl.
 * the IndexError semantics: the frames before the first offending one are applied, it and the later ones are not
 * (F3D_ERR_INDEX from the host variant; the _dev variant records it for f3d_take_device_error).  One launch pair per 2^25
 * lookups; duplicates die in an LDS set per 32 x 32 tile, the global set is generation-stamped and never cleared. */
int f3d_vote_uv2pt_batch(f3d_ctx* ctx, const int32_t* luts, const uint8_t* masks, int64_t nframes, int h, int w,
                         double* votes, int64_t npts, int ncols);
int f3d_vote_uv2pt_batch_dev(f3d_ctx* ctx, const int32_t* luts, const uint8_t* masks, int64_t nframes, int h, int w,
                             double* votes, int64_t npts, int ncols, void* stream);

/* ---- a8: VotingSegmentation.segment (voting.py:106-137) ----------------------------------- */
int f3d_segment_votes(f3d_ctx* ctx, const double* votes, int64_t npts, int ncols, int nclasses,
                      double threshold, const int32_t* filter, int nfilter, int64_t* classes);
int f3d_segment_votes_dev(f3d_ctx* ctx, const double* votes, int64_t npts, int ncols, int nclasses,
                          double threshold, const int32_t* filter /*host*/, int nfilter,
                          int64_t* classes, void* stream);

/* ---- a9: mask post-processing of SegmentImage (get2DSeg.py:110-118) ----------------------- */
/* sem: float32 [c, hw] logits -> uint8 mask[hw]: argmax over c; softmax max < conf -> `low`. */
int f3d_sem_logits_to_mask(f3d_ctx* ctx, const float* sem, int c, int64_t hw, float conf_threshold,
                           int low_label, uint8_t* mask);
int f3d_sem_logits_to_mask_dev(f3d_ctx* ctx, const float* sem, int c, int64_t hw,
                               float conf_threshold, int low_label, uint8_t* mask, void* stream);
/* The device-resident 2D -> 3D hand-off (SegmentImage's loop, get2DSeg.py:106-126, without the PNG round trip): `nimg` images of
 * logits, float32 [nimg, c, hw], become `nimg` consecutive planes of the caller's uint8 [V, H, W] mask tensor (masks = its base
 * + first_plane * hw) -- the tensor f3d_project_vote_argmax_dev reads.  One launch, enqueued on `stream`; no synchronisation and
 * no copy to the host. */
int f3d_sem_logits_to_masks_dev(f3d_ctx* ctx, const float* sem, int nimg, int c, int64_t hw,
                                float conf_threshold, int low_label, uint8_t* masks, void* stream);

/* ---- a10/a11: oriented-box membership for merge_bb (merge_intersecting_bb.py:64-91) ------- */
/* inside_bits: uint32 [n, ceil(b/32)] -- bit k of word j of point i set iff point i lies in box
 * 32*j+k (|Rt(p-c)|_a <= extent_a/2, a = 0..2).  cooc (optional): uint8 [b,b], cooc[i][j] = 1 iff
 * some point lies in both boxes i and j ("index lists share an element", :64-66,88-90). */
int f3d_points_in_obb(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n,
                      const f3d_obb* boxes, int b, uint32_t* inside_bits, uint8_t* cooc);
int f3d_points_in_obb_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n,
                          const f3d_obb* boxes /*host*/, int b, uint32_t* inside_bits,
                          uint8_t* cooc /*device, zeroed by the call*/, void* stream);
/* update_id_info's relabel (merge_intersecting_bb.py:59-61): ids[ids == from] = to; returns the
 * number of relabelled points through *count (may be NULL). */
int f3d_relabel(f3d_ctx* ctx, int64_t* ids, int64_t n, int64_t from, int64_t to, int64_t* count);
int f3d_relabel_dev(f3d_ctx* ctx, int64_t* ids, int64_t n, int64_t from, int64_t to,
                    int64_t* count_dev, void* stream);

/* ---- a10/a11: per-instance point lists and hull candidates for the oriented-box fits of merge_bb -------- */
/* The reference builds points[ids == id] for every instance (merge_intersecting_bb.py:72-74,81-82,124-125; get3DSeg.py:434)
 * and fits a box on it (Open3D: convex hull -> PCA of the hull vertices).
 * f3d_group_by_id: stable grouping of the point indices by id.  order int32 [n] lists the members of id 0, then id 1, ...
 * (ids outside [0, nids) last), each in ascending point index; starts int64 [nids + 2]: id k owns
 * order[starts[k] .. starts[k + 1]), starts[nids + 1] = n.  sorted_ids (device variant) uint32 [n] = the id of every position.
 * f3d_obb_extremes: extremes int32 [nids, 26] = for every id the member that is extreme along +-x, +-y, +-z and the 10 face /
 * body diagonals (float32 dot products; -1 for an id without members).
 * f3d_obb_hull_filter: with the facets (n . p + o <= 0 inside; double [F, 4], facet_start int32 [nids + 1] per id) of a convex
 * polytope spanned by MEMBERS of the id (e.g. the hull of its <= 26 extremes), drops every member strictly inside it by more
 * than margin[id]: such a point is interior to the hull of all members, so hull and box of the survivors are those of the
 * whole instance.  cand int32 [n]: the survivors of id k at cand[starts[k] ..], cand_count[k] of them, in no particular order.
 * The host-pointer variants form a sequence (group -> extremes -> hull_filter); the grouping and the cloud stay in the context. */
int f3d_group_by_id(f3d_ctx* ctx, const int64_t* ids, int64_t n, int64_t nids, int32_t* order, int64_t* starts);
int f3d_obb_extremes(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, int32_t* extremes);
int f3d_obb_hull_filter(f3d_ctx* ctx, int64_t n, const int32_t* facet_start, const double* facets, const double* margin,
                        int32_t* cand, int32_t* cand_count);
int f3d_group_by_id_dev(f3d_ctx* ctx, const int64_t* ids, int64_t n, int64_t nids, int32_t* order, uint32_t* sorted_ids,
                        int64_t* starts, void* stream);
int f3d_obb_extremes_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const int32_t* order,
                         const uint32_t* sorted_ids, int64_t nids, int32_t* extremes, void* stream);
int f3d_obb_hull_filter_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const int32_t* order,
                            const uint32_t* sorted_ids, const int64_t* starts, int64_t nids, const int32_t* facet_start,
                            const double* facets, const double* margin, int32_t* cand, int32_t* cand_count, void* stream);

/* ---- a11: the oriented-box fit itself (Open3D's OrientedBoundingBox.create_from_points at merge_intersecting_bb.py:75,86,126 and
 * get3DSeg.py:434-435): convex hull -> mean / covariance of the hull vertices -> eigenvectors by descending eigenvalue, third axis =
 * first x second -> extents of the hull vertices in that frame.  One wavefront per instance: hull vertex set by gift wrapping with
 * CERTIFIED orientation signs (float64 + static error bound), cyclic Jacobi for the 3 x 3 eigen-problem.
 * pts: float64 [total, 3], the instances' points back to back (a superset of each hull's vertices is enough: f3d_obb_candidates_dev);
 * start int64 [nfit + 1]: instance k owns pts[start[k] .. start[k + 1]).  boxes: f3d_obb [nfit] (center, R row-major with the axes
 * as columns, extent).  status int32 [nfit]: F3D_OBB_OK; F3D_OBB_FEW = fewer than 4 points; F3D_OBB_DEFERRED = a sign could not be
 * certified (coplanar / duplicate points, non-finite coordinates, a facet with more than 3 vertices) or the frontier outgrew its
 * LDS table -- nothing is guessed, the caller fits that instance on the host (Qhull raises for the truly flat ones, like Open3D).
 * isvert (optional) uint8 [total]: 1 for the hull vertices.  nvert (optional) int32 [nfit].
 * Axis signs: the largest component of the first two axes is positive (the box as a point set does not depend on them; LAPACK /
 * Eigen do not specify theirs).  Open3D itself is absent from the build image: parity with it is unpinned. */
#define F3D_OBB_OK 0
#define F3D_OBB_FEW 1
#define F3D_OBB_DEFERRED 2
int f3d_obb_fit(f3d_ctx* ctx, const double* pts, const int64_t* start, int nfit, double* boxes, int32_t* status,
                uint8_t* isvert, int32_t* nvert);
int f3d_obb_fit_dev(f3d_ctx* ctx, const double* pts, const int64_t* start, int nfit, int64_t total /* = start[nfit] */,
                    double* boxes, int32_t* status, uint8_t* isvert /*device, required*/, int32_t* nvert, void* stream);
/* Hull candidates of EVERY instance in device passes (follows f3d_group_by_id_dev of the same cloud): directional extremes ->
 * hulls of the <= 26 extremes on the device (the same certified wave code) -> members strictly inside that inner polytope are
 * dropped -> the survivors compacted in ascending point index.  cand int32 [n] (capacity): the candidates of id 0, id 1, ... back to
 * back; cand_start int64 [nids + 1].  Instances with fewer than min_members members, or whose inner hull cannot be certified, keep
 * every member.  The hull -- hence the box -- of an instance's candidates is that of all of its members. */
int f3d_obb_candidates_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, const int32_t* order,
                           const uint32_t* sorted_ids, const int64_t* starts, int64_t nids, int min_members,
                           int32_t* cand, int64_t* cand_start, void* stream);
/* out[j] = (double) xyz[idx[j]], j < count: the compact point array f3d_obb_fit_dev reads */
int f3d_gather_points_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, const int32_t* idx, int64_t count, double* out,
                          void* stream);

/* ---- a12 / (f)#4: the other primitives of Fusion3DSeg/intersections.py (host pointers) ---- */
/* ray_x_lines (:6-38): points [n,3], within uint8 [n] */
int f3d_ray_x_lines(f3d_ctx* ctx, const double origin[3], const double direction[3], const double* starts,
                    const double* ends, int64_t n, double* points, uint8_t* within);
/* rays_x_plane (:41-63): points [n,3], valid uint8 [n] */
int f3d_rays_x_plane(f3d_ctx* ctx, const double plane_point[3], const double plane_normal[3], const double* origins,
                     const double* directions, int64_t n, double* points, uint8_t* valid);
/* lines_x_planes (:66-94): points [n,m,3], valid uint8 [n,m].  The reference subtracts [n,3] from [n,m,3] without
 * a new axis (:89-90): it only broadcasts for n == 1 or n == m (then the segment test uses line m); any other
 * shape is F3D_ERR_INVALID, where NumPy raises ValueError. */
int f3d_lines_x_planes(f3d_ctx* ctx, const double* line_origins, const double* line_ends, int64_t n,
                       const double* plane_points, const double* plane_normals, int m, double* points, uint8_t* valid);
/* point_inside_polygon (:97-119): inside uint8 [n], within uint8 [m,n] */
int f3d_point_inside_polygon(f3d_ctx* ctx, const double* points, int64_t n, const double* vertices, int m,
                             uint8_t* inside, uint8_t* within);
/* points_plane_projection (:167-180): out [n,3] */
int f3d_points_plane_projection(f3d_ctx* ctx, const double* points, int64_t n, const double plane_point[3],
                                const double normal[3], double* out);
/* lines_plane_projection (:183-204): start / end projections and unit directions, each [n,3] */
int f3d_lines_plane_projection(f3d_ctx* ctx, const double* starts, const double* ends, int64_t n,
                               const double plane_point[3], const double normal[3], double* start_proj,
                               double* end_proj, double* directions);

/* ---- (f)#1: flood fill of split_into_instances (Fusion3DSeg/segUtils/cv.py:425-440) ------ */
/* Connected components of the adjacency (CSR: offsets int64 [n+1], neighbours int32 [offsets[n]]) restricted to
 * edges whose end points have the same class; root[i] = smallest point index of i's component (= the seed the
 * reference's "lowest remaining index" rule picks).  The adjacency must be symmetric, as KDTree.query_radius
 * (fusion.py:369-377) produces it.  A neighbour index outside [0, n) -> F3D_ERR_INDEX. */
int f3d_components_same_class(f3d_ctx* ctx, const int64_t* classes, int64_t n, const int64_t* offsets,
                              const int32_t* neighbours, int64_t* root);
int f3d_components_same_class_dev(f3d_ctx* ctx, const int64_t* classes, int64_t n, const int64_t* offsets,
                                  const int32_t* neighbours, int32_t* parent_scratch /*int32 [n]*/, int64_t* root,
                                  void* stream);

/* ---- (f)#1: the adjacency itself, Fusion.save_data (Fusion3DSeg/fusion.py:374-375) -------- */
/* tree = KDTree(points); adj = tree.query_radius(points, r=2*ds_radius): for every point the indices of all points
 * (itself included) whose float64 squared distance ((dx*dx + dy*dy) + dz*dz, sklearn's euclidean_rdist order) is
 * <= r*r.  Returned as CSR in two passes because the size is data dependent:
 *   count: offsets int64 [n+1] (exclusive scan, offsets[n] = *nnz); the grid stays in the context,
 *   fill : neighbours int32 [*nnz], row i = offsets[i] .. offsets[i+1]; must follow the count pass of the same cloud.
 * Order inside a row: by grid cell, then ascending index (sklearn's tree-traversal order is unspecified as well;
 * split_into_instances, the only consumer, does not depend on it).  NaN / infinite coordinates -> F3D_ERR_INVALID
 * (sklearn raises ValueError).  The result is symmetric and feeds f3d_components_same_class directly. */
int f3d_radius_graph_count(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, double radius,
                           int64_t* offsets /*[n+1]*/, int64_t* nnz);
int f3d_radius_graph_fill(f3d_ctx* ctx, int64_t n, int32_t* neighbours /*[nnz]*/);
int f3d_radius_graph_count_dev(f3d_ctx* ctx, const void* xyz, f3d_dtype dtype, int64_t n, double radius,
                               int64_t* offsets /*device [n+1]*/, int64_t* nnz /*host*/, void* stream);
int f3d_radius_graph_fill_dev(f3d_ctx* ctx, int64_t n, const int64_t* offsets /*device*/,
                              int32_t* neighbours /*device [nnz]*/, void* stream);

/* ---- a5: patch matching of Fusion.fuse (Fusion3DSeg/fusion.py:269-298) ------------------- */
/* The loop over the in-frustum points ("seeds", in index order) of one frame: seed k takes the still-free depth pixels of
 * the (2*half+1)^2 window around its projection uv[:,k] that lie within `radius` of it and whose normals satisfy
 * dot > min_cosine, judged with the seed's position / normal from before the frame.  Equivalent, and what is computed:
 * owner[p] = the lowest k whose window covers free pixel p and whose test accepts it, -1 if none (or not free).
 * uv int32 [2,m] (row 0 = u, row 1 = v, as points2pixel returns it), seeds [m,3], frame points / normals [h*w,3],
 * free uint8 [h*w] (non_merged), owner int32 [h*w].  The test reproduces NumPy's evaluation order (norm(axis=-1),
 * einsum 'ij,j->i') and the window the reference's slice arithmetic, negative stops included. */
int f3d_patch_owner(f3d_ctx* ctx, const int32_t* uv, int64_t m, int h, int w, int half, double radius, double min_cosine,
                    const double* seed_pts, const double* seed_normals, const double* frame_pts,
                    const double* frame_normals, const uint8_t* free_px, int32_t* owner);
int f3d_patch_owner_dev(f3d_ctx* ctx, const int32_t* uv, int64_t m, int h, int w, int half, double radius,
                        double min_cosine, const double* seed_pts, const double* seed_normals, const double* frame_pts,
                        const double* frame_normals, const uint8_t* free_px, int32_t* owner, void* stream);

/* Fusion.patch_downsample (fusion.py:172-208): the pixels are visited in a shuffled order (prio[p] = position of pixel p); a
 * pixel that is still free becomes a seed and takes the free pixels of its window that pass the same test.  owner[p] = the
 * pixel index of the seed that takes p (a seed owns itself), -1 for pixels nobody takes; seeds are the pixels with
 * owner[p] == p, in ascending prio.  Resolved in data-parallel rounds (*rounds, diagnostic).  Host pointers. */
int f3d_patch_seeds(f3d_ctx* ctx, const double* frame_pts, const double* frame_normals, const int32_t* prio,
                    const uint8_t* free_px, int h, int w, int half, double radius, double min_cosine,
                    int32_t* owner, int32_t* rounds);

/* The same two steps with the ORDERED SUMS of what every seed takes (fusion.py:195-201, 289-298: np.mean over the accepted pixels
 * stacked in window order): one thread per seed adds the rows of its pixels in ascending pixel index, the order NumPy adds them in,
 * so the fused points / normals / colours stay bit-identical.  sums double [.., 9] = rows of frame_pts, frame_normals, frame_colors
 * (colours may be NULL -> zeros), counts int32 = pixels taken.  f3d_patch_match: per seed (m rows); f3d_patch_seeds_sums: per pixel
 * (h*w rows, meaningful where owner[p] == p).  The frame is uploaded once per call. */
int f3d_patch_match(f3d_ctx* ctx, const int32_t* uv, int64_t m, int h, int w, int half, double radius, double min_cosine,
                    const double* seed_pts, const double* seed_normals, const double* frame_pts, const double* frame_normals,
                    const double* frame_colors, const uint8_t* free_px, int32_t* owner, double* sums, int32_t* counts);
int f3d_patch_seeds_sums(f3d_ctx* ctx, const double* frame_pts, const double* frame_normals, const double* frame_colors,
                         const int32_t* prio, const uint8_t* free_px, int h, int w, int half, double radius, double min_cosine,
                         int32_t* owner, double* sums, int32_t* counts, int32_t* rounds);

/* ---- (f)#3: depth frame -> world points (RTAB_utils/ios_rtab.py) -------------------------- */
/* RTAB2Cache.__getRGBP3d (:171-173): x = (px - cx) * (d / fx), y = (py - cy) * (d / fy), z = d with the scaled
 * intrinsics K and the integer pixel grid; __getModP3d: divided by depth_scale (1000: mm -> m, :187), rotated by the
 * frame's camera->world quaternion (w,x,y,z; the pose file stores x,y,z,w, :190) with SpatQuadranion.rotate and
 * translated (:191-192).  depth [h,w] row-major -> xyz float64 [h*w,3], pixel order = row-major, every operation in
 * the reference's order (float64, IEEE division). */
#define F3D_DEPTH_U16 2          /* 16-bit PNG depth as PIL loads it */
#define F3D_DEPTH_F32 1
#define F3D_DEPTH_F64 0
int f3d_unproject_depth(f3d_ctx* ctx, const void* depth, int depth_type, int h, int w, const double K[9],
                        double depth_scale, const double q_wxyz[4], const double t[3], double* xyz /*[h*w*3]*/);
int f3d_unproject_depth_dev(f3d_ctx* ctx, const void* depth, int depth_type, int h, int w, const double K[9],
                            double depth_scale, const double q_wxyz[4], const double t[3], double* xyz, void* stream);
/* F frames per launch (a single 1024 x 1024 frame is launch-bound): depth [F, h, w], q_wxyz host [F, 4], t host [F, 3] ->
 * xyz float64 [F, h*w, 3].  The poses travel in the kernel argument block, 64 frames per launch: enqueue only, no staging. */
int f3d_unproject_depth_batch_dev(f3d_ctx* ctx, const void* depth, int depth_type, int nframes, int h, int w,
                                  const double K[9], double depth_scale, const double* q_wxyz, const double* t,
                                  double* xyz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* F3D_H */
