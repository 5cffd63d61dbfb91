// Internal interface between the C-ABI layer (f3d_capi.cpp) and the kernels (f3d_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "f3d.h"

// sticky device error word of a context: one bit per operation, so that an IndexError recorded by one operation is
// neither blamed on nor silently skips another one that shares the context
#define F3D_DEVERR_FUSE 1                  // project_vote_argmax: a sampled label > nclasses (voting.py:98)
#define F3D_DEVERR_VOTE 2                  // vote_uv2pt: point index or label out of bounds (voting.py:98)
#define F3D_DEVERR_CC 4                    // components_same_class: neighbour index out of bounds
#define F3D_DEVERR_ALL 7
#define F3D_PLANES_PER_LAUNCH 16
#define F3D_OBB_MAX_BOXES 4096
#define F3D_SORT_MAX_CELLS 32767            // + 1 overflow cell = 2^15 keys -> 16 key bits sorted

struct f3d_cellgrid {                      // device-resident description of the cell-sort grid
    double lo[3];
    double inv_cell[3];
    int dim[3];                            // cells per axis = 1 << bits[c]
    int bits[3];                           // key bits given to each axis
    int ncells;                            // key space (last key: non-finite points)
};

struct f3d_graphgrid {                     // uniform grid of the radius graph (by-value kernel argument)
    double lo[3];
    double inv_cell;                       // 1 / cell edge; the edge is a hair above the query radius
    int dim[3];
    int pad;
};

struct f3d_plane_args {                    // by-value kernel argument of k_inside_polyhedra
    int m;
    int accumulate;                        // 1: AND into the existing `inside` bytes (chained launches)
    double pt[F3D_PLANES_PER_LAUNCH][3];
    double n[F3D_PLANES_PER_LAUNCH][3];
};

struct f3d_filter_args {                   // filter_classes of VotingSegmentation.segment
    int nfilter;                           // 0 = no filter
    int cls[8];                            // the list itself when nfilter <= 8 (unused slots = -1)
    const int* cls_dev;                    // device copy of the list when nfilter > 8
    const int* cls_host;                   // host copy of the whole list (the context's staging array; NULL when nfilter == 0)
};

// device-resident code book of the fused path's vote bins (built per call by k_mask_presence + k_code_lut, f3d_fuse.hip)
struct f3d_codebook {
    uint8_t lut[256];                      // label -> bin code (0 = no sample / absent label, 1 = rejected label, ...)
    uint8_t inv[256];                      // bin code -> label (must follow lut directly: the kernels stage both with one copy)
    uint16_t cmin[256];                    // (must follow inv directly) cmin[t]: votes below which max / t < threshold (k_threshold_table)
    int ncodes, words, book, pad;          // bins in use, histogram dwords per thread = (ncodes + 3) / 4, 1 presence / 2 filter book
    unsigned presence[8];                  // bit l: label l occurs in the masks
};

hipError_t f3d_launch_clear_error_bits(int* err, int bits, hipStream_t s);
hipError_t f3d_launch_rotate(const double* xyz, int64_t n, const double q[4], double* out, hipStream_t s);
hipError_t f3d_launch_unproject_depth(const void* depth, int depth_type, int h, int w, const double K[9], double scale,
                                      const double q[4], const double t[3], double* out, hipStream_t s);
hipError_t f3d_launch_project_view(const void* xyz, int dtype, int64_t n, const f3d_view& vw, int32_t* uv, uint8_t* inside,
                                   hipStream_t s);
hipError_t f3d_launch_inside_polyhedra(const void* xyz, int dtype, int64_t n, const f3d_plane_args& pa, uint8_t* inside,
                                       hipStream_t s);
size_t f3d_fuse_lds_bytes(int mode, int nclasses);
int f3d_fuse_pick_mode(int nviews, int nfilter, bool want_votes);
// perm (device, may be NULL): caller-order index of sorted point i.  gather_xyz = false: xyz is already the sorted copy;
// gather_xyz = true: xyz is the caller's cloud and the kernel reads point perm[i].  todo_count / todo: device scratch
// (1 counter + n int32) for the points the fast kernel hands to the exact kernel.  masks: the caller's [V,H,W] labels
// (read by the exact kernel); cmasks: their coded, tiled copy made by f3d_launch_code_masks (read by the fast kernel),
// or NULL when nclasses > F3D_CODE_MAX_NCLASSES -- the exact kernel then labels every point.
#define F3D_CODE_MAX_NCLASSES 253            // labels 0..nclasses + "rejected" + "no sample" must fit the 256 byte codes
size_t f3d_fuse_todo_bytes(int64_t n, int nviews, int nclasses);   // the todo_count / todo scratch of f3d_launch_fuse (counters, lists, parked bins)
hipError_t f3d_launch_fuse(const void* xyz, int dtype, int64_t n, const f3d_view* views_dev, int nviews,
                           const uint8_t* masks, const uint8_t* cmasks, int h, int w, int nclasses, const f3d_filter_args& flt,
                           double threshold, int64_t* classes, uint16_t* votes, int* err, const int32_t* perm, bool gather_xyz,
                           unsigned int* todo_count, int32_t* todo, const f3d_codebook* cb, void* tables /* f3d_fuse_tables_bytes(nviews) */,
                           int v0, int v1, uint32_t* carry /* NULL: all views in one launch */, void* xyz_keep, hipStream_t s);
size_t f3d_fuse_tables_bytes(int nviews);
hipError_t f3d_launch_fuse_setup(const f3d_view* views_dev, int v0, int v1, void* tables, f3d_codebook* cb, double threshold,
                                 unsigned int* todo_count, hipStream_t s, int book_nclasses = 0, const f3d_filter_args* book_flt = nullptr,
                                 bool book_want_votes = false);
hipError_t f3d_launch_code_masks_with_setup(const uint8_t* src, uint8_t* dst, int nviews, int h, int w, int nclasses, const f3d_filter_args& flt,
                                            bool want_votes, f3d_codebook* cb, const f3d_view* views_dev, void* tables, double threshold,
                                            unsigned int* todo_count /* NULL: leave the deferred lists' counters */, hipStream_t s);
size_t f3d_fuse_carry_bytes(int64_t n, int nclasses);
hipError_t f3d_launch_mask_presence(const uint8_t* src, int64_t nbytes, f3d_codebook* cb, hipStream_t s);
hipError_t f3d_launch_presence_bytes(f3d_codebook* cb, uint8_t* bytes256, bool to_bytes, hipStream_t s);
hipError_t f3d_launch_code_book(f3d_codebook* cb, int nclasses, const f3d_filter_args& flt, bool want_votes, hipStream_t s);
hipError_t f3d_launch_code_planes(const uint8_t* src, uint8_t* dst, int nviews, int h, int w, const f3d_codebook* cb, hipStream_t s);
// masks [V,H,W] row-major labels -> 8x8-pixel tiles of vote-bin codes (any H, W); dst holds f3d_coded_masks_bytes()
size_t f3d_coded_masks_bytes(int nviews, int h, int w);
// (builds the code book `cb` first: from filter_classes when it is short and no vote rows are wanted, else from the labels present)
hipError_t f3d_launch_code_masks(const uint8_t* src, uint8_t* dst, int nviews, int h, int w, int nclasses, const f3d_filter_args& flt,
                                 bool want_votes, f3d_codebook* cb, hipStream_t s);
// audit of the fast projection (tests only): for every (point, view) pair inside the frustum counts
// stats[0] pairs, stats[1] pairs sent to the exact fallback, stats[2] accepted pairs whose floor differs from the
// canonical path (must stay 0), stats[3] pairs rejected/accepted by the f32 cull that the exact test contradicts (0)
hipError_t f3d_launch_fastpath_audit(const void* xyz, int dtype, int64_t n, const f3d_view* views_dev, int nviews, int w, int h,
                                     unsigned long long* stats_dev, hipStream_t s);
// cell sort (f3d_sort.hip): perm (and sorted_xyz unless NULL) receive the cloud in grid-cell order; scratch >= f3d_sort_scratch_bytes(n)
size_t f3d_sort_scratch_bytes(int64_t n);
hipError_t f3d_launch_cell_sort(const void* xyz, int dtype, int64_t n, void* sorted_xyz, int32_t* perm, void* scratch, hipStream_t s);
// same-class connected components (f3d_cc.hip): root[i] = smallest index of i's component; parent = int32 [n] scratch
hipError_t f3d_launch_components(const int64_t* classes, int64_t n, const int64_t* offs, const int32_t* nbrs, int32_t* parent,
                                 int64_t* root, int* err, hipStream_t s);
// radius graph (f3d_graph.hip): KDTree.query_radius(points, r) as CSR.  bbox partials -> host picks the grid -> count pass
// (offsets[n + 1], exclusive scan) -> fill pass; `scratch` (f3d_graph_scratch_bytes) carries the grid between the passes
size_t f3d_graph_bbox_bytes(void);
hipError_t f3d_launch_graph_bbox(const void* xyz, int dtype, int64_t n, void* partial, int* nblocks, hipStream_t s);
int f3d_graph_reduce_bbox(const void* partial_host, int nblocks, double lo[3], double hi[3]);      // 1: non-finite coordinates seen
size_t f3d_graph_scratch_bytes(int64_t n, int64_t ncells);
hipError_t f3d_launch_graph_count(const void* xyz, int dtype, int64_t n, const f3d_graphgrid& g, double r2, void* scratch,
                                  int64_t* offsets, hipStream_t s);
hipError_t f3d_launch_graph_fill(int64_t n, const f3d_graphgrid& g, double r2, const void* scratch, const int64_t* offsets,
                                 int32_t* nbrs, hipStream_t s);
// a5 patch matching (f3d_patch.hip): owner[p] = first seed (lowest index) whose window covers free pixel p and accepts it, -1 if none
size_t f3d_patch_scratch_bytes(int h, int w, int64_t m);
hipError_t f3d_launch_patch_owner(const int32_t* uv, int64_t m, int h, int w, int half, double radius, double min_cosine,
                                  const double* seed_pts, const double* seed_nrm, const double* q_pts, const double* q_nrm,
                                  const uint8_t* free_px, int32_t* owner, void* scratch, hipStream_t s);
hipError_t f3d_launch_patch_seeds(const double* pts, const double* nrm, const int32_t* prio, const uint8_t* free_px, int h, int w, int half,
                                  double radius, double min_cosine, int32_t* status, int32_t* owner, int32_t* counter, int* rounds,
                                  hipStream_t s);
// ordered per-seed sums of the rows of up to three [h*w, 3] arrays over the pixels each seed owns (uv != NULL: seeds of Fusion.fuse at
// their projections, m of them; uv == NULL: the self-owning pixels of patch_downsample, m = h*w); sums [m, 9], counts [m]
hipError_t f3d_launch_patch_sums(const int32_t* owner, const int32_t* uv, int64_t m, int h, int w, int half, const double* rows_a,
                                 const double* rows_b, const double* rows_c, double* sums, int32_t* counts, hipStream_t s);
// a12: remaining intersections.py primitives (f3d_geom.hip), device pointers
hipError_t f3d_launch_ray_x_lines(const double o[3], const double d[3], const double* starts, const double* ends, int64_t n, double* pts,
                                  uint8_t* within, hipStream_t s);
hipError_t f3d_launch_rays_x_plane(const double pp[3], const double pn[3], const double* origins, const double* dirs, int64_t n, double* pts,
                                   uint8_t* valid, hipStream_t s);
hipError_t f3d_launch_lines_x_planes(const double* lo, const double* le, int64_t n, const double* pps, const double* pns, int m, int bmode,
                                     double* pts, uint8_t* valid, hipStream_t s);
hipError_t f3d_launch_point_inside_polygon(const double* points, int64_t n, const double* verts, int m, uint8_t* inside, uint8_t* within,
                                           hipStream_t s);
hipError_t f3d_launch_points_plane_projection(const double* points, int64_t n, const double pp[3], const double nr[3], double* out, hipStream_t s);
hipError_t f3d_launch_unit_difference(const double* a, const double* b, int64_t n, double* out, hipStream_t s);
hipError_t f3d_launch_segment_votes(const double* votes, int64_t npts, int ncols, int nclasses, double threshold,
                                    const f3d_filter_args& flt, int64_t* classes, hipStream_t s);
hipError_t f3d_launch_vote_uv2pt(const int32_t* uv2pt, const uint8_t* mask, int64_t hw, double* votes, int64_t npts, int ncols,
                                 unsigned long long* table, uint64_t table_slots, int* err, hipStream_t s);
// batched vote: frames [frame0, frame0 + nframes) of a call; table slots carry `gen` (< 16383), first_bad = device int (INT_MAX = none)
hipError_t f3d_launch_vote_uv2pt_batch(const int32_t* luts, const uint8_t* masks, int nframes, int h, int w, double* votes, int64_t npts,
                                       int ncols, unsigned long long* table, uint64_t table_slots, unsigned gen, int frame0, int* first_bad,
                                       int* err, hipStream_t s);
hipError_t f3d_launch_sem_to_mask(const float* sem, int nimg, int c, int64_t hw, float conf, int low_label, uint8_t* mask, hipStream_t s);
// aabb: float [6 * b] of device scratch (the boxes' padded float32 bounds, filled by the launch); cells: f3d_obb_cells_bytes() of
// 8-byte aligned device scratch (the cell table of a call with 8 .. 64 boxes; NULL: every point visits every box)
size_t f3d_obb_cells_bytes();
hipError_t f3d_launch_points_in_obb(const void* xyz, int dtype, int64_t n, const f3d_obb* boxes_dev, int b, float* aabb, void* cells, uint32_t* bits,
                                    uint8_t* cooc, hipStream_t s);
hipError_t f3d_launch_unproject_depth_batch(const void* depth, int depth_type, int nframes, int h, int w, const double K[9], double scale,
                                            const double* q_host, const double* t_host, double* out, hipStream_t s);
hipError_t f3d_launch_relabel(int64_t* ids, int64_t n, int64_t from, int64_t to, unsigned long long* count, hipStream_t s);

// merge_bb support (f3d_obb.hip): grouping of the points by instance id, extreme members along 26 directions, inner-hull filter
#define F3D_OBB_NDIR 26
size_t f3d_group_scratch_bytes(int64_t n, int64_t nids);
hipError_t f3d_launch_group_by_id(const int64_t* ids, int64_t n, int64_t nids, int32_t* order, uint32_t* sorted_keys, int64_t* starts,
                                  void* scratch, hipStream_t s);
hipError_t f3d_launch_obb_extremes(const void* xyz, int dtype, int64_t n, const int32_t* order, const uint32_t* sorted_keys, int64_t nseg,
                                   unsigned long long* table, int32_t* extremes, hipStream_t s);
hipError_t f3d_launch_obb_hull_filter(const void* xyz, int dtype, int64_t n, const int32_t* order, const uint32_t* sorted_keys,
                                      const int64_t* starts, int64_t nseg, const int32_t* fstart, const double* facets, const double* margin,
                                      int32_t* cand, int32_t* cand_count, hipStream_t s);
// the box fit on the device (f3d_hull.hip) and the all-device candidate pipeline (f3d_obb.hip)
#define F3D_OBB_SMALL_FACETS 48              // a triangulated hull of 26 points has at most 2 * 26 - 4 facets
hipError_t f3d_launch_obb_fit(const double* pts, const int64_t* start, int nfit, double* boxes, int32_t* status, uint8_t* isvert, int32_t* vlist,
                              int32_t* nvert, hipStream_t s);
hipError_t f3d_launch_obb_small_hulls(const void* xyz, int dtype, const int32_t* extremes, const int64_t* starts, int64_t nids, int min_members,
                                      double* gathered, uint8_t* isvert, double* facets, int32_t* nfacets, double* margin, hipStream_t s);
size_t f3d_obb_candidates_scratch_bytes(int64_t n, int64_t nids);
hipError_t f3d_launch_obb_candidates(const void* xyz, int dtype, int64_t n, const int32_t* order, const uint32_t* sorted_keys, const int64_t* starts,
                                     int64_t nids, int min_members, unsigned long long* table, void* scratch, int32_t* cand, int64_t* cand_start,
                                     hipStream_t s);
hipError_t f3d_launch_gather_points(const void* xyz, int dtype, const int32_t* idx, int64_t count, double* out, hipStream_t s);
