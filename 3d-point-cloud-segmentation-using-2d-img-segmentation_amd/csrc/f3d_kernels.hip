// HIP kernels of libf3d_hip.so for gfx950 (MI355X, CDNA4).  Wave = 64 lanes.
//
// Everything here is fp64 VALU + memory work: there is no dense contraction on this path, so no
// MFMA.  The kernels are organised for coalesced HBM streams, scalar-register (SGPR) residency of
// the per-view camera records, LDS vote histograms and ≫256 workgroups per launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "f3d.h"
#include "f3d_math.h"
#include "f3d_kernels.h"

#pragma clang fp contract(off)

#define F3D_BLOCK 256

namespace {

template <typename T>
__device__ __forceinline__ f3d_p3 load_point(const T* __restrict__ xyz, int64_t i) {
    const T* p = xyz + 3 * i;
    f3d_p3 r;
    r.x = (double)p[0]; r.y = (double)p[1]; r.z = (double)p[2];
    return r;
}

// ------------------------------------------------------------------------------------------
// a1: rotate
// ------------------------------------------------------------------------------------------
struct quat_arg { double q[4]; };

__global__ __launch_bounds__(F3D_BLOCK) void k_rotate(const double* __restrict__ xyz, int64_t n,
                                                       quat_arg qa, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * F3D_BLOCK) {
        const f3d_p3 o = f3d_rotate(qa.q, load_point(xyz, i));
        out[3 * i] = o.x; out[3 * i + 1] = o.y; out[3 * i + 2] = o.z;
    }
}

// (f)#3 depth frame -> world points: RTAB2Cache.__getRGBP3d (RTAB_utils/ios_rtab.py:171-173), the /1000 of __getModP3d (:187)
// and its rotate + translate (:190-192), fused.  Streaming: 2-8 B in, 24 B out per pixel.
struct unproject_arg { double fx, fy, cx, cy, scale, q[4], t[3]; };

template <typename D>
__global__ __launch_bounds__(F3D_BLOCK) void k_unproject_depth(const D* __restrict__ depth, int h, int w, unproject_arg a,
                                                                double* __restrict__ out) {
    const int64_t n = (int64_t)h * w;
    for (int64_t i = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * F3D_BLOCK) {
        const int64_t row = i / w;
        const double px = (double)(i - row * w), py = (double)row;      // np.linspace(0, W-1, W): the integers, exactly
        const double d = (double)depth[i];
        f3d_p3 c;
        c.x = ((px - a.cx) * (d / a.fx)) / a.scale;                     // :171 then :187
        c.y = ((py - a.cy) * (d / a.fy)) / a.scale;                     // :172
        c.z = d / a.scale;                                              // :173
        const f3d_p3 o = f3d_rotate(a.q, c);                            // :190-191
        out[3 * i] = o.x + a.t[0]; out[3 * i + 1] = o.y + a.t[1]; out[3 * i + 2] = o.z + a.t[2];   // :192
    }
}

// up to F3D_UNPROJECT_BATCH frames in one launch (blockIdx.y = frame): a 1024 x 1024 frame alone is launch-bound (11 us for 27 MB).
// The poses (q (w, x, y, z) then t per frame) travel in the kernel argument block: no staging buffer, no synchronisation.
#define F3D_UNPROJECT_BATCH 64
struct pose_batch { double p[F3D_UNPROJECT_BATCH][7]; };
template <typename D>
__global__ __launch_bounds__(F3D_BLOCK) void k_unproject_depth_batch(const D* __restrict__ depth, int h, int w, unproject_arg a,
                                                                      pose_batch poses, double* __restrict__ out) {
    const int64_t n = (int64_t)h * w;
    const double* ps = poses.p[blockIdx.y];
    double q[4] = {ps[0], ps[1], ps[2], ps[3]};
    const double t0 = ps[4], t1 = ps[5], t2 = ps[6];
    depth += (size_t)blockIdx.y * n;
    out += 3 * (size_t)blockIdx.y * n;
    for (int64_t i = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * F3D_BLOCK) {
        const int64_t row = i / w;
        const double px = (double)(i - row * w), py = (double)row;
        const double d = (double)depth[i];
        f3d_p3 c;
        c.x = ((px - a.cx) * (d / a.fx)) / a.scale;
        c.y = ((py - a.cy) * (d / a.fy)) / a.scale;
        c.z = d / a.scale;
        const f3d_p3 o = f3d_rotate(q, c);
        out[3 * i] = o.x + t0; out[3 * i + 1] = o.y + t1; out[3 * i + 2] = o.z + t2;
    }
}

// ------------------------------------------------------------------------------------------
// a2 (+a4): one view, streaming.  24 B (f64) or 12 B (f32) in, 8 B uv + 1 B inside out per point.
// The view record arrives in the kernarg segment -> SGPRs (wave-uniform).
// ------------------------------------------------------------------------------------------
template <typename T, bool WRITE_UV, bool WRITE_INSIDE>
__global__ __launch_bounds__(F3D_BLOCK) void k_project_view(const T* __restrict__ xyz, int64_t n, f3d_view vw,
                                                             int32_t* __restrict__ uv, uint8_t* __restrict__ inside) {
    for (int64_t i = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * F3D_BLOCK) {
        const f3d_p3 p = load_point(xyz, i);
        if (WRITE_UV) {
            const f3d_p3 h = f3d_project_h(vw.K, vw.qinv, vw.t, p);
            uv[i] = f3d_floor_to_i32(h.x / h.z);                       // camera_utils.py:24-25
            uv[n + i] = f3d_floor_to_i32(h.y / h.z);
        }
        if (WRITE_INSIDE) inside[i] = f3d_inside_view(vw, p) ? 1 : 0;
    }
}

// a4 with an arbitrary plane list (<= F3D_PLANES_PER_LAUNCH per launch, chained with `accumulate`)
template <typename T>
__global__ __launch_bounds__(F3D_BLOCK) void k_inside_polyhedra(const T* __restrict__ xyz, int64_t n, f3d_plane_args pa,
                                                                 uint8_t* __restrict__ inside) {
    for (int64_t i = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * F3D_BLOCK) {
        const f3d_p3 p = load_point(xyz, i);
        bool in = pa.accumulate ? (inside[i] != 0) : true;
        for (int m = 0; m < pa.m; ++m) in = in & (f3d_plane_dp(pa.pt[m], pa.n[m], p) >= 0.0);
        inside[i] = in ? 1 : 0;
    }
}

// (the fused multi-view path -- mask coding, k_fuse, k_fuse_exact, the accelerator audit -- lives in f3d_fuse.hip)

// k-th entry of filter_classes: short lists travel in the kernarg, long ones in device memory
__device__ __forceinline__ int filter_at(const f3d_filter_args& flt, int k) {
    return (flt.nfilter <= 8) ? flt.cls[k & 7] : flt.cls_dev[k];
}

// ------------------------------------------------------------------------------------------
// a8: segment over a dense float64 votes matrix (HBM streaming: ncols*8 B in, 8 B out per point).
// 16 lanes per row, 4 rows per wave; each lane streams 16-B (ncols even) or 8-B pieces of its row.
// ------------------------------------------------------------------------------------------
template <bool VEC2>
__global__ __launch_bounds__(F3D_BLOCK) void k_segment_votes(const double* __restrict__ votes, int64_t npts, int ncols,
                                                              int nclasses, double threshold, f3d_filter_args flt,
                                                              int64_t* __restrict__ classes) {
    const int lane16 = threadIdx.x & 15;
    const int64_t group = ((int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x) >> 4;
    const int64_t ngroups = ((int64_t)gridDim.x * F3D_BLOCK) >> 4;
    for (int64_t row = group; row < npts; row += ngroups) {
        const double* r = votes + (size_t)row * ncols;
        double total = 0.0, best = -INFINITY;
        int besti = 0x7fffffff;
        if (VEC2) {
            const double2* r2 = reinterpret_cast<const double2*>(r);
            const int n2 = ncols >> 1;
            for (int c = lane16; c < n2; c += 16) {
                const double2 x = r2[c];
                total += x.x; total += x.y;
                if (x.x > best) { best = x.x; besti = 2 * c; }
                if (x.y > best) { best = x.y; besti = 2 * c + 1; }
            }
        } else {
            for (int c = lane16; c < ncols; c += 16) {
                const double x = r[c];
                total += x;
                if (x > best) { best = x; besti = c; }
            }
        }
        if (flt.nfilter > 0) {                       // votes[:, filter_classes]: position in the list is the index
            best = -INFINITY; besti = 0x7fffffff;
            for (int k = lane16; k < flt.nfilter; k += 16) {
                const int l = filter_at(flt, k);
                const double x = r[l];
                if (x > best) { best = x; besti = k; }
            }
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            total += __shfl_xor(total, off, 16);
            const double ob = __shfl_xor(best, off, 16);
            const int oi = __shfl_xor(besti, off, 16);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if (lane16 == 0) {
            int64_t cls = besti;
            if (!(total > 0.0)) cls = nclasses;
            else if (best / total < threshold) cls = nclasses;
            if (best == 0.0) cls = nclasses;
            if (flt.nfilter > 0) {
                int64_t q = cls;
                for (int k = 0; k < flt.nfilter; ++k) {
                    if (q == k) q = filter_at(flt, k);
                }
                cls = q;
            }
            classes[row] = cls;
        }
    }
}

// ------------------------------------------------------------------------------------------
// a7: one frame of the uv2pt scatter vote with quirk Q1 (each distinct (point,label) pair of a
// frame adds exactly 1).  Pass 1 validates every index (NumPy raises IndexError before writing
// anything); pass 2 inserts the 64-bit key point*ncols+label into an open-addressing set with
// atomicCAS -- the lane whose insert creates the key performs the single, race-free increment.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(F3D_BLOCK) void k_vote_validate(const int32_t* __restrict__ uv2pt, const uint8_t* __restrict__ mask,
                                                              int64_t hw, int64_t npts, int ncols, int* __restrict__ err) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; i < hw; i += (int64_t)gridDim.x * F3D_BLOCK) {
        const int64_t p = uv2pt[i];
        if (p == -1) continue;
        if (p >= npts || p < -npts || (int)mask[i] >= ncols) bad = true;
    }
    if (bad) atomicOr(err, F3D_DEVERR_VOTE);
}

__global__ __launch_bounds__(F3D_BLOCK) void k_vote_uv2pt(const int32_t* __restrict__ uv2pt, const uint8_t* __restrict__ mask,
                                                           int64_t hw, double* __restrict__ votes, int64_t npts, int ncols,
                                                           unsigned long long* __restrict__ table, uint64_t table_mask,
                                                           const int* __restrict__ err) {
    if (*err & F3D_DEVERR_VOTE) return;
    for (int64_t i = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; i < hw; i += (int64_t)gridDim.x * F3D_BLOCK) {
        int64_t p = uv2pt[i];
        if (p == -1) continue;
        if (p < 0) p += npts;                                         // NumPy negative-index wrap
        const unsigned long long key = (unsigned long long)p * (unsigned long long)ncols + mask[i];
        uint64_t slot = (key * 0x9E3779B97F4A7C15ull) >> 20;
        for (;;) {
            slot &= table_mask;
            const unsigned long long prev = atomicCAS(&table[slot], ~0ull, key);
            if (prev == ~0ull) { votes[key] += 1.0; break; }          // first of its pair in this frame
            if (prev == key) break;                                   // duplicate pair: adds nothing (Q1)
            ++slot;
        }
    }
}

// ------------------------------------------------------------------------------------------
// a7, batched: F frames of the uv2pt scatter vote in one launch (VotingSegmentation.vote's loop, voting.py:88-98).
// Quirk Q1 holds PER FRAME (a (point, label) pair adds 1 per frame however many of the frame's pixels carry it), so the
// de-duplication key is (frame, point, label).  Three levels, cheapest first:
//   1. a block owns a 32 x 32 pixel tile of one frame and de-duplicates in an LDS set -- a fused point's pixels are a patch
//      window (fusion.py:269-298), i.e. they sit next to each other, so most duplicates die here without touching HBM;
//   2. the survivors enter a global open-addressing set whose slots carry a GENERATION stamp: [14 bit generation | 10 bit
//      frame | 39 bit point * ncols + label].  A slot of an older generation counts as empty, so the set is never cleared
//      between calls (the per-frame entry point memsets 16 MiB per 1024^2 frame);
//   3. the lane whose insert creates the key adds 1 to the vote cell (float64 atomic: two frames of a batch may hit one cell).
// NumPy raises IndexError for the first frame with an out-of-range index and leaves the earlier frames applied: pass 1 finds
// the first bad frame of the batch, pass 2 skips it and everything after it.
// ------------------------------------------------------------------------------------------
#define F3D_VOTE_TILE 32
#define F3D_VOTE_LDS_SLOTS 2048
#define F3D_VOTE_EMPTY 0xFFFFFFFFFFFFFFFFull

__global__ __launch_bounds__(F3D_BLOCK) void k_vote_validate_batch(const int32_t* __restrict__ luts, const uint8_t* __restrict__ masks,
                                                                    int64_t nframes, int64_t hw, int64_t npts, int ncols, int frame0,
                                                                    int* __restrict__ first_bad) {
    int bad = 0x7fffffff;
    const int64_t total = nframes * hw;
    for (int64_t i = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * F3D_BLOCK) {
        const int64_t p = luts[i];
        if (p == -1) continue;
        if (p >= npts || p < -npts || (int)masks[i] >= ncols) { const int f = frame0 + (int)(i / hw); bad = f < bad ? f : bad; }
    }
    if (bad != 0x7fffffff) atomicMin(first_bad, bad);
}

__global__ __launch_bounds__(F3D_BLOCK) void k_vote_uv2pt_batch(const int32_t* __restrict__ luts, const uint8_t* __restrict__ masks,
                                                                 int nframes, int h, int w, int tiles_x, int tiles_y,
                                                                 double* __restrict__ votes, int64_t npts, int ncols,
                                                                 unsigned long long* __restrict__ table, uint64_t table_mask, unsigned gen,
                                                                 int frame0, const int* __restrict__ first_bad, int* __restrict__ err) {
    __shared__ unsigned long long lset[F3D_VOTE_LDS_SLOTS];
    // An earlier, untaken IndexError: nothing is written any more.  The bit is only ever set BETWEEN vote launches (k_vote_batch_flag):
    // when blocks of this very launch set it, blocks that started later returned here and dropped their share of the frames before
    // the offending one (an intermittent wrong count, found by running the suite repeatedly).
    if (*err & F3D_DEVERR_VOTE) return;
    const int fb = *first_bad;
    const int tiles = tiles_x * tiles_y;
    const int64_t hw = (int64_t)h * w;
    for (int64_t b = blockIdx.x; b < (int64_t)nframes * tiles; b += gridDim.x) {
        const int f = (int)(b / tiles), t = (int)(b - (int64_t)f * tiles);
        if (frame0 + f >= fb) continue;                               // the reference raised at frame fb: this frame never ran
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        for (int k = threadIdx.x; k < F3D_VOTE_LDS_SLOTS; k += F3D_BLOCK) lset[k] = F3D_VOTE_EMPTY;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < (F3D_VOTE_TILE * F3D_VOTE_TILE) / F3D_BLOCK; ++r) {
            const int local = r * F3D_BLOCK + threadIdx.x;
            const int y = ty * F3D_VOTE_TILE + local / F3D_VOTE_TILE, x = tx * F3D_VOTE_TILE + (local & (F3D_VOTE_TILE - 1));
            if (y >= h || x >= w) continue;
            const int64_t i = (int64_t)f * hw + (int64_t)y * w + x;
            int64_t p = luts[i];
            if (p == -1) continue;
            if (p < 0) p += npts;                                     // NumPy negative-index wrap
            const unsigned long long cell = (unsigned long long)p * (unsigned long long)ncols + masks[i];   // < 2^39
            // 1. block-local set
            unsigned ls = (unsigned)((cell * 0x9E3779B97F4A7C15ull) >> 40);
            bool fresh = false;
            for (;;) {
                ls &= F3D_VOTE_LDS_SLOTS - 1;
                const unsigned long long prev = atomicCAS(&lset[ls], F3D_VOTE_EMPTY, cell);
                if (prev == F3D_VOTE_EMPTY) { fresh = true; break; }
                if (prev == cell) break;
                ++ls;
            }
            if (!fresh) continue;
            // 2. global generation-stamped set
            const unsigned long long mine = ((unsigned long long)gen << 49) | ((unsigned long long)f << 39) | cell;
            uint64_t slot = (mine * 0x9E3779B97F4A7C15ull) >> 20;
            bool created = false;
            for (;;) {
                slot &= table_mask;
                unsigned long long cur = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == mine) break;
                if ((unsigned)(cur >> 49) != gen) {                  // empty or left over from an earlier call: claim it
                    const unsigned long long prev = atomicCAS(&table[slot], cur, mine);
                    if (prev == cur) { created = true; break; }
                    if (prev == mine) break;
                    continue;                                         // somebody else took the slot meanwhile: look at it again
                }
                ++slot;
            }
            // 3. the creator of the key votes
            if (created) atomicAdd(&votes[cell], 1.0);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// a9: [C, HW] float32 logits -> uint8 mask (get2DSeg.py:110-118).  HBM streaming: 4*C B in, 1 B out per pixel.
// A thread owns 4 consecutive pixels and walks the C class planes once with 16-B loads (a wave reads 1 KiB
// contiguous per plane, 8 planes in flight): running argmax (first maximum) and an online softmax denominator
// s = sum exp(x - m) (rescaled when the maximum moves), so the logits are read exactly once.  max prob = 1 / s.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sem_update(float x, int c, float& m, int& mi, float& s) {
    // one exponential per logit: exp(-|x - m|) is the rescale factor when the maximum moves and the new term when it does not
    const float d = x - m;
    const float e = __expf(-fabsf(d));
    const bool up = d > 0.0f;                                    // first maximum wins (strict), NaN never does
    s = up ? s * e + 1.0f : s + e;
    m = up ? x : m;
    mi = up ? c : mi;
}

// blockIdx.y = image of a batch: sem [B][C][hw] -> mask [B][hw] (consecutive planes of the caller's [V,H,W] mask tensor)
template <bool VEC4>
__global__ __launch_bounds__(F3D_BLOCK) void k_sem_to_mask(const float* __restrict__ sem, int C, int64_t hw, float conf,
                                                            int low_label, uint8_t* __restrict__ mask) {
    constexpr int PX = VEC4 ? 4 : 1;
    sem += (size_t)blockIdx.y * (size_t)C * (size_t)hw;
    mask += (size_t)blockIdx.y * (size_t)hw;
    const int64_t ngroups = (hw + PX - 1) / PX;
    for (int64_t gidx = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; gidx < ngroups; gidx += (int64_t)gridDim.x * F3D_BLOCK) {
        const int64_t px0 = gidx * PX;
        float m[PX], s[PX]; int mi[PX];
#pragma unroll
        for (int k = 0; k < PX; ++k) { m[k] = -INFINITY; s[k] = 0.0f; mi[k] = 0; }
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            if (VEC4) {
                const float4 x = *reinterpret_cast<const float4*>(sem + (size_t)c * hw + px0);
                sem_update(x.x, c, m[0], mi[0], s[0]); sem_update(x.y, c, m[PX > 1 ? 1 : 0], mi[PX > 1 ? 1 : 0], s[PX > 1 ? 1 : 0]);
                sem_update(x.z, c, m[PX > 2 ? 2 : 0], mi[PX > 2 ? 2 : 0], s[PX > 2 ? 2 : 0]);
                sem_update(x.w, c, m[PX > 3 ? 3 : 0], mi[PX > 3 ? 3 : 0], s[PX > 3 ? 3 : 0]);
            } else {
                sem_update(sem[(size_t)c * hw + px0], c, m[0], mi[0], s[0]);
            }
        }
        uint8_t lab[PX];
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            int l = mi[k];
            if (conf != 0.0f && 1.0f / s[k] < conf) l = low_label;        // softmax max < conf_threshold -> 133
            lab[k] = (uint8_t)l;
        }
        if (VEC4) *reinterpret_cast<uchar4*>(mask + px0) = make_uchar4(lab[0], lab[PX > 1 ? 1 : 0], lab[PX > 2 ? 2 : 0], lab[PX > 3 ? 3 : 0]);
        else mask[px0] = lab[0];
    }
}

// ------------------------------------------------------------------------------------------
// a10/a11: oriented-box membership.  Thread per point; boxes (c, R, e) staged through LDS in
// chunks of 64; the point's membership bitset lives in LDS as [word][thread] (conflict-free), is
// optionally streamed out (uint32 [n, ceil(B/32)]) and feeds the B x B co-occurrence matrix
// ("the two index lists share a point", merge_intersecting_bb.py:64-66,88-90): only points that
// lie in at least one box do any pair work, and cooc bytes are written once (benign same-value race).
// ------------------------------------------------------------------------------------------
struct obb_consts { double c[3], R[9], e[3]; };
__device__ __forceinline__ obb_consts load_obb(const f3d_obb& b) {
    obb_consts o;
#pragma unroll
    for (int k = 0; k < 3; ++k) { o.c[k] = b.center[k]; o.e[k] = b.extent[k]; }
#pragma unroll
    for (int k = 0; k < 9; ++k) o.R[k] = b.R[k];
    return o;
}
__device__ __forceinline__ void pin(obb_consts& o) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { asm volatile("" : "+s"(o.c[k])); asm volatile("" : "+s"(o.e[k])); }
#pragma unroll
    for (int k = 0; k < 9; ++k) asm volatile("" : "+s"(o.R[k]));
}

// float32 bounds of box k, padded so that a point outside them is certainly outside the box: aabb[k] = lo[3], hi[3].
// The box is {p : |R^T (p - c)|_a <= e_a / 2}: p - c = (R^T)^-1 y with |y_a| <= e_a / 2, so the half width along world axis i is
// sum_a |(R^T)^-1[i][a]| e_a / 2 (= sum_a |R[i][a]| e_a / 2 for the orthonormal R of a fitted box; any other R is handled too, a
// singular one gets infinite bounds: the exact test alone decides).
__global__ __launch_bounds__(F3D_BLOCK) void k_obb_aabb(const f3d_obb* __restrict__ boxes, int B, float* __restrict__ aabb) {
    const int k = blockIdx.x * F3D_BLOCK + threadIdx.x;
    if (k >= B) return;
    const f3d_obb b = boxes[k];
    // M = R^T: M[a][i] = R[3 i + a]
    const double m00 = b.R[0], m01 = b.R[3], m02 = b.R[6], m10 = b.R[1], m11 = b.R[4], m12 = b.R[7], m20 = b.R[2], m21 = b.R[5], m22 = b.R[8];
    const double c00 = m11 * m22 - m12 * m21, c01 = m02 * m21 - m01 * m22, c02 = m01 * m12 - m02 * m11;
    const double c10 = m12 * m20 - m10 * m22, c11 = m00 * m22 - m02 * m20, c12 = m02 * m10 - m00 * m12;
    const double c20 = m10 * m21 - m11 * m20, c21 = m01 * m20 - m00 * m21, c22 = m00 * m11 - m01 * m10;
    const double det = m00 * c00 + m01 * c10 + m02 * c20;
    const double inv[3][3] = {{c00 / det, c01 / det, c02 / det}, {c10 / det, c11 / det, c12 / det}, {c20 / det, c21 / det, c22 / det}};
    const double scale = fabs(m00) + fabs(m01) + fabs(m02) + fabs(m10) + fabs(m11) + fabs(m12) + fabs(m20) + fabs(m21) + fabs(m22);
    const bool usable = fabs(det) > 1e-9 * scale * scale * scale;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double h = (fabs(inv[i][0]) * fabs(b.extent[0]) + fabs(inv[i][1]) * fabs(b.extent[1]) + fabs(inv[i][2]) * fabs(b.extent[2])) * 0.5;
        const double pad = 1e-6 * (fabs(b.center[i]) + h) + 1e-30;        // float32 rounding of the point and of the bounds, the in-box test's own rounding
        float lo = (float)(b.center[i] - h - pad), hi = (float)(b.center[i] + h + pad);
        if (!usable || !(h == h) || !(lo <= hi)) { lo = -INFINITY; hi = INFINITY; }      // the exact test alone decides
        aabb[6 * k + i] = lo; aabb[6 * k + 3 + i] = hi;
    }
}

// the in-box test both membership kernels share (one expression: the same roundings whichever kernel runs)
__device__ __forceinline__ bool obb_inside(const f3d_p3& p, const obb_consts& bx) {
    const double d0 = p.x - bx.c[0], d1 = p.y - bx.c[1], d2 = p.z - bx.c[2];
    bool in = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double pr = (d0 * bx.R[a] + d1 * bx.R[3 + a]) + d2 * bx.R[6 + a];
        in = in & (fabs(pr) <= bx.e[a] / 2);
    }
    return in;
}

// ---- cell table of the boxes (8 .. 64 boxes per call): which boxes can a point of cell c be in? ----
// A grid of F3D_OBB_CELLS cells over the union of the boxes' (finite) bounds; cell c holds the bit set of the boxes whose padded
// bounds meet the cell.  A point then tests only the boxes of its own cell -- for boxes that are small against the scene one or
// two instead of all of them -- whatever order the cloud is in (the per-(wave, box) pre-test of k_points_in_obb prunes little when a
// wave's 64 points are spread over the scene).  Results are those of the exact test alone: the table is a superset argument only.
//   * a point's cell index is computed in float64 (error < 1e-14 cells); a cell's declared range is widened by 1e-6 cells + 2^-48 of the
//     grid's coordinates, its outermost cells reach to infinity (points outside the grid clamp into them);
//   * a box's float32 bounds contain every point the exact test accepts AFTER the point is rounded to float32 (k_obb_aabb's pad): as
//     real intervals they are widened here by 2^-22 of their magnitude; infinite bounds (singular R) meet every cell.
#define F3D_OBB_CELLS 2048
#define F3D_OBB_CELL_MIN_BOXES 8
struct obb_grid { double lo[3], inv[3]; int dim[3]; int pad; };

__global__ __launch_bounds__(F3D_BLOCK) void k_obb_cells(const float* __restrict__ aabb, int B, obb_grid* __restrict__ grid,
                                                          unsigned long long* __restrict__ table) {
    __shared__ float sab[64 * 6];
    const int tid = threadIdx.x;
    for (int k = tid; k < B * 6; k += F3D_BLOCK) sab[k] = aabb[k];
    __syncthreads();
    // union of the finite bounds (every thread: B <= 64, LDS broadcasts)
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < B; ++k) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float l = sab[6 * k + a], h = sab[6 * k + 3 + a];
            if (l > -INFINITY && h < INFINITY) { lo[a] = fmin(lo[a], (double)l); hi[a] = fmax(hi[a], (double)h); }
        }
    }
    double ext[3];
    int dim[3] = {1, 1, 1};
#pragma unroll
    for (int a = 0; a < 3; ++a) { if (!(lo[a] <= hi[a])) { lo[a] = 0.0; hi[a] = 1.0; } ext[a] = hi[a] - lo[a]; }
    for (int it = 0; it < 11; ++it) {                         // 2^11 cells: double the axis with the longest cells
        const double s0 = ext[0] / dim[0], s1 = ext[1] / dim[1], s2 = ext[2] / dim[2];
        if (s0 >= s1 && s0 >= s2) dim[0] *= 2; else if (s1 >= s2) dim[1] *= 2; else dim[2] *= 2;
    }
    double inv[3], h[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { h[a] = ext[a] / dim[a]; inv[a] = ext[a] > 0.0 ? dim[a] / ext[a] : 0.0; }
    if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { grid->lo[a] = lo[a]; grid->inv[a] = inv[a]; grid->dim[a] = dim[a]; }
        grid->pad = 0;
    }
    const int c = blockIdx.x * F3D_BLOCK + tid;
    if (c >= F3D_OBB_CELLS) return;
    const int ci[3] = {c % dim[0], (c / dim[0]) % dim[1], c / (dim[0] * dim[1])};
    double clo[3], chi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double slack = 1e-6 * h[a] + 0x1p-48 * (fabs(lo[a]) + fabs(hi[a]));
        clo[a] = (ci[a] == 0 || inv[a] == 0.0) ? -INFINITY : lo[a] + ci[a] * h[a] - slack;
        chi[a] = (ci[a] == dim[a] - 1 || inv[a] == 0.0) ? INFINITY : lo[a] + (ci[a] + 1) * h[a] + slack;
    }
    unsigned long long m = 0ull;
    for (int k = 0; k < B; ++k) {
        bool meet = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double l = sab[6 * k + a], u = sab[6 * k + 3 + a];
            const double sk = 0x1p-22 * fmax(fabs(l), fabs(u));
            meet = meet & (l - sk <= chi[a]) & (u + sk >= clo[a]);
        }
        m |= meet ? (1ull << k) : 0ull;
    }
    table[c] = m;
}

template <typename T>
__global__ __launch_bounds__(F3D_BLOCK) void k_points_in_obb_cells(const T* __restrict__ xyz, int64_t n, const f3d_obb* __restrict__ boxes, int B,
                                                                    const obb_grid* __restrict__ grid, const unsigned long long* __restrict__ table,
                                                                    uint32_t* __restrict__ bits, uint8_t* __restrict__ cooc) {
    __shared__ unsigned long long stab[F3D_OBB_CELLS];      // 16 KB
    __shared__ double sbox[64 * 15];                        // 7.5 KB: the boxes as they lie in memory (center, R, extent)
    const int tid = threadIdx.x;
    for (int k = tid; k < F3D_OBB_CELLS; k += F3D_BLOCK) stab[k] = table[k];
    for (int k = tid; k < B * 15; k += F3D_BLOCK) sbox[k] = reinterpret_cast<const double*>(boxes)[k];
    const double l0 = grid->lo[0], l1 = grid->lo[1], l2 = grid->lo[2], v0 = grid->inv[0], v1 = grid->inv[1], v2 = grid->inv[2];
    const int n0 = grid->dim[0], n1 = grid->dim[1], n2 = grid->dim[2];
    __syncthreads();
    const int words = (B + 31) >> 5;
    const int64_t ntiles = (n + F3D_BLOCK - 1) / F3D_BLOCK;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t i = tile * F3D_BLOCK + tid;
        const bool live = i < n;
        f3d_p3 p = {0, 0, 0};
        if (live) p = load_point(xyz, i);
        auto cell1 = [](double x, double lo, double inv, int dim) {
            const double t = (x - lo) * inv;
            return t >= (double)dim ? dim - 1 : (t > 0.0 ? (int)t : 0);          // NaN: 0
        };
        const int c = cell1(p.x, l0, v0, n0) + n0 * (cell1(p.y, l1, v1, n1) + n1 * cell1(p.z, l2, v2, n2));
        unsigned long long m = live ? stab[c] : 0ull, res = 0ull;
        while (m) {                                          // per lane: the boxes of this point's cell
            const int k = __builtin_ctzll(m);
            m &= m - 1ull;
            const double* bp = sbox + 15 * k;
            obb_consts bx;
#pragma unroll
            for (int j = 0; j < 3; ++j) { bx.c[j] = bp[j]; bx.e[j] = bp[12 + j]; }
#pragma unroll
            for (int j = 0; j < 9; ++j) bx.R[j] = bp[3 + j];
            res |= obb_inside(p, bx) ? (1ull << k) : 0ull;
        }
        if (live && bits) {
            bits[(size_t)i * words] = (uint32_t)res;
            if (words > 1) bits[(size_t)i * words + 1] = (uint32_t)(res >> 32);
        }
        if (cooc && res) {
            for (unsigned long long a = res; a; a &= a - 1ull) {
                const size_t ia = (size_t)__builtin_ctzll(a);
                for (unsigned long long cc = res; cc; cc &= cc - 1ull) {
                    const size_t ic = (size_t)__builtin_ctzll(cc);
                    if (!cooc[ia * B + ic]) cooc[ia * B + ic] = 1;
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(F3D_BLOCK) void k_points_in_obb(const T* __restrict__ xyz, int64_t n,
                                                              const f3d_obb* __restrict__ boxes, int B, const float* __restrict__ aabb,
                                                              uint32_t* __restrict__ bits, uint8_t* __restrict__ cooc) {
    extern __shared__ uint32_t obb_lds[];                   // [words][F3D_BLOCK] bitset of this tile's points
    const int words = (B + 31) >> 5;
    uint32_t* myb = obb_lds;
    const int tid = threadIdx.x;
    const int64_t ntiles = (n + F3D_BLOCK - 1) / F3D_BLOCK;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t i = tile * F3D_BLOCK + tid;
        const bool live = i < n;
        f3d_p3 p = {0, 0, 0};
        if (live) p = load_point(xyz, i);
        uint32_t any = 0;
        // the boxes are wave-uniform: scalar loads straight from global memory (SGPR operands).  Staging them in LDS made the
        // kernel LDS-bound (15 broadcast 8-B reads per point-box test against ~25 VALU instructions).
        // Pre-test per (wave, box): six float32 compares against the box's padded axis-aligned bounds; the float64 in-box test (27
        // flop) runs only for a wave with a point inside those bounds -- for boxes that are small against the scene almost never.
        const float px = (float)p.x, py = (float)p.y, pz = (float)p.z;
        for (int w0 = 0; w0 < B; w0 += 32) {
            uint32_t word = 0;
            const int lim = min(32, B - w0);
            for (int k = 0; k < lim; ++k) {                             // (four boxes' bounds per scalar round trip, pinned: measured slower, 0.44 vs 0.36 ms)
                const float* ab = aabb + 6 * (size_t)(w0 + k);
                const bool near = live & (px >= ab[0]) & (py >= ab[1]) & (pz >= ab[2]) & (px <= ab[3]) & (py <= ab[4]) & (pz <= ab[5]);
                if (!__any(near)) continue;                              // wave-uniform: a scalar branch
                obb_consts bx = load_obb(boxes[w0 + k]);
                pin(bx);
                word |= (near & obb_inside(p, bx)) ? (1u << (k & 31)) : 0u;
            }
            myb[(w0 >> 5) * F3D_BLOCK + tid] = word;
            any |= word;
            if (live && bits) bits[(size_t)i * words + (w0 >> 5)] = word;
        }
        if (cooc && any) {
            for (int wa = 0; wa < words; ++wa) {
                const uint32_t A = myb[wa * F3D_BLOCK + tid];
                if (!A) continue;
                for (int wc = 0; wc < words; ++wc) {
                    const uint32_t Cw = myb[wc * F3D_BLOCK + tid];
                    if (!Cw) continue;
                    for (uint32_t a = A; a; a &= a - 1) {
                        const size_t ia = (size_t)(wa * 32 + __builtin_ctz(a));
                        for (uint32_t c = Cw; c; c &= c - 1) {
                            const size_t ic = (size_t)(wc * 32 + __builtin_ctz(c));
                            if (!cooc[ia * B + ic]) cooc[ia * B + ic] = 1;
                        }
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(F3D_BLOCK) void k_relabel(int64_t* __restrict__ ids, int64_t n, int64_t from, int64_t to,
                                                        unsigned long long* __restrict__ count) {
    unsigned long long local = 0;
    for (int64_t i = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * F3D_BLOCK) {
        if (ids[i] == from) { ids[i] = to; ++local; }
    }
    if (count) {
        for (int off = 32; off >= 1; off >>= 1) local += __shfl_xor(local, off, 64);
        if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
    }
}

__global__ void k_clear_error_bits(int* err, int bits) { atomicAnd(err, ~bits); }

// after a batched vote launch: a frame of the batch had an out-of-range index -> record the IndexError
__global__ void k_vote_batch_flag(const int* __restrict__ first_bad, int* __restrict__ err) {
    if (*first_bad != 0x7f7f7f7f) atomicOr(err, F3D_DEVERR_VOTE);
}

inline int grid_for(int64_t n, int per_block, int cap) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

// =============================================================================================
// launchers (called from f3d_capi.cpp)
// =============================================================================================
#define F3D_GRID_CAP (256 * 8 * 4)      // 256 CUs x 8 blocks, x4 so that tails stay short

hipError_t f3d_launch_clear_error_bits(int* err, int bits, hipStream_t s) {
    hipLaunchKernelGGL(k_clear_error_bits, dim3(1), dim3(1), 0, s, err, bits);
    return hipGetLastError();
}

hipError_t f3d_launch_rotate(const double* xyz, int64_t n, const double q[4], double* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    quat_arg qa; for (int k = 0; k < 4; ++k) qa.q[k] = q[k];
    hipLaunchKernelGGL(k_rotate, dim3(grid_for(n, F3D_BLOCK, F3D_GRID_CAP)), dim3(F3D_BLOCK), 0, s, xyz, n, qa, out);
    return hipGetLastError();
}

hipError_t f3d_launch_unproject_depth(const void* depth, int depth_type, int h, int w, const double K[9], double scale,
                                      const double q[4], const double t[3], double* out, hipStream_t s) {
    const int64_t n = (int64_t)h * w;
    if (n <= 0) return hipSuccess;
    unproject_arg a;
    a.fx = K[0]; a.fy = K[4]; a.cx = K[2]; a.cy = K[5]; a.scale = scale;
    for (int k = 0; k < 4; ++k) a.q[k] = q[k];
    for (int k = 0; k < 3; ++k) a.t[k] = t[k];
    const dim3 g(grid_for(n, F3D_BLOCK, F3D_GRID_CAP)), b(F3D_BLOCK);
    if (depth_type == F3D_DEPTH_U16) hipLaunchKernelGGL(k_unproject_depth<uint16_t>, g, b, 0, s, (const uint16_t*)depth, h, w, a, out);
    else if (depth_type == F3D_DEPTH_F32) hipLaunchKernelGGL(k_unproject_depth<float>, g, b, 0, s, (const float*)depth, h, w, a, out);
    else if (depth_type == F3D_DEPTH_F64) hipLaunchKernelGGL(k_unproject_depth<double>, g, b, 0, s, (const double*)depth, h, w, a, out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t f3d_launch_unproject_depth_batch(const void* depth, int depth_type, int nframes, int h, int w, const double K[9], double scale,
                                            const double* q_host, const double* t_host, double* out, hipStream_t s) {
    const int64_t n = (int64_t)h * w;
    if (n <= 0 || nframes <= 0) return hipSuccess;
    unproject_arg a;
    a.fx = K[0]; a.fy = K[4]; a.cx = K[2]; a.cy = K[5]; a.scale = scale;
    for (int k = 0; k < 4; ++k) a.q[k] = 0.0;
    for (int k = 0; k < 3; ++k) a.t[k] = 0.0;
    const size_t esz = depth_type == F3D_DEPTH_U16 ? 2 : depth_type == F3D_DEPTH_F32 ? 4 : 8;
    for (int f0 = 0; f0 < nframes; f0 += F3D_UNPROJECT_BATCH) {
        const int nf = nframes - f0 < F3D_UNPROJECT_BATCH ? nframes - f0 : F3D_UNPROJECT_BATCH;
        pose_batch pb;
        for (int f = 0; f < nf; ++f) {
            for (int k = 0; k < 4; ++k) pb.p[f][k] = q_host[4 * (size_t)(f0 + f) + k];
            for (int k = 0; k < 3; ++k) pb.p[f][4 + k] = t_host[3 * (size_t)(f0 + f) + k];
        }
        const int cap = (F3D_GRID_CAP + nf - 1) / nf;
        const dim3 g(grid_for(n, F3D_BLOCK, cap < 64 ? 64 : cap), nf), b(F3D_BLOCK);
        const char* din = reinterpret_cast<const char*>(depth) + (size_t)f0 * n * esz;
        double* dout = out + 3 * (size_t)f0 * n;
        if (depth_type == F3D_DEPTH_U16) hipLaunchKernelGGL(k_unproject_depth_batch<uint16_t>, g, b, 0, s, (const uint16_t*)din, h, w, a, pb, dout);
        else if (depth_type == F3D_DEPTH_F32) hipLaunchKernelGGL(k_unproject_depth_batch<float>, g, b, 0, s, (const float*)din, h, w, a, pb, dout);
        else if (depth_type == F3D_DEPTH_F64) hipLaunchKernelGGL(k_unproject_depth_batch<double>, g, b, 0, s, (const double*)din, h, w, a, pb, dout);
        else return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t f3d_launch_project_view(const void* xyz, int dtype, int64_t n, const f3d_view& vw, int32_t* uv, uint8_t* inside,
                                   hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const dim3 g(grid_for(n, F3D_BLOCK, F3D_GRID_CAP)), b(F3D_BLOCK);
#define F3D_PV(T, U, I) hipLaunchKernelGGL((k_project_view<T, U, I>), g, b, 0, s, (const T*)xyz, n, vw, uv, inside)
    if (dtype == F3D_F64) {
        if (uv && inside) F3D_PV(double, true, true); else if (uv) F3D_PV(double, true, false); else F3D_PV(double, false, true);
    } else {
        if (uv && inside) F3D_PV(float, true, true); else if (uv) F3D_PV(float, true, false); else F3D_PV(float, false, true);
    }
#undef F3D_PV
    return hipGetLastError();
}

hipError_t f3d_launch_inside_polyhedra(const void* xyz, int dtype, int64_t n, const f3d_plane_args& pa, uint8_t* inside,
                                       hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const dim3 g(grid_for(n, F3D_BLOCK, F3D_GRID_CAP)), b(F3D_BLOCK);
    if (dtype == F3D_F64) hipLaunchKernelGGL(k_inside_polyhedra<double>, g, b, 0, s, (const double*)xyz, n, pa, inside);
    else hipLaunchKernelGGL(k_inside_polyhedra<float>, g, b, 0, s, (const float*)xyz, n, pa, inside);
    return hipGetLastError();
}

hipError_t f3d_launch_segment_votes(const double* votes, int64_t npts, int ncols, int nclasses, double threshold,
                                    const f3d_filter_args& flt, int64_t* classes, hipStream_t s) {
    if (npts <= 0) return hipSuccess;
    const dim3 g(grid_for(npts, F3D_BLOCK / 16, F3D_GRID_CAP)), b(F3D_BLOCK);
    const bool vec2 = (ncols % 2 == 0) && ((reinterpret_cast<uintptr_t>(votes) & 15) == 0);
    if (vec2) hipLaunchKernelGGL(k_segment_votes<true>, g, b, 0, s, votes, npts, ncols, nclasses, threshold, flt, classes);
    else hipLaunchKernelGGL(k_segment_votes<false>, g, b, 0, s, votes, npts, ncols, nclasses, threshold, flt, classes);
    return hipGetLastError();
}

hipError_t f3d_launch_vote_uv2pt(const int32_t* uv2pt, const uint8_t* mask, int64_t hw, double* votes, int64_t npts, int ncols,
                                 unsigned long long* table, uint64_t table_slots, int* err, hipStream_t s) {
    if (hw <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(table, 0xFF, table_slots * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const dim3 g(grid_for(hw, F3D_BLOCK, F3D_GRID_CAP)), b(F3D_BLOCK);
    hipLaunchKernelGGL(k_vote_validate, g, b, 0, s, uv2pt, mask, hw, npts, ncols, err);
    hipLaunchKernelGGL(k_vote_uv2pt, g, b, 0, s, uv2pt, mask, hw, votes, npts, ncols, table, (uint64_t)(table_slots - 1), err);
    return hipGetLastError();
}

hipError_t f3d_launch_vote_uv2pt_batch(const int32_t* luts, const uint8_t* masks, int nframes, int h, int w, double* votes, int64_t npts,
                                       int ncols, unsigned long long* table, uint64_t table_slots, unsigned gen, int frame0, int* first_bad,
                                       int* err, hipStream_t s) {
    if (nframes <= 0 || h <= 0 || w <= 0) return hipSuccess;
    const int64_t hw = (int64_t)h * w;
    hipLaunchKernelGGL(k_vote_validate_batch, dim3(grid_for(hw * nframes, F3D_BLOCK, F3D_GRID_CAP)), dim3(F3D_BLOCK), 0, s, luts, masks, (int64_t)nframes, hw,
                       npts, ncols, frame0, first_bad);
    const int tx = (w + F3D_VOTE_TILE - 1) / F3D_VOTE_TILE, ty = (h + F3D_VOTE_TILE - 1) / F3D_VOTE_TILE;
    const int64_t blocks = (int64_t)nframes * tx * ty;
    hipLaunchKernelGGL(k_vote_uv2pt_batch, dim3((unsigned)(blocks < 1048576 ? blocks : 1048576)), dim3(F3D_BLOCK), 0, s, luts, masks, nframes, h, w, tx, ty,
                       votes, npts, ncols, table, (uint64_t)(table_slots - 1), gen, frame0, first_bad, err);
    hipLaunchKernelGGL(k_vote_batch_flag, dim3(1), dim3(1), 0, s, first_bad, err);
    return hipGetLastError();
}

hipError_t f3d_launch_sem_to_mask(const float* sem, int nimg, int c, int64_t hw, float conf, int low_label, uint8_t* mask, hipStream_t s) {
    if (hw <= 0 || nimg <= 0) return hipSuccess;
    if (nimg > 65535) return hipErrorInvalidValue;
    const bool vec4 = (hw % 4 == 0) && ((reinterpret_cast<uintptr_t>(sem) & 15) == 0) && ((reinterpret_cast<uintptr_t>(mask) & 3) == 0);
    const dim3 b(F3D_BLOCK);
    const int cap = nimg > 1 ? (F3D_GRID_CAP + nimg - 1) / nimg : F3D_GRID_CAP;
    if (vec4) hipLaunchKernelGGL(k_sem_to_mask<true>, dim3(grid_for(hw / 4, F3D_BLOCK, cap), nimg), b, 0, s, sem, c, hw, conf, low_label, mask);
    else hipLaunchKernelGGL(k_sem_to_mask<false>, dim3(grid_for(hw, F3D_BLOCK, cap), nimg), b, 0, s, sem, c, hw, conf, low_label, mask);
    return hipGetLastError();
}

size_t f3d_obb_cells_bytes() { return sizeof(obb_grid) + (size_t)F3D_OBB_CELLS * sizeof(unsigned long long); }

hipError_t f3d_launch_points_in_obb(const void* xyz, int dtype, int64_t n, const f3d_obb* boxes_dev, int b, float* aabb, void* cells, uint32_t* bits,
                                    uint8_t* cooc, hipStream_t s) {
    if (n <= 0 || b <= 0) return hipSuccess;
    if (b > F3D_OBB_MAX_BOXES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_obb_aabb, dim3((b + F3D_BLOCK - 1) / F3D_BLOCK), dim3(F3D_BLOCK), 0, s, boxes_dev, b, aabb);
    if (cooc) {
        hipError_t e = hipMemsetAsync(cooc, 0, (size_t)b * b, s);
        if (e != hipSuccess) return e;
    }
    const dim3 g(grid_for(n, F3D_BLOCK, F3D_GRID_CAP)), blk(F3D_BLOCK);
    if (b >= F3D_OBB_CELL_MIN_BOXES && b <= 64 && cells) {     // a point visits the boxes of its cell only
        obb_grid* grid = reinterpret_cast<obb_grid*>(cells);
        unsigned long long* table = reinterpret_cast<unsigned long long*>(grid + 1);
        hipLaunchKernelGGL(k_obb_cells, dim3(F3D_OBB_CELLS / F3D_BLOCK), blk, 0, s, aabb, b, grid, table);
        if (dtype == F3D_F64) hipLaunchKernelGGL(k_points_in_obb_cells<double>, g, blk, 0, s, (const double*)xyz, n, boxes_dev, b, grid, table, bits, cooc);
        else hipLaunchKernelGGL(k_points_in_obb_cells<float>, g, blk, 0, s, (const float*)xyz, n, boxes_dev, b, grid, table, bits, cooc);
        return hipGetLastError();
    }
    const size_t lds = (size_t)((b + 31) / 32) * F3D_BLOCK * sizeof(uint32_t);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (dtype == F3D_F64) {
        if (lds > 48 * 1024) {                             // beyond the default dynamic-LDS limit: the launch needs the attribute
            hipError_t e = hipFuncSetAttribute((const void*)k_points_in_obb<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_points_in_obb<double>, g, blk, lds, s, (const double*)xyz, n, boxes_dev, b, aabb, bits, cooc);
    } else {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)k_points_in_obb<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_points_in_obb<float>, g, blk, lds, s, (const float*)xyz, n, boxes_dev, b, aabb, bits, cooc);
    }
    return hipGetLastError();
}

hipError_t f3d_launch_relabel(int64_t* ids, int64_t n, int64_t from, int64_t to, unsigned long long* count, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_relabel, dim3(grid_for(n, F3D_BLOCK, F3D_GRID_CAP)), dim3(F3D_BLOCK), 0, s, ids, n, from, to, count);
    return hipGetLastError();
}
