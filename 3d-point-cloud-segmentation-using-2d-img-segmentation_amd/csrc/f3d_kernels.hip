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

// ------------------------------------------------------------------------------------------
// fused multi-view kernel: project -> sample -> vote -> segment.
//
// One thread owns one point of a 256-point tile (4 wavefronts of 64 consecutive points) and keeps
// its xyz in registers.  Results are exactly those of the reference arithmetic (oracle order); the
// speed comes from three accelerators that never decide a result unless a rigorous margin says the
// exact arithmetic would agree, and otherwise fall back to it:
//
//  (A) tile pre-cull, lanes-over-views.  Each wave reduces the bounding box of its 64 points, then
//      lane j tests the box against the 5 planes of view 64g+j (float32 planes staged in LDS):
//      box entirely behind a plane -> the whole wave skips that view; box entirely inside all planes
//      -> no per-point cull for that view.  With a cell-sorted cloud ~55 % of (wave, view) pairs are
//      skipped by a scalar bit-scan and ~30 % need no per-lane plane test.
//  (B) per-point pre-cull in float32 for the remaining "mixed" views; lanes within the rounding
//      margin of a plane evaluate the reference's exact float64 plane test (f3d_inside_view).
//  (C) fast projection: h = M (p - t) with M = K Rot(qinv) (3 FMAs per row), one reciprocal; the
//      pixel floor(u), floor(v) is accepted only if u and v are farther from an integer than a bound
//      on |fast - canonical| (both are within ~50 eps * mnorm * |p-t|_1 / |h2| of the real value;
//      the bound uses 2^-43, a >10x margin); otherwise the canonical sequence (un-normalised
//      quaternion sandwich, K @ c, IEEE divisions -- camera_utils.py:21-25) is evaluated.
//
// The view record is read through scalar loads (wave-uniform) -> SGPR operands; the 1-byte mask
// gather is software-pipelined (voted one view later); votes go to a per-thread LDS histogram laid
// out [label/4][thread] (one dword per lane on 64 consecutive banks, conflict-free) with a running
// argmax (count desc, label asc = first-maximum rule of np.argmax), or to 8 register counters when
// filter_classes has <= 8 entries.
// ------------------------------------------------------------------------------------------
enum { MODE_HIST8 = 0, MODE_HIST16 = 1 };

// k-th entry of filter_classes: short lists travel in the kernarg, long ones in device memory
__device__ __forceinline__ int filter_at(const f3d_filter_args& flt, int k) {
    return (flt.nfilter <= 8) ? flt.cls[k & 7] : flt.cls_dev[k];
}

template <int MODE>
struct hist_traits;
template <> struct hist_traits<MODE_HIST8> { static constexpr int per_word = 4, shift = 2, bits = 8; static constexpr uint32_t mask = 0xFFu; };
template <> struct hist_traits<MODE_HIST16> { static constexpr int per_word = 2, shift = 1, bits = 16; static constexpr uint32_t mask = 0xFFFFu; };

#ifndef F3D_CHUNK
#define F3D_CHUNK 2                          // all-in views projected per gather batch (3+ pushes the kernel past 128 VGPRs)
#endif
#define F3D_CULL_ROW 23                       // floats per view in the LDS cull table (22 used, odd stride = no bank conflicts)
#define F3D_FAST_EPS 1.1368683772161603e-13   // 2^-43

// float32 cull planes of one view, copied by value (wave-uniform -> scalar loads -> SGPRs); see load_proj / pin below
struct cull_consts { float n[F3D_NPLANES][3]; float off[F3D_NPLANES]; float rel, abs; };
__device__ __forceinline__ cull_consts load_cull(const f3d_view& vw) {
    cull_consts cc;
#pragma unroll
    for (int m = 0; m < F3D_NPLANES; ++m) {
        cc.n[m][0] = vw.cull_n32[m][0]; cc.n[m][1] = vw.cull_n32[m][1]; cc.n[m][2] = vw.cull_n32[m][2]; cc.off[m] = vw.cull_off32[m];
    }
    cc.rel = vw.cull_rel32; cc.abs = vw.cull_abs32;
    return cc;
}
__device__ __forceinline__ void pin(cull_consts& cc) {
#pragma unroll
    for (int m = 0; m < F3D_NPLANES; ++m) {
        asm volatile("" : "+s"(cc.n[m][0])); asm volatile("" : "+s"(cc.n[m][1])); asm volatile("" : "+s"(cc.n[m][2])); asm volatile("" : "+s"(cc.off[m]));
    }
    asm volatile("" : "+s"(cc.rel)); asm volatile("" : "+s"(cc.abs));
}

// per-point float32 cull against one view (SGPR-resident record): maybe = not surely outside, sure = surely inside
__device__ __forceinline__ void cull_point32(const cull_consts& vw, float px, float py, float pz, float ps, bool small,
                                             bool& maybe, bool& sure) {
    const float marg = __builtin_fmaf(vw.rel, ps, vw.abs);
    bool mb = true, sr = true;
#pragma unroll
    for (int m = 0; m < F3D_NPLANES; ++m) {
        const float a = __builtin_fmaf(vw.n[m][0], px,
                        __builtin_fmaf(vw.n[m][1], py,
                        __builtin_fmaf(vw.n[m][2], pz, -vw.off[m])));
        mb = mb & (a > -marg);
        sr = sr & (a > marg);
    }
    maybe = mb | !small;                      // huge / non-finite coordinates: only the exact test may decide
    sure = sr & small;
}

// byte offset of pixel (iu, iv) inside one view's mask: row-major as the caller hands it over (the exact kernel), or in
// the 8x8-pixel tiled copy (one 64-B line per tile) made by k_code_masks -- neighbouring points of a wave then share
// cache lines in BOTH directions
template <bool TILED>
__device__ __forceinline__ unsigned mask_offset(int iu, int iv, int W) {          // TILED: W = tiles per row
    if (TILED) return (((unsigned)(iv >> 3) * (unsigned)W + (unsigned)(iu >> 3)) << 6) | ((unsigned)(iv & 7) << 3) | (unsigned)(iu & 7);
    return (unsigned)(iv * W + iu);
}

// Bin codes of the fast kernel's vote histogram (nclasses <= F3D_CODE_MAX_NCLASSES): label l <= nclasses lives in bin
// nclasses + 2 - l (so that, among equal counts, the LARGER code is the smaller label), any label > nclasses (IndexError
// in the reference) in bin 1, and bin 0 means "no sample": k_code_masks ends every view with 64 such bytes, and a lane
// without a pixel gathers from there instead of carrying a validity flag through the vote.
#define F3D_CODE_NONE 0u
#define F3D_CODE_BAD 1u
__host__ __device__ inline size_t f3d_coded_plane(int H, int W) { return (size_t)((H + 7) >> 3) * (size_t)((W + 7) >> 3) * 64 + 64; }

// [V,H,W] row-major labels -> [V][ceil(H/8)][ceil(W/8)][8][8] bin codes + the 64-B "no sample" tail; one thread moves one
// tile row (8 bytes), a wave writes 512 contiguous bytes.  VEC: W % 8 == 0 and src 8-B aligned (one 8-B load per thread).
template <bool VEC>
__global__ __launch_bounds__(F3D_BLOCK) void k_code_masks(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int V, int H, int W,
                                                           int nclasses) {
    const int tw = (W + 7) >> 3, th = (H + 7) >> 3;
    const int64_t per_view = (int64_t)th * tw * 8 + 8;             // 8-byte pieces per view, tail included
    const int64_t total = per_view * V;
    const size_t tplane = f3d_coded_plane(H, W);
    for (int64_t k = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; k < total; k += (int64_t)gridDim.x * F3D_BLOCK) {
        const int64_t v = k / per_view; const int64_t r = k - v * per_view;      // r indexes (tile, row-in-tile) of the destination
        uint64_t out = 0;                                                         // tail pieces and padding: F3D_CODE_NONE
        const int tile = (int)(r >> 3), ry = (int)(r & 7);
        const int ty = tile / tw, tx = tile - ty * tw;
        const int y = ty * 8 + ry;
        if (ty < th && y < H) {
            const uint8_t* row = src + (size_t)v * H * W + (size_t)y * W + tx * 8;
            uint64_t x = 0;
            if (VEC) x = *reinterpret_cast<const uint64_t*>(row);
            else for (int c = 0; c < 8; ++c) if (tx * 8 + c < W) x |= (uint64_t)row[c] << (8 * c);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const unsigned l = (unsigned)(x >> (8 * c)) & 0xFFu;
                const unsigned b = l <= (unsigned)nclasses ? (unsigned)nclasses + 2u - l : F3D_CODE_BAD;
                out |= (uint64_t)b << (8 * c);
            }
        }
        *reinterpret_cast<uint64_t*>(dst + (size_t)v * tplane + (size_t)r * 8) = out;
    }
}

// float64 refinement of the point cull (3 FMAs per plane), for the rare lanes inside the float32 margin
__device__ __forceinline__ void cull_point64(const f3d_view& vw, f3d_p3 p, double pscale, bool& maybe, bool& sure) {
    const double marg = __builtin_fma(vw.cull_rel64, pscale, vw.cull_abs64);
    bool mb = true, sr = true;
#pragma unroll
    for (int m = 0; m < F3D_NPLANES; ++m) {
        const double a = __builtin_fma(vw.plane_n[m][0], p.x, __builtin_fma(vw.plane_n[m][1], p.y,
                         __builtin_fma(vw.plane_n[m][2], p.z, -vw.plane_off[m])));
        mb = mb & (a > -marg);
        sr = sr & (a > marg);
    }
    maybe = mb; sure = sr;
}

// The per-view constants of the fast projection, copied by value: wave-uniform, so they are scalar loads into SGPRs, and
// copying them BEFORE the arithmetic that may or may not need them (the cull of a mixed view, the other view of a batch)
// leaves one scalar-memory round trip exposed instead of one per use.
struct proj_consts { double M[9]; double t[3]; };
__device__ __forceinline__ proj_consts load_proj(const f3d_view& vw) {
    proj_consts pc;
#pragma unroll
    for (int k = 0; k < 9; ++k) pc.M[k] = vw.M[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) pc.t[k] = vw.t[k];
    return pc;
}
// pin(): the copies exist in SGPRs at this point of the program.  Without it the compiler sinks each load next to its first
// use (behind the cull of a mixed view, behind the other view's arithmetic of a batch) and the scalar-memory latency is
// paid once per use; with all loads of an iteration requested first and pinned together it is paid once.
__device__ __forceinline__ void pin(proj_consts& pc) {
#pragma unroll
    for (int k = 0; k < 9; ++k) asm volatile("" : "+s"(pc.M[k]));
#pragma unroll
    for (int k = 0; k < 3; ++k) asm volatile("" : "+s"(pc.t[k]));
}

// c1 of the bound below for one view, and 2^-43 times its maximum over all views of the launch (a larger bound only
// defers a few more points): computed once per block, lanes over views
__device__ __forceinline__ double view_c1(const f3d_view& vw, double umax) { return __builtin_fma(umax, vw.mnorm[2], fmax(vw.mnorm[0], vw.mnorm[1])); }
__device__ __forceinline__ double launch_ec1(const f3d_view* __restrict__ views, int nviews, double umax, int lane) {
    double c = 0.0;
    for (int v = lane; v < nviews; v += 64) c = fmax(c, view_c1(views[v], umax));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c = fmax(c, __shfl_xor(c, off, 64));
    return F3D_FAST_EPS * c;
}

// fast projection (C).  Returns true when (iu, iv) are proven equal to the canonical floor(u), floor(v) AND lie inside
// the W x H image; `unsure` is set when the canonical arithmetic has to decide.  umax >= max(W, H): for |u| <= umax
// the bound is rigorous; beyond it both paths are out of the image anyway.  ec1 = 2^-43 c1, eumax = 2^-43 umax.
__device__ __forceinline__ bool project_fast(const proj_consts& pc, double ec1, double eumax, f3d_p3 p, int W, int H, int& iu, int& iv,
                                             bool& unsure) {
    const double d0 = p.x - pc.t[0], d1 = p.y - pc.t[1], d2 = p.z - pc.t[2];
    const double h0 = __builtin_fma(pc.M[0], d0, __builtin_fma(pc.M[1], d1, pc.M[2] * d2));
    const double h1 = __builtin_fma(pc.M[3], d0, __builtin_fma(pc.M[4], d1, pc.M[5] * d2));
    const double h2 = __builtin_fma(pc.M[6], d0, __builtin_fma(pc.M[7], d1, pc.M[8] * d2));
    double r = __builtin_amdgcn_rcp(h2);
    r = __builtin_fma(__builtin_fma(-h2, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-h2, r, 1.0), r, r);
    const double uf = h0 * r, vf = h1 * r;
    const double fu = floor(uf), fv = floor(vf);
    // |fast - canonical| <= 2^-43 * (|d|_1 |r| (mnorm_k + |u| mnorm_2) + |u|), evaluated with |u| <= umax
    const double b = __builtin_fma(((fabs(d0) + fabs(d1)) + fabs(d2)) * fabs(r), ec1, eumax);
    // frac in (b, 1-b)  <=>  |frac - 0.5| < 0.5 - b      (NaN / inf -> false)
    const bool safe = (fabs((uf - fu) - 0.5) < 0.5 - b) && (fabs((vf - fv) - 0.5) < 0.5 - b);
    unsure = !safe;
    iu = (int)fu; iv = (int)fv;                              // saturating conversions; only used when safe
    return safe & ((unsigned)iu < (unsigned)W) & ((unsigned)iv < (unsigned)H);
}

__device__ __forceinline__ void project_exact(const f3d_view& vw, f3d_p3 p, double& fu, double& fv) {
    const f3d_p3 h = f3d_project_h(vw.K, vw.qinv, vw.t, p);
    fu = floor(h.x / h.z); fv = floor(h.y / h.z);
}

// ---- vote state shared by the fast kernel (k_fuse) and the exact kernel (k_fuse_exact)
template <int MODE>
struct vote_state {
    int total = 0;
    unsigned best = 0;                   // (count << 16) | (0xFFFF - label): the maximum is the highest count, then the lowest label
    bool bad = false;
};

// branch-free vote: lanes without a sample vote into a spare bin (index ncols) that nothing reads
template <int MODE>
__device__ __forceinline__ void vote_add(vote_state<MODE>& st, uint32_t* hist, int tid, const f3d_filter_args& flt, int nclasses,
                                         bool valid, int label) {
    using HT = hist_traits<MODE>;
    st.bad = st.bad | (valid & (label > nclasses));                              // IndexError in the reference (flagged per tile)
    valid = valid & (label <= nclasses);
    st.total += valid ? 1 : 0;
    const unsigned l = valid ? (unsigned)label : (unsigned)nclasses + 1u;
    const unsigned sh = (l & (HT::per_word - 1)) * HT::bits;
    const uint32_t old = atomicAdd(&hist[(l >> HT::shift) * F3D_BLOCK + tid], 1u << sh);
    const unsigned c = ((old >> sh) & HT::mask) + 1u;
    const unsigned key = valid ? ((c << 16) | (0xFFFFu - l)) : 0u;
    st.best = st.best > key ? st.best : key;
}

// VotingSegmentation.segment (voting.py:120-135) for one point, then the stores
template <int MODE, bool WRITE_VOTES>
__device__ __forceinline__ void finish_point(const vote_state<MODE>& st, const uint32_t* hist, int tid, const f3d_filter_args& flt,
                                             int nclasses, double threshold, bool store, int64_t orig,
                                             int64_t* __restrict__ classes, uint16_t* __restrict__ votes_out) {
    using HT = hist_traits<MODE>;
    const int ncols = nclasses + 1;
    int64_t cls;
    int win_c, win_i;
    if (flt.nfilter > 0) {                                                       // votes[:, filter_classes]: first maximum wins
        win_c = -1; win_i = 0;
        for (int k = 0; k < flt.nfilter; ++k) {
            const int l = filter_at(flt, k);
            int c = 0;
            if (l >= 0 && l < ncols) c = (int)((hist[(l >> HT::shift) * F3D_BLOCK + tid] >> ((l & (HT::per_word - 1)) * HT::bits)) & HT::mask);
            if (c > win_c) { win_c = c; win_i = k; }
        }
    } else {
        win_c = (int)(st.best >> 16); win_i = (int)(0xFFFFu - (st.best & 0xFFFFu));
    }
    if (st.total == 0) cls = nclasses;                                             // :126
    else {
        cls = win_i;
        if ((double)win_c / (double)st.total < threshold) cls = nclasses;          // :128-130
        if (win_c == 0) cls = nclasses;                                            // :131
    }
    if (flt.nfilter > 0) {                                                         // sequential remap (Q3)
        int64_t r = cls;
        for (int k = 0; k < flt.nfilter; ++k) if (r == k) r = filter_at(flt, k);
        cls = r;
    }
    if (store) classes[orig] = cls;
    if (WRITE_VOTES && store) {
        for (int l = 0; l < ncols; ++l)
            votes_out[(size_t)orig * ncols + l] =
                (uint16_t)((hist[(l >> HT::shift) * F3D_BLOCK + tid] >> ((l & (HT::per_word - 1)) * HT::bits)) & HT::mask);
    }
}

// ---- the fast kernel's vote: `b` is a bin code read from the coded masks (see k_code_masks).  No validity flag, no label
// range test: "no sample" adds 0 to bin 0 and yields key 0, a label the reference would reject lands in bin 1 and is
// reported by finish_coded.  key = (count before this vote << 16) | code: its running maximum is the plurality, ties
// going to the smaller label (the larger code), exactly the first-maximum rule of np.argmax.
struct coded_state {
    unsigned total = 0;
    unsigned best = 0;
};

template <int MODE>
__device__ __forceinline__ void vote_coded(coded_state& st, uint32_t* hist, int tid, unsigned b) {
    using HT = hist_traits<MODE>;
    const unsigned one = b < 1u ? b : 1u;                                          // 0 for F3D_CODE_NONE
    st.total += one;
    const unsigned sh = (b & (HT::per_word - 1)) * HT::bits;
    const uint32_t old = atomicAdd(&hist[((b >> HT::shift) * F3D_BLOCK) | (unsigned)tid], one << sh);
    const unsigned key = (((old >> sh) & HT::mask) << 16) | b;
    st.best = st.best > key ? st.best : key;
}

template <int MODE>
__device__ __forceinline__ unsigned coded_count(const uint32_t* hist, int tid, unsigned b) {
    using HT = hist_traits<MODE>;
    return (hist[((b >> HT::shift) * F3D_BLOCK) | (unsigned)tid] >> ((b & (HT::per_word - 1)) * HT::bits)) & HT::mask;
}

// VotingSegmentation.segment (voting.py:120-135) for one point of the fast kernel, then the stores; returns whether a label
// the reference would raise IndexError for was sampled
template <int MODE, bool WRITE_VOTES>
__device__ __forceinline__ bool finish_coded(const coded_state& st, const uint32_t* hist, int tid, const f3d_filter_args& flt,
                                             int nclasses, double threshold, bool store, int64_t orig,
                                             int64_t* __restrict__ classes, uint16_t* __restrict__ votes_out) {
    const int ncols = nclasses + 1;
    const unsigned top = (unsigned)nclasses + 2u;                                  // code of label 0
    int64_t cls;
    int win_c, win_i;
    if (flt.nfilter > 0) {                                                         // votes[:, filter_classes]: first maximum wins
        win_c = -1; win_i = 0;
        for (int k = 0; k < flt.nfilter; ++k) {
            const int l = filter_at(flt, k);
            int c = 0;
            if (l >= 0 && l < ncols) c = (int)coded_count<MODE>(hist, tid, top - (unsigned)l);
            if (c > win_c) { win_c = c; win_i = k; }
        }
    } else {
        win_c = (int)(st.best >> 16) + 1; win_i = (int)(top - (st.best & 0xFFFFu));
    }
    if (st.total == 0) cls = nclasses;                                             // :126
    else {
        cls = win_i;
        if ((double)win_c / (double)st.total < threshold) cls = nclasses;          // :128-130
        if (win_c == 0) cls = nclasses;                                            // :131
    }
    if (flt.nfilter > 0) {                                                         // sequential remap (Q3)
        int64_t r = cls;
        for (int k = 0; k < flt.nfilter; ++k) if (r == k) r = filter_at(flt, k);
        cls = r;
    }
    if (store) classes[orig] = cls;
    if (WRITE_VOTES && store) {
        for (int l = 0; l < ncols; ++l) votes_out[(size_t)orig * ncols + l] = (uint16_t)coded_count<MODE>(hist, tid, top - (unsigned)l);
    }
    return coded_count<MODE>(hist, tid, F3D_CODE_BAD) != 0u;
}

// Wave-wide min / max of a float through DPP (no LDS traffic, no s_waitcnt): xor-1 and xor-2 inside quads, mirror the half
// rows and the rows (every lane of a 16-lane row then holds the row's result), row_bcast15 / row_bcast31 carry it across the
// rows into lane 63, which is read back as a scalar.  6 VALU + 1 v_readlane per value; the __shfl_xor butterfly it replaces
// was 6 ds_bpermute (LDS crossbar, each with its wait) + 6 VALU.  Inputs are never NaN (+-inf for lanes without a point).
template <bool MAX>
__device__ __forceinline__ float wave_reduce(float v) {
    // written as one asm block: the compiler's own lowering of update_dpp + fminf spends 4 VALU per step (copy, v_mov_dpp,
    // canonicalise, min); v_min/v_max take the DPP operand directly.  s_nop 1 = the 2 wait states a DPP read needs after
    // a VALU write of the same register.  Must be called with all 64 lanes active.
#define F3D_DPP_CHAIN(op)                                                                  \
    asm volatile("s_nop 1\n\t" op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
                 "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
                 "s_nop 1\n\t" op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
                 "s_nop 1\n\t" op " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"          \
                 "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
                 "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"        \
                 "s_nop 1" : "+v"(v))
    if (MAX) F3D_DPP_CHAIN("v_max_f32_dpp"); else F3D_DPP_CHAIN("v_min_f32_dpp");
#undef F3D_DPP_CHAIN
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ------------------------------------------------------------------------------------------
// k_fuse: the fast kernel.  It contains NO exact arithmetic: a point for which any accelerator cannot prove its
// decision (a plane within the float32 margin, a pixel within the fast-projection bound of an integer, huge or
// non-finite coordinates) is not stored; its caller-order index is appended to `todo` and k_fuse_exact, launched
// right behind, recomputes that point entirely with the reference's arithmetic (about 1e-3 of the points of a
// random cloud).  Keeping the canonical sequences out of this kernel is what keeps its register budget small.
// ------------------------------------------------------------------------------------------
template <typename T, bool WRITE_VOTES>
__global__ __launch_bounds__(F3D_BLOCK) void k_fuse(const T* __restrict__ xyz, int64_t n,
                                                     const f3d_view* __restrict__ views, int nviews,
                                                     const uint8_t* __restrict__ cmasks, int H, int W,
                                                     int nclasses, f3d_filter_args flt, double threshold,
                                                     int64_t* __restrict__ classes, uint16_t* __restrict__ votes_out,
                                                     int* __restrict__ err, const int32_t* __restrict__ perm, int gather_xyz,
                                                     unsigned int* __restrict__ todo_count, int32_t* __restrict__ todo) {
    // 8-bit bins whatever the number of views: a bin that reaches 255 shows up in the running maximum (count field 255 = a
    // vote that found the bin full) and sends the point to k_fuse_exact, which counts in 16 bits when V > 255
    constexpr int MODE = MODE_HIST8;
    using HT = hist_traits<MODE>;
    extern __shared__ uint32_t lds_u32[];
    float* ctab = reinterpret_cast<float*>(lds_u32);                      // [64][F3D_CULL_ROW] cull planes of one view group
    uint32_t* hist = lds_u32 + 64 * F3D_CULL_ROW;                         // [words_per_thread][F3D_BLOCK]
    const int tid = threadIdx.x, lane = threadIdx.x & 63;
    const int ncols = nclasses + 1;
    const int words = (ncols + 2 + HT::per_word - 1) >> HT::shift;       // + 2: codes F3D_CODE_NONE and F3D_CODE_BAD
    const int64_t ntiles = (n + F3D_BLOCK - 1) / F3D_BLOCK;
    const size_t plane = f3d_coded_plane(H, W);                           // bytes per view of the coded, tiled masks
    const int wt = (W + 7) >> 3;                                          // tiles per row
    const unsigned none_off = (unsigned)(plane - 64);                     // where a lane without a pixel gathers F3D_CODE_NONE
    const int ngroups = (nviews + 63) >> 6;
    const double umax = (double)(W > H ? W : H);

    auto stage_group = [&](int g) {                                       // whole block; caller brackets with barriers
        const int nv = min(64, nviews - 64 * g);
        for (int k = tid; k < nv * 22; k += F3D_BLOCK) {
            const int vi = k / 22, f = k - vi * 22;
            ctab[vi * F3D_CULL_ROW + f] = reinterpret_cast<const float*>(&views[64 * g + vi].cull_n32[0][0])[f];
        }
    };
    if (ngroups == 1) { stage_group(0); __syncthreads(); }
    const double ec1 = launch_ec1(views, nviews, umax, lane), eumax = F3D_FAST_EPS * umax;

    // XCD-aware tile mapping: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch), so XCD x walks the
    // contiguous tile range [x*q, (x+1)*q): with a cell-sorted cloud that is one compact region of space, whose pixels
    // in every mask stay resident in that XCD's 4 MiB L2 (measured: FETCH_SIZE 5.1 GB -> 1.8 GB per launch, L2 hit 77 %).
    // Placement affects speed only, never results.
    const int64_t tiles_per_xcd = (ntiles + 7) / 8;
    const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, gx = gridDim.x >> 3;
    // Two-deep software pipeline over this block's tiles: perm[] of tile k+2 and xyz of tile k+1 are requested at the top of
    // tile k.  With the in-step sort the point of a lane is xyz[perm[i]], a random 24-B read that nothing else overlaps.
    auto tile_index = [&](int64_t jj) -> int64_t {          // first point index of this thread in tile jj of the block's walk, or -1
        if (jj >= tiles_per_xcd) return -1;
        const int64_t t = (int64_t)xcd * tiles_per_xcd + jj;
        if (t >= ntiles) return -1;
        const int64_t ii = t * F3D_BLOCK + tid;
        return ii < n ? ii : -1;
    };
    auto fetch_orig = [&](int64_t ii) -> int64_t { return ii < 0 ? -1 : (perm ? (int64_t)perm[ii] : ii); };
    auto fetch_point = [&](int64_t ii, int64_t oo) -> f3d_p3 {
        f3d_p3 q = {0.0, 0.0, 0.0};
        if (ii >= 0) q = load_point(xyz, gather_xyz ? oo : ii);
        return q;
    };
    int64_t orig_n1 = fetch_orig(tile_index(bx));
    f3d_p3 p_n1 = fetch_point(tile_index(bx), orig_n1);
    int64_t orig_n2 = fetch_orig(tile_index(bx + gx));
    for (int64_t j = bx; j < tiles_per_xcd; j += gx) {
        const int64_t tile = (int64_t)xcd * tiles_per_xcd + j;
        if (tile >= ntiles) break;
        const int64_t i = tile * F3D_BLOCK + tid;
        const bool live = i < n;
        const f3d_p3 p = p_n1;
        const int64_t orig = live ? orig_n1 : i;                                  // caller-order index of this point
        orig_n1 = orig_n2;
        p_n1 = fetch_point(tile_index(j + gx), orig_n1);
        orig_n2 = fetch_orig(tile_index(j + 2 * gx));
        const double pscale = (fabs(p.x) + fabs(p.y)) + fabs(p.z);
        const float px32 = (float)p.x, py32 = (float)p.y, pz32 = (float)p.z, ps32 = (float)pscale;
        const bool small = pscale < 1.0e30;                // float32 culls are meaningful (no overflow, no NaN)
        bool defer = live & !small;                        // this point goes to k_fuse_exact

        // ---- (A) bounding box of this wave's live, well-behaved points
        float lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;
        if (live & small) { lo0 = hi0 = px32; lo1 = hi1 = py32; lo2 = hi2 = pz32; }
        lo0 = wave_reduce<false>(lo0); hi0 = wave_reduce<true>(hi0);
        lo1 = wave_reduce<false>(lo1); hi1 = wave_reduce<true>(hi1);
        lo2 = wave_reduce<false>(lo2); hi2 = wave_reduce<true>(hi2);
        const bool wave_any = __any(live & small);
        const float c0 = 0.5f * (lo0 + hi0), c1 = 0.5f * (lo1 + hi1), c2 = 0.5f * (lo2 + hi2);
        const float e0 = 0.5f * (hi0 - lo0) * 1.000002f + 1e-30f, e1 = 0.5f * (hi1 - lo1) * 1.000002f + 1e-30f,
                    e2 = 0.5f * (hi2 - lo2) * 1.000002f + 1e-30f;
        const float ps_box = ((fabsf(c0) + fabsf(c1)) + fabsf(c2)) + ((e0 + e1) + e2);

        for (int wd = 0; wd < words; ++wd) hist[wd * F3D_BLOCK + tid] = 0u;       // own column only: no barrier
        coded_state st;
        unsigned pend_code = F3D_CODE_NONE;                 // software-pipelined gather: vote one view later
        unsigned ccode[F3D_CHUNK];
#pragma unroll
        for (int k = 0; k < F3D_CHUNK; ++k) ccode[k] = F3D_CODE_NONE;

        for (int g = 0; g < ngroups; ++g) {
            if (ngroups > 1) { __syncthreads(); stage_group(g); __syncthreads(); }
            // lane j <-> view 64g + j: classify the wave's box against that view's planes
            const int vj = 64 * g + lane;
            bool box_out = false, box_in = true;
            if (vj < nviews) {
                const float* row = ctab + lane * F3D_CULL_ROW;
                const float marg = 2.0f * __builtin_fmaf(row[20], ps_box, row[21]);
#pragma unroll
                for (int m = 0; m < F3D_NPLANES; ++m) {
                    const float n0 = row[3 * m], n1 = row[3 * m + 1], n2 = row[3 * m + 2];
                    const float base = __builtin_fmaf(n0, c0, __builtin_fmaf(n1, c1, __builtin_fmaf(n2, c2, -row[15 + m])));
                    const float spread = __builtin_fmaf(fabsf(n0), e0, __builtin_fmaf(fabsf(n1), e1, fabsf(n2) * e2));
                    box_out = box_out | (base + spread < -marg);
                    box_in = box_in & (base - spread > marg);
                }
            }
            const unsigned long long valid_m = __ballot(vj < nviews);
            unsigned long long out_m = __ballot(vj < nviews && box_out);
            const unsigned long long in_m = __ballot(vj < nviews && box_in && !box_out);
            if (!wave_any) out_m = valid_m;                 // nothing but deferred / dead lanes in this wave
            // views whose planes all contain the wave's box: every live lane is inside, no cull, no divergence.
            // Taken F3D_CHUNK at a time: project the chunk, retire the previous chunk's votes, then issue the chunk's mask
            // gathers back to back -- F3D_CHUNK gathers in flight per wave, each with a whole chunk of arithmetic to land
            // (in-kernel stamps showed ~90 % of a view iteration waiting for the previous gather with a 1-deep pipeline).
            unsigned long long todo_v = valid_m & ~out_m & in_m;
            while (todo_v) {
                int cv[F3D_CHUNK]; unsigned coff[F3D_CHUNK];
                bool use[F3D_CHUNK];
                proj_consts pc[F3D_CHUNK];
#pragma unroll
                for (int k = 0; k < F3D_CHUNK; ++k) {       // the batch's scalar loads go out together
                    use[k] = todo_v != 0ull;
                    const int bit = use[k] ? __builtin_ctzll(todo_v) : 0;
                    if (use[k]) todo_v &= todo_v - 1ull;
                    cv[k] = 64 * g + bit;                    // an unused slot re-reads a valid record and gathers "no sample"
                    pc[k] = load_proj(views[cv[k]]);
                }
#pragma unroll
                for (int k = 0; k < F3D_CHUNK; ++k) pin(pc[k]);
#pragma unroll
                for (int k = 0; k < F3D_CHUNK; ++k) {
                    int iu, iv;
                    bool unsure;
                    const bool hit = project_fast(pc[k], ec1, eumax, p, W, H, iu, iv, unsure) & live & small & use[k];
                    defer = defer | (unsure & live & use[k]);
                    coff[k] = hit ? mask_offset<true>(iu, iv, wt) : none_off;
                }
#pragma unroll
                for (int k = 0; k < F3D_CHUNK; ++k) vote_coded<MODE>(st, hist, tid, ccode[k]);
#pragma unroll
                for (int k = 0; k < F3D_CHUNK; ++k) ccode[k] = (cmasks + (size_t)cv[k] * plane)[coff[k]];
            }
#pragma unroll
            for (int k = 0; k < F3D_CHUNK; ++k) { vote_coded<MODE>(st, hist, tid, ccode[k]); ccode[k] = F3D_CODE_NONE; }
            // mixed views: per-point float32 cull; a lane inside the rounding margin of a plane is deferred
            todo_v = valid_m & ~out_m & ~in_m;
            while (todo_v) {
                const int bit = __builtin_ctzll(todo_v);
                todo_v &= todo_v - 1ull;
                const int v = 64 * g + bit;
                const f3d_view& vw = views[v];
                cull_consts cc = load_cull(vw);
                proj_consts pc = load_proj(vw);            // requested with the cull planes, not after the cull
                pin(cc); pin(pc);
                bool maybe, sure;
                cull_point32(cc, px32, py32, pz32, ps32, small, maybe, sure);
                bool inside = live & small & sure;
                const bool unc = live & small & maybe & !sure;
                if (__any(unc)) {                           // inside the float32 margin: decide with float64 FMAs
                    if (unc) {
                        bool m64, s64;
                        cull_point64(vw, p, pscale, m64, s64);
                        inside = s64;
                        defer = defer | (m64 & !s64);       // within rounding of the plane itself: the exact kernel decides
                    }
                }
                bool hit = false;
                int iu = 0, iv = 0;
                if (inside) {
                    bool unsure;
                    hit = project_fast(pc, ec1, eumax, p, W, H, iu, iv, unsure);
                    defer = defer | unsure;
                }
                vote_coded<MODE>(st, hist, tid, pend_code);
                pend_code = (cmasks + (size_t)v * plane)[hit ? mask_offset<true>(iu, iv, wt) : none_off];
            }
        }
        vote_coded<MODE>(st, hist, tid, pend_code);
        defer = defer | (live & ((st.best >> 16) >= 0xFFu));                               // an 8-bit bin overflowed (needs > 255 views)
        if (defer) todo[atomicAdd(todo_count, 1u)] = (int32_t)(gather_xyz ? orig : i);     // index into xyz as this launch sees it
        const bool bad = finish_coded<MODE, WRITE_VOTES>(st, hist, tid, flt, nclasses, threshold, live & !defer, orig, classes, votes_out);
        if (bad & !defer) atomicOr(err, F3D_DEVERR_FUSE);
    }
}

// k_fuse_exact: the reference's arithmetic, nothing else, for the points k_fuse deferred (and the whole path of the
// oracle in kernel form): exact 5-plane test, canonical projection with IEEE divisions, gather, vote, segment.
template <typename T, int MODE, bool WRITE_VOTES>
__global__ __launch_bounds__(F3D_BLOCK) void k_fuse_exact(const T* __restrict__ xyz, int64_t n_all, const unsigned int* __restrict__ todo_count,
                                                           const int32_t* __restrict__ todo,
                                                           const f3d_view* __restrict__ views, int nviews,
                                                           const uint8_t* __restrict__ masks, int H, int W,
                                                           int nclasses, f3d_filter_args flt, double threshold,
                                                           int64_t* __restrict__ classes, uint16_t* __restrict__ votes_out,
                                                           int* __restrict__ err, const int32_t* __restrict__ perm,
                                                           int gather_xyz) {
    using HT = hist_traits<MODE>;
    extern __shared__ uint32_t lds_u32[];
    uint32_t* hist = lds_u32;
    const int tid = threadIdx.x;
    const int ncols = nclasses + 1;
    const int words = (ncols + 1 + HT::per_word - 1) >> HT::shift;
    const size_t plane = (size_t)H * (size_t)W;
    const int64_t count = todo ? (int64_t)*todo_count : n_all;                  // todo == NULL: every point (no fast kernel ran)
    for (int64_t base = (int64_t)blockIdx.x * F3D_BLOCK; base < count; base += (int64_t)gridDim.x * F3D_BLOCK) {
        const int64_t k = base + tid;
        const bool live = k < count;
        const int64_t src = live ? (todo ? (int64_t)todo[k] : k) : 0;             // index into xyz as k_fuse saw it
        const int64_t orig = (live && perm && !gather_xyz) ? (int64_t)perm[src] : src;
        f3d_p3 p = {0.0, 0.0, 0.0};
        if (live) p = load_point(xyz, src);
        for (int wd = 0; wd < words; ++wd) hist[wd * F3D_BLOCK + tid] = 0u;
        vote_state<MODE> st;
        for (int v = 0; v < nviews; ++v) {
            const f3d_view& vw = views[v];
            bool hit = false;
            unsigned off = 0u;
            if (live && f3d_inside_view(vw, p)) {
                double fu, fv;
                project_exact(vw, p, fu, fv);
                if (fu >= 0.0 && fu < (double)W && fv >= 0.0 && fv < (double)H) {   // NaN compares false
                    hit = true; off = mask_offset<false>((int)fu, (int)fv, W);
                }
            }
            const int label = hit ? (int)(masks + (size_t)v * plane)[off] : 0;
            vote_add<MODE>(st, hist, tid, flt, nclasses, hit, label);
        }
        if (st.bad) atomicOr(err, F3D_DEVERR_FUSE);
        finish_point<MODE, WRITE_VOTES>(st, hist, tid, flt, nclasses, threshold, live, orig, classes, votes_out);
    }
}

// Audit of accelerators (B) and (C) against the exact arithmetic, every (point, view) pair: see f3d_kernels.h
template <typename T>
__global__ __launch_bounds__(F3D_BLOCK) void k_fastpath_audit(const T* __restrict__ xyz, int64_t n,
                                                               const f3d_view* __restrict__ views, int nviews,
                                                               unsigned long long* __restrict__ stats) {
    unsigned long long pairs = 0, fallback = 0, wrong = 0, cullwrong = 0;
    for (int64_t i = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * F3D_BLOCK) {
        const f3d_p3 p = load_point(xyz, i);
        const double pscale = (fabs(p.x) + fabs(p.y)) + fabs(p.z);
        const bool small = pscale < 1.0e30;
        for (int v = 0; v < nviews; ++v) {
            const f3d_view& vw = views[v];
            const bool in_exact = f3d_inside_view(vw, p);
            bool maybe, sure;
            cull_point32(load_cull(vw), (float)p.x, (float)p.y, (float)p.z, (float)pscale, small, maybe, sure);
            if ((sure && !in_exact) || (!maybe && in_exact)) ++cullwrong;
            if (!in_exact) continue;
            ++pairs;
            // audit with a 1024 x 1024 image, and with a tiny one so that the out-of-image rule is exercised as well
            double eu, ev;
            project_exact(vw, p, eu, ev);
            for (int dim = 1024; dim >= 16; dim >>= 6) {
                int iu, iv; bool unsure;
                const bool hit = project_fast(load_proj(vw), F3D_FAST_EPS * view_c1(vw, (double)dim), F3D_FAST_EPS * (double)dim, p, dim, dim, iu, iv, unsure);
                const bool ehit = (eu >= 0.0) & (eu < (double)dim) & (ev >= 0.0) & (ev < (double)dim);
                if (unsure) { if (dim == 1024) ++fallback; }
                else if (hit != ehit || (hit && !((double)iu == eu && (double)iv == ev))) ++wrong;
            }
        }
    }
    atomicAdd(&stats[0], pairs); atomicAdd(&stats[1], fallback); atomicAdd(&stats[2], wrong); atomicAdd(&stats[3], cullwrong);
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

template <bool VEC4>
__global__ __launch_bounds__(F3D_BLOCK) void k_sem_to_mask(const float* __restrict__ sem, int C, int64_t hw, float conf,
                                                            int low_label, uint8_t* __restrict__ mask) {
    constexpr int PX = VEC4 ? 4 : 1;
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

template <typename T>
__global__ __launch_bounds__(F3D_BLOCK) void k_points_in_obb(const T* __restrict__ xyz, int64_t n,
                                                              const f3d_obb* __restrict__ boxes, int B,
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
        // kernel LDS-bound (15 broadcast 8-B reads per point-box test against ~25 VALU instructions); two boxes are requested
        // and pinned together so that one scalar-memory round trip covers both
        for (int w0 = 0; w0 < B; w0 += 32) {
            uint32_t word = 0;
            const int lim = min(32, B - w0);
            for (int k = 0; k < lim; k += 2) {
                obb_consts bx[2];
                bx[0] = load_obb(boxes[w0 + k]);
                bx[1] = load_obb(boxes[w0 + min(k + 1, lim - 1)]);
                pin(bx[0]); pin(bx[1]);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const double d0 = p.x - bx[u].c[0], d1 = p.y - bx[u].c[1], d2 = p.z - bx[u].c[2];
                    bool in = live & (k + u < lim);
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const double pr = (d0 * bx[u].R[a] + d1 * bx[u].R[3 + a]) + d2 * bx[u].R[6 + a];
                        in = in & (fabs(pr) <= bx[u].e[a] / 2);
                    }
                    word |= in ? (1u << ((k + u) & 31)) : 0u;
                }
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
#ifndef F3D_FUSE_GRID
#define F3D_FUSE_GRID (256 * 4 * 8)     // k_fuse: 4 resident blocks per CU (LDS limit), 8 rounds so that the tail stays short
#endif

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

size_t f3d_fuse_lds_bytes(int mode, int nclasses) {
    const int ncols = nclasses + 1;
    const int per_word = (mode == MODE_HIST8) ? 4 : 2;
    const size_t hist = (size_t)((ncols + 2 + per_word - 1) / per_word) * F3D_BLOCK * sizeof(uint32_t);   // + codes NONE and BAD
    return 64 * F3D_CULL_ROW * sizeof(float) + hist;
}

int f3d_fuse_pick_mode(int nviews, int nfilter, bool want_votes) {
    // (a register-counter mode for <= 8 filter classes was tried: it needs more VGPRs than the LDS histogram, drops the kernel
    // to 3 waves per SIMD and measured 1.74 ms against 1.52 ms per C3 step)
    (void)nfilter; (void)want_votes;
    return nviews <= 255 ? MODE_HIST8 : MODE_HIST16;
}

hipError_t f3d_launch_fuse(const void* xyz, int dtype, int64_t n, const f3d_view* views_dev, int nviews,
                           const uint8_t* masks, const uint8_t* cmasks, int h, int w, int nclasses, const f3d_filter_args& flt,
                           double threshold, int64_t* classes, uint16_t* votes, int* err, const int32_t* perm, bool gather_xyz,
                           unsigned int* todo_count, int32_t* todo, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int mode = f3d_fuse_pick_mode(nviews, flt.nfilter, votes != nullptr);      // bins of the exact kernel; the fast one uses 8 bits
    const size_t lds = f3d_fuse_lds_bytes(MODE_HIST8, nclasses);
    const size_t lds_exact = f3d_fuse_lds_bytes(mode, nclasses) - 64 * F3D_CULL_ROW * sizeof(float);
    if (lds > 160 * 1024 || lds_exact > 160 * 1024) return hipErrorInvalidValue;
    const int64_t ntiles = (n + F3D_BLOCK - 1) / F3D_BLOCK;
    int grid = (int)(ntiles < F3D_FUSE_GRID ? ntiles : F3D_FUSE_GRID);
    grid = (grid + 7) & ~7;                                  // the XCD-aware tile mapping needs a multiple of 8 blocks
    const bool fast = cmasks != nullptr;                     // no coded masks (nclasses > F3D_CODE_MAX_NCLASSES): exact kernel only
    const dim3 g(grid), b(F3D_BLOCK), ge(fast ? 512 : grid);
    if (fast) {
        hipError_t e0 = hipMemsetAsync(todo_count, 0, sizeof(unsigned int), s);
        if (e0 != hipSuccess) return e0;
    }
#define F3D_FUSE(T, M, V)                                                                                      \
    do {                                                                                                       \
        if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)k_fuse<T, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return hipErrorInvalidValue;                                                                       \
        if (lds_exact > 48 * 1024 &&                                                                           \
            hipFuncSetAttribute((const void*)k_fuse_exact<T, M, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_exact) != hipSuccess) \
            return hipErrorInvalidValue;                                                                       \
        if (fast)                                                                                              \
            hipLaunchKernelGGL((k_fuse<T, V>), g, b, lds, s, (const T*)xyz, n, views_dev, nviews, cmasks, h, w, \
                               nclasses, flt, threshold, classes, votes, err, perm, gather_xyz ? 1 : 0, todo_count, todo); \
        hipLaunchKernelGGL((k_fuse_exact<T, M, V>), ge, b, lds_exact, s, (const T*)xyz, n, todo_count, fast ? todo : nullptr, \
                           views_dev, nviews, masks, h, w, nclasses, flt, threshold, classes, votes, err, perm, gather_xyz ? 1 : 0); \
    } while (0)
#define F3D_FUSE_T(T)                                                                                          \
    do {                                                                                                       \
        if (mode == MODE_HIST8) { if (votes) F3D_FUSE(T, MODE_HIST8, true); else F3D_FUSE(T, MODE_HIST8, false); } \
        else { if (votes) F3D_FUSE(T, MODE_HIST16, true); else F3D_FUSE(T, MODE_HIST16, false); }              \
    } while (0)
    if (dtype == F3D_F64) F3D_FUSE_T(double); else F3D_FUSE_T(float);
#undef F3D_FUSE_T
#undef F3D_FUSE
    return hipGetLastError();
}

size_t f3d_coded_masks_bytes(int nviews, int h, int w) { return (size_t)nviews * f3d_coded_plane(h, w); }

hipError_t f3d_launch_code_masks(const uint8_t* src, uint8_t* dst, int nviews, int h, int w, int nclasses, hipStream_t s) {
    if (nviews <= 0) return hipSuccess;
    if (nclasses < 0 || nclasses > F3D_CODE_MAX_NCLASSES || ((uintptr_t)dst & 7)) return hipErrorInvalidValue;
    const int64_t total = (int64_t)nviews * (int64_t)(f3d_coded_plane(h, w) / 8);
    const dim3 g(grid_for(total, F3D_BLOCK, F3D_GRID_CAP)), b(F3D_BLOCK);
    if (!(w & 7) && !((uintptr_t)src & 7)) hipLaunchKernelGGL(k_code_masks<true>, g, b, 0, s, src, dst, nviews, h, w, nclasses);
    else hipLaunchKernelGGL(k_code_masks<false>, g, b, 0, s, src, dst, nviews, h, w, nclasses);
    return hipGetLastError();
}

hipError_t f3d_launch_fastpath_audit(const void* xyz, int dtype, int64_t n, const f3d_view* views_dev, int nviews,
                                     unsigned long long* stats_dev, hipStream_t s) {
    hipError_t e = hipMemsetAsync(stats_dev, 0, 4 * sizeof(unsigned long long), s);
    if (e != hipSuccess || n <= 0) return e;
    const dim3 g(grid_for(n, F3D_BLOCK, F3D_GRID_CAP)), b(F3D_BLOCK);
    if (dtype == F3D_F64) hipLaunchKernelGGL(k_fastpath_audit<double>, g, b, 0, s, (const double*)xyz, n, views_dev, nviews, stats_dev);
    else hipLaunchKernelGGL(k_fastpath_audit<float>, g, b, 0, s, (const float*)xyz, n, views_dev, nviews, stats_dev);
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

hipError_t f3d_launch_sem_to_mask(const float* sem, int c, int64_t hw, float conf, int low_label, uint8_t* mask, hipStream_t s) {
    if (hw <= 0) return hipSuccess;
    const bool vec4 = (hw % 4 == 0) && ((reinterpret_cast<uintptr_t>(sem) & 15) == 0) && ((reinterpret_cast<uintptr_t>(mask) & 3) == 0);
    const dim3 b(F3D_BLOCK);
    if (vec4) hipLaunchKernelGGL(k_sem_to_mask<true>, dim3(grid_for(hw / 4, F3D_BLOCK, F3D_GRID_CAP)), b, 0, s, sem, c, hw, conf, low_label, mask);
    else hipLaunchKernelGGL(k_sem_to_mask<false>, dim3(grid_for(hw, F3D_BLOCK, F3D_GRID_CAP)), b, 0, s, sem, c, hw, conf, low_label, mask);
    return hipGetLastError();
}

hipError_t f3d_launch_points_in_obb(const void* xyz, int dtype, int64_t n, const f3d_obb* boxes_dev, int b, uint32_t* bits,
                                    uint8_t* cooc, hipStream_t s) {
    if (n <= 0 || b <= 0) return hipSuccess;
    if (b > F3D_OBB_MAX_BOXES) return hipErrorInvalidValue;
    if (cooc) {
        hipError_t e = hipMemsetAsync(cooc, 0, (size_t)b * b, s);
        if (e != hipSuccess) return e;
    }
    const size_t lds = (size_t)((b + 31) / 32) * F3D_BLOCK * sizeof(uint32_t);
    const dim3 g(grid_for(n, F3D_BLOCK, F3D_GRID_CAP)), blk(F3D_BLOCK);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (dtype == F3D_F64) {
        if (lds > 48 * 1024) {                             // beyond the default dynamic-LDS limit: the launch needs the attribute
            hipError_t e = hipFuncSetAttribute((const void*)k_points_in_obb<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_points_in_obb<double>, g, blk, lds, s, (const double*)xyz, n, boxes_dev, b, bits, cooc);
    } else {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)k_points_in_obb<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_points_in_obb<float>, g, blk, lds, s, (const float*)xyz, n, boxes_dev, b, bits, cooc);
    }
    return hipGetLastError();
}

hipError_t f3d_launch_relabel(int64_t* ids, int64_t n, int64_t from, int64_t to, unsigned long long* count, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_relabel, dim3(grid_for(n, F3D_BLOCK, F3D_GRID_CAP)), dim3(F3D_BLOCK), 0, s, ids, n, from, to, count);
    return hipGetLastError();
}
