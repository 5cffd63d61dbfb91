// The fused multi-view path of libf3d_hip.so for gfx950 (MI355X, CDNA4; wave = 64 lanes):
//   project -> sample -> vote -> segment   (fusion.py:254-266 + voting.py:94-137 composed per point)
//
//   k_mask_presence / k_code_lut / k_code_masks : the masks are rewritten once per call into 8x8-pixel tiles of vote-BIN
//       CODES.  Only labels that occur in the masks (or, with filter_classes, only the filter labels) get a bin, so a
//       thread's vote histogram is a handful of LDS dwords instead of 34.
//   k_fuse        : the fast kernel.  One lane owns two points; all per-view arithmetic is FLOAT32: one lane per view
//       projects the CENTRE of the wave's bounding box in float64, every lane adds the float32 offset term of its own
//       points (|p - c| is a few centimetres after the cell sort, so float32 carries the pixel to ~1e-5 px) and accepts
//       the pixel only when it is farther than a rigorous bound from a pixel border.  A point with a decision that
//       cannot be proven is appended to a list -- with the votes of its proven views parked and a mask of the open ones --
//   k_fuse_mid    : float64 on the open (point, view) pairs of that list; a pair it cannot prove either is decided on the spot by the
//       reference's arithmetic and nothing else (exact 5-plane test, un-normalised quaternion sandwich, K @ c, IEEE divisions).
//   k_fuse_exact  : that arithmetic as a kernel of its own over the raw masks: the whole path when no coded masks exist, and the
//       points whose 8-bit vote bin wrapped (more than 255 views).
// Results are exactly those of the reference arithmetic (oracle order).  No MFMA: nothing here is a dense contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "f3d.h"
#include "f3d_math.h"
#include "f3d_kernels.h"
#include <cstdlib>

#pragma clang fp contract(off)

#define F3D_BLOCK 256
#define F3D_NO_PREFILL ((int64_t)0x7fffffffffffffffLL)   // "the label vector was not pre-filled": every label is stored
#define F3D_PART_MAX_GROUPS 16            // view groups whose open-view masks a parked point carries (more: the point is redone from nothing)
#ifndef F3D_MID_DENSE
#define F3D_MID_DENSE 8                  // k_fuse_mid: more open views per point than this on a block's average -> the views run in the outer loop
#endif
#ifndef F3D_NT_CLASSES
#define F3D_NT_CLASSES 1                 // the (scattered) label stores carry the non-temporal hint (measured: -2.5 % of the step)
#endif
#if defined(F3D_EXP_STORE) && F3D_EXP_STORE == 2          // timing experiments (results wrong): no label store at all
#define F3D_STORE_CLASS(p, v) ((void)(v))
#elif defined(F3D_EXP_STORE) && F3D_EXP_STORE == 3        // ... a one-byte label at the same (scattered) index
#define F3D_STORE_CLASS(p, v) (reinterpret_cast<uint8_t*>(classes)[(p) - classes] = (uint8_t)(v))
#elif defined(F3D_EXP_STORE) && F3D_EXP_STORE == 4        // ... one byte, non-temporal
#define F3D_STORE_CLASS(p, v) __builtin_nontemporal_store((uint8_t)(v), reinterpret_cast<uint8_t*>(classes) + ((p) - classes))
#elif F3D_NT_CLASSES
#define F3D_STORE_CLASS(p, v) __builtin_nontemporal_store((int64_t)(v), (p))
#else
#define F3D_STORE_CLASS(p, v) (*(p) = (v))
#endif

namespace {

template <typename T>
__device__ __forceinline__ f3d_p3 load_point(const T* __restrict__ xyz, int64_t i) {
    const T* p = xyz + 3 * i;
    f3d_p3 r;
    r.x = (double)p[0]; r.y = (double)p[1]; r.z = (double)p[2];
    return r;
}
enum { MODE_HIST8 = 0, MODE_HIST16 = 1 };

// k-th entry of filter_classes: short lists travel in the kernarg, long ones in device memory
__device__ __forceinline__ int filter_at(const f3d_filter_args& flt, int k) {
    return (flt.nfilter <= 8) ? flt.cls[k & 7] : flt.cls_dev[k];
}

template <int MODE>
struct hist_traits;
template <> struct hist_traits<MODE_HIST8> { static constexpr int per_word = 4, shift = 2, bits = 8; static constexpr uint32_t mask = 0xFFu; };
template <> struct hist_traits<MODE_HIST16> { static constexpr int per_word = 2, shift = 1, bits = 16; static constexpr uint32_t mask = 0xFFFFu; };

#ifndef F3D_GATHER_DEPTH
#define F3D_GATHER_DEPTH 1                   // chunks of gathers in flight per wave (2: measured below)
#endif
#ifndef F3D_CHUNK
#define F3D_CHUNK 2                          // whole-wave views projected per gather batch
#endif
#define F3D_CULL_ROW 25                       // floats per view in the LDS cull table (24 used, odd stride = no bank conflicts)
#define F3D_FAST_EPS 1.1368683772161603e-13   // 2^-43
#define F3D_U24 5.9604644775390625e-08        // 2^-24

// float32 cull planes of one view, copied by value (wave-uniform -> scalar loads -> SGPRs)
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
// pin(): the copies exist in SGPRs at this point of the program.  Without it the compiler sinks each scalar load next to
// its first use and the scalar-memory latency is paid once per use; with all loads of an iteration requested first and
// pinned together it is paid once.
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
__device__ __forceinline__ unsigned mask_offset(int iu, int iv, int W) {          // TILED: W = tiles per row of the coded plane (border included)
    if (TILED) return (((unsigned)((iv + 8) >> 3) * (unsigned)W + (unsigned)((iu + 8) >> 3)) << 6) | ((unsigned)(iv & 7) << 3) | (unsigned)(iu & 7);
    return (unsigned)(iv * W + iu);
}

// ------------------------------------------------------------------------------------------
// Vote-bin codes.  Code 0 = "no sample" (k_code_masks ends every view with 64 such bytes, and a lane without a pixel
// gathers from there instead of carrying a validity flag through the vote), code 1 = a label the reference would
// reject (> nclasses, IndexError at voting.py:98), then
//   presence book (no filter, or votes requested): one code per label that occurs in the masks, the SMALLEST label
//       getting the LARGEST code, so that the maximum of (count << 8 | code) is count desc, label asc = np.argmax's
//       first-maximum rule;
//   filter book: code 2 = any other valid label (it only counts towards the total, voting.py:120), codes 3.. = the
//       distinct entries of filter_classes.
// The book is built on the device (k_mask_presence + k_code_lut), so no host round trip decides the histogram size:
// the launcher enqueues a SMALL-histogram and a FULL-histogram instance of k_fuse and each returns at once unless the
// book's size is in its range.
// ------------------------------------------------------------------------------------------
#define F3D_CODE_NONE 0u
#define F3D_CODE_BAD 1u
#define F3D_CODE_OTHER 2u
#define F3D_PACKED_SMALL_WORDS 12             // packed 8-bit bins: alphabets up to 48 codes get the medium-LDS instance
#ifndef F3D_PACKED_LARGE_WORDS
#define F3D_PACKED_LARGE_WORDS 25            // ... up to 100 codes two points per lane with the view tables in global memory (3 blocks per CU)
#endif
#define F3D_BOOK_DWORDS 256                  // lut + inv + cmin of f3d_codebook, staged in LDS by k_fuse
#define F3D_BIN32_MAX_CODES 12               // alphabets up to this many codes vote into dword bins (12 KiB of LDS per 256 points: 4 blocks per CU)

// One view of the coded masks: (ceil(H/8) + 2) x (ceil(W/8) + 2) tiles of 8x8 pixels (64 B each): the image plus a one-tile border of
// "no sample" all around, so that pixel (-1, -1) .. (W, H) are addressable -- tile (0, 0) of view 0, at offset 0, is such a border tile
// and serves as the address a lane without a pixel gathers from.
__host__ __device__ inline int f3d_coded_pitch(int W) { return ((W + 7) >> 3) + 2; }
__host__ __device__ inline size_t f3d_coded_plane(int H, int W) { return (size_t)(((H + 7) >> 3) + 2) * (size_t)f3d_coded_pitch(W) * 64; }

// which labels occur in the masks.  A thread keeps a 256-bit set in four 64-bit registers; masks are piecewise constant,
// so an 8-byte word whose bytes all equal the previous label costs two instructions.
template <bool VEC>
__global__ __launch_bounds__(F3D_BLOCK) void k_mask_presence(const uint8_t* __restrict__ src, int64_t nbytes, f3d_codebook* __restrict__ cb) {
    unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    auto mark = [&](unsigned l) {
        const unsigned long long bit = 1ull << (l & 63u);
        const unsigned q = l >> 6;
        m0 |= q == 0u ? bit : 0ull; m1 |= q == 1u ? bit : 0ull; m2 |= q == 2u ? bit : 0ull; m3 |= q == 3u ? bit : 0ull;
    };
    if (VEC) {
        const int64_t nw = nbytes >> 3;
        const uint64_t* w = reinterpret_cast<const uint64_t*>(src);
        uint64_t last = ~0ull;                                     // the previous word when all of its bytes were equal
        for (int64_t k = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; k < nw; k += (int64_t)gridDim.x * F3D_BLOCK) {
            const uint64_t x = w[k];
            if (x == last) continue;
            if (x == (x & 0xFFull) * 0x0101010101010101ull) { mark((unsigned)(x & 0xFFull)); last = x; continue; }
#pragma unroll
            for (int c = 0; c < 8; ++c) mark((unsigned)(x >> (8 * c)) & 0xFFu);
        }
        if (blockIdx.x == 0 && threadIdx.x < (nbytes & 7)) mark(src[(nw << 3) + threadIdx.x]);
    } else {
        for (int64_t k = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; k < nbytes; k += (int64_t)gridDim.x * F3D_BLOCK) mark(src[k]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        m0 |= __shfl_xor(m0, off, 64); m1 |= __shfl_xor(m1, off, 64); m2 |= __shfl_xor(m2, off, 64); m3 |= __shfl_xor(m3, off, 64);
    }
    // block-level OR in LDS, then one thread per dword adds only the bits the global set still lacks: thousands of waves
    // hammering the same 8 dwords with atomics cost 0.38 ms; masks share their alphabet, so almost every block finds its bits set
    __shared__ unsigned blk[8];
    if (threadIdx.x < 8) blk[threadIdx.x] = 0u;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long m[4] = {m0, m1, m2, m3};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if ((unsigned)m[q]) atomicOr(&blk[2 * q], (unsigned)m[q]);
            if ((unsigned)(m[q] >> 32)) atomicOr(&blk[2 * q + 1], (unsigned)(m[q] >> 32));
        }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const unsigned want = blk[threadIdx.x];
        const unsigned have = __hip_atomic_load(&cb->presence[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (want & ~have) atomicOr(&cb->presence[threadIdx.x], want);
    }
}

// the presence set as 256 bytes of 0 / 1 (to_bytes) or back (one block of 256 threads)
__global__ __launch_bounds__(F3D_BLOCK) void k_presence_bytes(f3d_codebook* __restrict__ cb, uint8_t* __restrict__ bytes256, int to_bytes) {
    const unsigned l = threadIdx.x;
    if (to_bytes) {
        bytes256[l] = (uint8_t)((cb->presence[l >> 5] >> (l & 31u)) & 1u);
        __syncthreads();
        if (l < 8) cb->presence[l] = 0u;                          // consumed (see f3d_launch_mask_presence)
        return;
    }
    const unsigned long long m = __ballot(bytes256[l] != 0);
    if ((l & 63u) == 0u) { cb->presence[l >> 5] = (unsigned)m; cb->presence[(l >> 5) + 1] = (unsigned)(m >> 32); }
}

// cb->cmin for this call's threshold (see segment_point): thread t finds, by bisection with the reference's own float64 division,
// how many c in 0..t give c / t < threshold.  One block of 256 threads.
__device__ __forceinline__ void threshold_table(f3d_codebook* __restrict__ cb, double threshold) {
    const int t = threadIdx.x;
    int lo = 0, hi = t + 1;                                   // c in [lo, hi): the first c that is NOT below the threshold
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((double)mid / (double)t < threshold) lo = mid + 1; else hi = mid;
    }
    cb->cmin[t] = (uint16_t)lo;                               // t = 0 is never looked up (no votes: voting.py:126)
}

// one block of 256 threads: thread l decides the code of label l.  book: 0 = every label 0..nclasses has a bin (no
// presence pass ran), 1 = presence book, 2 = filter book.
__device__ __forceinline__ void code_lut_block(f3d_codebook* __restrict__ cb, int nclasses, int book, const f3d_filter_args& flt) {
    __shared__ unsigned pres[8];
    __shared__ int first_of[256];                                  // filter book: position of label l's first occurrence, -1 = not listed
    const unsigned l = threadIdx.x;
    if (l < 8) { pres[l] = book == 1 ? cb->presence[l] : 0xFFFFFFFFu; cb->presence[l] = 0u; }   // consumed (see f3d_launch_mask_presence)
    first_of[l] = -1;
    cb->inv[l] = 0;
    __syncthreads();
    if (book == 2) {
        if (l == 0)
            for (int k = flt.nfilter - 1; k >= 0; --k) { const int f = filter_at(flt, k); if (f >= 0 && f < 256) first_of[f] = k; }
        __syncthreads();
        const bool listed = first_of[l] >= 0 && (int)l <= nclasses;
        int rank = 0;                                              // distinct listed labels below l
        for (unsigned j = 0; j < l; ++j) rank += (first_of[j] >= 0 && (int)j <= nclasses) ? 1 : 0;
        int distinct = 0;
        for (unsigned j = 0; j < 256; ++j) distinct += (first_of[j] >= 0 && (int)j <= nclasses) ? 1 : 0;
        const unsigned code = (int)l > nclasses ? F3D_CODE_BAD : (listed ? 3u + (unsigned)rank : F3D_CODE_OTHER);
        cb->lut[l] = (uint8_t)code;
        if (listed) cb->inv[code] = (uint8_t)l;
        if (l == 0) { cb->ncodes = 3 + distinct; cb->words = (3 + distinct + 3) >> 2; cb->book = 2; }
        return;
    }
    const bool pv = (int)l <= nclasses && ((pres[l >> 5] >> (l & 31u)) & 1u);     // label l is valid and present
    __shared__ unsigned long long wb[4];
    const unsigned long long bal = __ballot(pv);
    if ((l & 63u) == 0u) wb[l >> 6] = bal;
    __syncthreads();
    int K = 0, below = __popcll(bal & ((1ull << (l & 63u)) - 1ull));
    for (unsigned w = 0; w < 4; ++w) { const int c = __popcll(wb[w]); K += c; below += (w < (l >> 6)) ? c : 0; }
    unsigned code;
    if ((int)l > nclasses) code = F3D_CODE_BAD;                    // only meaningful when such a byte really occurs
    else if (pv) code = (unsigned)(K + 1 - below);                 // smallest present label -> K + 1, largest -> 2
    else code = F3D_CODE_NONE;                                     // never gathered; its vote row reads the always-zero bin 0
    cb->lut[l] = (uint8_t)code;
    if (code >= 2u) cb->inv[code] = (uint8_t)l;
    if (l == 0) { cb->ncodes = K + 2; cb->words = (K + 2 + 3) >> 2; cb->book = book; }
}

__global__ __launch_bounds__(F3D_BLOCK) void k_code_lut(f3d_codebook* __restrict__ cb, int nclasses, int book, f3d_filter_args flt) {
    code_lut_block(cb, nclasses, book, flt);
}

// [V,H,W] row-major labels -> [V][ceil(H/8) + 2][ceil(W/8) + 2][8][8] bin codes (border tiles = "no sample"); one thread moves one
// tile row (8 bytes), a wave writes 512 contiguous bytes.  VEC: W % 8 == 0 and src 8-B aligned (one 8-B load per thread).
template <bool VEC>
__global__ __launch_bounds__(F3D_BLOCK) void k_code_masks(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int V, int H, int W,
                                                           const f3d_codebook* __restrict__ cb) {
    __shared__ uint32_t lut_w[64];
    if (threadIdx.x < 64) lut_w[threadIdx.x] = reinterpret_cast<const uint32_t*>(cb->lut)[threadIdx.x];
    __syncthreads();
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(lut_w);
    const int tw = f3d_coded_pitch(W), th = ((H + 7) >> 3) + 2;    // border included
    const int64_t per_view = (int64_t)th * tw * 8;                 // 8-byte pieces per view
    const int64_t total = per_view * V;
    const size_t tplane = f3d_coded_plane(H, W);
    for (int64_t k = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; k < total; k += (int64_t)gridDim.x * F3D_BLOCK) {
        const int64_t v = k / per_view; const int64_t r = k - v * per_view;      // r indexes (tile, row-in-tile) of the destination
        uint64_t out = 0;                                                         // border and padding: F3D_CODE_NONE
        const int tile = (int)(r >> 3), ry = (int)(r & 7);
        const int ty = tile / tw, tx = tile - ty * tw;
        const int y = (ty - 1) * 8 + ry, x0 = (tx - 1) * 8;
        if (y >= 0 && y < H && x0 >= 0 && x0 < W) {
            const uint8_t* row = src + (size_t)v * H * W + (size_t)y * W + x0;
            if (VEC) {
                const uint64_t x = *reinterpret_cast<const uint64_t*>(row);
                if (x == (x & 0xFFull) * 0x0101010101010101ull) out = (uint64_t)lut[x & 0xFFull] * 0x0101010101010101ull;
                else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) out |= (uint64_t)lut[(x >> (8 * c)) & 0xFFull] << (8 * c);
                }
            } else {
                for (int c = 0; c < 8; ++c) if (x0 + c < W) out |= (uint64_t)lut[row[c]] << (8 * c);
            }
        }
        *reinterpret_cast<uint64_t*>(dst + (size_t)v * tplane + (size_t)r * 8) = out;
    }
}

// ------------------------------------------------------------------------------------------
// Centre + offset projection.  For a wave whose 64 points lie in the box c +- E (float32 centre, half extents) and a
// view with operator M = K Rot(qinv) and translation t, the view's lane projects the CENTRE in float64:
//     h = M (c - t),   z_c = h_2,   u_c = h_0 / h_2 = U + uf   (U integer, uf in [0, 1)),   likewise v_c = V + vf.
// With d = p - c the pixel coordinate of a point of the wave is, exactly,
//     u(p) - U = uf + (sum_j a0_j d_j) / (1 + sum_j b_j d_j),     a0_j = (M_0j - u_c M_2j) / z_c,   b_j = M_2j / z_c
// (and a1_j = (M_1j - v_c M_2j) / z_c for v).  The lane rounds a0, a1, b, uf, vf to float32 ONCE per (wave, view); every
// lane then evaluates  w = uf + (a0 . d) * rcp(1 + b . d)  in float32 -- |d| is a few centimetres after the cell sort, so
// w is a small number carried to ~1e-5 px -- and the pixel U + floor(w) is accepted when w is farther than the bound
// below from an integer.  The 14 floats of a row reach the other lanes through v_readlane when the view's turn comes:
// the view loop touches neither scalar memory nor LDS for its constants.
// Error budget, u = 2^-24, per coordinate, with X = sum_j |a_j| E_j (>= |a . d| over the box), Q = sum_j |b_j| E_j < 1/2,
// smax = 1 / (1 - Q) (>= rcp(1 + b . d)), D = X smax (>= |w - uf|):
//   a . d : a and (float)(p - c) carry u each, the 3-FMA chain 3 more            -> 5.01 u X
//   1 + q : the same for b . d (5.01 u Q), the add (u (1 + Q)), v_rcp_f32 (1 ulp)  -> relative u (1 + (1 + 6.01 Q) smax)
//   w     : the final FMA (u (D + 8)) and (float)uf (4 u: uf is taken from a tile border, 0 <= uf < 8)
//   B32 = 1.5 u (D (7.01 + (1 + 6.01 Q) smax) + 12.5)                              (1.5: safety factor)
// On top of it twice the float64 bound 2^-43 (|d|_1 |r| (mnorm_k + |u| mnorm_2) + |u|) of "M (p - t) in FMAs against the
// canonical operation order" (both are within ~50 eps of the real number; 2^-43 is a > 10x margin), once for the centre
// the row is derived from and once for the point, evaluated over the box.
// A view whose box comes close to the camera plane (Q >= 1/2), is huge, or projects absurdly far gets no row (ok = false):
// its visible points go to the next tier.
// ------------------------------------------------------------------------------------------
// uf, vf: u_c - U8, v_c - V8 in [0, 8) with U8 = floor(u_c) rounded down to a multiple of 8 (a tile border), so that the integer part of
// a pixel only enters the tile arithmetic through obase = the offset of pixel (U8, V8) inside the view's coded plane; U8, V8 themselves are
// only needed for the image-range test of the mixed views.  hb: half width of the accepted fraction band.
struct centre_row { float b0, b1, b2, a00, a01, a02, a10, a11, a12, uf, vf, hb; unsigned obase; int U8, V8; };

__device__ __forceinline__ double rcp_newton(double x) {            // 1/x to ~1 ulp without the IEEE division's register appetite
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}

// vt: the view's first 15 doubles (M[9], t[3], mnorm[3] -- the head of f3d_view) with stride `vs` between them: straight from the
// record (vs = 1) or from the block's LDS copy laid out [field][view] (vs = 64; lanes-over-views reads are then conflict-free
// instead of 64 scattered cache lines per load)
#define F3D_VHEAD 15
#ifndef F3D_VT
#define F3D_VT                           // (volatile was tried to stop the hoisting of these tile-invariant reads: more spills, not fewer)
#endif
__device__ __forceinline__ bool centre_precompute(const F3D_VT double* vt, int vs, float c0, float c1, float c2, float e0, float e1, float e2,
                                                   float umaxf, int pitch, centre_row& row) {
    const double M0 = vt[0], M1 = vt[vs], M2 = vt[2 * vs], M3 = vt[3 * vs], M4 = vt[4 * vs], M5 = vt[5 * vs], M6 = vt[6 * vs], M7 = vt[7 * vs],
                 M8 = vt[8 * vs];
    const double d0 = (double)c0 - vt[9 * vs], d1 = (double)c1 - vt[10 * vs], d2 = (double)c2 - vt[11 * vs];
    const double h0 = __builtin_fma(M0, d0, __builtin_fma(M1, d1, M2 * d2));
    const double h1 = __builtin_fma(M3, d0, __builtin_fma(M4, d1, M5 * d2));
    const double h2 = __builtin_fma(M6, d0, __builtin_fma(M7, d1, M8 * d2));
    const double rc = rcp_newton(h2);
    const double uc = h0 * rc, vc = h1 * rc;
    const double U = floor(uc), V = floor(vc);
    row.b0 = (float)(M6 * rc); row.b1 = (float)(M7 * rc); row.b2 = (float)(M8 * rc);
    row.a00 = (float)(__builtin_fma(-uc, M6, M0) * rc); row.a01 = (float)(__builtin_fma(-uc, M7, M1) * rc); row.a02 = (float)(__builtin_fma(-uc, M8, M2) * rc);
    row.a10 = (float)(__builtin_fma(-vc, M6, M3) * rc); row.a11 = (float)(__builtin_fma(-vc, M7, M4) * rc); row.a12 = (float)(__builtin_fma(-vc, M8, M5) * rc);
    const double U8 = floor(uc * 0.125) * 8.0, V8 = floor(vc * 0.125) * 8.0;
    row.uf = (float)(uc - U8); row.vf = (float)(vc - V8);
    row.U8 = (int)U8; row.V8 = (int)V8;                                            // |U|, |V| < 32768 is required below
    row.obase = (unsigned)(((row.V8 + 8) >> 3) * pitch + ((row.U8 + 8) >> 3)) << 6;
    // the bounds in float32, every step rounded up by the factor k (they only have to be upper bounds); the float32 row entries
    // are within 2^-24 of the real a, b, which k covers as well
    const float k = 1.00001f;
    const float E0 = (e0 + 1.2e-7f * (fabsf(c0) + e0)) * k, E1 = (e1 + 1.2e-7f * (fabsf(c1) + e1)) * k,
                E2 = (e2 + 1.2e-7f * (fabsf(c2) + e2)) * k;                       // the box is of float32-rounded coordinates
    const float X0 = (fabsf(row.a00) * E0 + fabsf(row.a01) * E1 + fabsf(row.a02) * E2) * k;
    const float X1 = (fabsf(row.a10) * E0 + fabsf(row.a11) * E1 + fabsf(row.a12) * E2) * k;
    const float Q = (fabsf(row.b0) * E0 + fabsf(row.b1) * E1 + fabsf(row.b2) * E2) * k;
    const float smax = k * k / (1.0f - Q);                                         // float32 division: 2.5 ulp, covered by k * k
    const float D = fmaxf(X0, X1) * smax * k;
    const float B32 = 1.5f * (float)F3D_U24 * (D * (7.01f + (1.0f + 6.01f * Q) * smax) + 12.5f) * k;   // |w| <= D + 8, (float)uf carries 4 u
    const float arc = fabsf((float)rc) * k;
    const float d1max = ((fabsf((float)d0) + fabsf((float)d1)) + fabsf((float)d2) + ((E0 + E1) + E2)) * k;
    const float c1b = (2.0f * umaxf * (float)vt[14 * vs] + fmaxf((float)vt[12 * vs], (float)vt[13 * vs])) * k;
    const float b64 = (float)F3D_FAST_EPS * (d1max * arc * smax * c1b + 2.0f * umaxf) * k;   // |r| <= |rc| smax over the box
    const float hb = (0.5f - (B32 + 2.0f * b64) * k - 4.0f * (float)F3D_U24) * 0.99999f;
    row.hb = hb;
    // every comparison is false for NaN: a degenerate box or view simply is not taken this way
    return (h2 > 0.0) && (Q < 0.5f) && (B32 < 1.0e-3f) && (hb > 0.25f) && (fabs(U) < 32768.0) && (fabs(V) < 32768.0) &&
           (fabs(uc) <= 2.0 * (double)umaxf) && (fabs(vc) <= 2.0 * (double)umaxf);
}

// the row of view `bit` of the current 64-view group, broadcast out of the lane that computed it (v_readlane -> SGPRs)
__device__ __forceinline__ centre_row read_row(const centre_row& mine, int bit) {
    centre_row r;
#define F3D_RL(f) r.f = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.f), bit))
    F3D_RL(b0); F3D_RL(b1); F3D_RL(b2); F3D_RL(a00); F3D_RL(a01); F3D_RL(a02); F3D_RL(a10); F3D_RL(a11); F3D_RL(a12);
    F3D_RL(uf); F3D_RL(vf); F3D_RL(hb);
#undef F3D_RL
    r.obase = (unsigned)__builtin_amdgcn_readlane((int)mine.obase, bit);
    r.U8 = 0; r.V8 = 0;                                            // (read separately where the range test needs them)
    return r;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float x) { return (f32x2){x, x}; }
#define F3D_FMA2(a, b, c) __builtin_elementwise_fma((a), (b), (c))

// The two points of a lane against one row, in packed float32 (v_pk_fma_f32: one issue slot for both points).  fi0, fi1 = the pixel
// relative to the row's tile border (U8, V8) -- small integers, possibly negative; safe = "this pixel is proven".
__device__ __forceinline__ void project_offset2(const centre_row& r, f32x2 DX, f32x2 DY, f32x2 DZ, int fi0[2], int fi1[2], bool safe[2]) {
    const f32x2 q = F3D_FMA2(splat2(r.b0), DX, F3D_FMA2(splat2(r.b1), DY, F3D_FMA2(splat2(r.b2), DZ, splat2(1.0f))));
    const f32x2 x0 = F3D_FMA2(splat2(r.a00), DX, F3D_FMA2(splat2(r.a01), DY, splat2(r.a02) * DZ));
    const f32x2 x1 = F3D_FMA2(splat2(r.a10), DX, F3D_FMA2(splat2(r.a11), DY, splat2(r.a12) * DZ));
    const f32x2 sr = {__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
    const f32x2 w0 = F3D_FMA2(x0, sr, splat2(r.uf)), w1 = F3D_FMA2(x1, sr, splat2(r.vf));
    const f32x2 f0 = {__builtin_floorf(w0.x), __builtin_floorf(w0.y)}, f1 = {__builtin_floorf(w1.x), __builtin_floorf(w1.y)};
    const f32x2 g0 = (w0 - f0) - splat2(0.5f), g1 = (w1 - f1) - splat2(0.5f);
    safe[0] = __builtin_fmaxf(__builtin_fabsf(g0.x), __builtin_fabsf(g1.x)) < r.hb;   // NaN -> false (fmax drops one NaN, not two:
    safe[1] = __builtin_fmaxf(__builtin_fabsf(g0.y), __builtin_fabsf(g1.y)) < r.hb;   //  a NaN pixel has both coordinates NaN -- q is shared)
    fi0[0] = (int)f0.x; fi0[1] = (int)f0.y; fi1[0] = (int)f1.x; fi1[1] = (int)f1.y;
}

// offset of pixel (U8 + fi0, V8 + fi1) in the view's coded plane = obase + this; c_row = 64 * pitch - 64.
// (fi + 56 (fi >> 3) = (fi & 7) + 64 (fi >> 3), and 8 fi + (64 pitch - 64)(fi >> 3) = 8 (fi & 7) + 64 pitch (fi >> 3); arithmetic shifts.)
__device__ __forceinline__ unsigned rel_offset(int fi0, int fi1, int c_row) {
    // 24-bit multiplies (v_mad_i32_i24, full rate; a 32-bit v_mul_lo_u32 costs four issue slots): |fi| >> 3 and c_row are far below 2^23
    return (unsigned)((fi0 + __mul24(fi0 >> 3, 56)) + ((fi1 << 3) + __mul24(fi1 >> 3, c_row)));
}

// single point (audit kernel): absolute pixel and "proven"
__device__ __forceinline__ bool project_offset(const centre_row& r, float dx, float dy, float dz, int& iu, int& iv) {
    int fi0[2], fi1[2]; bool safe[2];
    project_offset2(r, splat2(dx), splat2(dy), splat2(dz), fi0, fi1, safe);
    iu = fi0[0] + r.U8; iv = fi1[0] + r.V8;
    return safe[0];
}

__device__ __forceinline__ void project_exact(const f3d_view& vw, f3d_p3 p, double& fu, double& fv) {
    const f3d_p3 h = f3d_project_h(vw.K, vw.qinv, vw.t, p);
    fu = floor(h.x / h.z); fv = floor(h.y / h.z);
}

// ---- vote state of the exact kernel (k_fuse_exact): bins indexed by the label itself
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

// VotingSegmentation.segment (voting.py:120-135) for one point given the winner, then the store
// cmin (optional, totals up to 255 only): cmin[t] = the number of c in 0..t with (double)c / (double)t < threshold -- the float64
// division of voting.py:128 is monotone in c, so "max / total < threshold" is "max < cmin[total]" (k_threshold_table builds it with
// that very division; a table look-up instead of a ~40-instruction IEEE division per point)
template <typename FilterAt>
__device__ __forceinline__ int64_t segment_point(int win_c, int win_i, int total, int nfilter, FilterAt fat, int nclasses, double threshold,
                                                 const uint16_t* cmin = nullptr) {
    int64_t cls;
    if (total == 0) cls = nclasses;                                                // :126
    else {
        cls = win_i;
        const bool low = cmin ? win_c < (int)cmin[total] : (double)win_c / (double)total < threshold;
        if (low) cls = nclasses;                                                   // :128-130
        if (win_c == 0) cls = nclasses;                                            // :131
    }
    if (nfilter > 0) {                                                             // sequential remap (Q3)
        int64_t r = cls;
        for (int k = 0; k < nfilter; ++k) if (r == k) r = fat(k);
        cls = r;
    }
    return cls;
}

template <int MODE, bool WRITE_VOTES>
__device__ __forceinline__ void finish_point(const vote_state<MODE>& st, const uint32_t* hist, int tid, const f3d_filter_args& flt,
                                             int nclasses, double threshold, bool store, int64_t orig,
                                             int64_t* __restrict__ classes, uint16_t* __restrict__ votes_out) {
    using HT = hist_traits<MODE>;
    const int ncols = nclasses + 1;
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
    const int64_t cls = segment_point(win_c, win_i, st.total, flt.nfilter, [&](int k) { return filter_at(flt, k); }, nclasses, threshold);
    if (store) F3D_STORE_CLASS(&classes[orig], cls);
    if (WRITE_VOTES && store) {
        for (int l = 0; l < ncols; ++l)
            votes_out[(size_t)orig * ncols + l] =
                (uint16_t)((hist[(l >> HT::shift) * F3D_BLOCK + tid] >> ((l & (HT::per_word - 1)) * HT::bits)) & HT::mask);
    }
}

// ---- the fast kernel's vote: `b` is a bin code gathered from the coded masks; hcol = the point's histogram column (bins of
// word w at hcol[w * F3D_BLOCK]).  No validity flag, no label range test, no returned value to wait for (ds_add_u32): "no
// sample" lands in bin 0, which nothing reads; the plurality is found by one scan over the few bins at the end.
// WRAP (more than 255 views): an 8-bit bin may wrap, so bin 0 must stay 0 and the votes are counted (see finish_coded).
// dword bins (small alphabets): bin of code b at hcol[b * F3D_BLOCK]; the vote is an address computation and a ds_add_u32
__device__ __forceinline__ void vote_bin32(uint32_t* hcol, unsigned b) { atomicAdd(&hcol[b * F3D_BLOCK], 1u); }

// VotingSegmentation.segment (voting.py:120-135) for one point from dword bins, then the stores
template <bool WRITE_VOTES>
__device__ __forceinline__ void finish_bin32(const uint32_t* hcol, int ncodes, const uint8_t* lut, const uint8_t* inv, int nfilter,
                                             const int* __restrict__ fcls, int nclasses, double threshold, bool store, int orig,
                                             int64_t* __restrict__ classes, uint16_t* __restrict__ votes_out, bool& bad,
                                             const uint16_t* cmin = nullptr, int64_t prefilled = F3D_NO_PREFILL) {
    const int ncols = nclasses + 1;
    const unsigned cbad = hcol[F3D_BLOCK];
    bad = cbad != 0u;
    unsigned best = 0, sum = cbad;
    for (int c = 2; c < ncodes; ++c) {
        const unsigned n = hcol[c * F3D_BLOCK];
        const unsigned key = (n << 8) | (unsigned)c;                               // count desc, then code desc = label asc
        sum += n;
        best = best > key ? best : key;
    }
    int win_c, win_i;
    if (nfilter > 0) {                                                             // votes[:, filter_classes]: first maximum wins
        win_c = -1; win_i = 0;
        for (int k = 0; k < nfilter; ++k) {
            const int l = fcls[k];                                                 // wave-uniform: scalar load
            int c = 0;
            if (l >= 0 && l < ncols) { const unsigned b = lut[l]; c = b >= 2u ? (int)hcol[b * F3D_BLOCK] : 0; }
            if (c > win_c) { win_c = c; win_i = k; }
        }
    } else {
        win_c = (int)(best >> 8); win_i = (int)inv[best & 0xFFu];
    }
    const int64_t cls = segment_point(win_c, win_i, (int)sum, nfilter, [&](int k) { return fcls[k]; }, nclasses, threshold, cmin);
    if (store && cls != prefilled) F3D_STORE_CLASS(&classes[orig], cls);      // (the label vector was filled with `prefilled` by a streaming kernel)
    if (WRITE_VOTES && store) {
        for (int l = 0; l < ncols; ++l) { const unsigned b = lut[l]; votes_out[(size_t)orig * ncols + l] = (uint16_t)(b >= 2u ? hcol[b * F3D_BLOCK] : 0u); }
    }
}

// GUARD: code 0 ("no sample") adds nothing.  A point casts more votes than there are views -- placeholders of the software pipeline (two per
// 64-view group and one per odd chunk), padding slots, the final pend vote: up to nviews + 3 ngroups + 2 -- so from ~240 views on
// byte 0 of word 0 could carry into byte 1, the rejected-label bin (a spurious IndexError and a total off by one).
template <bool GUARD>
__device__ __forceinline__ void vote_coded(unsigned& nvalid, uint32_t* hcol, unsigned b) {
    if (GUARD) {
        const unsigned one = b < 1u ? b : 1u;                                      // 0 for F3D_CODE_NONE
        nvalid += one;
        atomicAdd(&hcol[(b >> 2) * F3D_BLOCK], one << ((b & 3u) * 8u));
    } else {
        atomicAdd(&hcol[(b >> 2) * F3D_BLOCK], 1u << ((b & 3u) * 8u));
    }
}

__device__ __forceinline__ unsigned coded_count(const uint32_t* hcol, unsigned b) {
    return (hcol[(b >> 2) * F3D_BLOCK] >> ((b & 3u) * 8u)) & 0xFFu;
}

// VotingSegmentation.segment (voting.py:120-135) for one point of the fast kernel, then the stores.  Returns false when the
// 8-bit bins cannot be trusted (one of them wrapped: more than 255 agreeing views) -- the point then goes to the exact kernel.
template <bool WRITE_VOTES, bool WRAP>
__device__ __forceinline__ bool finish_coded(unsigned nvalid, const uint32_t* hcol, int words, const uint8_t* lut, const uint8_t* inv,
                                             int nfilter, const int* __restrict__ fcls, int nclasses, double threshold, bool store,
                                             int orig, int64_t* __restrict__ classes, uint16_t* __restrict__ votes_out, bool& bad,
                                             const uint16_t* cmin = nullptr, int64_t prefilled = F3D_NO_PREFILL) {
    const int ncols = nclasses + 1;
    unsigned best = 0, sum = 0;
    {
        const unsigned x = hcol[0];                                                // codes 0 (no sample: ignored), 1 (rejected label), 2, 3
        bad = ((x >> 8) & 0xFFu) != 0u;
        const unsigned c2 = (x >> 16) & 0xFFu, c3 = x >> 24;
        sum = ((x >> 8) & 0xFFu) + c2 + c3;
        best = (c2 << 8) | 2u;
        const unsigned k3 = (c3 << 8) | 3u;
        best = best > k3 ? best : k3;
    }
    // the other words, four bins at a time: the bytes of a word are split into two pairs of 16-bit lanes (count in the high byte, the
    // word's index below it), v_pk_max_u16 keeps per byte position the highest count and among equal counts the highest word = the
    // highest code = the smallest label; v_sad_u8 adds the four counts to the total.  6 instructions per word instead of 14.
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 me = {0, 0}, mo = {0, 0};
    for (int w = 1; w < words; ++w) {
        const unsigned x = hcol[w * F3D_BLOCK];
        const unsigned tag = (unsigned)w * 0x00010001u;
        sum = __builtin_amdgcn_sad_u8(x, 0u, sum);
        const unsigned o = (x & 0xFF00FF00u) | tag, e = ((x << 8) & 0xFF00FF00u) | tag;
        mo = __builtin_elementwise_max(mo, __builtin_bit_cast(u16x2, o));
        me = __builtin_elementwise_max(me, __builtin_bit_cast(u16x2, e));
    }
    {
        const unsigned cand[4] = {me.x, mo.x, me.y, mo.y};               // byte positions 0..3: (count << 8) | word
#pragma unroll
        for (unsigned j = 0; j < 4; ++j) {
            const unsigned key = (cand[j] & 0xFF00u) | ((cand[j] & 0x3Fu) << 2) | j;
            best = best > key ? best : key;
        }
    }
    // a bin that wrapped (256 votes) drops 256 from its own field and adds at most 1 to its neighbour's: the sum of the fields
    // then falls short of the number of votes cast.  Only possible with more than 255 views.
    if (WRAP && sum != nvalid) return false;
    int win_c, win_i;
    if (nfilter > 0) {                                                             // votes[:, filter_classes]: first maximum wins
        win_c = -1; win_i = 0;
        for (int k = 0; k < nfilter; ++k) {
            const int l = fcls[k];                                                 // wave-uniform: scalar load
            int c = 0;
            if (l >= 0 && l < ncols) { const unsigned b = lut[l]; c = b >= 2u ? (int)coded_count(hcol, b) : 0; }
            if (c > win_c) { win_c = c; win_i = k; }
        }
    } else {
        win_c = (int)(best >> 8); win_i = (int)inv[best & 0xFFu];
    }
    // total = the votes cast (every bin but "no sample"; a rejected label raises IndexError anyway)
    const int64_t cls = segment_point(win_c, win_i, (int)sum, nfilter, [&](int k) { return fcls[k]; }, nclasses, threshold, cmin);
    if (store && cls != prefilled) F3D_STORE_CLASS(&classes[orig], cls);
    if (WRITE_VOTES && store) {                                                    // presence book: an absent label reads bin 0 = 0
        for (int l = 0; l < ncols; ++l) { const unsigned b = lut[l]; votes_out[(size_t)orig * ncols + l] = (uint16_t)(b >= 2u ? coded_count(hcol, b) : 0u); }
    }
    return true;
}

// Wave-wide min / max of a float through DPP (no LDS traffic, no s_waitcnt): xor-1 and xor-2 inside quads, mirror the half
// rows and the rows (every lane of a 16-lane row then holds the row's result), row_bcast15 / row_bcast31 carry it across the
// rows into lane 63, which is read back as a scalar.  6 VALU + 1 v_readlane per value.  Inputs are never NaN (+-inf for
// lanes without a point).  Must be called with all 64 lanes active.
template <bool MAX>
__device__ __forceinline__ float wave_reduce(float v) {
    // written as one asm block: the compiler's own lowering of update_dpp + fminf spends 4 VALU per step (copy, v_mov_dpp,
    // canonicalise, min); v_min/v_max take the DPP operand directly.  s_nop 1 = the 2 wait states a DPP read needs after
    // a VALU write of the same register.
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

// The bounding box of a wave: three minima and three maxima reduced together.  The six chains are interleaved step by step, so the
// two wait states a DPP read needs after a VALU write of the same register are filled with the other chains' instructions instead
// of s_nop (36 VALU for the box instead of 36 + 42 idle issue slots).
__device__ __forceinline__ void wave_box(float& lo0, float& hi0, float& lo1, float& hi1, float& lo2, float& hi2) {
#define F3D_STEP(ctrl)                                                \
    "v_min_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl "\n\t" \
    "v_min_f32_dpp %2, %2, %2 " ctrl "\n\tv_max_f32_dpp %3, %3, %3 " ctrl "\n\t" \
    "v_min_f32_dpp %4, %4, %4 " ctrl "\n\tv_max_f32_dpp %5, %5, %5 " ctrl "\n\t"
    asm volatile("s_nop 1\n\t"
                 F3D_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 F3D_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 F3D_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
                 F3D_STEP("row_mirror row_mask:0xf bank_mask:0xf")
                 F3D_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 F3D_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1), "+v"(lo2), "+v"(hi2));
#undef F3D_STEP
#define F3D_L63(x) x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63))
    F3D_L63(lo0); F3D_L63(hi0); F3D_L63(lo1); F3D_L63(hi1); F3D_L63(lo2); F3D_L63(hi2);
#undef F3D_L63
}

// ------------------------------------------------------------------------------------------
// k_fuse: the fast kernel.  It contains NO exact arithmetic and no float64 in its view loop: a point for which any
// accelerator cannot prove its decision (a plane within the rounding margin, a pixel within the bound of a pixel border, a
// view without a usable centre row, huge or non-finite coordinates, a wrapped 8-bit bin) is not stored; its index is
// appended to `todo` and the next tier recomputes that point entirely.
//  * The kernel's time follows the NUMBER of vector instructions issued (measured: ~4.5 cycles per wave instruction whatever
//    its width; replacing float64 by float32 at equal count changed nothing), so the view loop is written for few of them:
//    a lane owns TWO points (a wave = 128 consecutive points of the cell-sorted cloud) and evaluates both with packed float32
//    (v_pk_fma_f32: one issue slot for two points); the 13 wave-uniform numbers of a view arrive through v_readlane once for
//    both; the integer part of the pixel is folded into a scalar tile offset (no per-lane "+ U"); a vote into dword bins is
//    two instructions.
//  (A) tile pre-cull, lanes-over-views: each wave reduces the bounding box of its points, lane j tests the box against the 5
//      planes of view 64g+j (float32 planes staged in LDS): box behind a plane -> the wave skips the view (scalar bit-scan);
//      box inside all planes -> no per-point cull and -- the frustum being the image's own pyramid -- no image-range test:
//      the coded masks carry a one-tile "no sample" border for the last-bit cases.
//  (B) per-point cull for the remaining "mixed" views in OFFSET form: the view's lane evaluates n . (c - plane point) once per
//      wave, every lane adds n32 . (p - c) -- the float32 offset arithmetic adds only ~1e-7 m to the margin.
//  (D) centre + offset projection (above) for every visible view.
// Vote histograms live in LDS as [point][bin][thread] (conflict-free): BIN32 = one dword per code for alphabets of at most
// F3D_BIN32_MAX_CODES codes, otherwise 8-bit bins packed 4 per dword.  The launcher enqueues one instance of each kind; an
// instance returns at once unless the code book's size is in its range [cmin, cmax].
// ------------------------------------------------------------------------------------------
#ifndef F3D_VTAB_GLOBAL
#define F3D_VTAB_GLOBAL 0                // 1: M, t, mnorm of the views always from the transposed global table (7.7 KB less LDS per block)
#endif
#ifndef F3D_XCD_CHUNK_LOG2
#define F3D_XCD_CHUNK_LOG2 6             // 64 tiles per chunk of the XCD-aware tile mapping (64 .. 256 measured alike)
#endif
#ifndef F3D_FUSE_WAVES
#define F3D_FUSE_WAVES 3                 // waves per SIMD the register allocation of k_fuse must allow (4 spills: measured slower)
#endif
// [group][field][view] copies of the per-view constants the lanes-over-views prologue reads (24 floats: cull planes, margins, image
// size; 15 doubles: M, t, mnorm), for the instance that cannot afford them in LDS: consecutive lanes read consecutive addresses
__device__ __forceinline__ void threshold_table(f3d_codebook* __restrict__ cb, double threshold);
__global__ __launch_bounds__(F3D_BLOCK) void k_fuse_setup(const f3d_view* __restrict__ views, int nviews, float* __restrict__ ctabT,
                                                           double* __restrict__ vtabT, f3d_codebook* __restrict__ cb, double threshold,
                                                           unsigned int* __restrict__ todo_count, int lut_nclasses, int lut_book, f3d_filter_args flt) {
    // (the one-shot call builds the code book in the same launch: its last block is k_code_lut's block -- one launch less per step)
    if (lut_book >= 0 && blockIdx.x == gridDim.x - 1) { code_lut_block(cb, lut_nclasses, lut_book, flt); return; }
    if (blockIdx.x == 0) {
        threshold_table(cb, threshold);                       // ... the call's threshold table (segment_point)
        if (todo_count && threadIdx.x < 4) todo_count[threadIdx.x] = 0u;   // ... and empty deferred lists
    }
    const int ngroups = (nviews + 63) >> 6;
    const int nsetup = lut_book >= 0 ? (int)gridDim.x - 1 : (int)gridDim.x;
    for (int k = blockIdx.x * F3D_BLOCK + threadIdx.x; k < ngroups * 64 * 39; k += nsetup * F3D_BLOCK) {
        const int v = k / 39, f = k - v * 39, g = v >> 6, l = v & 63;
        const f3d_view& vw = views[v < nviews ? v : nviews - 1];
        if (f < 24) ctabT[(g * 24 + f) * 64 + l] = reinterpret_cast<const float*>(&vw.cull_n32[0][0])[f];
        else vtabT[(g * F3D_VHEAD + (f - 24)) * 64 + l] = reinterpret_cast<const double*>(&vw)[f - 24];
    }
}

// CARRY: the instance of the view-chunked call (f3d_fuse_chunk_dev): the views arrive in several launches; between two of them a
// point's vote bins live in HBM as `carry` [tile][point slot][word][F3D_BLOCK], 8-bit bins packed four to a dword whatever the LDS
// form (at most 255 views in total: no bin can wrap); byte 0 of word 0 -- the "no sample" bin nothing reads -- carries the point's
// "deferred" flag.  chunk_flags bit 0: load the carry (not the first chunk), bit 1: store it instead of finishing (not the last).
// xyz_keep (first chunk of a cloud read through perm): the points are written back in cell order for the chunks that follow.
template <typename T, int PPL, bool WRITE_VOTES, bool BIN32, bool WRAP, bool TLDS, bool CARRY>
__global__ __launch_bounds__(F3D_BLOCK, F3D_FUSE_WAVES) void k_fuse(const T* __restrict__ xyz, int64_t n,
                                                     const f3d_view* __restrict__ views, int nviews,
                                                     const uint8_t* __restrict__ cmasks, int H, int W,
                                                     int nclasses, int nfilter, const int* __restrict__ fcls, double threshold,
                                                     int64_t* __restrict__ classes, uint16_t* __restrict__ votes_out,
                                                     int* __restrict__ err, const int32_t* __restrict__ perm, int gather_xyz,
                                                     unsigned int* __restrict__ todo_count, int32_t* __restrict__ todo,
                                                     const f3d_codebook* __restrict__ cb, int cmin, int cmax,
                                                     const float* __restrict__ ctabT, const double* __restrict__ vtabT,
                                                     uint32_t* __restrict__ carry, int chunk_flags, T* __restrict__ xyz_keep,
                                                     unsigned long long* __restrict__ umask, uint32_t* __restrict__ park, int park_slots, int park_stride,
                                                     int64_t prefilled) {
    const int ncodes = cb->ncodes;                                        // wave-uniform: scalar load
    if (ncodes > cmax || ncodes < cmin) return;                           // the other instance's book
    const int words = (ncodes + 3) >> 2;
    const int hdw = BIN32 ? ncodes : words;                               // histogram dwords per point
    extern __shared__ uint32_t lds_u32[];
    // TLDS: the per-view constants of a 64-view group are staged in LDS; otherwise (the large-alphabet instances, whose histograms need the
    // room) they are read from the transposed global tables ctabT / vtabT
    float* ctab = reinterpret_cast<float*>(lds_u32);                      // [64][F3D_CULL_ROW] cull planes (+ image size) of one view group
    uint32_t* lutw = TLDS ? lds_u32 + 64 * F3D_CULL_ROW : lds_u32;        // lut[256], inv[256] (bytes), cmin[256] (uint16): F3D_BOOK_DWORDS
    double* vtab = reinterpret_cast<double*>(lutw + F3D_BOOK_DWORDS);     // [F3D_VHEAD][64]: M, t, mnorm of the group's views
    uint32_t* hist = (TLDS && !F3D_VTAB_GLOBAL) ? lutw + F3D_BOOK_DWORDS + 2 * F3D_VHEAD * 64 : lutw + F3D_BOOK_DWORDS; // [PPL][hdw][F3D_BLOCK]
    const uint16_t* vmin = (WRAP || nviews > 255) ? nullptr : reinterpret_cast<const uint16_t*>(lutw + 128);   // totals beyond 255: the division itself
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(lutw);
    const uint8_t* inv = lut + 256;
    const int tid = threadIdx.x, lane = threadIdx.x & 63;
    constexpr int TILE = F3D_BLOCK * PPL;                                 // PPL = points per lane: every instance in use has 2 (a one-point
                                                                          // instance for huge histograms was measured slower than two points at 2 blocks per CU)
    const int npts = (int)n;                                              // n < 2^31 - TILE (checked by the launcher): 32-bit indices
    const int ntiles = (int)((n + TILE - 1) / TILE);
    const unsigned plane = (unsigned)f3d_coded_plane(H, W);               // bytes per view of the coded masks (V * plane < 2^32: launcher)
    const int pitch = f3d_coded_pitch(W);                                 // tiles per row, border included
    const int c_row = 64 * pitch - 64;
    const int ngroups = (nviews + 63) >> 6;
    const float umaxf = (float)(W > H ? W : H), Wf = (float)W, Hf = (float)H;

    // the coded masks as a raw buffer resource: a gather is then buffer_load_ubyte with the 32-bit offset as it is (a flat global load
    // wants a 64-bit address: two more vector instructions per gather)
    const __amdgpu_buffer_rsrc_t cm_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(cmasks), 0, (int)((unsigned)nviews * plane), 0x00020000);
#if defined(F3D_EXP_GATHER) && F3D_EXP_GATHER == 1     // timing experiments only (results are wrong): every lane reads the same 64 bytes
    auto gather = [&](unsigned off) -> unsigned { return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(cm_rsrc, (int)(off & 63u), 0, 0); };
#elif defined(F3D_EXP_GATHER) && F3D_EXP_GATHER == 2   // no memory instruction at all
    auto gather = [&](unsigned off) -> unsigned { return off & 2u; };
#else
    auto gather = [&](unsigned off) -> unsigned { return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(cm_rsrc, (int)off, 0, 0); };
#endif
    lutw[tid] = reinterpret_cast<const uint32_t*>(cb->lut)[tid];          // lut, inv and cmin are adjacent in the book (256 dwords)
    auto stage_group = [&](int g) {                                       // whole block; caller brackets with barriers
        const int nv = min(64, nviews - 64 * g);
        for (int k = tid; k < nv * 24; k += F3D_BLOCK) {
            const int vi = k / 24, f = k - vi * 24;
            ctab[vi * F3D_CULL_ROW + f] = reinterpret_cast<const float*>(&views[64 * g + vi].cull_n32[0][0])[f];
        }
        if (!F3D_VTAB_GLOBAL)
            for (int k = tid; k < nv * F3D_VHEAD; k += F3D_BLOCK) {
                const int vi = k / F3D_VHEAD, f = k - vi * F3D_VHEAD;
                vtab[f * 64 + vi] = reinterpret_cast<const double*>(&views[64 * g + vi])[f];
            }
    };
    if (TLDS && ngroups > 1) return;                                      // (never launched: LDS-staged tables serve one 64-view group, see launch_fuse_t)
    if (TLDS) stage_group(0);
    __syncthreads();

    // XCD-aware tile mapping: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch).  The XCDs take CHUNKS of
    // 64 consecutive tiles in turn (chunk c -> XCD c mod 8): with a cell-sorted cloud a chunk is a compact region of
    // space whose pixels in every mask stay resident in that XCD's 4 MiB L2, and the chunks deal the cloud's cheap and expensive
    // regions (points seen by few / by many views) evenly.  r1/r2 gave every XCD ONE contiguous eighth of the cloud: the XCD
    // with the busiest region finished last while others idled -- 0.98 -> 0.85 ms for the fused call at C3, 1.46 -> 1.23 ms with
    // iid masks (profiles/r03_summary.md).  Placement affects speed only, never results.
    int xlog = F3D_XCD_CHUNK_LOG2;                                        // chunk = 2^xlog tiles, smaller for small clouds: at least four chunks per XCD
#ifndef F3D_XCD_MIN_ROWS
#define F3D_XCD_MIN_ROWS 8               // chunk rows per XCD at least (small clouds: 1.25M points 0.349 -> 0.314 ms per step with 8 instead of 4)
#endif
    while (xlog > 0 && ntiles < ((8 * F3D_XCD_MIN_ROWS) << xlog)) --xlog;
    const int chunk_rows = (ntiles + (8 << xlog) - 1) >> (xlog + 3);
    const int tiles_per_xcd = chunk_rows << xlog;                         // positions j of one XCD's list (the last row may hold fewer tiles)
    const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, gx = gridDim.x >> 3;
    uint32_t* const hcol0 = hist + tid;
    uint32_t* const hcol1 = PPL == 2 ? hist + hdw * F3D_BLOCK + tid : hcol0;
    for (int j = bx; j < tiles_per_xcd; j += gx) {
        const int tile = ((j >> xlog) << (xlog + 3)) + (xcd << xlog) + (j & ((1 << xlog) - 1));
        if (tile >= ntiles) continue;
        const int i0 = tile * TILE + (tid >> 6) * (64 * PPL) + lane;      // this lane's first point; its second is i0 + 64
        bool live[2], act[2], defer[2];
        int orig[2];
        f32x2 DX, DY, DZ;                                                  // offsets of the lane's two points from the box centre
        float c0, c1, c2, e0, e1, e2, ps_box;
        bool wave_any;
        {
            f3d_p3 p[2];                                                           // float64 only inside this scope
            float lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = i0 + 64 * q;
                live[q] = (q < PPL) & (i < npts);
                orig[q] = live[q] ? (perm ? perm[i] : i) : i;                      // caller-order index of this point
                p[q].x = p[q].y = p[q].z = 0.0;
                if (live[q]) p[q] = load_point(xyz, (int64_t)(gather_xyz ? orig[q] : i));
                // first of several view chunks of a gathered cloud: leave the points behind in cell order, the later chunks stream them
                if (CARRY && xyz_keep && live[q]) { xyz_keep[3 * (size_t)i] = (T)p[q].x; xyz_keep[3 * (size_t)i + 1] = (T)p[q].y; xyz_keep[3 * (size_t)i + 2] = (T)p[q].z; }
                const double pscale = (fabs(p[q].x) + fabs(p[q].y)) + fabs(p[q].z);
                act[q] = live[q] & (pscale < 1.0e30);          // float32 work is meaningful (no overflow, no NaN)
                defer[q] = live[q] & !act[q];                  // this point goes to the next tier
                const float x32 = (float)p[q].x, y32 = (float)p[q].y, z32 = (float)p[q].z;
                if (act[q]) {
                    lo0 = fminf(lo0, x32); hi0 = fmaxf(hi0, x32); lo1 = fminf(lo1, y32); hi1 = fmaxf(hi1, y32);
                    lo2 = fminf(lo2, z32); hi2 = fmaxf(hi2, z32);
                }
            }
            // ---- (A) bounding box of this wave's live, well-behaved points
            wave_box(lo0, hi0, lo1, hi1, lo2, hi2);
            wave_any = __any(act[0] | act[1]);
            c0 = 0.5f * (lo0 + hi0); c1 = 0.5f * (lo1 + hi1); c2 = 0.5f * (lo2 + hi2);
            e0 = 0.5f * (hi0 - lo0) * 1.000002f + 1e-30f; e1 = 0.5f * (hi1 - lo1) * 1.000002f + 1e-30f;
            e2 = 0.5f * (hi2 - lo2) * 1.000002f + 1e-30f;
            ps_box = ((fabsf(c0) + fabsf(c1)) + fabsf(c2)) + ((e0 + e1) + e2);
            // offsets from the box centre (c is a float32, hence exact as a double; the difference is rounded once)
            DX = (f32x2){(float)(p[0].x - (double)c0), (float)(p[1].x - (double)c0)};
            DY = (f32x2){(float)(p[0].y - (double)c1), (float)(p[1].y - (double)c1)};
            DZ = (f32x2){(float)(p[0].z - (double)c2), (float)(p[1].z - (double)c2)};
            // a lane without a usable point carries NaN offsets: every comparison of the view loop is then false for it (no cull
            // passed, no pixel proven, nothing gathered), so the loop needs no "active" flag
            if (!act[0]) { DX.x = __builtin_nanf(""); }
            if (!act[1]) { DX.y = __builtin_nanf(""); }
        }
        // PART (every call that is not view-chunked): a point with unproven decisions keeps the votes of its proven views.  um = the views of
        // the current 64-view group that left one of the point's decisions unproven.  At the end of a group a point with such views takes its
        // slot in the deferred list (once) and leaves the group's mask in umask[slot][group]; at the end its bins are parked in park[slot],
        // and the float64 tier visits the marked views only (typically one of all).  Points without usable float32 coordinates, with a
        // bin beyond 255, beyond park_slots deferred ones or F3D_PART_MAX_GROUPS view groups are redone from nothing (sign bit of the list entry).
        constexpr bool PART = !CARRY;
        const bool part_rt = PART && park_slots > 0 && ngroups <= F3D_PART_MAX_GROUPS;
        unsigned long long um[2] = {0ull, 0ull};             // PART
        int slot[2] = {-1, -1};                              // PART: the point's place in the deferred list, taken at its first unproven view
        unsigned unsure[2] = {0u, 0u};                       // !PART (or no room for masks): a visible view left one of the point's decisions unproven
        if (CARRY && (chunk_flags & 1)) {                                   // the bins (and the deferred flag) of the earlier view chunks
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                uint32_t* hc = q ? hcol1 : hcol0;
                const uint32_t* cq = carry + ((size_t)(tile * PPL + q) * words) * F3D_BLOCK + tid;
                for (int wd = 0; wd < words; ++wd) {
                    uint32_t x = cq[wd * F3D_BLOCK];
                    if (wd == 0) { defer[q] = defer[q] | (live[q] & ((x & 0xFFu) != 0u)); x &= ~0xFFu; }
                    if (BIN32) {
#pragma unroll
                        for (int b4 = 0; b4 < 4; ++b4) if (4 * wd + b4 < ncodes) hc[(4 * wd + b4) * F3D_BLOCK] = (x >> (8 * b4)) & 0xFFu;
                    } else {
                        hc[wd * F3D_BLOCK] = x;
                    }
                }
            }
        } else {
            for (int wd = 0; wd < hdw; ++wd) { hcol0[wd * F3D_BLOCK] = 0u; if (PPL == 2) hcol1[wd * F3D_BLOCK] = 0u; }   // own columns only: no barrier
        }
        unsigned nvalid[2] = {0u, 0u};
        unsigned pend[2] = {F3D_CODE_NONE, F3D_CODE_NONE};   // software-pipelined gathers: a code is voted one view (chunk) later
        unsigned ccode[2][F3D_CHUNK];
#pragma unroll
        for (int k = 0; k < F3D_CHUNK; ++k) ccode[0][k] = ccode[1][k] = F3D_CODE_NONE;
#if F3D_GATHER_DEPTH == 2                                    // codes are voted two chunks after their gathers were issued
        unsigned ccode2[2][F3D_CHUNK];
#pragma unroll
        for (int k = 0; k < F3D_CHUNK; ++k) ccode2[0][k] = ccode2[1][k] = F3D_CODE_NONE;
#define F3D_OLDEST ccode2
#else
#define F3D_OLDEST ccode
#endif
        auto vote = [&](int q, unsigned b) {
            if (q >= PPL) return;
            if (BIN32) vote_bin32(q ? hcol1 : hcol0, b);
            else vote_coded<WRAP || CARRY>(nvalid[q], q ? hcol1 : hcol0, b);   // a chunk of a chunked call is guarded too (any number of views per chunk)
        };

        for (int g = 0; g < ngroups; ++g) {
            // lane j <-> view 64g + j: classify the wave's box against that view's planes, project the box centre
            const int vj = 64 * g + lane;
            bool box_out = false, box_in = true, own_image = false;
            float pb0 = 0.f, pb1 = 0.f, pb2 = 0.f, pb3 = 0.f, pb4 = 0.f, pmarg = 0.f;    // (B): n . (c - plane point) per plane, margin
            unsigned pneed = 0u;                                                         // (B): the planes that need a per-point test at all
            if (vj < nviews) {
                // field f of this lane's view: LDS row (odd stride) or the transposed global table
                const float* rowp = TLDS ? ctab + lane * F3D_CULL_ROW : ctabT + (size_t)g * 24 * 64 + lane;
                auto rowf = [&](int f) { return TLDS ? rowp[f] : rowp[f * 64]; };
                const float marg = 2.0f * __builtin_fmaf(rowf(20), ps_box, rowf(21));
                float bmax = 0.f, smax = 0.f, base[F3D_NPLANES];
                unsigned need = 0u;
#pragma unroll
                for (int m = 0; m < F3D_NPLANES; ++m) {
                    const float n0 = rowf(3 * m), n1 = rowf(3 * m + 1), n2 = rowf(3 * m + 2);
                    base[m] = __builtin_fmaf(n0, c0, __builtin_fmaf(n1, c1, __builtin_fmaf(n2, c2, -rowf(15 + m))));
                    const float spread = __builtin_fmaf(fabsf(n0), e0, __builtin_fmaf(fabsf(n1), e1, fabsf(n2) * e2));
                    box_out = box_out | (base[m] + spread < -marg);
                    box_in = box_in & (base[m] - spread > marg);
                    need |= (base[m] - spread > marg) ? 0u : (1u << m);         // planes the box is not entirely in front of
                    bmax = fmaxf(bmax, fabsf(base[m])); smax = fmaxf(smax, spread);
                }
                // offset-form cull of a mixed view: a = base + n32 . d.  base itself is the float32 world-coordinate value, off the
                // real n . (c - pp) by at most marg / 2 (that is what marg bounds); on top of it the float32 offset arithmetic:
                // 2^-24 (4 |base| + 5 sum |n_j| E_j).
                pb0 = base[0]; pb1 = base[1]; pb2 = base[2]; pb3 = base[3]; pb4 = base[4];
                pneed = need;
                pmarg = 0.5f * marg + 1.01f * (float)F3D_U24 * (4.0f * bmax + 5.0f * smax);
                own_image = (int)(rowf(22) == Wf) & (int)(rowf(23) == Hf);  // the frustum was built for this mask size: inside the planes = inside the image
            }
            centre_row mine = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0, 0};
            bool row_ok = false;
            if (vj < nviews && !box_out && wave_any)
                row_ok = centre_precompute((TLDS && !F3D_VTAB_GLOBAL) ? vtab + lane : vtabT + (size_t)g * F3D_VHEAD * 64 + lane, 64, c0, c1, c2, e0, e1, e2, umaxf, pitch, mine);
            mine.obase += (unsigned)vj * plane;             // absolute: the view's plane included
            const unsigned long long valid_m = __ballot(vj < nviews);
            unsigned long long out_m = __ballot(vj < nviews && box_out);
            const unsigned long long in_m = __ballot(vj < nviews && box_in && !box_out && own_image);   // whole-wave views without any test
            const unsigned long long row_m = __ballot(row_ok);
            if (!wave_any) out_m = valid_m;                 // nothing but deferred / dead lanes in this wave
            // views whose planes all contain the wave's box: every live lane is inside, no cull, no divergence, no range test.
            // Taken F3D_CHUNK at a time: project the chunk, retire the previous chunk's votes, then issue the chunk's mask
            // gathers back to back -- 2 * F3D_CHUNK gathers in flight per wave, each with a whole chunk of arithmetic to land.
            unsigned long long todo_v = valid_m & ~out_m & in_m & row_m;
#if defined(F3D_EXP_SKIP) && (F3D_EXP_SKIP == 1 || F3D_EXP_SKIP == 3)   // timing experiment: no whole-wave views
            todo_v = 0ull;
#endif
            while (todo_v) {
                int cbit[F3D_CHUNK]; unsigned coff[2][F3D_CHUNK];
                bool use[F3D_CHUNK];
                unsigned long long unsm[2][F3D_CHUNK];                         // PART: the lanes a view left unproven, as a (scalar) lane mask
#pragma unroll
                for (int k = 0; k < F3D_CHUNK; ++k) {
                    use[k] = todo_v != 0ull;
                    cbit[k] = use[k] ? __builtin_ctzll(todo_v) : (k ? cbit[0] : 0);   // an unused slot re-reads a valid row and gathers "no sample"
                    if (use[k]) todo_v &= todo_v - 1ull;
                }
#pragma unroll
                for (int k = 0; k < F3D_CHUNK; ++k) {
                    const centre_row r = read_row(mine, cbit[k]);
                    int fi0[2], fi1[2]; bool safe[2];
                    project_offset2(r, DX, DY, DZ, fi0, fi1, safe);
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        if (PART) unsm[q][k] = __ballot(act[q] & !safe[q]);    // (an unused slot repeats a row of this chunk: same answer)
                        else unsure[q] = safe[q] ? unsure[q] : 1u;
                        const unsigned o = r.obase + rel_offset(fi0[q], fi1[q], c_row);   // computed for every lane: a select, not a branch
                        coff[q][k] = (safe[q] & use[k]) ? o : 0u;              // offset 0: a border tile, "no sample"
                    }
                }
#ifndef F3D_NO_VOTE_PIN
                // the previous chunk's codes are consumed only now, after this chunk's arithmetic: without this (empty) dependency the
                // scheduler hoists the votes -- and with them the s_waitcnt for the gathers -- to the top of the iteration
#pragma unroll
                for (int k = 0; k < F3D_CHUNK; ++k) {
                    asm volatile("" : "+v"(F3D_OLDEST[0][k]) : "v"(coff[0][k]));
                    if (PPL == 2) asm volatile("" : "+v"(F3D_OLDEST[1][k]) : "v"(coff[1][k]));
                }
#endif
#pragma unroll
                for (int k = 0; k < F3D_CHUNK; ++k) { vote(0, F3D_OLDEST[0][k]); vote(1, F3D_OLDEST[1][k]); }
#if F3D_GATHER_DEPTH == 2
#pragma unroll
                for (int k = 0; k < F3D_CHUNK; ++k) { ccode2[0][k] = ccode[0][k]; ccode2[1][k] = ccode[1][k]; }
#endif
#pragma unroll
                for (int k = 0; k < F3D_CHUNK; ++k) { ccode[0][k] = gather(coff[0][k]); if (PPL == 2) ccode[1][k] = gather(coff[1][k]); }
                if (PART) {                                                    // scalar in the common case: OR of the lane masks, one branch
                    unsigned long long au = 0ull;
#pragma unroll
                    for (int k = 0; k < F3D_CHUNK; ++k) au |= unsm[0][k] | (PPL == 2 ? unsm[1][k] : 0ull);
                    if (__builtin_expect(au != 0ull, 0)) {                     // rare
                        asm volatile("" ::: "memory");                         // (keeps the compiler from flattening this branch into the loop body: +34 VALU per chunk)
                        const unsigned long long me = 1ull << lane;
#pragma unroll
                        for (int k = 0; k < F3D_CHUNK; ++k)
#pragma unroll
                            for (int q = 0; q < PPL; ++q) um[q] |= (unsm[q][k] & me) ? (1ull << cbit[k]) : 0ull;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < F3D_CHUNK; ++k) {
#if F3D_GATHER_DEPTH == 2
                vote(0, ccode2[0][k]); vote(1, ccode2[1][k]);
                ccode2[0][k] = ccode2[1][k] = F3D_CODE_NONE;
#endif
                vote(0, ccode[0][k]); vote(1, ccode[1][k]);
                ccode[0][k] = ccode[1][k] = F3D_CODE_NONE;
            }
            // every other visible view: per-point cull in offset form, image-range test; a lane inside the rounding margin of a
            // plane, or a view without a usable centre row, sends the point to the next tier
            todo_v = valid_m & ~out_m & ~(in_m & row_m);
#if defined(F3D_EXP_SKIP) && F3D_EXP_SKIP == 2         // timing experiment: no mixed views
            todo_v = 0ull;
#elif defined(F3D_EXP_SKIP) && F3D_EXP_SKIP == 3       // neither: the per-tile prologue and epilogue alone
            todo_v = 0ull;
#endif
            while (todo_v) {
                const int bit = __builtin_ctzll(todo_v);
                todo_v &= todo_v - 1ull;
                const int v = 64 * g + bit;
                const f3d_view& vw = views[v];
                float nn[F3D_NPLANES][3];                   // wave-uniform: scalar loads, requested together
#pragma unroll
                for (int m = 0; m < F3D_NPLANES; ++m) { nn[m][0] = vw.cull_n32[m][0]; nn[m][1] = vw.cull_n32[m][1]; nn[m][2] = vw.cull_n32[m][2]; }
#pragma unroll
                for (int m = 0; m < F3D_NPLANES; ++m) { asm volatile("" : "+s"(nn[m][0])); asm volatile("" : "+s"(nn[m][1])); asm volatile("" : "+s"(nn[m][2])); }
#define F3D_RL1(x) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), bit))
                const float mg = F3D_RL1(pmarg);
                // only the planes the wave's box straddles are tested per point (usually one or two of the five): the box is entirely in
                // front of the others, by the same margin that makes a view a whole-wave view
                const unsigned need = (unsigned)__builtin_amdgcn_readlane((int)pneed, bit);
                const float* pbs[F3D_NPLANES] = {&pb0, &pb1, &pb2, &pb3, &pb4};
                const bool has_row = (row_m >> bit) & 1ull;  // wave-uniform
                bool maybe[2] = {true, true}, sure[2] = {true, true};
#pragma unroll
                for (int m = 0; m < F3D_NPLANES; ++m) {
                    if (!((need >> m) & 1u)) continue;       // wave-uniform: a scalar branch
                    const float bbm = F3D_RL1(*pbs[m]);
                    const f32x2 a = F3D_FMA2(splat2(nn[m][0]), DX, F3D_FMA2(splat2(nn[m][1]), DY, F3D_FMA2(splat2(nn[m][2]), DZ, splat2(bbm))));
                    maybe[0] = maybe[0] & (a.x > -mg); sure[0] = sure[0] & (a.x > mg);
                    maybe[1] = maybe[1] & (a.y > -mg); sure[1] = sure[1] & (a.y > mg);
                }
#undef F3D_RL1
                unsigned off[2] = {0u, 0u};
                bool unsv[2] = {false, false};
                if (has_row) {
                    centre_row r = read_row(mine, bit);
                    const int U8 = __builtin_amdgcn_readlane(mine.U8, bit), V8 = __builtin_amdgcn_readlane(mine.V8, bit);
                    int fi0[2], fi1[2]; bool safe[2];
                    project_offset2(r, DX, DY, DZ, fi0, fi1, safe);
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        const bool inside = sure[q];
                        const bool hit = inside & safe[q] & ((unsigned)(fi0[q] + U8) < (unsigned)W) & ((unsigned)(fi1[q] + V8) < (unsigned)H);
                        unsv[q] = (maybe[q] & !sure[q]) | (inside & !safe[q]);
                        const unsigned o = r.obase + rel_offset(fi0[q], fi1[q], c_row);
                        off[q] = hit ? o : 0u;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < PPL; ++q) unsv[q] = maybe[q];                         // the box comes too close to this view's camera plane
                }
                if (PART) {
                    // (act: with no plane to test -- a box entirely inside the view -- a lane without a usable point is "inside" and never "safe")
                    const unsigned long long u0 = __ballot(unsv[0] & act[0]), u1 = __ballot(unsv[1] & act[1]);
                    if (__builtin_expect((u0 | u1) != 0ull, 0)) {              // rare
                        asm volatile("" ::: "memory");                         // (as above)
                        const unsigned long long me = 1ull << lane;
                        um[0] |= (u0 & me) ? (1ull << bit) : 0ull;
                        if (PPL == 2) um[1] |= (u1 & me) ? (1ull << bit) : 0ull;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < PPL; ++q) unsure[q] = unsv[q] ? 1u : unsure[q];
                }
#ifndef F3D_NO_VOTE_PIN
                asm volatile("" : "+v"(pend[0]) : "v"(off[0]));
                if (PPL == 2) asm volatile("" : "+v"(pend[1]) : "v"(off[1]));
#endif
                vote(0, pend[0]); vote(1, pend[1]);
                pend[0] = gather(off[0]); if (PPL == 2) pend[1] = gather(off[1]);
            }
            // end of the group: the points with unproven views leave the group's mask behind (rare)
            if (PART && __any(((um[0] | um[1]) != 0ull) | (slot[0] >= 0) | (slot[1] >= 0))) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    if (!part_rt) { unsure[q] = um[q] != 0ull ? 1u : unsure[q]; }
                    else if (act[q] && (um[q] != 0ull || slot[q] >= 0)) {         // (act: only a usable point ever takes a slot)
                        if (slot[q] < 0) {
                            slot[q] = (int)atomicAdd(todo_count, 1u);
                            if (slot[q] < park_slots) for (int gg = 0; gg < g; ++gg) umask[(size_t)slot[q] * ngroups + gg] = 0ull;
                        }
                        if (slot[q] < park_slots) umask[(size_t)slot[q] * ngroups + g] = um[q];
                    }
                    um[q] = 0ull;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            vote(q, pend[q]);
            const bool part = act[q] & ((PART && slot[q] >= 0) | (unsure[q] != 0u));   // unproven views: the float64 tier adds their votes to the parked bins
            bool full = defer[q] | (part & !part_rt);          // ... or redoes the point from nothing
            defer[q] = defer[q] | part;
            uint32_t* hc = q ? hcol1 : hcol0;
            if (CARRY && (chunk_flags & 2)) {                                 // more views to come: park the bins in HBM
                uint32_t* cq = carry + ((size_t)(tile * PPL + q) * words) * F3D_BLOCK + tid;
                for (int wd = 0; wd < words; ++wd) {
                    uint32_t x = 0u;
                    if (BIN32) {
#pragma unroll
                        for (int b4 = 0; b4 < 4; ++b4) if (4 * wd + b4 < ncodes) x |= (hc[(4 * wd + b4) * F3D_BLOCK] & 0xFFu) << (8 * b4);   // (the "no sample" dword may exceed 255)
                    } else {
                        x = hc[wd * F3D_BLOCK];
                    }
                    if (wd == 0) x = (x & ~0xFFu) | (defer[q] ? 1u : 0u);
                    cq[wd * F3D_BLOCK] = x;
                }
                continue;
            }
            bool bad = false, trusted = true;
            if (BIN32) finish_bin32<WRITE_VOTES>(hc, ncodes, lut, inv, nfilter, fcls, nclasses, threshold, live[q] & !defer[q], orig[q], classes, votes_out, bad, vmin, prefilled);
            else trusted = finish_coded<WRITE_VOTES, WRAP>(nvalid[q], hc, words, lut, inv, nfilter, fcls, nclasses, threshold, live[q] & !defer[q], orig[q],
                                                           classes, votes_out, bad, vmin, prefilled);
            full = full | (live[q] & !trusted);
            const bool d = defer[q] | full;
            if (d) {
                const int sl = (PART && slot[q] >= 0) ? slot[q] : (int)atomicAdd(todo_count, 1u);
                full = full | !part_rt | (sl >= park_slots) | !(PART && slot[q] >= 0);
                if (BIN32 && !full && nviews > 255)                           // (a dword bin beyond 255 does not fit the parked byte)
                    for (int c = 1; c < ncodes; ++c) full = full | (hc[c * F3D_BLOCK] > 255u);
                todo[sl] = (gather_xyz ? orig[q] : (i0 + 64 * q)) | (full ? (int)0x80000000 : 0);   // index into xyz as this launch sees it
                if (PART && !full) {                                          // the bins, four 8-bit counts to a dword as the other tiers hold them
                    uint32_t* pk = park + (size_t)sl * park_stride;
                    for (int wd = 0; wd < words; ++wd) {
                        uint32_t x = 0u;
                        if (BIN32) {
#pragma unroll
                            for (int b4 = 0; b4 < 4; ++b4) if (4 * wd + b4 < ncodes) x |= (hc[(4 * wd + b4) * F3D_BLOCK] & 0xFFu) << (8 * b4);
                        } else {
                            x = hc[wd * F3D_BLOCK];
                        }
                        pk[wd] = wd == 0 ? (x & ~0xFFu) : x;                  // ("no sample" is nobody's vote)
                    }
                }
            }
            if (bad & !d) atomicOr(err, F3D_DEVERR_FUSE);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_fuse_mid: the middle tier, on the points k_fuse deferred (1-5 % of a cloud: the float32 bound leaves a band of ~1e-4 px
// around every pixel border undecided) and, of those, on the views k_fuse left open.  Per (point, view) pair, float64:
// float32 cull -> float64 FMA refinement -> fast projection h = M (p - t) (9 FMAs, reciprocal + 2 Newton steps), accepted
// when farther than 2^-43 (|p-t|_1 |r| (mnorm_k + umax mnorm_2) + umax) from a pixel border (> 10x the distance between
// this and the canonical operation order).  What it still cannot prove (a point within rounding of a plane or of a pixel
// border: ~1e-3 of the cloud) goes on to the exact tier through the second list.  Same coded masks and vote bins as k_fuse.
// ------------------------------------------------------------------------------------------
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

// Returns true when (iu, iv) are proven equal to the canonical floor(u), floor(v) AND lie inside the W x H image; `unsure`
// is set when the canonical arithmetic has to decide.  umax >= max(W, H): for |u| <= umax the bound is rigorous; beyond it
// both paths are out of the image anyway.
__device__ __forceinline__ bool project_fast(const f3d_view& vw, double umax, f3d_p3 p, int W, int H, int& iu, int& iv, bool& unsure) {
    const double d0 = p.x - vw.t[0], d1 = p.y - vw.t[1], d2 = p.z - vw.t[2];
    const double h0 = __builtin_fma(vw.M[0], d0, __builtin_fma(vw.M[1], d1, vw.M[2] * d2));
    const double h1 = __builtin_fma(vw.M[3], d0, __builtin_fma(vw.M[4], d1, vw.M[5] * d2));
    const double h2 = __builtin_fma(vw.M[6], d0, __builtin_fma(vw.M[7], d1, vw.M[8] * d2));
    const double r = rcp_newton(h2);
    const double uf = h0 * r, vf = h1 * r;
    const double fu = floor(uf), fv = floor(vf);
    const double c1 = __builtin_fma(umax, vw.mnorm[2], fmax(vw.mnorm[0], vw.mnorm[1]));
    // |fast - canonical| <= 2^-43 * (|d|_1 |r| (mnorm_k + |u| mnorm_2) + |u|), evaluated with |u| <= umax
    const double b = F3D_FAST_EPS * __builtin_fma(((fabs(d0) + fabs(d1)) + fabs(d2)) * fabs(r), c1, umax);
    // frac in (b, 1-b)  <=>  |frac - 0.5| < 0.5 - b      (NaN / inf -> false)
    const bool safe = (fabs((uf - fu) - 0.5) < 0.5 - b) && (fabs((vf - fv) - 0.5) < 0.5 - b);
    unsure = !safe;
    iu = (int)fu; iv = (int)fv;                              // saturating conversions; only used when safe
    return safe & ((unsigned)iu < (unsigned)W) & ((unsigned)iv < (unsigned)H);
}

// A pair this tier cannot prove either (a point within rounding of a plane or of a pixel border, a point without usable float32
// coordinates) is decided right here by the reference's arithmetic (exact 5-plane test, canonical projection, IEEE divisions) -- the rare
// branch of `pair` below.  r3 first ran that arithmetic as a launch of its own over a second list (4.7 us when empty, every step).
// Only a point whose 8-bit bin wrapped (more than 255 views) leaves this kernel unlabelled: out_list -> k_fuse_exact (16-bit bins).
// diag_count: the number of points that needed the reference's arithmetic (f3d_debug_fuse_deferred).
template <typename T, bool WRITE_VOTES>
__global__ __launch_bounds__(F3D_BLOCK) void k_fuse_mid(const T* __restrict__ xyz, const unsigned int* __restrict__ in_count, const int32_t* __restrict__ in_list,
                                                         const f3d_view* __restrict__ views, int nviews,
                                                         const uint8_t* __restrict__ cmasks, int H, int W,
                                                         int nclasses, int nfilter, const int* __restrict__ fcls, double threshold,
                                                         int64_t* __restrict__ classes, uint16_t* __restrict__ votes_out,
                                                         int* __restrict__ err, const int32_t* __restrict__ perm, int gather_xyz,
                                                         unsigned int* __restrict__ out_count, int32_t* __restrict__ out_list,
                                                         unsigned int* __restrict__ diag_count, const f3d_codebook* __restrict__ cb,
                                                         const unsigned long long* __restrict__ umask, const uint32_t* __restrict__ park,
                                                         int park_slots, int park_stride) {
    // A block takes 256 deferred points at a time.  A point k_fuse parked (list entry without the sign bit, slot < park_slots) comes with
    // the bins of its proven views and the mask of the views still to be decided -- typically ONE of 64; the others start from empty
    // bins and need every view.  The (point, view) pairs of the 256 points are numbered through a block-wide prefix sum and dealt to the
    // threads round robin: the points with many open views (all points of a wave whose box straddles a jump of the cell order) would
    // otherwise hold their whole wave back.  A thread votes into the bins of the pair's point with LDS atomics; the view records are
    // per-lane loads.  r3: 50 us (every deferred point over all 64 views, wave-uniform records) -> 35 us (a thread walks the open views
    // of its own point) -> this form, at C3.
    extern __shared__ uint32_t lds_u32[];
    uint32_t* lutw = lds_u32;                                             // lut[256] then inv[256] (bytes)
    uint32_t* spre = lutw + 128;                                          // [F3D_BLOCK + 4] exclusive prefix of the open views per point, wave totals
    uint32_t* snv = spre + F3D_BLOCK + 4;                                 // [F3D_BLOCK] votes cast
    uint32_t* sdefer = snv + F3D_BLOCK;                                   // [F3D_BLOCK] the point needed the reference's arithmetic (diagnostic)
    unsigned long long* smask = reinterpret_cast<unsigned long long*>(sdefer + F3D_BLOCK);   // [F3D_BLOCK] open views of the current group
    double* spt = reinterpret_cast<double*>(smask + F3D_BLOCK);          // [3][F3D_BLOCK] the points
    uint32_t* hist = reinterpret_cast<uint32_t*>(spt + 3 * F3D_BLOCK);   // [words][F3D_BLOCK]
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(lutw);
    const uint8_t* inv = lut + 256;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int words = cb->words;
    const size_t plane = f3d_coded_plane(H, W);
    const int wt = f3d_coded_pitch(W);
    const unsigned none_off = 0u;                                         // a border tile: "no sample"
    const double umax = (double)(W > H ? W : H);
    const int ngroups = (nviews + 63) >> 6;
    if (tid < 128) lutw[tid] = reinterpret_cast<const uint32_t*>(cb->lut)[tid];
    const int count = (int)*in_count;
    for (int base = blockIdx.x * F3D_BLOCK; base < count; base += gridDim.x * F3D_BLOCK) {     // block-uniform bounds: every thread meets every barrier
        const int k = base + tid;
        const bool live = k < count;
        const int entry = live ? in_list[k] : 0;
        const int src = entry & 0x7fffffff;                                       // index into xyz as k_fuse saw it
        const bool full = (entry < 0) | (k >= park_slots) | (umask == nullptr);
        {
            f3d_p3 p = {0.0, 0.0, 0.0};
            if (live) p = load_point(xyz, (int64_t)src);
            spt[tid] = p.x; spt[F3D_BLOCK + tid] = p.y; spt[2 * F3D_BLOCK + tid] = p.z;
            sdefer[tid] = 0u;
            unsigned nv0 = 0;
            if (live && !full) {                                                  // the votes of the views k_fuse proved
                const uint32_t* pk = park + (size_t)k * park_stride;
                for (int wd = 0; wd < words; ++wd) {
                    const uint32_t x = pk[wd];
                    hist[wd * F3D_BLOCK + tid] = x;
                    nv0 = __builtin_amdgcn_sad_u8(x, 0u, nv0);                    // (byte 0 of word 0, "no sample", is parked as 0)
                }
            } else {
                for (int wd = 0; wd < words; ++wd) hist[wd * F3D_BLOCK + tid] = 0u;
            }
            snv[tid] = nv0;
        }
        for (int g = 0; g < ngroups; ++g) {
            const int nvg = min(64, nviews - 64 * g);
            unsigned long long m = !live ? 0ull : full ? (nvg == 64 ? ~0ull : (1ull << nvg) - 1ull) : umask[(size_t)k * ngroups + g];
#if defined(F3D_EXP_MID) && F3D_EXP_MID == 1          // timing experiment: no view work at all
            m = 0ull;
#elif defined(F3D_EXP_MID) && F3D_EXP_MID == 2        // timing experiment: one view per point at most
            m &= 0ull - m;
#endif
            // exclusive prefix of the open-view counts over the block
            const unsigned c = (unsigned)__builtin_popcountll(m);
            unsigned incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(incl, off, 64); incl += lane >= off ? o : 0u; }
            __syncthreads();                                              // (the previous round's readers of smask / spre are done)
            smask[tid] = m;
            if (lane == 63) spre[F3D_BLOCK + wv] = incl;
            __syncthreads();
            unsigned wbase = 0, total = 0;
#pragma unroll
            for (int w = 0; w < F3D_BLOCK / 64; ++w) { const unsigned t = spre[F3D_BLOCK + w]; wbase += w < wv ? t : 0u; total += t; }
            spre[tid] = wbase + incl - c;
            __syncthreads();
            unsigned pend_code = F3D_CODE_NONE;
            int pend_pt = 0;
            auto cast = [&](int pt, unsigned b) {                             // several threads may vote for one point: LDS atomics
                if (b != F3D_CODE_NONE) { atomicAdd(&snv[pt], 1u); atomicAdd(&hist[(b >> 2) * F3D_BLOCK + pt], 1u << ((b & 3u) * 8u)); }
            };
            // one (point, view) pair: the decision of this tier, then the gather of the pair's code (voted one pair later)
            auto pair = [&](int pt, const f3d_p3& p, int v, const f3d_view& vw, bool on) {
                bool hit = false;
                int iu = 0, iv = 0;
                const double pscale = (fabs(p.x) + fabs(p.y)) + fabs(p.z);
                const bool small = on & (pscale < 1.0e30);
                bool exact = on & !small;                                         // no usable float32 coordinates
                bool maybe, sure;
                cull_point32(load_cull(vw), (float)p.x, (float)p.y, (float)p.z, (float)pscale, small, maybe, sure);
                bool inside = small & sure;
                if (small & maybe & !sure) {                                      // inside the float32 margin: decide with float64 FMAs
                    bool m64, s64;
                    cull_point64(vw, p, pscale, m64, s64);
                    inside = s64;
                    exact = exact | (m64 & !s64);                                 // within rounding of the plane itself
                }
                if (inside) {
                    bool unsure;
                    hit = project_fast(vw, umax, p, W, H, iu, iv, unsure);
                    exact = exact | unsure;
                }
                if (exact) {                                                      // rare: the reference's arithmetic decides this pair
                    hit = false;
                    if (f3d_inside_view(vw, p)) {
                        double fu, fv;
                        project_exact(vw, p, fu, fv);
                        if (fu >= 0.0 && fu < (double)W && fv >= 0.0 && fv < (double)H) { hit = true; iu = (int)fu; iv = (int)fv; }   // NaN compares false
                    }
                    atomicOr(&sdefer[pt], 1u);
                }
                cast(pend_pt, pend_code);
                pend_code = (cmasks + (size_t)v * plane)[hit ? mask_offset<true>(iu, iv, wt) : none_off];
                pend_pt = pt;
            };
            if (total > (unsigned)F3D_BLOCK * F3D_MID_DENSE) {
                // DENSE (block-uniform decision): most points want most views -- a view-chunked call, a cloud full of unusable points, more
                // view groups than masks are kept for.  Every thread keeps its own point and the views run in the outer loop: their records
                // are wave-uniform again (scalar loads); dealing 64 pairs per point with per-lane records took 242 us where this takes ~50.
                const f3d_p3 p = {spt[tid], spt[F3D_BLOCK + tid], spt[2 * F3D_BLOCK + tid]};
                for (int bit = 0; bit < nvg; ++bit) {
                    const bool on = (m >> bit) & 1ull;
                    if (!__any(on)) continue;                                     // wave-uniform
                    pair(tid, p, 64 * g + bit, views[64 * g + bit], on);
                }
            } else {
                for (unsigned j = tid; j < total; j += F3D_BLOCK) {
                    int lo = 0;                                                   // the last point whose prefix is <= j (points without open views are skipped by it)
#pragma unroll
                    for (int step = F3D_BLOCK / 2; step >= 1; step >>= 1) lo += (spre[lo + step] <= j) ? step : 0;
                    unsigned long long mm = smask[lo];
                    for (unsigned r = j - spre[lo]; r > 0; --r) mm &= mm - 1ull;
                    const int v = 64 * g + __builtin_ctzll(mm);
                    const f3d_p3 p = {spt[lo], spt[F3D_BLOCK + lo], spt[2 * F3D_BLOCK + lo]};
                    pair(lo, p, v, views[v], true);
                }
            }
            cast(pend_pt, pend_code);
        }
        __syncthreads();                                                  // every vote is in
        const int orig = (live && perm && !gather_xyz) ? perm[src] : src;
        bool bad = false;
        const bool trusted = finish_coded<WRITE_VOTES, true>(snv[tid], hist + tid, words, lut, inv, nfilter, fcls, nclasses, threshold,
                                                             live, orig, classes, votes_out, bad);
        const bool defer = live & !trusted;                               // an 8-bit bin wrapped: more than 255 views
        if (defer && out_list) out_list[atomicAdd(out_count, 1u)] = src;
        if (live && diag_count && sdefer[tid] != 0u) atomicAdd(diag_count, 1u);
        if (bad & !defer) atomicOr(err, F3D_DEVERR_FUSE);
        __syncthreads();                                                  // (the next round overwrites the bins)
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

// Audit of the accelerators against the exact arithmetic, every (point, view) pair of a CELL-SORTED cloud: a wave of 64
// consecutive points forms the box exactly as in k_fuse.  stats[0] pairs inside the frustum, stats[1] pairs the offset
// projection does not decide (bound, or no usable centre row), stats[2] decided pairs whose pixel differs from the
// canonical path (must stay 0), stats[3] float32 cull decisions (point or box) the exact plane test contradicts (0).
template <typename T>
__global__ __launch_bounds__(F3D_BLOCK) void k_fastpath_audit(const T* __restrict__ xyz, int64_t n,
                                                               const f3d_view* __restrict__ views, int nviews, int W, int H,
                                                               unsigned long long* __restrict__ stats) {
    unsigned long long pairs = 0, fallback = 0, wrong = 0, cullwrong = 0;
    const int lane = threadIdx.x & 63;
    const int64_t ntiles = (n + F3D_BLOCK - 1) / F3D_BLOCK;
    const float umaxf = (float)(W > H ? W : H);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t i = tile * F3D_BLOCK + threadIdx.x;
        const bool live = i < n;
        f3d_p3 p = {0.0, 0.0, 0.0};
        if (live) p = load_point(xyz, i);
        const double pscale = (fabs(p.x) + fabs(p.y)) + fabs(p.z);
        const bool small = pscale < 1.0e30;
        const float px32 = (float)p.x, py32 = (float)p.y, pz32 = (float)p.z;
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
        const float dx32 = (float)(p.x - (double)c0), dy32 = (float)(p.y - (double)c1), dz32 = (float)(p.z - (double)c2);
        for (int g = 0; g < (nviews + 63) / 64; ++g) {
            const int vj = 64 * g + lane;
            centre_row mine = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0, 0};
            bool row_ok = false, box_out = false, box_in = true;
            if (vj < nviews && wave_any) {
                const f3d_view& vw = views[vj];
                const float marg = 2.0f * __builtin_fmaf(vw.cull_rel32, ps_box, vw.cull_abs32);
                for (int m = 0; m < F3D_NPLANES; ++m) {
                    const float n0 = vw.cull_n32[m][0], n1 = vw.cull_n32[m][1], n2 = vw.cull_n32[m][2];
                    const float base = __builtin_fmaf(n0, c0, __builtin_fmaf(n1, c1, __builtin_fmaf(n2, c2, -vw.cull_off32[m])));
                    const float spread = __builtin_fmaf(fabsf(n0), e0, __builtin_fmaf(fabsf(n1), e1, fabsf(n2) * e2));
                    box_out = box_out | (base + spread < -marg);
                    box_in = box_in & (base - spread > marg);
                }
                row_ok = centre_precompute(reinterpret_cast<const double*>(&vw), 1, c0, c1, c2, e0, e1, e2, umaxf, f3d_coded_pitch(W), mine);
            }
            const unsigned long long row_m = __ballot(row_ok), out_m = __ballot(box_out), in_m = __ballot(box_in && !box_out);
            for (int bit = 0; bit < 64 && 64 * g + bit < nviews; ++bit) {
                const f3d_view& vw = views[64 * g + bit];
                const bool in_exact = live && f3d_inside_view(vw, p);
                bool maybe, sure;
                cull_point32(load_cull(vw), px32, py32, pz32, (float)pscale, small, maybe, sure);
                if (live && ((sure && !in_exact) || (!maybe && in_exact))) ++cullwrong;
                if (live && small && wave_any && ((((out_m >> bit) & 1ull) && in_exact) || (((in_m >> bit) & 1ull) && !in_exact))) ++cullwrong;
                if (!in_exact) continue;
                ++pairs;
                double eu, ev;
                project_exact(vw, p, eu, ev);
                const bool ehit = (eu >= 0.0) & (eu < (double)W) & (ev >= 0.0) & (ev < (double)H);
                if (!((row_m >> bit) & 1ull) || !small) { ++fallback; continue; }
                centre_row r = read_row(mine, bit);
                r.U8 = __builtin_amdgcn_readlane(mine.U8, bit); r.V8 = __builtin_amdgcn_readlane(mine.V8, bit);
                int iu, iv;
                const bool safe = project_offset(r, dx32, dy32, dz32, iu, iv);
                if (!safe) { ++fallback; continue; }
                if (((unsigned)iu < (unsigned)W) & ((unsigned)iv < (unsigned)H)) {      // the tile arithmetic of k_fuse gives the same address
                    const unsigned off = r.obase + rel_offset(iu - r.U8, iv - r.V8, 64 * f3d_coded_pitch(W) - 64);
                    if (off != mask_offset<true>(iu, iv, f3d_coded_pitch(W))) ++wrong;
                }
                const bool hit = ((unsigned)iu < (unsigned)W) & ((unsigned)iv < (unsigned)H);
                if (hit != ehit || (hit && !((double)iu == eu && (double)iv == ev))) ++wrong;
            }
        }
    }
    atomicAdd(&stats[0], pairs); atomicAdd(&stats[1], fallback); atomicAdd(&stats[2], wrong); atomicAdd(&stats[3], cullwrong);
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
#ifndef F3D_FUSE_GRID
#define F3D_FUSE_GRID (256 * 4 * 8)     // k_fuse: blocks per launch (multiple of 8; several rounds so that the tail stays short)
#endif

static size_t fuse_lds_bytes(int hist_dwords_per_point, int ppl, bool tables = true) {   // k_fuse: [tables +] code book + histograms of `ppl` points per lane
    return ((tables ? 64 * F3D_CULL_ROW + (F3D_VTAB_GLOBAL ? 0 : 2 * F3D_VHEAD * 64) : 0) + F3D_BOOK_DWORDS) * sizeof(uint32_t) + (size_t)hist_dwords_per_point * ppl * F3D_BLOCK * sizeof(uint32_t);
}
size_t f3d_fuse_carry_bytes(int64_t n, int nclasses) {     // packed bins of every point slot of the 256- or 512-point tiles
    const size_t words_max = (size_t)((nclasses + 1 + 2 + 3) >> 2);
    return (size_t)((n + 2 * F3D_BLOCK - 1) / (2 * F3D_BLOCK)) * 2 * F3D_BLOCK * words_max * sizeof(uint32_t);
}
size_t f3d_fuse_tables_bytes(int nviews) { return (size_t)((nviews + 63) / 64) * 64 * (24 * sizeof(float) + F3D_VHEAD * sizeof(double)); }

size_t f3d_fuse_lds_bytes(int mode, int nclasses) {                 // LDS of k_fuse_exact
    const int ncols = nclasses + 1;
    const int per_word = (mode == MODE_HIST8) ? 4 : 2;
    return (size_t)((ncols + 1 + per_word - 1) / per_word) * F3D_BLOCK * sizeof(uint32_t);
}

int f3d_fuse_pick_mode(int nviews, int nfilter, bool want_votes) {
    (void)nfilter; (void)want_votes;
    return nviews <= 255 ? MODE_HIST8 : MODE_HIST16;
}

size_t f3d_coded_masks_bytes(int nviews, int h, int w) { return (size_t)nviews * f3d_coded_plane(h, w); }
extern "C" size_t f3d_coded_plane_bytes(int h, int w) { return (h > 0 && w > 0) ? f3d_coded_plane(h, w) : 0; }

// which book: the filter's own labels when there are few of them (nothing else is ever looked at, voting.py:121-124);
// otherwise the labels that occur in the masks
static int pick_book(const f3d_filter_args& flt, bool want_votes) {
    int distinct = 0;
    if (flt.nfilter > 0 && flt.nfilter <= 8 && !want_votes) {
        for (int k = 0; k < flt.nfilter; ++k) {
            bool seen = false;
            for (int j = 0; j < k; ++j) seen = seen || flt.cls[j] == flt.cls[k];
            distinct += seen ? 0 : 1;
        }
    }
    return (distinct > 0) ? 2 : 1;
}

// labels present in `nbytes` mask bytes -> cb->presence.  The set is OR-ed into: it is zero when the context is created and whoever
// consumes it (k_code_lut, k_presence_bytes) leaves it zero again -- one 5 us fill kernel less per call.  (A call that failed between
// the two would leave a superset behind: a few unused vote bins in the next call, never a wrong label.)
hipError_t f3d_launch_mask_presence(const uint8_t* src, int64_t nbytes, f3d_codebook* cb, hipStream_t s) {
    if (nbytes <= 0) return hipSuccess;
    const bool vec = !((uintptr_t)src & 7);
    const dim3 g(grid_for(nbytes >> 3, F3D_BLOCK, 256 * 8)), b(F3D_BLOCK);
    if (vec) hipLaunchKernelGGL(k_mask_presence<true>, g, b, 0, s, src, nbytes, cb);
    else hipLaunchKernelGGL(k_mask_presence<false>, g, b, 0, s, src, nbytes, cb);
    return hipGetLastError();
}

// cb->presence <-> 256 bytes of 0 / 1 (the form ranks can combine with an all-reduce MAX: RCCL has no bitwise OR)
hipError_t f3d_launch_presence_bytes(f3d_codebook* cb, uint8_t* bytes256, bool to_bytes, hipStream_t s) {
    hipLaunchKernelGGL(k_presence_bytes, dim3(1), dim3(F3D_BLOCK), 0, s, cb, bytes256, to_bytes ? 1 : 0);
    return hipGetLastError();
}

// the code book from cb->presence (presence book) or from the filter list (filter book)
hipError_t f3d_launch_code_book(f3d_codebook* cb, int nclasses, const f3d_filter_args& flt, bool want_votes, hipStream_t s) {
    if (nclasses < 0 || nclasses > F3D_CODE_MAX_NCLASSES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_code_lut, dim3(1), dim3(F3D_BLOCK), 0, s, cb, nclasses, pick_book(flt, want_votes), flt);
    return hipGetLastError();
}

// `nviews` planes of src -> coded, bordered, tiled planes at dst, with the book as it stands
hipError_t f3d_launch_code_planes(const uint8_t* src, uint8_t* dst, int nviews, int h, int w, const f3d_codebook* cb, hipStream_t s) {
    if (nviews <= 0) return hipSuccess;
    if ((uintptr_t)dst & 7) return hipErrorInvalidValue;
    const int64_t total = (int64_t)nviews * (int64_t)(f3d_coded_plane(h, w) / 8);
    const dim3 g(grid_for(total, F3D_BLOCK, F3D_GRID_CAP)), b(F3D_BLOCK);
    if (!(w & 7) && !((uintptr_t)src & 7)) hipLaunchKernelGGL(k_code_masks<true>, g, b, 0, s, src, dst, nviews, h, w, cb);
    else hipLaunchKernelGGL(k_code_masks<false>, g, b, 0, s, src, dst, nviews, h, w, cb);
    return hipGetLastError();
}

// the one-shot call: presence -> [setup + book in one launch] -> coded planes
hipError_t f3d_launch_code_masks_with_setup(const uint8_t* src, uint8_t* dst, int nviews, int h, int w, int nclasses, const f3d_filter_args& flt,
                                            bool want_votes, f3d_codebook* cb, const f3d_view* views_dev, void* tables, double threshold,
                                            unsigned int* todo_count, hipStream_t s) {
    if (nviews <= 0) return hipSuccess;
    if (nclasses < 0 || nclasses > F3D_CODE_MAX_NCLASSES || ((uintptr_t)dst & 7)) return hipErrorInvalidValue;
    hipError_t e;
    if (pick_book(flt, want_votes) == 1 && (e = f3d_launch_mask_presence(src, (int64_t)nviews * h * w, cb, s)) != hipSuccess) return e;
    if ((e = f3d_launch_fuse_setup(views_dev, 0, nviews, tables, cb, threshold, todo_count, s, nclasses, &flt, want_votes)) != hipSuccess) return e;
    return f3d_launch_code_planes(src, dst, nviews, h, w, cb, s);
}

hipError_t f3d_launch_code_masks(const uint8_t* src, uint8_t* dst, int nviews, int h, int w, int nclasses, const f3d_filter_args& flt,
                                 bool want_votes, f3d_codebook* cb, hipStream_t s) {
    if (nviews <= 0) return hipSuccess;
    if (nclasses < 0 || nclasses > F3D_CODE_MAX_NCLASSES || ((uintptr_t)dst & 7)) return hipErrorInvalidValue;
    hipError_t e;
    if (pick_book(flt, want_votes) == 1 && (e = f3d_launch_mask_presence(src, (int64_t)nviews * h * w, cb, s)) != hipSuccess) return e;
    if ((e = f3d_launch_code_book(cb, nclasses, flt, want_votes, s)) != hipSuccess) return e;
    return f3d_launch_code_planes(src, dst, nviews, h, w, cb, s);
}

// k_fuse_mid: code book, prefix / vote count / flag per point, open-view masks, the points, the bins
static size_t mid_lds_bytes(int words_max) {
    return (size_t)(128 + (F3D_BLOCK + 4) + 2 * F3D_BLOCK) * 4 + (size_t)F3D_BLOCK * 8 + (size_t)3 * F3D_BLOCK * 8 + (size_t)words_max * F3D_BLOCK * 4;
}

// slots of the first deferred list whose points can be parked (bins + view masks); deferred points beyond them are redone from nothing
static int64_t fuse_park_slots(int64_t n, int nviews) {
    if (nviews > 64 * F3D_PART_MAX_GROUPS || nviews <= 0) return 0;
    int64_t k = n / 16 + 4096;
    if (const char* e = getenv("F3D_DEBUG_PARK_SLOTS")) k = atoll(e);   // tests: force the overflow path (deferred points beyond the parked slots)
    return k < n ? (k < 0 ? 0 : k) : n;
}
// 4 counters, two index lists of n entries (fast -> float64 tier -> exact), the view masks and the parked bins of the first list's slots
size_t f3d_fuse_todo_bytes(int64_t n, int nviews, int nclasses) {
    const size_t k = (size_t)fuse_park_slots(n, nviews);
    return 16 + (size_t)n * 8 + k * ((size_t)((nviews + 63) / 64) * 8 + (size_t)((nclasses + 1 + 2 + 3) >> 2) * 4);
}

// The label vector is filled with the label of a point nobody votes for ("unlabelled": nclasses, through the filter remap) by
// streaming stores, and k_fuse scatters only the labels that differ.  A scattered 8-byte store is one write transaction on the fabric
// whatever its size: 10M of them are ~350 us of the memory system's time (96-142 us of the step once overlapped with the view loops,
// profiles/r03_summary.md); the fill is 80 MB at streaming speed.  At the reference's threshold 0.5 most points of a real scene and
// 99.9 % of the synthetic one stay unlabelled; at threshold 0 the fill is 15 us spent for nothing.
__global__ __launch_bounds__(F3D_BLOCK) void k_fill_labels(int64_t* __restrict__ classes, int64_t n, int64_t value) {
    for (int64_t i = (int64_t)blockIdx.x * F3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * F3D_BLOCK)
        __builtin_nontemporal_store(value, &classes[i]);
}

static int64_t unlabelled_after_remap(int nclasses, const f3d_filter_args& flt) {   // segment_point's answer for a point without votes
    int64_t r = nclasses;
    if (flt.nfilter > 0 && !flt.cls_host) return F3D_NO_PREFILL;               // (no host copy of the list: store every label)
    for (int k = 0; k < flt.nfilter; ++k) if (r == k) r = flt.cls_host[k];
    return r;
}

template <typename KernelT>
static hipError_t raise_lds(KernelT kernel, size_t lds) {
    if (lds <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <typename T, bool V, bool CARRY>
static hipError_t launch_fuse_t(const void* xyz, int64_t n, const f3d_view* views_dev, int nviews, const uint8_t* masks, const uint8_t* cmasks,
                                int h, int w, int nclasses, const f3d_filter_args& flt, double threshold, int64_t* classes, uint16_t* votes,
                                int* err, const int32_t* perm, bool gather_xyz, unsigned int* todo_count, int32_t* todo,
                                unsigned int* todo2_count, int32_t* todo2, const f3d_codebook* cb, void* tables, int mode, int grid,
                                int v0, int v1, uint32_t* carry, void* xyz_keep, unsigned long long* umask, uint32_t* park, int park_slots,
                                int park_stride, hipStream_t s) {
    // CARRY: the fast kernel runs over the views [v0, v1) only and parks / resumes the vote bins in `carry`; the float64 tier and
    // the exact kernel follow the last chunk (v1 == nviews) and see every view.  Otherwise v0 = 0, v1 = nviews.
    const bool fast = cmasks != nullptr;                     // no coded masks (nclasses > F3D_CODE_MAX_NCLASSES): exact kernel only
    if (CARRY && (!fast || nviews > 255 || !carry || v0 < 0 || v1 <= v0 || v1 > nviews)) return hipErrorInvalidValue;
    const int chunk_flags = CARRY ? ((v0 > 0 ? 1 : 0) | (v1 < nviews ? 2 : 0)) : 0;
    const f3d_view* cviews = views_dev + v0;
    const uint8_t* ccm = fast ? cmasks + (size_t)v0 * f3d_coded_plane(h, w) : nullptr;
    const int cnv = v1 - v0;
    if (fast && (n > 0x7ffff000LL || (flt.nfilter > 0 && !flt.cls_dev) || (uint64_t)nviews * f3d_coded_plane(h, w) >= (1ull << 32)))
        return hipErrorInvalidValue;                         // 32-bit point indices and mask offsets; filter list in device memory
    const dim3 g(grid), b(F3D_BLOCK), ge(fast ? 512 : grid);
    const int words_max = (nclasses + 1 + 2 + 3) >> 2;      // every label 0..nclasses present, plus the codes "no sample" and "rejected"
    const size_t lds_small = fuse_lds_bytes(F3D_BIN32_MAX_CODES, 2, cnv <= 64), lds_full = fuse_lds_bytes(words_max, 2, false);
    float* ctabT = reinterpret_cast<float*>(tables);
    double* vtabT = reinterpret_cast<double*>(reinterpret_cast<char*>(tables) + (size_t)((cnv + 63) / 64) * 64 * 24 * sizeof(float));
    const size_t lds_exact = f3d_fuse_lds_bytes(mode, nclasses);
    if (lds_exact > 160 * 1024 || lds_full > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e;
    int64_t prefilled = F3D_NO_PREFILL;
    if (fast && !(chunk_flags & 2)) {                        // (a chunk that parks its bins writes no labels)
        prefilled = unlabelled_after_remap(nclasses, flt);
        if (prefilled != F3D_NO_PREFILL)
            hipLaunchKernelGGL(k_fill_labels, dim3(grid_for(n, F3D_BLOCK, 2048)), dim3(F3D_BLOCK), 0, s, classes, n, prefilled);
    }
    if (fast) {
        // the guarded vote: an 8-bit bin of a real code can wrap (more than 255 views), or the "no sample" byte could (a point casts up to
        // nviews + 3 ngroups + 2 votes, placeholders included: from ~240 views on)
        const bool wrap = nviews + 3 * ((nviews + 63) / 64) + 2 > 255;
        // the view tables of ONE 64-view group are staged in LDS; with more views every instance reads the transposed global copies (the
        // tile loop then has no block-wide barrier, and the register allocation does not have to serve a path C3 never takes)
        const bool one_group = cnv <= 64;
        auto ks = one_group ? k_fuse<T, 2, V, true, false, true, CARRY> : k_fuse<T, 2, V, true, false, false, CARRY>;     // dword bins: at most F3D_BIN32_MAX_CODES codes
        // 8-bit bins, 4 per dword, 2 points per lane: <= 48 codes with the view tables in LDS; <= 100 codes / any alphabet with the tables in
        // global memory and 25 / all bin words per point (3 / 2 blocks per CU)
        auto km2 = k_fuse<T, 2, V, false, false, true, CARRY>;
        auto km3 = k_fuse<T, 2, V, false, false, false, CARRY>;
        auto kf = km3;                                           // any alphabet: the same code with room for every bin (2 blocks per CU)
        if (!CARRY && wrap) { km2 = k_fuse<T, 2, V, false, true, true, false>; km3 = kf = k_fuse<T, 2, V, false, true, false, false>; }
        if (!one_group) km2 = km3;
        const size_t lds_large = fuse_lds_bytes(words_max < F3D_PACKED_LARGE_WORDS ? words_max : F3D_PACKED_LARGE_WORDS, 2, false);
        if ((e = raise_lds(ks, lds_small)) != hipSuccess || (e = raise_lds(kf, lds_full > lds_large ? lds_full : lds_large)) != hipSuccess) return e;
        // up to four instances are enqueued (dword bins for tiny alphabets; packed 8-bit bins for up to 48, up to 100 and for any number
        // of codes): LDS per block decides how many blocks a CU holds, and only the device
        // knows how many labels the masks contain -- the code book says which instance runs, the others return at once
        const size_t lds_mid = fuse_lds_bytes(F3D_PACKED_SMALL_WORDS, 2, one_group);
        hipLaunchKernelGGL(ks, g, b, lds_small, s, (const T*)xyz, n, cviews, cnv, ccm, h, w, nclasses, flt.nfilter, flt.cls_dev, threshold,
                           classes, votes, err, perm, gather_xyz ? 1 : 0, todo_count, todo, cb, 0, F3D_BIN32_MAX_CODES, ctabT, vtabT, carry, chunk_flags, (T*)xyz_keep, umask, park, park_slots, park_stride, prefilled);
        if (nclasses + 3 > F3D_BIN32_MAX_CODES)
            hipLaunchKernelGGL(km2, g, b, lds_mid, s, (const T*)xyz, n, cviews, cnv, ccm, h, w, nclasses, flt.nfilter, flt.cls_dev, threshold,
                               classes, votes, err, perm, gather_xyz ? 1 : 0, todo_count, todo, cb, F3D_BIN32_MAX_CODES + 1, 4 * F3D_PACKED_SMALL_WORDS,
                               ctabT, vtabT, carry, chunk_flags, (T*)xyz_keep, umask, park, park_slots, park_stride, prefilled);
        if (nclasses + 3 > 4 * F3D_PACKED_SMALL_WORDS)
            hipLaunchKernelGGL(km3, g, b, lds_large, s, (const T*)xyz, n, cviews, cnv, ccm, h, w, nclasses, flt.nfilter, flt.cls_dev, threshold,
                               classes, votes, err, perm, gather_xyz ? 1 : 0, todo_count, todo, cb, 4 * F3D_PACKED_SMALL_WORDS + 1, 4 * F3D_PACKED_LARGE_WORDS,
                               ctabT, vtabT, carry, chunk_flags, (T*)xyz_keep, umask, park, park_slots, park_stride, prefilled);
        if (nclasses + 3 > 4 * F3D_PACKED_LARGE_WORDS) {
            hipLaunchKernelGGL(kf, g, b, lds_full, s, (const T*)xyz, n, cviews, cnv, ccm, h, w, nclasses, flt.nfilter, flt.cls_dev,
                               threshold, classes, votes, err, perm, gather_xyz ? 1 : 0, todo_count, todo, cb, 4 * F3D_PACKED_LARGE_WORDS + 1, 256,
                               ctabT, vtabT, carry, chunk_flags, (T*)xyz_keep, umask, park, park_slots, park_stride, prefilled);
        }
        if (chunk_flags & 2) return hipGetLastError();       // more view chunks to come
        // float64 tier on the deferred points' open views, the reference's arithmetic for what it cannot prove either.  Beyond 255 views a
        // point whose 8-bit bin wrapped goes on to k_fuse_exact (16-bit bins, raw masks) through a list in the second list's storage.
        const bool wide = nviews > 255;
        if (wide && !masks) return hipErrorInvalidValue;     // (only coded planes: f3d_fuse_chunk_coded_dev, at most 255 views)
        auto km = k_fuse_mid<T, V>;
        const size_t lds_tier2 = mid_lds_bytes(words_max);
        if ((e = raise_lds(km, lds_tier2)) != hipSuccess) return e;
        hipLaunchKernelGGL(km, dim3(1024), b, lds_tier2, s, (const T*)xyz, todo_count, todo, views_dev, nviews, cmasks, h, w, nclasses, flt.nfilter,
                           flt.cls_dev, threshold, classes, votes, err, perm, gather_xyz ? 1 : 0, wide ? todo_count + 2 : (unsigned int*)nullptr,
                           wide ? todo2 : (int32_t*)nullptr, todo2_count, cb,
                           (const unsigned long long*)umask, (const uint32_t*)park, park_slots, park_stride);
        if (!wide) return hipGetLastError();
    }
    const unsigned int* exact_count = fast ? todo_count + 2 : todo2_count;   // the list k_fuse_exact works on (NULL list: every point)
    const int32_t* exact_list = fast ? todo2 : nullptr;
    if (mode == MODE_HIST8) {
        auto ke = k_fuse_exact<T, MODE_HIST8, V>;
        if ((e = raise_lds(ke, lds_exact)) != hipSuccess) return e;
        hipLaunchKernelGGL(ke, ge, b, lds_exact, s, (const T*)xyz, n, exact_count, exact_list, views_dev, nviews, masks, h, w, nclasses,
                           flt, threshold, classes, votes, err, perm, gather_xyz ? 1 : 0);
    } else {
        auto ke = k_fuse_exact<T, MODE_HIST16, V>;
        if ((e = raise_lds(ke, lds_exact)) != hipSuccess) return e;
        hipLaunchKernelGGL(ke, ge, b, lds_exact, s, (const T*)xyz, n, exact_count, exact_list, views_dev, nviews, masks, h, w, nclasses,
                           flt, threshold, classes, votes, err, perm, gather_xyz ? 1 : 0);
    }
    return hipGetLastError();
}

// Before f3d_launch_fuse, on any stream that is joined into its stream: the threshold table of the code book (segment_point) and the
// transposed per-view tables of the views [v0, v1) the large-alphabet instances read (tables: f3d_fuse_tables_bytes(nviews) of scratch).
// todo_count (the 4 counters of the deferred lists; NULL for a later chunk of a view-chunked call) is zeroed here.
hipError_t f3d_launch_fuse_setup(const f3d_view* views_dev, int v0, int v1, void* tables, f3d_codebook* cb, double threshold,
                                 unsigned int* todo_count, hipStream_t s, int book_nclasses, const f3d_filter_args* book_flt, bool book_want_votes) {
    // book_flt != NULL: the code book (f3d_launch_code_book's work) is built by an extra block of the same launch
    const int cnv = v1 - v0;
    if (cnv <= 0) return hipSuccess;
    float* ctabT = reinterpret_cast<float*>(tables);
    double* vtabT = reinterpret_cast<double*>(reinterpret_cast<char*>(tables) + (size_t)((cnv + 63) / 64) * 64 * 24 * sizeof(float));
    f3d_filter_args none; none.nfilter = 0; none.cls_dev = nullptr; none.cls_host = nullptr; for (int k = 0; k < 8; ++k) none.cls[k] = -1;
    if (book_flt && (book_nclasses < 0 || book_nclasses > F3D_CODE_MAX_NCLASSES)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_fuse_setup, dim3(book_flt ? 9 : 8), dim3(F3D_BLOCK), 0, s, views_dev + v0, cnv, ctabT, vtabT, cb, threshold, todo_count,
                       book_nclasses, book_flt ? pick_book(*book_flt, book_want_votes) : -1, book_flt ? *book_flt : none);
    return hipGetLastError();
}

hipError_t f3d_launch_fuse(const void* xyz, int dtype, int64_t n, const f3d_view* views_dev, int nviews,
                           const uint8_t* masks, const uint8_t* cmasks, int h, int w, int nclasses, const f3d_filter_args& flt,
                           double threshold, int64_t* classes, uint16_t* votes, int* err, const int32_t* perm, bool gather_xyz,
                           unsigned int* todo_count, int32_t* todo, const f3d_codebook* cb, void* tables, int v0, int v1, uint32_t* carry,
                           void* xyz_keep, hipStream_t s) {
    // carry == NULL: one launch over all views (v0, v1 ignored).  Otherwise the views [v0, v1) of a view-chunked call: `carry` holds
    // f3d_fuse_carry_bytes(n, nclasses) of device scratch that must survive from the chunk with v0 == 0 to the one with v1 == nviews;
    // chunks in ascending order without gaps; no vote output.  xyz_keep (may be NULL; first chunk with gather_xyz only): n points of
    // xyz's type, receives the cloud in perm order -- the caller passes it as xyz (gather_xyz = false, same perm) from then on.
    // tables: f3d_fuse_tables_bytes(nviews) of device scratch.  todo_count points at 4 counters (first list, second list, 2 spare) followed by the two index lists of n entries each
    if (n <= 0) return hipSuccess;
    unsigned int* todo2_count = todo_count + 1;
    int32_t* todo2 = todo + n;
    // behind the two lists (f3d_fuse_todo_bytes): per slot of the first list the masks of the views still to be decided, then the parked bins
    const int park_slots = (int)fuse_park_slots(n, nviews), park_stride = (nclasses + 1 + 2 + 3) >> 2, ngroups = (nviews + 63) / 64;
    unsigned long long* umask = reinterpret_cast<unsigned long long*>(todo + 2 * n);
    uint32_t* park = reinterpret_cast<uint32_t*>(umask + (size_t)park_slots * ngroups);
    const int mode = f3d_fuse_pick_mode(nviews, flt.nfilter, votes != nullptr);      // bins of the exact kernel; the fast one uses 8 bits
    const int64_t ntiles = (n + F3D_BLOCK * 2 - 1) / (F3D_BLOCK * 2);          // k_fuse: 2 points per lane
    int grid = (int)(ntiles < F3D_FUSE_GRID ? ntiles : F3D_FUSE_GRID);
    grid = (grid + 7) & ~7;                                  // the XCD-aware tile mapping needs a multiple of 8 blocks
    if (carry && votes) return hipErrorInvalidValue;
    if (!carry) { v0 = 0; v1 = nviews; }
    // (with coded masks the counters of the deferred lists were zeroed by f3d_launch_fuse_setup, which must precede this call)
#define F3D_ARGS xyz, n, views_dev, nviews, masks, cmasks, h, w, nclasses, flt, threshold, classes, votes, err, perm, gather_xyz, todo_count, todo, todo2_count, todo2, cb, tables, mode, grid, v0, v1, carry, xyz_keep, umask, park, park_slots, park_stride, s
    if (carry) return dtype == F3D_F64 ? launch_fuse_t<double, false, true>(F3D_ARGS) : launch_fuse_t<float, false, true>(F3D_ARGS);
    if (dtype == F3D_F64) return votes ? launch_fuse_t<double, true, false>(F3D_ARGS) : launch_fuse_t<double, false, false>(F3D_ARGS);
    return votes ? launch_fuse_t<float, true, false>(F3D_ARGS) : launch_fuse_t<float, false, false>(F3D_ARGS);
#undef F3D_ARGS
}

hipError_t f3d_launch_fastpath_audit(const void* xyz, int dtype, int64_t n, const f3d_view* views_dev, int nviews, int w, int h,
                                     unsigned long long* stats_dev, hipStream_t s) {
    hipError_t e = hipMemsetAsync(stats_dev, 0, 4 * sizeof(unsigned long long), s);
    if (e != hipSuccess || n <= 0) return e;
    const dim3 g(grid_for(n, F3D_BLOCK, F3D_GRID_CAP)), b(F3D_BLOCK);
    if (dtype == F3D_F64) hipLaunchKernelGGL(k_fastpath_audit<double>, g, b, 0, s, (const double*)xyz, n, views_dev, nviews, w, h, stats_dev);
    else hipLaunchKernelGGL(k_fastpath_audit<float>, g, b, 0, s, (const float*)xyz, n, views_dev, nviews, w, h, stats_dev);
    return hipGetLastError();
}
