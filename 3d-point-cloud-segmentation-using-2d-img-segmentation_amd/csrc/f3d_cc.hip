// Same-class connected components for split_into_instances (reference segUtils/cv.py:425-440, 473-499).
//
// The reference flood-fills from the lowest remaining point index through neighbours of the same class.  On a
// symmetric adjacency (the only producer, KDTree.query_radius at fusion.py:369-377, is symmetric) the cluster of a
// seed is its connected component in the graph restricted to same-class edges, and "lowest remaining index" is the
// component's minimum index.  That is what this file computes: a lock-free union-find where the larger root is
// always linked under the smaller one (so every component ends rooted at its minimum index), one pass over the
// CSR edges, then a compression pass.  Parent reads are agent-scope relaxed atomic loads (served by L2): a CU's L1 is
// not coherent with other CUs' atomics, and the CAS return value -- always current -- drives the retry.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "f3d.h"
#include "f3d_kernels.h"

namespace {

constexpr int CB = 256;

__device__ __forceinline__ int32_t ld(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int32_t find_root(int32_t* parent, int32_t x) {
    for (;;) {
        const int32_t p = ld(parent + x);
        if (p == x) return x;
        const int32_t gp = ld(parent + p);
        if (gp != p) __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving (any ancestor is valid)
        x = p;
    }
}

__global__ __launch_bounds__(CB) void k_cc_init(int32_t* __restrict__ parent, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * CB + threadIdx.x; i < n; i += (int64_t)gridDim.x * CB) parent[i] = (int32_t)i;
}

__global__ __launch_bounds__(CB) void k_cc_hook(const int64_t* __restrict__ classes, int64_t n, const int64_t* __restrict__ offs,
                                                const int32_t* __restrict__ nbrs, int32_t* parent, int* __restrict__ err) {
    for (int64_t i = (int64_t)blockIdx.x * CB + threadIdx.x; i < n; i += (int64_t)gridDim.x * CB) {
        const int64_t ci = classes[i];
        for (int64_t e = offs[i]; e < offs[i + 1]; ++e) {
            const int64_t j = nbrs[e];
            if (j < 0 || j >= n) { atomicOr(err, F3D_DEVERR_CC); continue; }
            if (j == i || classes[j] != ci) continue;
            int32_t a = (int32_t)i, b = (int32_t)j;
            for (;;) {
                a = find_root(parent, a); b = find_root(parent, b);
                if (a == b) break;
                if (a < b) { const int32_t t = a; a = b; b = t; }            // a = larger root, goes under b
                const int32_t old = atomicCAS(parent + a, a, b);
                if (old == a) break;
                a = old;                                                     // someone else linked a first: continue from there
            }
        }
    }
}

__global__ __launch_bounds__(CB) void k_cc_compress(int32_t* parent, int64_t n, int64_t* __restrict__ root) {
    for (int64_t i = (int64_t)blockIdx.x * CB + threadIdx.x; i < n; i += (int64_t)gridDim.x * CB) {
        int32_t x = (int32_t)i;
        for (;;) { const int32_t p = ld(parent + x); if (p == x) break; x = p; }
        root[i] = x;
    }
}

}  // namespace

hipError_t f3d_launch_components(const int64_t* classes, int64_t n, const int64_t* offs, const int32_t* nbrs, int32_t* parent,
                                 int64_t* root, int* err, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n > 0x7fffffffLL) return hipErrorInvalidValue;
    int64_t gb = (n + CB - 1) / CB;
    const dim3 g((unsigned)(gb < 16384 ? gb : 16384)), b(CB);
    hipLaunchKernelGGL(k_cc_init, g, b, 0, s, parent, n);
    hipLaunchKernelGGL(k_cc_hook, g, b, 0, s, classes, n, offs, nbrs, parent, err);
    hipLaunchKernelGGL(k_cc_compress, g, b, 0, s, parent, n, root);
    return hipGetLastError();
}
