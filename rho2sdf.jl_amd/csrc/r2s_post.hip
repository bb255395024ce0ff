// Post-processing stages of the hot path on gfx950:
//   remove_sdf_artifacts!       (src/SignedDistances/SdfArtifactRemoval.jl:134-245)
//   calculate_volume_from_sdf   (src/SdfSmoothing/CalcVolumeFromSDF.jl:26-125)
//   RBFs_smoothing              (src/SdfSmoothing/RBFs4Smoothing.jl:321-377)
// All grid sweeps, x-fastest, HBM-bound except the quadrature on cut cells.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "r2s_common.hpp"
#include "r2s_internal.hpp"

// ====================================================================================
// connected components of {sdf >= threshold}, 6-connectivity: lock-free union-find with
// the smallest linear index as root (the serial reference semantics, analyze_sdf_components
// :271-285; the reference's threaded union-find is racy).
// ====================================================================================
#define NOLABEL 0xFFFFFFFFu

__device__ __forceinline__ uint32_t uf_load(const uint32_t* L, uint32_t i)
{
    // parents are rewritten by other CUs during the kernel: bypass the per-CU L1
    return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t uf_find(uint32_t* L, uint32_t x)
{
    for (;;) {
        const uint32_t p = uf_load(L, x);
        if (p == x) return x;
        x = p;
    }
}

__device__ __forceinline__ void uf_union(uint32_t* L, uint32_t a, uint32_t b)
{
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }   // link the larger root under the smaller
        const uint32_t old = atomicCAS(&L[a], a, b);
        if (old == a) return;                                // a was still a root
    }
}

// Runs of consecutive voxels along x are components by themselves, so only their first voxels ("heads") take part in
// the union-find: pass 1 points every voxel of a run piece at the piece's head (a piece = a run cut at the boundaries of
// the 64-voxel wavefronts, found with one ballot), pass 2 joins a piece to the piece before it across a wavefront boundary
// and runs to the runs of the next row / plane ONCE per overlap interval (at its first voxel), pass 3 points the heads at
// their roots.  The result - every voxel labelled with the smallest index of its component - is that of one union per
// pair of neighbouring voxels (round 1: three CAS loops per voxel, 5.0 + 5.3 ms at 512^3), with ~100 times fewer
// atomics and no pointer chasing outside the heads.  All three kernels use the same voxel <-> lane mapping.
// piece_len != nullptr: the head of every run piece also gets the piece's length (the single-device labelling sums
// component sizes over heads, ccl_compress_heads_sizes_kernel)
__global__ void __launch_bounds__(256) ccl_init_kernel(const double* __restrict__ sdf, uint32_t n, int nx, double thr, uint32_t* __restrict__ L,
                                                      uint32_t* __restrict__ piece_len = nullptr)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in = v < n && sdf[v] >= thr;
    const unsigned long long m_in = __ballot(in);
    const bool prev = lane > 0 && ((m_in >> (lane - 1)) & 1ull);
    const bool head = in && (lane == 0 || v % (uint32_t)nx == 0u || !prev);
    const unsigned long long m_head = __ballot(head);
    if (v < n) {
        const unsigned long long below = m_head & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));   // heads at or below this lane
        L[v] = in ? v - (uint32_t)lane + (uint32_t)(63 - __clzll((long long)below)) : NOLABEL;
        if (piece_len && head) {
            // the piece ends before the next lane that is outside or a head itself (or with the wavefront)
            const unsigned long long stop = lane == 63 ? 0ull : ((~m_in | m_head) & (~0ull << (lane + 1)));
            piece_len[v] = (uint32_t)((stop ? __ffsll((long long)stop) - 1 : 64) - lane);
        }
    }
}

__global__ void __launch_bounds__(256) ccl_union_kernel(uint32_t* __restrict__ L, int nx, int ny, int nz)
{
    const uint32_t n = (uint32_t)nx * ny * nz;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    // (whether a voxel is labelled never changes; the labels of heads do: those are read inside uf_union only)
    if (L[v] == NOLABEL) return;
    const uint32_t plane = (uint32_t)nx * ny;
    const int i = v % nx, j = (v / nx) % ny, k = v / plane;
    const bool left = i > 0 && L[v - 1] != NOLABEL;
    if (left && (threadIdx.x & 63) == 0) uf_union(L, v, v - 1);   // a piece that continues the run of the previous wavefront
    if (j + 1 < ny && L[v + nx] != NOLABEL && (!left || L[v + nx - 1] == NOLABEL)) uf_union(L, v, v + nx);
    if (k + 1 < nz && L[v + plane] != NOLABEL && (!left || L[v + plane - 1] == NOLABEL)) uf_union(L, v, v + plane);
}

__global__ void __launch_bounds__(256) ccl_compress_heads_kernel(uint32_t* __restrict__ L, uint32_t n, int nx)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    if (L[v] == NOLABEL) return;
    const bool head = (threadIdx.x & 63) == 0 || v % (uint32_t)nx == 0u || L[v - 1] == NOLABEL;
    if (head) {
        const uint32_t r = uf_find(L, v);
        if (r != v) __hip_atomic_store(L + v, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (an ancestor: walks through v stay valid)
    }
}

__global__ void __launch_bounds__(256) ccl_flatten_count_kernel(const uint32_t* __restrict__ L, uint32_t n, uint32_t* __restrict__ root,
                                                               uint32_t* __restrict__ size)
{
    // (after ccl_compress_heads_kernel: L[v] = head of the voxel's run piece, L[head] = root; the loop ends at once)
    // Sizes: a fixed grid strides over the voxels and every wavefront keeps a running (root, count) pair that it only
    // flushes when the root changes - a few thousand same-address atomics instead of one per wavefront of the DATA
    // (420 k on the one big component at 512^3: 5 ms).
    const int lane = threadIdx.x & 63;
    uint32_t cur = NOLABEL, cnt = 0;   // wave-uniform
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += stride) {
        const uint64_t v = base + threadIdx.x;
        uint32_t r = (v < n) ? L[v] : NOLABEL;
        if (r != NOLABEL)
            while (L[r] != r) r = L[r];
        if (v < n) root[v] = r;
        bool pending = r != NOLABEL;
        while (__any(pending)) {
            const unsigned long long todo = __ballot(pending);
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lr = __shfl(r, leader, 64);
            const unsigned long long same = __ballot(pending && r == lr);
            const uint32_t c = (uint32_t)__popcll(same);
            if (lr == cur) {
                cnt += c;
            } else {
                if (cnt && lane == 0) atomicAdd(&size[cur], cnt);
                cur = lr;
                cnt = c;
            }
            if (r == lr) pending = false;
        }
    }
    if (cnt && lane == 0) atomicAdd(&size[cur], cnt);
}

// counters: [0] largest size, [1] smallest root having it, [2] flipped count, [3] interior count
__global__ void ccl_max_kernel(const uint32_t* __restrict__ size, uint32_t n, uint32_t* __restrict__ counters)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n && size[v]) {
        atomicMax(&counters[0], size[v]);
        atomicAdd(&counters[3], size[v]);
    }
}
__global__ void ccl_argmax_kernel(const uint32_t* __restrict__ size, uint32_t n, uint32_t* __restrict__ counters)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n && size[v] == counters[0] && size[v]) atomicMin(&counters[1], v);
}
__global__ void ccl_flip_kernel(double* __restrict__ sdf, const uint32_t* __restrict__ root,
                                const uint32_t* __restrict__ size, uint32_t n, uint32_t largest_root,
                                uint32_t min_size, uint32_t* __restrict__ counters)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t r = root[v];
    if (r != NOLABEL && r != largest_root && size[r] < min_size) {   // SdfArtifactRemoval.jl:220
        sdf[v] = -fabs(sdf[v]);                                       // :234
        atomicAdd(&counters[2], 1u);
    }
}

// ---- single device: the same labelling with less traffic around it ------------------------------------------------
// (the Z-slab version below keeps the kernels above: it merges labels over the slab interfaces through root[] / size[])
// * the heads that are their own root are listed as they are found (a few hundred components), their size counters
//   zeroed there: no 0.5 GB memset, and the largest size / smallest root having it / interior count come from the list
//   in one small kernel instead of two sweeps over 134 M counters;
// * no root[] array: the count and the flip read a voxel's root through L[L[v]] (the second read hits the head's line).
__global__ void __launch_bounds__(256) ccl_compress_heads_roots_kernel(uint32_t* __restrict__ L, uint32_t n, int nx, uint32_t* __restrict__ size,
                                                                      uint32_t* __restrict__ roots, uint32_t roots_cap,
                                                                      uint32_t* __restrict__ nroots)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    if (L[v] == NOLABEL) return;
    const bool head = (threadIdx.x & 63) == 0 || v % (uint32_t)nx == 0u || L[v - 1] == NOLABEL;
    if (head) {
        const uint32_t r = uf_find(L, v);
        if (r != v) {
            __hip_atomic_store(L + v, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (an ancestor: walks through v stay valid)
        } else {
            size[v] = 0u;
            const uint32_t at = atomicAdd(nroots, 1u);
            if (at < roots_cap) roots[at] = v;
        }
    }
}
// ccl_compress_heads_roots_kernel + the component sizes in the same pass: size[] holds the length of every run piece at
// its head (ccl_init_kernel); a head that is not a root adds its length to its root's counter - a fixed grid strides over
// the voxels and every wavefront keeps a running (root, count) pair that it flushes when the root changes, so the one big
// component costs a few thousand same-address atomics, not one per piece - and a root keeps its own length as the start
// of the sum.  No pass over all voxels that resolves every voxel's root just to count it (ccl_count_kernel: 0.29 ms).
__global__ void __launch_bounds__(256) ccl_compress_heads_sizes_kernel(uint32_t* __restrict__ L, uint32_t n, int nx, uint32_t* __restrict__ size,
                                                                      uint32_t* __restrict__ roots, uint32_t roots_cap,
                                                                      uint32_t* __restrict__ nroots)
{
    const int lane = threadIdx.x & 63;
    uint32_t cur = NOLABEL, cnt = 0;   // wave-uniform
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += stride) {
        const uint64_t v64 = base + threadIdx.x;
        const uint32_t v = (uint32_t)v64;
        const uint32_t lv = v64 < n ? L[v] : NOLABEL;
        const uint32_t lp = __shfl_up(lv, 1, 64);   // (lane 0 is a head whatever came before: pieces end with the wavefront)
        const bool head = lv != NOLABEL && (lane == 0 || v % (uint32_t)nx == 0u || lp == NOLABEL);
        uint32_t r = NOLABEL, len = 0;
        if (head) {
            r = uf_find(L, v);
            if (r != v) {
                __hip_atomic_store(L + v, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (an ancestor: walks through v stay valid)
                len = size[v];
            } else {
                const uint32_t at = atomicAdd(nroots, 1u);
                if (at < roots_cap) roots[at] = v;
                r = NOLABEL;   // (its own length is in size[v] already)
            }
        }
        bool pending = r != NOLABEL;
        while (__any(pending)) {
            const unsigned long long todo = __ballot(pending);
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lr = __shfl(r, leader, 64);
            unsigned long long same = __ballot(pending && r == lr);
            uint32_t c = 0;
            for (unsigned long long m = same; m; m &= m - 1) c += __shfl(len, __ffsll((long long)m) - 1, 64);   // (a few heads per wavefront)
            if (lr == cur) {
                cnt += c;
            } else {
                if (cnt && lane == 0) atomicAdd(&size[cur], cnt);
                cur = lr;
                cnt = c;
            }
            if (r == lr) pending = false;
        }
    }
    if (cnt && lane == 0) atomicAdd(&size[cur], cnt);
}
// sizes only (ccl_flatten_count_kernel without the root[] array)
__global__ void __launch_bounds__(256) ccl_count_kernel(const uint32_t* __restrict__ L, uint32_t n, uint32_t* __restrict__ size)
{
    const int lane = threadIdx.x & 63;
    uint32_t cur = NOLABEL, cnt = 0;   // wave-uniform
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += stride) {
        const uint64_t v = base + threadIdx.x;
        uint32_t r = (v < n) ? L[v] : NOLABEL;
        if (r != NOLABEL)
            while (L[r] != r) r = L[r];
        bool pending = r != NOLABEL;
        while (__any(pending)) {
            const unsigned long long todo = __ballot(pending);
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lr = __shfl(r, leader, 64);
            const unsigned long long same = __ballot(pending && r == lr);
            const uint32_t c = (uint32_t)__popcll(same);
            if (lr == cur) {
                cnt += c;
            } else {
                if (cnt && lane == 0) atomicAdd(&size[cur], cnt);
                cur = lr;
                cnt = c;
            }
            if (r == lr) pending = false;
        }
    }
    if (cnt && lane == 0) atomicAdd(&size[cur], cnt);
}
// counters: [0] largest size, [1] smallest root having it, [3] interior count - from the list of roots (one workgroup)
__global__ void __launch_bounds__(1024) ccl_roots_max_kernel(const uint32_t* __restrict__ size, const uint32_t* __restrict__ roots,
                                                            const uint32_t* __restrict__ nroots, uint32_t* __restrict__ counters)
{
    __shared__ uint32_t s_max, s_arg, s_sum;
    if (threadIdx.x == 0) { s_max = 0u; s_arg = NOLABEL; s_sum = 0u; }
    __syncthreads();
    const uint32_t m = *nroots;
    uint32_t mx = 0, sum = 0;
    for (uint32_t q = threadIdx.x; q < m; q += blockDim.x) {
        const uint32_t sz = size[roots[q]];
        mx = sz > mx ? sz : mx;
        sum += sz;
    }
    atomicMax(&s_max, mx);
    atomicAdd(&s_sum, sum);
    __syncthreads();
    const uint32_t gmax = s_max;
    uint32_t arg = NOLABEL;
    for (uint32_t q = threadIdx.x; q < m; q += blockDim.x) {
        const uint32_t r = roots[q];
        if (gmax && size[r] == gmax && r < arg) arg = r;
    }
    atomicMin(&s_arg, arg);
    __syncthreads();
    if (threadIdx.x == 0) { counters[0] = s_max; counters[1] = s_arg; counters[3] = s_sum; }
}
__global__ void ccl_flip_l_kernel(double* __restrict__ sdf, const uint32_t* __restrict__ L, const uint32_t* __restrict__ size, uint32_t n,
                                  uint32_t largest_root, uint32_t min_size, uint32_t* __restrict__ counters)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    bool flip = false;
    if (v < n) {
        uint32_t r = L[v];
        if (r != NOLABEL) {
            while (L[r] != r) r = L[r];   // (head -> root: one step after ccl_compress_heads_roots_kernel)
            flip = r != largest_root && size[r] < min_size;   // SdfArtifactRemoval.jl:220
        }
        if (flip) sdf[v] = -fabs(sdf[v]);                     // :234
    }
    const unsigned long long m = __ballot(flip);
    if (m && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(&counters[2], (uint32_t)__popcll(m));
}

// the work arrays of remove_artifacts_dev, kept between calls per device (r2s_release_cache frees them): allocating and
// freeing 1.6 GB per call cost as much as a kernel of the stage
struct CclWork {
    DevBuf L, size, roots, cnt;
};
static std::mutex g_ccl_mutex;
static std::map<int, CclWork> g_ccl_work;
static void release_ccl_work()
{
    std::lock_guard<std::mutex> lock(g_ccl_mutex);
    for (auto& kv : g_ccl_work) {
        (void)hipSetDevice(kv.first);
        kv.second.L.release(); kv.second.size.release(); kv.second.roots.release(); kv.second.cnt.release();
    }
    g_ccl_work.clear();
}

static int remove_artifacts_dev(double* d_sdf, const r2s_grid* g, double threshold, double min_ratio,
                                hipStream_t st, int64_t* n_flipped)
{
    const int64_t n64 = g->ngp;
    if (n64 <= 0 || n64 >= 0xFFFFFFFFll) return fail(R2S_ERR_ARG, "grid too large for 32-bit labels");
    const uint32_t n = (uint32_t)n64;
    const int nx = (int)g->N[0] + 1, ny = (int)g->N[1] + 1, nz = (int)g->N[2] + 1;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ccl_mutex);   // (one call at a time uses the device's work arrays)
    CclWork& Wk = g_ccl_work[dev];
    DevBuf &L = Wk.L, &size = Wk.size, &roots = Wk.roots, &cnt = Wk.cnt;
    // roots_cap: a component has a head that is its own root; more roots than this (noise at the voxel scale) -> the counters
    // of the overflow are still zeroed, only the list is short: the sweeps over all counters take over (below)
    const uint32_t roots_max = 1u << 22;
    const char* cap_env = getenv("R2S_CCL_ROOTS_CAP");   // (tests: a short list exercises the overflow path)
    const uint32_t roots_cap = cap_env && atoi(cap_env) > 0 ? std::min<uint32_t>((uint32_t)atoi(cap_env), roots_max) : roots_max;
    ENSURE(L, sizeof(uint32_t) * (size_t)n);
    ENSURE(size, sizeof(uint32_t) * (size_t)n);
    ENSURE(roots, sizeof(uint32_t) * (size_t)roots_max);
    ENSURE(cnt, 64);
    const unsigned nb = (n + 255) / 256;
    uint32_t h[8] = {0, NOLABEL, 0, 0, 0, 0, 0, 0};   // [4]: number of roots
    HIP_TRY(hipMemcpyAsync(cnt.p, h, sizeof h, hipMemcpyHostToDevice, st));
    static const bool split_env = getenv("R2S_CCL_SPLIT") && atoi(getenv("R2S_CCL_SPLIT"));   // (tests / A-B: sizes in a pass of their own)
    ccl_init_kernel<<<nb, 256, 0, st>>>(d_sdf, n, nx, threshold, L.as<uint32_t>(), split_env ? nullptr : size.as<uint32_t>());
    ccl_union_kernel<<<nb, 256, 0, st>>>(L.as<uint32_t>(), nx, ny, nz);
    if (split_env) {
        ccl_compress_heads_roots_kernel<<<nb, 256, 0, st>>>(L.as<uint32_t>(), n, nx, size.as<uint32_t>(), roots.as<uint32_t>(), roots_cap,
                                                           cnt.as<uint32_t>() + 4);
        ccl_count_kernel<<<(nb < 4096u ? nb : 4096u), 256, 0, st>>>(L.as<uint32_t>(), n, size.as<uint32_t>());
    } else {
        ccl_compress_heads_sizes_kernel<<<(nb < 8192u ? nb : 8192u), 256, 0, st>>>(L.as<uint32_t>(), n, nx, size.as<uint32_t>(), roots.as<uint32_t>(),
                                                                                  roots_cap, cnt.as<uint32_t>() + 4);
    }
    ccl_roots_max_kernel<<<1, 1024, 0, st>>>(size.as<uint32_t>(), roots.as<uint32_t>(), cnt.as<uint32_t>() + 4, cnt.as<uint32_t>());
    HIP_TRY(hipMemcpyAsync(h, cnt.p, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h[4] > roots_cap) {   // (noise at the voxel scale: more components than the list holds - the sweeps over all counters)
        uint32_t h0[8] = {0, NOLABEL, 0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(cnt.p, h0, sizeof h0, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(size.p, 0, sizeof(uint32_t) * (size_t)n, st));
        ccl_count_kernel<<<(nb < 4096u ? nb : 4096u), 256, 0, st>>>(L.as<uint32_t>(), n, size.as<uint32_t>());
        ccl_max_kernel<<<nb, 256, 0, st>>>(size.as<uint32_t>(), n, cnt.as<uint32_t>());
        ccl_argmax_kernel<<<nb, 256, 0, st>>>(size.as<uint32_t>(), n, cnt.as<uint32_t>());
        HIP_TRY(hipMemcpyAsync(h, cnt.p, sizeof h, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    int64_t flipped = 0;
    if (h[3] != 0) {   // interior_count == 0 -> nothing to do (:150-153)
        // min_component_size = max(1, round(Int, ratio*largest)), Julia round = ties to even (:206)
        long long ms = (long long)std::nearbyint(min_ratio * (double)h[0]);
        if (ms < 1) ms = 1;
        const uint32_t min_size = ms > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)ms;
        ccl_flip_l_kernel<<<nb, 256, 0, st>>>(d_sdf, L.as<uint32_t>(), size.as<uint32_t>(), n, h[1], min_size, cnt.as<uint32_t>());
        HIP_TRY(hipMemcpyAsync(h, cnt.p, sizeof h, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        flipped = h[2];
    }
    if (n_flipped) *n_flipped = flipped;
    return 0;
}

// ====================================================================================
// Gauss-Legendre tables (FastGaussQuadrature.gausslegendre stand-in, host side)
// ====================================================================================
static void gauss_legendre(int n, double* x, double* w)
{
    if (n == 3) {
        x[0] = -std::sqrt(3.0 / 5.0); x[1] = 0.0; x[2] = std::sqrt(3.0 / 5.0);
        w[0] = 5.0 / 9.0; w[1] = 8.0 / 9.0; w[2] = 5.0 / 9.0;
        return;
    }
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < n; ++i) {
        double z = std::cos(pi * ((double)(n - 1 - i) + 0.75) / ((double)n + 0.5));
        double pp = 1.0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= n; ++j) {
                const double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            const double dz = p1 / pp;
            z -= dz;
            if (std::fabs(dz) < 1e-16) break;
        }
        if ((n % 2) && i == n / 2) z = 0.0;
        double p1 = 1.0, p2 = 0.0;
        for (int j = 1; j <= n; ++j) {
            const double p3 = p2;
            p2 = p1;
            p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
        }
        pp = n * (z * p1 - p2) / (z * z - 1.0);
        x[i] = z;
        w[i] = 2.0 / ((1.0 - z * z) * pp * pp);
    }
    for (int i = 0; i < n / 2; ++i) {
        const double xa = 0.5 * (x[n - 1 - i] - x[i]), wa = 0.5 * (w[i] + w[n - 1 - i]);
        x[i] = -xa; x[n - 1 - i] = xa; w[i] = wa; w[n - 1 - i] = wa;
    }
}
extern "C" void r2s_internal_gauss_legendre(int n, double* x, double* w) { gauss_legendre(n, x, w); }

// ====================================================================================
// calculate_volume_from_sdf: 1 thread / cell; full cells add a^3, cut cells run the order^3
// tensor quadrature of the trilinear interpolant (all Float32, same expressions).  Block
// partial sums (Float32) are reduced by a second kernel in a fixed order (the reference's
// Float32 atomics make its own sum order-dependent).
// ====================================================================================
struct QuadTab {
    float gp[32];
    float gw[32];
    int order;
};

__global__ void __launch_bounds__(256) volume_cells_kernel(const float* __restrict__ sdf, int nx, int ny, int nz,
                                                          float shift, float iso, float elvol, float jac,
                                                          QuadTab q, float* __restrict__ partial)
{
    __shared__ float red[256];
    const int64_t ncell = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
    float acc = 0.0f;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(c % (nx - 1)), j = (int)((c / (nx - 1)) % (ny - 1)), k = (int)(c / ((int64_t)(nx - 1) * (ny - 1)));
        const int64_t b = ((int64_t)k * ny + j) * nx + i, sy = nx, sz = (int64_t)nx * ny;
        // `shifted_sdf .= sdf .- th` (RBFs4Smoothing.jl:286) folded into the loads
        const float c000 = sdf[b] - shift, c100 = sdf[b + 1] - shift, c010 = sdf[b + sy] - shift,
                    c110 = sdf[b + sy + 1] - shift, c001 = sdf[b + sz] - shift, c101 = sdf[b + sz + 1] - shift,
                    c011 = sdf[b + sz + sy] - shift, c111 = sdf[b + sz + sy + 1] - shift;
        const float mn = fminf(fminf(fminf(c000, c100), fminf(c010, c110)), fminf(fminf(c001, c101), fminf(c011, c111)));
        const float mx = fmaxf(fmaxf(fmaxf(c000, c100), fmaxf(c010, c110)), fmaxf(fmaxf(c001, c101), fmaxf(c011, c111)));
        if (mx < iso) continue;
        if (mn >= iso) { acc += elvol; continue; }
        float part = 0.0f;
        for (int kq = 0; kq < q.order; ++kq) {
            const float zeta = (q.gp[kq] + 1) / 2;
            for (int jq = 0; jq < q.order; ++jq) {
                const float eta = (q.gp[jq] + 1) / 2;
                for (int iq = 0; iq < q.order; ++iq) {
                    const float xi = (q.gp[iq] + 1) / 2;
                    const float c00 = c000 * (1.0f - xi) + c100 * xi;
                    const float c01 = c001 * (1.0f - xi) + c101 * xi;
                    const float c10 = c010 * (1.0f - xi) + c110 * xi;
                    const float c11 = c011 * (1.0f - xi) + c111 * xi;
                    const float c0 = c00 * (1.0f - eta) + c10 * eta;
                    const float c1 = c01 * (1.0f - eta) + c11 * eta;
                    const float p = c0 * (1.0f - zeta) + c1 * zeta;
                    if (p >= iso) part += q.gw[iq] * q.gw[jq] * q.gw[kq] * jac;
                }
            }
        }
        acc += part;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// (one workgroup of 1 024 threads: the row sums of a volume are summed once per bisection level, and with 256 threads the
// 1 024 serial additions per thread took 0.24 ms of the 1.3 ms of a level.  Fixed order: thread t adds the elements
// t, t + 1024, ... in that order, then the tree.)
__global__ void __launch_bounds__(1024) sum_f32_kernel(const float* __restrict__ in, int n, float* __restrict__ out)
{
    __shared__ float red[1024];
    float acc = 0.0f;
    // (16 loads in flight, the additions in the same order: one load at a time the loop waited 250 ns per element, 65 us
    //  per level of the bisection)
    int i = threadIdx.x;
    for (; i + 15 * 1024 < n; i += 16 * 1024) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = in[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    for (; i < n; i += 1024) acc += in[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

// Row version: one workgroup per (j,k) row of cells, threads along x (coalesced corner loads, no
// integer division).  Full cells are summed per thread; cut cells are listed in LDS in x order and
// then worked off one WAVEFRONT per cell (lanes split the order^3 Gauss points), so a few cut cells
// do not stall 63 idle lanes for 729 iterations.  Every reduction has a fixed order (no float atomics).
// The order^3 quadrature points of a cut cell as a table: {xi, 1-xi, eta, 1-eta, zeta, 1-zeta, weight, -} per point, in
// the order the lanes of volume_rows_kernel visit them.  The values are those the kernel used to form per point and cell
// (index decoding with three integer divisions took more instructions than the interpolation itself).
__global__ void quad_points_kernel(QuadTab q, float jac, float* __restrict__ out)
{
    const int npts = q.order * q.order * q.order;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npts) return;
    const int iq = p % q.order, jq = (p / q.order) % q.order, kq = p / (q.order * q.order);
    const float zeta = (q.gp[kq] + 1) / 2, eta = (q.gp[jq] + 1) / 2, xi = (q.gp[iq] + 1) / 2;
    float* o = out + (size_t)p * 8;
    o[0] = xi; o[1] = 1.0f - xi; o[2] = eta; o[3] = 1.0f - eta; o[4] = zeta; o[5] = 1.0f - zeta;
    o[6] = q.gw[iq] * q.gw[jq] * q.gw[kq] * jac;
    o[7] = 0.0f;
}

// smallest / largest corner value over the cells of every 64-cell segment of every cell row: the level bisection
// evaluates the volume of the same field at up to 40 levels, and a segment whose values all lie on one side of the
// level needs no loads (volume_rows_kernel)
__global__ void __launch_bounds__(256) volume_seg_minmax_kernel(const float* __restrict__ sdf, int nx, int ny, int nz,
                                                               float* __restrict__ segmn, float* __restrict__ segmx, int row0)
{
    const int row = row0 + blockIdx.x;
    const int j = row % (ny - 1), k = row / (ny - 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t sy = nx, sz = (int64_t)nx * ny;
    const int nseg = (nx - 1 + 63) / 64;
    for (int sg = wave; sg < nseg; sg += 4) {
        const int i = sg * 64 + lane;
        float mn = INFINITY, mx = -INFINITY;
        if (i < nx - 1) {
            const int64_t b = ((int64_t)k * ny + j) * nx + i;
            const float c000 = sdf[b], c100 = sdf[b + 1], c010 = sdf[b + sy], c110 = sdf[b + sy + 1], c001 = sdf[b + sz],
                        c101 = sdf[b + sz + 1], c011 = sdf[b + sz + sy], c111 = sdf[b + sz + sy + 1];
            mn = fminf(fminf(fminf(c000, c100), fminf(c010, c110)), fminf(fminf(c001, c101), fminf(c011, c111)));
            mx = fmaxf(fmaxf(fmaxf(c000, c100), fmaxf(c010, c110)), fmaxf(fmaxf(c001, c101), fmaxf(c011, c111)));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, off, 64));
            mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        }
        if (lane == 0) {
            segmn[(size_t)row * nseg + sg] = mn;
            segmx[(size_t)row * nseg + sg] = mx;
        }
    }
}

// sum over the lanes of a wavefront in a fixed order (xor butterfly: every lane ends with the same value)
__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// S(segment) for a segment whose cells are all full: the butterfly of `elvol` over its cells
__global__ void volume_cfull_kernel(int nx, float elvol, float* __restrict__ cfull)
{
    const int lane = threadIdx.x & 63;
    const int nseg = (nx - 1 + 63) / 64, nlast = (nx - 1) - 64 * (nseg - 1);
    const float a = wave_sum_f32(elvol), b = wave_sum_f32(lane < nlast ? elvol : 0.0f);
    if (lane == 0) { cfull[0] = a; cfull[1] = b; }
}

// Row version: one WAVEFRONT per (j,k) row of cells, 64-cell segments along x (coalesced corner loads, no integer
// division, no workgroup barrier: rows are independent, so the wavefronts of a CU overlap each other's loads and
// quadratures).  Per segment S = the butterfly sum over its cells of {a^3 for a full cell, the quadrature sum for a cut
// cell, 0}; the row's value is the sum of its S in x order.  A cut cell's order^3 Gauss points are split over the lanes
// (point p to lane p mod 64, summed per lane in p order, then the butterfly).  Every reduction has a fixed order (no
// float atomics), and a segment's S does not depend on how it was decided.
// segmn / segmx (volume_seg_minmax_kernel, optional): rounding is monotonic, so (segment max - shift) < iso puts every
// cell of the segment outside and (segment min - shift) >= iso makes every cell full - the same decisions the cells
// would take one by one, without their loads.
// rows / nlist (optional): the rows to work on (level bisection: the rows that can still change, volume_narrow_kernel);
// else the rows row0 .. row0 + nrows - 1 (a Z-slab of a multi-device run works on the rows of its planes, with `sdf`,
// `partial` and the segment arrays addressed as the whole grid's).
__global__ void __launch_bounds__(256) volume_rowwave_kernel(const float* __restrict__ sdf, int nx, int ny, int nz,
                                                            float shift, float iso, float elvol, float jac, QuadTab q,
                                                            float* __restrict__ partial, int row0, int nrows,
                                                            const float* __restrict__ segmn, const float* __restrict__ segmx,
                                                            const float4* __restrict__ qpts, const int* __restrict__ rows,
                                                            const uint32_t* __restrict__ nlist)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t sy = nx, sz = (int64_t)nx * ny;
    const int n = q.order, nn = n * n, npts = nn * n;
    const int nseg = (nx - 1 + 63) / 64;
    const bool segs = segmn != nullptr;
    // the tensor form of a cut cell's quadrature (order <= 9: its tables fit the LDS arrays below): the trilinear value at
    // point (iq, jq, kq) is a chain of seven linear interpolations, of which only the last depends on kq and the first four
    // only on iq - formed once per cell (4 n along x, 2 n^2 along y, n^3 along z) instead of once per point (7 n^3), from the
    // same operands by the same expressions: the same values, a fifth of the arithmetic
    const bool tensor = qpts != nullptr && npts <= 12 * 64 && 4 * n <= 64;
    __shared__ float sW[12 * 64];      // weight per quadrature point (kq n^2 + jq n + iq)
    __shared__ float sG[32];           // (gp + 1) / 2 per index
    if (tensor) {
        for (int p = tid; p < npts; p += 256) sW[p] = qpts[2 * p + 1].z;   // (zeta, 1 - zeta, weight, -)
        if (tid < n) sG[tid] = (q.gp[tid] + 1) / 2;
        __syncthreads();
    }
    const float cfull64 = wave_sum_f32(elvol);
    const int total = rows ? (int)*nlist : nrows;
    for (int idx = (int)blockIdx.x * 4 + wave; idx < total; idx += (int)gridDim.x * 4) {
        const int row = rows ? rows[idx] : row0 + idx;
        const int j = row % (ny - 1), k = row / (ny - 1);
        float psum = 0.0f;
        for (int sg0 = 0; sg0 < nseg; sg0 += 64) {
            // the states of (up to) 64 segments at once: 0 outside, 1 full, 2 look at the cells
            int st = 0;
            if (sg0 + lane < nseg) {
                st = 2;
                if (segs) {
                    const float smn = segmn[(size_t)row * nseg + sg0 + lane] - shift, smx = segmx[(size_t)row * nseg + sg0 + lane] - shift;
                    st = (smx < iso) ? 0 : (smn >= iso ? 1 : 2);
                }
            }
            const uint64_t mfull = __ballot(st == 1), mcut = __ballot(st == 2);
            uint64_t mwork = mfull | mcut;
            while (mwork) {   // the segments that add something, in x order
                const int sl = __ffsll((long long)mwork) - 1;
                mwork &= mwork - 1;
                const int sg = sg0 + sl;
                const int i = sg * 64 + lane;
                const bool cell = i < nx - 1;
                if ((mfull >> sl) & 1ull) {
                    psum += (sg * 64 + 64 <= nx - 1) ? cfull64 : wave_sum_f32(cell ? elvol : 0.0f);
                    continue;
                }
                float v = 0.0f;
                float c000 = 0, c100 = 0, c010 = 0, c110 = 0, c001 = 0, c101 = 0, c011 = 0, c111 = 0;
                bool cut = false;
                if (cell) {
                    const int64_t b = ((int64_t)k * ny + j) * nx + i;
                    // `shifted_sdf .= sdf .- th` (RBFs4Smoothing.jl:286) folded into the loads
                    c000 = sdf[b] - shift; c100 = sdf[b + 1] - shift; c010 = sdf[b + sy] - shift; c110 = sdf[b + sy + 1] - shift;
                    c001 = sdf[b + sz] - shift; c101 = sdf[b + sz + 1] - shift; c011 = sdf[b + sz + sy] - shift; c111 = sdf[b + sz + sy + 1] - shift;
                    const float mn = fminf(fminf(fminf(c000, c100), fminf(c010, c110)), fminf(fminf(c001, c101), fminf(c011, c111)));
                    const float mx = fmaxf(fmaxf(fmaxf(c000, c100), fmaxf(c010, c110)), fmaxf(fmaxf(c001, c101), fmaxf(c011, c111)));
                    if (!(mx < iso)) {
                        if (mn >= iso) v = elvol;
                        else cut = true;
                    }
                }
                uint64_t m = __ballot(cut);
                while (m) {   // the cut cells of the segment, one after the other, all lanes on the Gauss points of one cell
                    const int src = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const float d000 = __shfl(c000, src, 64), d100 = __shfl(c100, src, 64), d010 = __shfl(c010, src, 64),
                                d110 = __shfl(c110, src, 64), d001 = __shfl(c001, src, 64), d101 = __shfl(c101, src, 64),
                                d011 = __shfl(c011, src, 64), d111 = __shfl(c111, src, 64);
                    float part = 0.0f;
                    if (tensor) {
                        // along x: lane (pair n + iq) holds c00, c01, c10, c11 (pair 0 .. 3) at xi_iq
                        float s1 = 0.0f;
                        {
                            const int pr = lane / n, iq = lane - pr * n;
                            const float xi = sG[iq < n ? iq : 0];
                            const float cA = pr == 0 ? d000 : (pr == 1 ? d001 : (pr == 2 ? d010 : d011));
                            const float cB = pr == 0 ? d100 : (pr == 1 ? d101 : (pr == 2 ? d110 : d111));
                            s1 = cA * (1.0f - xi) + cB * xi;
                        }
                        // along y and z: lane e (and e + 64, ...) holds c0, c1 of its (jq, iq) pair - every lane takes part in
                        // the shuffles - and adds the weights of ITS column of Gauss points (kq ascending) that lie inside
                        for (int e0 = 0; e0 < nn; e0 += 64) {
                            const int e = e0 + lane < nn ? e0 + lane : nn - 1;
                            const int jq = e / n, iq = e - jq * n;
                            const float eta = sG[jq];
                            const float c00 = __shfl(s1, iq, 64), c01 = __shfl(s1, n + iq, 64), c10 = __shfl(s1, 2 * n + iq, 64),
                                        c11 = __shfl(s1, 3 * n + iq, 64);
                            const float c0 = c00 * (1.0f - eta) + c10 * eta;
                            const float c1 = c01 * (1.0f - eta) + c11 * eta;
                            if (e0 + lane < nn) {
                                for (int kq = 0; kq < n; ++kq) {
                                    const float zeta = sG[kq];
                                    const float pv = c0 * (1.0f - zeta) + c1 * zeta;
                                    if (pv >= iso) part += sW[kq * nn + e];
                                }
                            }
                        }
                    } else {
                        for (int p = lane; p < npts; p += 64) {
                            const int iq = p % q.order, jq = (p / q.order) % q.order, kq = p / (q.order * q.order);
                            const float zeta = (q.gp[kq] + 1) / 2, eta = (q.gp[jq] + 1) / 2, xi = (q.gp[iq] + 1) / 2;
                            const float c00 = d000 * (1.0f - xi) + d100 * xi;
                            const float c01 = d001 * (1.0f - xi) + d101 * xi;
                            const float c10 = d010 * (1.0f - xi) + d110 * xi;
                            const float c11 = d011 * (1.0f - xi) + d111 * xi;
                            const float c0 = c00 * (1.0f - eta) + c10 * eta;
                            const float c1 = c01 * (1.0f - eta) + c11 * eta;
                            const float pv = c0 * (1.0f - zeta) + c1 * zeta;
                            if (pv >= iso) part += q.gw[iq] * q.gw[jq] * q.gw[kq] * jac;
                        }
                    }
                    part = wave_sum_f32(part);
                    if (lane == src) v = part;
                }
                psum += wave_sum_f32(v);
            }
        }
        if (lane == 0) partial[row] = psum;
    }
}

// Level bisection on one field: every level th lies in [lo, hi].  A segment with max < lo stays outside and one with
// min >= hi stays full at every further level; a row of such segments keeps its value - written here once, from the
// constants of volume_cfull_kernel - and leaves the list of rows the next levels work on.
// in / n_in: the current list (null: all rows 0 .. nrows - 1); out / n_out: the rows that can still change.
__global__ void __launch_bounds__(256) volume_narrow_kernel(const float* __restrict__ segmn, const float* __restrict__ segmx, int nseg,
                                                           int nrows, float lo, float hi, const float* __restrict__ cfull,
                                                           const int* __restrict__ in, const uint32_t* __restrict__ n_in,
                                                           int* __restrict__ out, uint32_t* __restrict__ n_out,
                                                           float* __restrict__ partial)
{
    const int total = in ? (int)*n_in : nrows;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = false;
    int row = 0;
    if (idx < total) {
        row = in ? in[idx] : idx;
        float psum = 0.0f;
        for (int sg = 0; sg < nseg; ++sg) {
            const float smn = segmn[(size_t)row * nseg + sg], smx = segmx[(size_t)row * nseg + sg];
            if (smx < lo) continue;
            if (smn >= hi) psum += cfull[sg == nseg - 1 ? 1 : 0];
            else live = true;
        }
        if (!live) partial[row] = psum;
    }
    // append the live rows (one atomic per wavefront; the order of the list does not enter any sum)
    const uint64_t m = __ballot(live);
    if (m) {
        const int lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(n_out, (uint32_t)__popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1, 64);
        if (live) out[base + __popcll(m & ((1ull << lane) - 1ull))] = row;
    }
}

// Small results back to the host WITHOUT the copy engine: a kernel writes them into pinned host memory and the host waits
// for the stream.  (A hipMemcpy of a few KB queues behind whatever bulk transfer the engine is busy with - in r2s_rho2sdf the
// cleaned field, 1.07 GB in 32 MB pieces, travels while the CG runs, and each of its ~25 scalar read-backs waited up to
// 0.6 ms for a piece to finish: the RBF stage took 38 or 46 ms depending on which engine the runtime had picked.)
struct HostMailbox {
    void* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes)
    {
        if (bytes <= cap) return true;
        release();
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
        cap = bytes;
        return true;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};
__global__ void __launch_bounds__(256) mailbox_copy_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t nwords)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nwords) dst[i] = src[i];
}
// host <- device, `bytes` a multiple of 4 and small; orders like a hipMemcpy on `st` followed by a stream synchronisation
static int d2h_small(void* host, const void* dev, size_t bytes, hipStream_t st, HostMailbox& mb)
{
    static const bool off = getenv("R2S_MAILBOX") && atoi(getenv("R2S_MAILBOX")) == 0;   // (A/B switch)
    if (off || bytes == 0 || (bytes & 3u) || bytes > ((size_t)1 << 20) || !mb.ensure(std::max<size_t>(bytes, (size_t)64 << 10))) {
        HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return 0;
    }
    const size_t nw = bytes / 4;
    mailbox_copy_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>((const uint32_t*)dev, (uint32_t*)mb.p, nw);
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(host, mb.p, bytes);
    return 0;
}

static int vol_grid()
{
    static const int g = getenv("R2S_VOL_GRID") ? atoi(getenv("R2S_VOL_GRID")) : 16384;
    return g > 0 ? g : 16384;
}
struct VolumeWork {
    DevBuf partial, result, segmn, segmx, qpts, live[2], cnt, cfull;
    HostMailbox mb;
    float qpts_jac = -1.0f;             // the Jacobian the point table was built with
    const float* seg_field = nullptr;   // the field the segment extrema were computed for (prepare)
    int cur = -1;                       // live[cur]: the rows the levels still work on (-1: all rows)
    QuadTab q;
    int init(int order)
    {
        if (order < 1 || order > 32) return fail(R2S_ERR_ARG, "quadrature order %d not in 1..32", order);
        double x[32], w[32];
        gauss_legendre(order, x, w);
        for (int i = 0; i < order; ++i) { q.gp[i] = (float)x[i]; q.gw[i] = (float)w[i]; }   // CalcVolumeFromSDF.jl:43-44
        q.order = order;
        ENSURE(result, 64);
        return 0;
    }
    // table of the quadrature points for cells of this size (quad_points_kernel)
    int points(float jac, hipStream_t st)
    {
        if (qpts.p && qpts_jac == jac) return 0;
        const int npts = q.order * q.order * q.order;
        ENSURE(qpts, sizeof(float) * 8 * (size_t)npts);
        quad_points_kernel<<<(npts + 255) / 256, 256, 0, st>>>(q, jac, qpts.as<float>());
        qpts_jac = jac;
        return 0;
    }
    // before a series of run() calls on the same field: segment extrema, so that each level only looks at the cells near it
    int prepare(const float* d_sdf, int nx, int ny, int nz, hipStream_t st)
    {
        const int nrows = (ny - 1) * (nz - 1), nseg = (nx - 1 + 63) / 64;
        seg_field = nullptr;
        cur = -1;
        if (nrows <= 0 || nseg <= 0) return 0;
        ENSURE(segmn, sizeof(float) * (size_t)nrows * nseg);
        ENSURE(segmx, sizeof(float) * (size_t)nrows * nseg);
        volume_seg_minmax_kernel<<<nrows, 256, 0, st>>>(d_sdf, nx, ny, nz, segmn.as<float>(), segmx.as<float>(), 0);
        seg_field = d_sdf;
        return 0;
    }
    // level bisection (after prepare): every further run() has its level (shift + iso) in [lo, hi] - rows that cannot change
    // any more get their final value and leave the list of rows run() works on
    int narrow(int nx, int ny, int nz, float edge, float lo, float hi, hipStream_t st)
    {
        const int nrows = (ny - 1) * (nz - 1), nseg = (nx - 1 + 63) / 64;
        if (!seg_field || nrows <= 0) return 0;
        ENSURE(partial, sizeof(float) * (size_t)nrows);
        ENSURE(live[0], sizeof(int) * (size_t)nrows);
        ENSURE(live[1], sizeof(int) * (size_t)nrows);
        ENSURE(cnt, 64);
        if (!cfull.p) {
            ENSURE(cfull, 64);
        }
        volume_cfull_kernel<<<1, 64, 0, st>>>(nx, edge * edge * edge, cfull.as<float>());
        const int nxt = cur < 0 ? 0 : 1 - cur;
        uint32_t* counts = cnt.as<uint32_t>();   // [0], [1]: the lengths of live[0], live[1]
        HIP_TRY(hipMemsetAsync(counts + nxt, 0, 4, st));
        // (the launch covers the longest possible list: the current length is read on the device)
        volume_narrow_kernel<<<(nrows + 255) / 256, 256, 0, st>>>(segmn.as<float>(), segmx.as<float>(), nseg, nrows, lo, hi, cfull.as<float>(),
                                                                 cur < 0 ? nullptr : live[cur].as<int>(), cur < 0 ? nullptr : counts + cur,
                                                                 live[nxt].as<int>(), counts + nxt, partial.as<float>());
        cur = nxt;
        return 0;
    }
    // volume of {sdf - shift >= iso}; synchronises the stream
    int run(const float* d_sdf, int nx, int ny, int nz, float edge, float shift, float iso, hipStream_t st, float* out)
    {
        const float elvol = edge * edge * edge;       // element_edge_length^3 (:40)
        const float jac = elvol / 8.0f;               // :51
        const int nrows = (ny - 1) * (nz - 1);
        if (nrows <= 0) { *out = 0.0f; return 0; }
        ENSURE(partial, sizeof(float) * (size_t)nrows);
        const bool segs = seg_field == d_sdf && seg_field != nullptr;
        const bool listed = segs && cur >= 0;
        {
            int rc = points(jac, st);
            if (rc) return rc;
        }
        // (a bounded grid: workgroups set up the quadrature tables once and take rows in turn; 16 384 measured best at 512^3)
        volume_rowwave_kernel<<<std::min((nrows + 3) / 4, vol_grid()), 256, 0, st>>>(
            d_sdf, nx, ny, nz, shift, iso, elvol, jac, q, partial.as<float>(), 0, nrows, segs ? segmn.as<float>() : nullptr,
            segs ? segmx.as<float>() : nullptr, qpts.as<float4>(), listed ? live[cur].as<int>() : nullptr,
            listed ? cnt.as<uint32_t>() + cur : nullptr);
        sum_f32_kernel<<<1, 1024, 0, st>>>(partial.as<float>(), nrows, result.as<float>());
        return d2h_small(out, result.p, sizeof(float), st, mb);
    }
    void release()
    {
        mb.release();
        partial.release(); result.release(); segmn.release(); segmx.release(); qpts.release(); live[0].release(); live[1].release();
        cnt.release(); cfull.release();
        seg_field = nullptr; qpts_jac = -1.0f; cur = -1;
    }
};

// ====================================================================================
// RBF smoothing.  KDTree inrange/knn on a regular lattice = fixed stencil; neighbours are
// visited by increasing lattice distance (ties dz,dy,dx), distances from the reference's
// Float32 coordinates, Float64 sigma inside exp, Float32 accumulation (RBFs4Smoothing.jl:238-243).
// ====================================================================================
struct Stencil {
    int n;
    signed char off[512][3];
};

static void build_stencil(int s, const int frac[3], Stencil* st, double R2)
{
    // candidates: lattice distance^2 (in 1/s cells) up to 5 % beyond the support radius; the run-time
    // `dist <= max_distance` test (RBFs4Smoothing.jl:240) decides
    const int d2max = (int)std::floor(R2 * 1.05 * s * s + 0.25);
    int cand[512][4], n = 0;
    for (int dz = -3; dz <= 4; ++dz)
        for (int dy = -3; dy <= 4; ++dy)
            for (int dx = -3; dx <= 4; ++dx) {
                const int ex = dx * s - frac[0], ey = dy * s - frac[1], ez = dz * s - frac[2];
                const int d2 = ex * ex + ey * ey + ez * ez;
                if (d2 > d2max) continue;
                cand[n][0] = d2; cand[n][1] = dz; cand[n][2] = dy; cand[n][3] = dx;
                n++;
            }
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) {
        for (int q = 0; q < 4; ++q)
            if (cand[a][q] != cand[b][q]) return cand[a][q] < cand[b][q];
        return false;
    });
    st->n = n;
    for (int i = 0; i < n; ++i) {
        st->off[i][0] = (signed char)cand[idx[i]][3];
        st->off[i][1] = (signed char)cand[idx[i]][2];
        st->off[i][2] = (signed char)cand[idx[i]][1];
    }
}

struct RbfGeom {
    int nx, ny, nz;          // coarse lattice
    const float *cx, *cy, *cz;
    double sigma;
    float max_distance;
    double thr;
    int tap_r, tap_d2;       // stencil radius / largest lattice distance^2 that can reach the threshold
};

// exp(-t) for 0 <= t <= 64 (kernel arguments: t = (dist/sigma)^2 <= -ln(threshold)): t = k ln2/64 + r, exp(-t) =
// 2^(-k/64) exp(-r) with 2^(-j/64) from a 64-entry table and a degree-6 Taylor polynomial of exp(-r), |r| <= ln2/128.
// Relative error <= 2 ulp(Float64) - the class of difference that already separates libm's exp from the device
// library's (the smoothing stage is compared at Float32 round-off) - at a quarter of the device
// library's instruction count; the RBF evaluations spend most of their time in it.
__constant__ double c_exp2_neg_64[64] = {   // 2^(-j/64), j = 0..63, correctly rounded
    1, 0.98922801319397546, 0.97857206208770009, 0.96803089674614717,
    0.9576032806985737, 0.9472879907934828, 0.93708381705514998, 0.92698956254169274,
    0.91700404320467122, 0.90712608775019943, 0.89735453750155358, 0.88768824626326059,
    0.87812608018664973, 0.86866691763685311, 0.85930964906123897, 0.85005317685926174,
    0.8408964152537145, 0.83183829016336819, 0.82287773907698247, 0.81401371092867392,
    0.80524516597462714, 0.7965710756711335, 0.78799042255394325, 0.77950220011891846,
    0.77110541270397037, 0.76279907537226921, 0.75458221379671142, 0.74645386414563242,
    0.73841307296974967, 0.73045889709032352, 0.72259040348852333, 0.71480666919598501,
    0.70710678118654757, 0.69948983626915562, 0.69195494098191601, 0.68450121148729526,
    0.67712777346844633, 0.66983376202665146, 0.66261832157987066, 0.65548060576238221,
    0.64841977732550482, 0.64143500803938913, 0.63452547859586661, 0.62769037851234555,
    0.620928906036742, 0.61424026805343501, 0.60762367999023448, 0.60107836572635154,
    0.59460355750136051, 0.58819849582514061, 0.58186242938878874, 0.57559461497649134,
    0.56939431737834578, 0.56326080930412092, 0.55719337129794622, 0.55119129165392045,
    0.54525386633262884, 0.53938039887855993, 0.53357020033841185, 0.52782258918027858,
    0.52213689121370688, 0.51651243951061421, 0.51094857432705831, 0.50544464302585024,
};
__device__ __forceinline__ double exp_neg_fast(double t, const double* __restrict__ tab)
{
    const double kf = rint(t * 92.33248261689366);              // 64 / ln 2
    double r = fma(-kf, 0.010830424696248286, t);               // ln2/64, high part (11 trailing zero bits: exact product for k < 2^11)
    r = fma(-kf, 8.59050471673183e-16, r);                      // low part
    const int k = (int)kf;
    double p = fma(r, -1.0 / 720.0, 1.0 / 120.0);
    p = fma(r, -p, 1.0 / 24.0);
    p = fma(r, -p, 1.0 / 6.0);
    p = fma(r, -p, 0.5);
    p = fma(r, -p, 1.0);
    p = fma(r, -p, 1.0);
    return ldexp(tab[k & 63] * p, -(k >> 6));
}

__device__ __forceinline__ float rbf_apply_point(const RbfGeom& G, const float* __restrict__ w, int s, int tnx, int tny,
                                                 const float* __restrict__ tx, const float* __restrict__ ty,
                                                 const float* __restrict__ tz, const Stencil* __restrict__ stencils,
                                                 const double* __restrict__ etab, int64_t t);
// rbf_interpolation_kdtree (:219-248): 1 thread / target point
__global__ void __launch_bounds__(256) rbf_apply_kernel(RbfGeom G, const float* __restrict__ w, int s, int tnx, int tny,
                                                       int tnz, const float* __restrict__ tx,
                                                       const float* __restrict__ ty, const float* __restrict__ tz,
                                                       const Stencil* __restrict__ stencils, float add,
                                                       float* __restrict__ out, int64_t t_begin = 0, int64_t t_end = -1)
{
    // [t_begin, t_end): the targets of this launch (a Z-slab of a multi-device run; `w` and `out` are addressed as
    // the whole grids')
    __shared__ double etab[64];
    if (threadIdx.x < 64) etab[threadIdx.x] = c_exp2_neg_64[threadIdx.x];
    __syncthreads();
    const int64_t t = t_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nt = t_end >= 0 ? t_end : (int64_t)tnx * tny * tnz;
    if (t >= nt) return;
    out[t] = rbf_apply_point(G, w, s, tnx, tny, tx, ty, tz, stencils, etab, t) + add;
}
__device__ __forceinline__ float rbf_apply_point(const RbfGeom& G, const float* __restrict__ w, int s, int tnx, int tny,
                                                 const float* __restrict__ tx, const float* __restrict__ ty,
                                                 const float* __restrict__ tz, const Stencil* __restrict__ stencils,
                                                 const double* __restrict__ etab, int64_t t)
{
    const double inv_sigma = 1.0 / G.sigma;
    const int i = (int)(t % tnx), j = (int)((t / tnx) % tny), k = (int)(t / ((int64_t)tnx * tny));
    const Stencil& S = stencils[((k % s) * s + (j % s)) * s + (i % s)];
    const int bi = i / s, bj = j / s, bk = k / s;
    const float px = tx[i], py = ty[j], pz = tz[k];
    float acc = 0.0f;
    for (int q = 0; q < S.n; ++q) {
        const int ci = bi + S.off[q][0], cj = bj + S.off[q][1], ck = bk + S.off[q][2];
        if (ci < 0 || cj < 0 || ck < 0 || ci >= G.nx || cj >= G.ny || ck >= G.nz) continue;
        const float dx = px - G.cx[ci], dy = py - G.cy[cj], dz = pz - G.cz[ck];
        const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
        if (dist <= G.max_distance) {
            // (dist / sigma)^2 through the reciprocal and exp through exp_neg_fast: Float64 values within 2-3 ulp of the
            // reference's, far below the Float32 accumulation they feed
            const double u = (double)dist * inv_sigma;
            acc = (float)((double)acc + (double)w[((int64_t)ck * G.ny + cj) * G.nx + ci] * exp_neg_fast(u * u, etab));
        }
    }
    return acc;
}

// y = K x, K = compute_sparse_kernel_matrix (:142-176); row accumulation in ascending linear index
__global__ void __launch_bounds__(256) rbf_matvec_kernel(RbfGeom G, const float* __restrict__ x, float* __restrict__ y,
                                                        int64_t t_begin = 0, int64_t t_end = -1)
{
    const int64_t t = t_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)G.nx * G.ny * G.nz;
    if (t >= (t_end >= 0 ? t_end : n)) return;
    const int i = (int)(t % G.nx), j = (int)((t / G.nx) % G.ny), k = (int)(t / ((int64_t)G.nx * G.ny));
    const float px = G.cx[i], py = G.cy[j], pz = G.cz[k];
    float acc = 0.0f;
    // taps beyond lattice distance^2 = tap_d2 are far outside the kernel support (next possible value of a
    // sum of three squares leaves a margin of > 5 % in r) and cannot pass `val > threshold`
    for (int ck = k - G.tap_r; ck <= k + G.tap_r; ++ck)
        for (int cj = j - G.tap_r; cj <= j + G.tap_r; ++cj)
            for (int ci = i - G.tap_r; ci <= i + G.tap_r; ++ci) {
                if ((ck - k) * (ck - k) + (cj - j) * (cj - j) + (ci - i) * (ci - i) > G.tap_d2) continue;
                if (ci < 0 || cj < 0 || ck < 0 || ci >= G.nx || cj >= G.ny || ck >= G.nz) continue;
                const float dx = px - G.cx[ci], dy = py - G.cy[cj], dz = pz - G.cz[ck];
                const float r = sqrtf(dx * dx + dy * dy + dz * dz);
                const double u = (double)r / G.sigma;
                const double val = exp(-(u * u));
                if (val > G.thr) acc += (float)val * x[((int64_t)ck * G.ny + cj) * G.nx + ci];
            }
    y[t] = acc;
}

// The CG of compute_rbf_weights applies the same sparse matrix every iteration: when HBM has room its
// entries are materialised once, tap-major ([tap][voxel] Float32, 288 GB make 512^3 x 81 taps = 43 GB
// affordable), and each matvec becomes a coalesced stream instead of 81 exp() per voxel.  Taps are listed
// in the loop order of rbf_matvec_kernel and absent entries are stored as 0, so the row sums are formed
// in the same order from the same values (bit-identical results).
// The buffer is kept for the life of the process (r2s_release_cache frees it): giving tens of GB back to the
// driver and asking for them again costs seconds per call (freed VRAM is scrubbed), far more than it saves.
static DevBuf g_rbf_kv;
static std::mutex g_rbf_kv_mutex;   // one smoothing call at a time may use (or regrow) the shared buffer
#define RBF_MAX_TAPS 160
struct RbfTaps {
    int n;
    signed char off[RBF_MAX_TAPS][3];
};
__global__ void __launch_bounds__(256) rbf_kbuild_kernel(RbfGeom G, RbfTaps T, float* __restrict__ kv)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)G.nx * G.ny * G.nz;
    if (t >= n) return;
    const int i = (int)(t % G.nx), j = (int)((t / G.nx) % G.ny), k = (int)(t / ((int64_t)G.nx * G.ny));
    const float px = G.cx[i], py = G.cy[j], pz = G.cz[k];
    for (int q = 0; q < T.n; ++q) {
        const int ci = i + T.off[q][0], cj = j + T.off[q][1], ck = k + T.off[q][2];
        float v = 0.0f;
        if (!(ci < 0 || cj < 0 || ck < 0 || ci >= G.nx || cj >= G.ny || ck >= G.nz)) {
            const float dx = px - G.cx[ci], dy = py - G.cy[cj], dz = pz - G.cz[ck];
            const float r = sqrtf(dx * dx + dy * dy + dz * dz);
            const double u = (double)r / G.sigma;
            const double val = exp(-(u * u));
            if (val > G.thr) v = (float)val;
        }
        kv[(int64_t)q * n + t] = v;
    }
}
__global__ void __launch_bounds__(256) rbf_matvec_k_kernel(RbfGeom G, RbfTaps T, const float* __restrict__ kv,
                                                          const float* __restrict__ x, float* __restrict__ y)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)G.nx * G.ny * G.nz;
    if (t >= n) return;
    float acc = 0.0f;
    // eight taps per trip: their matrix entries and x values are independent loads in flight together; an
    // absent entry is stored as 0 and its x index is clamped, so the products are added in tap order
    // exactly as the on-the-fly kernel adds the present ones
    for (int q0 = 0; q0 < T.n; q0 += 8) {
        float v[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u;
            v[u] = 0.0f;
            xv[u] = 0.0f;
            if (q < T.n) {
                v[u] = kv[(int64_t)q * n + t];
                int64_t o = t + ((int64_t)T.off[q][2] * G.ny + T.off[q][1]) * G.nx + T.off[q][0];
                o = o < 0 ? 0 : (o >= n ? n - 1 : o);
                xv[u] = x[o];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v[u] != 0.0f) acc += v[u] * xv[u];
    }
    y[t] = acc;
}

// ---- K x through a table of the DISTINCT matrix entries --------------------------------------------------------
// An entry of K depends on its row and column only through the three Float32 coordinate differences
// cx[i] - cx[i+di], cy[j] - cy[j+dj], cz[k] - cz[k+dk].  On the reference's Float32 `range` lattice each of them takes
// only a handful of distinct values per offset (rounding patterns of the coordinates: <= 10 on a 512-point axis), so
// the whole matrix holds at most (2R+1)^3 x NV^3 distinct numbers.  They are evaluated ONCE with the reference's
// arithmetic (Float32 distance, Float64 exp, threshold test, rounding to Float32 - rbf_lut_build_kernel) and the
// matvec looks them up: per axis and offset a byte per lattice index names the variant.  Same values, same order
// of accumulation => bit-identical to rbf_matvec_kernel, without its 81 exp() per row and without the 43 GB
// materialised matrix of rbf_matvec_k_kernel; the table (<= 2 MB for R = 2) lives in L2.
#define RBF_NV 16   // variants per (axis, offset) the matvec table has room for; more -> the materialised / on-the-fly paths
#define RBF_FINE_CHUNKS 4   // Z chunks of the output field when finished chunks are forwarded to the host
#ifndef RBF_MATVEC_BATCH
#define RBF_MATVEC_BATCH 16
#endif
#ifndef RBF_APPLY_BATCH
#define RBF_APPLY_BATCH 12
#endif
#define RBF_NVA 64  // ... and the evaluation tables (the differences against the separately rounded output grid take 20-40)
struct RbfLutGeom {
    int nx, ny, nz, R, tap_d2;
    const uint8_t *vx, *vy, *vz;   // [2R+1][n_axis]: variant id of (index, offset), 255 = neighbour outside the lattice
    const float* T;                // [(2R+1)^3][NV][NV][NV], x variant fastest; 0 = entry absent (val <= threshold)
    const double* TA;              // [(2R+1)^3][NVA]^3: the kernel values of the EVALUATION (rbf_apply_kernel's arithmetic); 0 = beyond max_distance
    const double* TA16;            // the same with RBF_NV slots per axis (lattice-to-lattice evaluation with <= 15 variants), or null
    // the same values in the layout of the row-walk kernels (r2s_rbf_walk.hpp), or null
    const float* WT;               // matrix entries, RBF_NV slots per axis
    const double* WA;              // evaluation, wa_nv slots per axis
    int wa_nv;
};
struct RbfLutVals {
    float v[3][7][RBF_NVA];        // the variant values per axis and offset
    int R;
    double sigma, thr;
    float max_distance;
};
// (the variant values travel through device memory: the struct is larger than a kernel argument block)
__global__ void __launch_bounds__(256) rbf_lut_build_kernel(const RbfLutVals* __restrict__ Vp, float* __restrict__ T)
{
    const RbfLutVals& V = *Vp;
    const int W = 2 * V.R + 1;
    const int64_t n = (int64_t)W * W * W * RBF_NV * RBF_NV * RBF_NV;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int a = (int)(t % RBF_NV), b = (int)((t / RBF_NV) % RBF_NV), c = (int)((t / (RBF_NV * RBF_NV)) % RBF_NV);
    const int tap = (int)(t / (RBF_NV * RBF_NV * RBF_NV));
    const int di = tap % W, dj = (tap / W) % W, dk = tap / (W * W);
    const float dx = V.v[0][di][a], dy = V.v[1][dj][b], dz = V.v[2][dk][c];
    const float r = sqrtf(dx * dx + dy * dy + dz * dz);
    const double u = (double)r / V.sigma;
    const double val = exp(-(u * u));
    T[t] = (val > V.thr) ? (float)val : 0.0f;   // (unused variant slots hold NaN values -> comparisons false -> 0)
}
// the neighbours of a lattice point in the order of build_stencil at smooth = 1: lattice distance^2, then dz, dy, dx
template <int R>
struct RbfTapOrder {
    int n;
    signed char dk[(2 * R + 1) * (2 * R + 1) * (2 * R + 1)], dj[(2 * R + 1) * (2 * R + 1) * (2 * R + 1)], di[(2 * R + 1) * (2 * R + 1) * (2 * R + 1)];
};
template <int R, int D2>
constexpr RbfTapOrder<R> rbf_tap_order_rowwise()   // dk, dj, di ascending: the order of the matrix rows (rbf_matvec_kernel)
{
    RbfTapOrder<R> t{};
    t.n = 0;
    for (int dk = -R; dk <= R; ++dk)
        for (int dj = -R; dj <= R; ++dj)
            for (int di = -R; di <= R; ++di)
                if (dk * dk + dj * dj + di * di <= D2) {
                    t.dk[t.n] = (signed char)(dk + R); t.dj[t.n] = (signed char)(dj + R); t.di[t.n] = (signed char)(di + R);
                    t.n++;
                }
    return t;
}
template <int R, int D2>
constexpr RbfTapOrder<R> rbf_tap_order()
{
    RbfTapOrder<R> t{};
    t.n = 0;
    for (int d2 = 0; d2 <= D2; ++d2)
        for (int dk = -R; dk <= R; ++dk)
            for (int dj = -R; dj <= R; ++dj)
                for (int di = -R; di <= R; ++di)
                    if (dk * dk + dj * dj + di * di == d2) {
                        t.dk[t.n] = (signed char)(dk + R); t.dj[t.n] = (signed char)(dj + R); t.di[t.n] = (signed char)(di + R);
                        t.n++;
                    }
    return t;
}

// one row through per-lane clamped addresses and predicates: rows of the first / last R planes, of wavefronts that
// straddle two planes and of a slab whose halo ends nearby
// the table of the evaluation (rbf_apply_kernel with targets = lattice points): same differences, that kernel's arithmetic
// (nv: variant slots per axis of this table - RBF_NVA, or RBF_NV for the compact table of the lattice-to-lattice evaluation)
__global__ void __launch_bounds__(256) rbf_lut_build_apply_kernel(const RbfLutVals* __restrict__ Vp, double* __restrict__ TA, int nv = RBF_NVA)
{
    const RbfLutVals& V = *Vp;
    __shared__ double etab[64];
    if (threadIdx.x < 64) etab[threadIdx.x] = c_exp2_neg_64[threadIdx.x];
    __syncthreads();
    const int W = 2 * V.R + 1;
    const int64_t n = (int64_t)W * W * W * nv * nv * nv;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int a = (int)(t % nv), b = (int)((t / nv) % nv), c = (int)((t / (nv * nv)) % nv);
    const int tap = (int)(t / ((int64_t)nv * nv * nv));
    const int di = tap % W, dj = (tap / W) % W, dk = tap / (W * W);
    const float dx = V.v[0][di][a], dy = V.v[1][dj][b], dz = V.v[2][dk][c];
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    const double inv_sigma = 1.0 / V.sigma;
    double val = 0.0;
    if (dist <= V.max_distance) {   // (NaN differences of unused variant slots: false)
        const double u = (double)dist * inv_sigma;
        val = exp_neg_fast(u * u, etab);
    }
    TA[t] = val;
}
template <int R>
__device__ __forceinline__ float rbf_lut_row_clamped(const RbfLutGeom& G, const float* __restrict__ x, int64_t t, int64_t first,
                                                     int64_t last)
{
    constexpr int W = 2 * R + 1;
    const int i = (int)(t % G.nx), j = (int)((t / G.nx) % G.ny), k = (int)(t / ((int64_t)G.nx * G.ny));
    uint32_t ax[W], by[W], cz[W];
#pragma unroll
    for (int d = 0; d < W; ++d) {
        ax[d] = G.vx[d * G.nx + i];
        by[d] = G.vy[d * G.ny + j];
        cz[d] = G.vz[d * G.nz + k];
    }
    float acc = 0.0f;
#pragma unroll
    for (int dk = 0; dk < W; ++dk) {
#pragma unroll
        for (int dj = 0; dj < W; ++dj) {
            if ((dk - R) * (dk - R) + (dj - R) * (dj - R) > G.tap_d2) continue;
            const bool row_ok = cz[dk] != 255u && by[dj] != 255u;
            const uint32_t c = row_ok ? cz[dk] : 0u, b = row_ok ? by[dj] : 0u;
            const float* __restrict__ row = G.T + ((((size_t)((dk * W + dj) * W)) * RBF_NV + c) * RBF_NV + b) * RBF_NV;
            const int64_t base = t + ((int64_t)(dk - R) * G.ny + (dj - R)) * G.nx - R;
            float w[W], xv[W];
#pragma unroll
            for (int di = 0; di < W; ++di) {
                const bool ok = row_ok && ax[di] != 255u &&
                                (dk - R) * (dk - R) + (dj - R) * (dj - R) + (di - R) * (di - R) <= G.tap_d2;
                const float tv = row[(size_t)di * RBF_NV * RBF_NV * RBF_NV + (ax[di] != 255u ? ax[di] : 0u)];
                int64_t o = base + di;
                o = o < first ? first : (o > last ? last : o);
                xv[di] = x[o];
                w[di] = ok ? tv : 0.0f;
            }
#pragma unroll
            for (int di = 0; di < W; ++di)
                if (w[di] != 0.0f) acc += w[di] * xv[di];
        }
    }
    return acc;
}
template <int R, int D2>   // D2: G.tap_d2 as a compile-time constant (no branches between the tap groups), or -1
__global__ void __launch_bounds__(256) rbf_matvec_lut_kernel(RbfLutGeom G, const float* __restrict__ x, float* __restrict__ y,
                                                            int64_t t_begin = 0, int64_t t_end = -1, int64_t x_lo = 0,
                                                            int64_t x_hi = -1)
{
    // [t_begin, t_end): rows of this launch; [x_lo, x_hi]: the part of x that exists on this device (a Z-slab with its
    // halo; x and y are addressed as the whole vectors) - the clamp of absent neighbours stays inside it.
    // The loop order of rbf_matvec_kernel (ck, cj, ci ascending) is kept: the row sum is formed from the same values in
    // the same order.
    //
    // A wavefront = 64 consecutive rows.  When they lie in one Z plane with all 2R+1 neighbour planes addressable, every
    // address is (wave-uniform pointer) + (32-bit lane offset) + (immediate): the vector loads carry their base in SGPRs
    // and the loop spends its VALU work on the products only (the row-by-row form above needs ~20 VALU instructions per
    // tap for 64-bit offsets, clamps and predicates, and was bound by them).  Neighbours outside the lattice in x or y
    // take variant slot RBF_NV-1, which rbf_lut_axis leaves unused on every axis: its table entries are 0 (built from
    // NaN differences), i.e. "entry absent", and the x they multiply is some other element of the vector that is never
    // used.
    constexpr int W = 2 * R + 1;
    constexpr uint32_t NV3 = RBF_NV * RBF_NV * RBF_NV;
    const int d2max = D2 >= 0 ? D2 : G.tap_d2;
    const int64_t n = (int64_t)G.nx * G.ny * G.nz;
    const int64_t tend = t_end >= 0 ? t_end : n;
    const int64_t first = x_lo, last = x_hi >= 0 ? x_hi : n - 1;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tw0 = t_begin + (int64_t)blockIdx.x * 256 + (int64_t)wv * 64;   // first row of the wavefront (in SGPRs)
    const int64_t t = tw0 + lane;
    if (tw0 >= tend) return;
    const int64_t tw1 = tw0 + 63 < tend ? tw0 + 63 : tend - 1;
    const int64_t plane = (int64_t)G.nx * G.ny;
    const int64_t reach = (int64_t)R * plane + (int64_t)R * G.nx + R;
    const int k = __builtin_amdgcn_readfirstlane((int)(tw0 / plane));
    const bool fast = (tw1 / plane == k) && k >= R && k < G.nz - R && tw0 - reach >= first && tw1 + reach <= last;
    if (!fast) {
        if (t < tend) y[t] = rbf_lut_row_clamped<R>(G, x, t, first, last);
        return;
    }
    const int64_t tc = t < tend ? t : tend - 1;   // (idle lanes of the last wavefront repeat its last row)
    const uint32_t r2 = (uint32_t)(tc - (int64_t)k * plane);
    const uint32_t j = r2 / (uint32_t)G.nx, i = r2 - j * (uint32_t)G.nx;
    uint32_t ab[W][W];   // (y variant, x variant) part of the table index per (dj, di), as a BYTE offset (saddr + voffset loads)
    {
        uint32_t a[W], b[W];
#pragma unroll
        for (int d = 0; d < W; ++d) {
            const uint32_t va = G.vx[d * G.nx + i], vb = G.vy[d * G.ny + j];
            a[d] = va != 255u ? va : (uint32_t)(RBF_NV - 1);
            b[d] = vb != 255u ? vb : (uint32_t)(RBF_NV - 1);
        }
#pragma unroll
        for (int dj = 0; dj < W; ++dj)
#pragma unroll
            for (int di = 0; di < W; ++di) ab[dj][di] = (b[dj] * RBF_NV + a[di]) * 4u;
    }
    // buffer loads: descriptor + scalar offset in SGPRs, the lane part as a 32-bit byte offset - no address arithmetic
    // in the VALU.  (x: a window of the vector around the wavefront's rows, so that the offsets fit 32 bits on any grid)
    const uint32_t lane4 = lane * 4u;
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(
        (void*)G.T, 0, (int)((uint32_t)(W * W * W) * NV3 * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x + (tw0 - reach)), 0, (int)((2 * reach + 64) * 4), 0x00020000);
    float acc = 0.0f;
    if constexpr (D2 >= 0) {
        // batches of RBF_MATVEC_BATCH entries: the loads of a batch are in flight together, then the (serial) row sum
        constexpr RbfTapOrder<R> TO = rbf_tap_order_rowwise<R, D2>();
        uint32_t c[W];
#pragma unroll
        for (int d = 0; d < W; ++d) c[d] = __builtin_amdgcn_readfirstlane((uint32_t)G.vz[d * G.nz + k]);   // (k is interior: never 255)
#pragma unroll
        for (int q0 = 0; q0 < TO.n; q0 += RBF_MATVEC_BATCH) {
            uint32_t wb[RBF_MATVEC_BATCH], xb[RBF_MATVEC_BATCH];
#pragma unroll
            for (int u = 0; u < RBF_MATVEC_BATCH; ++u) {
                const int q = q0 + u < TO.n ? q0 + u : TO.n - 1;
                const int dk = TO.dk[q], dj = TO.dj[q], di = TO.di[q];
                const uint32_t toff = ((uint32_t)((dk * W + dj) * W + di) * RBF_NV + c[dk]) * (RBF_NV * RBF_NV) * 4u;
                const uint32_t xoff = (uint32_t)(reach + ((int64_t)(dk - R) * G.ny + (dj - R)) * G.nx + (di - R)) * 4u;
#if defined(RBF_DIAG) && RBF_DIAG == 1   // timing-only builds: what the kernel costs without its table / vector loads
                wb[u] = __float_as_uint(1.0f + (float)ab[dj][di]);
#else
                wb[u] = __builtin_amdgcn_raw_buffer_load_b32(rT, (int)ab[dj][di], (int)toff, 0);
#endif
#if defined(RBF_DIAG) && RBF_DIAG == 2
                xb[u] = __float_as_uint(1.0f + (float)lane4);
#else
                xb[u] = __builtin_amdgcn_raw_buffer_load_b32(rX, (int)lane4, (int)xoff, 0);
#endif
            }
#pragma unroll
            for (int u = 0; u < RBF_MATVEC_BATCH; ++u) {
                if (q0 + u >= TO.n) continue;
                const float wq = __uint_as_float(wb[u]);
                if (wq != 0.0f) acc += wq * __uint_as_float(xb[u]);
            }
        }
    } else {
    #pragma unroll
        for (int dk = 0; dk < W; ++dk) {
            const uint32_t c = __builtin_amdgcn_readfirstlane((uint32_t)G.vz[dk * G.nz + k]);   // (k is interior: never 255)
    #pragma unroll
            for (int dj = 0; dj < W; ++dj) {
                if ((dk - R) * (dk - R) + (dj - R) * (dj - R) > d2max) continue;
                const uint32_t Trow = (((uint32_t)((dk * W + dj) * W) * RBF_NV + c) * (RBF_NV * RBF_NV)) * 4u;
                const uint32_t xrow = (uint32_t)(reach + ((int64_t)(dk - R) * G.ny + (dj - R)) * G.nx - R) * 4u;
                float w[W], xv[W];
    #pragma unroll
                for (int di = 0; di < W; ++di) {
                    w[di] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rT, (int)ab[dj][di], (int)(Trow + (uint32_t)di * NV3 * 4u), 0));
                    xv[di] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, (int)lane4, (int)(xrow + (uint32_t)di * 4u), 0));
                }
    #pragma unroll
                for (int di = 0; di < W; ++di) {
                    const bool in = (dk - R) * (dk - R) + (dj - R) * (dj - R) + (di - R) * (di - R) <= d2max;   // (uniform)
                    if (in && w[di] != 0.0f) acc += w[di] * xv[di];
                }
            }
        }
    }
    if (t < tend) y[t] = acc;
}

// The same product with the table rows of a workgroup staged in LDS.  A workgroup = 256 consecutive rows of one Z
// plane: they span 1-2 (at most RBF_MV_ROWS) lattice rows j, so per neighbour offset it needs one 64-byte table row
// (16 x variants) per j - 81 x 64 B = 5 KB per j at the default threshold.  The lanes then pick their entry with an LDS
// read (address = per-lane x variant + immediate) instead of a 64-lane gather from L1/L2, which cost twice what the
// loads of the vector itself do (timing-only builds without either: 2.5 / 3.9 ms of 5.25 ms at 512^3).
#define RBF_MV_ROWS 4
template <int R, int D2>
__global__ void __launch_bounds__(256) rbf_matvec_lds_kernel(RbfLutGeom G, const float* __restrict__ x, float* __restrict__ y,
                                                            int64_t t_begin, int64_t t_end, int64_t x_lo, int64_t x_hi)
{
    constexpr int W = 2 * R + 1;
    constexpr RbfTapOrder<R> TO = rbf_tap_order_rowwise<R, D2>();
    constexpr int NT = TO.n;
    __shared__ float sT[RBF_MV_ROWS * NT * RBF_NV];
    const int64_t n = (int64_t)G.nx * G.ny * G.nz;
    const int64_t tend = t_end >= 0 ? t_end : n;
    const int64_t first = x_lo, last = x_hi >= 0 ? x_hi : n - 1;
    const int64_t tb0 = t_begin + (int64_t)blockIdx.x * 256;   // first row of the workgroup
    if (tb0 >= tend) return;
    const int64_t tb1 = tb0 + 255 < tend ? tb0 + 255 : tend - 1;
    const int64_t plane = (int64_t)G.nx * G.ny;
    const int64_t reach = (int64_t)R * plane + (int64_t)R * G.nx + R;
    const int k = (int)(tb0 / plane);
    const int jA = (int)((tb0 - (int64_t)k * plane) / G.nx), jB = (int)((tb1 - (int64_t)k * plane) / G.nx);
    const bool fast = (tb1 / plane == k) && k >= R && k < G.nz - R && tb0 - reach >= first && tb1 + reach <= last &&
                      jB - jA + 1 <= RBF_MV_ROWS;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const int64_t t = tb0 + tid;
    if (!fast) {   // (uniform over the workgroup)
        if (t < tend) y[t] = rbf_lut_row_clamped<R>(G, x, t, first, last);
        return;
    }
    // ---- table rows of this workgroup -> LDS: sT[jr][q][x variant] ----
    // two steps so that the loads of a step are independent of each other: (1) one thread per (j row, neighbour) chases
    // offset -> variants of its plane / row -> position of the 64-byte table row; (2) all threads copy the rows, six
    // 4-byte loads in flight each (one serial loop of offset look-ups + table load per element spent more time waiting
    // than the products take)
    const int nj = jB - jA + 1;
    __shared__ uint32_t sBase[RBF_MV_ROWS * NT];
    for (int e = (int)tid; e < nj * NT; e += 256) {
        const int jr = e / NT, q = e - jr * NT;
        const int dk = TO.dk[q], dj = TO.dj[q], di = TO.di[q];
        const uint32_t c = G.vz[dk * G.nz + k];   // (k is interior: never 255)
        uint32_t b = G.vy[dj * G.ny + jA + jr];
        b = b != 255u ? b : (uint32_t)(RBF_NV - 1);
        sBase[e] = ((((uint32_t)((dk * W + dj) * W + di)) * RBF_NV + c) * RBF_NV + b) * RBF_NV;
    }
    __syncthreads();
    const int total = nj * NT * RBF_NV;
    for (int e0 = 0; e0 < total; e0 += 6 * 256) {
        float v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int e = e0 + u * 256 + (int)tid;
            v[u] = e < total ? G.T[sBase[e / RBF_NV] + (uint32_t)(e % RBF_NV)] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int e = e0 + u * 256 + (int)tid;
            if (e < total) sT[e] = v[u];
        }
    }
    __syncthreads();
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tw0 = tb0 + (int64_t)wv * 64;
    if (tw0 >= tend) return;
    const int64_t tc = t < tend ? t : tend - 1;   // (idle lanes of the last wavefront repeat its last row)
    const uint32_t r2 = (uint32_t)(tc - (int64_t)k * plane);
    const uint32_t j = r2 / (uint32_t)G.nx, i = r2 - j * (uint32_t)G.nx;
    uint32_t a4[W];   // byte offset of the lane's entry inside a staged table row, plus the offset of its j block
#pragma unroll
    for (int d = 0; d < W; ++d) {
        const uint32_t va = G.vx[d * G.nx + i];
        a4[d] = (va != 255u ? va : (uint32_t)(RBF_NV - 1)) * 4u + (j - (uint32_t)jA) * (uint32_t)(NT * RBF_NV * 4);
    }
    const uint32_t lane4 = lane * 4u;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x + (tw0 - reach)), 0, (int)((2 * reach + 64) * 4), 0x00020000);
    float acc = 0.0f;
#pragma unroll
    for (int q0 = 0; q0 < NT; q0 += RBF_MATVEC_BATCH) {
        uint32_t xb[RBF_MATVEC_BATCH];
        float wb[RBF_MATVEC_BATCH];
#pragma unroll
        for (int u = 0; u < RBF_MATVEC_BATCH; ++u) {
            const int q = q0 + u < NT ? q0 + u : NT - 1;
            const int dk = TO.dk[q], dj = TO.dj[q], di = TO.di[q];
            const uint32_t xoff = (uint32_t)(reach + ((int64_t)(dk - R) * G.ny + (dj - R)) * G.nx + (di - R)) * 4u;
            xb[u] = __builtin_amdgcn_raw_buffer_load_b32(rX, (int)lane4, (int)xoff, 0);
            wb[u] = *(const float*)((const char*)sT + a4[di] + (uint32_t)(q * RBF_NV * 4));
        }
#pragma unroll
        for (int u = 0; u < RBF_MATVEC_BATCH; ++u) {
            if (q0 + u >= NT) continue;
            if (wb[u] != 0.0f) acc += wb[u] * __uint_as_float(xb[u]);
        }
    }
    if (t < tend) y[t] = acc;
}

#include "r2s_rbf_walk.hpp"

// [t0, t1) / [xlo, xhi] as whole planes, or false
static bool rbf_walk_planes(const RbfLutGeom& LG, int64_t t0, int64_t t1, int64_t xlo, int64_t xhi, RbfWalkArgs* A)
{
    const int64_t plane = (int64_t)LG.nx * LG.ny, n = plane * LG.nz;
    if (t1 < 0) t1 = n;
    if (xhi < 0) xhi = n - 1;
    if (t0 % plane || t1 % plane || xlo % plane || (xhi + 1) % plane) return false;
    memset(A, 0, sizeof *A);
    A->nx = LG.nx; A->ny = LG.ny; A->nz = LG.nz;
    A->vx = LG.vx; A->vy = LG.vy; A->vz = LG.vz;
    A->k_begin = (int)(t0 / plane); A->k_end = (int)(t1 / plane);
    A->xk_lo = (int)(xlo / plane); A->xk_hi = (int)((xhi + 1) / plane) - 1;
    return true;
}
static void launch_rbf_matvec_lut(const RbfLutGeom& LG, unsigned nb, hipStream_t st, const float* x, float* y, int64_t t0 = 0,
                                  int64_t t1 = -1, int64_t xlo = 0, int64_t xhi = -1, double* dot_partial = nullptr)
{
    const char* mv_env = getenv("R2S_RBF_MATVEC");
    RbfWalkArgs WAr;
    if (!mv_env && LG.WT && rbf_walk_supported(LG.R, LG.tap_d2, LG.nx, LG.ny) && rbf_walk_planes(LG, t0, t1, xlo, xhi, &WAr)) {
        WAr.T = LG.WT; WAr.x = x; WAr.y = y; WAr.dot_partial = dot_partial;
        rbf_walk_launch(0, RBF_NV, WAr, st);
        return;
    }
    const bool global_only = mv_env && !strcmp(mv_env, "lutg");   // table entries gathered from L1/L2 (the tests compare)
    if (LG.R == 2 && LG.tap_d2 == 7 && !global_only) rbf_matvec_lds_kernel<2, 7><<<nb, 256, 0, st>>>(LG, x, y, t0, t1, xlo, xhi);   // threshold 1e-3 (default)
    else if (LG.R == 2 && LG.tap_d2 == 7) rbf_matvec_lut_kernel<2, 7><<<nb, 256, 0, st>>>(LG, x, y, t0, t1, xlo, xhi);
    else if (LG.R == 1) rbf_matvec_lut_kernel<1, -1><<<nb, 256, 0, st>>>(LG, x, y, t0, t1, xlo, xhi);
    else if (LG.R == 2) rbf_matvec_lut_kernel<2, -1><<<nb, 256, 0, st>>>(LG, x, y, t0, t1, xlo, xhi);
    else rbf_matvec_lut_kernel<3, -1><<<nb, 256, 0, st>>>(LG, x, y, t0, t1, xlo, xhi);
}

// rbf_apply_kernel for one target per lattice point (same grid; tx, ty, tz: the targets' coordinates, which the table's
// variants were formed with) through the table TA: the neighbours in the order of the
// stencil (lattice distance^2, then dz, dy, dx - build_stencil), Float64 product and sum rounded to Float32 per
// neighbour as there.  Wavefronts away from the first / last R planes take the table; the others evaluate as before.
template <int R, int D2>
__global__ void __launch_bounds__(256) rbf_apply_lut_kernel(RbfGeom G, RbfLutGeom L, const float* __restrict__ w,
                                                           const float* __restrict__ tx, const float* __restrict__ ty,
                                                           const float* __restrict__ tz, const Stencil* __restrict__ stencils,
                                                           float add, float* __restrict__ out, int64_t t_begin, int64_t t_end,
                                                           int64_t x_lo, int64_t x_hi)
{
    constexpr int W = 2 * R + 1;
    constexpr uint32_t NV3 = RBF_NVA * RBF_NVA * RBF_NVA;
    __shared__ double etab[64];
    if (threadIdx.x < 64) etab[threadIdx.x] = c_exp2_neg_64[threadIdx.x];
    __syncthreads();
    const int64_t n = (int64_t)G.nx * G.ny * G.nz;
    const int64_t tend = t_end >= 0 ? t_end : n;
    const int64_t first = x_lo, last = x_hi >= 0 ? x_hi : n - 1;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tw0 = t_begin + (int64_t)blockIdx.x * 256 + (int64_t)wv * 64;
    const int64_t t = tw0 + lane;
    if (tw0 >= tend) return;
    const int64_t tw1 = tw0 + 63 < tend ? tw0 + 63 : tend - 1;
    const int64_t plane = (int64_t)G.nx * G.ny;
    const int64_t reach = (int64_t)R * plane + (int64_t)R * G.nx + R;
    const int k = __builtin_amdgcn_readfirstlane((int)(tw0 / plane));
    const bool fast = (tw1 / plane == k) && k >= R && k < G.nz - R && tw0 - reach >= first && tw1 + reach <= last;
    if (!fast) {
        if (t < tend) out[t] = rbf_apply_point(G, w, 1, G.nx, G.ny, tx, ty, tz, stencils, etab, t) + add;
        return;
    }
    const int64_t tc = t < tend ? t : tend - 1;
    const uint32_t r2 = (uint32_t)(tc - (int64_t)k * plane);
    const uint32_t j = r2 / (uint32_t)G.nx, i = r2 - j * (uint32_t)G.nx;
    uint32_t ab[W][W];   // byte offsets of the (y variant, x variant) pair in a table row of doubles
    {
        uint32_t a[W], b[W];
#pragma unroll
        for (int d = 0; d < W; ++d) {
            const uint32_t va = L.vx[d * G.nx + i], vb = L.vy[d * G.ny + j];
            a[d] = va != 255u ? va : (uint32_t)(RBF_NVA - 1);
            b[d] = vb != 255u ? vb : (uint32_t)(RBF_NVA - 1);
        }
#pragma unroll
        for (int dj = 0; dj < W; ++dj)
#pragma unroll
            for (int di = 0; di < W; ++di) ab[dj][di] = (b[dj] * RBF_NVA + a[di]) * 8u;
    }
    uint32_t c[W];
#pragma unroll
    for (int d = 0; d < W; ++d) c[d] = __builtin_amdgcn_readfirstlane((uint32_t)L.vz[d * G.nz + k]);   // (k interior: never 255)
    const uint32_t lane4 = lane * 4u;
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(
        (void*)L.TA, 0, (int)((uint32_t)(W * W * W) * NV3 * 8u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(w + (tw0 - reach)), 0, (int)((2 * reach + 64) * 4), 0x00020000);
    float acc = 0.0f;
    // batches of RBF_APPLY_BATCH neighbours: all loads of a batch are in flight together, then the (serial) sums
    constexpr RbfTapOrder<R> TO = rbf_tap_order<R, D2>();
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q0 = 0; q0 < TO.n; q0 += RBF_APPLY_BATCH) {
        u32x2 eb[RBF_APPLY_BATCH];
        uint32_t wb[RBF_APPLY_BATCH];
#pragma unroll
        for (int u = 0; u < RBF_APPLY_BATCH; ++u) {
            const int q = q0 + u < TO.n ? q0 + u : TO.n - 1;
            const int dk = TO.dk[q], dj = TO.dj[q], di = TO.di[q];
            const uint32_t toff = ((uint32_t)((dk * W + dj) * W + di) * RBF_NVA + c[dk]) * (RBF_NVA * RBF_NVA) * 8u;
            const uint32_t xoff = (uint32_t)(reach + ((int64_t)(dk - R) * G.ny + (dj - R)) * G.nx + (di - R)) * 4u;
            eb[u] = __builtin_amdgcn_raw_buffer_load_b64(rT, (int)ab[dj][di], (int)toff, 0);
            wb[u] = __builtin_amdgcn_raw_buffer_load_b32(rX, (int)lane4, (int)xoff, 0);
        }
#pragma unroll
        for (int u = 0; u < RBF_APPLY_BATCH; ++u) {
            if (q0 + u >= TO.n) continue;
            const double e = __hiloint2double((int)eb[u].y, (int)eb[u].x);
            if (e != 0.0) acc = (float)((double)acc + (double)__uint_as_float(wb[u]) * e);
        }
    }
    if (t < tend) out[t] = acc + add;
}
// The lattice-to-lattice evaluation (the LSF of the level bisection) with the table rows of a workgroup staged in LDS, as
// in rbf_matvec_lds_kernel: with <= 15 variants per offset a row of the compact table TA16 is 128 bytes, 81 rows = 10 KB
// per lattice row j; a workgroup of 256 consecutive targets spans at most RBF_AP_ROWS of them (else: the gathered form).
#define RBF_AP_ROWS 2
template <int R, int D2>
__global__ void __launch_bounds__(256) rbf_apply_lds_kernel(RbfGeom G, RbfLutGeom L, const float* __restrict__ w,
                                                           const Stencil* __restrict__ stencils, float add, float* __restrict__ out,
                                                           int64_t t_begin, int64_t t_end, int64_t x_lo, int64_t x_hi)
{
    constexpr int W = 2 * R + 1;
    constexpr RbfTapOrder<R> TO = rbf_tap_order<R, D2>();
    constexpr int NT = TO.n;
    __shared__ double sT[RBF_AP_ROWS * NT * RBF_NV];
    __shared__ uint32_t sBase[RBF_AP_ROWS * NT];
    __shared__ double etab[64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const int64_t n = (int64_t)G.nx * G.ny * G.nz;
    const int64_t tend = t_end >= 0 ? t_end : n;
    const int64_t first = x_lo, last = x_hi >= 0 ? x_hi : n - 1;
    const int64_t tb0 = t_begin + (int64_t)blockIdx.x * 256;
    if (tb0 >= tend) return;
    const int64_t tb1 = tb0 + 255 < tend ? tb0 + 255 : tend - 1;
    const int64_t plane = (int64_t)G.nx * G.ny;
    const int64_t reach = (int64_t)R * plane + (int64_t)R * G.nx + R;
    const int k = (int)(tb0 / plane);
    const int jA = (int)((tb0 - (int64_t)k * plane) / G.nx), jB = (int)((tb1 - (int64_t)k * plane) / G.nx);
    const bool fast = (tb1 / plane == k) && k >= R && k < G.nz - R && tb0 - reach >= first && tb1 + reach <= last &&
                      jB - jA + 1 <= RBF_AP_ROWS;
    const int64_t t = tb0 + tid;
    if (!fast) {   // (uniform over the workgroup)
        if (tid < 64) etab[tid] = c_exp2_neg_64[tid];
        __syncthreads();
        if (t < tend) out[t] = rbf_apply_point(G, w, 1, G.nx, G.ny, G.cx, G.cy, G.cz, stencils, etab, t) + add;
        return;
    }
    const int nj = jB - jA + 1;
    for (int e = (int)tid; e < nj * NT; e += 256) {
        const int jr = e / NT, q = e - jr * NT;
        const int dk = TO.dk[q], dj = TO.dj[q], di = TO.di[q];
        const uint32_t c = L.vz[dk * G.nz + k];   // (k is interior: never 255)
        uint32_t b = L.vy[dj * G.ny + jA + jr];
        b = b != 255u ? b : (uint32_t)(RBF_NV - 1);
        sBase[e] = ((((uint32_t)((dk * W + dj) * W + di)) * RBF_NV + c) * RBF_NV + b) * RBF_NV;
    }
    __syncthreads();
    const int total = nj * NT * RBF_NV;
    for (int e0 = 0; e0 < total; e0 += 6 * 256) {
        double v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int e = e0 + u * 256 + (int)tid;
            v[u] = e < total ? L.TA16[sBase[e / RBF_NV] + (uint32_t)(e % RBF_NV)] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int e = e0 + u * 256 + (int)tid;
            if (e < total) sT[e] = v[u];
        }
    }
    __syncthreads();
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tw0 = tb0 + (int64_t)wv * 64;
    if (tw0 >= tend) return;
    const int64_t tc = t < tend ? t : tend - 1;
    const uint32_t r2 = (uint32_t)(tc - (int64_t)k * plane);
    const uint32_t j = r2 / (uint32_t)G.nx, i = r2 - j * (uint32_t)G.nx;
    uint32_t a8[W];   // byte offset of the lane's entry inside a staged row of doubles, plus the offset of its j block
#pragma unroll
    for (int d = 0; d < W; ++d) {
        const uint32_t va = L.vx[d * G.nx + i];
        a8[d] = (va != 255u ? va : (uint32_t)(RBF_NV - 1)) * 8u + (j - (uint32_t)jA) * (uint32_t)(NT * RBF_NV * 8);
    }
    const uint32_t lane4 = lane * 4u;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(w + (tw0 - reach)), 0, (int)((2 * reach + 64) * 4), 0x00020000);
    float acc = 0.0f;
#pragma unroll
    for (int q0 = 0; q0 < NT; q0 += RBF_APPLY_BATCH) {
        uint32_t wb[RBF_APPLY_BATCH];
#pragma unroll
        for (int u = 0; u < RBF_APPLY_BATCH; ++u) {
            const int q = q0 + u < NT ? q0 + u : NT - 1;
            const int dk = TO.dk[q], dj = TO.dj[q], di = TO.di[q];
            const uint32_t xoff = (uint32_t)(reach + ((int64_t)(dk - R) * G.ny + (dj - R)) * G.nx + (di - R)) * 4u;
            wb[u] = __builtin_amdgcn_raw_buffer_load_b32(rX, (int)lane4, (int)xoff, 0);
        }
#pragma unroll
        for (int u = 0; u < RBF_APPLY_BATCH; ++u) {
            if (q0 + u >= NT) continue;
            const double e = *(const double*)((const char*)sT + a8[TO.di[q0 + u]] + (uint32_t)((q0 + u) * RBF_NV * 8));
            if (e != 0.0) acc = (float)((double)acc + (double)__uint_as_float(wb[u]) * e);
        }
    }
    if (t < tend) out[t] = acc + add;
}

// does the run-time stencil visit the neighbours in the order the kernel above has compiled in?
static bool rbf_stencil_is_canonical(const Stencil& S, int R, int D2)
{
    int q = 0;
    for (int d2 = 0; d2 <= D2; ++d2)
        for (int dz = -R; dz <= R; ++dz)
            for (int dy = -R; dy <= R; ++dy)
                for (int dx = -R; dx <= R; ++dx) {
                    if (dz * dz + dy * dy + dx * dx != d2) continue;
                    if (q >= S.n || S.off[q][0] != dx || S.off[q][1] != dy || S.off[q][2] != dz) return false;
                    ++q;
                }
    return q == S.n;
}
// true if launched; false: this (R, tap_d2) has no table kernel -> the caller evaluates on the fly
static bool launch_rbf_apply_lut(const RbfGeom& G, const RbfLutGeom& LG, const Stencil& host_stencil, unsigned nb, hipStream_t st,
                                 const float* w, const float* tx, const float* ty, const float* tz, const Stencil* d_stencil, float add,
                                 float* out, int64_t t0 = 0, int64_t t1 = -1, int64_t xlo = 0, int64_t xhi = -1)
{
    if (!rbf_stencil_is_canonical(host_stencil, LG.R, LG.tap_d2)) return false;
    const char* ap_env = getenv("R2S_RBF_APPLY");
    RbfWalkArgs WAr;
    if (!ap_env && LG.WA && rbf_walk_supported(LG.R, LG.tap_d2, LG.nx, LG.ny) && rbf_walk_planes(LG, t0, t1, xlo, xhi, &WAr)) {
        WAr.T = LG.WA; WAr.x = w; WAr.y = out; WAr.add = add;
        rbf_walk_launch(1, LG.wa_nv, WAr, st);
        return true;
    }
    if (!LG.TA) return false;
    const bool gathered = ap_env && !strcmp(ap_env, "lutg");   // table entries gathered from L1 / L2 (the tests compare)
    if (LG.R == 2 && LG.tap_d2 == 7 && LG.TA16 && LG.nx >= 256 && tx == G.cx && ty == G.cy && tz == G.cz && !gathered)   // (nx >= 256: a workgroup spans <= 2 rows)
        rbf_apply_lds_kernel<2, 7><<<nb, 256, 0, st>>>(G, LG, w, d_stencil, add, out, t0, t1, xlo, xhi);
    else if (LG.R == 2 && LG.tap_d2 == 7) rbf_apply_lut_kernel<2, 7><<<nb, 256, 0, st>>>(G, LG, w, tx, ty, tz, d_stencil, add, out, t0, t1, xlo, xhi);
    else if (LG.R == 1 && LG.tap_d2 <= 3) rbf_apply_lut_kernel<1, 3><<<nb, 256, 0, st>>>(G, LG, w, tx, ty, tz, d_stencil, add, out, t0, t1, xlo, xhi);
    else if (LG.R == 2 && LG.tap_d2 <= 8) rbf_apply_lut_kernel<2, 8><<<nb, 256, 0, st>>>(G, LG, w, tx, ty, tz, d_stencil, add, out, t0, t1, xlo, xhi);
    else return false;
    return true;
}

// host side of the table: distinct Float32 differences per axis and offset, variant ids per lattice index
// (t: coordinates of the rows / targets, c: of the columns / sources - the same lattice for K, possibly a separately
// rounded one for the evaluation on the "fine" grid at smooth = 1)
// returns the largest number of variants of an offset (<= RBF_NVA - 1: the last slot stays unused on every axis and
// means "no such neighbour"), or 0 when there are more.  The matvec table has room for RBF_NV - 1.
static int rbf_lut_axis(const std::vector<float>& t, const std::vector<float>& c, int R, float vals[7][RBF_NVA], std::vector<uint8_t>& ids)
{
    if (t.size() != c.size()) return 0;
    const int n = (int)c.size(), W = 2 * R + 1;
    int most = 1;
    ids.assign((size_t)W * n, 255);
    for (int d = 0; d < W; ++d) {
        std::vector<float> u;
        for (int i = 0; i < n; ++i) {
            const int ci = i + d - R;
            if (ci < 0 || ci >= n) continue;
            u.push_back(t[i] - c[ci]);   // px - G.cx[ci] of rbf_matvec_kernel (IEEE single subtraction on both sides)
        }
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
        if (u.size() > RBF_NVA - 1) return 0;
        most = std::max(most, (int)u.size());
        for (int q = 0; q < RBF_NVA; ++q) vals[d][q] = q < (int)u.size() ? u[q] : NAN;
        for (int i = 0; i < n; ++i) {
            const int ci = i + d - R;
            if (ci < 0 || ci >= n) continue;
            const float v = t[i] - c[ci];
            ids[(size_t)d * n + i] = (uint8_t)(std::lower_bound(u.begin(), u.end(), v) - u.begin());
        }
    }
    return most;
}
// ---- the evaluation on a REFINED grid (rbf_grid = :fine, smooth = s >= 2) through tables ----------------------------
// A target (i, j, k) of parity class (i % s, j % s, k % s) sees the sources around (i / s, j / s, k / s) through the
// stencil of its class (build_stencil); the Float32 coordinate differences per axis, parity and offset take a few dozen
// distinct values, so the kernel values of a class are a table [neighbour][z variant][y variant][x variant] evaluated once
// with rbf_apply_kernel's arithmetic.  One WAVEFRONT = 64 targets of ONE parity in x of one row: the stencil, the y / z
// variants and the table row are wave-uniform (scalar loads), the sources of a neighbour are 64 consecutive weights, and a
// neighbour costs two vector loads + the Float64 product and sum instead of ~60 instructions of distance and exp.  Same
// values, same order, same skips (sources outside the lattice; dist > max_distance <=> entry 0) => the same numbers.
struct FineTap {
    int32_t woff;              // (dk ny + dj) nx + di
    signed char di, dj, dk, pad;
};
struct FineLut {
    int s, fx, fy, nx, ny, nzc, nmx, nmy, nmz, nvf, nrec;
    const uint64_t *idx, *idy, *idz;   // [parity][index / s]: the variant ids of the 8 offsets -3..4, a byte each (255: no such source)
    const double* T;
    const FineTap* taps;       // [class][512]
    const int64_t* tbase;      // [class]: first entry of the class's table
    const int* ntaps;          // [class]
};

static bool fine_lut_axis(const std::vector<float>& t, const std::vector<float>& c, int s, std::vector<uint64_t>& packs,
                          std::vector<float>& vals /* [s][8][64] */, int& most)
{
    const int tn = (int)t.size(), cn = (int)c.size(), nm = (tn + s - 1) / s;
    packs.assign((size_t)s * nm, ~0ull);
    vals.assign((size_t)s * 8 * 64, NAN);
    for (int p = 0; p < s; ++p)
        for (int d8 = 0; d8 < 8; ++d8) {
            const int d = d8 - 3;
            std::vector<float> u;
            for (int m = 0; m * s + p < tn; ++m)
                if (m + d >= 0 && m + d < cn) u.push_back(t[(size_t)m * s + p] - c[(size_t)(m + d)]);   // px - G.cx[ci] of rbf_apply_point
            std::sort(u.begin(), u.end());
            u.erase(std::unique(u.begin(), u.end()), u.end());
            if (u.size() > 63) return false;
            most = std::max(most, (int)u.size());
            for (size_t q = 0; q < u.size(); ++q) vals[((size_t)p * 8 + d8) * 64 + q] = u[q];
            for (int m = 0; m * s + p < tn; ++m) {
                if (m + d < 0 || m + d >= cn) continue;
                const float v = t[(size_t)m * s + p] - c[(size_t)(m + d)];
                const uint64_t id = (uint64_t)(std::lower_bound(u.begin(), u.end(), v) - u.begin());
                uint64_t& pk = packs[(size_t)p * nm + m];
                pk = (pk & ~(0xffull << (8 * d8))) | (id << (8 * d8));
            }
        }
    return true;
}

__global__ void __launch_bounds__(256) fine_lut_build_kernel(const FineTap* __restrict__ taps, int ntaps, const float* __restrict__ vx,
                                                            const float* __restrict__ vy, const float* __restrict__ vz, double sigma,
                                                            float max_distance, int nvf, double* __restrict__ T)
{
    __shared__ double etab[64];
    if (threadIdx.x < 64) etab[threadIdx.x] = c_exp2_neg_64[threadIdx.x];
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)ntaps * nvf * nvf * nvf) return;
    const int a = (int)(t % nvf), b = (int)((t / nvf) % nvf), c = (int)((t / ((int64_t)nvf * nvf)) % nvf);
    const int q = (int)(t / ((int64_t)nvf * nvf * nvf));
    const FineTap tp = taps[q];
    const float dx = vx[(tp.di + 3) * 64 + a], dy = vy[(tp.dj + 3) * 64 + b], dz = vz[(tp.dk + 3) * 64 + c];
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    const double inv_sigma = 1.0 / sigma;
    double val = 0.0;
    if (dist <= max_distance) {   // (NaN differences of unused variant slots: false)
        const double u = (double)dist * inv_sigma;
        val = exp_neg_fast(u * u, etab);
    }
    T[t] = val;
}

__global__ void __launch_bounds__(256) rbf_apply_fine_lut_kernel(FineLut F, const float* __restrict__ w, float add, float* __restrict__ out,
                                                                int kf0, int kf1)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    // per wavefront and neighbour: byte offset of its table row, byte offset of its weights, shift of the x variant id
    // (F.nrec entries per wavefront: the longest list of the classes, rounded up to batches)
    extern __shared__ u32x4 s_rec_all[];
    u32x4* const s_rec = s_rec_all + (size_t)(threadIdx.x >> 6) * F.nrec;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t nseg = (uint32_t)(F.nmx + 63) / 64u;
    const uint32_t per_row = (uint32_t)F.s * nseg;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wv);   // (the host keeps the count below 2^32)
    const uint32_t row = wave / per_row;
    const uint32_t rem = wave - row * per_row;
    const int k = kf0 + (int)(row / (uint32_t)F.fy), j = (int)(row % (uint32_t)F.fy);
    const bool live = k < kf1;   // (wavefronts past the end of the chunk keep the barrier company)
    const int kk = live ? k : kf0;
    const int pi = (int)(rem / nseg), seg = (int)(rem % nseg);
    const int pk = kk % F.s, bk = kk / F.s, pj = j % F.s, bj = j / F.s;
    const int m = seg * 64 + (int)lane, i = m * F.s + pi;
    const bool active = live && i < F.fx;
    const int cls = (pk * F.s + pj) * F.s + pi;
    const FineTap* __restrict__ taps = F.taps + (size_t)cls * 512;
    const int ntaps = __builtin_amdgcn_readfirstlane(F.ntaps[cls]);
    const uint32_t nvf = (uint32_t)F.nvf;
    // "no such source" is variant slot nvf - 1 on every axis: no difference has that id, its table entries are 0 (built
    // from NaN differences) = "beyond max_distance", and whatever weight the address of such a neighbour holds is not used
    const uint64_t pz = F.idz[(size_t)pk * F.nmz + bk], py = F.idy[(size_t)pj * F.nmy + bj];
    const uint64_t px = active ? F.idx[(size_t)pi * F.nmx + m] : F.idx[(size_t)F.s * F.nmx];   // (an all-absent pack behind the axis)
    // what is the same for the 64 targets of the wavefront, once per neighbour (lane q prepares neighbour q): everything
    // but the x variant.  A batch of the loop below may run past the end of the list: such entries get a table offset
    // beyond the class's table
    const int nrec = (ntaps + 7) & ~7;
    for (int q = (int)lane; q < nrec; q += 64) {
        const FineTap tp = taps[q];
        const uint32_t c = (uint32_t)(pz >> (8 * (tp.dk + 3))) & 255u, b = (uint32_t)(py >> (8 * (tp.dj + 3))) & 255u;
        u32x4 r;
        r.x = q < ntaps ? ((((uint32_t)q * nvf + c) * nvf + b) * nvf) * 8u : 0x80000000u;   // (no wrap-around when the x variant is added)
        r.y = (uint32_t)tp.woff * 4u;
        r.z = (uint32_t)(8 * (tp.di + 3));
        r.w = 0u;
        s_rec[q] = r;
    }
    __syncthreads();
    if (!live) return;
    // buffer loads: out-of-range offsets read 0
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((uint32_t)F.nx * (uint32_t)F.ny * (uint32_t)F.nzc * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void*)(F.T + F.tbase[cls]), 0, (int)((uint32_t)ntaps * nvf * nvf * nvf * 8u), 0x00020000);
    const uint32_t wlane = (uint32_t)((((int64_t)bk * F.ny + bj) * F.nx + m) * 4);
    float acc = 0.0f;
    // batches of FINE_BATCH neighbours: the loads of a batch are in flight together (a neighbour at a time the kernel waits
    // for two dependent loads per neighbour: 26 ms instead of the 22 of the on-the-fly evaluation at 257^3 -> 513^3)
    // (issuing the loads of the next batch before the sums of this one: 9.8 instead of 8.8 ms)
    constexpr int FINE_BATCH = 8;
    for (int q0 = 0; q0 < ntaps; q0 += FINE_BATCH) {
        uint32_t wb[FINE_BATCH];
        u32x2 eb[FINE_BATCH];
#pragma unroll
        for (int u = 0; u < FINE_BATCH; ++u) {
            const u32x4 r = s_rec[q0 + u];   // (one address for the wavefront: a broadcast read)
            const uint32_t a = (uint32_t)(px >> r.z) & 255u;
            wb[u] = __builtin_amdgcn_raw_buffer_load_b32(rW, (int)(wlane + r.y), 0, 0);
            eb[u] = __builtin_amdgcn_raw_buffer_load_b64(rT, (int)(r.x + a * 8u), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < FINE_BATCH; ++u) {
            const double e = __hiloint2double((int)eb[u].y, (int)eb[u].x);
            const float nv = (float)((double)acc + (double)__uint_as_float(wb[u]) * e);
            acc = e != 0.0 ? nv : acc;   // (a select, not a branch: the skipped neighbours are few)
        }
    }
    if (active) out[((int64_t)k * F.fy + j) * F.fx + i] = acc + add;
}

// process_vector (:15-22), pass 1: max |v| over |v| < 1e9 (as Float32 bits, all non-negative)
__global__ void pv_max_kernel(const double* __restrict__ v, int64_t n, float* __restrict__ f, uint32_t* __restrict__ maxbits,
                              uint32_t* __restrict__ any)
{
    // grid-stride: a fixed grid keeps the number of same-address atomics at one per wavefront of the GRID
    // (one per wavefront of the DATA was 2.1 M atomics = 19 ms at 512^3)
    uint32_t bits = 0;
    bool has = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = (float)v[i];
        f[i] = x;
        const float a = fabsf(x);
        if (a < 1.0e9f) {
            const uint32_t b = __float_as_uint(a);
            bits = b > bits ? b : bits;
            has = true;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(bits, off, 64);
        bits = o > bits ? o : bits;
    }
    const bool anyw = __any(has);
    // one same-address atomic per BLOCK (through LDS): 2 048 of them instead of 8 192 (8 ms at 512^3)
    __shared__ uint32_t s_bits[4], s_any[4];
    if ((threadIdx.x & 63) == 0) { s_bits[threadIdx.x >> 6] = bits; s_any[threadIdx.x >> 6] = anyw ? 1u : 0u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t b = 0, a = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { b = s_bits[w] > b ? s_bits[w] : b; a |= s_any[w]; }
        if (a) {
            atomicMax(maxbits, b);
            *any = 1u;
        }
    }
}
__global__ void pv_replace_kernel(float* __restrict__ f, int64_t n, const uint32_t* __restrict__ maxbits)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float mx = __uint_as_float(*maxbits);
    const float x = f[i], a = fabsf(x);
    const float rtol = 3.4526698e-4f;   // sqrt(eps(Float32)): isapprox default
    const float big = a > 1.0e10f ? a : 1.0e10f;
    if (fabsf(a - 1.0e10f) <= rtol * big) f[i] = (x > 0 ? 1.0f : (x < 0 ? -1.0f : 0.0f)) * mx;
}

// Float32 vectors, dot products accumulated in Float64 (BLAS sdot/snrm2 stand-in)
__global__ void __launch_bounds__(256) dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                 double* __restrict__ partial)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc += (double)a[i] * (double)b[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// Dot product as DOT_PARTS Float64 partial sums per Z plane (fixed tree inside each part); the parts are added in
// (k, part) order on the host.  The result does not depend on how the planes are spread over devices: a Z-slab run of
// the CG follows the single-device run bit for bit.
#define DOT_PARTS 4
__global__ void __launch_bounds__(256) dot_planes_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t plane,
                                                        double* __restrict__ partial)
{
    __shared__ double red[256];
    const int64_t k = blockIdx.x / DOT_PARTS, part = blockIdx.x % DOT_PARTS;
    const int64_t chunk = (plane + DOT_PARTS - 1) / DOT_PARTS, lo = part * chunk, hi = (lo + chunk < plane) ? lo + chunk : plane;
    const float* __restrict__ pa = a + k * plane;
    const float* __restrict__ pb = b + k * plane;
    double acc = 0.0;
    int64_t i = lo + threadIdx.x;
    for (; i + 7 * 256 < hi; i += 8 * 256) {   // (8 pairs of loads in flight, the additions in the same order)
        float va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { va[u] = pa[i + u * 256]; vb[u] = pb[i + u * 256]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (double)va[u] * (double)vb[u];
    }
    for (; i < hi; i += 256) acc += (double)pa[i] * (double)pb[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// x += alpha u, r -= alpha q and the parts of dot(r, r) in one pass: the grid, the element -> thread assignment and the
// order of the additions are those of dot_planes_kernel(r, r), so the partial sums come out bit for bit the same
__global__ void __launch_bounds__(256) cg_update_xr_dot_kernel(float* __restrict__ x, float* __restrict__ r, const float* __restrict__ u,
                                                              const float* __restrict__ q, float alpha, int64_t plane,
                                                              double* __restrict__ partial)
{
    __shared__ double red[256];
    const int64_t k = blockIdx.x / DOT_PARTS, part = blockIdx.x % DOT_PARTS;
    const int64_t chunk = (plane + DOT_PARTS - 1) / DOT_PARTS, lo = part * chunk, hi = (lo + chunk < plane) ? lo + chunk : plane;
    const int64_t o = k * plane;
    double acc = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        x[o + i] += alpha * u[o + i];
        const float rn = r[o + i] - alpha * q[o + i];
        r[o + i] = rn;
        acc += (double)rn * (double)rn;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// r -= alpha q and the parts of dot(r, r): cg_update_xr_dot_kernel without the weights' step (which the next product takes
// over, cg_update_wu_kernel) - the same grid, assignment and order of additions
__global__ void __launch_bounds__(256) cg_update_r_dot_kernel(float* __restrict__ r, const float* __restrict__ q, float alpha, int64_t plane,
                                                             double* __restrict__ partial)
{
    __shared__ double red[256];
    const int64_t k = blockIdx.x / DOT_PARTS, part = blockIdx.x % DOT_PARTS;
    const int64_t chunk = (plane + DOT_PARTS - 1) / DOT_PARTS, lo = part * chunk, hi = (lo + chunk < plane) ? lo + chunk : plane;
    const int64_t o = k * plane;
    double acc = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float rn = r[o + i] - alpha * q[o + i];
        r[o + i] = rn;
        acc += (double)rn * (double)rn;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// w += alpha_prev u (the step of the previous iteration) and u = r + beta u in one pass (16 bytes read, 8 written per voxel)
__global__ void __launch_bounds__(256) cg_update_wu_kernel(float* __restrict__ w, float* __restrict__ u, const float* __restrict__ r,
                                                          float alpha_prev, float beta, int64_t n)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n) {
        float4 W = *(const float4*)(w + i), U = *(const float4*)(u + i);
        const float4 Rr = *(const float4*)(r + i);
        W.x += alpha_prev * U.x; W.y += alpha_prev * U.y; W.z += alpha_prev * U.z; W.w += alpha_prev * U.w;
        U.x = Rr.x + beta * U.x; U.y = Rr.y + beta * U.y; U.z = Rr.z + beta * U.z; U.w = Rr.w + beta * U.w;
        *(float4*)(w + i) = W;
        *(float4*)(u + i) = U;
    } else {
        for (int64_t q = i; q < n; ++q) {
            const float uo = u[q];
            w[q] += alpha_prev * uo;
            u[q] = r[q] + beta * uo;
        }
    }
}
__global__ void cg_axpy_kernel(float* __restrict__ x, const float* __restrict__ u, float alpha, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += alpha * u[i];
}
__global__ void __launch_bounds__(256) sum_f64_kernel(const double* __restrict__ in, int n, double* __restrict__ out)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += in[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}
__global__ void cg_update_u_kernel(float* __restrict__ u, const float* __restrict__ r, float beta, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) u[i] = r[i] + beta * u[i];
}
__global__ void minmax_kernel(const float* __restrict__ v, int64_t n, int* __restrict__ mm)
{
    // order-preserving int encoding of floats
    int lo = 0x7FFFFFFF, hi = (int)0x80000000;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int b = __float_as_int(v[i]);
        b = b >= 0 ? b : (b ^ 0x7FFFFFFF);
        lo = b < lo ? b : lo;
        hi = b > hi ? b : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int ol = __shfl_xor(lo, off, 64), oh = __shfl_xor(hi, off, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}

static void coarse_coords(double mn, double mx, int n, std::vector<float>& c)
{
    // create_grid (:36-46): Float32 range(min, max, length)
    c.resize(n);
    const double a = (double)(float)mn, b = (double)(float)mx;
    for (int i = 0; i < n; ++i) c[i] = (float)(a + (double)i * (b - a) / (double)(n - 1));
    c[n - 1] = (float)mx;
}

struct RbfWork {   // the device buffers of one rbf_smooth_host call
    DevBuf b[40];
    VolumeWork vw;
    HostMailbox mb;
    void release()
    {
        mb.release();
        for (DevBuf& x : b) x.release();
        vw.release();
    }
};
// sdf_dev / out_dev: `sdf` / `fine_out` are device pointers (device-resident chaining of the stages)
static int rbf_smooth_host(const double* sdf, const r2s_grid* g, int is_interp, int smooth, double kthr,
                           double target_volume, float* fine_out, float* th_out, int* cg_iters, float* lsf_out,
                           bool sdf_dev = false, bool out_dev = false,
                           const std::function<int(int64_t, int64_t)>* fine_chunk = nullptr, RbfWork* ws = nullptr,
                           bool fine_early = false)
{
    if (!sdf || !g || !fine_out) return fail(R2S_ERR_ARG, "null argument");
    if (smooth < 1 || smooth > 4) return fail(R2S_ERR_ARG, "smooth must be 1..4");
    if (!(kthr > 0.0 && kthr < 1.0)) return fail(R2S_ERR_ARG, "kernel threshold must be in (0,1)");
    const int nx = (int)g->N[0] + 1, ny = (int)g->N[1] + 1, nz = (int)g->N[2] + 1;
    const int64_t n = (int64_t)nx * ny * nz;
    const int fx = (int)g->N[0] * smooth + 1, fy = (int)g->N[1] * smooth + 1, fz = (int)g->N[2] * smooth + 1;
    const int64_t nf = (int64_t)fx * fy * fz;
    hipStream_t st = nullptr;
    // (a caller that repeats the call keeps the ~30 buffers - 4.6 GB at 512^3 - in a workspace: allocating and freeing
    //  them costs ~5 ms per call)
    RbfWork local;
    RbfWork& W = ws ? *ws : local;
    DevBuf &d_sdf = W.b[0], &d_f = W.b[1], &d_w = W.b[2], &d_lsf = W.b[3], &d_fine = W.b[4], &d_cx = W.b[5], &d_cy = W.b[6], &d_cz = W.b[7],
           &d_tx = W.b[8], &d_ty = W.b[9], &d_tz = W.b[10], &d_st = W.b[11], &d_cnt = W.b[12], &d_r = W.b[13], &d_u = W.b[14], &d_q = W.b[15],
           &d_part = W.b[16], &d_sum = W.b[17], &d_lut = W.b[18], &d_luta = W.b[19], &d_vx = W.b[20], &d_vy = W.b[21], &d_vz = W.b[22],
           &d_lutf = W.b[23], &d_fvx = W.b[24], &d_fvy = W.b[25], &d_fvz = W.b[26], &d_lv = W.b[27], &d_lvf = W.b[28], &d_luta16 = W.b[29],
           &d_wt = W.b[30], &d_wa = W.b[31], &d_waf = W.b[32], &d_part2 = W.b[33], &d_fl_ids = W.b[34], &d_fl_vals = W.b[35],
           &d_fl_T = W.b[36], &d_fl_taps = W.b[37], &d_fl_meta = W.b[38];
    VolumeWork& vw = W.vw;
    auto cleanup = [&]() {
        if (!ws) W.release();
    };
#define TRY_C(expr)                                                                   \
    do {                                                                              \
        int rc_ = (expr);                                                             \
        if (rc_) { cleanup(); return rc_; }                                           \
    } while (0)
#define HIP_C(expr)                                                                   \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) { cleanup(); return fail(R2S_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } \
    } while (0)
#define ENSURE_C(buf, bytes)                                                          \
    do {                                                                              \
        if ((buf).ensure(bytes)) { cleanup(); return fail(R2S_ERR_NOMEM, "hipMalloc of %zu bytes failed", (size_t)(bytes)); } \
    } while (0)
    if (!sdf_dev) ENSURE_C(d_sdf, sizeof(double) * (size_t)n);
    ENSURE_C(d_f, sizeof(float) * (size_t)n);
    ENSURE_C(d_w, sizeof(float) * (size_t)n);
    ENSURE_C(d_lsf, sizeof(float) * (size_t)n);
    if (!out_dev) ENSURE_C(d_fine, sizeof(float) * (size_t)nf);
    ENSURE_C(d_cnt, 64);
    if (!sdf_dev) HIP_C(hipMemcpy(d_sdf.p, sdf, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    const double* dsdf = sdf_dev ? sdf : d_sdf.as<double>();
    float* dfine = out_dev ? fine_out : d_fine.as<float>();
    // ---- process_vector ----
    HIP_C(hipMemset(d_cnt.p, 0, 64));
    const unsigned nb = (unsigned)((n + 255) / 256);
    pv_max_kernel<<<(nb < 2048u ? nb : 2048u), 256, 0, st>>>(dsdf, n, d_f.as<float>(), d_cnt.as<uint32_t>(), d_cnt.as<uint32_t>() + 1);
    uint32_t hc[2];
    TRY_C(d2h_small(hc, d_cnt.p, 8, st, W.mb));
    if (!hc[1]) { cleanup(); return fail(R2S_ERR_ARG, "every SDF value is a sentinel: nothing to smooth"); }   // A15
    pv_replace_kernel<<<nb, 256, 0, st>>>(d_f.as<float>(), n, d_cnt.as<uint32_t>());
    // ---- geometry ----
    std::vector<float> cx, cy, cz, tx(fx), ty(fy), tz(fz);
    coarse_coords(g->aabb_min[0], g->aabb_max[0], nx, cx);
    coarse_coords(g->aabb_min[1], g->aabb_max[1], ny, cy);
    coarse_coords(g->aabb_min[2], g->aabb_max[2], nz, cz);
    {
        // create_smooth_grid (:60-74): one step dx (from the x axis) for all three axes
        const float xmin = (float)g->aabb_min[0], xmax = (float)g->aabb_max[0], ymin = (float)g->aabb_min[1],
                    zmin = (float)g->aabb_min[2];
        const float dx = (xmax - xmin) / (float)(fx - 1);
        for (int i = 0; i < fx; ++i) tx[i] = xmin + (float)i * dx;
        for (int i = 0; i < fy; ++i) ty[i] = ymin + (float)i * dx;
        for (int i = 0; i < fz; ++i) tz[i] = zmin + (float)i * dx;
    }
    auto up = [&](DevBuf& b, const std::vector<float>& v) -> int {
        if (b.ensure(sizeof(float) * v.size())) return fail(R2S_ERR_NOMEM, "hipMalloc failed");
        if (hipMemcpy(b.p, v.data(), sizeof(float) * v.size(), hipMemcpyHostToDevice) != hipSuccess)
            return fail(R2S_ERR_HIP, "hipMemcpy failed");
        return 0;
    };
    TRY_C(up(d_cx, cx)); TRY_C(up(d_cy, cy)); TRY_C(up(d_cz, cz));
    TRY_C(up(d_tx, tx)); TRY_C(up(d_ty, ty)); TRY_C(up(d_tz, tz));
    // stencils for smooth = 1 (one class) and for the fine grid (smooth^3 classes)
    std::vector<Stencil> sts(1 + (size_t)smooth * smooth * smooth);
    {
        int fr0[3] = {0, 0, 0};
        build_stencil(1, fr0, &sts[0], -std::log(kthr));
        for (int a = 0; a < smooth; ++a)
            for (int b = 0; b < smooth; ++b)
                for (int c = 0; c < smooth; ++c) {
                    int fr[3] = {c, b, a};
                    build_stencil(smooth, fr, &sts[1 + (a * smooth + b) * smooth + c], -std::log(kthr));
                }
    }
    ENSURE_C(d_st, sizeof(Stencil) * sts.size());
    HIP_C(hipMemcpy(d_st.p, sts.data(), sizeof(Stencil) * sts.size(), hipMemcpyHostToDevice));
    RbfGeom G;
    G.nx = nx; G.ny = ny; G.nz = nz;
    G.cx = d_cx.as<float>(); G.cy = d_cy.as<float>(); G.cz = d_cz.as<float>();
    G.sigma = g->cell_size;                                                    // :346
    G.thr = kthr;
    G.max_distance = (float)std::sqrt(-std::log(kthr) * G.sigma * G.sigma);     // :221
    {
        const double R2 = -std::log(kthr);                 // support radius^2 in cells (sigma = cell size)
        G.tap_d2 = (int)std::floor(R2 * 1.05 + 0.25);      // 1e-3 -> 7 (i.e. 6: 7 is not a sum of three squares)
        G.tap_r = (int)std::floor(std::sqrt((double)G.tap_d2));
    }
    // ---- refined output grid: the tables of its parity classes (R2S_RBF_APPLY=fly: on the fly, the tests compare) ----
    FineLut FL;
    memset(&FL, 0, sizeof FL);
    if (smooth >= 2 && !(getenv("R2S_RBF_APPLY") && getenv("R2S_RBF_APPLY")[0] == 'f')) {
        const int ncls = smooth * smooth * smooth;
        std::vector<uint64_t> pkx, pky, pkz;
        std::vector<float> vlx, vly, vlz;
        int most = 1;
        const bool fits = fine_lut_axis(tx, cx, smooth, pkx, vlx, most) && fine_lut_axis(ty, cy, smooth, pky, vly, most) &&
                          fine_lut_axis(tz, cz, smooth, pkz, vlz, most);
        const int nvf = most <= 15 ? 16 : (most <= 31 ? 32 : 64);
        for (std::vector<uint64_t>* pkp : {&pkx, &pky, &pkz})   // "no such source" = the unused slot nvf - 1 (see the kernel)
            for (uint64_t& v : *pkp)
                for (int d8 = 0; d8 < 8; ++d8)
                    if (((v >> (8 * d8)) & 255u) == 255u) v = (v & ~(0xffull << (8 * d8))) | ((uint64_t)(nvf - 1) << (8 * d8));
        {
            uint64_t none = 0;
            for (int d8 = 0; d8 < 8; ++d8) none |= (uint64_t)(nvf - 1) << (8 * d8);
            pkx.push_back(none);   // (the pack of the lanes beyond the end of a row)
        }
        std::vector<FineTap> taps((size_t)ncls * 512);
        std::vector<int64_t> tbase((size_t)ncls);
        std::vector<int> ntaps((size_t)ncls);
        int64_t total = 0;
        bool ok = fits;
        for (int c = 0; c < ncls && ok; ++c) {
            const Stencil& S = sts[1 + (size_t)c];
            if (S.n > 512 - 8) ok = false;   // (the kernel reads its batches past the end of a list)
            tbase[(size_t)c] = total;
            ntaps[(size_t)c] = S.n;
            total += (int64_t)S.n * nvf * nvf * nvf;
            for (int q = 0; q < S.n && ok; ++q) {
                FineTap& tp = taps[(size_t)c * 512 + q];
                tp.di = S.off[q][0]; tp.dj = S.off[q][1]; tp.dk = S.off[q][2]; tp.pad = 0;
                if (tp.di < -3 || tp.di > 4 || tp.dj < -3 || tp.dj > 4 || tp.dk < -3 || tp.dk > 4) ok = false;
                tp.woff = (int32_t)(((int64_t)tp.dk * ny + tp.dj) * nx + tp.di);
            }
        }
        const int64_t waves = (int64_t)fz * fy * smooth * (((fx + smooth - 1) / smooth + 63) / 64);
        if (ok && total * 8 <= ((int64_t)8 << 30) && n * 4 < ((int64_t)1 << 32) && waves < ((int64_t)1 << 32)) {
            const size_t nid = pkx.size() + pky.size() + pkz.size();
            ENSURE_C(d_fl_ids, 8 * nid);
            ENSURE_C(d_fl_vals, 4 * (vlx.size() + vly.size() + vlz.size()));
            ENSURE_C(d_fl_T, 8 * (size_t)total);
            ENSURE_C(d_fl_taps, sizeof(FineTap) * taps.size());
            ENSURE_C(d_fl_meta, 8 * (size_t)ncls + 4 * (size_t)ncls);
            uint64_t* dids = d_fl_ids.as<uint64_t>();
            HIP_C(hipMemcpy(dids, pkx.data(), 8 * pkx.size(), hipMemcpyHostToDevice));
            HIP_C(hipMemcpy(dids + pkx.size(), pky.data(), 8 * pky.size(), hipMemcpyHostToDevice));
            HIP_C(hipMemcpy(dids + pkx.size() + pky.size(), pkz.data(), 8 * pkz.size(), hipMemcpyHostToDevice));
            float* dv = d_fl_vals.as<float>();
            HIP_C(hipMemcpy(dv, vlx.data(), 4 * vlx.size(), hipMemcpyHostToDevice));
            HIP_C(hipMemcpy(dv + vlx.size(), vly.data(), 4 * vly.size(), hipMemcpyHostToDevice));
            HIP_C(hipMemcpy(dv + vlx.size() + vly.size(), vlz.data(), 4 * vlz.size(), hipMemcpyHostToDevice));
            HIP_C(hipMemcpy(d_fl_taps.p, taps.data(), sizeof(FineTap) * taps.size(), hipMemcpyHostToDevice));
            HIP_C(hipMemcpy(d_fl_meta.p, tbase.data(), 8 * (size_t)ncls, hipMemcpyHostToDevice));
            HIP_C(hipMemcpy((char*)d_fl_meta.p + 8 * (size_t)ncls, ntaps.data(), 4 * (size_t)ncls, hipMemcpyHostToDevice));
            const double sigma = g->cell_size;
            const float maxd = (float)std::sqrt(-std::log(kthr) * sigma * sigma);
            for (int c = 0; c < ncls; ++c) {
                const int pi = c % smooth, pj = (c / smooth) % smooth, pk = c / (smooth * smooth);
                const int64_t nt = (int64_t)ntaps[(size_t)c] * nvf * nvf * nvf;
                if (!nt) continue;
                fine_lut_build_kernel<<<(unsigned)((nt + 255) / 256), 256, 0, st>>>(
                    d_fl_taps.as<FineTap>() + (size_t)c * 512, ntaps[(size_t)c], dv + (size_t)pi * 8 * 64, dv + vlx.size() + (size_t)pj * 8 * 64,
                    dv + vlx.size() + vly.size() + (size_t)pk * 8 * 64, sigma, maxd, nvf, d_fl_T.as<double>() + tbase[(size_t)c]);
            }
            FL.s = smooth; FL.fx = fx; FL.fy = fy; FL.nx = nx; FL.ny = ny; FL.nzc = nz;
            FL.nmx = (fx + smooth - 1) / smooth; FL.nmy = (fy + smooth - 1) / smooth; FL.nmz = (fz + smooth - 1) / smooth;
            FL.nvf = nvf;
            FL.nrec = 8;
            for (int c = 0; c < ncls; ++c) FL.nrec = std::max(FL.nrec, (ntaps[(size_t)c] + 7) & ~7);
            FL.idx = dids; FL.idy = dids + pkx.size(); FL.idz = dids + pkx.size() + pky.size();
            FL.T = d_fl_T.as<double>();
            FL.taps = d_fl_taps.as<FineTap>();
            FL.tbase = (const int64_t*)d_fl_meta.p;
            FL.ntaps = (const int*)((const char*)d_fl_meta.p + 8 * (size_t)ncls);
        }
    }
    // ---- tables of the distinct kernel values: the CG matvec (T) and the evaluation on the same grid (TA) ----
    // (exact, no matrix in memory; R2S_RBF_MATVEC=k|fly and R2S_RBF_APPLY=fly force the other paths - the tests compare them)
    const char* mv_env = getenv("R2S_RBF_MATVEC");   // (read per call: the tests switch)
    const char* ap_env = getenv("R2S_RBF_APPLY");
    const bool want_mv_lut = is_interp && !(mv_env && (mv_env[0] == 'k' || mv_env[0] == 'f'));
    const bool want_ap_lut = !(ap_env && ap_env[0] == 'f');
    RbfLutGeom LG, LGF;   // LGF: the evaluation of the output field at smooth = 1 (its grid is rounded separately)
    memset(&LG, 0, sizeof LG);
    memset(&LGF, 0, sizeof LGF);
    const bool fine_one_to_one = smooth == 1 && fx == nx && fy == ny && fz == nz;
    const bool walk_ok = rbf_walk_supported(G.tap_r, G.tap_d2, nx, ny);
    if (G.tap_r >= 1 && G.tap_r <= 3 && want_ap_lut && fine_one_to_one) {
        RbfLutVals LV;
        memset(&LV, 0, sizeof LV);
        LV.R = G.tap_r; LV.sigma = G.sigma; LV.thr = G.thr; LV.max_distance = G.max_distance;
        std::vector<uint8_t> ix, iy, iz;
        const int f0 = rbf_lut_axis(tx, cx, G.tap_r, LV.v[0], ix), f1 = rbf_lut_axis(ty, cy, G.tap_r, LV.v[1], iy),
                  f2 = rbf_lut_axis(tz, cz, G.tap_r, LV.v[2], iz);
        if (f0 && f1 && f2) {
            const int W = 2 * G.tap_r + 1;
            ENSURE_C(d_fvx, ix.size()); ENSURE_C(d_fvy, iy.size()); ENSURE_C(d_fvz, iz.size());
            HIP_C(hipMemcpy(d_fvx.p, ix.data(), ix.size(), hipMemcpyHostToDevice));
            HIP_C(hipMemcpy(d_fvy.p, iy.data(), iy.size(), hipMemcpyHostToDevice));
            HIP_C(hipMemcpy(d_fvz.p, iz.data(), iz.size(), hipMemcpyHostToDevice));
            LGF.nx = nx; LGF.ny = ny; LGF.nz = nz; LGF.R = G.tap_r; LGF.tap_d2 = G.tap_d2;
            LGF.vx = d_fvx.as<uint8_t>(); LGF.vy = d_fvy.as<uint8_t>(); LGF.vz = d_fvz.as<uint8_t>();
            ENSURE_C(d_lvf, sizeof LV);
            HIP_C(hipMemcpy(d_lvf.p, &LV, sizeof LV, hipMemcpyHostToDevice));
            if (walk_ok && !ap_env) {   // the table of the row-walk kernel (r2s_rbf_walk.hpp)
                const int nv = std::max(f0, std::max(f1, f2)) <= RBF_NV - 1 ? RBF_NV : RBF_NVA;
                const size_t nT = (size_t)W * W * W * nv * nv * nv;
                ENSURE_C(d_waf, sizeof(double) * nT);
                rbf_walk_table_kernel<1><<<(unsigned)((nT + 255) / 256), 256, 0, st>>>(d_lvf.as<RbfLutVals>(), d_waf.p, nv);
                LGF.WA = d_waf.as<double>(); LGF.wa_nv = nv;
            } else {
                const size_t nT = (size_t)W * W * W * RBF_NVA * RBF_NVA * RBF_NVA;
                ENSURE_C(d_lutf, sizeof(double) * nT);
                rbf_lut_build_apply_kernel<<<(unsigned)((nT + 255) / 256), 256, 0, st>>>(d_lvf.as<RbfLutVals>(), d_lutf.as<double>());
                LGF.TA = d_lutf.as<double>();
            }
        }
    }
    if (G.tap_r >= 1 && G.tap_r <= 3 && (want_mv_lut || want_ap_lut)) {
        RbfLutVals LV;
        memset(&LV, 0, sizeof LV);
        LV.R = G.tap_r; LV.sigma = G.sigma; LV.thr = G.thr; LV.max_distance = G.max_distance;
        std::vector<uint8_t> ix, iy, iz;
        const int m0 = rbf_lut_axis(cx, cx, G.tap_r, LV.v[0], ix), m1 = rbf_lut_axis(cy, cy, G.tap_r, LV.v[1], iy),
                  m2 = rbf_lut_axis(cz, cz, G.tap_r, LV.v[2], iz);
        const bool mv_fits = std::max(m0, std::max(m1, m2)) <= RBF_NV - 1;
        if (m0 && m1 && m2) {
            const int W = 2 * G.tap_r + 1;
            const size_t nT = (size_t)W * W * W * RBF_NV * RBF_NV * RBF_NV;
            ENSURE_C(d_vx, ix.size()); ENSURE_C(d_vy, iy.size()); ENSURE_C(d_vz, iz.size());
            HIP_C(hipMemcpy(d_vx.p, ix.data(), ix.size(), hipMemcpyHostToDevice));
            HIP_C(hipMemcpy(d_vy.p, iy.data(), iy.size(), hipMemcpyHostToDevice));
            HIP_C(hipMemcpy(d_vz.p, iz.data(), iz.size(), hipMemcpyHostToDevice));
            LG.nx = nx; LG.ny = ny; LG.nz = nz; LG.R = G.tap_r; LG.tap_d2 = G.tap_d2;
            LG.vx = d_vx.as<uint8_t>(); LG.vy = d_vy.as<uint8_t>(); LG.vz = d_vz.as<uint8_t>();
            ENSURE_C(d_lv, sizeof LV);
            HIP_C(hipMemcpy(d_lv.p, &LV, sizeof LV, hipMemcpyHostToDevice));
            if (want_mv_lut && mv_fits && walk_ok && !mv_env) {
                ENSURE_C(d_wt, sizeof(float) * nT);
                rbf_walk_table_kernel<0><<<(unsigned)((nT + 255) / 256), 256, 0, st>>>(d_lv.as<RbfLutVals>(), d_wt.p, RBF_NV);
                LG.WT = d_wt.as<float>();
            } else if (want_mv_lut && mv_fits) {
                ENSURE_C(d_lut, sizeof(float) * nT);
                rbf_lut_build_kernel<<<(unsigned)((nT + 255) / 256), 256, 0, st>>>(d_lv.as<RbfLutVals>(), d_lut.as<float>());
                LG.T = d_lut.as<float>();
            }
            if (want_ap_lut && walk_ok && !ap_env) {
                const int nv = mv_fits ? RBF_NV : RBF_NVA;
                const size_t nW = (size_t)W * W * W * nv * nv * nv;
                ENSURE_C(d_wa, sizeof(double) * nW);
                rbf_walk_table_kernel<1><<<(unsigned)((nW + 255) / 256), 256, 0, st>>>(d_lv.as<RbfLutVals>(), d_wa.p, nv);
                LG.WA = d_wa.as<double>(); LG.wa_nv = nv;
            } else if (want_ap_lut) {
                const size_t nTA = (size_t)W * W * W * RBF_NVA * RBF_NVA * RBF_NVA;
                ENSURE_C(d_luta, sizeof(double) * nTA);
                rbf_lut_build_apply_kernel<<<(unsigned)((nTA + 255) / 256), 256, 0, st>>>(d_lv.as<RbfLutVals>(), d_luta.as<double>());
                LG.TA = d_luta.as<double>();
                if (mv_fits) {   // compact table for the kernel that stages its rows in LDS
                    ENSURE_C(d_luta16, sizeof(double) * nT);
                    rbf_lut_build_apply_kernel<<<(unsigned)((nT + 255) / 256), 256, 0, st>>>(d_lv.as<RbfLutVals>(), d_luta16.as<double>(), RBF_NV);
                    LG.TA16 = d_luta16.as<double>();
                }
            }
        }
    }
    // ---- weights ----
    int its = 0;
    if (is_interp) {   // compute_rbf_weights (:191-202): cg(K, b), IterativeSolvers 0.9.4 defaults
        ENSURE_C(d_r, sizeof(float) * (size_t)n);
        ENSURE_C(d_u, sizeof(float) * (size_t)n);
        ENSURE_C(d_q, sizeof(float) * (size_t)n);
        ENSURE_C(d_part, sizeof(double) * (size_t)std::max(nz * DOT_PARTS, 1024));
        ENSURE_C(d_sum, 64);
        auto dot = [&](const float* a, const float* b, float* out) -> int {
            dot_planes_kernel<<<nz * DOT_PARTS, 256, 0, st>>>(a, b, (int64_t)nx * ny, d_part.as<double>());
            std::vector<double> hp((size_t)nz * DOT_PARTS);
            { const int rc_ = d2h_small(hp.data(), d_part.p, sizeof(double) * hp.size(), st, W.mb); if (rc_) return rc_; }
            double h = 0.0;
            for (double v : hp) h += v;   // parts in (k, part) order (see dot_planes_kernel)
            *out = (float)h;
            return 0;
        };
        auto xr_dot = [&](float alpha, float* out) -> int {   // weights / residual update with dot(r, r)
            cg_update_xr_dot_kernel<<<nz * DOT_PARTS, 256, 0, st>>>(d_w.as<float>(), d_r.as<float>(), d_u.as<float>(), d_q.as<float>(), alpha,
                                                                   (int64_t)nx * ny, d_part.as<double>());
            std::vector<double> hp((size_t)nz * DOT_PARTS);
            { const int rc_ = d2h_small(hp.data(), d_part.p, sizeof(double) * hp.size(), st, W.mb); if (rc_) return rc_; }
            double h = 0.0;
            for (double v : hp) h += v;
            *out = (float)h;
            return 0;
        };
        // materialise K when it fits comfortably (see rbf_kbuild_kernel)
        RbfTaps taps;
        taps.n = 0;
        bool taps_ok = true;
        for (int dk = -G.tap_r; dk <= G.tap_r && taps_ok; ++dk)
            for (int dj = -G.tap_r; dj <= G.tap_r && taps_ok; ++dj)
                for (int di = -G.tap_r; di <= G.tap_r; ++di) {
                    if (dk * dk + dj * dj + di * di > G.tap_d2) continue;
                    if (taps.n == RBF_MAX_TAPS) { taps_ok = false; break; }
                    taps.off[taps.n][0] = (signed char)di; taps.off[taps.n][1] = (signed char)dj; taps.off[taps.n][2] = (signed char)dk;
                    taps.n++;
                }
        // first choice: the table of distinct entries; then the materialised matrix; then on the fly
        const bool use_lut = LG.T != nullptr || LG.WT != nullptr;
        bool use_k = false;
        std::unique_lock<std::mutex> kv_lock(g_rbf_kv_mutex, std::defer_lock);
        if (!use_lut && !(mv_env && mv_env[0] == 'f')) kv_lock.try_lock();   // busy: fall back to on-the-fly
        if (!use_lut && taps_ok && kv_lock.owns_lock()) {
            size_t free_b = 0, total_b = 0;
            const size_t need = sizeof(float) * (size_t)n * (size_t)taps.n;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= total_b / 4 &&
                (need <= g_rbf_kv.cap || need <= free_b / 2) && g_rbf_kv.ensure_exact(need) == 0) {
                rbf_kbuild_kernel<<<nb, 256, 0, st>>>(G, taps, g_rbf_kv.as<float>());
                use_k = true;
            }
        }
        HIP_C(hipMemcpy(d_r.p, d_f.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice));
        HIP_C(hipMemset(d_u.p, 0, sizeof(float) * (size_t)n));
        HIP_C(hipMemset(d_w.p, 0, sizeof(float) * (size_t)n));
        // dot(u, q): the partial sums of the row-walk product's workgroups (formed by that kernel, or by rbf_walk_dot_kernel
        // for the other forms of the product: every form of the CG follows the same numbers)
        const size_t nparts = rbf_walk_nparts(nx, ny, nz, 0, nz);
        ENSURE_C(d_part2, sizeof(double) * std::max(nparts, (size_t)1));
        std::vector<double> hp2(nparts);
        auto sum_parts2 = [&](float* out) -> int {
            { const int rc_ = d2h_small(hp2.data(), d_part2.p, sizeof(double) * nparts, st, W.mb); if (rc_) return rc_; }
            double h = 0.0;
            for (double v : hp2) h += v;   // in (k, walk, row piece) order
            *out = (float)h;
            return 0;
        };
        float rr;
        TRY_C(dot(d_r.as<float>(), d_r.as<float>(), &rr));
        float residual = std::sqrt(rr), prev = 1.0f;
        const float tol = 3.4526698e-4f * residual;   // reltol = sqrt(eps(Float32)), abstol = 0
        const int64_t its_cap = n;
        RbfWalkArgs WAr;
        const bool fused = !mv_env && LG.WT && walk_ok && rbf_walk_planes(LG, 0, -1, 0, -1, &WAr) && !getenv("R2S_RBF_CG_UNFUSED");
        if (fused) {
            // three passes per iteration: [w += alpha_prev u, u = r + beta u] (the weights take the step of the PREVIOUS
            // iteration here, where u is read anyway), the product with the parts of dot(u, q), [r -= alpha q with the parts of
            // dot(r, r)].  Same operations on the same numbers as the loop below (five passes).
            float* ucur = d_u.as<float>();
            float alpha_prev = 0.0f;
            auto r_dot = [&](float alpha, float* out) -> int {
                cg_update_r_dot_kernel<<<nz * DOT_PARTS, 256, 0, st>>>(d_r.as<float>(), d_q.as<float>(), alpha, (int64_t)nx * ny, d_part.as<double>());
                std::vector<double> hp((size_t)nz * DOT_PARTS);
                { const int rc_ = d2h_small(hp.data(), d_part.p, sizeof(double) * hp.size(), st, W.mb); if (rc_) return rc_; }
                double h = 0.0;
                for (double v : hp) h += v;
                *out = (float)h;
                return 0;
            };
            while (!(residual <= tol) && its < its_cap) {
                const float beta = (residual * residual) / (prev * prev);
                cg_update_wu_kernel<<<(unsigned)((n / 4 + 256) / 256), 256, 0, st>>>(d_w.as<float>(), ucur, d_r.as<float>(), alpha_prev, beta, n);
                WAr.T = LG.WT; WAr.x = ucur; WAr.y = d_q.as<float>(); WAr.dot_partial = d_part2.as<double>();
                rbf_walk_launch(0, RBF_NV, WAr, st);
                float uq;
                TRY_C(sum_parts2(&uq));
                const float alpha = (residual * residual) / uq;
                prev = residual;
                TRY_C(r_dot(alpha, &rr));
                residual = std::sqrt(rr);
                its++;
                alpha_prev = alpha;
            }
            if (its > 0) cg_axpy_kernel<<<nb, 256, 0, st>>>(d_w.as<float>(), ucur, alpha_prev, n);   // the last iteration's step
        } else
        while (!(residual <= tol) && its < its_cap) {
            const float beta = (residual * residual) / (prev * prev);
            cg_update_u_kernel<<<nb, 256, 0, st>>>(d_u.as<float>(), d_r.as<float>(), beta, n);
            double* dp = nullptr;   // (the row-walk kernel leaves the parts of dot(u, q) itself)
            if (use_lut && LG.WT && !mv_env) dp = d_part2.as<double>();
            if (use_lut) launch_rbf_matvec_lut(LG, nb, st, d_u.as<float>(), d_q.as<float>(), 0, -1, 0, -1, dp);
            else if (use_k) rbf_matvec_k_kernel<<<nb, 256, 0, st>>>(G, taps, g_rbf_kv.as<float>(), d_u.as<float>(), d_q.as<float>());
            else rbf_matvec_kernel<<<nb, 256, 0, st>>>(G, d_u.as<float>(), d_q.as<float>());
            if (!dp) rbf_walk_dot_launch(d_u.as<float>(), d_q.as<float>(), nx, ny, nz, 0, nz, d_part2.as<double>(), st);
            float uq;
            TRY_C(sum_parts2(&uq));
            const float alpha = (residual * residual) / uq;
            prev = residual;
            TRY_C(xr_dot(alpha, &rr));
            residual = std::sqrt(rr);
            its++;
        }
    } else {
        HIP_C(hipMemcpy(d_w.p, d_f.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice));   // :353
    }
    if (cg_iters) *cg_iters = its;
    // the output field (:363-366) + `add`
    // (smooth = 1: one target per lattice point - through the table of ITS coordinate differences)
    // In RBF_FINE_CHUNKS Z chunks when the caller wants to forward finished chunks (fine_chunk), else in one launch.
    auto eval_fine = [&](float add) -> int {
        static const int chunks_env = getenv("R2S_FINE_CHUNKS") ? std::min(std::max(atoi(getenv("R2S_FINE_CHUNKS")), 1), 64) : 0;   // tuning knob
        const int want = chunks_env ? chunks_env : RBF_FINE_CHUNKS;
        const int nchunk = (fine_chunk && fz >= 4 * want) ? want : 1;
        for (int c = 0; c < nchunk; ++c) {
            const int f0 = (int)((int64_t)fz * c / nchunk), f1 = (int)((int64_t)fz * (c + 1) / nchunk);
            const int64_t t0 = (int64_t)f0 * fx * fy, t1 = (int64_t)f1 * fx * fy;
            const unsigned nbc = (unsigned)((t1 - t0 + 255) / 256);
            if (FL.T) {
                const int64_t waves = (int64_t)(f1 - f0) * fy * smooth * ((FL.nmx + 63) / 64);
                rbf_apply_fine_lut_kernel<<<(unsigned)((waves + 3) / 4), 256, (size_t)FL.nrec * 4 * 16, st>>>(FL, d_w.as<float>(), add, dfine, f0, f1);
            } else if (!(fine_one_to_one && launch_rbf_apply_lut(G, LGF, sts[1], nbc, st, d_w.as<float>(), d_tx.as<float>(), d_ty.as<float>(),
                                                                 d_tz.as<float>(), d_st.as<Stencil>() + 1, add, dfine, t0, t1)))
                rbf_apply_kernel<<<nbc, 256, 0, st>>>(G, d_w.as<float>(), smooth, fx, fy, fz, d_tx.as<float>(), d_ty.as<float>(),
                                                     d_tz.as<float>(), d_st.as<Stencil>() + 1, add, dfine, t0, t1);
            if (fine_chunk) {
                const int rcc = (*fine_chunk)(t0, t1);
                if (rcc) return rcc;
            }
        }
        return 0;
    };
    // fine_early: the field WITHOUT the level shift first (it needs the weights only), so that its chunks travel to the host
    // while the level is found; the caller adds the shift to what it received (x + 0 + th = x + th: the same Float32 sum).
    // The device array stays WITHOUT it (the chunks may still be on their way when the level is known)
    if (fine_early) TRY_C(eval_fine(0.0f));
    // ---- LSF on the coarse grid (:357) and the volume-preserving level (:359, :265-300) ----
    if (!launch_rbf_apply_lut(G, LG, sts[0], nb, st, d_w.as<float>(), d_cx.as<float>(), d_cy.as<float>(), d_cz.as<float>(),
                              d_st.as<Stencil>(), 0.0f, d_lsf.as<float>()))
        rbf_apply_kernel<<<nb, 256, 0, st>>>(G, d_w.as<float>(), 1, nx, ny, nz, d_cx.as<float>(), d_cy.as<float>(),
                                            d_cz.as<float>(), d_st.as<Stencil>(), 0.0f, d_lsf.as<float>());
    if (lsf_out) HIP_C(hipMemcpy(lsf_out, d_lsf.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    TRY_C(vw.init(9));
    TRY_C(vw.prepare(d_lsf.as<float>(), nx, ny, nz, st));
    // the range of the field = the start of the bisection: from the extrema of the 64-cell segments that `prepare` has just
    // formed (every lattice point is a corner of some cell: the same two numbers as a sweep over the field, which cost
    // 0.28 ms at 512^3); a lattice without cells: the sweep
    int mmh[4] = {0x7FFFFFFF, (int)0x80000000, 0x7FFFFFFF, (int)0x80000000};
    HIP_C(hipMemcpyAsync(d_cnt.p, mmh, 16, hipMemcpyHostToDevice, st));
    const int64_t nsegs = vw.seg_field ? (int64_t)(ny - 1) * (nz - 1) * ((nx - 1 + 63) / 64) : 0;
    if (nsegs > 0) {
        const unsigned nbs = (unsigned)std::min<int64_t>((nsegs + 255) / 256, 128);   // (two same-address atomics per wavefront at the end: few wavefronts)
        minmax_kernel<<<nbs, 256, 0, st>>>(vw.segmn.as<float>(), nsegs, d_cnt.as<int>());
        minmax_kernel<<<nbs, 256, 0, st>>>(vw.segmx.as<float>(), nsegs, d_cnt.as<int>() + 2);
    } else {
        minmax_kernel<<<(nb < 2048u ? nb : 2048u), 256, 0, st>>>(d_lsf.as<float>(), n, d_cnt.as<int>());
    }
    TRY_C(d2h_small(mmh, d_cnt.p, 16, st, W.mb));
    auto dec = [](int b) { b = b >= 0 ? b : (b ^ 0x7FFFFFFF); float f; memcpy(&f, &b, 4); return f; };
    float lo = dec(mmh[0]), hi = dec(nsegs > 0 ? mmh[3] : mmh[1]);
    const float edge = std::sqrt((cx[1] - cx[0]) * (cx[1] - cx[0]));   // norm(fine_grid[2,1,1] - fine_grid[1,1,1])
    double eps = 1.0;
    float th = 0.0f;
    int it = 0;
    while (it < 40 && eps > 1.0e-4) {
        th = (lo + hi) / 2;
        float vol;
        if (it > 0) TRY_C(vw.narrow(nx, ny, nz, edge, lo, hi, st));   // (level 0: [lo, hi] = the range of the field, every row is live)
        TRY_C(vw.run(d_lsf.as<float>(), nx, ny, nz, edge, th, 0.0f, st, &vol));
        eps = std::fabs(target_volume - (double)vol);
        if ((double)vol > target_volume) lo = th; else hi = th;
        it++;
    }
    th = -th;
    if (th_out) *th_out = th;
    // ---- fine grid (:363-366) ----
    if (!fine_early) TRY_C(eval_fine(th));   // (fine_early: done above, the shift is the caller's)
    HIP_C(hipGetLastError());
    if (out_dev) HIP_C(hipDeviceSynchronize());
    else HIP_C(hipMemcpy(fine_out, d_fine.p, sizeof(float) * (size_t)nf, hipMemcpyDeviceToHost));
    cleanup();
    return 0;
}

// ====================================================================================
// Z-slab distributed post-processing (single process, one slab per device)
// ====================================================================================
__global__ void gather_u32_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
__global__ void scatter_u32_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ val, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = val[i];
}

namespace {
#define SLAB_HIP(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) { rc = fail(R2S_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)
#define SLAB_TRY(expr)                                                                                       \
    do {                                                                                                     \
        rc = (expr);                                                                                         \
        if (rc) goto done;                                                                                   \
    } while (0)

struct SlabBufs {   // per-slab device buffers, allocated / freed on their device
    std::vector<DevBuf> b;
    const std::vector<r2s_int::Slab>* S;
    explicit SlabBufs(const std::vector<r2s_int::Slab>& s) : b(s.size()), S(&s) {}
    int ensure(size_t q, size_t bytes)
    {
        if (hipSetDevice((*S)[q].device) != hipSuccess) return fail(R2S_ERR_HIP, "hipSetDevice failed");
        if (b[q].ensure_exact(std::max<size_t>(bytes, 16))) return fail(R2S_ERR_NOMEM, "hipMalloc of %zu bytes failed on device %d", bytes, (*S)[q].device);
        return 0;
    }
    template <class T>
    T* at(size_t q) { return b[q].as<T>(); }
    ~SlabBufs()
    {
        for (size_t q = 0; q < b.size(); ++q) {
            (void)hipSetDevice((*S)[q].device);
            b[q].release();
        }
    }
};

int sync_slabs(const std::vector<r2s_int::Slab>& S)
{
    for (const auto& sl : S) {
        HIP_TRY(hipSetDevice(sl.device));
        HIP_TRY(hipStreamSynchronize(sl.stream));
        HIP_TRY(hipGetLastError());
    }
    return 0;
}
}  // namespace

namespace r2s_int {

int exchange_halo_slabs(const std::vector<Slab>& S, const std::vector<void*>& base, size_t elem, int64_t plane, int radius)
{
    const size_t pb = (size_t)plane * elem;
    for (size_t q = 0; q < S.size(); ++q) {
        const Slab& d = S[q];
        if (d.k1 <= d.k0) continue;
        HIP_TRY(hipSetDevice(d.device));
        for (int k = std::max(d.h0, d.k0 - radius); k < std::min(d.h1, d.k1 + radius); ++k) {
            if (k >= d.k0 && k < d.k1) continue;
            for (size_t o = 0; o < S.size(); ++o) {
                const Slab& src = S[o];
                if (k < src.k0 || k >= src.k1) continue;
                HIP_TRY(hipMemcpyPeerAsync((char*)base[q] + (size_t)(k - d.h0) * pb, d.device,
                                           (const char*)base[o] + (size_t)(k - src.h0) * pb, src.device, pb, d.stream));
            }
        }
    }
    return sync_slabs(S);
}

int remove_artifacts_slabs(const std::vector<Slab>& S, const r2s_grid* g, double threshold, double min_ratio, int64_t* n_flipped)
{
    const int nx = (int)g->N[0] + 1, ny = (int)g->N[1] + 1;
    const int64_t plane = (int64_t)nx * ny;
    const size_t G = S.size();
    int rc = 0;
    SlabBufs L(S), root(S), size(S), cnt(S), idx(S), val(S);
    std::vector<uint32_t> nvox(G, 0);
    std::vector<std::vector<uint32_t>> broots(G);                    // roots of slab q that touch an interface
    std::vector<std::vector<uint32_t>> top(G), bot(G);               // root labels of the last / first owned plane
    struct Cls { uint64_t size = 0, min_gid = ~0ull; };
    std::vector<std::pair<uint64_t, uint64_t>> parent;               // (key, parent key), sorted by key after collection
    auto key_of = [](size_t q, uint32_t r) { return ((uint64_t)q << 32) | r; };
    std::vector<uint64_t> keys;
    std::vector<uint64_t> par;
    auto find = [&](uint64_t k) -> size_t {
        size_t i = (size_t)(std::lower_bound(keys.begin(), keys.end(), k) - keys.begin());
        while (par[i] != i) { par[i] = par[par[i]]; i = (size_t)par[i]; }
        return i;
    };
    uint64_t interior = 0, largest = 0, largest_gid = ~0ull;
    size_t largest_cls = (size_t)-1;
    bool largest_merged = false;
    std::vector<uint32_t> largest_local(G, NOLABEL - 1u);
    uint64_t min_size64 = 1;
    int64_t flipped = 0;
    std::vector<size_t> order;   // non-empty slabs in k order
    for (size_t q = 0; q < G; ++q)
        if (S[q].k1 > S[q].k0) order.push_back(q);
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return S[a].k0 < S[b].k0; });

    // ---- per slab: labels, roots, sizes ----
    for (size_t q : order) {
        const Slab& d = S[q];
        const int64_t n64 = (int64_t)(d.k1 - d.k0) * plane;
        if (n64 >= 0xFFFFFFFFll) return fail(R2S_ERR_ARG, "slab too large for 32-bit labels");
        nvox[q] = (uint32_t)n64;
        SLAB_TRY(L.ensure(q, 4 * (size_t)n64)); SLAB_TRY(root.ensure(q, 4 * (size_t)n64)); SLAB_TRY(size.ensure(q, 4 * (size_t)n64));
        SLAB_TRY(cnt.ensure(q, 64));
        const unsigned nb = (unsigned)((n64 + 255) / 256);
        double* sdf = d.d_sdf + (int64_t)(d.k0 - d.h0) * plane;
        uint32_t h[4] = {0, NOLABEL, 0, 0};
        SLAB_HIP(hipMemcpyAsync(cnt.at<uint32_t>(q), h, sizeof h, hipMemcpyHostToDevice, d.stream));
        SLAB_HIP(hipMemsetAsync(size.at<uint32_t>(q), 0, 4 * (size_t)n64, d.stream));
        ccl_init_kernel<<<nb, 256, 0, d.stream>>>(sdf, nvox[q], nx, threshold, L.at<uint32_t>(q));
        ccl_union_kernel<<<nb, 256, 0, d.stream>>>(L.at<uint32_t>(q), nx, ny, d.k1 - d.k0);
        ccl_compress_heads_kernel<<<nb, 256, 0, d.stream>>>(L.at<uint32_t>(q), nvox[q], nx);
        ccl_flatten_count_kernel<<<(nb < 4096u ? nb : 4096u), 256, 0, d.stream>>>(L.at<uint32_t>(q), nvox[q], root.at<uint32_t>(q), size.at<uint32_t>(q));
        ccl_max_kernel<<<nb, 256, 0, d.stream>>>(size.at<uint32_t>(q), nvox[q], cnt.at<uint32_t>(q));   // [3]: interior voxels
    }
    SLAB_TRY(sync_slabs(S));
    for (size_t q : order) {
        const Slab& d = S[q];
        uint32_t h[4];
        SLAB_HIP(hipSetDevice(d.device));
        SLAB_HIP(hipMemcpy(h, cnt.at<uint32_t>(q), sizeof h, hipMemcpyDeviceToHost));
        interior += h[3];
        top[q].resize((size_t)plane); bot[q].resize((size_t)plane);
        SLAB_HIP(hipMemcpy(bot[q].data(), root.at<uint32_t>(q), 4 * (size_t)plane, hipMemcpyDeviceToHost));
        SLAB_HIP(hipMemcpy(top[q].data(), root.at<uint32_t>(q) + (size_t)(nvox[q] - plane), 4 * (size_t)plane, hipMemcpyDeviceToHost));
    }
    if (interior == 0) {   // SdfArtifactRemoval.jl:150-153
        if (n_flipped) *n_flipped = 0;
        return 0;
    }
    // ---- interface merge on the host: union-find over the roots that touch an interface ----
    for (size_t a = 0; a + 1 < order.size(); ++a) {
        const size_t q = order[a], p = order[a + 1];
        for (int64_t c = 0; c < plane; ++c) {
            if (top[q][(size_t)c] != NOLABEL) keys.push_back(key_of(q, top[q][(size_t)c]));
            if (bot[p][(size_t)c] != NOLABEL) keys.push_back(key_of(p, bot[p][(size_t)c]));
        }
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    par.resize(keys.size());
    for (size_t i = 0; i < par.size(); ++i) par[i] = i;
    for (size_t a = 0; a + 1 < order.size(); ++a) {
        const size_t q = order[a], p = order[a + 1];
        for (int64_t c = 0; c < plane; ++c) {
            const uint32_t ra = top[q][(size_t)c], rb = bot[p][(size_t)c];
            if (ra == NOLABEL || rb == NOLABEL) continue;
            const size_t ia = find(key_of(q, ra)), ib = find(key_of(p, rb));
            if (ia != ib) par[std::max(ia, ib)] = std::min(ia, ib);
        }
    }
    {
        // sizes of the interface roots from their devices, class sizes, and back
        std::vector<Cls> cls(keys.size());
        std::vector<std::vector<uint32_t>> bsz(G);
        for (size_t i = 0; i < keys.size(); ++i) broots[(size_t)(keys[i] >> 32)].push_back((uint32_t)keys[i]);
        for (size_t q : order) {
            const uint32_t nbq = (uint32_t)broots[q].size();
            bsz[q].resize(nbq);
            if (!nbq) continue;
            const Slab& d = S[q];
            SLAB_TRY(idx.ensure(q, 4 * (size_t)nbq)); SLAB_TRY(val.ensure(q, 4 * (size_t)nbq));
            SLAB_HIP(hipMemcpy(idx.at<uint32_t>(q), broots[q].data(), 4 * (size_t)nbq, hipMemcpyHostToDevice));
            gather_u32_kernel<<<(nbq + 255) / 256, 256, 0, d.stream>>>(size.at<uint32_t>(q), idx.at<uint32_t>(q), nbq, val.at<uint32_t>(q));
            SLAB_HIP(hipMemcpyAsync(bsz[q].data(), val.at<uint32_t>(q), 4 * (size_t)nbq, hipMemcpyDeviceToHost, d.stream));
        }
        SLAB_TRY(sync_slabs(S));
        {
            std::vector<size_t> pos(G, 0);
            for (size_t i = 0; i < keys.size(); ++i) {
                const size_t q = (size_t)(keys[i] >> 32);
                const size_t c = find(keys[i]);
                cls[c].size += bsz[q][pos[q]++];
                const uint64_t gid = (uint64_t)S[q].k0 * (uint64_t)plane + (uint32_t)keys[i];   // the root's index in the whole grid
                cls[c].min_gid = std::min(cls[c].min_gid, gid);
            }
        }
        // every interface root now carries its class size (saturated): the keep test of a slab sees the whole component
        for (size_t q : order) {
            const uint32_t nbq = (uint32_t)broots[q].size();
            if (!nbq) continue;
            const Slab& d = S[q];
            std::vector<uint32_t> v(nbq);
            for (uint32_t i = 0; i < nbq; ++i) {
                const uint64_t sz = cls[find(key_of(q, broots[q][i]))].size;
                v[i] = sz > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)sz;
            }
            SLAB_HIP(hipSetDevice(d.device));
            SLAB_HIP(hipMemcpy(val.at<uint32_t>(q), v.data(), 4 * (size_t)nbq, hipMemcpyHostToDevice));
            scatter_u32_kernel<<<(nbq + 255) / 256, 256, 0, d.stream>>>(size.at<uint32_t>(q), idx.at<uint32_t>(q), val.at<uint32_t>(q), nbq);
        }
        // ---- the largest component: per slab (max size, smallest root having it), then the smallest grid index among ties ----
        for (size_t q : order) {
            const Slab& d = S[q];
            const unsigned nb = (unsigned)((nvox[q] + 255u) / 256u);
            uint32_t h[4] = {0, NOLABEL, 0, 0};
            SLAB_HIP(hipSetDevice(d.device));
            SLAB_HIP(hipMemcpyAsync(cnt.at<uint32_t>(q), h, sizeof h, hipMemcpyHostToDevice, d.stream));
            ccl_max_kernel<<<nb, 256, 0, d.stream>>>(size.at<uint32_t>(q), nvox[q], cnt.at<uint32_t>(q));
            ccl_argmax_kernel<<<nb, 256, 0, d.stream>>>(size.at<uint32_t>(q), nvox[q], cnt.at<uint32_t>(q));
        }
        SLAB_TRY(sync_slabs(S));
        for (size_t q : order) {
            uint32_t h[4];
            SLAB_HIP(hipSetDevice(S[q].device));
            SLAB_HIP(hipMemcpy(h, cnt.at<uint32_t>(q), sizeof h, hipMemcpyDeviceToHost));
            if (h[1] == NOLABEL) continue;
            uint64_t sz = h[0], gid = (uint64_t)S[q].k0 * (uint64_t)plane + h[1];
            size_t c = (size_t)-1;
            const uint64_t k = key_of(q, h[1]);
            const auto it = std::lower_bound(keys.begin(), keys.end(), k);
            if (it != keys.end() && *it == k) {   // an interface root: its class decides (true size, smallest member index)
                c = find(k);
                sz = cls[c].size;
                gid = cls[c].min_gid;
            }
            if (sz > largest || (sz == largest && gid < largest_gid)) {
                largest = sz; largest_gid = gid; largest_cls = c; largest_merged = (c != (size_t)-1);
                std::fill(largest_local.begin(), largest_local.end(), NOLABEL - 1u);
                if (!largest_merged) largest_local[q] = h[1];
            }
        }
        // a saturated size hides the true maximum from the slabs: an interface class larger than any saturated value wins here
        for (size_t i = 0; i < keys.size(); ++i)
            if (par[i] == i && (cls[i].size > largest || (cls[i].size == largest && cls[i].min_gid < largest_gid))) {
                largest = cls[i].size; largest_gid = cls[i].min_gid; largest_cls = i; largest_merged = true;
                std::fill(largest_local.begin(), largest_local.end(), NOLABEL - 1u);
            }
        {
            // min_component_size = max(1, round(Int, ratio * largest)), ties to even (SdfArtifactRemoval.jl:206)
            const double ms = std::nearbyint(min_ratio * (double)largest);
            min_size64 = ms < 1.0 ? 1 : (ms > 1.8e19 ? ~0ull : (uint64_t)ms);
        }
        if (largest_merged) {
            // its members are kept whatever the ratio (root == largest, :220): mark them with a size no threshold exceeds
            for (size_t q : order) {
                std::vector<uint32_t> ids;
                for (uint32_t r : broots[q])
                    if (find(key_of(q, r)) == largest_cls) ids.push_back(r);
                if (ids.empty()) continue;
                const Slab& d = S[q];
                std::vector<uint32_t> v(ids.size(), 0xFFFFFFFFu);
                SLAB_HIP(hipSetDevice(d.device));
                SLAB_HIP(hipMemcpy(idx.at<uint32_t>(q), ids.data(), 4 * ids.size(), hipMemcpyHostToDevice));
                SLAB_HIP(hipMemcpy(val.at<uint32_t>(q), v.data(), 4 * v.size(), hipMemcpyHostToDevice));
                scatter_u32_kernel<<<(unsigned)((ids.size() + 255) / 256), 256, 0, d.stream>>>(size.at<uint32_t>(q), idx.at<uint32_t>(q),
                                                                                             val.at<uint32_t>(q), (uint32_t)ids.size());
            }
        }
    }
    // ---- flip ----
    for (size_t q : order) {
        const Slab& d = S[q];
        const unsigned nb = (unsigned)((nvox[q] + 255u) / 256u);
        const uint32_t min_size = min_size64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)min_size64;
        uint32_t h[4] = {0, NOLABEL, 0, 0};
        SLAB_HIP(hipSetDevice(d.device));
        SLAB_HIP(hipMemcpyAsync(cnt.at<uint32_t>(q), h, sizeof h, hipMemcpyHostToDevice, d.stream));
        ccl_flip_kernel<<<nb, 256, 0, d.stream>>>(d.d_sdf + (int64_t)(d.k0 - d.h0) * plane, root.at<uint32_t>(q), size.at<uint32_t>(q), nvox[q],
                                                largest_local[q], min_size, cnt.at<uint32_t>(q));
    }
    SLAB_TRY(sync_slabs(S));
    for (size_t q : order) {
        uint32_t h[4];
        SLAB_HIP(hipSetDevice(S[q].device));
        SLAB_HIP(hipMemcpy(h, cnt.at<uint32_t>(q), sizeof h, hipMemcpyDeviceToHost));
        flipped += h[2];
    }
    if (n_flipped) *n_flipped = flipped;
done:
    return rc;
}

}  // namespace r2s_int

namespace r2s_int {
// RBFs_smoothing over Z-slabs (see r2s_internal.hpp).  Every step is the single-device step restricted to the
// slab's planes; the vectors are addressed as the whole grid's (base pointer shifted by the first held plane).
int rbf_smooth_slabs(const std::vector<Slab>& S, const r2s_grid* g, int is_interp, int smooth, double kthr, double target_volume,
                     float* fine_out_host, float* th_out, int* cg_iters)
{
    if (!g || !fine_out_host) return fail(R2S_ERR_ARG, "null argument");
    if (smooth < 1 || smooth > 4) return fail(R2S_ERR_ARG, "smooth must be 1..4");
    if (!(kthr > 0.0 && kthr < 1.0)) return fail(R2S_ERR_ARG, "kernel threshold must be in (0,1)");
    const int nx = (int)g->N[0] + 1, ny = (int)g->N[1] + 1, nz = (int)g->N[2] + 1;
    const int64_t plane = (int64_t)nx * ny;
    const int fx = (int)g->N[0] * smooth + 1, fy = (int)g->N[1] * smooth + 1, fz = (int)g->N[2] * smooth + 1;
    const int64_t fplane = (int64_t)fx * fy;
    const size_t G = S.size();
    int rc = 0;
    std::vector<size_t> order;
    for (size_t q = 0; q < G; ++q)
        if (S[q].k1 > S[q].k0) order.push_back(q);
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return S[a].k0 < S[b].k0; });
    if (order.empty()) return fail(R2S_ERR_ARG, "no slab owns a plane");
    // ---- host-side geometry (as rbf_smooth_host) ----
    std::vector<float> cx, cy, cz, tx(fx), ty(fy), tz(fz);
    coarse_coords(g->aabb_min[0], g->aabb_max[0], nx, cx);
    coarse_coords(g->aabb_min[1], g->aabb_max[1], ny, cy);
    coarse_coords(g->aabb_min[2], g->aabb_max[2], nz, cz);
    {
        const float xmin = (float)g->aabb_min[0], xmax = (float)g->aabb_max[0], ymin = (float)g->aabb_min[1], zmin = (float)g->aabb_min[2];
        const float dx = (xmax - xmin) / (float)(fx - 1);
        for (int i = 0; i < fx; ++i) tx[i] = xmin + (float)i * dx;
        for (int i = 0; i < fy; ++i) ty[i] = ymin + (float)i * dx;
        for (int i = 0; i < fz; ++i) tz[i] = zmin + (float)i * dx;
    }
    std::vector<Stencil> sts(1 + (size_t)smooth * smooth * smooth);
    {
        int fr0[3] = {0, 0, 0};
        build_stencil(1, fr0, &sts[0], -std::log(kthr));
        for (int a = 0; a < smooth; ++a)
            for (int b = 0; b < smooth; ++b)
                for (int c = 0; c < smooth; ++c) {
                    int fr[3] = {c, b, a};
                    build_stencil(smooth, fr, &sts[1 + (a * smooth + b) * smooth + c], -std::log(kthr));
                }
    }
    RbfGeom G0;
    memset(&G0, 0, sizeof G0);
    G0.nx = nx; G0.ny = ny; G0.nz = nz;
    G0.sigma = g->cell_size;
    G0.thr = kthr;
    G0.max_distance = (float)std::sqrt(-std::log(kthr) * G0.sigma * G0.sigma);
    {
        const double R2 = -std::log(kthr);
        G0.tap_d2 = (int)std::floor(R2 * 1.05 + 0.25);
        G0.tap_r = (int)std::floor(std::sqrt((double)G0.tap_d2));
    }
    int halo = G0.tap_r;   // planes a stencil reaches beyond the slab
    for (const Stencil& st : sts)
        for (int q = 0; q < st.n; ++q) halo = std::max(halo, std::max((int)st.off[q][2], -(int)st.off[q][2]));
    for (size_t q : order)
        if (S[q].h0 > std::max(0, S[q].k0 - halo) || S[q].h1 < std::min(nz, S[q].k1 + halo))
            return fail(R2S_ERR_ARG, "slab [%d,%d) holds [%d,%d): the smoothing stencils need a halo of %d planes", S[q].k0, S[q].k1,
                        S[q].h0, S[q].h1, halo);
    // matvec and same-grid evaluation through the tables of distinct kernel values when the lattice allows it (else on the fly)
    RbfLutVals LV;
    memset(&LV, 0, sizeof LV);
    std::vector<uint8_t> ix, iy, iz;
    bool lut_axes = false, mv_fits = false;
    const char* ap_env = getenv("R2S_RBF_APPLY");
    const bool want_ap_lut = !(ap_env && ap_env[0] == 'f');
    if ((is_interp || want_ap_lut) && G0.tap_r >= 1 && G0.tap_r <= 3) {
        LV.R = G0.tap_r; LV.sigma = G0.sigma; LV.thr = G0.thr; LV.max_distance = G0.max_distance;
        const int m0 = rbf_lut_axis(cx, cx, G0.tap_r, LV.v[0], ix), m1 = rbf_lut_axis(cy, cy, G0.tap_r, LV.v[1], iy),
                  m2 = rbf_lut_axis(cz, cz, G0.tap_r, LV.v[2], iz);
        lut_axes = m0 && m1 && m2;
        mv_fits = std::max(m0, std::max(m1, m2)) <= RBF_NV - 1;
    }
    const bool use_lut = is_interp && lut_axes && mv_fits;
    const bool fine_one_to_one = smooth == 1 && fx == nx && fy == ny && fz == nz;
    RbfLutVals LVF;
    memset(&LVF, 0, sizeof LVF);
    std::vector<uint8_t> fix, fiy, fiz;
    bool lutf_axes = false;
    int lutf_most = 0;
    if (want_ap_lut && fine_one_to_one && G0.tap_r >= 1 && G0.tap_r <= 3) {
        LVF.R = G0.tap_r; LVF.sigma = G0.sigma; LVF.thr = G0.thr; LVF.max_distance = G0.max_distance;
        const int f0 = rbf_lut_axis(tx, cx, G0.tap_r, LVF.v[0], fix), f1 = rbf_lut_axis(ty, cy, G0.tap_r, LVF.v[1], fiy),
                  f2 = rbf_lut_axis(tz, cz, G0.tap_r, LVF.v[2], fiz);
        lutf_axes = f0 && f1 && f2;
        lutf_most = std::max(f0, std::max(f1, f2));
    }
    const bool walk_ok = rbf_walk_supported(G0.tap_r, G0.tap_d2, nx, ny);
    const char* mv_env = getenv("R2S_RBF_MATVEC");
    SlabBufs blutf(S), bfvx(S), bfvy(S), bfvz(S), blv(S), blvf(S), bsegmn(S), bsegmx(S), bluta16(S), bwt(S), bwa(S), bwaf(S), bpart2(S);
    SlabBufs bf(S), bw(S), br(S), bu(S), bq(S), blsf(S), bfine(S), bcx(S), bcy(S), bcz(S), btx(S), bty(S), btz(S), bst(S), bcnt(S),
        bpart(S), blut(S), bluta(S), bvx(S), bvy(S), bvz(S), brows(S);
    std::vector<RbfGeom> Gq(G, G0);
    std::vector<RbfLutGeom> LG(G), LGF(G);
    std::vector<VolumeWork> vw(G);
    std::vector<void*> base(G, nullptr);
    auto nheld = [&](size_t q) { return (size_t)(S[q].h1 - S[q].h0) * (size_t)plane; };
    auto vptr = [&](float* p, size_t q) { return p - (int64_t)S[q].h0 * plane; };   // whole-grid addressing of a held array
    auto owned = [&](float* p, size_t q) { return p + (int64_t)(S[q].k0 - S[q].h0) * plane; };
    auto nowned = [&](size_t q) { return (int64_t)(S[q].k1 - S[q].k0) * plane; };
    auto halo_xchg = [&](SlabBufs& B, int radius) -> int {
        for (size_t q = 0; q < G; ++q) base[q] = B.b[q].p;
        return exchange_halo_slabs(S, base, sizeof(float), plane, radius);
    };
    // plane-wise dot product over all slabs, summed in k order on the host (dot_planes_kernel)
    std::vector<std::vector<double>> hp(G), hp2(G);
    auto dot = [&](SlabBufs& A, SlabBufs& B, float* out) -> int {
        for (size_t q : order) {
            const Slab& d = S[q];
            HIP_TRY(hipSetDevice(d.device));
            hp[q].resize((size_t)(d.k1 - d.k0) * DOT_PARTS);
            dot_planes_kernel<<<(d.k1 - d.k0) * DOT_PARTS, 256, 0, d.stream>>>(owned(A.at<float>(q), q), owned(B.at<float>(q), q), plane, bpart.at<double>(q));
            HIP_TRY(hipMemcpyAsync(hp[q].data(), bpart.at<double>(q), sizeof(double) * hp[q].size(), hipMemcpyDeviceToHost, d.stream));
        }
        int r2 = sync_slabs(S);
        if (r2) return r2;
        double h = 0.0;
        for (size_t q : order)
            for (double v : hp[q]) h += v;
        *out = (float)h;
        return 0;
    };
    int its = 0;
    float th = 0.0f;
    uint32_t gmax_bits = 0;
    bool any_real = false;
    // ---- per-slab set-up: buffers, coordinates, stencils, tables, process_vector pass 1 ----
    for (size_t q : order) {
        const Slab& d = S[q];
        const size_t nh = nheld(q);
        SLAB_TRY(bf.ensure(q, 4 * nh)); SLAB_TRY(bw.ensure(q, 4 * nh)); SLAB_TRY(blsf.ensure(q, 4 * nh));
        SLAB_TRY(bcnt.ensure(q, 64)); SLAB_TRY(bpart.ensure(q, sizeof(double) * (size_t)std::max(nz * DOT_PARTS, 1024)));
        if (is_interp) { SLAB_TRY(br.ensure(q, 4 * nh)); SLAB_TRY(bu.ensure(q, 4 * nh)); SLAB_TRY(bq.ensure(q, 4 * nh)); }
        auto up = [&](SlabBufs& B, const void* src, size_t bytes) -> int {
            int r2 = B.ensure(q, bytes);
            if (r2) return r2;
            HIP_TRY(hipMemcpyAsync(B.b[q].p, src, bytes, hipMemcpyHostToDevice, d.stream));
            return 0;
        };
        SLAB_TRY(up(bcx, cx.data(), 4 * cx.size())); SLAB_TRY(up(bcy, cy.data(), 4 * cy.size())); SLAB_TRY(up(bcz, cz.data(), 4 * cz.size()));
        SLAB_TRY(up(btx, tx.data(), 4 * tx.size())); SLAB_TRY(up(bty, ty.data(), 4 * ty.size())); SLAB_TRY(up(btz, tz.data(), 4 * tz.size()));
        SLAB_TRY(up(bst, sts.data(), sizeof(Stencil) * sts.size()));
        Gq[q].cx = bcx.at<float>(q); Gq[q].cy = bcy.at<float>(q); Gq[q].cz = bcz.at<float>(q);
        memset(&LG[q], 0, sizeof(RbfLutGeom));
        if (lut_axes) {
            const int W = 2 * G0.tap_r + 1;
            const size_t nT = (size_t)W * W * W * RBF_NV * RBF_NV * RBF_NV;
            SLAB_TRY(up(bvx, ix.data(), ix.size())); SLAB_TRY(up(bvy, iy.data(), iy.size())); SLAB_TRY(up(bvz, iz.data(), iz.size()));
            SLAB_TRY(up(blv, &LV, sizeof LV));
            LG[q].nx = nx; LG[q].ny = ny; LG[q].nz = nz; LG[q].R = G0.tap_r; LG[q].tap_d2 = G0.tap_d2;
            LG[q].vx = bvx.at<uint8_t>(q); LG[q].vy = bvy.at<uint8_t>(q); LG[q].vz = bvz.at<uint8_t>(q);
            if (use_lut && walk_ok && !mv_env) {
                SLAB_TRY(bwt.ensure(q, sizeof(float) * nT));
                rbf_walk_table_kernel<0><<<(unsigned)((nT + 255) / 256), 256, 0, d.stream>>>(blv.at<RbfLutVals>(q), bwt.b[q].p, RBF_NV);
                LG[q].WT = bwt.at<float>(q);
            } else if (use_lut) {
                SLAB_TRY(blut.ensure(q, sizeof(float) * nT));
                rbf_lut_build_kernel<<<(unsigned)((nT + 255) / 256), 256, 0, d.stream>>>(blv.at<RbfLutVals>(q), blut.at<float>(q));
                LG[q].T = blut.at<float>(q);
            }
            if (want_ap_lut && walk_ok && !ap_env) {
                const int nv = mv_fits ? RBF_NV : RBF_NVA;
                const size_t nW = (size_t)W * W * W * nv * nv * nv;
                SLAB_TRY(bwa.ensure(q, sizeof(double) * nW));
                rbf_walk_table_kernel<1><<<(unsigned)((nW + 255) / 256), 256, 0, d.stream>>>(blv.at<RbfLutVals>(q), bwa.b[q].p, nv);
                LG[q].WA = bwa.at<double>(q); LG[q].wa_nv = nv;
            } else if (want_ap_lut) {
                const size_t nTA = (size_t)W * W * W * RBF_NVA * RBF_NVA * RBF_NVA;
                SLAB_TRY(bluta.ensure(q, sizeof(double) * nTA));
                rbf_lut_build_apply_kernel<<<(unsigned)((nTA + 255) / 256), 256, 0, d.stream>>>(blv.at<RbfLutVals>(q), bluta.at<double>(q));
                LG[q].TA = bluta.at<double>(q);
                if (mv_fits) {
                    SLAB_TRY(bluta16.ensure(q, sizeof(double) * nT));
                    rbf_lut_build_apply_kernel<<<(unsigned)((nT + 255) / 256), 256, 0, d.stream>>>(blv.at<RbfLutVals>(q), bluta16.at<double>(q), RBF_NV);
                    LG[q].TA16 = bluta16.at<double>(q);
                }
            }
        }
        memset(&LGF[q], 0, sizeof(RbfLutGeom));
        if (lutf_axes) {
            const int W = 2 * G0.tap_r + 1;
            const size_t nT = (size_t)W * W * W * RBF_NVA * RBF_NVA * RBF_NVA;
            SLAB_TRY(up(bfvx, fix.data(), fix.size())); SLAB_TRY(up(bfvy, fiy.data(), fiy.size())); SLAB_TRY(up(bfvz, fiz.data(), fiz.size()));
            LGF[q].nx = nx; LGF[q].ny = ny; LGF[q].nz = nz; LGF[q].R = G0.tap_r; LGF[q].tap_d2 = G0.tap_d2;
            LGF[q].vx = bfvx.at<uint8_t>(q); LGF[q].vy = bfvy.at<uint8_t>(q); LGF[q].vz = bfvz.at<uint8_t>(q);
            SLAB_TRY(up(blvf, &LVF, sizeof LVF));
            if (walk_ok && !ap_env) {
                const int nv = lutf_most <= RBF_NV - 1 ? RBF_NV : RBF_NVA;
                const size_t nW = (size_t)W * W * W * nv * nv * nv;
                SLAB_TRY(bwaf.ensure(q, sizeof(double) * nW));
                rbf_walk_table_kernel<1><<<(unsigned)((nW + 255) / 256), 256, 0, d.stream>>>(blvf.at<RbfLutVals>(q), bwaf.b[q].p, nv);
                LGF[q].WA = bwaf.at<double>(q); LGF[q].wa_nv = nv;
            } else {
                SLAB_TRY(blutf.ensure(q, sizeof(double) * nT));
                rbf_lut_build_apply_kernel<<<(unsigned)((nT + 255) / 256), 256, 0, d.stream>>>(blvf.at<RbfLutVals>(q), blutf.at<double>(q));
                LGF[q].TA = blutf.at<double>(q);
            }
        }
        // process_vector pass 1 on the OWNED planes (every plane counts once for the maximum)
        SLAB_HIP(hipMemsetAsync(bcnt.b[q].p, 0, 64, d.stream));
        const int64_t no = nowned(q);
        const unsigned nbo = (unsigned)((no + 255) / 256);
        pv_max_kernel<<<(nbo < 2048u ? nbo : 2048u), 256, 0, d.stream>>>(d.d_sdf + (int64_t)(d.k0 - d.h0) * plane, no, owned(bf.at<float>(q), q),
                                                                         bcnt.at<uint32_t>(q), bcnt.at<uint32_t>(q) + 1);
    }
    SLAB_TRY(sync_slabs(S));
    for (size_t q : order) {
        uint32_t hc[2];
        SLAB_HIP(hipSetDevice(S[q].device));
        SLAB_HIP(hipMemcpy(hc, bcnt.b[q].p, 8, hipMemcpyDeviceToHost));
        if (hc[1]) { any_real = true; gmax_bits = std::max(gmax_bits, hc[0]); }   // non-negative floats order like their bits
    }
    if (!any_real) { rc = fail(R2S_ERR_ARG, "every SDF value is a sentinel: nothing to smooth"); goto done; }
    for (size_t q : order) {
        const Slab& d = S[q];
        const int64_t no = nowned(q);
        SLAB_HIP(hipSetDevice(d.device));
        SLAB_HIP(hipMemcpyAsync(bcnt.b[q].p, &gmax_bits, 4, hipMemcpyHostToDevice, d.stream));
        pv_replace_kernel<<<(unsigned)((no + 255) / 256), 256, 0, d.stream>>>(owned(bf.at<float>(q), q), no, bcnt.at<uint32_t>(q));
    }
    SLAB_TRY(sync_slabs(S));
    // ---- weights ----
    if (is_interp) {   // compute_rbf_weights (:191-202), the CG of rbf_smooth_host slab by slab
        for (size_t q : order) {
            const Slab& d = S[q];
            SLAB_HIP(hipSetDevice(d.device));
            SLAB_HIP(hipMemcpyAsync(owned(br.at<float>(q), q), owned(bf.at<float>(q), q), 4 * (size_t)nowned(q), hipMemcpyDeviceToDevice, d.stream));
            SLAB_HIP(hipMemsetAsync(bu.b[q].p, 0, 4 * nheld(q), d.stream));
            SLAB_HIP(hipMemsetAsync(bw.b[q].p, 0, 4 * nheld(q), d.stream));
        }
        float rr;
        SLAB_TRY(dot(br, br, &rr));
        float residual = std::sqrt(rr), prev = 1.0f;
        const float tol = 3.4526698e-4f * residual;
        const int64_t ntot = (int64_t)nz * plane;
        while (!(residual <= tol) && its < ntot) {
            const float beta = (residual * residual) / (prev * prev);
            for (size_t q : order) {
                const Slab& d = S[q];
                const int64_t no = nowned(q);
                SLAB_HIP(hipSetDevice(d.device));
                cg_update_u_kernel<<<(unsigned)((no + 255) / 256), 256, 0, d.stream>>>(owned(bu.at<float>(q), q), owned(br.at<float>(q), q), beta, no);
            }
            SLAB_TRY(sync_slabs(S));
            SLAB_TRY(halo_xchg(bu, G0.tap_r));
            for (size_t q : order) {
                const Slab& d = S[q];
                const int64_t t0 = (int64_t)d.k0 * plane, t1 = (int64_t)d.k1 * plane;
                const unsigned nb = (unsigned)((t1 - t0 + 255) / 256);
                SLAB_HIP(hipSetDevice(d.device));
                float* xv = vptr(bu.at<float>(q), q);
                float* yv = vptr(bq.at<float>(q), q);
                const int64_t xlo = (int64_t)d.h0 * plane, xhi = (int64_t)d.h1 * plane - 1;
                // dot(u, q): the partial sums of the row-walk product's workgroups, as on one device (rbf_smooth_host)
                const size_t np = rbf_walk_nparts(nx, ny, nz, d.k0, d.k1);
                SLAB_TRY(bpart2.ensure(q, sizeof(double) * std::max(np, (size_t)1)));
                hp2[q].resize(np);
                const bool walk_mv = use_lut && LG[q].WT && !mv_env;
                if (use_lut) launch_rbf_matvec_lut(LG[q], nb, d.stream, xv, yv, t0, t1, xlo, xhi, walk_mv ? bpart2.at<double>(q) : nullptr);
                else rbf_matvec_kernel<<<nb, 256, 0, d.stream>>>(Gq[q], xv, yv, t0, t1);
                if (!walk_mv) rbf_walk_dot_launch(xv, yv, nx, ny, nz, d.k0, d.k1, bpart2.at<double>(q), d.stream);
                SLAB_HIP(hipMemcpyAsync(hp2[q].data(), bpart2.at<double>(q), sizeof(double) * np, hipMemcpyDeviceToHost, d.stream));
            }
            float uq;
            SLAB_TRY(sync_slabs(S));
            {
                double h = 0.0;
                for (size_t q : order)   // (slabs in k order: the order of the single-device run)
                    for (double v : hp2[q]) h += v;
                uq = (float)h;
            }
            const float alpha = (residual * residual) / uq;
            for (size_t q : order) {   // weights / residual update with the parts of dot(r, r)
                const Slab& d = S[q];
                SLAB_HIP(hipSetDevice(d.device));
                hp[q].resize((size_t)(d.k1 - d.k0) * DOT_PARTS);
                cg_update_xr_dot_kernel<<<(d.k1 - d.k0) * DOT_PARTS, 256, 0, d.stream>>>(owned(bw.at<float>(q), q), owned(br.at<float>(q), q),
                                                                                         owned(bu.at<float>(q), q), owned(bq.at<float>(q), q), alpha,
                                                                                         plane, bpart.at<double>(q));
                SLAB_HIP(hipMemcpyAsync(hp[q].data(), bpart.at<double>(q), sizeof(double) * hp[q].size(), hipMemcpyDeviceToHost, d.stream));
            }
            prev = residual;
            SLAB_TRY(sync_slabs(S));
            {
                double h = 0.0;
                for (size_t q : order)
                    for (double v : hp[q]) h += v;
                rr = (float)h;
            }
            residual = std::sqrt(rr);
            its++;
        }
    } else {
        for (size_t q : order) {
            SLAB_HIP(hipSetDevice(S[q].device));
            SLAB_HIP(hipMemcpyAsync(owned(bw.at<float>(q), q), owned(bf.at<float>(q), q), 4 * (size_t)nowned(q), hipMemcpyDeviceToDevice, S[q].stream));
        }
        SLAB_TRY(sync_slabs(S));
    }
    if (cg_iters) *cg_iters = its;
    // ---- LSF on the coarse grid (:357) ----
    SLAB_TRY(halo_xchg(bw, halo));
    for (size_t q : order) {
        const Slab& d = S[q];
        const int64_t t0 = (int64_t)d.k0 * plane, t1 = (int64_t)d.k1 * plane;
        SLAB_HIP(hipSetDevice(d.device));
        if (!launch_rbf_apply_lut(Gq[q], LG[q], sts[0], (unsigned)((t1 - t0 + 255) / 256), d.stream, vptr(bw.at<float>(q), q), bcx.at<float>(q),
                                  bcy.at<float>(q), bcz.at<float>(q), bst.at<Stencil>(q), 0.0f, vptr(blsf.at<float>(q), q), t0, t1,
                                  (int64_t)d.h0 * plane, (int64_t)d.h1 * plane - 1))
            rbf_apply_kernel<<<(unsigned)((t1 - t0 + 255) / 256), 256, 0, d.stream>>>(Gq[q], vptr(bw.at<float>(q), q), 1, nx, ny, nz, bcx.at<float>(q),
                                                                                    bcy.at<float>(q), bcz.at<float>(q), bst.at<Stencil>(q), 0.0f,
                                                                                    vptr(blsf.at<float>(q), q), t0, t1);
    }
    SLAB_TRY(sync_slabs(S));
    SLAB_TRY(halo_xchg(blsf, 1));   // the cells of the last owned plane reach into the next one
    // ---- volume-preserving level (:359, :265-300): bisection on the summed row volumes ----
    {
        int lo_i = 0x7FFFFFFF, hi_i = (int)0x80000000;
        for (size_t q : order) {
            const Slab& d = S[q];
            const int64_t no = nowned(q);
            const unsigned nbo = (unsigned)((no + 255) / 256);
            int mmh[2] = {0x7FFFFFFF, (int)0x80000000};
            SLAB_HIP(hipSetDevice(d.device));
            SLAB_HIP(hipMemcpyAsync(bcnt.b[q].p, mmh, 8, hipMemcpyHostToDevice, d.stream));
            minmax_kernel<<<(nbo < 2048u ? nbo : 2048u), 256, 0, d.stream>>>(owned(blsf.at<float>(q), q), no, bcnt.at<int>(q));
        }
        SLAB_TRY(sync_slabs(S));
        for (size_t q : order) {
            int mmh[2];
            SLAB_HIP(hipSetDevice(S[q].device));
            SLAB_HIP(hipMemcpy(mmh, bcnt.b[q].p, 8, hipMemcpyDeviceToHost));
            lo_i = std::min(lo_i, mmh[0]);
            hi_i = std::max(hi_i, mmh[1]);
        }
        auto dec = [](int b) { b = b >= 0 ? b : (b ^ 0x7FFFFFFF); float f; memcpy(&f, &b, 4); return f; };
        float lo = dec(lo_i), hi = dec(hi_i);
        const size_t q0 = order[0];
        const int nrows = (ny - 1) * (nz - 1);
        for (size_t q : order) {
            SLAB_HIP(hipSetDevice(S[q].device));
            SLAB_TRY(vw[q].init(9));
            SLAB_TRY(brows.ensure(q, sizeof(float) * (size_t)std::max(nrows, 1)));
            // segment extrema of the slab's rows (arrays addressed by the row numbers of the whole grid)
            const int nseg = (nx - 1 + 63) / 64, kc1 = std::min(S[q].k1, nz - 1);
            SLAB_TRY(bsegmn.ensure(q, sizeof(float) * (size_t)std::max(nrows, 1) * nseg));
            SLAB_TRY(bsegmx.ensure(q, sizeof(float) * (size_t)std::max(nrows, 1) * nseg));
            if (kc1 > S[q].k0)
                volume_seg_minmax_kernel<<<(kc1 - S[q].k0) * (ny - 1), 256, 0, S[q].stream>>>(vptr(blsf.at<float>(q), q), nx, ny, nz, bsegmn.at<float>(q),
                                                                                             bsegmx.at<float>(q), S[q].k0 * (ny - 1));
        }
        const float edge = std::sqrt((cx[1] - cx[0]) * (cx[1] - cx[0]));
        const float elvol = edge * edge * edge, jac = elvol / 8.0f;
        for (size_t q : order) {
            SLAB_HIP(hipSetDevice(S[q].device));
            SLAB_TRY(vw[q].points(jac, S[q].stream));
        }
        double eps = 1.0;
        int it = 0;
        while (it < 40 && eps > 1.0e-4) {
            th = (lo + hi) / 2;
            for (size_t q : order) {   // rows of the cells between the slab's planes, written at their place in the whole row array
                const Slab& d = S[q];
                const int kc1 = std::min(d.k1, nz - 1);
                if (kc1 <= d.k0) continue;
                const int row0 = d.k0 * (ny - 1), nr = (kc1 - d.k0) * (ny - 1);
                SLAB_HIP(hipSetDevice(d.device));
                volume_rowwave_kernel<<<std::min((nr + 3) / 4, vol_grid()), 256, 0, d.stream>>>(vptr(blsf.at<float>(q), q), nx, ny, nz, th, 0.0f, elvol, jac,
                                                                                          vw[q].q, brows.at<float>(q), row0, nr, bsegmn.at<float>(q),
                                                                                          bsegmx.at<float>(q), vw[q].qpts.as<float4>(), nullptr, nullptr);
                if (q != q0)
                    SLAB_HIP(hipMemcpyPeerAsync(brows.at<float>(q0) + row0, S[q0].device, brows.at<float>(q) + row0, d.device,
                                                sizeof(float) * (size_t)nr, d.stream));
            }
            SLAB_TRY(sync_slabs(S));
            float vol;
            SLAB_HIP(hipSetDevice(S[q0].device));
            sum_f32_kernel<<<1, 1024, 0, S[q0].stream>>>(brows.at<float>(q0), nrows, vw[q0].result.as<float>());
            SLAB_HIP(hipMemcpyAsync(&vol, vw[q0].result.p, sizeof(float), hipMemcpyDeviceToHost, S[q0].stream));
            SLAB_HIP(hipStreamSynchronize(S[q0].stream));
            eps = std::fabs(target_volume - (double)vol);
            if ((double)vol > target_volume) lo = th; else hi = th;
            it++;
        }
        th = -th;
        if (th_out) *th_out = th;
    }
    // ---- fine grid (:363-366): every slab evaluates the fine planes of its coarse planes and sends them to the caller ----
    for (size_t q : order) {
        const Slab& d = S[q];
        const int f0 = d.k0 * smooth, f1 = std::min(fz, (d.k1 == nz) ? fz : d.k1 * smooth);
        if (f1 <= f0) continue;
        const int64_t t0 = (int64_t)f0 * fplane, t1 = (int64_t)f1 * fplane;
        SLAB_HIP(hipSetDevice(d.device));
        SLAB_TRY(bfine.ensure(q, 4 * (size_t)(t1 - t0)));
        if (!(fine_one_to_one && launch_rbf_apply_lut(Gq[q], LGF[q], sts[1], (unsigned)((t1 - t0 + 255) / 256), d.stream, vptr(bw.at<float>(q), q),
                                                      btx.at<float>(q), bty.at<float>(q), btz.at<float>(q), bst.at<Stencil>(q) + 1, th,
                                                      bfine.at<float>(q) - t0, t0, t1, (int64_t)d.h0 * plane, (int64_t)d.h1 * plane - 1)))
            rbf_apply_kernel<<<(unsigned)((t1 - t0 + 255) / 256), 256, 0, d.stream>>>(Gq[q], vptr(bw.at<float>(q), q), smooth, fx, fy, fz, btx.at<float>(q),
                                                                                    bty.at<float>(q), btz.at<float>(q), bst.at<Stencil>(q) + 1, th,
                                                                                    bfine.at<float>(q) - t0, t0, t1);
        SLAB_HIP(hipMemcpyAsync(fine_out_host + t0, bfine.at<float>(q), 4 * (size_t)(t1 - t0), hipMemcpyDeviceToHost, d.stream));
    }
    SLAB_TRY(sync_slabs(S));
done:
    for (size_t q = 0; q < G; ++q) {
        (void)hipSetDevice(S[q].device);
        vw[q].release();
    }
    return rc;
}

int remove_artifacts_dev(double* d_sdf, const r2s_grid* g, double threshold, double min_ratio, hipStream_t st,
                         int64_t* n_flipped)
{
    return ::remove_artifacts_dev(d_sdf, g, threshold, min_ratio, st, n_flipped);
}
int rbf_smooth_dev(const double* d_sdf, const r2s_grid* g, int is_interp, int smooth, double kthr, double target_volume,
                   float* d_fine_out, float* th_out, int* cg_iters, const std::function<int(int64_t, int64_t)>* fine_chunk,
                   void* workspace, bool fine_early)
{
    return rbf_smooth_host(d_sdf, g, is_interp, smooth, kthr, target_volume, d_fine_out, th_out, cg_iters, nullptr, true, true,
                           fine_chunk, (RbfWork*)workspace, fine_early);
}
void* rbf_workspace_create() { return new RbfWork(); }
void rbf_workspace_release(void* w)
{
    if (!w) return;
    ((RbfWork*)w)->release();
    delete (RbfWork*)w;
}
}  // namespace r2s_int

extern "C" {

int r2s_remove_artifacts(double* sdf_inout, const r2s_grid* grid, double threshold, double min_ratio, int32_t device,
                         int64_t* n_flipped)
{
    if (!sdf_inout || !grid) return fail(R2S_ERR_ARG, "null argument");
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf d;
    ENSURE(d, sizeof(double) * (size_t)grid->ngp);
    hipError_t e = hipMemcpy(d.p, sdf_inout, sizeof(double) * (size_t)grid->ngp, hipMemcpyHostToDevice);
    if (e != hipSuccess) { d.release(); return fail(R2S_ERR_HIP, "%s", hipGetErrorString(e)); }
    rc = remove_artifacts_dev(d.as<double>(), grid, threshold, min_ratio, nullptr, n_flipped);
    if (!rc) {
        e = hipMemcpy(sdf_inout, d.p, sizeof(double) * (size_t)grid->ngp, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(R2S_ERR_HIP, "%s", hipGetErrorString(e));
    }
    d.release();
    return rc;
}

int r2s_remove_artifacts_dev(double* d_sdf, const r2s_grid* grid, double threshold, double min_ratio, void* stream,
                             int64_t* n_flipped)
{
    if (!d_sdf || !grid) return fail(R2S_ERR_ARG, "null argument");
    return remove_artifacts_dev(d_sdf, grid, threshold, min_ratio, (hipStream_t)stream, n_flipped);
}

int r2s_volume_from_sdf(const float* sdf, int64_t nx, int64_t ny, int64_t nz, float edge, float iso,
                        int32_t quad_order, int32_t device, float* vol_out)
{
    if (!sdf || !vol_out || nx < 2 || ny < 2 || nz < 2) return fail(R2S_ERR_ARG, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    VolumeWork vw;
    DevBuf d;
    rc = vw.init(quad_order);
    if (rc) return rc;
    const size_t bytes = sizeof(float) * (size_t)(nx * ny * nz);
    if (d.ensure(bytes)) { vw.release(); return fail(R2S_ERR_NOMEM, "hipMalloc failed"); }
    hipError_t e = hipMemcpy(d.p, sdf, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(R2S_ERR_HIP, "%s", hipGetErrorString(e));
    if (!rc) rc = vw.run(d.as<float>(), (int)nx, (int)ny, (int)nz, edge, 0.0f, iso, nullptr, vol_out);
    d.release();
    vw.release();
    return rc;
}

int r2s_rbf_smooth(const double* sdf, const r2s_grid* grid, int32_t is_interp, int32_t smooth, double kernel_threshold,
                   double target_volume, int32_t device, float* fine_sdf_out, float* level_shift_out,
                   int32_t* cg_iters_out, float* lsf_out)
{
    int rc = use_device(device);
    if (rc) return rc;
    int its = 0;
    rc = rbf_smooth_host(sdf, grid, is_interp, smooth, kernel_threshold, target_volume, fine_sdf_out, level_shift_out,
                         &its, lsf_out);
    if (cg_iters_out) *cg_iters_out = its;
    return rc;
}

/* device-resident variant: d_sdf (Float64, grid points) and d_fine_out (Float32, fine grid points) are device
 * pointers on the current device; work queued on `stream` is waited for first; synchronous on return */
int r2s_rbf_smooth_dev(const double* d_sdf, const r2s_grid* grid, int32_t is_interp, int32_t smooth, double kernel_threshold,
                       double target_volume, float* d_fine_out, float* level_shift_out, int32_t* cg_iters_out, void* stream)
{
    if (!d_sdf || !grid || !d_fine_out) return fail(R2S_ERR_ARG, "null argument");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    int its = 0;
    int rc = rbf_smooth_host(d_sdf, grid, is_interp, smooth, kernel_threshold, target_volume, d_fine_out, level_shift_out,
                             &its, nullptr, true, true);
    if (cg_iters_out) *cg_iters_out = its;
    return rc;
}

/* frees the process-wide work buffers kept between calls (the materialised RBF matrix) */
void r2s_release_cache(void)
{
    {
        std::lock_guard<std::mutex> lock(g_rbf_kv_mutex);
        g_rbf_kv.release();
    }
    release_ccl_work();
    r2s_int::release_host_sessions();
}

}  // extern "C"
