// MI355X (gfx950) signed-distance extraction engine: kernels + C ABI.
//
// Pipeline of one r2s_plan_run_dev() call (stream 1 = the caller's stream, stream 2 = the plan's own):
//   node_degree / ine_fill        node -> element CSR                       (MeshInformations.jl:69-77)
//   elem_prep / face_mask         element records (+ solver constants), solid/iso class, boundary faces
//   item_build                    band work items (boundary triangles + iso projections) and their boxes
//   band_bin / sign_hot / sign_bin / sign_box / scan / active_tiles / bin_sort
//                                 4x4x4-voxel tile -> item list and (hot tiles) -> candidate element list,
//                                 boxes of the candidate elements, compact lists of the tiles to gather
//   fill                          sentinel sweep (HBM-bound)                                  [stream 2]
//   sign_project                  item-major Newton inverse maps of the sign pass (HEX8)      [stream 2]
//   iso_project_hex_pl            persistent lane-refill SQP projection onto the iso-surface  [stream 1]
//   sdf_tiles                     one wavefront per active tile: ordered gather over the tile's lists
//                                 (look-ups of the item-major results, triangles inline), writes dist*sign
// See DESIGN.md for the data layout, the flow and the measured numbers of each kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <utility>
#include <string>
#include <vector>

#include "../../include/rho2sdf_hip.h"
#include "r2s_common.hpp"
#include "r2s_device_math.hpp"

#include <type_traits>

using namespace r2s;

// ------------------------------------------------------------------------------------
// element topology (src/ElementTypes/ElementTypes.jl:15-78), 0-based
// ------------------------------------------------------------------------------------
__constant__ int c_hex_isn[6][4] = {{0, 3, 2, 1}, {0, 1, 5, 4}, {1, 2, 6, 5},
                                    {2, 3, 7, 6}, {3, 0, 4, 7}, {4, 5, 6, 7}};

// ------------------------------------------------------------------------------------
// exclusive scan of uint32 (block = 1024 threads x 4 items)
// ------------------------------------------------------------------------------------
#define SCAN_BLOCK 256   // (4 wavefronts: a 1024-thread workgroup needs 4 free wave slots with registers on EVERY SIMD of
#define SCAN_ITEMS 16    //  one CU at once and waited 2.8 ms for them beside the persistent projection kernel)
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

// up to two arrays of the same length per launch (blockIdx.y picks the array): the pipeline's count arrays
// come in pairs (band / sign lists, work / storage chunks) and every launch saved is ~5 us of a 0.5 ms prologue
struct ScanPair {
    const uint32_t* in[2];
    uint32_t* out[2];
    uint32_t* sums[2];   // per-block totals (nullptr: not wanted)
};
__global__ void __launch_bounds__(SCAN_BLOCK) scan_block_kernel(ScanPair sp, int64_t n)
{
    __shared__ uint32_t wave_sums[SCAN_BLOCK / 64];
    const uint32_t* __restrict__ in = sp.in[blockIdx.y];
    uint32_t* __restrict__ out = sp.out[blockIdx.y];
    uint32_t* __restrict__ block_sums = sp.sums[blockIdx.y];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0u;
        sum += v[i];
    }
    // inclusive scan of per-thread sums inside the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        if (w < wave) wave_off += wave_sums[w];
        total += wave_sums[w];
    }
    uint32_t run = wave_off + incl - sum;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == 0 && block_sums) block_sums[blockIdx.x] = total;
}

__global__ void scan_add_kernel(ScanPair sp, int64_t n)
{
    uint32_t* __restrict__ out = sp.out[blockIdx.y];
    const int64_t i = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x;
    const uint32_t add = sp.sums[blockIdx.y][blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t j = i + (int64_t)k * SCAN_BLOCK;
        if (j < n) out[j] += add;
    }
}

// zero up to four buffers with one launch (blockIdx.y picks the buffer); sizes are in 32-bit words and may
// be rounded up - every buffer comes from DevBuf::ensure, which over-allocates by >= 256 bytes
struct ZeroSet {
    uint32_t* p[4];
    int64_t nwords[4];
};
__global__ void zero_many_kernel(ZeroSet z)
{
    uint32_t* __restrict__ p = z.p[blockIdx.y];
    const int64_t n = z.nwords[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static void zero_many(hipStream_t st, std::initializer_list<std::pair<void*, size_t>> bufs)
{
    ZeroSet z;
    unsigned nb = 0;
    int64_t mx = 0;
    for (const auto& b : bufs) {
        z.p[nb] = static_cast<uint32_t*>(b.first);
        z.nwords[nb] = (int64_t)((b.second + 3) / 4);
        mx = std::max(mx, z.nwords[nb]);
        nb++;
    }
    for (unsigned i = nb; i < 4; ++i) { z.p[i] = nullptr; z.nwords[i] = 0; }
    const unsigned gx = (unsigned)std::min<int64_t>((mx + 1023) / 1024, 2048);
    if (nb && gx) zero_many_kernel<<<dim3(gx, nb), 256, 0, st>>>(z);
}

// ------------------------------------------------------------------------------------
// mesh preparation kernels
// ------------------------------------------------------------------------------------
// The scalars the host needs between stages (array sizes) travel to pinned host memory in ONE small kernel
// per read-back point instead of one blit kernel per value.
struct ReadBack {
    const uint32_t* src[8];
    uint32_t n[8];     // words per entry (<= 8)
    uint32_t dst[8];   // first word in the pinned block
    int cnt = 0;
    void add(const uint32_t* p, uint32_t words, uint32_t at) { src[cnt] = p; n[cnt] = words; dst[cnt] = at; ++cnt; }
};
// Speculation: a plan remembers the sizes the host read back in its last call with the same shapes.  The next
// call launches everything with those sizes and never waits in the middle; this kernel compares them with the
// sizes that really came out and raises the call's abort flag when one differs - every kernel that could then
// write (or read) outside what was allocated returns at once, and the host repeats the call the slow way.
struct SpecExpect {
    uint32_t v[16];
    uint32_t on;
};
__global__ void read_back_kernel(ReadBack rb, uint32_t* __restrict__ host, SpecExpect ex, uint32_t* __restrict__ abort_flag)
{
    const int q = threadIdx.x >> 3, w = threadIdx.x & 7;
    if (q < rb.cnt && (uint32_t)w < rb.n[q]) {
        const uint32_t v = rb.src[q][w];
        host[rb.dst[q] + w] = v;
        if (ex.on && v != ex.v[rb.dst[q] + w]) *abort_flag = 1u;
    }
}

__global__ void node_degree_kernel(const int64_t* __restrict__ IEN, int64_t nel, int nen, int64_t nnp,
                                   uint32_t* __restrict__ deg, int* __restrict__ bad)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nel * nen) return;
    int64_t n = IEN[t] - 1;
    if (n < 0 || n >= nnp) { *bad = 1; return; }
    atomicAdd(&deg[n], 1u);
}

__global__ void ine_fill_kernel(const int64_t* __restrict__ IEN, int64_t nel, int nen, int64_t nnp,
                                const uint32_t* __restrict__ ptr, uint32_t* __restrict__ cursor,
                                uint32_t* __restrict__ ine)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nel * nen) return;
    int64_t n = IEN[t] - 1;
    if (n < 0 || n >= nnp) return;
    uint32_t pos = atomicAdd(&cursor[n], 1u);
    ine[ptr[n] + pos] = (uint32_t)(t / nen);
}

// element classes
#define CLS_SKIP 0
#define CLS_SOLID 1
#define CLS_ISO 2

// The part of the grid one call (one rank) computes, as LOCAL planes 0..nzl-1:
//   contiguous  (G == 1): local plane kl = lattice plane k0 + kl, planes [k0,k1)
//   interleaved (G  > 1): the rank owns the 4-plane tile layers tz with tz % G == r of the whole grid;
//                         local layer t = global layer t*G + r (balanced work for the Z-slab all-gather)
struct SlabInfo {
    int32_t k0, k1;          // lattice planes [k0,k1) (whole grid when interleaved)
    int32_t ntx, nty, ntz;   // local tiles
    int32_t ntiles;
    int32_t G, r;            // interleave stride / phase
    int32_t nzl;             // local planes
};

__host__ __device__ __forceinline__ int slab_global_k(const SlabInfo& s, int kl)
{
    return s.G == 1 ? kl + s.k0 : ((kl >> 2) * s.G + s.r) * 4 + (kl & 3);
}

// lattice planes [a,b] (inclusive) -> local planes [la,lb]; false if the slab holds none of them
__host__ __device__ __forceinline__ bool slab_local_range(const SlabInfo& s, int a, int b, int& la, int& lb)
{
    if (a < s.k0) a = s.k0;
    if (b >= s.k1) b = s.k1 - 1;
    if (a > b) return false;
    if (s.G == 1) {
        la = a - s.k0;
        lb = b - s.k0;
        return true;
    }
    const int layer_a = a >> 2, layer_b = b >> 2;
    int ta = layer_a - s.r, tb = layer_b - s.r;
    ta = ta <= 0 ? 0 : (ta + s.G - 1) / s.G;          // first owned layer >= layer_a
    if (tb < 0) return false;
    tb = tb / s.G;                                      // last owned layer <= layer_b
    if (ta > tb) return false;
    la = 4 * ta + ((ta * s.G + s.r == layer_a) ? (a & 3) : 0);
    lb = 4 * tb + ((tb * s.G + s.r == layer_b) ? (b & 3) : 3);
    return la <= lb;
}

// n x n LU with partial pivoting, same operation order as the oracle's lu_solve (stands in for
// LAPACK getrf behind Julia's `\`); piv[c] = row swapped with row c.  Prep kernels only.
template <int N>
__device__ void lu_factor(double A[N][N], int32_t* piv, int32_t& sing)
{
    sing = 0;
    for (int c = 0; c < N; ++c) {
        int p = c;
        double best = fabs(A[c][c]);
        for (int r = c + 1; r < N; ++r)
            if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); p = r; }
        if (best == 0.0) sing = 1;
        if (c < N - 1) piv[c] = p;
        if (p != c)
            for (int k = 0; k < N; ++k) { const double t = A[c][k]; A[c][k] = A[p][k]; A[p][k] = t; }
        for (int r = c + 1; r < N; ++r) {
            const double l = A[r][c] / A[c][c];
            A[r][c] = l;
            for (int k = c + 1; k < N; ++k) A[r][k] -= l * A[c][k];
        }
    }
}

// element-type traits: record type, topology (src/ElementTypes/ElementTypes.jl:15-78) and the
// per-element precomputation the per-voxel tests need
struct HexT {
    using Rec = ElemRec;
    static constexpr int NEN = 8, NES = 6, NSN = 4;
    static __device__ __forceinline__ int face(int sg, int a) { return c_hex_isn[sg][a]; }
    static __device__ void finish(Rec& R, const GridDev&, double, bool) { hex8_monomials(R); hex8_newton0(R); }   // + hex_planes_kernel
    static __device__ __forceinline__ void store(Rec& D, const Rec& R, bool) { D = R; }
};
struct TetT {
    using Rec = TetRec;
    static constexpr int NEN = 4, NES = 4, NSN = 3;
    static __device__ __forceinline__ int face(int sg, int a) { return c_tet_isn[sg][a]; }
    // (iso: the iso-surface passes through the element - only then are the projection constants needed)
    static __device__ __forceinline__ void store(Rec& D, const Rec& R, bool iso)
    {
        static_cast<TetGeom&>(D) = R;
        if (iso) static_cast<TetIso&>(D) = R;
        static_cast<TetPlanes&>(D) = R;
    }
    static __device__ void finish(Rec& R, const GridDev& g, double rho_t, bool iso)
    {
        double A[3][3], T[4][4];
        for (int i = 0; i < 3; ++i) {   // FindLocalCoordinates.jl:124: hcat(x2-x1, x3-x1, x4-x1)
            A[i][0] = R.X[1][i] - R.X[0][i];
            A[i][1] = R.X[2][i] - R.X[0][i];
            A[i][2] = R.X[3][i] - R.X[0][i];
        }
        lu_factor<3>(A, R.p3, R.sing3);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R.lu3[r][c] = A[r][c];
        for (int r = 0; r < 3; ++r)     // SignDetection.jl:236: [v1 v2 v3 v4; 1 1 1 1]
            for (int c = 0; c < 4; ++c) T[r][c] = R.X[c][r];
        for (int c = 0; c < 4; ++c) T[3][c] = 1.0;
        lu_factor<4>(T, R.p4, R.sing4);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) R.lu4[r][c] = T[r][c];
        const int dims[3] = {g.nx, g.ny, g.nz};
        for (int ax = 0; ax < 3; ++ax) {   // SignDetection.jl:191-192 (1-based bin indices)
            double a = floor((R.mn[ax] - g.amin[ax]) / g.cell) - 1.0;
            double b = ceil((R.mx[ax] - g.amin[ax]) / g.cell) + 1.0;
            if (a < 1.0) a = 1.0;
            if (b > (double)dims[ax]) b = (double)dims[ax];
            R.blo[ax] = (int32_t)fmin(a, 2.0e9);
            R.bhi[ax] = (int32_t)fmax(b, -2.0e9);
        }
        R.pad = 0;
        if (iso) tet4_iso_constants(R, rho_t);
        tet4_face_planes(R);
    }
};

// an element outside this call's share of the grid: what of its (cleared) record says "no sign candidate anywhere"
__device__ __forceinline__ void mark_skipped(ElemRec&) {}   // (HEX8: the sign bins go by the AABB, which the skip path stores)
__device__ __forceinline__ void mark_skipped(TetRec& E)     // (TET4: by the bin range of ET::finish - an empty one)
{
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { E.blo[ax] = 1; E.bhi[ax] = 0; }
}

// 1 thread / element: gather record, classify (sdfOnDensityField.jl:197-201,312), boundary
// faces (:511-519), work-item count, lattice range of the element AABB for the sign bins.
template <class ET>
__global__ void elem_prep_kernel(const double* __restrict__ X, const int64_t* __restrict__ IEN,
                                 const double* __restrict__ rho_n, int64_t nel, int64_t nnp, double rho_t,
                                 GridDev g, typename ET::Rec* __restrict__ erec, uint8_t* __restrict__ cls,
                                 uint32_t* __restrict__ fmask, uint32_t* __restrict__ nitems, SlabInfo sl, double delta)
{
    int64_t el = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (el >= nel) return;
    int64_t nd[ET::NEN];
    typename ET::Rec R;
    double rmin = INFINITY, rmax = -INFINITY;
#pragma unroll
    for (int a = 0; a < ET::NEN; ++a) {
        nd[a] = IEN[el * ET::NEN + a] - 1;
        if ((uint64_t)nd[a] >= (uint64_t)nnp) nd[a] = 0;   // reported by node_degree_kernel; keep the loads in range
#pragma unroll
        for (int i = 0; i < 3; ++i) R.X[a][i] = X[3 * nd[a] + i];
        R.r[a] = rho_n[nd[a]];
        if (R.r[a] < rmin) rmin = R.r[a];
        if (R.r[a] > rmax) rmax = R.r[a];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double mn = R.X[0][i], mx = R.X[0][i];
#pragma unroll
        for (int a = 1; a < ET::NEN; ++a) {
            if (R.X[a][i] < mn) mn = R.X[a][i];
            if (R.X[a][i] > mx) mx = R.X[a][i];
        }
        R.mn[i] = mn;
        R.mx[i] = mx;
    }
    R.rmax = rmax;
    R.rmin = rmin;
    {
        // An element none of whose lattice planes (AABB widened by the band, one plane of slack on either side) belongs
        // to this call's share of the grid contributes nothing here: no record, no items.  One rank of eight sees an
        // eighth (Z-slabs) to a half (interleaved tile layers) of the elements.
        int a = (int)floor((R.mn[2] - delta - g.amin[2]) / g.cell) - 1, b = (int)ceil((R.mx[2] + delta - g.amin[2]) / g.cell) + 1;
        if (a < 0) a = 0;
        if (b > g.nz - 1) b = g.nz - 1;
        int la, lb;
        if (a > b || !slab_local_range(sl, a, b, la, lb)) {
            // The rest of the record is CLEARED, not left as it was, and marked as holding no sign candidate: the plan's
            // arrays outlive a call, and the TET4 sign bins go by a field this path did not write (TetRec::blo / bhi) -
            // what an earlier call (another mesh, another share of the grid) had stored at this index became a
            // candidate of a later one.  Found by tools/fuzz_fan_out.py: a TET4 call on 8 devices after a TET4 call on
            // 5 - 145 signs differed from the one-device result until the plans were released.
            {
                uint64_t* w = reinterpret_cast<uint64_t*>(&erec[el]);
                static_assert(sizeof(typename ET::Rec) % 8 == 0, "element records are cleared in 8-byte words");
#pragma unroll 8
                for (size_t q = 0; q < sizeof(typename ET::Rec) / 8; ++q) w[q] = 0ull;
            }
            mark_skipped(erec[el]);
#pragma unroll
            for (int i = 0; i < 3; ++i) { erec[el].mn[i] = R.mn[i]; erec[el].mx[i] = R.mx[i]; }   // (the bin kernels look at the AABB)
            erec[el].rmax = rmax;
            erec[el].rmin = rmin;
            cls[el] = CLS_SKIP;
            fmask[el] = 0u;
            nitems[el] = 0u;
            return;
        }
    }
    int c = CLS_SKIP;
    if (rmin >= rho_t) c = CLS_SOLID;
    else if (rmax > rho_t) c = CLS_ISO;
    ET::finish(R, g, rho_t, c == CLS_ISO);
    ET::store(erec[el], R, c == CLS_ISO);
    cls[el] = (uint8_t)c;
    fmask[el] = 0u;
    nitems[el] = (c == CLS_ISO) ? 1u : 0u;   // + the boundary-face triangles, added by face_mask_kernel
}

// bounding half-spaces of the inflated elements (ElemRec::pn / po, used by sign_project_kernel only): one thread
// per (element, face), on the second stream - off the critical path of the mesh preparation
__global__ void hex_planes_kernel(ElemRec* __restrict__ erec, int64_t nel, int no_inner)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t el = gid / 6;
    const int f = (int)(gid % 6);
    if (el >= nel) return;
    double n[3], po, pin;
    hex8_plane(erec[el], f >> 1, f & 1, n, po, pin);
    erec[el].pn[f][0] = n[0]; erec[el].pn[f][1] = n[1]; erec[el].pn[f][2] = n[2];
    erec[el].po[f] = po;
    // no_inner (R2S_SIGN_NO_INNER=1): no inner region, every candidate pair of the sign pass runs its Newton solve - the
    // shortcut assumes a CONFORMING mesh (include/rho2sdf_hip.h, r2s_sign_detection)
    erec[el].pi[f] = no_inner ? -INFINITY : pin;
}

// boundary faces (sdfOnDensityField.jl:511-519): a face is on the boundary when exactly one element (this one)
// holds all its nodes.  One thread per (element, face) - the search is a chain of dependent look-ups in the
// node -> element lists, so it wants many threads, and elements that are neither solid nor iso leave at once.
template <class ET>
__global__ void face_mask_kernel(const int64_t* __restrict__ IEN, int64_t nel, const uint32_t* __restrict__ ine_ptr,
                                 const uint32_t* __restrict__ ine, const uint8_t* __restrict__ cls,
                                 uint32_t* __restrict__ fmask, uint32_t* __restrict__ nitems,
                                 const int* __restrict__ bad)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t el = gid / ET::NES;
    const int sg = (int)(gid % ET::NES);
    if (el >= nel || cls[el] == CLS_SKIP || *bad) return;   // bad: IEN holds ids outside 1..nnp, the call fails after this kernel
    // elements that hold all nodes of the face: the candidates are the elements of its first node; whether a candidate
    // holds the other nodes is read off its own connectivity row (one contiguous row instead of a scan of the other
    // nodes' element lists)
    int64_t fn[ET::NSN];
#pragma unroll
    for (int a = 0; a < ET::NSN; ++a) fn[a] = IEN[el * ET::NEN + ET::face(sg, a)];
    const int64_t n0 = fn[0] - 1;
    int common = 0;
    for (uint32_t p = ine_ptr[n0]; p < ine_ptr[n0 + 1]; ++p) {
        const int64_t* __restrict__ row = IEN + (int64_t)ine[p] * ET::NEN;
        int64_t rn[ET::NEN];
#pragma unroll
        for (int m = 0; m < ET::NEN; ++m) rn[m] = row[m];
        bool all = true;
#pragma unroll
        for (int a = 1; a < ET::NSN; ++a) {
            bool found = false;
#pragma unroll
            for (int m = 0; m < ET::NEN; ++m) found = found || rn[m] == fn[a];
            all = all && found;
        }
        common += all ? 1 : 0;
    }
    if (common == 1) {
        atomicOr(&fmask[el], 1u << sg);
        atomicAdd(&nitems[el], (uint32_t)ET::NSN);
    }
}

// mini-AABB cell range of a coordinate interval (Grid.jl:130-145)
__device__ __forceinline__ void mini_range(const GridDev& g, int ax, double mn, double mx, double delta,
                                           int32_t& imin, int32_t& imax)
{
    double lo = cell_of(g, ax, mn - delta);
    double hi = cell_of(g, ax, mx + delta);
    if (lo < 0) lo = 0;
    if (hi >= (double)g.N[ax]) hi = (double)g.N[ax];
    // keep empty ranges empty, but inside int32
    lo = fmin(lo, 2.0e9);
    hi = fmax(hi, -2.0e9);
    imin = (int32_t)lo;
    imax = (int32_t)hi;
}

// 1 thread / element with items: writes its triangles (face order sg, fan order a) and then the
// iso item - the reference's processing order inside one element (:584-624).
// ---- storage of item-major results ---------------------------------------------------------
// Results of an item (iso projections, inverse maps of the sign pass) are stored per 4x4x4 voxel tile of
// the item's lattice box, in the lane order of the per-tile gather kernel (x + 4 y + 16 z inside the tile):
// the gather reads one contiguous 512 B line per (tile, item), with a wave-uniform base address.
struct TileBox {
    int32_t t0[3];   // first tile per axis (Z: tiles of LOCAL planes)
    int32_t td[3];   // tiles per axis
};
__host__ __device__ __forceinline__ TileBox tile_box(const int32_t lo[3], const int32_t dim[3])
{
    TileBox b;
    for (int ax = 0; ax < 3; ++ax) {
        b.t0[ax] = lo[ax] >> 2;
        b.td[ax] = (dim[ax] > 0) ? (((lo[ax] + dim[ax] - 1) >> 2) - b.t0[ax] + 1) : 0;
    }
    return b;
}
__device__ __forceinline__ size_t tile_slot(uint32_t store_off, const TileBox& b, int i, int j, int k)
{
    const uint32_t tile = (uint32_t)((((k >> 2) - b.t0[2]) * b.td[1] + ((j >> 2) - b.t0[1])) * b.td[0] + ((i >> 2) - b.t0[0]));
    return ((size_t)store_off + tile) * 64u + (uint32_t)((i & 3) | ((j & 3) << 2) | ((k & 3) << 4));
}

// wave-uniform form for the per-tile gather: storage chunk of tile (tx,ty,tz) inside an item's tile box
__device__ __forceinline__ uint32_t tile_chunk(uint32_t store_off, const TileBox& b, int tx, int ty, int tz)
{
    return store_off + (uint32_t)(((tz - b.t0[2]) * b.td[1] + (ty - b.t0[1])) * b.td[0] + (tx - b.t0[0]));
}

template <class ET>
__global__ void item_build_kernel(const typename ET::Rec* __restrict__ erec, const uint8_t* __restrict__ cls,
                                  const uint32_t* __restrict__ fmask, const uint32_t* __restrict__ item_off,
                                  int64_t nel, GridDev g, SlabInfo sl, double delta,
                                  BandItem* __restrict__ items, uint32_t* __restrict__ nchunks,
                                  uint32_t* __restrict__ nstore, uint8_t* __restrict__ hard, double rho_t, const uint32_t* __restrict__ abort_flag)
{
    if (*abort_flag) return;   // speculated sizes of this call did not hold (run_impl)
    int64_t el = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (el >= nel) return;
    const int c = cls[el];
    if (c == CLS_SKIP) return;
    const typename ET::Rec& E = erec[el];
    uint32_t w = item_off[el];
    const uint32_t fm = fmask[el];
    for (int sg = 0; sg < ET::NES; ++sg) {
        if (!(fm & (1u << sg))) continue;
        double Xs[4][3], Xc[3];
        for (int a = 0; a < ET::NSN; ++a)
            for (int i = 0; i < 3; ++i) Xs[a][i] = E.X[ET::face(sg, a)][i];
        for (int i = 0; i < 3; ++i) {   // Xc = vec(mean(Xs, dims = 2)) (sdfOnDensityField.jl:521)
            double s = Xs[0][i];
            for (int a = 1; a < ET::NSN; ++a) s += Xs[a][i];
            Xc[i] = s / (double)ET::NSN;
        }
        for (int a = 0; a < ET::NSN; ++a) {
            BandItem T;
            const int b = (a + 1) % ET::NSN;
            for (int i = 0; i < 3; ++i) { T.tri[0][i] = Xs[a][i]; T.tri[1][i] = Xs[b][i]; T.tri[2][i] = Xc[i]; }
            double Et[3][3];
            for (int i = 0; i < 3; ++i) {
                Et[0][i] = T.tri[1][i] - T.tri[0][i];
                Et[1][i] = T.tri[2][i] - T.tri[1][i];
                Et[2][i] = T.tri[0][i] - T.tri[2][i];
            }
            double n0 = Et[0][1] * Et[1][2] - Et[0][2] * Et[1][1];
            double n1 = Et[0][2] * Et[1][0] - Et[0][0] * Et[1][2];
            double n2 = Et[0][0] * Et[1][1] - Et[0][1] * Et[1][0];
            const double nn = norm3(n0, n1, n2);
            n0 /= nn; n1 /= nn; n2 /= nn;
            T.n[0] = n0; T.n[1] = n1; T.n[2] = n2;
            for (int j = 0; j < 3; ++j) {
                const double L = norm3(Et[j][0], Et[j][1], Et[j][2]);
                T.L[j] = L;
                for (int i = 0; i < 3; ++i) T.eh[j][i] = Et[j][i] / L;
            }
            // barycentric matrix (TriangularMeshUtils.jl:9-21) and its partial-pivot LU
            double A[3][3];
            for (int v = 0; v < 3; ++v) {
                const double* p = T.tri[v];
                A[0][v] = p[1] * n2 - p[2] * n1;
                A[1][v] = p[2] * n0 - p[0] * n2;
                A[2][v] = p[0] * n1 - p[1] * n0;
            }
            int im = 0;
            double nm = fabs(n0);
            if (fabs(n1) > nm) { nm = fabs(n1); im = 1; }
            if (fabs(n2) > nm) { nm = fabs(n2); im = 2; }
            for (int v = 0; v < 3; ++v) A[im][v] = 1.0;
            T.im = im;
            int sing = 0, p0 = 0, p1 = 1;
            {
                double best = fabs(A[0][0]);
                if (fabs(A[1][0]) > best) { best = fabs(A[1][0]); p0 = 1; }
                if (fabs(A[2][0]) > best) { best = fabs(A[2][0]); p0 = 2; }
                if (best == 0.0) sing = 1;
                if (p0 != 0)
                    for (int k = 0; k < 3; ++k) { double t = A[0][k]; A[0][k] = A[p0][k]; A[p0][k] = t; }
                for (int r = 1; r < 3; ++r) {
                    const double l = A[r][0] / A[0][0];
                    A[r][0] = l;
                    A[r][1] -= l * A[0][1];
                    A[r][2] -= l * A[0][2];
                }
                best = fabs(A[1][1]);
                if (fabs(A[2][1]) > best) { best = fabs(A[2][1]); p1 = 2; }
                if (best == 0.0) sing = 1;
                if (p1 != 1)
                    for (int k = 0; k < 3; ++k) { double t = A[1][k]; A[1][k] = A[2][k]; A[2][k] = t; }
                const double l = A[2][1] / A[1][1];
                A[2][1] = l;
                A[2][2] -= l * A[1][2];
                if (A[2][2] == 0.0) sing = 1;
            }
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) T.lu[r][k] = A[r][k];
            T.p0 = p0; T.p1 = p1; T.sing = sing;
            T.el = (int32_t)el;
            T.kind = (c == CLS_SOLID) ? 1 : 2;
            for (int ax = 0; ax < 3; ++ax) {
                double mn = fmin(fmin(T.tri[0][ax], T.tri[1][ax]), T.tri[2][ax]);
                double mx = fmax(fmax(T.tri[0][ax], T.tri[1][ax]), T.tri[2][ax]);
                mini_range(g, ax, mn, mx, delta, T.imin[ax], T.imax[ax]);
            }
            T.lo[0] = T.lo[1] = T.lo[2] = 0;
            T.dim[0] = T.dim[1] = T.dim[2] = 0;
            T.chunk_off = 0;
            T.store_off = 0;
            nchunks[w] = 0;
            nstore[w] = 0;
            items[w++] = T;
        }
    }
    if (c == CLS_ISO) {
        BandItem T;
        memset(&T, 0, sizeof T);
        T.el = (int32_t)el;
        T.kind = 0;
        for (int ax = 0; ax < 3; ++ax) mini_range(g, ax, E.mn[ax], E.mx[ax], delta, T.imin[ax], T.imax[ax]);
        // lattice box swept by iso_project_kernel: voxels i with cell(i) in [imin,imax] have
        // lattice index in [imin, imax+1]; Z clipped to the slab
        const int nmax[3] = {g.nx - 1, g.ny - 1, g.nz - 1};
        uint64_t vol = 1;
        for (int ax = 0; ax < 3; ++ax) {
            int a = T.imin[ax], b = T.imax[ax];
            if (a > b || a > nmax[ax]) { a = 0; b = -1; }
            else {
                b = (b >= nmax[ax]) ? nmax[ax] : b + 1;
                // cell(i) (Grid.jl:58) is monotone in i and equals i or i-1: trim the two ends
                // to the lattice indices whose cell really lies in [imin,imax]
                if ((int)cell_of(g, ax, grid_coord(g, ax, a)) < T.imin[ax]) a += 1;
                if ((int)cell_of(g, ax, grid_coord(g, ax, b)) > T.imax[ax]) b -= 1;
                if (ax == 2) {   // Z in LOCAL planes of the slab
                    int la, lb;
                    if (a <= b && slab_local_range(sl, a, b, la, lb)) { a = la; b = lb; }
                    else { a = 0; b = -1; }
                }
            }
            T.lo[ax] = a;
            T.dim[ax] = (b >= a) ? (b - a + 1) : 0;
            vol *= (uint64_t)T.dim[ax];
        }
        nchunks[w] = (uint32_t)((vol + 63) / 64);
        {
            const TileBox tb = tile_box(T.lo, T.dim);
            nstore[w] = vol ? (uint32_t)(tb.td[0] * tb.td[1] * tb.td[2]) : 0u;
        }
        if constexpr (std::is_same<typename ET::Rec, ElemRec>::value) {
            // iso-surface close to a node: the elements it only clips near a corner or follows along a face, where the
            // SQP needs tens of iterations - they go first in the work order (work_order_kernel)
            double dmin = INFINITY;
            for (int k = 0; k < 8; ++k) dmin = fmin(dmin, fabs(E.r[k] - rho_t));
            // rank 0 (hardest) .. 15: distance of the nearest nodal density from the threshold in 1/32 of the element's
            // density range (round 2: the dozen wavefronts that used to end the kernel 0.3-0.7 ms after the work ran dry
            // each held ONE voxel of an element with a ratio of 0.10-0.17, just beyond the former yes/no limit of 0.1)
            const double ratio = dmin / (E.rmax - E.rmin);
            hard[w] = (uint8_t)(ratio < 15.0 / 32.0 ? (int)(ratio * 32.0) : 15);
        }
        items[w++] = T;
    }
}

// Work order of the persistent projection kernel (HEX8): items whose iso-surface passes close to a node come
// first (16 ranks, see item_build_kernel).  Those are the elements the surface only clips near a corner or follows along a face, and that is where
// the SQP needs tens of iterations (degenerate or nearly infeasible sub-problems) instead of four.  A lane works
// through such a voxel alone, so when one of them is handed out near the end of the kernel everything waits for
// it (the drain phase used to be a quarter of the kernel's run time); handed out first, it finishes in the
// shadow of the bulk.  One block; the order inside the two classes is whatever the atomics give - it has no
// influence on the results (every pair writes its own slot).
__global__ void __launch_bounds__(1024) work_order_kernel(const uint8_t* __restrict__ hard, const uint32_t* __restrict__ nchunks,
                                                          uint32_t nitems, uint32_t* __restrict__ perm,
                                                          uint32_t* __restrict__ wchunks)
{
    // counting sort over the 16 hardness ranks (rank 0 first); items that are not iso items carry rank 0 and no chunks
    __shared__ uint32_t cnt[16], base[16], cur[16];
    if (threadIdx.x < 16) { cnt[threadIdx.x] = 0; cur[threadIdx.x] = 0; }
    __syncthreads();
    for (uint32_t it = threadIdx.x; it < nitems; it += blockDim.x) atomicAdd(&cnt[hard[it] & 15], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t b = 0;
        for (int q = 0; q < 16; ++q) { base[q] = b; b += cnt[q]; }
    }
    __syncthreads();
    for (uint32_t it = threadIdx.x; it < nitems; it += blockDim.x) {
        const uint32_t r = hard[it] & 15;
        const uint32_t pos = base[r] + atomicAdd(&cur[r], 1u);
        perm[pos] = it;
        wchunks[pos] = nchunks[it];
    }
    if (threadIdx.x == 0) wchunks[nitems] = 0;
}

// writes the scanned chunk offsets back into the items
__global__ void item_chunks_kernel(BandItem* __restrict__ items, const uint32_t* __restrict__ chunk_off,
                                   const uint32_t* __restrict__ store_off, uint32_t nitems)
{
    uint32_t it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it < nitems) {
        items[it].chunk_off = chunk_off[it];
        items[it].store_off = store_off[it];
    }
}

// Item-major projection onto the iso-surface (process_isocontour_element!, :606-624): one
// wavefront per 64-voxel chunk of an iso item's lattice box, element record in SGPRs, every
// lane a different voxel of the same element.  Results go to `res` (distance) and, when
// requested, `res_xp`; the ordered per-voxel gather (sdf_tiles_kernel) consumes them in the
// reference's element order, so the strict-'<' update semantics are unchanged.
template <class Rec>
// 205 VGPRs -> 2 waves/SIMD, no scratch.  Forcing 3 waves/SIMD (168 VGPRs) is 2.5 % faster but spills
// 156 B/lane, i.e. ~2.8 GB of scratch traffic per launch against 0.17 GB of algorithmic writes.
__global__ void __launch_bounds__(256) iso_project_kernel(const BandItem* __restrict__ items, uint32_t nitems,
                                                         const uint32_t* __restrict__ chunk_off,
                                                         uint32_t nchunks, const Rec* __restrict__ erec,
                                                         GridDev g, SlabInfo sl, double rho_t, double* __restrict__ res,
                                                         double* __restrict__ res_xp, const uint32_t* __restrict__ abort_flag,
                                                         uint32_t cpw)
{
    // `cpw` consecutive chunks per wavefront: the search for the item of a chunk (a chain of dependent scalar loads,
    // ~10 us of latency for 30 000 items - four times what the projection of the chunk's 64 voxels takes) and the
    // element record are paid once per run of chunks instead of once per chunk
    const uint32_t ab = *abort_flag;
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    uint32_t c = w * cpw;
    if (c >= nchunks) return;
    const uint32_t c_end = (c + cpw < nchunks) ? c + cpw : nchunks;
    const int lane = threadIdx.x & 63;
    // last item with chunk_off[it] <= c  (chunk_off has nitems+1 entries, non-decreasing)
    uint32_t lo = 0, hi = nitems;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunk_off[mid] <= c) lo = mid; else hi = mid;
    }
    if (ab) return;   // speculated sizes of this call did not hold (run_impl)
    for (; c < c_end; ++c) {
        while (lo + 1 < nitems && chunk_off[lo + 1] <= c) ++lo;   // (items without chunks are stepped over)
        const BandItem& T = items[lo];
        const Rec& E = erec[T.el];
        const uint32_t local = (c - chunk_off[lo]) * 64u + lane;
        const uint32_t bx = T.dim[0], by = T.dim[1], bz = T.dim[2];
        if (local >= bx * by * bz) continue;
        const int li = local % bx, lj = (local / bx) % by, lk = local / (bx * by);
        double x[3];
        x[0] = grid_coord(g, 0, T.lo[0] + li);
        x[1] = grid_coord(g, 1, T.lo[1] + lj);
        x[2] = grid_coord(g, 2, slab_global_k(sl, T.lo[2] + lk));   // T.lo[2] is a local plane
        const int ci = (int)cell_of(g, 0, x[0]), cj = (int)cell_of(g, 1, x[1]), ck = (int)cell_of(g, 2, x[2]);
        const bool in = ci >= T.imin[0] && ci <= T.imax[0] && cj >= T.imin[1] && cj <= T.imax[1] &&
                        ck >= T.imin[2] && ck <= T.imax[2];
        if (!in) continue;
        double xp[3];
        const double d = iso_candidate(E, rho_t, x, xp);
        const TileBox tb = tile_box(T.lo, T.dim);
        const size_t slot = tile_slot(T.store_off, tb, T.lo[0] + li, T.lo[1] + lj, T.lo[2] + lk);
        res[slot] = d;
        if (res_xp) {
            res_xp[3 * slot] = xp[0];
            res_xp[3 * slot + 1] = xp[1];
            res_xp[3 * slot + 2] = xp[2];
        }
    }
}

// v -> (li, lj, lk) of a box with row length bx and plane size bxy.  Float reciprocals (exact to +-1,
// corrected) instead of two integer divisions when the box is small enough for exact float indices.
struct BoxDecode {
    uint32_t bx, bxy;
    float rbx, rbxy;
    bool small;
};
__device__ __forceinline__ BoxDecode box_decode_make(uint32_t bx, uint32_t by, uint32_t bz)
{
    BoxDecode d;
    d.bx = bx; d.bxy = bx * by;
    d.rbx = 1.0f / (float)bx; d.rbxy = 1.0f / (float)d.bxy;
    d.small = (uint64_t)d.bxy * bz < (1u << 22);
    return d;
}
__device__ __forceinline__ void box_decode(const BoxDecode& d, uint32_t v, uint32_t& li, uint32_t& lj, uint32_t& lk)
{
    if (d.small) {
        uint32_t q = (uint32_t)((float)v * d.rbxy);
        int32_t rem = (int32_t)(v - q * d.bxy);
        if (rem < 0) { q -= 1; rem += (int32_t)d.bxy; }
        else if (rem >= (int32_t)d.bxy) { q += 1; rem -= (int32_t)d.bxy; }
        lk = q;
        uint32_t q2 = (uint32_t)((float)rem * d.rbx);
        int32_t r2 = rem - (int32_t)(q2 * d.bx);
        if (r2 < 0) { q2 -= 1; r2 += (int32_t)d.bx; }
        else if (r2 >= (int32_t)d.bx) { q2 += 1; r2 -= (int32_t)d.bx; }
        lj = q2;
        li = (uint32_t)r2;
    } else {
        lk = v / d.bxy;
        const uint32_t rem = v - lk * d.bxy;
        lj = rem / d.bx;
        li = rem - lj * d.bx;
    }
}

// HEX8 iso-surface projection with lane refill (DESIGN.md "iso_project"): every lane runs the SQP state
// machine of r2s_device_math.hpp (IsoLane) on its own voxel and, when it has finished, takes the next
// voxel, so that slowly converging voxels, extra active-set steps and line-search trials of one lane do not
// stall the other 63.
#ifndef R2S_ISO_LB2
#define R2S_ISO_LB2 , 3   // <= 168 VGPRs: 3 waves/SIMD measured 3-7 % faster than 2
#endif
#ifndef R2S_ISO_REFILL_MIN
#define R2S_ISO_REFILL_MIN 16   // finished lanes wait until this many can be finalised + refilled together
#endif
// ------------------------------------------------------------------------------------
// tile binning.  A tile is a 4x4x4 block of VOXELS (lattice indices); a voxel's cell
// index (Grid.jl:58) is its lattice index or one less, so an item with cell box
// [imin,imax] can touch voxels with lattice index in [imin, imax+1].
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bool band_tile_range(const BandItem& T, const GridDev& g, const SlabInfo& s,
                                                int lo[3], int hi[3])
{
    const int nmax[3] = {g.nx - 1, g.ny - 1, g.nz - 1};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        int a = T.imin[ax], b = T.imax[ax];
        if (a > b || a > nmax[ax]) return false;
        b = (b >= nmax[ax]) ? nmax[ax] : b + 1;
        if (ax == 2) {
            int la, lb;
            if (!slab_local_range(s, a, b, la, lb)) return false;
            a = la;
            b = lb;
        }
        lo[ax] = a >> 2;
        hi[ax] = b >> 2;
    }
    return true;
}

// lattice range that can contain points of the element AABB (conservative by one)
__device__ __forceinline__ bool sign_tile_range(const ElemRec& E, const GridDev& g, const SlabInfo& s,
                                                int lo[3], int hi[3])
{
    const int nmax[3] = {g.nx - 1, g.ny - 1, g.nz - 1};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        double fa = floor((E.mn[ax] - g.amin[ax]) / g.cell) - 1.0;
        double fb = ceil((E.mx[ax] - g.amin[ax]) / g.cell) + 1.0;
        if (!(fb >= 0.0) || !(fa <= (double)nmax[ax])) return false;
        int a = (fa < 0.0) ? 0 : (int)fa;
        int b = (fb > (double)nmax[ax]) ? nmax[ax] : (int)fb;
        if (ax == 2) {
            int la, lb;
            if (!slab_local_range(s, a, b, la, lb)) return false;
            a = la;
            b = lb;
        }
        lo[ax] = a >> 2;
        hi[ax] = b >> 2;
    }
    return true;
}

// TET4: voxels whose bin index (SignDetection.jl:258-268) lies in [blo,bhi]; the bin index of
// lattice point i is i or i+1 (1-based), so the lattice range is [blo-1, bhi]
__device__ __forceinline__ bool sign_tile_range(const TetRec& E, const GridDev& g, const SlabInfo& s,
                                                int lo[3], int hi[3])
{
    const int nmax[3] = {g.nx - 1, g.ny - 1, g.nz - 1};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        if (E.blo[ax] > E.bhi[ax]) return false;
        int a = E.blo[ax] - 2, b = E.bhi[ax];
        if (a < 0) a = 0;
        if (b > nmax[ax]) b = nmax[ax];
        if (a > b) return false;
        if (ax == 2) {
            int la, lb;
            if (!slab_local_range(s, a, b, la, lb)) return false;
            a = la;
            b = lb;
        }
        lo[ax] = a >> 2;
        hi[ax] = b >> 2;
    }
    return true;
}

// the binning kernels spread the tile box of one item / element over a small lane group (the boxes hold
// ~50-100 tiles; one thread per box left most of the GPU waiting on a serial chain of atomics)
#define BIN_LANES 8
__device__ __forceinline__ uint32_t tile_of(const int lo[3], const int hi[3], uint32_t q, const SlabInfo& s)
{
    const uint32_t n0 = (uint32_t)(hi[0] - lo[0] + 1), n1 = (uint32_t)(hi[1] - lo[1] + 1);
    const uint32_t tz = q / (n0 * n1), r = q - tz * (n0 * n1), ty = r / n0, tx = r - ty * n0;
    return ((uint32_t)(lo[2] + (int)tz) * s.nty + (uint32_t)(lo[1] + (int)ty)) * s.ntx + (uint32_t)(lo[0] + (int)tx);
}
__device__ __forceinline__ uint32_t tile_count(const int lo[3], const int hi[3])
{
    return (uint32_t)(hi[0] - lo[0] + 1) * (uint32_t)(hi[1] - lo[1] + 1) * (uint32_t)(hi[2] - lo[2] + 1);
}

template <bool FILL>
__global__ void band_bin_kernel(const BandItem* __restrict__ items, uint32_t nitems, GridDev g, SlabInfo s,
                                uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
                                uint32_t* __restrict__ entries, uint8_t* __restrict__ tri, const uint32_t* __restrict__ abort_flag)
{
    if (*abort_flag) return;   // speculated sizes of this call did not hold (run_impl)
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t it = gid / BIN_LANES, sub = gid % BIN_LANES;
    if (it >= nitems) return;
    int lo[3], hi[3];
    if (!band_tile_range(items[it], g, s, lo, hi)) return;
    const uint32_t n = tile_count(lo, hi);
    for (uint32_t q = sub; q < n; q += BIN_LANES) {
        const uint32_t t = tile_of(lo, hi, q, s);
        const uint32_t pos = atomicAdd(&cnt[t], 1u);
        if (FILL) entries[off[t] + pos] = it;
        else if (items[it].kind != 0) tri[t] = 1;   // the tile's gather needs the triangle code
    }
}

// TET4: is every voxel of the (local) tile beyond one face plane of the element by the margin of TetRec::fo?  Then
// is_point_in_tetrahedron rejects all 64 of them and the element need not be in the tile's list (its bounding box
// holds 3-6 times the tiles the element itself touches).  HEX8 lists are tested by the inverse maps instead.
__device__ __forceinline__ bool tile_clear_of(const ElemRec&, const GridDev&, const SlabInfo&, int, int, int) { return false; }
__device__ __forceinline__ bool tile_clear_of(const TetRec& E, const GridDev& g, const SlabInfo& s, int tx, int ty, int tz)
{
    const double half = 1.5 * g.cell;
    const double c[3] = {grid_coord(g, 0, 4 * tx) + half, grid_coord(g, 1, 4 * ty) + half,
                         grid_coord(g, 2, slab_global_k(s, 4 * tz)) + half};
    bool clear = false;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const double t = E.fn[f][0] * c[0] + E.fn[f][1] * c[1] + E.fn[f][2] * c[2];
        const double r = (fabs(E.fn[f][0]) + fabs(E.fn[f][1]) + fabs(E.fn[f][2])) * half;
        clear = clear || (t - r > E.fo[f]);
    }
    return clear;
}
__device__ __forceinline__ void tile_xyz(const int lo[3], const int hi[3], uint32_t q, int& tx, int& ty, int& tz)
{
    const uint32_t n0 = (uint32_t)(hi[0] - lo[0] + 1), n1 = (uint32_t)(hi[1] - lo[1] + 1);
    const uint32_t z = q / (n0 * n1), r = q - z * (n0 * n1), y = r / n0;
    tx = lo[0] + (int)(r - y * n0);
    ty = lo[1] + (int)y;
    tz = lo[2] + (int)z;
}

// a voxel is only examined when some candidate reaches rho_t (SignDetection.jl:36): tiles whose lists
// would hold no such element keep sign = -1 without any work ("hot" = the others).  Pass 1 marks them.
template <class Rec>
__global__ void sign_hot_kernel(const Rec* __restrict__ erec, uint32_t nel, GridDev g, SlabInfo s, double rho_t,
                                uint8_t* __restrict__ hot)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t el = gid / BIN_LANES, sub = gid % BIN_LANES;
    if (el >= nel) return;
    if (erec[el].rmax < rho_t) return;
    int lo[3], hi[3];
    if (!sign_tile_range(erec[el], g, s, lo, hi)) return;
    const uint32_t n = tile_count(lo, hi);
    for (uint32_t q = sub; q < n; q += BIN_LANES) {
        int tx, ty, tz;
        tile_xyz(lo, hi, q, tx, ty, tz);
        if (tile_clear_of(erec[el], g, s, tx, ty, tz)) continue;
        hot[((uint32_t)tz * s.nty + (uint32_t)ty) * s.ntx + (uint32_t)tx] = 1;
    }
}

// candidate lists of the hot tiles only (count pass, then fill pass)
template <class Rec, bool FILL>
__global__ void sign_bin_kernel(const Rec* __restrict__ erec, uint32_t nel, GridDev g, SlabInfo s,
                                uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
                                uint32_t* __restrict__ entries, const uint8_t* __restrict__ hot,
                                const uint32_t* __restrict__ abort_flag, double rmax_needed)
{
    if (FILL && *abort_flag) return;   // speculated sizes of this call did not hold (run_impl)
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t el = gid / BIN_LANES, sub = gid % BIN_LANES;
    if (el >= nel) return;
    // TET4: only a hit with rho >= rho_t counts (SignDetection.jl:143-146) and rho <= max nodal density of the
    // element (validated shape functions are >= 0 and sum to <= 1), so elements below rho_t are no candidates.
    // (HEX8 passes -inf: there every candidate takes part in the cmax test, SignDetection.jl:36)
    if (erec[el].rmax < rmax_needed) return;
    int lo[3], hi[3];
    if (!sign_tile_range(erec[el], g, s, lo, hi)) return;
    const uint32_t n = tile_count(lo, hi);
    for (uint32_t q = sub; q < n; q += BIN_LANES) {
        int tx, ty, tz;
        tile_xyz(lo, hi, q, tx, ty, tz);
        const uint32_t t = ((uint32_t)tz * s.nty + (uint32_t)ty) * s.ntx + (uint32_t)tx;
        if (!hot[t] || tile_clear_of(erec[el], g, s, tx, ty, tz)) continue;
        const uint32_t pos = atomicAdd(&cnt[t], 1u);
        if (FILL) entries[off[t] + pos] = el;
    }
}

// Persistent kernel: the element record of a lane sits in LDS (a few slots per wavefront), so the lanes
// of one wavefront may work on voxels of DIFFERENT items.  Nothing drains at item boundaries any more, and
// the work is handed out dynamically (atomic counter over small chunk groups), which also removes the
// tail that a few slowly converging elements used to cause.
struct IsoElemLds {
    double C[8][3];
    double Cr[8];
    double X[8][3];
    double rtol;   // 1e-14 max(|rho_t|, max |nodal density|): rounding residue of the density field (see the oracle)
    int32_t el;
};

struct IsoElemCoef {   // what the solvers read of an element: the monomial coefficients of the two trilinear maps
    double C[8][3];
    double Cr[8];
};

// a pair the fast lane machine hands over to the complete solver (iso_straggler_kernel)
struct alignas(8) IsoStraggler {
    double x[3];
    uint64_t slot;
    uint32_t el;
    int32_t it;       // state at the start of the iteration the fast path gave up in (0, 0, 2, 0, 0 = from the start)
    double xi[3], mu, Delta;
    int32_t pat;
    int32_t pad;
};
#define R2S_ISO_UNSOLVED (-1.0)   // result slot of a handed-over pair until the complete solver has been there
#define R2S_ISO_SLOTS 4
#ifndef R2S_ISO_QP2_MIN
#define R2S_ISO_QP2_MIN 24
#endif
#ifdef R2S_ISO_WAVE_END   // diagnostic build only (tools/iso_phase_stats.py): when the wavefronts of the kernel leave
__device__ unsigned long long g_iso_wave_end[4096];   // wall_clock64 at the exit of every wavefront; [4094]: a start
__device__ unsigned long long g_iso_wave_info[4096 * 4];   // per wavefront: time the work ran dry for it, last item, trips and busy lanes at that time
extern "C" int r2s_debug_iso_wave_end(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_iso_wave_end), sizeof(unsigned long long) * 4096) != hipSuccess;
}
extern "C" int r2s_debug_iso_wave_info(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_iso_wave_info), sizeof(unsigned long long) * 4096 * 4) != hipSuccess;
}
#endif
#ifdef R2S_ISO_STATS   // diagnostic build only (tools/iso_phase_stats.py): visits and active lanes per phase
__device__ unsigned long long g_iso_stats[56];   // [0..15] ISO_STAT pairs, [16..31] histogram of SQP iterations / 4,
                                                 // [32..47] of log2(trips of a pair), [48..51] sums over pairs with >= 128 trips
#define ISO_STAT(p, cond)                                                                  \
    {                                                                                      \
        const uint64_t m__ = __ballot(cond);                                               \
        if (m__ && lane == 0) {                                                            \
            atomicAdd(&g_iso_stats[2 * (p)], 1ull);                                        \
            atomicAdd(&g_iso_stats[2 * (p) + 1], (unsigned long long)__popcll(m__));       \
        }                                                                                  \
    }
__device__ unsigned long long g_iso_long[8 * 64];   // [0] count; 8 words per long-running pair
extern "C" int r2s_debug_iso_long(unsigned long long* out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_iso_long), sizeof(unsigned long long) * 8 * 64) != hipSuccess) return 1;
    unsigned long long z[8 * 64] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_iso_long), z, sizeof z) != hipSuccess;
}
extern "C" int r2s_debug_iso_stats(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_iso_stats), sizeof(unsigned long long) * 56) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[56] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_iso_stats), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#else
#define ISO_STAT(p, cond)
#endif
template <int WPS>
__global__ void __launch_bounds__(64, WPS) iso_project_hex_pl_kernel(
    const BandItem* __restrict__ items, uint32_t nitems, const uint32_t* __restrict__ chunk_off, uint32_t nchunks,
    uint32_t group, const ElemRec* __restrict__ erec, GridDev g, SlabInfo sl, double rho_t, double* __restrict__ res,
    double* __restrict__ res_xp, uint32_t* __restrict__ counter, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ abort_flag,
    IsoStraggler* __restrict__ strag, uint32_t strag_cap, uint32_t* __restrict__ strag_cnt /* entries */,
    uint32_t* __restrict__ strag_ovf)
{
    if (*abort_flag) return;   // speculated sizes of this call did not hold (run_impl)
    __shared__ IsoElemLds slots[R2S_ISO_SLOTS];
    const uint32_t lane = threadIdx.x;
#ifdef R2S_ISO_WAVE_END
    if (lane == 0 && blockIdx.x == 1) g_iso_wave_end[4094] = wall_clock64();   // (about) the start of the kernel
#endif
    // wave-uniform bookkeeping
    uint32_t c = 0, c_end = 0;        // rest of the fetched chunk group
    uint32_t it = 0, co = 0, cn = 0;  // newest item and its chunk range
    bool have_item = false, exhausted = false;
    uint32_t next = 0, v_end = 0, vol = 0;   // rest of the current segment (box voxels of item `it`)
    int cur_slot = 0, lo0 = 0, lo1 = 0, lo2 = 0;
    uint32_t st_off = 0;
    TileBox tb = {{0, 0, 0}, {0, 0, 0}};
    BoxDecode dec = box_decode_make(1u, 1u, 1u);
    // per lane
    IsoLane s;
    s.phase = ISO_IDLE;
#ifdef R2S_ISO_STATS
    int dbg_trips = 0, dbg_qp = 0, dbg_ls = 0, dbg_el = 0;
#endif
    size_t my = 0;
    int eslot = 0;

    for (;;) {
        const uint64_t m_done = __ballot(s.phase == ISO_DONE || s.phase == ISO_BAIL);
        const uint64_t m_busy = __ballot(s.phase != ISO_DONE && s.phase != ISO_BAIL && s.phase != ISO_IDLE);
        if (m_busy == 0 || __popcll(m_done) >= R2S_ISO_REFILL_MIN) {
            ISO_STAT(6, s.phase == ISO_DONE)
            ISO_STAT(3, s.phase == ISO_BAIL)
            if (s.phase == ISO_BAIL) {
                // not a plain Newton-SQP run: the complete solver takes the pair from the start (iso_straggler_kernel);
                // one atomic per wavefront reserves the list entries
                const uint64_t m_bail = __ballot(true);
                const uint32_t nb = (uint32_t)__popcll(m_bail);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_bail >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_bail, 0u));
                uint32_t base = 0;
                if (rank == 0) base = atomicAdd(&strag_cnt[0], nb);
                base = __builtin_amdgcn_readfirstlane(base);   // (the first active lane is the one with rank 0)
                const uint32_t idx = base + rank;
                if (idx < strag_cap) {
                    IsoStraggler e;
                    e.x[0] = s.x[0]; e.x[1] = s.x[1]; e.x[2] = s.x[2];
                    e.slot = (uint64_t)my;
                    e.el = (uint32_t)slots[eslot].el;
                    const bool resume = !s.seen;
                    e.it = resume ? s.it : 0;
                    e.xi[0] = resume ? s.xi[0] : 0.0; e.xi[1] = resume ? s.xi[1] : 0.0; e.xi[2] = resume ? s.xi[2] : 0.0;
                    e.mu = resume ? s.mu : 0.0;
                    e.Delta = resume ? s.Delta : 2.0;
                    e.pat = resume ? s.pat : 0;
                    e.pad = 0;
                    strag[idx] = e;
                } else {
                    *strag_ovf = 1u;   // list full: iso_sweep_kernel finds the pair by its result slot
                }
                res[my] = R2S_ISO_UNSOLVED;
                s.phase = ISO_IDLE;
            }
            if (s.phase == ISO_DONE) {
#ifdef R2S_ISO_STATS
                atomicAdd(&g_iso_stats[16 + (s.it / 4 > 15 ? 15 : s.it / 4)], 1ull);
                atomicAdd(&g_iso_stats[32 + (31 - __clz(dbg_trips | 1))], 1ull);
                if (dbg_trips >= 128) {   // who are they: element, voxel coordinates, iterations, trips of up to 63 of them
                    const unsigned long long k = atomicAdd(&g_iso_long[0], 1ull);
                    if (k < 63) {
                        g_iso_long[8 * (k + 1) + 0] = (unsigned long long)dbg_el;
                        g_iso_long[8 * (k + 1) + 1] = (unsigned long long)__double_as_longlong(s.x[0]);
                        g_iso_long[8 * (k + 1) + 2] = (unsigned long long)__double_as_longlong(s.x[1]);
                        g_iso_long[8 * (k + 1) + 3] = (unsigned long long)__double_as_longlong(s.x[2]);
                        g_iso_long[8 * (k + 1) + 4] = (unsigned long long)s.it;
                        g_iso_long[8 * (k + 1) + 5] = (unsigned long long)dbg_trips;
                        g_iso_long[8 * (k + 1) + 6] = (unsigned long long)dbg_qp;
                        g_iso_long[8 * (k + 1) + 7] = (unsigned long long)dbg_ls;
                    }
                }
                if (dbg_trips >= 128) {   // the pairs behind the kernel's tail: where do their trips go?
                    atomicAdd(&g_iso_stats[48], (unsigned long long)dbg_trips);
                    atomicAdd(&g_iso_stats[49], (unsigned long long)dbg_qp);
                    atomicAdd(&g_iso_stats[50], (unsigned long long)dbg_ls);
                    atomicAdd(&g_iso_stats[51], (unsigned long long)s.it);
                }
#endif
                const IsoElemLds& E = slots[eslot];
                double N[8], xp[3];
                hex8_shape(s.xi, N);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) t = fma(E.X[k][i], N[k], t);
                    xp[i] = t;
                }
                res[my] = norm3(s.x[0] - xp[0], s.x[1] - xp[1], s.x[2] - xp[2]);
                if (res_xp) {
                    res_xp[3 * my] = xp[0];
                    res_xp[3 * my + 1] = xp[1];
                    res_xp[3 * my + 2] = xp[2];
                }
                s.phase = ISO_IDLE;
            }
            // hand out voxels; open new segments / items / chunk groups as needed
            for (;;) {
                const uint64_t m_idle = __ballot(s.phase == ISO_IDLE);
                if (m_idle == 0) break;
                if (next >= v_end) {
                    if (exhausted) break;
                    if (c >= c_end) {
                        uint32_t cc = 0;
                        if (lane == 0) cc = atomicAdd(counter, group);
                        c = __builtin_amdgcn_readfirstlane(cc);
                        if (c >= nchunks) {
#ifdef R2S_ISO_WAVE_END
                            if (!exhausted && lane == 0 && blockIdx.x < 4094) {
                                g_iso_wave_info[4 * blockIdx.x] = wall_clock64();
                                g_iso_wave_info[4 * blockIdx.x + 1] = it;
                                g_iso_wave_info[4 * blockIdx.x + 2] = (unsigned long long)__popcll(__ballot(s.phase != ISO_IDLE && s.phase != ISO_DONE));
                            }
#endif
                            exhausted = true;
                            break;
                        }
                        c_end = (c + group < nchunks) ? c + group : nchunks;
                    }
                    if (!(have_item && c >= co && c < cn)) {
                        // another item: its record needs an LDS slot no lane is working from
                        int fs = -1;
#pragma unroll
                        for (int q = 0; q < R2S_ISO_SLOTS; ++q)
                            if (fs < 0 && __ballot(s.phase != ISO_IDLE && eslot == q) == 0) fs = q;
                        if (fs < 0) break;   // all slots busy: the idle lanes wait
                        uint32_t lo = 0, hi = nitems;   // last item with chunk_off[it] <= c
                        while (hi - lo > 1) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (chunk_off[mid] <= c) lo = mid; else hi = mid;
                        }
                        it = perm[lo]; co = chunk_off[lo]; cn = chunk_off[lo + 1];   // chunk_off is in work order
                        have_item = true;
                        const BandItem& T = items[it];
                        const double* src = reinterpret_cast<const double*>(&erec[T.el]);
                        double* dst = reinterpret_cast<double*>(&slots[fs]);
                        if (lane < 24) dst[lane] = src[offsetof(ElemRec, C) / 8 + lane];
                        else if (lane < 32) dst[lane] = src[offsetof(ElemRec, Cr) / 8 + (lane - 24)];
                        else if (lane < 56) dst[lane] = src[offsetof(ElemRec, X) / 8 + (lane - 32)];
                        else if (lane == 56) {
                            const ElemRec& R = erec[T.el];
                            slots[fs].rtol = fmax(fabs(rho_t), fmax(fabs(R.rmax), fabs(R.rmin))) * 1e-14;
                            slots[fs].el = T.el;
                        }
                        __syncthreads();
                        cur_slot = fs;
                        dec = box_decode_make((uint32_t)T.dim[0], (uint32_t)T.dim[1], (uint32_t)T.dim[2]);
                        vol = dec.bxy * (uint32_t)T.dim[2];
                        lo0 = T.lo[0]; lo1 = T.lo[1]; lo2 = T.lo[2];
                        st_off = T.store_off;
                        tb = tile_box(T.lo, T.dim);
                    }
                    const uint32_t seg_end = (cn < c_end) ? cn : c_end;
                    next = (c - co) * 64u;
                    v_end = ((seg_end - co) * 64u < vol) ? (seg_end - co) * 64u : vol;
                    c = seg_end;
                    if (next >= v_end) continue;
                }
                if (s.phase == ISO_IDLE) {
                    const uint32_t v = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_idle >> 32),
                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)m_idle, 0u));
                    if (v < v_end) {
                        uint32_t li, lj, lk;
                        box_decode(dec, v, li, lj, lk);
                        const int i = lo0 + (int)li, j = lo1 + (int)lj, kl = lo2 + (int)lk;   // kl: local plane
                        double x[3];
                        x[0] = grid_coord(g, 0, i);
                        x[1] = grid_coord(g, 1, j);
                        x[2] = grid_coord(g, 2, slab_global_k(sl, kl));
                        iso_lane_start(s, x);
#ifdef R2S_ISO_STATS
                        dbg_trips = 0; dbg_qp = 0; dbg_ls = 0; dbg_el = items[it].el;
#endif
                        my = tile_slot(st_off, tb, i, j, kl);
                        eslot = cur_slot;
                    }
                }
                next += (uint32_t)__popcll(m_idle);
            }
            if (exhausted && __ballot(s.phase != ISO_IDLE) == 0) {
#ifdef R2S_ISO_WAVE_END
                if (lane == 0 && blockIdx.x < 4094) g_iso_wave_end[blockIdx.x] = wall_clock64();
#endif
                break;
            }
        }
        // ---- one visit of each phase ----
        {
            const IsoElemLds& E = slots[eslot];
            ISO_STAT(0, s.phase == ISO_EVAL)
            if (s.phase == ISO_EVAL) iso_lane_eval(E, rho_t, E.rtol, s);
            ISO_STAT(1, s.phase == ISO_QP)
            if (s.phase == ISO_QP) iso_lane_qp(s);
            // lanes whose active set changed need another pattern: worth a second visit in this trip only when
            // many of them do (an unconditional second visit costs what it saves)
            if (__popcll(__ballot(s.phase == ISO_QP)) >= R2S_ISO_QP2_MIN) {
                ISO_STAT(7, s.phase == ISO_QP)
                if (s.phase == ISO_QP) iso_lane_qp(s);
            }
            ISO_STAT(2, s.phase == ISO_FINISH)
            if (s.phase == ISO_FINISH) iso_lane_finish(E, rho_t, E.rtol, s);
            ISO_STAT(5, true)
#ifdef R2S_ISO_STATS
            if (s.phase != ISO_IDLE) dbg_trips += 1;
#endif
        }
    }
}

#ifdef R2S_STRAG_DIAG   // diagnostic build only (tools/strag_diag.py)
__device__ unsigned long long g_strag_diag[3 * 16384];
extern "C" int r2s_debug_strag_diag(unsigned long long* out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_strag_diag), sizeof(unsigned long long) * 3 * 16384) != hipSuccess) return 1;
    static unsigned long long z[3 * 16384];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_strag_diag), z, sizeof z) != hipSuccess;
}
__device__ unsigned long long g_strag_hist[256];   // [0,128): SQP iterations of a pair at its end; [128,256): its trips
extern "C" int r2s_debug_strag_hist(unsigned long long* out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_strag_hist), sizeof(unsigned long long) * 256) != hipSuccess) return 1;
    static unsigned long long z[256];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_strag_hist), z, sizeof z) != hipSuccess;
}
#endif
// The pairs the fast path handed over, compacted.  Persistent one-wave workgroups with lane refill run the COMPLETE
// solver as a lane machine (iso_full_* in r2s_device_math.hpp = iso_project_full = the oracle's iteration, operation for
// operation): a lane takes the next list entry when its pair is done, and a trip costs what the phases of its lanes cost.
// (Round 3 started with one lane per pair running iso_project_full straight through: a wavefront then lasts as long as
// its slowest lane and executes the union of its lanes' branches in every iteration - 14 us per wavefront-iteration,
// 0.45-0.6 ms for 3 % of the fast kernel's instructions.)  Every lane has its own element: the 32 coefficients of the
// solver live in a padded LDS column per lane (33 doubles: two-way bank conflicts at most).
struct IsoCoefPad {
    double C[8][3];
    double Cr[8];
    double pad;
};
#ifndef R2S_STRAG_REFILL_MIN
#define R2S_STRAG_REFILL_MIN 8
#endif
#ifndef R2S_ENUM_COOP_MAX
#define R2S_ENUM_COOP_MAX 6   // lanes in the pattern search up to which the wavefront serves them one by one
#endif
__global__ void __launch_bounds__(64, 2) iso_straggler_kernel(const IsoStraggler* __restrict__ strag, uint32_t strag_cap,
                                                           const uint32_t* __restrict__ strag_cnt, const ElemRec* __restrict__ erec,
                                                           double rho_t, double* __restrict__ res, double* __restrict__ res_xp,
                                                           const uint32_t* __restrict__ abort_flag, uint32_t* __restrict__ head,
                                                           uint32_t* __restrict__ n_fail)
{
    if (*abort_flag) return;
    __shared__ IsoCoefPad coef[64];
    const uint32_t n = min(strag_cnt[0], strag_cap);
    const uint32_t lane = threadIdx.x;
    IsoFullLane s;
    s.phase = FS_IDLE;
    uint32_t my_el = 0;
    uint64_t my_slot = 0;
    double rtol = 0.0;
    bool exhausted = false;
    const uint32_t quota = min(64u, max(4u, (n + gridDim.x - 1u) / gridDim.x));
#ifdef R2S_STRAG_DIAG
    const unsigned long long t0 = wall_clock64();
    unsigned long long trips = 0, pairs = 0;
    int my_trips = 0;
#endif
    for (;;) {
        const uint64_t m_done = __ballot(s.phase == FS_DONE);
        const uint64_t m_busy = __ballot(s.phase != FS_DONE && s.phase != FS_IDLE);
        // (quota: lanes a wavefront keeps busy - all 64 on a long list; a short list, e.g. one rank's share of 8, is spread
        //  over all wavefronts instead: a trip costs what the phases of its lanes cost, and the kernel lasts as long as
        //  the trips of its slowest pair)
        const uint32_t refill_min = min((uint32_t)R2S_STRAG_REFILL_MIN, max(1u, quota / 4u));
        if (m_busy == 0 || __popcll(m_done) >= R2S_STRAG_REFILL_MIN ||
            (!exhausted && (uint32_t)__popcll(m_busy) + refill_min <= quota &&
             (uint32_t)__popcll(__ballot(s.phase == FS_IDLE)) + (quota < 64u ? (uint32_t)__popcll(m_done) : 0u) >= refill_min)) {
            {   // runs that ended without a KKT point (the nearest on-surface iterate is used, as the reference uses
                // whatever NLopt returns, ComputeCoordsOnIso.jl:79-86): counted for r2s_stats
                const uint64_t mf = __ballot(s.phase == FS_DONE && s.it > R2S_ISO_MAXIT);
                if (mf && lane == (uint32_t)__builtin_ctzll(mf)) atomicAdd(n_fail, (uint32_t)__popcll(mf));
            }
            if (s.phase == FS_DONE) {
                const ElemRec& E = erec[my_el];
                double N[8], xp[3];
                hex8_shape(s.xi, N);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) t = fma(E.X[k][q], N[k], t);
                    xp[q] = t;
                }
                res[my_slot] = norm3(s.x[0] - xp[0], s.x[1] - xp[1], s.x[2] - xp[2]);
                if (res_xp) {
                    res_xp[3 * my_slot] = xp[0];
                    res_xp[3 * my_slot + 1] = xp[1];
                    res_xp[3 * my_slot + 2] = xp[2];
                }
#ifdef R2S_STRAG_DIAG
                if (R2S_STRAG_DIAG >= 2) {   // (same-address atomics: the wavefront times of such a build mean nothing)
                    atomicAdd(&g_strag_hist[min(max(s.it, 0), 127)], 1ull);
                    atomicAdd(&g_strag_hist[128 + min(my_trips, 127)], 1ull);
                }
                my_trips = 0;
#endif
                s.phase = FS_IDLE;
            }
            uint64_t m_idle = __ballot(s.phase == FS_IDLE);
            {   // at most `quota` busy lanes: the lowest idle lanes take the new entries
                const uint32_t busy_now = 64u - (uint32_t)__popcll(m_idle);
                uint32_t room = quota > busy_now ? quota - busy_now : 0u;
                uint64_t keep = 0;
                for (uint64_t m = m_idle; m && room; m &= m - 1, --room) keep |= m & (~m + 1);
                m_idle = keep;
            }
            if (m_idle && !exhausted) {
                const uint32_t nid = (uint32_t)__popcll(m_idle);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(head, nid);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base + nid >= n) exhausted = true;
                const uint32_t i = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_idle, 0u));
                if (s.phase == FS_IDLE && ((m_idle >> lane) & 1ull) && i < n) {
                    const IsoStraggler e = strag[i];
                    const ElemRec& E = erec[e.el];
                    IsoCoefPad& K = coef[lane];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        K.C[k][0] = E.C[k][0]; K.C[k][1] = E.C[k][1]; K.C[k][2] = E.C[k][2];
                        K.Cr[k] = E.Cr[k];
                    }
                    rtol = fmax(fabs(rho_t), fmax(fabs(E.rmax), fabs(E.rmin))) * 1e-14;
                    my_el = e.el;
                    my_slot = e.slot;
                    iso_full_start(s, e.x, e.xi, e.mu, e.Delta, e.pat, e.it);
#ifdef R2S_STRAG_DIAG
                    pairs += 1;
#endif
                }
            }
            if (exhausted && __ballot(s.phase != FS_IDLE) == 0) break;
        }
        // ---- one visit of each phase that some lane is in ----
        {
            const IsoCoefPad& E = coef[lane];
            // The short phases a pair may visit several times per iteration (active-set walk: <= 8 patterns; second
            // sigma: back to EVAL once; line search: <= 30 trials) are repeated until no lane is left in them, instead
            // of one visit per trip of ALL phases: a pair took 2.5 trips per SQP iteration, and the kernel lasts as
            // long as the trips of its slowest pairs (0.455 -> 0.41 ms behind the fast kernel)
            for (int rep = 0; rep < 2; ++rep) {
                if (s.phase == FS_EVAL) iso_full_eval(E, rho_t, rtol, s);
                while (__ballot(s.phase == FS_QP)) {
                    if (s.phase == FS_QP) iso_full_qp(s);
                }
                if (!__ballot(s.phase == FS_EVAL)) break;
            }
            {
                // exhaustive pattern search: one lane at a time by the whole wavefront while few lanes need it (the tail
                // of the kernel is a handful of pairs with dozens of non-convex iterations), lane by lane otherwise
                uint64_t m_enum = __ballot(s.phase == FS_ENUM);
                if (m_enum && __popcll(m_enum) <= R2S_ENUM_COOP_MAX) {
                    for (; m_enum; m_enum &= m_enum - 1) iso_full_enum_coop(s, (int)__builtin_ctzll(m_enum));
                } else if (s.phase == FS_ENUM) {
                    iso_full_enum(s);
                }
            }
            if (s.phase == FS_POST) iso_full_post(E, rho_t, rtol, s);
            while (__ballot(s.phase == FS_LS)) {
                if (s.phase == FS_LS) iso_full_ls(E, rho_t, s);
            }
            if (s.phase == FS_UPD) iso_full_upd(E, rho_t, rtol, s);
#ifdef R2S_STRAG_DIAG
            trips += 1;
            if (s.phase != FS_IDLE) my_trips += 1;
#endif
        }
    }
#ifdef R2S_STRAG_DIAG
    if (blockIdx.x < 16384) {
        atomicAdd(&g_strag_diag[3 * blockIdx.x + 2], pairs);
        if (lane == 0) { g_strag_diag[3 * blockIdx.x] = wall_clock64() - t0; g_strag_diag[3 * blockIdx.x + 1] = trips; }
    }
#endif
}

// Only when the straggler list overflowed (a mesh on which more than ~6 % of the pairs leave the fast path - tiny grids
// are sized so that it cannot happen): every pair of every item is visited item-major and the slots still marked
// unsolved are worked off in place.  Slow (a wavefront waits for its few unsolved lanes), rare, correct.
__global__ void __launch_bounds__(256) iso_sweep_kernel(const BandItem* __restrict__ items, uint32_t nitems,
                                                        const uint32_t* __restrict__ chunk_off, uint32_t nchunks,
                                                        const uint32_t* __restrict__ perm, const ElemRec* __restrict__ erec,
                                                        GridDev g, SlabInfo sl, double rho_t, double* __restrict__ res,
                                                        double* __restrict__ res_xp, const uint32_t* __restrict__ strag_cnt,
                                                        const uint32_t* __restrict__ abort_flag, uint32_t* __restrict__ n_fail)
{
    if (*abort_flag || *strag_cnt == 0u) return;   // (strag_cnt: the overflow flag)
    const uint32_t nw = (gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (uint32_t c = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6); c < nchunks; c += nw) {
        uint32_t lo = 0, hi = nitems;   // last work-order position with chunk_off[pos] <= c
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (chunk_off[mid] <= c) lo = mid; else hi = mid;
        }
        const BandItem& T = items[perm[lo]];
        const uint32_t local = (c - chunk_off[lo]) * 64u + lane;
        const uint32_t bx = T.dim[0], by = T.dim[1], bz = T.dim[2];
        if (local >= bx * by * bz) continue;
        const int li = local % bx, lj = (local / bx) % by, lk = local / (bx * by);
        const TileBox tb = tile_box(T.lo, T.dim);
        const size_t slot = tile_slot(T.store_off, tb, T.lo[0] + li, T.lo[1] + lj, T.lo[2] + lk);
        if (res[slot] != R2S_ISO_UNSOLVED) continue;
        double x[3], xi[3] = {0.0, 0.0, 0.0};
        x[0] = grid_coord(g, 0, T.lo[0] + li);
        x[1] = grid_coord(g, 1, T.lo[1] + lj);
        x[2] = grid_coord(g, 2, slab_global_k(sl, T.lo[2] + lk));
        const ElemRec& E = erec[T.el];
        IsoElemCoef K;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            K.C[k][0] = E.C[k][0]; K.C[k][1] = E.C[k][1]; K.C[k][2] = E.C[k][2];
            K.Cr[k] = E.Cr[k];
        }
        if (iso_project_full(K, fmax(fabs(E.rmax), fabs(E.rmin)), x, rho_t, xi) > R2S_ISO_MAXIT) atomicAdd(n_fail, 1u);
        double N[8], xp[3];
        hex8_shape(xi, N);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) t = fma(E.X[k][q], N[k], t);
            xp[q] = t;
        }
        res[slot] = norm3(x[0] - xp[0], x[1] - xp[1], x[2] - xp[2]);
        if (res_xp) {
            res_xp[3 * slot] = xp[0];
            res_xp[3 * slot + 1] = xp[1];
            res_xp[3 * slot + 2] = xp[2];
        }
    }
}

// ---- item-major inverse maps for Sign_Detection_HEX8 ----------------------------------------
// The reference walks, per grid point, the elements whose AABB holds the point and runs the Newton
// inverse map for each (SignDetection.jl:27-70).  Here the inverse maps are computed element-major
// (sign_project_kernel: element record in SGPRs, lanes = lattice points of the element's AABB box, dense
// lanes) into `sres`, and the ordered per-voxel state machine (sdf_tiles_kernel) only looks them up.
struct alignas(16) SignBox {
    int32_t lo[3];    // first lattice index per axis (Z: local plane of the slab)
    int32_t dim[3];   // 0 in any axis = no box (element irrelevant for the sign pass)
    double rmax;      // max nodal density (SignDetection.jl:33-36): all the ordered gather needs of an element
};

// lattice indices i with mn <= grid_coord(i) <= mx: exactly the reference's AABB comparisons
__device__ __forceinline__ bool coord_range(const GridDev& g, int ax, double mn, double mx, int nmax, int& a, int& b)
{
    const double fa = ceil((mn - g.amin[ax]) / g.cell), fb = floor((mx - g.amin[ax]) / g.cell);
    if (!(fa <= (double)nmax + 2.0) || !(fb >= -2.0)) return false;   // outside the grid (or NaN)
    a = (fa < 0.0) ? 0 : ((fa > (double)nmax) ? nmax : (int)fa);
    b = (fb > (double)nmax) ? nmax : ((fb < 0.0) ? 0 : (int)fb);
    while (a > 0 && grid_coord(g, ax, a - 1) >= mn) --a;
    while (a <= nmax && grid_coord(g, ax, a) < mn) ++a;
    while (b < nmax && grid_coord(g, ax, b + 1) <= mx) ++b;
    while (b >= 0 && grid_coord(g, ax, b) > mx) --b;
    return a <= b;
}

// one thread per element: relevant (its tile range holds a hot tile) -> lattice box + chunk count
__global__ void sign_box_kernel(const ElemRec* __restrict__ erec, uint32_t nel, GridDev g, SlabInfo s,
                                const uint8_t* __restrict__ hot, SignBox* __restrict__ sbox,
                                uint32_t* __restrict__ nchunks, uint32_t* __restrict__ nstore)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t el = gid / BIN_LANES, sub = gid % BIN_LANES;
    const bool live = el < nel;
    SignBox B;
    memset(&B, 0, sizeof B);
    const ElemRec& E = erec[live ? el : 0];
    int lo[3], hi[3];
    bool found = false;
    if (live && sign_tile_range(E, g, s, lo, hi)) {
        const uint32_t n = tile_count(lo, hi);
        for (uint32_t q = sub; q < n && !found; q += BIN_LANES) found = hot[tile_of(lo, hi, q, s)] != 0;
    }
    // any lane of the element's group
    const unsigned long long m = __ballot(found);
    const bool rel = ((m >> ((threadIdx.x & 63u) & ~(unsigned)(BIN_LANES - 1))) & ((1ull << BIN_LANES) - 1ull)) != 0;
    if (!live || sub != 0) return;
    if (rel) {
        const int nmax[3] = {g.nx - 1, g.ny - 1, g.nz - 1};
        uint64_t vol = 1;
        for (int ax = 0; ax < 3; ++ax) {
            int a = 0, b = -1;
            if (coord_range(g, ax, E.mn[ax], E.mx[ax], nmax[ax], a, b)) {
                if (ax == 2) {   // Z in LOCAL planes of the slab
                    int la, lb;
                    if (slab_local_range(s, a, b, la, lb)) { a = la; b = lb; }
                    else { a = 0; b = -1; }
                }
            } else { a = 0; b = -1; }
            B.lo[ax] = a;
            B.dim[ax] = (b >= a) ? (b - a + 1) : 0;
            vol *= (uint64_t)B.dim[ax];
        }
        if (vol == 0) B.dim[0] = B.dim[1] = B.dim[2] = 0;
        nchunks[el] = (uint32_t)((vol + 63) / 64);
        const TileBox tb = tile_box(B.lo, B.dim);
        nstore[el] = vol ? (uint32_t)(tb.td[0] * tb.td[1] * tb.td[2]) : 0u;
    }
    B.rmax = E.rmax;
    sbox[el] = B;
}

// one wavefront per `cpw` chunks (64 lattice points each) of the boxes.  64 points at a time are tested
// (hot tile, inside the element's bounding half-spaces - ElemRec::pn; the others cannot pass the
// max|xi| < 1.01 test); the survivors are compacted through a small LDS queue and the Newton inverse
// maps run on full wavefronts.  Stored per slot: +m when the interpolated density reaches rho_t, -m when
// it does not, +inf when the candidate has no effect.
__global__ void __launch_bounds__(64) sign_project_kernel(const SignBox* __restrict__ sbox, uint32_t nel,
                                                          const uint32_t* __restrict__ chunk_off, uint32_t nchunks,
                                                          const uint32_t* __restrict__ store_off,
                                                          uint32_t cpw, const ElemRec* __restrict__ erec, GridDev g,
                                                          SlabInfo sl, double rho_t, const uint8_t* __restrict__ hot,
                                                          double* __restrict__ res, const uint32_t* __restrict__ abort_flag)
{
    const uint32_t ab = *abort_flag;   // (tested after the search below: its loads do not wait for this one)
    __shared__ uint32_t queue[128];
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    uint32_t c = w * cpw;
    if (c >= nchunks) return;
    const uint32_t c_end = (c + cpw < nchunks) ? c + cpw : nchunks;
    uint32_t lo = 0, hi = nel;   // last element with chunk_off[el] <= c
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunk_off[mid] <= c) lo = mid; else hi = mid;
    }
    if (ab) return;   // speculated sizes of this call did not hold (run_impl)
    const uint32_t lane = threadIdx.x & 63;
    while (c < c_end) {
        lo = __builtin_amdgcn_readfirstlane(lo);
        const SignBox& B = sbox[lo];
        const ElemRec& E = erec[lo];
        const uint32_t co = chunk_off[lo], cn = chunk_off[lo + 1];
        const uint32_t seg_end = (cn < c_end) ? cn : c_end;
        const BoxDecode dec = box_decode_make((uint32_t)B.dim[0], (uint32_t)B.dim[1], (uint32_t)B.dim[2]);
        const uint32_t vol = dec.bxy * (uint32_t)B.dim[2];
        const uint32_t v_begin = (c - co) * 64u;
        const uint32_t v_end = ((seg_end - co) * 64u < vol) ? (seg_end - co) * 64u : vol;
        const int lo0 = B.lo[0], lo1 = B.lo[1], lo2 = B.lo[2];
        const uint32_t st_off = store_off[lo];
        const TileBox tb = tile_box(B.lo, B.dim);
        // (margin: sum N_k r_k of the reference is a rounded convex combination; with r_min strictly above rho_t by more
        //  than its rounding it is >= rho_t for every xi in the cube)
        const double margin = 1e-9 * fmax(1.0, fabs(rho_t));
        const double uniform_sign = (E.rmin > rho_t + margin) ? 1.0 : ((E.rmax < rho_t - margin) ? -1.0 : 0.0);
        uint32_t qn = 0;   // wave-uniform queue length
        for (uint32_t v0 = v_begin; v0 < v_end || qn > 0; v0 += 64u) {
            if (v0 < v_end) {
                const uint32_t v = v0 + lane;
                bool pass = false;
                if (v < v_end) {
                    uint32_t li, lj, lk;
                    box_decode(dec, v, li, lj, lk);
                    const int i = lo0 + (int)li, j = lo1 + (int)lj, kl = lo2 + (int)lk;
                    double x[3];
                    x[0] = grid_coord(g, 0, i);
                    x[1] = grid_coord(g, 1, j);
                    x[2] = grid_coord(g, 2, slab_global_k(sl, kl));
                    const uint32_t t = ((uint32_t)(kl >> 2) * sl.nty + (uint32_t)(j >> 2)) * sl.ntx + (uint32_t)(i >> 2);
                    pass = hot[t] && !hex8_outside(E, x);
                    if (!pass) res[tile_slot(st_off, tb, i, j, kl)] = INFINITY;
                    else if (uniform_sign != 0.0 && hex8_inner(E, x)) {
                        // inside an element whose nodal densities lie all above (or all below) rho_t: max|xi| <= 1 and the
                        // comparison rho(xi) >= rho_t comes out the same whatever xi is.  The state machine of the gather only
                        // needs "holds the point, counts as +1 (or leaves the sign alone), ends the walk": in a conforming mesh
                        // no other candidate holds a point this deep inside the element with a smaller max|xi|, so nothing
                        // after it can change the outcome and the order of what came before it is untouched.
                        res[tile_slot(st_off, tb, i, j, kl)] = uniform_sign > 0.0 ? 0.0 : -0.0;
                        pass = false;
                    }
                }
                const uint64_t m = __ballot(pass);
                if (pass) queue[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = v;
                qn += (uint32_t)__popcll(m);
                __syncthreads();
            }
            // a full wavefront of survivors (or the rest once the segment's points are exhausted)
            if (qn >= 64u || (v0 + 64u >= v_end && qn > 0)) {
                const uint32_t take = qn < 64u ? qn : 64u;
                const bool have = lane < take;
                const uint32_t v = queue[lane];
                const uint32_t spill = queue[64u + lane];
                __syncthreads();
                if (lane + 64u < qn) queue[lane] = spill;
                qn -= take;
                __syncthreads();
                if (have) {
                    uint32_t li, lj, lk;
                    box_decode(dec, v, li, lj, lk);
                    const int i = lo0 + (int)li, j = lo1 + (int)lj, kl = lo2 + (int)lk;
                    double x[3], xi[3], N[8];
                    x[0] = grid_coord(g, 0, i);
                    x[1] = grid_coord(g, 1, j);
                    x[2] = grid_coord(g, 2, slab_global_k(sl, kl));
                    inv_map_hex8(E, x, xi);
                    const double m = fmax(fabs(xi[0]), fmax(fabs(xi[1]), fabs(xi[2])));
                    double val = INFINITY;
                    if (m < 1.01) {
                        hex8_shape(xi, N);
                        double rho = 0.0;
#pragma unroll
                        for (int k = 0; k < 8; ++k) rho += N[k] * E.r[k];
                        val = (rho >= rho_t) ? m : -m;
                    }
                    res[tile_slot(st_off, tb, i, j, kl)] = val;
                }
            }
        }
        c = seg_end;
        if (c < c_end) {
            do { ++lo; } while (chunk_off[lo + 1] <= c);
        }
    }
}

// active tiles = tiles with a non-empty band list, or a sign list whose elements reach rho_t
#define AT_ITEMS 8   // tiles per thread
__global__ void __launch_bounds__(256) active_tiles_kernel(const uint32_t* __restrict__ band_cnt,
                                                           const uint32_t* __restrict__ sign_cnt,
                                                           const uint8_t* __restrict__ hot, uint32_t ntiles,
                                                           uint32_t* __restrict__ active_band,
                                                           uint32_t* __restrict__ active_sign,
                                                           uint32_t* __restrict__ active_any,
                                                           uint32_t* __restrict__ active_sonly,
                                                           const uint8_t* __restrict__ tri,
                                                           uint32_t* __restrict__ active_lean,
                                                           uint32_t* __restrict__ active_tri,
                                                           uint32_t* __restrict__ counters)
{
    // block-aggregated append: every wavefront counts its members of the six lists per pass (ballots), one thread per
    // list turns the 4 x AT_ITEMS counts into offsets and reserves the block's output range with ONE global atomic per
    // list (2.1 M tiles used to mean ~100 k same-address atomics = 0.49 ms), then the members are written straight to
    // their places.  (Round 1 staged the lists in 48 KB of LDS: beside the persistent projection kernel, whose one-wave
    // workgroups fragment the LDS, such a workgroup waited 2.6 ms for a contiguous piece.)
    __shared__ uint32_t s_cnt[6][AT_ITEMS][4], s_base[6], s_max[2];
    if (threadIdx.x < 2) s_max[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t t0 = blockIdx.x * (256u * AT_ITEMS);
    uint32_t mxb = 0, mxs = 0;
    uint32_t flags[AT_ITEMS];
#pragma unroll
    for (int i = 0; i < AT_ITEMS; ++i) {
        const uint32_t t = t0 + (uint32_t)i * 256u + threadIdx.x;
        const bool in = t < ntiles;
        const uint32_t bc = in ? band_cnt[t] : 0u, sc = in ? sign_cnt[t] : 0u;
        const bool f[3] = {bc != 0, sc != 0 && hot[t] != 0, false};
        // [2]: union = what a sparse all-gather has to move; [3]: sign-only tiles (gathered as soon as the
        // inverse maps are there, beside the iso-surface projection)
        // [4] / [5]: band tiles without / with boundary triangles (lean / full gather kernel)
        const bool ft = f[0] && tri[t] != 0;
        const bool fl[6] = {f[0], f[1], f[0] || f[1], f[1] && !f[0], f[0] && !ft, ft};
        uint32_t fb = 0;
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            const unsigned long long m = __ballot(fl[l]);
            if (lane == 0) s_cnt[l][i][wave] = (uint32_t)__popcll(m);
            fb |= fl[l] ? (1u << l) : 0u;
        }
        flags[i] = fb;
        if (fl[0] && bc > mxb) mxb = bc;
        if (fl[1] && sc > mxs) mxs = sc;
    }
    __syncthreads();
    // longest lists (decides whether the wave-per-tile sort has to run at all)
    if (mxb > 64u) atomicMax(&s_max[0], mxb);
    if (mxs > 64u) atomicMax(&s_max[1], mxs);
    if (threadIdx.x < 6) {
        const int idx[6] = {1, 2, 5, 6, 10, 11};
        uint32_t n = 0;
        for (int i = 0; i < AT_ITEMS; ++i)
            for (int w = 0; w < 4; ++w) {
                const uint32_t c = s_cnt[threadIdx.x][i][w];
                s_cnt[threadIdx.x][i][w] = n;   // count -> offset inside the block's range
                n += c;
            }
        s_base[threadIdx.x] = n ? atomicAdd(&counters[idx[threadIdx.x]], n) : 0u;
    }
    __syncthreads();
    if (threadIdx.x >= 6 && threadIdx.x < 8) {
        const uint32_t v = s_max[threadIdx.x - 6];
        if (v > 64u) atomicMax(&counters[threadIdx.x - 3], v);
    }
    uint32_t* const out[6] = {active_band, active_sign, active_any, active_sonly, active_lean, active_tri};
#pragma unroll
    for (int i = 0; i < AT_ITEMS; ++i) {
        const uint32_t t = t0 + (uint32_t)i * 256u + threadIdx.x;
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            const bool f = (flags[i] >> l) & 1u;
            const unsigned long long m = __ballot(f);
            if (f) out[l][s_base[l] + s_cnt[l][i][wave] + (uint32_t)__popcll(m & below)] = t;
        }
    }
}

// short lists (<= 64 entries): SORT_LANES lanes per active tile, rank sort (every lane ranks every SORT_LANES-th
// entry; the lanes of a group read the same addresses in the inner loop)
#define SORT_LANES 8
__global__ void __launch_bounds__(256) bin_sort_small_kernel(const uint32_t* __restrict__ active, uint32_t n_active,
                                                            const uint32_t* __restrict__ off,
                                                            const uint32_t* __restrict__ in,
                                                            uint32_t* __restrict__ out, const uint32_t* __restrict__ abort_flag)
{
    if (*abort_flag) return;   // speculated sizes of this call did not hold (run_impl)
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t w = gid / SORT_LANES, sub = gid % SORT_LANES;
    if (w >= n_active) return;
    const uint32_t t = active[w];
    const uint32_t b = off[t], n = off[t + 1] - b;
    if (n > 64) return;
    for (uint32_t i = sub; i < n; i += SORT_LANES) {
        const uint32_t v = in[b + i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += (in[b + j] < v) ? 1u : 0u;
        out[b + rank] = v;
    }
}

// one wavefront per active tile: rank-sort both lists ascending (the reference visits
// elements 1..nel in order; the ordered gather needs the same order).
__global__ void __launch_bounds__(256) bin_sort_kernel(const uint32_t* __restrict__ active, uint32_t n_active,
                                                      const uint32_t* __restrict__ off,
                                                      const uint32_t* __restrict__ in,
                                                      uint32_t* __restrict__ out, const uint32_t* __restrict__ abort_flag)
{
    if (*abort_flag) return;   // speculated sizes of this call did not hold (run_impl)
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n_active) return;
    const uint32_t t = active[w];
    const uint32_t b = off[t], n = off[t + 1] - b;
    if (n <= 64) return;   // handled by bin_sort_small_kernel
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t v = in[b + i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += (in[b + j] < v) ? 1u : 0u;
        out[b + rank] = v;
    }
}

// ------------------------------------------------------------------------------------
// sentinel sweep: untouched voxels are dist = 1e10, sign = -1 (sdfOnDensityField.jl:172-173,483;
// SignDetection.jl:13) => sdf = -1e10.  Pure HBM write stream, 16 B per lane.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fill_kernel(double* __restrict__ p, int64_t n, double v)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n2 = n >> 1;
    double2* p2 = reinterpret_cast<double2*>(p);
    const double2 vv = make_double2(v, v);
    for (int64_t j = i; j < n2; j += stride) p2[j] = vv;
    if (i == 0 && (n & 1)) p[n - 1] = v;
}

// ------------------------------------------------------------------------------------
// main kernel: one wavefront = one 4x4x4 voxel tile
// ------------------------------------------------------------------------------------
struct MainArgs {
    GridDev g;
    SlabInfo s;
    double rho_t;
    const uint32_t* active;
    uint32_t n_active;
    const uint32_t* band_off;
    const uint32_t* band_ent;
    const uint32_t* sign_off;
    const uint32_t* sign_ent;
    const BandItem* items;
    const void* erec;   // ElemRec[] or TetRec[]
    double* dist;
    double* sign;
    double* sdf;
    double* xp;
    const double* iso_res;     // per (iso item, box voxel) distances from iso_project_kernel
    const double* iso_res_xp;  // projection points (only when xp is requested)
    const void* sbox;          // HEX8: per-element boxes / chunk offsets / results of sign_project_kernel
    const uint32_t* s_store_off;
    const double* sres;
    const uint8_t* hot;        // per tile: some candidate reaches rho_t
    const uint32_t* abort_flag;   // set on the device when the speculated sizes of this call did not hold (run_impl)
    int true_min;  // order-independent semantics (see write_value / process_triangle)
    int sdf_mode;  // 1: sdf = dist*sign in one kernel; 2: dist pass stores -dist; 3: sign pass flips;
                   // 4: dist pass after the sign pass (keeps the sign already stored)
};

// TRI = false: instantiation for tiles whose band lists hold no boundary triangles (tile classification in
// band_bin / active_tiles): without the triangle code the kernel needs half the registers and twice as many
// wavefronts hide the dependent-load latency it is bound by
// SYM = true: the order-independent instantiation of the triangle code (MainArgs::true_min is set with it)
template <class Rec, bool DO_DIST, bool DO_SIGN, bool TRI = true, bool SYM = false>
__global__ void __launch_bounds__(256) sdf_tiles_kernel(MainArgs A)
{
    const Rec* __restrict__ erec = static_cast<const Rec*>(A.erec);
    // wave-uniform tile id: everything derived from it lives in SGPRs / scalar loads
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= A.n_active) return;
    const int lane = threadIdx.x & 63;
    // (the abort flag travels with the tile id: two independent scalar loads, one wait - a dependent load of its own
    // at the top of this latency-bound kernel cost 15 % of its run time)
    const uint32_t ab = *A.abort_flag;
    const uint32_t t = A.active[w];
    if (ab) return;
    const int tx = t % A.s.ntx, ty = (t / A.s.ntx) % A.s.nty, tz = t / (A.s.ntx * A.s.nty);
    const int i = tx * 4 + (lane & 3), j = ty * 4 + ((lane >> 2) & 3), kl = tz * 4 + (lane >> 4);
    const int k = slab_global_k(A.s, kl);
    const bool valid = (i < A.g.nx) && (j < A.g.ny) && (kl < A.s.nzl) && (k < A.s.k1);
    double x[3];
    x[0] = grid_coord(A.g, 0, i);
    x[1] = grid_coord(A.g, 1, j);
    x[2] = grid_coord(A.g, 2, k);

    VoxState st;
    st.cur = 1.0e10;
    st.xp[0] = st.xp[1] = st.xp[2] = 0.0;
    if (DO_DIST) {
        const int ci = (int)cell_of(A.g, 0, x[0]), cj = (int)cell_of(A.g, 1, x[1]),
                  ck = (int)cell_of(A.g, 2, x[2]);
        const uint32_t b = A.band_off[t], e = A.band_off[t + 1];
        // Lane l fetches the header of list entry l (two dependent vector loads for the whole list instead of
        // two dependent scalar loads per item); the headers are then broadcast item by item (v_readlane) and
        // the iso look-ups of four items are issued together.  Items are consumed in list order, which is all
        // the update rules depend on.
        for (uint32_t p0 = b; p0 < e; p0 += 64u) {
            const uint32_t n = (e - p0 < 64u) ? e - p0 : 64u;
            int32_t my_it = 0;
            int32_t h[15];   // imin[3], imax[3], el, kind, lo[3], dim[3], store_off
#pragma unroll
            for (int q = 0; q < 15; ++q) h[q] = 0;
            if ((uint32_t)lane < n) {
                my_it = (int32_t)A.band_ent[p0 + lane];
                const int32_t* w = reinterpret_cast<const int32_t*>(&A.items[my_it]);
#pragma unroll
                for (int q = 0; q < 8; ++q) h[q] = w[q];
#pragma unroll
                for (int q = 0; q < 6; ++q) h[8 + q] = w[offsetof(BandItem, lo) / 4 + q];
                h[14] = w[offsetof(BandItem, store_off) / 4];
            }
            for (uint32_t q0 = 0; q0 < n; q0 += 4u) {
                bool in[4];
                int kind[4];
                size_t slot[4];
                double d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = q0 + u < n;
                    const int qq = (int)(ok ? q0 + u : n - 1u);
                    int32_t hh[15];
#pragma unroll
                    for (int q = 0; q < 15; ++q) hh[q] = __builtin_amdgcn_readlane(h[q], qq);
                    in[u] = valid && ok && ci >= hh[0] && ci <= hh[3] && cj >= hh[1] && cj <= hh[4] && ck >= hh[2] &&
                            ck <= hh[5];
                    kind[u] = hh[7];
                    slot[u] = 0;
                    d[u] = INFINITY;
                    if (kind[u] == 0) {
                        const int32_t lo3[3] = {hh[8], hh[9], hh[10]}, dim3[3] = {hh[11], hh[12], hh[13]};
                        slot[u] = (size_t)tile_chunk((uint32_t)hh[14], tile_box(lo3, dim3), tx, ty, tz) * 64u + (uint32_t)lane;
                        if (in[u]) d[u] = A.iso_res[slot[u]];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (in[u]) {
                        if (kind[u] == 0) {
                            // WriteValue of the pre-computed iso candidate (sdfOnDensityField.jl:617-621)
                            if (A.true_min && A.iso_res_xp && fabs(d[u]) == st.cur) {
                                const double q[3] = {A.iso_res_xp[3 * slot[u]], A.iso_res_xp[3 * slot[u] + 1], A.iso_res_xp[3 * slot[u] + 2]};
                                write_value(st, d[u], q, true);   // symmetric tie rule
                            } else if (fabs(d[u]) < st.cur) {
                                st.cur = d[u];
                                if (A.iso_res_xp) {
                                    st.xp[0] = A.iso_res_xp[3 * slot[u]];
                                    st.xp[1] = A.iso_res_xp[3 * slot[u] + 1];
                                    st.xp[2] = A.iso_res_xp[3 * slot[u] + 2];
                                }
                            }
                        } else if constexpr (TRI) {
                            const BandItem& T = A.items[__builtin_amdgcn_readlane(my_it, (int)(q0 + u))];
                            process_triangle<Rec, SYM>(st, T, erec[T.el], A.rho_t, x);
                        }
                    }
                }
            }
        }
    }
    double sg = -1.0;
    if (DO_SIGN) {
        const uint32_t b = A.sign_off[t], e = A.sign_off[t + 1];
        if constexpr (std::is_same<Rec, ElemRec>::value) {
            // Sign_Detection_HEX8 (SignDetection.jl:27-70): wave-uniform walk over the tile's candidate
            // list in ascending element order; the inverse maps were computed item-major by
            // sign_project_kernel, so a visit is a lookup.  The state machine (:41-70) runs while the
            // candidates' max density is collected; the test of :36 is applied at the end (same result).
            if (A.hot[t]) {
                const SignBox* __restrict__ sbox = static_cast<const SignBox*>(A.sbox);
                bool any = false, done = false;
                double cmax = -INFINITY, max_local = 10.0;
                // lane l fetches the record of candidate l; records are broadcast one by one (v_readlane), four
                // look-ups are issued together, and the state machine consumes them in order.  "Point inside the element's AABB"
                // (SignDetection.jl:30) == "lattice index inside the element's box" by construction of the box.
                for (uint32_t p0 = b; p0 < e; p0 += 64u) {
                    const uint32_t n = (e - p0 < 64u) ? e - p0 : 64u;
                    int32_t hs[9];   // lo[3], dim[3], rmax (2 words), store_off of candidate l
#pragma unroll
                    for (int q = 0; q < 9; ++q) hs[q] = 0;
                    if ((uint32_t)lane < n) {
                        const uint32_t el = A.sign_ent[p0 + lane];
                        const int32_t* w = reinterpret_cast<const int32_t*>(&sbox[el]);
#pragma unroll
                        for (int q = 0; q < 8; ++q) hs[q] = w[q];
                        hs[8] = (int32_t)A.s_store_off[el];
                    }
                    for (uint32_t q0 = 0; q0 < n; q0 += 4u) {
                        bool in[4];
                        double v[4], rmax[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool ok = q0 + u < n;
                            const int qq = (int)(ok ? q0 + u : n - 1u);
                            int32_t hh[9];
#pragma unroll
                            for (int q = 0; q < 9; ++q) hh[q] = __builtin_amdgcn_readlane(hs[q], qq);
                            const int32_t lo3[3] = {hh[0], hh[1], hh[2]}, dim3[3] = {hh[3], hh[4], hh[5]};
                            const uint32_t di = (uint32_t)(i - lo3[0]), dj = (uint32_t)(j - lo3[1]), dk = (uint32_t)(kl - lo3[2]);
                            in[u] = valid && ok && di < (uint32_t)dim3[0] && dj < (uint32_t)dim3[1] && dk < (uint32_t)dim3[2];
                            rmax[u] = __hiloint2double(hh[7], hh[6]);
                            v[u] = INFINITY;
                            if (in[u] && !done)   // a lane that is done (as of the previous batch) needs no more look-ups
                                v[u] = A.sres[(size_t)tile_chunk((uint32_t)hh[8], tile_box(lo3, dim3), tx, ty, tz) * 64u + (uint32_t)lane];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (in[u]) {
                                any = true;
                                if (rmax[u] > cmax) cmax = rmax[u];
                                const double m = fabs(v[u]);
                                if (A.true_min) {   // any element that holds the point decides for +1
                                    if (m < 1.01 && !__builtin_signbit(v[u])) sg = 1.0;
                                } else if (!done && m < 1.01 && max_local > m) {
                                    if (!__builtin_signbit(v[u])) sg = 1.0;
                                    if (m < 0.95) done = true;
                                    else max_local = m;
                                }
                            }
                        }
                    }
                }
                if (!(any && !(cmax < A.rho_t))) sg = -1.0;   // SignDetection.jl:36
            }
        } else {
            // Sign_Detection_TET4 (SignDetection.jl:116-151): bin index of the point (:258-268), candidates of
            // that bin, +1 when one of them holds the point with rho >= rho_t (the reference stops at the first
            // such element of its ascending list - which one it is does not change the sign)
            const int dims[3] = {A.g.nx, A.g.ny, A.g.nz};
            int gi[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                double f = floor((x[ax] - A.g.amin[ax]) / A.g.cell) + 1.0;
                if (f > (double)dims[ax]) f = (double)dims[ax];
                if (f < 1.0) f = 1.0;
                gi[ax] = (int)f;
            }
            bool done = false;
            for (uint32_t p = b; p < e; ++p) {
                if (!__any(valid && !done)) break;   // every voxel of the tile has its first hit
                const TetRec& E = erec[A.sign_ent[p]];
                const bool in = valid && !done && gi[0] >= E.blo[0] && gi[0] <= E.bhi[0] && gi[1] >= E.blo[1] &&
                                gi[1] <= E.bhi[1] && gi[2] >= E.blo[2] && gi[2] <= E.bhi[2];
                // the 4x4 solve of is_point_in_tetrahedron only for the points the face planes leave undecided
                bool outside, pit;
                tet4_classify(E, x, outside, pit);
                if (in && !outside && !pit) pit = point_in_tet(E, x);
                if (in && pit) {
                    double loc[3], N[4];
                    if (find_local_tet4(E, x, loc)) {
                        tet4_shape(loc, N);
                        double rho = 0.0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) rho += N[q] * E.r[q];
                        if (rho >= A.rho_t) { sg = 1.0; done = true; }
                    }
                }
            }
        }
    }
    if (valid) {
        const int64_t v = ((int64_t)kl * A.g.ny + j) * A.g.nx + i;
        if (A.dist) A.dist[v] = st.cur;
        if (A.sign) A.sign[v] = sg;
        if (A.sdf) {
            // fused output dist*sign (RhoToSDF.jl:171) in two passes: the distance pass stores
            // dist * (-1), the sign pass negates where the sign is +1 (bit-identical products)
            if (A.sdf_mode == 1) A.sdf[v] = st.cur * sg;
            else if (A.sdf_mode == 2) A.sdf[v] = -st.cur;
            else if (A.sdf_mode == 4) A.sdf[v] = (A.sdf[v] > 0.0) ? st.cur : -st.cur;   // sign pass ran first
            else if (sg > 0.0) A.sdf[v] = -A.sdf[v];
        }
        if (A.xp) {
            A.xp[3 * v] = st.xp[0];
            A.xp[3 * v + 1] = st.xp[1];
            A.xp[3 * v + 2] = st.xp[2];
        }
    }
}

// ------------------------------------------------------------------------------------
// sparse stitching of the volume across GPUs: only tiles that can differ from the sentinel travel.
// A packed tile is 64 doubles in lane order (x + 4 y + 16 z inside the tile) plus its id in the
// tile numbering of the WHOLE grid.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_tiles_kernel(const uint32_t* __restrict__ tiles, uint32_t n, SlabInfo s,
                                                        GridDev g, const double* __restrict__ local, double sentinel,
                                                        double* __restrict__ payload, uint32_t* __restrict__ ids)
{
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= n) return;
    const int lane = threadIdx.x & 63;
    const uint32_t t = tiles[w];
    const int tx = t % s.ntx, ty = (t / s.ntx) % s.nty, tz = t / (s.ntx * s.nty);
    const int i = tx * 4 + (lane & 3), j = ty * 4 + ((lane >> 2) & 3), kl = tz * 4 + (lane >> 4);
    const int k = slab_global_k(s, kl);
    const bool valid = (i < g.nx) && (j < g.ny) && (kl < s.nzl) && (k < s.k1);
    payload[(size_t)w * 64 + lane] = valid ? local[((int64_t)kl * g.ny + j) * g.nx + i] : sentinel;
    if (lane == 0) ids[w] = ((uint32_t)(k >> 2) * s.nty + ty) * s.ntx + tx;   // tile layer of the whole grid
}

__global__ void __launch_bounds__(256) unpack_tiles_kernel(const double* __restrict__ payload,
                                                          const uint32_t* __restrict__ ids, uint32_t n, int nx, int ny,
                                                          int nz, double* __restrict__ volume)
{
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= n) return;
    const int lane = threadIdx.x & 63;
    const int ntx = (nx + 3) / 4, nty = (ny + 3) / 4;
    const uint32_t t = ids[w];
    const int tx = t % ntx, ty = (t / ntx) % nty, tz = t / (ntx * nty);
    const int i = tx * 4 + (lane & 3), j = ty * 4 + ((lane >> 2) & 3), k = tz * 4 + (lane >> 4);
    if (i < nx && j < ny && k < nz) volume[((int64_t)k * ny + j) * nx + i] = payload[(size_t)w * 64 + lane];
}

// sign-only tiles: every voxel is +-1e10, one 64-bit mask per tile
__global__ void __launch_bounds__(256) pack_masks_kernel(const uint32_t* __restrict__ tiles, uint32_t n, SlabInfo s,
                                                        GridDev g, const double* __restrict__ local,
                                                        uint64_t* __restrict__ masks, uint32_t* __restrict__ ids)
{
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= n) return;
    const int lane = threadIdx.x & 63;
    const uint32_t t = tiles[w];
    const int tx = t % s.ntx, ty = (t / s.ntx) % s.nty, tz = t / (s.ntx * s.nty);
    const int i = tx * 4 + (lane & 3), j = ty * 4 + ((lane >> 2) & 3), kl = tz * 4 + (lane >> 4);
    const int k = slab_global_k(s, kl);
    const bool valid = (i < g.nx) && (j < g.ny) && (kl < s.nzl) && (k < s.k1);
    const uint64_t m = __ballot(valid && local[((int64_t)kl * g.ny + j) * g.nx + i] > 0.0);
    if (lane == 0) {
        masks[w] = m;
        ids[w] = ((uint32_t)(k >> 2) * s.nty + ty) * s.ntx + tx;
    }
}

__global__ void __launch_bounds__(256) unpack_masks_kernel(const uint64_t* __restrict__ masks,
                                                          const uint32_t* __restrict__ ids, uint32_t n, int nx, int ny,
                                                          int nz, double magnitude, double* __restrict__ volume)
{
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= n) return;
    const int lane = threadIdx.x & 63;
    const int ntx = (nx + 3) / 4, nty = (ny + 3) / 4;
    const uint32_t t = ids[w];
    const uint64_t m = masks[w];
    const int tx = t % ntx, ty = (t / ntx) % nty, tz = t / (ntx * nty);
    const int i = tx * 4 + (lane & 3), j = ty * 4 + ((lane >> 2) & 3), k = tz * 4 + (lane >> 4);
    if (i < nx && j < ny && k < nz) volume[((int64_t)k * ny + j) * nx + i] = ((m >> lane) & 1ull) ? magnitude : -magnitude;
}

// The whole exchange buffer of the sparse stitching in two launches: every rank's segment is
// [n_full, n_mask (two int64) | n_cap_full x 64 doubles | ids | masks | mask ids] (slabs.py: SlabGather), the counts
// are read from the segment headers ON THE DEVICE, so the host neither waits for them nor launches per rank.
// A count beyond the capacity marks a segment that was not packed (the step is repeated with larger segments).
__global__ void __launch_bounds__(256) unpack_segments_tiles_kernel(const double* __restrict__ buf, int64_t seglen, uint32_t cap_full,
                                                                   int nx, int ny, int nz, double* __restrict__ volume)
{
    const double* __restrict__ seg = buf + (int64_t)blockIdx.y * seglen;
    const int64_t nf = reinterpret_cast<const int64_t*>(seg)[0];
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (nf < 0 || nf > (int64_t)cap_full || (int64_t)w >= nf) return;
    const double* __restrict__ payload = seg + 2;
    const uint32_t* __restrict__ ids = reinterpret_cast<const uint32_t*>(seg + 2 + (int64_t)cap_full * 64);
    const int lane = threadIdx.x & 63;
    const int ntx = (nx + 3) / 4, nty = (ny + 3) / 4;
    const uint32_t t = ids[w];
    const int tx = t % ntx, ty = (t / ntx) % nty, tz = t / (ntx * nty);
    const int i = tx * 4 + (lane & 3), j = ty * 4 + ((lane >> 2) & 3), k = tz * 4 + (lane >> 4);
    if (i < nx && j < ny && k < nz) volume[((int64_t)k * ny + j) * nx + i] = payload[(size_t)w * 64 + lane];
}
__global__ void __launch_bounds__(256) unpack_segments_masks_kernel(const double* __restrict__ buf, int64_t seglen, uint32_t cap_full,
                                                                   uint32_t cap_mask, int nx, int ny, int nz, double magnitude,
                                                                   double* __restrict__ volume)
{
    const double* __restrict__ seg = buf + (int64_t)blockIdx.y * seglen;
    const int64_t nf = reinterpret_cast<const int64_t*>(seg)[0], nm = reinterpret_cast<const int64_t*>(seg)[1];
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (nf < 0 || nf > (int64_t)cap_full || nm < 0 || nm > (int64_t)cap_mask || (int64_t)w >= nm) return;
    const uint64_t* __restrict__ masks = reinterpret_cast<const uint64_t*>(seg + 2 + (int64_t)cap_full * 64 + (cap_full + 1) / 2);
    const uint32_t* __restrict__ ids = reinterpret_cast<const uint32_t*>(masks + cap_mask);
    const int lane = threadIdx.x & 63;
    const int ntx = (nx + 3) / 4, nty = (ny + 3) / 4;
    const uint32_t t = ids[w];
    const int tx = t % ntx, ty = (t / ntx) % nty, tz = t / (ntx * nty);
    const int i = tx * 4 + (lane & 3), j = ty * 4 + ((lane >> 2) & 3), k = tz * 4 + (lane >> 4);
    if (i < nx && j < ny && k < nz) volume[((int64_t)k * ny + j) * nx + i] = ((masks[w] >> lane) & 1ull) ? magnitude : -magnitude;
}

// ------------------------------------------------------------------------------------
// plan: device workspace that survives across calls
// ------------------------------------------------------------------------------------
struct r2s_plan {
    int device = 0;
    int n_cu = 256;
    DevBuf deg, ine_ptr, ine, cursor, erec, cls, fmask, nitems, item_off, items;
    DevBuf band_cnt, band_off, band_raw, band_ent, sign_cnt, sign_off, sign_raw, sign_ent;
    DevBuf active, active_sign, active_any, active_sonly, active_lean, active_tri, tri, hot, counters, scan_tmp[3], scan_tmp2[3], nchunks, chunk_off, iso_res, iso_res_xp, strag;
    DevBuf perm, wchunks, hardflag;   // work order of the persistent projection kernel (HEX8)
    DevBuf sbox, s_nchunks, s_chunk_off, sres;   // item-major inverse maps of the sign pass (HEX8)
    DevBuf nstore, store_off, s_nstore, s_store_off;   // storage (tile) chunk counts / offsets
    // state of the last run, for r2s_plan_pack_tiles_dev
    SlabInfo last_s;
    GridDev last_g;
    uint32_t last_n_any = 0, last_n_band = 0, last_n_sonly = 0;
    bool has_last = false;
    uint32_t* h_pinned = nullptr;  // 32 words
    uint32_t* d_pinned = nullptr;  // the same block as the device sees it
    // sizes read back by the last completed call, and the shapes they belong to (speculation, see read_back_kernel)
    struct Spec {
        bool valid = false;
        int64_t key[12] = {0};
        uint32_t v[16] = {0};
    } spec;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // second stream: sentinel sweep + sign pass run beside the iso-surface projection (fused SDF output)
    hipStream_t st2 = nullptr;
    hipStream_t st3 = nullptr;   // bounding half-spaces (hex_planes_kernel), beside the chains of the other two during preparation
    hipEvent_t ev2[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fast[2] = {nullptr, nullptr};   // around iso_project_hex_pl_kernel alone (r2s_stats.ms_iso_fast)
    bool fast_timed = false;                      // ... recorded by the current call
};

// exclusive scans of one or two (in1 != nullptr) arrays of n entries each
// (bank 1: the workspace of scans that run on the second stream beside scans of the first)
static int scan_exclusive2(r2s_plan* P, const uint32_t* in0, uint32_t* out0, const uint32_t* in1, uint32_t* out1,
                           int64_t n, hipStream_t st, int level = 0, int bank = 0)
{
    DevBuf* tmp = bank ? P->scan_tmp2 : P->scan_tmp;
    if (n <= 0) return 0;
    const unsigned ny = in1 ? 2u : 1u;
    const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    ScanPair sp;
    sp.in[0] = in0; sp.out[0] = out0; sp.in[1] = in1; sp.out[1] = out1;
    sp.sums[0] = sp.sums[1] = nullptr;
    if (nb == 1) {
        scan_block_kernel<<<dim3(1, ny), SCAN_BLOCK, 0, st>>>(sp, n);
        return 0;
    }
    if (level >= 3) return fail(R2S_ERR_ARG, "scan too large");
    if (tmp[level].ensure(sizeof(uint32_t) * 2 * (size_t)nb)) return fail(R2S_ERR_NOMEM, "scan workspace");
    sp.sums[0] = tmp[level].as<uint32_t>();
    sp.sums[1] = sp.sums[0] + nb;
    scan_block_kernel<<<dim3((unsigned)nb, ny), SCAN_BLOCK, 0, st>>>(sp, n);
    int rc = scan_exclusive2(P, sp.sums[0], sp.sums[0], in1 ? sp.sums[1] : nullptr, in1 ? sp.sums[1] : nullptr, nb, st,
                             level + 1, bank);
    if (rc) return rc;
    scan_add_kernel<<<dim3((unsigned)nb, ny), SCAN_BLOCK, 0, st>>>(sp, n);
    return 0;
}
static int scan_exclusive(r2s_plan* P, const uint32_t* in, uint32_t* out, int64_t n, hipStream_t st)
{
    return scan_exclusive2(P, in, out, nullptr, nullptr, n, st);
}

extern "C" {

int r2s_version(void) { return R2S_VERSION; }
const char* r2s_last_error(void) { return g_err.c_str(); }

int r2s_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

void r2s_default_params(r2s_params* p)
{
    memset(p, 0, sizeof *p);
    p->band_factor = 1.1;
    p->elem_type = R2S_HEX8;
    p->device = -1;
}

int r2s_grid_make(const double xmin[3], const double xmax[3], int64_t n_max, int64_t margin, r2s_grid* out)
{
    if (!xmin || !xmax || !out || n_max <= 0 || margin < 0) return fail(R2S_ERR_ARG, "r2s_grid_make: bad argument");
    double ext = xmax[0] - xmin[0];
    for (int i = 1; i < 3; ++i) ext = std::max(ext, xmax[i] - xmin[i]);
    const double cell = ext / (double)n_max;       // Grid.jl:17
    const double m = (double)margin * cell;
    out->ngp = 1;
    for (int i = 0; i < 3; ++i) {
        const double lo = xmin[i] - m;              // Grid.jl:20
        const double hi = xmax[i] + m;              // Grid.jl:21
        out->N[i] = (int64_t)std::ceil((hi - lo) / cell);  // Grid.jl:24
        out->aabb_min[i] = lo;
        out->aabb_max[i] = lo + (double)out->N[i] * cell;  // Grid.jl:26
        out->ngp *= out->N[i] + 1;                  // Grid.jl:30
    }
    out->cell_size = cell;
    return 0;
}

int r2s_auto_grid(const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, int32_t elem_type,
                  r2s_grid* out, double* median_edge)
{
    if (!X || !IEN || !out || nnp <= 0 || nel <= 0) return fail(R2S_ERR_ARG, "r2s_auto_grid: bad argument");
    static const int hex_edges[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                         {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    static const int tet_edges[6][2] = {{0, 1}, {1, 2}, {2, 0}, {0, 3}, {1, 3}, {2, 3}};
    const int nen = elem_type == R2S_HEX8 ? 8 : 4, noe = elem_type == R2S_HEX8 ? 12 : 6;
    std::vector<double> d((size_t)noe * nel);
    for (int64_t e = 0; e < nel; ++e)
        for (int k = 0; k < noe; ++k) {  // calculate_edge_distances, Grid_setup.jl:28-51
            const int s = elem_type == R2S_HEX8 ? hex_edges[k][0] : tet_edges[k][0];
            const int f = elem_type == R2S_HEX8 ? hex_edges[k][1] : tet_edges[k][1];
            const int64_t ns = IEN[e * nen + s] - 1, nf = IEN[e * nen + f] - 1;
            if (ns < 0 || ns >= nnp || nf < 0 || nf >= nnp) return fail(R2S_ERR_ARG, "IEN out of range");
            const double dx = X[3 * nf] - X[3 * ns], dy = X[3 * nf + 1] - X[3 * ns + 1],
                         dz = X[3 * nf + 2] - X[3 * ns + 2];
            d[(size_t)e * noe + k] = std::sqrt(dx * dx + dy * dy + dz * dz);
        }
    const size_t n = d.size();
    double B;
    std::nth_element(d.begin(), d.begin() + n / 2, d.end());
    B = d[n / 2];
    if (n % 2 == 0) {  // median of an even count = mean of the middle pair
        const double lo = *std::max_element(d.begin(), d.begin() + n / 2);
        B = (lo + B) / 2;
    }
    double mn[3], mx[3];
    for (int i = 0; i < 3; ++i) { mn[i] = INFINITY; mx[i] = -INFINITY; }
    for (int64_t p = 0; p < nnp; ++p)
        for (int i = 0; i < 3; ++i) {
            mn[i] = std::min(mn[i], X[3 * p + i]);
            mx[i] = std::max(mx[i], X[3 * p + i]);
        }
    const double ext = std::max(mx[0] - mn[0], std::max(mx[1] - mn[1], mx[2] - mn[2]));
    const int64_t n_new = (int64_t)std::floor(ext / B);  // Grid_setup.jl:103
    if (median_edge) *median_edge = B;
    return r2s_grid_make(mn, mx, n_new, 3, out);
}

int r2s_plan_create(int32_t device, r2s_plan** out)
{
    if (!out) return fail(R2S_ERR_ARG, "r2s_plan_create: null out");
    if (device < 0) {
        int rc = check_device(0);
        if (rc) return rc;
        HIP_TRY(hipGetDevice(&device));
    }
    int rc = check_device(device);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    r2s_plan* P = new r2s_plan();
    P->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) P->n_cu = prop.multiProcessorCount;
    }
    HIP_TRY(hipHostMalloc((void**)&P->h_pinned, 128, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(hipHostGetDevicePointer((void**)&P->d_pinned, P->h_pinned, 0));
    for (int i = 0; i < 8; ++i) HIP_TRY(hipEventCreate(&P->ev[i]));
    for (int i = 0; i < 8; ++i) HIP_TRY(hipEventCreate(&P->ev2[i]));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreate(&P->ev_fast[i]));
    {
        // high priority: the short stages of the second stream (sentinel sweep, inverse maps of the sign pass,
        // sign-only gather) get wave slots ahead of the long persistent projection kernel.  Measured on the
        // north-star workload: high 6.1-6.2 ms/step, normal / low 6.3-6.45.
        int prio_lo = 0, prio_hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        (void)prio_lo;
        HIP_TRY(hipStreamCreateWithPriority(&P->st2, hipStreamNonBlocking, prio_hi));
        HIP_TRY(hipStreamCreateWithFlags(&P->st3, hipStreamNonBlocking));
    }
    *out = P;
    return 0;
}

void r2s_plan_destroy(r2s_plan* P)
{
    if (!P) return;
    (void)hipSetDevice(P->device);
    DevBuf* all[] = {&P->deg, &P->ine_ptr, &P->ine, &P->cursor, &P->erec, &P->cls, &P->fmask, &P->nitems,
                     &P->item_off, &P->items, &P->band_cnt, &P->band_off, &P->band_raw, &P->band_ent,
                     &P->sign_cnt, &P->sign_off, &P->sign_raw, &P->sign_ent, &P->active, &P->active_sign, &P->active_any, &P->active_sonly, &P->active_lean, &P->active_tri, &P->tri,
                     &P->hot, &P->counters, &P->nchunks, &P->chunk_off, &P->iso_res, &P->iso_res_xp, &P->strag,
                     &P->perm, &P->wchunks, &P->hardflag, &P->sbox, &P->s_nchunks, &P->s_chunk_off, &P->sres, &P->nstore, &P->store_off, &P->s_nstore, &P->s_store_off,
                     &P->scan_tmp[0], &P->scan_tmp[1], &P->scan_tmp[2], &P->scan_tmp2[0], &P->scan_tmp2[1], &P->scan_tmp2[2]};
    for (DevBuf* b : all) b->release();
    if (P->h_pinned) (void)hipHostFree(P->h_pinned);
    for (int i = 0; i < 8; ++i)
        if (P->ev[i]) (void)hipEventDestroy(P->ev[i]);
    for (int i = 0; i < 8; ++i)
        if (P->ev2[i]) (void)hipEventDestroy(P->ev2[i]);
    for (int i = 0; i < 2; ++i)
        if (P->ev_fast[i]) (void)hipEventDestroy(P->ev_fast[i]);
    if (P->st2) (void)hipStreamDestroy(P->st2);
    if (P->st3) (void)hipStreamDestroy(P->st3);
    delete P;
}


}  // extern "C"

// entries of the straggler list for `n_store` storage chunks (64 result slots each): 1/16 of the slots, but never
// fewer than min(slots, 2^20) - small grids cannot overflow whatever the mesh does
static uint32_t iso_straggler_cap(uint32_t n_store)
{
    static const int cap_env = getenv("R2S_ISO_STRAGGLER_CAP") ? atoi(getenv("R2S_ISO_STRAGGLER_CAP")) : 0;   // test hook: force the overflow path
    if (cap_env > 0) return (uint32_t)cap_env;
    const uint64_t slots = 64ull * std::max<uint32_t>(n_store, 1u);
    return (uint32_t)std::max<uint64_t>(slots / 16, std::min<uint64_t>(slots, 1ull << 20));
}

// The HEX8 projection on stream `st`: the persistent fast-path kernel, behind it the complete
// solver on the pairs it handed over, last the sweep that only does something when the list overflowed.  Nothing here
// waits for a count on the host.  counters: [8] chunk counter, [12] list entries, [13] overflow flag, [14] list head.
static int iso_project_hex(r2s_plan* P, hipStream_t st, uint32_t n_items, uint32_t n_chunks, uint32_t n_store, uint32_t wps,
                           const GridDev& g, const SlabInfo& s, double rho_t, double* res_xp, uint32_t* counters,
                           const uint32_t* abort_flag)
{
    static const int free_env = getenv("R2S_ISO_FREE") ? atoi(getenv("R2S_ISO_FREE")) : 0;   // tuning knob: fewer persistent wavefronts
    const uint32_t resident = (uint32_t)P->n_cu * 4u * wps - (uint32_t)std::min(std::max(free_env, 0), (int)P->n_cu * 4);
    // chunks per fetch (tools/rank_share.py, R2S_ISO_GROUP): consecutive chunks mostly belong to one item, so a group costs
    // one atomic and one item record (dependent loads with every lane of the wavefront waiting) instead of one per chunk.
    // 8 when a wavefront gets a hundred chunks and more (N = 1: 4.14 -> 4.06 ms against 4), never fewer than 4: one rank
    // of 8 with single chunks took 0.69 instead of 0.45 ms (the rule of round 2, "1 for a small share", dated from the
    // kernel without lane refill, where a coarse group left a wavefront 15 % behind the others)
    static const int group_env = getenv("R2S_ISO_GROUP") ? atoi(getenv("R2S_ISO_GROUP")) : 0;   // tuning knob
    const uint32_t group = group_env > 0 ? (uint32_t)group_env
                           : std::max(1u, std::min(std::min(8u, std::max(4u, n_chunks / (resident * 6u))), n_chunks / (resident * 2u)));   // (small grids: two fetches per wavefront at least)
    const uint32_t ngroups = (n_chunks + group - 1) / group;
    const uint32_t cap = iso_straggler_cap(n_store);
    ENSURE(P->strag, sizeof(IsoStraggler) * (size_t)cap);
    IsoStraggler* const list = P->strag.as<IsoStraggler>();
    const uint32_t grid = std::min(ngroups, resident);
    const BandItem* items = P->items.as<BandItem>();
    const uint32_t* chunk_off = P->chunk_off.as<uint32_t>();
    const uint32_t* perm = P->perm.as<uint32_t>();
    const ElemRec* erec = P->erec.as<ElemRec>();
    double* res = P->iso_res.as<double>();
    // two instantiations: 168 registers (3 wavefronts per SIMD, 48 B of scratch per lane) when the kernel has the GPU to
    // itself, 2 per SIMD without scratch when it starts beside the binning (early launch, see run_impl)
    HIP_TRY(hipEventRecord(P->ev_fast[0], st));
    if (wps >= 3)
        iso_project_hex_pl_kernel<3><<<grid, 64, 0, st>>>(items, n_items, chunk_off, n_chunks, group, erec, g, s, rho_t, res, res_xp,
                                                          counters + 8, perm, abort_flag, list, cap, counters + 12, counters + 13);
    else
        iso_project_hex_pl_kernel<2><<<grid, 64, 0, st>>>(items, n_items, chunk_off, n_chunks, group, erec, g, s, rho_t, res, res_xp,
                                                          counters + 8, perm, abort_flag, list, cap, counters + 12, counters + 13);
    HIP_TRY(hipEventRecord(P->ev_fast[1], st));
    P->fast_timed = true;
    // two persistent wavefronts per SIMD pull entries from the list (counters[14]); those that find none leave at once
    static const int strag_wpc = getenv("R2S_STRAG_WPC") ? std::min(std::max(atoi(getenv("R2S_STRAG_WPC")), 1), 32) : 8;   // tuning knob (4-16: +-1 %)
    iso_straggler_kernel<<<(uint32_t)P->n_cu * (uint32_t)strag_wpc, 64, 0, st>>>(list, cap, counters + 12, erec, rho_t, res, res_xp, abort_flag, counters + 14, counters + 16);
    iso_sweep_kernel<<<(uint32_t)P->n_cu * 2u, 256, 0, st>>>(items, n_items, chunk_off, n_chunks, perm, erec, g, s, rho_t, res, res_xp,
                                                             counters + 13, abort_flag, counters + 16);
    return 0;
}

template <class ET>
static int run_impl(r2s_plan* P, const double* dX, int64_t nnp, const int64_t* dIEN, int64_t nel,
                     const double* d_rho_n, double rho_t, const r2s_grid* grid, const r2s_params* params,
                     int64_t k_begin, int64_t k_end, int32_t mode, double* d_dist, double* d_sign,
                     double* d_sdf, double* d_xp, void* stream, r2s_stats* stats, bool allow_spec = true)
{
    if (!P || !dX || !dIEN || !d_rho_n || !grid) return fail(R2S_ERR_ARG, "r2s_plan_run_dev: null argument");
    r2s_params prm;
    if (params) prm = *params; else r2s_default_params(&prm);
    if (nnp <= 0 || nel <= 0) return fail(R2S_ERR_ARG, "empty mesh");
    if (nel * ET::NEN >= (int64_t)1 << 31) return fail(R2S_ERR_ARG, "mesh too large");
    for (int i = 0; i < 3; ++i)
        if (grid->N[i] < 1 || grid->N[i] > 100000) return fail(R2S_ERR_ARG, "bad grid dimension");
    if (k_begin < 0 || k_end > grid->N[2] + 1 || k_begin >= k_end) return fail(R2S_ERR_ARG, "bad slab [%lld,%lld)", (long long)k_begin, (long long)k_end);
    const bool want_dist = (mode & (R2S_OUT_DIST | R2S_OUT_SDF | R2S_OUT_XP)) != 0;
    const bool want_sign = (mode & (R2S_OUT_SIGN | R2S_OUT_SDF)) != 0;
    if (!want_dist && !want_sign) return fail(R2S_ERR_ARG, "mode selects no output");
    if ((mode & R2S_OUT_DIST) && !d_dist) return fail(R2S_ERR_ARG, "d_dist is null");
    if ((mode & R2S_OUT_SIGN) && !d_sign) return fail(R2S_ERR_ARG, "d_sign is null");
    if ((mode & R2S_OUT_SDF) && !d_sdf) return fail(R2S_ERR_ARG, "d_sdf is null");
    if ((mode & R2S_OUT_XP) && !d_xp) return fail(R2S_ERR_ARG, "d_xp is null");
    HIP_TRY(hipSetDevice(P->device));
    hipStream_t st = (hipStream_t)stream;
    P->fast_timed = false;

    GridDev g;
    for (int i = 0; i < 3; ++i) {
        g.amin[i] = grid->aabb_min[i];
        g.amax[i] = grid->aabb_max[i];
        g.N[i] = (int32_t)grid->N[i];
    }
    g.cell = grid->cell_size;
    g.nx = g.N[0] + 1; g.ny = g.N[1] + 1; g.nz = g.N[2] + 1;
    SlabInfo s;
    s.k0 = (int32_t)k_begin; s.k1 = (int32_t)k_end;
    s.G = prm.zstride > 1 ? prm.zstride : 1;
    s.r = prm.zstride > 1 ? prm.zphase : 0;
    if (s.G > 1) {
        // interleaved tile layers over the whole grid (balanced Z partition for the all-gather)
        if (k_begin != 0 || k_end != g.nz) return fail(R2S_ERR_ARG, "zstride > 1 needs the full Z range [0, N3+1)");
        if (s.r < 0 || s.r >= s.G) return fail(R2S_ERR_ARG, "zphase %d not in [0, zstride=%d)", s.r, s.G);
        const int layers = (g.nz + 3) / 4;
        const int owned = layers > s.r ? (layers - s.r + s.G - 1) / s.G : 0;
        if (owned == 0) return fail(R2S_ERR_ARG, "this rank owns no tile layer (grid has %d, zstride %d)", layers, s.G);
        s.nzl = 4 * owned;
    } else {
        s.nzl = (int32_t)(k_end - k_begin);
    }
    s.ntx = (g.nx + 3) / 4; s.nty = (g.ny + 3) / 4; s.ntz = (s.nzl + 3) / 4;
    const int64_t ntiles64 = (int64_t)s.ntx * s.nty * s.ntz;
    if (ntiles64 >= ((int64_t)1 << 31)) return fail(R2S_ERR_ARG, "slab too large");
    s.ntiles = (int32_t)ntiles64;
    const uint32_t ntiles = (uint32_t)s.ntiles;
    const int64_t nvox = (int64_t)s.nzl * g.nx * g.ny;   // output voxels of this call (local planes)
    const double delta = prm.band_factor * g.cell;  // sdfOnDensityField.jl:158

    // ---- workspace ----
    ENSURE(P->deg, sizeof(uint32_t) * (size_t)(nnp + 1));
    ENSURE(P->ine_ptr, sizeof(uint32_t) * (size_t)(nnp + 1));
    ENSURE(P->cursor, sizeof(uint32_t) * (size_t)(nnp + 1));
    ENSURE(P->ine, sizeof(uint32_t) * (size_t)(nel * ET::NEN));
    ENSURE(P->erec, sizeof(typename ET::Rec) * (size_t)nel);
    ENSURE(P->cls, (size_t)nel);
    ENSURE(P->fmask, sizeof(uint32_t) * (size_t)nel);
    ENSURE(P->nitems, sizeof(uint32_t) * (size_t)(nel + 1));
    ENSURE(P->item_off, sizeof(uint32_t) * (size_t)(nel + 1));
    ENSURE(P->band_cnt, sizeof(uint32_t) * (size_t)(ntiles + 1));
    ENSURE(P->band_off, sizeof(uint32_t) * (size_t)(ntiles + 1));
    ENSURE(P->sign_cnt, sizeof(uint32_t) * (size_t)(ntiles + 1));
    ENSURE(P->sign_off, sizeof(uint32_t) * (size_t)(ntiles + 1));
    ENSURE(P->active, sizeof(uint32_t) * (size_t)ntiles);
    ENSURE(P->active_sign, sizeof(uint32_t) * (size_t)ntiles);
    ENSURE(P->active_any, sizeof(uint32_t) * (size_t)ntiles);
    ENSURE(P->active_sonly, sizeof(uint32_t) * (size_t)ntiles);
    ENSURE(P->active_lean, sizeof(uint32_t) * (size_t)ntiles);
    ENSURE(P->active_tri, sizeof(uint32_t) * (size_t)ntiles);
    ENSURE(P->tri, (size_t)ntiles + 1);
    ENSURE(P->hot, (size_t)ntiles + 1);
    ENSURE(P->counters, 128);
    uint32_t* counters = P->counters.as<uint32_t>();  // [0] bad IEN flag, [1] band tiles, [2] sign tiles, [15] abort flag
    const uint32_t* abort_flag = counters + 15;
    // ---- speculation on the sizes the host used to wait for (see read_back_kernel) ----
    const int64_t key[12] = {nnp, nel, (int64_t)ET::NEN, grid->N[0], grid->N[1], grid->N[2], k_begin, k_end,
                             (int64_t)s.G * 65536 + s.r, (int64_t)mode, (int64_t)ntiles, 0};
    static const bool spec_env = !(getenv("R2S_NO_SPECULATION") && atoi(getenv("R2S_NO_SPECULATION")));
    const bool spec = allow_spec && spec_env && P->spec.valid && memcmp(key, P->spec.key, sizeof key) == 0;
    uint32_t cnt[16] = {0};   // the 16 words of the read-back block: speculated, or read after a wait
    SpecExpect ex;
    memset(&ex, 0, sizeof ex);
    for (int q = 0; q < 32; ++q) P->h_pinned[q] = 0u;   // (words this call does not report read as 0, now and when compared)
    if (spec) {
        memcpy(cnt, P->spec.v, sizeof cnt);
        memcpy(ex.v, P->spec.v, sizeof cnt);
        ex.on = 1u;
    }

    HIP_TRY(hipStreamWaitEvent(st, P->ev2[5], 0));   // (a previous call that failed early may have left hex_planes_kernel behind)
    HIP_TRY(hipEventRecord(P->ev[0], st));
    // ---- node -> element CSR (second stream) beside the element records, classes, item counts ----
    zero_many(st, {{P->deg.p, sizeof(uint32_t) * (size_t)(nnp + 1)}, {P->cursor.p, sizeof(uint32_t) * (size_t)(nnp + 1)},
                   {counters, 128}, {P->nitems.p, sizeof(uint32_t) * (size_t)(nel + 1)}});
    HIP_TRY(hipEventRecord(P->ev2[0], st));
    // (host order: the long kernel of this stream first, then the chain of short ones for the other stream)
    elem_prep_kernel<ET><<<(unsigned)((nel + 127) / 128), 128, 0, st>>>(
        dX, dIEN, d_rho_n, nel, nnp, rho_t, g, P->erec.as<typename ET::Rec>(),
        P->cls.as<uint8_t>(), P->fmask.as<uint32_t>(), P->nitems.as<uint32_t>(), s, delta);
    HIP_TRY(hipEventRecord(P->ev2[5], st));
    HIP_TRY(hipStreamWaitEvent(P->st2, P->ev2[0], 0));
    {
        const int64_t n = nel * ET::NEN;
        node_degree_kernel<<<(unsigned)((n + 255) / 256), 256, 0, P->st2>>>(dIEN, nel, ET::NEN, nnp, P->deg.as<uint32_t>(), (int*)counters);
        int rc = scan_exclusive(P, P->deg.as<uint32_t>(), P->ine_ptr.as<uint32_t>(), nnp + 1, P->st2);
        if (rc) return rc;
        ine_fill_kernel<<<(unsigned)((n + 255) / 256), 256, 0, P->st2>>>(dIEN, nel, ET::NEN, nnp, P->ine_ptr.as<uint32_t>(), P->cursor.as<uint32_t>(), P->ine.as<uint32_t>());
    }
    HIP_TRY(hipEventRecord(P->ev2[1], P->st2));
    if constexpr (std::is_same<typename ET::Rec, ElemRec>::value) {
        // bounding half-spaces for the sign pass: behind the element records, on a stream of their own (only sign_project
        // needs them; on the second stream in front of the sign counts they delay the projection kernel's launch by their
        // 43 us, after the sign counts they run beside its start: 4.27 instead of 4.13 ms)
        HIP_TRY(hipStreamWaitEvent(P->st2, P->ev2[5], 0));   // element records (first stream) before the sign counts
        HIP_TRY(hipStreamWaitEvent(P->st3, P->ev2[5], 0));
        static const int no_inner_env = getenv("R2S_SIGN_NO_INNER") ? atoi(getenv("R2S_SIGN_NO_INNER")) : 0;
        hex_planes_kernel<<<(unsigned)((nel * 6 + 255) / 256), 256, 0, P->st3>>>(P->erec.as<ElemRec>(), nel,
                                                                                 (no_inner_env || prm.sign_no_inner) ? 1 : 0);
        HIP_TRY(hipEventRecord(P->ev2[5], P->st3));   // from here on: "planes done"
    }
    // HEX8 sign pass: which tiles are hot, how long their candidate lists get and the boxes of the inverse maps depend on
    // the element records only - counted here on the second stream, beside the boundary-face test and the work items of
    // the first, instead of in the binning phase (0.12 ms off the chain fill -> inverse maps -> gathers)
    constexpr bool HEX = std::is_same<typename ET::Rec, ElemRec>::value;
    const double rmax_needed = HEX ? -INFINITY : rho_t - 1e-12 * (fabs(rho_t) + 1.0);   // (TET4: see sign_bin_kernel)
    const bool sign_items = HEX && want_sign && nel > 0;
    ReadBack rb2;   // second read-back point (after the count passes)
    // (host order matters: ~7 us per launch.  Issued after the first launches of the first stream's chain instead, these
    //  delay item_build and with it the projection kernel: 4.25 instead of 4.13 ms)
    if constexpr (HEX) {
        if (sign_items) {
            zero_many(P->st2, {{P->sign_cnt.p, sizeof(uint32_t) * (size_t)(ntiles + 1)}, {P->hot.p, (size_t)ntiles + 1}});
            sign_hot_kernel<ElemRec><<<(unsigned)((nel * BIN_LANES + 255) / 256), 256, 0, P->st2>>>(P->erec.as<ElemRec>(), (uint32_t)nel, g, s, rho_t, P->hot.as<uint8_t>());
            sign_bin_kernel<ElemRec, false><<<(unsigned)((nel * BIN_LANES + 255) / 256), 256, 0, P->st2>>>(P->erec.as<ElemRec>(), (uint32_t)nel, g, s, P->sign_cnt.as<uint32_t>(), nullptr, nullptr, P->hot.as<uint8_t>(), abort_flag, rmax_needed);
            int rc = scan_exclusive2(P, P->sign_cnt.as<uint32_t>(), P->sign_off.as<uint32_t>(), nullptr, nullptr, (int64_t)ntiles + 1, P->st2, 0, 1);
            if (rc) return rc;
            // boxes of the elements that can be candidates in a hot tile (hot[] is complete after the count pass)
            ENSURE(P->sbox, sizeof(SignBox) * (size_t)nel);
            ENSURE(P->s_nchunks, sizeof(uint32_t) * (size_t)(nel + 1));
            ENSURE(P->s_chunk_off, sizeof(uint32_t) * (size_t)(nel + 1));
            ENSURE(P->s_nstore, sizeof(uint32_t) * (size_t)(nel + 1));
            ENSURE(P->s_store_off, sizeof(uint32_t) * (size_t)(nel + 1));
            zero_many(P->st2, {{P->s_nchunks.p, sizeof(uint32_t) * (size_t)(nel + 1)}, {P->s_nstore.p, sizeof(uint32_t) * (size_t)(nel + 1)}});
            sign_box_kernel<<<(unsigned)((nel * BIN_LANES + 255) / 256), 256, 0, P->st2>>>(P->erec.as<ElemRec>(), (uint32_t)nel, g, s, P->hot.as<uint8_t>(), P->sbox.as<SignBox>(), P->s_nchunks.as<uint32_t>(), P->s_nstore.as<uint32_t>());
            rc = scan_exclusive2(P, P->s_nchunks.as<uint32_t>(), P->s_chunk_off.as<uint32_t>(), P->s_nstore.as<uint32_t>(),
                                 P->s_store_off.as<uint32_t>(), nel + 1, P->st2, 0, 1);
            if (rc) return rc;
            rb2.add(P->s_chunk_off.as<uint32_t>() + nel, 1, 11);
            rb2.add(P->s_store_off.as<uint32_t>() + nel, 1, 13);
            HIP_TRY(hipEventRecord(P->ev2[7], P->st2));   // sign counts, offsets and boxes complete
        }
    }
    HIP_TRY(hipStreamWaitEvent(st, P->ev2[1], 0));
    face_mask_kernel<ET><<<(unsigned)((nel * ET::NES + 255) / 256), 256, 0, st>>>(
        dIEN, nel, P->ine_ptr.as<uint32_t>(), P->ine.as<uint32_t>(), P->cls.as<uint8_t>(), P->fmask.as<uint32_t>(),
        P->nitems.as<uint32_t>(), (const int*)counters);
    {
        int rc = scan_exclusive(P, P->nitems.as<uint32_t>(), P->item_off.as<uint32_t>(), nel + 1, st);
        if (rc) return rc;
    }
    {
        ReadBack rb;
        rb.add(P->item_off.as<uint32_t>() + nel, 1, 0);
        rb.add(counters, 1, 1);
        read_back_kernel<<<1, 64, 0, st>>>(rb, P->d_pinned, ex, counters + 15);
    }
    if (!spec) {
        HIP_TRY(hipStreamSynchronize(st));
        if (P->h_pinned[1]) return fail(R2S_ERR_ARG, "IEN contains node ids outside 1..nnp");
        cnt[0] = P->h_pinned[0];
        cnt[1] = 0;
    }
    const uint32_t n_items = want_dist ? cnt[0] : 0;
    ENSURE(P->items, sizeof(BandItem) * (size_t)std::max<uint32_t>(n_items, 1));
    ENSURE(P->nchunks, sizeof(uint32_t) * (size_t)(n_items + 1));
    ENSURE(P->chunk_off, sizeof(uint32_t) * (size_t)(n_items + 1));
    ENSURE(P->nstore, sizeof(uint32_t) * (size_t)(n_items + 1));
    ENSURE(P->store_off, sizeof(uint32_t) * (size_t)(n_items + 1));
    if (n_items) {
        ENSURE(P->hardflag, (size_t)n_items + 1);
        zero_many(st, {{P->nchunks.p, sizeof(uint32_t) * (size_t)(n_items + 1)}, {P->nstore.p, sizeof(uint32_t) * (size_t)(n_items + 1)},
                       {P->hardflag.p, (size_t)n_items + 1}});
        item_build_kernel<ET><<<(unsigned)((nel + 63) / 64), 64, 0, st>>>(
            P->erec.as<typename ET::Rec>(), P->cls.as<uint8_t>(), P->fmask.as<uint32_t>(), P->item_off.as<uint32_t>(), nel,
            g, s, delta, P->items.as<BandItem>(), P->nchunks.as<uint32_t>(), P->nstore.as<uint32_t>(), P->hardflag.as<uint8_t>(), rho_t, abort_flag);
        const uint32_t* work_counts = P->nchunks.as<uint32_t>();
        if constexpr (std::is_same<typename ET::Rec, ElemRec>::value) {
            ENSURE(P->perm, sizeof(uint32_t) * (size_t)n_items);
            ENSURE(P->wchunks, sizeof(uint32_t) * (size_t)(n_items + 1));
            work_order_kernel<<<1, 1024, 0, st>>>(P->hardflag.as<uint8_t>(), P->nchunks.as<uint32_t>(), n_items,
                                                  P->perm.as<uint32_t>(), P->wchunks.as<uint32_t>());
            work_counts = P->wchunks.as<uint32_t>();   // chunk_off then runs in work order (HEX8)
        }
        int rc = scan_exclusive2(P, work_counts, P->chunk_off.as<uint32_t>(), P->nstore.as<uint32_t>(),
                                 P->store_off.as<uint32_t>(), (int64_t)n_items + 1, st);
        if (rc) return rc;
        item_chunks_kernel<<<(n_items + 255) / 256, 256, 0, st>>>(P->items.as<BandItem>(), P->chunk_off.as<uint32_t>(), P->store_off.as<uint32_t>(), n_items);
        rb2.add(P->chunk_off.as<uint32_t>() + n_items, 1, 10);
        rb2.add(P->store_off.as<uint32_t>() + n_items, 1, 12);
    }
    HIP_TRY(hipEventRecord(P->ev[1], st));

    // ---- HEX8, speculated sizes: the projection kernel starts NOW, beside the binning ----
    // It only needs the work items; the tile lists (0.27 ms of ~25 short, latency-bound kernels that leave the GPU almost
    // idle) are for the inverse maps and the gathers.  With the sizes of the previous call at hand nothing has to wait for
    // a read-back: the chunk / storage totals are compared on the device first (abort flag, as for every speculated size),
    // the projection kernel is launched with 2 wavefronts per SIMD (336 of 512 VGPRs: the short kernels, the sweep and
    // sign_project find room beside it; 3 per SIMD would lock them out for 3 ms), and the binning moves to the second
    // stream, in front of the sweep.  R2S_NO_EARLY_ISO=1: the old order.
    static const bool overlap_env0 = !(getenv("R2S_NO_OVERLAP") && atoi(getenv("R2S_NO_OVERLAP")));
    static const bool early_env = !(getenv("R2S_NO_EARLY_ISO") && atoi(getenv("R2S_NO_EARLY_ISO")));
    bool early_iso = false;
    if constexpr (std::is_same<typename ET::Rec, ElemRec>::value) {
        early_iso = spec && early_env && overlap_env0 && want_sign && want_dist && nel > 0 && n_items && cnt[10] && cnt[11] && !(mode & R2S_OUT_XP);
        if (early_iso) {
            ReadBack rbA;
            rbA.add(P->chunk_off.as<uint32_t>() + n_items, 1, 10);
            rbA.add(P->store_off.as<uint32_t>() + n_items, 1, 12);
            read_back_kernel<<<1, 64, 0, st>>>(rbA, P->d_pinned, ex, counters + 15);
            HIP_TRY(hipEventRecord(P->ev2[6], st));   // work items complete: the binning (second stream) may start
            const uint32_t n_chunks_e = cnt[10], n_store_e = cnt[12];
            ENSURE(P->iso_res, sizeof(double) * 64 * (size_t)std::max<uint32_t>(n_store_e, 1));
            static const int wps_env = getenv("R2S_ISO_WPS") ? atoi(getenv("R2S_ISO_WPS")) : 0;   // tuning knob
            const uint32_t wps = (wps_env >= 1 && wps_env <= 3) ? (uint32_t)wps_env : 2u;
            // (the sign counts of the second stream first: when the persistent wavefronts arrive in the middle of those
            //  kernels, one of them is left waiting until the projection kernel drains - 2 of 3 runs, +0.25 ms)
            if (sign_items) HIP_TRY(hipStreamWaitEvent(st, P->ev2[7], 0));
            HIP_TRY(hipEventRecord(P->ev[7], st));
            { const int rc = iso_project_hex(P, st, n_items, n_chunks_e, n_store_e, wps, g, s, rho_t, nullptr, counters, abort_flag); if (rc) return rc; }
            HIP_TRY(hipEventRecord(P->ev[6], st));
            HIP_TRY(hipStreamWaitEvent(P->st2, P->ev2[6], 0));
            // (the sentinel sweep on a third stream right here, beside the binning: its wavefronts delay the first short
            //  kernels by what it saves later - 4.24 vs 4.20 ms; it stays behind the binning on the second stream)
        }
    } else {
        // TET4, fused output: the same - the closed-form projection kernel (short-lived wavefronts, 44 VGPRs) runs on
        // the first stream while the high-priority second stream builds the tile lists (2.8 ms on the 1 M elements of
        // config 5), sweeps and runs the sign gather
        early_iso = spec && early_env && overlap_env0 && mode == R2S_OUT_SDF && want_sign && want_dist && n_items && cnt[10];
        if (early_iso) {
            ReadBack rbA;
            rbA.add(P->chunk_off.as<uint32_t>() + n_items, 1, 10);
            rbA.add(P->store_off.as<uint32_t>() + n_items, 1, 12);
            read_back_kernel<<<1, 64, 0, st>>>(rbA, P->d_pinned, ex, counters + 15);
            HIP_TRY(hipEventRecord(P->ev2[6], st));
            const uint32_t n_chunks_e = cnt[10], n_store_e = cnt[12];
            ENSURE(P->iso_res, sizeof(double) * 64 * (size_t)std::max<uint32_t>(n_store_e, 1));
            static const int cpw_env = getenv("R2S_ISO_CPW") ? atoi(getenv("R2S_ISO_CPW")) : 0;   // tuning knob
            const uint32_t cpw = cpw_env > 0 ? (uint32_t)cpw_env : 8u;
            const uint32_t nwaves = (n_chunks_e + cpw - 1) / cpw;
            HIP_TRY(hipEventRecord(P->ev[7], st));
            iso_project_kernel<typename ET::Rec><<<(nwaves + 3) / 4, 256, 0, st>>>(
                P->items.as<BandItem>(), n_items, P->chunk_off.as<uint32_t>(), n_chunks_e, P->erec.as<typename ET::Rec>(), g,
                s, rho_t, P->iso_res.as<double>(), nullptr, abort_flag, cpw);
            HIP_TRY(hipEventRecord(P->ev[6], st));
            HIP_TRY(hipStreamWaitEvent(P->st2, P->ev2[6], 0));
        }
    }
    hipStream_t bs = early_iso ? P->st2 : st;   // stream of the binning

    // ---- tile bins ----
    if (sign_items) {   // (HEX8: the sign counts are on their way on the second stream, see above)
        zero_many(bs, {{P->band_cnt.p, sizeof(uint32_t) * (size_t)(ntiles + 1)}, {P->tri.p, (size_t)ntiles + 1}});
        if (n_items)
            band_bin_kernel<false><<<(n_items * BIN_LANES + 127) / 128, 128, 0, bs>>>(P->items.as<BandItem>(), n_items, g, s, P->band_cnt.as<uint32_t>(), nullptr, nullptr, P->tri.as<uint8_t>(), abort_flag);
        int rc = scan_exclusive2(P, P->band_cnt.as<uint32_t>(), P->band_off.as<uint32_t>(), nullptr, nullptr, (int64_t)ntiles + 1, bs);
        if (rc) return rc;
        HIP_TRY(hipStreamWaitEvent(bs, P->ev2[7], 0));
    } else {
        zero_many(bs, {{P->band_cnt.p, sizeof(uint32_t) * (size_t)(ntiles + 1)}, {P->sign_cnt.p, sizeof(uint32_t) * (size_t)(ntiles + 1)},
                       {P->hot.p, (size_t)ntiles + 1}, {P->tri.p, (size_t)ntiles + 1}});
        if (n_items)
            band_bin_kernel<false><<<(n_items * BIN_LANES + 127) / 128, 128, 0, bs>>>(P->items.as<BandItem>(), n_items, g, s, P->band_cnt.as<uint32_t>(), nullptr, nullptr, P->tri.as<uint8_t>(), abort_flag);
        if (want_sign)
        {
            sign_hot_kernel<typename ET::Rec><<<(unsigned)((nel * BIN_LANES + 255) / 256), 256, 0, bs>>>(P->erec.as<typename ET::Rec>(), (uint32_t)nel, g, s, rho_t, P->hot.as<uint8_t>());
            sign_bin_kernel<typename ET::Rec, false><<<(unsigned)((nel * BIN_LANES + 255) / 256), 256, 0, bs>>>(P->erec.as<typename ET::Rec>(), (uint32_t)nel, g, s, P->sign_cnt.as<uint32_t>(), nullptr, nullptr, P->hot.as<uint8_t>(), abort_flag, rmax_needed);
        }
        int rc = scan_exclusive2(P, P->band_cnt.as<uint32_t>(), P->band_off.as<uint32_t>(), P->sign_cnt.as<uint32_t>(),
                                 P->sign_off.as<uint32_t>(), (int64_t)ntiles + 1, bs);
        if (rc) return rc;
    }
    active_tiles_kernel<<<(ntiles + 256 * AT_ITEMS - 1) / (256 * AT_ITEMS), 256, 0, bs>>>(P->band_cnt.as<uint32_t>(), P->sign_cnt.as<uint32_t>(), P->hot.as<uint8_t>(), ntiles, P->active.as<uint32_t>(), P->active_sign.as<uint32_t>(), P->active_any.as<uint32_t>(), P->active_sonly.as<uint32_t>(), P->tri.as<uint8_t>(), P->active_lean.as<uint32_t>(), P->active_tri.as<uint32_t>(), counters);
    rb2.add(P->band_off.as<uint32_t>() + ntiles, 1, 2);
    rb2.add(P->sign_off.as<uint32_t>() + ntiles, 1, 3);
    rb2.add(counters + 1, 6, 4);
    rb2.add(counters + 10, 2, 14);
    read_back_kernel<<<1, 64, 0, bs>>>(rb2, P->d_pinned, ex, counters + 15);
    if (!spec) {
        HIP_TRY(hipStreamSynchronize(bs));
        for (int q = 2; q < 16; ++q) cnt[q] = P->h_pinned[q];
    }
    const uint32_t n_band = cnt[2], n_sign = cnt[3], n_active = cnt[4], n_active_sign = cnt[5];
    P->last_s = s; P->last_g = g; P->last_n_any = cnt[8]; P->last_n_band = cnt[4]; P->last_n_sonly = cnt[9]; P->has_last = true;
    const uint32_t n_chunks = n_items ? cnt[10] : 0;
    const uint32_t n_store = n_items ? cnt[12] : 0;     // storage chunks (4x4x4 tiles of the item boxes)
    ENSURE(P->iso_res, sizeof(double) * 64 * (size_t)std::max<uint32_t>(n_store, 1));
    const uint32_t n_schunks = sign_items ? cnt[11] : 0;
    const uint32_t n_sstore = sign_items ? cnt[13] : 0;
    if (sign_items) ENSURE(P->sres, sizeof(double) * 64 * (size_t)std::max<uint32_t>(n_sstore, 1));
    if (mode & R2S_OUT_XP) ENSURE(P->iso_res_xp, sizeof(double) * 192 * (size_t)std::max<uint32_t>(n_store, 1));
    ENSURE(P->band_raw, sizeof(uint32_t) * (size_t)std::max<uint32_t>(n_band, 1));
    ENSURE(P->band_ent, sizeof(uint32_t) * (size_t)std::max<uint32_t>(n_band, 1));
    ENSURE(P->sign_raw, sizeof(uint32_t) * (size_t)std::max<uint32_t>(n_sign, 1));
    ENSURE(P->sign_ent, sizeof(uint32_t) * (size_t)std::max<uint32_t>(n_sign, 1));
    zero_many(bs, {{P->band_cnt.p, sizeof(uint32_t) * (size_t)(ntiles + 1)}, {P->sign_cnt.p, sizeof(uint32_t) * (size_t)(ntiles + 1)}});
    if (n_items)
        band_bin_kernel<true><<<(n_items * BIN_LANES + 127) / 128, 128, 0, bs>>>(P->items.as<BandItem>(), n_items, g, s, P->band_cnt.as<uint32_t>(), P->band_off.as<uint32_t>(), P->band_raw.as<uint32_t>(), nullptr, abort_flag);
    if (want_sign)
        sign_bin_kernel<typename ET::Rec, true><<<(unsigned)((nel * BIN_LANES + 255) / 256), 256, 0, bs>>>(P->erec.as<typename ET::Rec>(), (uint32_t)nel, g, s, P->sign_cnt.as<uint32_t>(), P->sign_off.as<uint32_t>(), HEX ? P->sign_raw.as<uint32_t>() : P->sign_ent.as<uint32_t>(), P->hot.as<uint8_t>(), abort_flag, rmax_needed);
    // (TET4: a voxel is +1 when ANY candidate holds it with rho >= rho_t - "the first one" of SignDetection.jl:128-147
    // only ends the search - so the order of a list does not matter and the lists are used as filled)
    // lists longer than 64 entries only occur when the grid is coarse relative to the mesh (few tiles)
    if (n_active) {
        bin_sort_small_kernel<<<(n_active * SORT_LANES + 255) / 256, 256, 0, bs>>>(P->active.as<uint32_t>(), n_active, P->band_off.as<uint32_t>(), P->band_raw.as<uint32_t>(), P->band_ent.as<uint32_t>(), abort_flag);
        if (cnt[6] > 64u)
            bin_sort_kernel<<<(n_active + 3) / 4, 256, 0, bs>>>(P->active.as<uint32_t>(), n_active, P->band_off.as<uint32_t>(), P->band_raw.as<uint32_t>(), P->band_ent.as<uint32_t>(), abort_flag);
    }
    if (n_active_sign && HEX) {
        bin_sort_small_kernel<<<(n_active_sign * SORT_LANES + 255) / 256, 256, 0, bs>>>(P->active_sign.as<uint32_t>(), n_active_sign, P->sign_off.as<uint32_t>(), P->sign_raw.as<uint32_t>(), P->sign_ent.as<uint32_t>(), abort_flag);
        if (cnt[7] > 64u)
            bin_sort_kernel<<<(n_active_sign + 3) / 4, 256, 0, bs>>>(P->active_sign.as<uint32_t>(), n_active_sign, P->sign_off.as<uint32_t>(), P->sign_raw.as<uint32_t>(), P->sign_ent.as<uint32_t>(), abort_flag);
    }
    HIP_TRY(hipEventRecord(P->ev[2], bs));

    // ---- fork ----
    // HEX8, distance + sign wanted: the inverse maps of the sign pass (and the sign-only gather) run on a
    // second stream beside the iso-surface projection; the band gather waits for both.  TET4, fused output:
    // the in-gather sign pass runs on the second stream.
    static const bool overlap_env = !(getenv("R2S_NO_OVERLAP") && atoi(getenv("R2S_NO_OVERLAP")));
    const bool overlap = !HEX && overlap_env && mode == R2S_OUT_SDF && want_sign && want_dist;   // TET4 flow
    const bool fork = HEX && overlap_env && want_sign && want_dist && n_chunks && n_schunks;      // HEX8 flow
    // ---- sentinel sweep ----
    // HBM-bound and only needed by the gathers: with two streams it goes first on the second stream, beside
    // the start of the (FP64-bound) projection kernel, and the sign stages follow it there.  Measured
    // alternatives (north-star workload, ms/step): sweep at the start of the call beside prep + binning 5.9
    // (its waves delay the short kernels, even with a quarter-size grid); sign_project before the sweep 6.5
    // and sign_project starting together with the projection kernel 6.7 (the projection kernel is the long
    // pole and should get its persistent waves placed first); sweep + sign_project already after the second
    // read-back, beside the list fill / sort kernels 6.4 (those short kernels then wait for wave slots and the
    // projection kernel starts 1.2 ms later); inverse maps split into elements near the band (which the band
    // gather waits for) and the rest: no change - what is left after the projection kernel is work-bound; this
    // order 5.15.
    {
        const bool two_streams = overlap || fork;
        hipStream_t fs = two_streams ? P->st2 : st;
        const unsigned fill_grid = 256 * 8;
        if (two_streams) HIP_TRY(hipStreamWaitEvent(P->st2, P->ev[2], 0));
        HIP_TRY(hipEventRecord(P->ev2[0], fs));
        if (mode & R2S_OUT_DIST) fill_kernel<<<fill_grid, 256, 0, fs>>>(d_dist, nvox, 1.0e10);
        if (mode & R2S_OUT_SIGN) fill_kernel<<<fill_grid, 256, 0, fs>>>(d_sign, nvox, -1.0);
        if (mode & R2S_OUT_SDF) fill_kernel<<<fill_grid, 256, 0, fs>>>(d_sdf, nvox, -1.0e10);
        if (mode & R2S_OUT_XP) HIP_TRY(hipMemsetAsync(d_xp, 0, sizeof(double) * 3 * (size_t)nvox, fs));
        HIP_TRY(hipEventRecord(P->ev2[3], fs));
        HIP_TRY(hipEventRecord(P->ev[3], st));
        if (two_streams) HIP_TRY(hipEventRecord(P->ev2[1], P->st2));
    }

    // ---- projection / sign kernel over the active tiles ----
    {
        MainArgs A;
        A.g = g; A.s = s; A.rho_t = rho_t;
        A.true_min = prm.true_min ? 1 : 0;
        A.band_off = P->band_off.as<uint32_t>(); A.band_ent = P->band_ent.as<uint32_t>();
        A.sign_off = P->sign_off.as<uint32_t>(); A.sign_ent = P->sign_ent.as<uint32_t>();
        A.items = P->items.as<BandItem>(); A.erec = P->erec.as<typename ET::Rec>();
        A.dist = (mode & R2S_OUT_DIST) ? d_dist : nullptr;
        A.sign = (mode & R2S_OUT_SIGN) ? d_sign : nullptr;
        A.sdf = (mode & R2S_OUT_SDF) ? d_sdf : nullptr;
        A.sbox = P->sbox.p; A.s_store_off = P->s_store_off.as<uint32_t>(); A.sres = P->sres.as<double>();
        A.hot = P->hot.as<uint8_t>();
        A.abort_flag = abort_flag;
        if constexpr (HEX) {
            // inverse maps of the sign pass (item-major), beside the iso-surface projection when both run
            hipStream_t ss = fork ? P->st2 : st;
            if (!fork) HIP_TRY(hipEventRecord(P->ev2[1], ss));
            if (want_sign && n_schunks) {
                const uint32_t cpw = 8, nwaves = (n_schunks + cpw - 1) / cpw;   // chunks per wavefront
                HIP_TRY(hipStreamWaitEvent(ss, P->ev2[5], 0));   // bounding half-spaces (hex_planes_kernel, second stream)
                sign_project_kernel<<<nwaves, 64, 0, ss>>>(P->sbox.as<SignBox>(), (uint32_t)nel, P->s_chunk_off.as<uint32_t>(), n_schunks, P->s_store_off.as<uint32_t>(), cpw,
                                                          P->erec.as<ElemRec>(), g, s, rho_t, P->hot.as<uint8_t>(), P->sres.as<double>(), abort_flag);
            }
            HIP_TRY(hipEventRecord(P->ev2[4], ss));   // inverse maps done: all the band gather waits for
            A.xp = (mode & R2S_OUT_XP) ? d_xp : nullptr;
            A.sdf_mode = 1;
            const uint32_t n_sonly = cnt[9];
            const bool early_sign_tiles = want_dist && want_sign && !(mode & R2S_OUT_XP);
            if (early_sign_tiles && n_sonly) {
                // tiles without band items only need the inverse maps: gathered right behind sign_project,
                // beside the iso-surface projection
                MainArgs B = A;
                B.active = P->active_sonly.as<uint32_t>(); B.n_active = n_sonly;
                sdf_tiles_kernel<ElemRec, false, true><<<n_sonly, 64, 0, ss>>>(B);
            }
            HIP_TRY(hipEventRecord(P->ev2[2], ss));
            A.iso_res = P->iso_res.as<double>();
            A.iso_res_xp = (mode & R2S_OUT_XP) ? P->iso_res_xp.as<double>() : nullptr;
            if (!early_iso) HIP_TRY(hipEventRecord(P->ev[7], st));
            if (want_dist && n_chunks && !early_iso) {
                static const int wps_env = getenv("R2S_ISO_WPS") ? atoi(getenv("R2S_ISO_WPS")) : 0;   // tuning knob
                // (2 waves/SIMD, or 3 on only part of the SIMDs so that sign_project finds room beside them from the
                // start: measured, no gain - the two kernels together are bound by their FP64 work either way)
                const uint32_t wps = (wps_env >= 1 && wps_env <= 3) ? (uint32_t)wps_env : 3u;
                // (counters[8], the chunk counter, is zero since the start of the call)
                // the sweep has to get its wavefronts placed before this kernel fills every SIMD for 3 ms (it is released
                // once the second stream has reached the sweep; high priority does the rest)
                if (fork) HIP_TRY(hipStreamWaitEvent(st, P->ev2[0], 0));
                double* const res_xp = (mode & R2S_OUT_XP) ? P->iso_res_xp.as<double>() : nullptr;
                { const int rc = iso_project_hex(P, st, n_items, n_chunks, n_store, wps, g, s, rho_t, res_xp, counters, abort_flag); if (rc) return rc; }
            }
            if (!early_iso) HIP_TRY(hipEventRecord(P->ev[6], st));
            if (fork) HIP_TRY(hipStreamWaitEvent(st, P->ev2[4], 0));   // (the sweep precedes it on that stream)
            // ordered per-voxel gather: band items (distance) and candidate elements (sign) of every tile
            if (want_dist && want_sign && early_sign_tiles) {
                // band tiles (the others are done): lean kernel where the lists hold no boundary triangles
                const uint32_t n_lean = cnt[14], n_tri = cnt[15];
                if (n_lean) {
                    A.active = P->active_lean.as<uint32_t>(); A.n_active = n_lean;
                    sdf_tiles_kernel<ElemRec, true, true, false><<<n_lean, 64, 0, st>>>(A);   // one-wave workgroups: ~10 % faster
                }
                if (n_tri) {
                    A.active = P->active_tri.as<uint32_t>(); A.n_active = n_tri;
                    if (A.true_min) sdf_tiles_kernel<ElemRec, true, true, true, true><<<(n_tri + 3) / 4, 256, 0, st>>>(A);
                    else sdf_tiles_kernel<ElemRec, true, true, true><<<(n_tri + 3) / 4, 256, 0, st>>>(A);
                }
            } else if (want_dist && want_sign) {
                const uint32_t n_any = cnt[8];
                A.active = P->active_any.as<uint32_t>(); A.n_active = n_any;
                if (n_any && A.true_min) sdf_tiles_kernel<ElemRec, true, true, true, true><<<(n_any + 3) / 4, 256, 0, st>>>(A);
                else if (n_any) sdf_tiles_kernel<ElemRec, true, true><<<(n_any + 3) / 4, 256, 0, st>>>(A);
            } else if (want_dist) {
                A.active = P->active.as<uint32_t>(); A.n_active = n_active;
                if (n_active && A.true_min) sdf_tiles_kernel<ElemRec, true, false, true, true><<<(n_active + 3) / 4, 256, 0, st>>>(A);
                else if (n_active) sdf_tiles_kernel<ElemRec, true, false><<<(n_active + 3) / 4, 256, 0, st>>>(A);
            } else if (want_sign) {
                A.active = P->active_sign.as<uint32_t>(); A.n_active = n_active_sign;
                if (n_active_sign) sdf_tiles_kernel<ElemRec, false, true><<<(n_active_sign + 3) / 4, 256, 0, st>>>(A);
            }
            if (fork) HIP_TRY(hipStreamWaitEvent(st, P->ev2[2], 0));   // sign-only gather (second stream), beside the band gather
            HIP_TRY(hipEventRecord(P->ev[4], st));
        } else {
            if (!early_iso) HIP_TRY(hipEventRecord(P->ev[7], st));
            if (overlap) {
                if (n_active_sign) {
                    MainArgs B = A;
                    B.active = P->active_sign.as<uint32_t>(); B.n_active = n_active_sign;
                    B.dist = nullptr; B.xp = nullptr; B.sign = nullptr;
                    B.sdf_mode = 3;
                    // one-wave workgroups: they take freed wave slots as readily as the projection kernel's
                    sdf_tiles_kernel<typename ET::Rec, false, true><<<n_active_sign, 64, 0, P->st2>>>(B);
                }
                HIP_TRY(hipEventRecord(P->ev2[2], P->st2));
            }
            // item-major iso-surface projections, then the ordered gather over the band tiles
            A.iso_res = P->iso_res.as<double>();
            A.iso_res_xp = (mode & R2S_OUT_XP) ? P->iso_res_xp.as<double>() : nullptr;
            if (want_dist && n_chunks && !early_iso)
            {
                static const int cpw_env = getenv("R2S_ISO_CPW") ? atoi(getenv("R2S_ISO_CPW")) : 0;   // tuning knob
                const uint32_t cpw = cpw_env > 0 ? (uint32_t)cpw_env : 8u;
                const uint32_t nwaves = (n_chunks + cpw - 1) / cpw;
                iso_project_kernel<typename ET::Rec><<<(nwaves + 3) / 4, 256, 0, st>>>(
                    P->items.as<BandItem>(), n_items, P->chunk_off.as<uint32_t>(), n_chunks, P->erec.as<typename ET::Rec>(), g,
                    s, rho_t, P->iso_res.as<double>(), (mode & R2S_OUT_XP) ? P->iso_res_xp.as<double>() : nullptr, abort_flag, cpw);
            }
            if (!early_iso) HIP_TRY(hipEventRecord(P->ev[6], st));
            if (want_dist && n_active) {
                A.active = P->active.as<uint32_t>(); A.n_active = n_active;
                A.sign = nullptr;
                A.xp = (mode & R2S_OUT_XP) ? d_xp : nullptr;
                A.sdf_mode = overlap ? 4 : 2;
                if (overlap) HIP_TRY(hipStreamWaitEvent(st, P->ev2[2], 0));
                if (A.true_min) sdf_tiles_kernel<typename ET::Rec, true, false, true, true><<<(n_active + 3) / 4, 256, 0, st>>>(A);
                else sdf_tiles_kernel<typename ET::Rec, true, false><<<(n_active + 3) / 4, 256, 0, st>>>(A);
            } else if (overlap) {
                HIP_TRY(hipStreamWaitEvent(st, P->ev2[2], 0));
            }
            HIP_TRY(hipEventRecord(P->ev[4], st));
            // sign pass over the tiles whose candidate elements reach rho_t
            if (!overlap && want_sign && n_active_sign) {
                A.active = P->active_sign.as<uint32_t>(); A.n_active = n_active_sign;
                A.dist = nullptr; A.xp = nullptr;
                A.sign = (mode & R2S_OUT_SIGN) ? d_sign : nullptr;
                A.sdf_mode = 3;
                sdf_tiles_kernel<typename ET::Rec, false, true><<<(n_active_sign + 3) / 4, 256, 0, st>>>(A);
            }
        }
    }
    HIP_TRY(hipStreamWaitEvent(st, P->ev2[5], 0));   // nothing of this call is left on the second stream
    HIP_TRY(hipEventRecord(P->ev[5], st));
    {
        ReadBack rb3;   // the abort flag travels with the sizes
        rb3.add(counters + 15, 1, 16);
        rb3.add(counters + 16, 1, 17);   // [16] runs of the complete solver that ended without a KKT point
        rb3.add(counters + 12, 1, 18);   // [12] pairs the fast lane machine handed over
        SpecExpect none;
        memset(&none, 0, sizeof none);
        read_back_kernel<<<1, 64, 0, st>>>(rb3, P->d_pinned, none, counters + 15);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    if (spec) {
        if (P->h_pinned[1]) {
            P->spec.valid = false;
            return fail(R2S_ERR_ARG, "IEN contains node ids outside 1..nnp");
        }
        bool same = P->h_pinned[16] == 0u;
        for (int q = 0; q < 16 && same; ++q)
            if (q != 1 && P->h_pinned[q] != cnt[q]) same = false;
        if (!same) {
            // the sizes of this call differ from the remembered ones (another mesh / density / threshold behind the
            // same shapes): nothing was written outside the buffers, the outputs are not valid - once more, waiting
            P->spec.valid = false;
            return run_impl<ET>(P, dX, nnp, dIEN, nel, d_rho_n, rho_t, grid, params, k_begin, k_end, mode, d_dist, d_sign,
                                d_sdf, d_xp, stream, stats, false);
        }
    } else {
        memcpy(P->spec.key, key, sizeof key);
        for (int q = 0; q < 16; ++q) P->spec.v[q] = P->h_pinned[q];
        P->spec.v[1] = 0;
        P->spec.valid = true;
    }

    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_items = n_items;
        stats->n_band_entries = n_band;
        stats->n_sign_entries = n_sign;
        stats->n_tiles = ntiles;
        stats->n_active_tiles = n_active;
        stats->n_active_sign_tiles = n_active_sign;
        stats->n_any_tiles = P->last_n_any;
        stats->n_sign_only_tiles = P->last_n_sonly;
        stats->n_iso_fail = P->h_pinned[17];
        stats->n_iso_straggler = P->h_pinned[18];
        float ms = 0;
        if (hipEventElapsedTime(&ms, P->ev[0], P->ev[1]) == hipSuccess) stats->ms_prep = ms;
        if (hipEventElapsedTime(&ms, P->ev[1], P->ev[2]) == hipSuccess) stats->ms_bins = ms;
        if (hipEventElapsedTime(&ms, P->ev2[0], P->ev2[3]) == hipSuccess) stats->ms_fill = ms;   // second stream: beside the projection kernel
        if (hipEventElapsedTime(&ms, P->ev[7], P->ev[6]) == hipSuccess) stats->ms_main = ms;
        if (P->fast_timed && hipEventElapsedTime(&ms, P->ev_fast[0], P->ev_fast[1]) == hipSuccess) stats->ms_iso_fast = ms;
        if (hipEventElapsedTime(&ms, P->ev[6], P->ev[4]) == hipSuccess) stats->ms_gather = ms;
        stats->n_iso_chunks = n_chunks;
        if (hipEventElapsedTime(&ms, P->ev[4], P->ev[5]) == hipSuccess) stats->ms_sign = ms;
        if (overlap || HEX) {   // sign stage on its own events (second stream: it overlaps ms_main)
            if (hipEventElapsedTime(&ms, P->ev2[1], P->ev2[2]) == hipSuccess) stats->ms_sign = ms;
        }
        (void)hipGetLastError();
    }
    return 0;
}

extern "C" {

int r2s_plan_run_dev(r2s_plan* P, const double* dX, int64_t nnp, const int64_t* dIEN, int64_t nel,
                     const double* d_rho_n, double rho_t, const r2s_grid* grid, const r2s_params* params,
                     int64_t k_begin, int64_t k_end, int32_t mode, double* d_dist, double* d_sign,
                     double* d_sdf, double* d_xp, void* stream, r2s_stats* stats)
{
    const int et = params ? params->elem_type : R2S_HEX8;
    if (et == R2S_HEX8)
        return run_impl<HexT>(P, dX, nnp, dIEN, nel, d_rho_n, rho_t, grid, params, k_begin, k_end, mode, d_dist,
                              d_sign, d_sdf, d_xp, stream, stats);
    if (et == R2S_TET4)
        return run_impl<TetT>(P, dX, nnp, dIEN, nel, d_rho_n, rho_t, grid, params, k_begin, k_end, mode, d_dist,
                              d_sign, d_sdf, d_xp, stream, stats);
    return fail(R2S_ERR_UNSUPPORTED, "unknown element type %d", et);
}

int r2s_plan_pack_tiles_dev(r2s_plan* P, const double* d_local, double* d_payload, uint32_t* d_ids,
                            int64_t capacity_tiles, int64_t* n_tiles_out, void* stream)
{
    if (!P || !P->has_last) return fail(R2S_ERR_ARG, "r2s_plan_pack_tiles_dev: no previous r2s_plan_run_dev on this plan");
    const uint32_t n = P->last_n_any;
    if (n_tiles_out) *n_tiles_out = n;
    if (!d_local || !d_payload || !d_ids) return fail(R2S_ERR_ARG, "null argument");
    if ((int64_t)n > capacity_tiles) return fail(R2S_ERR_ARG, "payload capacity %lld < %u tiles", (long long)capacity_tiles, n);
    if ((P->last_s.k0 & 3) != 0) return fail(R2S_ERR_ARG, "tile packing needs a Z partition aligned to 4-plane tile layers");
    HIP_TRY(hipSetDevice(P->device));
    if (n)
        pack_tiles_kernel<<<(n + 3) / 4, 256, 0, (hipStream_t)stream>>>(P->active_any.as<uint32_t>(), n, P->last_s, P->last_g,
                                                                      d_local, -1.0e10, d_payload, d_ids);
    HIP_TRY(hipGetLastError());
    return 0;
}

int r2s_unpack_tiles_dev(const double* d_payload, const uint32_t* d_ids, int64_t n_tiles, const r2s_grid* grid,
                         double* d_volume, void* stream)
{
    if (!grid || !d_volume || (n_tiles > 0 && (!d_payload || !d_ids)) || n_tiles < 0 || n_tiles >= ((int64_t)1 << 31))
        return fail(R2S_ERR_ARG, "r2s_unpack_tiles_dev: bad argument");
    if (n_tiles)
        unpack_tiles_kernel<<<(unsigned)((n_tiles + 3) / 4), 256, 0, (hipStream_t)stream>>>(
            d_payload, d_ids, (uint32_t)n_tiles, (int)grid->N[0] + 1, (int)grid->N[1] + 1, (int)grid->N[2] + 1, d_volume);
    HIP_TRY(hipGetLastError());
    return 0;
}

int r2s_plan_pack_tiles2_dev(r2s_plan* P, const double* d_local, double* d_payload, uint32_t* d_ids,
                             int64_t capacity_full, uint64_t* d_masks, uint32_t* d_mask_ids, int64_t capacity_mask,
                             int64_t* n_full_out, int64_t* n_mask_out, void* stream)
{
    if (!P || !P->has_last) return fail(R2S_ERR_ARG, "r2s_plan_pack_tiles2_dev: no previous r2s_plan_run_dev on this plan");
    const uint32_t nf = P->last_n_band, nm = P->last_n_sonly;
    if (n_full_out) *n_full_out = nf;
    if (n_mask_out) *n_mask_out = nm;
    if (!d_local || (nf && (!d_payload || !d_ids)) || (nm && (!d_masks || !d_mask_ids))) return fail(R2S_ERR_ARG, "null argument");
    if ((int64_t)nf > capacity_full || (int64_t)nm > capacity_mask)
        return fail(R2S_ERR_ARG, "capacity (%lld, %lld) < (%u, %u) tiles", (long long)capacity_full, (long long)capacity_mask, nf, nm);
    if ((P->last_s.k0 & 3) != 0) return fail(R2S_ERR_ARG, "tile packing needs a Z partition aligned to 4-plane tile layers");
    HIP_TRY(hipSetDevice(P->device));
    if (nf)
        pack_tiles_kernel<<<(nf + 3) / 4, 256, 0, (hipStream_t)stream>>>(P->active.as<uint32_t>(), nf, P->last_s, P->last_g, d_local,
                                                                       -1.0e10, d_payload, d_ids);
    if (nm)
        pack_masks_kernel<<<(nm + 3) / 4, 256, 0, (hipStream_t)stream>>>(P->active_sonly.as<uint32_t>(), nm, P->last_s, P->last_g,
                                                                       d_local, d_masks, d_mask_ids);
    HIP_TRY(hipGetLastError());
    return 0;
}

int r2s_unpack_masks_dev(const uint64_t* d_masks, const uint32_t* d_mask_ids, int64_t n_tiles, const r2s_grid* grid,
                         double magnitude, double* d_volume, void* stream)
{
    if (!grid || !d_volume || (n_tiles > 0 && (!d_masks || !d_mask_ids)) || n_tiles < 0 || n_tiles >= ((int64_t)1 << 31))
        return fail(R2S_ERR_ARG, "r2s_unpack_masks_dev: bad argument");
    if (n_tiles)
        unpack_masks_kernel<<<(unsigned)((n_tiles + 3) / 4), 256, 0, (hipStream_t)stream>>>(
            d_masks, d_mask_ids, (uint32_t)n_tiles, (int)grid->N[0] + 1, (int)grid->N[1] + 1, (int)grid->N[2] + 1, magnitude,
            d_volume);
    HIP_TRY(hipGetLastError());
    return 0;
}

int r2s_unpack_segments_dev(const double* d_buf, int32_t world, int64_t seglen, int64_t cap_full, int64_t cap_mask,
                            const r2s_grid* grid, double magnitude, double* d_volume, void* stream)
{
    if (!d_buf || !grid || !d_volume || world < 1 || world > 65535 || cap_full < 0 || cap_mask < 0 ||
        cap_full >= ((int64_t)1 << 31) || cap_mask >= ((int64_t)1 << 31))
        return fail(R2S_ERR_ARG, "r2s_unpack_segments_dev: bad argument");
    const int64_t need = 2 + cap_full * 64 + (cap_full + 1) / 2 + cap_mask + (cap_mask + 1) / 2;
    if (seglen < need) return fail(R2S_ERR_ARG, "segment length %lld < %lld for capacities (%lld, %lld)", (long long)seglen, (long long)need,
                                   (long long)cap_full, (long long)cap_mask);
    const int nx = (int)grid->N[0] + 1, ny = (int)grid->N[1] + 1, nz = (int)grid->N[2] + 1;
    if (cap_full)
        unpack_segments_tiles_kernel<<<dim3((unsigned)((cap_full + 3) / 4), (unsigned)world), 256, 0, (hipStream_t)stream>>>(
            d_buf, seglen, (uint32_t)cap_full, nx, ny, nz, d_volume);
    if (cap_mask)
        unpack_segments_masks_kernel<<<dim3((unsigned)((cap_mask + 3) / 4), (unsigned)world), 256, 0, (hipStream_t)stream>>>(
            d_buf, seglen, (uint32_t)cap_full, (uint32_t)cap_mask, nx, ny, nz, magnitude, d_volume);
    HIP_TRY(hipGetLastError());
    return 0;
}

int r2s_fill_dev(double* d, int64_t n, double value, void* stream)
{
    if (!d || n < 0) return fail(R2S_ERR_ARG, "r2s_fill_dev: bad argument");
    if (n) fill_kernel<<<256 * 8, 256, 0, (hipStream_t)stream>>>(d, n, value);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
